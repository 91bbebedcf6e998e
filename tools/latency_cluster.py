#!/usr/bin/env python3
"""Cluster kernel (G workgroups per trajectory) vs single-workgroup kernel: kernel time per solve,
SS, lambda0 = 0, max_iter of the reference table.  python tools/latency_cluster.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth

def run(N, B, G):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", G)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    lam = torch.zeros(B, 14 * N, device="cuda")
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    ts = []
    for i in range(25):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    itn = it.cpu().numpy().astype(np.int64)
    ms = float(np.median(ts[5:]))
    ok = "ok" if (itn < 1 << 30).all() else "TIMEOUT"
    print(f"N={N:3d} batch={B:3d} G={G:2d}: {ms*1e3:8.1f} us/solve-batch  {ms*1e3/itn.mean():6.2f} us/iter  "
          f"{itn.sum()/ms/1e3:7.3f} Miter/s  iters={int(itn.mean())} {ok}", flush=True)

for N, B, Gs in ((64, 1, (0, 2)), (128, 1, (0, 2, 3, 4)), (256, 1, (0, 2, 4, 8)), (512, 1, (0, 4, 8, 16)),
                 (256, 64, (0, 4)), (512, 32, (0, 8)), (512, 64, (0, 4)), (128, 128, (0, 2))):
    for G in Gs:
        run(N, B, G)
