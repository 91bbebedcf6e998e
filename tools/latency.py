#!/usr/bin/env python3
"""Single-trajectory (batch 1) latency of the default configuration for every horizon length of the
reference's table (include/common/settings.cuh:123-139): kernel time by HIP events, SS preconditioner,
lambda0 = 0, exit_tol 1e-4.   python tools/latency.py > gpurun_out/latency_b1.txt"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth

for N in (32, 64, 128, 256, 512):
    sol = PcgSolver(N, max_batch=1)
    dS, dP, dg = bench.build_inputs(sol, N, 1, 0, "ss", torch.device("cuda", 0))
    lam = torch.zeros(1, 14 * N, device="cuda")
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    ts = []
    for i in range(40):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it, ex = sol.solve(dS, dP, dg, lam, cfg)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    it = int(it.item())
    ms = float(np.median(ts[5:]))
    w, rt = sol.get_option("pcg_waves"), sol.get_option("pcg_reg_rows")
    print(f"N={N:3d} batch=1 waves={w} reg_triples={rt} iters={it} kernel_us={ms * 1e3:.1f} us_per_iter={ms * 1e3 / it:.2f}", flush=True)
