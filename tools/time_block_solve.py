#!/usr/bin/env python3
"""mpcg_block_solve (batched block-tridiagonal direct solve) against mpcg_pcg_solve on the same systems."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
for N, B in ((32, 1), (32, 4096), (128, 1), (128, 256), (128, 512), (128, 1024), (128, 2048), (128, 4096), (512, 1), (512, 1024)):
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    lam = torch.zeros(B, 14 * N, device="cuda")
    def t(fn, reps=6):
        ts = []
        for i in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return float(np.median(ts[1:]))
    sol.set_option("block_solve_wide", 0)
    ms_n = t(lambda: sol.block_solve(dS, dg, lam))
    sol.set_option("block_solve_wide", 1)
    ms_w = t(lambda: sol.block_solve(dS, dg, lam))
    sol.set_option("block_solve_wide", -1)
    ms_d = t(lambda: sol.block_solve(dS, dg, lam))
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    def pcg():
        lam.zero_(); sol.solve(dS, dP, dg, lam, cfg)
    ms_p = t(pcg)
    print(f"N={N:3d} batch={B:4d}: [4 traj/wave {ms_n:7.3f} ms | 1 traj/wave {ms_w:7.3f} ms] block_solve {ms_d:8.3f} ms = {B / ms_d * 1e3:10.0f} linsolves/s ({ms_d * 1e3 / B:7.2f} us each) | "
          f"pcg ({synth.pcg_max_iter(N)} it cap, cold) {ms_p:8.3f} ms = {B / ms_p * 1e3:10.0f} linsolves/s", flush=True)
