#!/usr/bin/env python3
"""Sweep bt_spmv launch knobs on 1024 x N=128 (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, synth
N, B = 128, 1024
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
x = torch.randn(B, 14 * N, device="cuda"); y = torch.empty_like(x)
bytes_ = synth.algorithmic_bytes(N)["spmv"] * B
for nt in (0, 1):
    for bpc in (2, 4, 6, 8, 12, 16, 32):
        sol.set_option("nt_loads", nt); sol.set_option("spmv_blocks_per_cu", bpc)
        for _ in range(3): sol.bt_spmv(dS, x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): sol.bt_spmv(dS, x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"nt={nt} blocks/cu={bpc:2d}: {ms*1e3:7.1f} us  {bytes_/ms/1e6:7.1f} GB/s  frac {bytes_/ms/1e6/8000:.3f}", flush=True)
