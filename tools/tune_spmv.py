#!/usr/bin/env python3
"""Sweep bt_spmv launch knobs on B x N=128 (run on the GPU box).  B = 4096: S = 1.2 GB, far beyond the 256 MiB
Infinity Cache, i.e. a true HBM stream (B = 1024 is partly L3-served)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, synth
N = 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sol = PcgSolver(N, max_batch=B)
dS0, dP, dg = bench.build_inputs(sol, N, 512, 0, "ss", torch.device("cuda", 0))
dS = dS0.repeat((B + 511) // 512, 1)[:B].contiguous()
del dS0, dP
x = torch.randn(B, 14 * N, device="cuda"); y = torch.empty_like(x)
y0 = None
bytes_ = synth.algorithmic_bytes(N)["spmv"] * B
for depth in (0,):
    for nt in (0, 1):
        for bpc in (4, 8, 16, 32, 64):
            sol.set_option("nt_loads", nt); sol.set_option("spmv_blocks_per_cu", bpc)
            for _ in range(3): sol.bt_spmv(dS, x, y)
            if y0 is None: y0 = y.clone()
            assert torch.equal(y, y0), "variants must agree bit for bit"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): sol.bt_spmv(dS, x, y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"B={B} nt={nt} blocks/cu={bpc:2d}: {ms*1e3:7.1f} us  {bytes_/ms/1e6:7.1f} GB/s  frac {bytes_/ms/1e6/8000:.3f}", flush=True)
