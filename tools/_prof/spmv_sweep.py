#!/usr/bin/env python3
"""bt_spmv: workgroups per CU sweep at the bench's SpMV shape (N=128, 4096 trajectories, 1,285 MB)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpcgpu_amd import PcgSolver, _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
def t(fn, reps=9):
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
N, B = 128, 4096
sol = PcgSolver(N, max_batch=B)
S = torch.randn(B, 588 * N, device="cuda"); x = torch.randn(B, 14 * N, device="cuda"); y = torch.empty_like(x)
for nt in (1, 0):
    sol.set_option("nt_loads", nt)
    for bpc in [int(a) for a in (sys.argv[1:] or "1 2 3 4 6 8 12 16".split())]:
        sol.set_option("spmv_blocks_per_cu", bpc)
        ms = t(lambda: sol.bt_spmv(S, x, y))
        print("nt_loads=%d spmv_blocks_per_cu=%2d: %.3f ms  %.0f GB/s" % (nt, bpc, ms, B * 313824 / ms / 1e6))
# back-to-back launches (the bench's way: 20 per timing) against one launch per timing
sol.set_option("nt_loads", 1)
for bpc in (2, 3, 4):
    sol.set_option("spmv_blocks_per_cu", bpc)
    def many():
        for _ in range(20):
            sol.bt_spmv(S, x, y)
    ms1, ms20 = t(lambda: sol.bt_spmv(S, x, y)), t(many, 5) / 20
    print("spmv_blocks_per_cu=%d: single launch %.3f ms (%.0f GB/s) | 20 back to back %.3f ms each (%.0f GB/s)" % (bpc, ms1, B * 313824 / ms1 / 1e6, ms20, B * 313824 / ms20 / 1e6))
