#!/usr/bin/env python3
"""HBM write-side ceilings on this box: fill (write only), copy (read + write), and a 2:1 write:read mix like form_schur's (752 MB out, 388 MB in)."""
import torch, numpy as np
def t(fn, reps=7):
    ts = []
    for i in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
n = 752 * 1024 * 1024 // 4
x = torch.empty(n, device="cuda"); y = torch.randn(n, device="cuda"); z = torch.randn(n // 2, device="cuda")
ms = t(lambda: x.zero_()); print("fill 752 MB: %.3f ms  %.0f GB/s written" % (ms, n * 4 / ms / 1e6))
ms = t(lambda: x.copy_(y)); print("copy 752 MB: %.3f ms  %.0f GB/s written, %.0f GB/s total" % (ms, n * 4 / ms / 1e6, 2 * n * 4 / ms / 1e6))
xv = x.view(2, -1)
ms = t(lambda: torch.mul(z.unsqueeze(0), 2.0, out=xv) if False else xv.copy_(z.unsqueeze(0).expand(2, -1)))
print("write 752 MB from 376 MB read: %.3f ms  %.0f GB/s written, %.0f GB/s total" % (ms, n * 4 / ms / 1e6, 1.5 * n * 4 / ms / 1e6))
# read side
r = torch.randn(1285 * 1024 * 1024 // 4, device="cuda")
ms = t(lambda: r.sum()); print("sum over 1285 MB: %.3f ms  %.0f GB/s read" % (ms, r.numel() * 4 / ms / 1e6))
ms = t(lambda: r.max()); print("max over 1285 MB: %.3f ms  %.0f GB/s read" % (ms, r.numel() * 4 / ms / 1e6))
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpcgpu_amd import PcgSolver
N, B = 128, 4096
sol = PcgSolver(N, max_batch=B)
S = torch.randn(B, 588 * N, device="cuda"); x = torch.randn(B, 14 * N, device="cuda"); y = torch.empty_like(x)
ms = t(lambda: sol.bt_spmv(S, x, y)); print("bt_spmv %d x N=%d: %.3f ms  %.0f GB/s on the 313,824 B/trajectory model" % (B, N, ms, B * 313824 / ms / 1e6))
