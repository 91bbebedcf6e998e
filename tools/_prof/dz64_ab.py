#!/usr/bin/env python3
"""A/B of mpcg_compute_dz_f64's grid cap (MPCG_DZ64_CAPMUL = 1 shipped vs 4 = the float kernel's): 1024 x 128 knots, hip events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpcgpu_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from mpcgpu_amd import PcgSolver, synth
N, B = 128, 1024
sol = PcgSolver(N, max_batch=B)
k = synth.make_kkt(N, 64, 4242)
Gh, Ch, gh, ch = synth.pack_kkt_dense(k, np.float64)
dev = torch.device("cuda", 0)
G, C_, g_, c_ = (torch.from_numpy(np.tile(a, (B // 64, 1)).copy()).to(dev) for a in (Gh, Ch, gh, ch))
lam = torch.randn(B, 14 * N, dtype=torch.float64, device=dev)
dz = torch.empty(B, 21 * N - 7, dtype=torch.float64, device=dev)
for _ in range(20):
    sol.compute_dz(G, C_, g_, lam, dz=dz)
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        sol.compute_dz(G, C_, g_, lam, dz=dz)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50)
print(f"{_lib.LIB_PATH}: compute_dz_f64 1024 x 128: {min(ts):.4f} ms (min of 5 x 50), median {sorted(ts)[2]:.4f}")
