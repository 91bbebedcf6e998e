#!/usr/bin/env python3
"""linsys_t = double: mpcg_form_schur_f64 / mpcg_compute_dz_f64 (the round-1 LDS kernels instantiated for double) at B x N knots."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
sol = PcgSolver(N, max_batch=B)
k = synth.make_kkt(N, 64, 5)
Gd, Cd, gd, cd = (torch.from_numpy(np.tile(a, (B // 64, 1))).to(dev) for a in synth.pack_kkt_dense(k, np.float64))
G0 = Gd.clone()
for pc in ("ss", "jacobi"):
    ms = bench.timed(lambda: (Gd.copy_(G0), sol.form_schur(Gd, Cd, gd, cd, 1e-3, pc)), 5, warm=2) - bench.timed(lambda: Gd.copy_(G0), 5, warm=2)
    print("form_schur_f64 %-6s %d x %d: %.3f ms" % (pc, B, N, ms))
Gd.copy_(G0); S, P, gam = sol.form_schur(Gd, Cd, gd, cd, 1e-3, "ss")
lam = torch.randn(B, 14 * N, dtype=torch.float64, device=dev)
print("compute_dz_f64 %d x %d: %.3f ms" % (B, N, bench.timed(lambda: sol.compute_dz(Gd, Cd, gd, lam), 5, warm=2)))
