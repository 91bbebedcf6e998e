#!/usr/bin/env python3
"""Driver for the PMC passes of the round-4 producer kernels: a few launches of the chunk-walking Schur formation (+ seam kernel) and of
the dz kernel on 1024 x 128 knots, nothing else.  run_walk.py [chunk]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, synth
N, B = 128, 1024
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sol = PcgSolver(N, max_batch=B)
k = synth.make_kkt(N, 64, 1)
G, C, g, c = (torch.from_numpy(a).cuda().repeat(B // 64, 1).contiguous() for a in synth.pack_kkt_dense(k, np.float32))
G0 = G.clone(); S = torch.empty(B, 3 * 196 * N, device="cuda"); P = torch.empty_like(S); gm = torch.empty(B, 14 * N, device="cuda")
lam = torch.randn(B, 14 * N, device="cuda")
sol.set_option("schur_chunk", L)
for i in range(5):
    G.copy_(G0)
    sol.form_schur(G, C, g, c, 1e-3, "ss", S=S, Pinv=P, gamma=gm)
    sol.compute_dz(G, C, g, lam)
torch.cuda.synchronize()
