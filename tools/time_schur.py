#!/usr/bin/env python3
"""Time mpcg_form_schur / mpcg_compute_dz on B trajectories x N knots (default 128 x 128):  time_schur.py [N [B]]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):                      # A/B against another build of the library (tools/_prof/ab/)
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sol = PcgSolver(N, max_batch=B)
k = synth.make_kkt(N, B, 1)
G, C, g, c = (torch.from_numpy(a).cuda() for a in synth.pack_kkt_dense(k, np.float32))
G0 = G.clone(); S = torch.empty(B, 3 * 196 * N, device="cuda"); P = torch.empty_like(S); gm = torch.empty(B, 14 * N, device="cuda")
lam = torch.randn(B, 14 * N, device="cuda")
def t(fn, reps=6):
    ts = []
    for i in range(reps):
        G.copy_(G0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])) * 1e3
for dpp in (0, 1, 2):
    sol.set_option("schur_dpp", 1 if dpp else 0)
    sol.set_option("schur_fma", 1 if dpp == 2 else 0)
    tag = ("register/DPP kernels, fused multiply-adds (schur_fma)" if dpp == 2 else "register/DPP kernels") if dpp else "LDS kernels"
    print("form_schur (ss)   %d traj x %d knots, %s: %.1f us" % (B, N, tag, t(lambda: sol.form_schur(G, C, g, c, 1e-3, "ss", S=S, Pinv=P, gamma=gm))))
    print("form_schur (jac)  %d traj x %d knots, %s: %.1f us" % (B, N, tag, t(lambda: sol.form_schur(G, C, g, c, 1e-3, "jacobi", S=S, Pinv=P, gamma=gm))))
sol.set_option("schur_fma", 0)
sol.form_schur(G, C, g, c, 1e-3, "ss", S=S, Pinv=P, gamma=gm)
print("compute_dz        %d traj x %d knots: %.1f us" % (B, N, t(lambda: sol.compute_dz(G, C, g, lam))))
