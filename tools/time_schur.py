#!/usr/bin/env python3
"""Time mpcg_form_schur / mpcg_compute_dz on B trajectories x N knots (default 1024 x 128):  time_schur.py [N [B]]
— the register-resident chunk-walking formation at its automatic and at forced chunk lengths, the LDS kernels, both dz kernels, each with
the rate on its algorithmic-bytes model."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):                      # A/B against another build of the library (tools/_prof/ab/)
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sol = PcgSolver(N, max_batch=B)
Bs = min(B, 64)
k = synth.make_kkt(N, Bs, 1)
rep = (B + Bs - 1) // Bs
G, C, g, c = (torch.from_numpy(a).cuda().repeat(rep, 1)[:B].contiguous() for a in synth.pack_kkt_dense(k, np.float32))
G0 = G.clone(); S = torch.empty(B, 3 * 196 * N, device="cuda"); P = torch.empty_like(S); gm = torch.empty(B, 14 * N, device="cuda")
lam = torch.randn(B, 14 * N, device="cuda")
def t(fn, reps=7):
    ts = []
    for i in range(reps):
        G.copy_(G0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:])) * 1e3
knots = B * N
def line(tag, us, bytes_per_knot):
    print("%-58s %8.1f us   %6.2f TB/s on the %d B/knot model" % (tag, us, knots * bytes_per_knot / us / 1e6, bytes_per_knot))
for pc, bpk in (("ss", 8036), ("jacobi", 6468), ("none", 5684)):
    sol.set_option("schur_dpp", 1)
    for L in (0,) + ((1, 2, 4, 8, 16) if B * N < 65536 else (8, 16, 32)):
        sol.set_option("schur_chunk", L)
        us = t(lambda: sol.form_schur(G, C, g, c, 1e-3, pc, S=S, Pinv=P, gamma=gm))
        line("form_schur %-6s chunk %s" % (pc, "auto = %d" % sol.get_option("last_schur_chunk") if L == 0 else str(L)), us, bpk)
    sol.set_option("schur_dpp", 0)
    line("form_schur %-6s LDS kernels" % pc, t(lambda: sol.form_schur(G, C, g, c, 1e-3, pc, S=S, Pinv=P, gamma=gm)), bpk)
sol.set_option("schur_dpp", 1); sol.set_option("schur_chunk", 0)
sol.form_schur(G, C, g, c, 1e-3, "ss", S=S, Pinv=P, gamma=gm)
for d in (0, 1):
    sol.set_option("dz_dpp", d)
    line("compute_dz dz_dpp=%d" % d, t(lambda: sol.compute_dz(G, C, g, lam)), 980 + 1176 + 84 + 56 + 84)
