#!/usr/bin/env python3
"""form_schur (ss) on 1024 x 128 knots at several chunk lengths, for A/B builds of the library (AB_LIB=...): walk_chunks.py L1 L2 ..."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, synth
N, B = 128, 1024
sol = PcgSolver(N, max_batch=B)
k = synth.make_kkt(N, 64, 1)
G, C, g, c = (torch.from_numpy(a).cuda().repeat(B // 64, 1).contiguous() for a in synth.pack_kkt_dense(k, np.float32))
G0 = G.clone(); S = torch.empty(B, 3 * 196 * N, device="cuda"); P = torch.empty_like(S); gm = torch.empty(B, 14 * N, device="cuda")
def t(fn, reps=9):
    ts = []
    for i in range(reps):
        G.copy_(G0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:])) * 1e3
out = []
for L in [int(a) for a in sys.argv[1:]] or [8, 11, 16]:
    sol.set_option("schur_chunk", L)
    out.append("L=%d %.1f us" % (L, t(lambda: sol.form_schur(G, C, g, c, 1e-3, "ss", S=S, Pinv=P, gamma=gm))))
print(os.environ.get("AB_LIB", "default"), " | ".join(out))
