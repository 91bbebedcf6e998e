"""Timing of ablated builds of the lane-per-block kernel (results are meaningless, only the time counts):
   for m in 0 1 2 3 4 8 15; do hipcc ... -DMPCG_ABLATE=$m -o tools/_prof/ab/libmpcg_abl$m.so; done  (tools/lpb_ablate.sh)
MPCG_ABLATE bits: 1 no direct product, 2 no transposed product, 4 no element-wise vector updates, 8 no wave fold."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpcgpu_amd._lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
N, B = 128, 1024
k = synth.make_kkt(N, 16, 1)
S0, P0, g0 = synth.form_schur(k)
S = torch.from_numpy(np.tile(S0, (B // 16, 1))).to(dev); P = torch.from_numpy(np.tile(P0, (B // 16, 1))).to(dev)
g = torch.from_numpy(np.tile(g0, (B // 16, 1))).to(dev)
sol = PcgSolver(N, max_batch=B)
cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=167)
lam = torch.zeros(B, 14 * N, device=dev)
ts = []
for i in range(8):
    lam.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, "ss"); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts[2:]))
print(os.environ.get("AB_LIB", "product"), f"{ms:.4f} ms  {ms * 1e3 / 4 / 167:.3f} us/iteration/CU  ({ms / 4 / 167 * 2.38e6:.0f} cycles)", flush=True)
