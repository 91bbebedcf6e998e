#!/usr/bin/env python3
"""Time mpcg_generate_kkt (HIP twin of generate_kkt_submatrices) on B windows of N knots (run on the GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):                      # A/B against another build of the library (tools/_prof/ab/)
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, Plant, iiwa
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
xu, ee, xs = iiwa.random_windows(N, B, seed=3)
sol = PcgSolver(N, max_batch=B)
plant = Plant()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
dxu, dee, dxs = t(xu), t(ee), t(xs)
out = sol.generate_kkt(plant, dee.reshape(B, -1), dxs, dxu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): out = sol.generate_kkt(plant, dee.reshape(B, -1), dxs, dxu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"generate_kkt N={N} B={B}: {ms:.3f} ms  ({B*(N-1)/ms/1e3:.2f} M knots/s)")
