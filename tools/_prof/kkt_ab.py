#!/usr/bin/env python3
"""generate_kkt: analytic gradient recursion vs one-sided differences — time at 1024 x 128 knots and accuracy of C (= -A, -B) against the
float64 host restatement with central differences (oracle/iiwa_ref.py) on 6 windows of 16 knots."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import iiwa_ref
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, Plant, iiwa
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
plant = Plant()
N, B = 128, 1024
xu, ee, xs = iiwa.random_windows(N, B, seed=3)
dxu, dee, dxs = f32(xu), f32(ee.reshape(B, -1)), f32(xs)
sol = PcgSolver(N, max_batch=B)
def t(fn, reps=9):
    ts = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:]))
M = iiwa_ref.Model()
Ns, Bs = 16, 6
xus, ees, xss = iiwa.random_windows(Ns, Bs, seed=8)
ref = [iiwa_ref.generate_kkt(M, xus[b].astype(np.float32).astype(np.float64), ees[b].astype(np.float32).astype(np.float64), xss[b].astype(np.float32).astype(np.float64), Ns)[1] for b in range(Bs)]
sols = PcgSolver(Ns, max_batch=Bs)
for analytic in (1, 0, 1, 0):
    sol.set_option("kkt_analytic", analytic); sols.set_option("kkt_analytic", analytic)
    ms = t(lambda: sol.generate_kkt(plant, dee, dxs, dxu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N)))
    C = sols.generate_kkt(plant, f32(ees.reshape(Bs, -1)), f32(xss), f32(xus), iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(Ns))[1].cpu().numpy()
    err = max(np.abs(C[b] - ref[b]).max() for b in range(Bs))
    print("kkt_analytic=%d: %.4f ms per %d x %d knots (%.0f M knots/s); max |C - C_ref(float64, central differences)| = %.2e" % (analytic, ms, B, N - 1, B * (N - 1) / ms / 1e3, err))
