#!/usr/bin/env python3
"""mpcg_generate_kkt in float arithmetic ("kkt_f32" = 1: two knots per lane in packed float; = 2: one knot per lane) against the float64-inside kernel: worst difference per output array (relative to the array's
largest entry, and to the float64 host restatement on a few knots) and time per 1024 x 127 knots at steady clocks."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, Plant, iiwa
N, B = 128, 1024
dev = torch.device("cuda", 0)
xu, ee, xs = iiwa.random_windows(N, B, seed=3)
plant = Plant()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
dxu, dee, dxs = t(xu), t(ee), t(xs)
outs = {}
for f32 in (0, 1, 2):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("kkt_f32", f32)
    call = lambda: sol.generate_kkt(plant, dee.reshape(B, -1), dxs, dxu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
    out = call(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < 0.05:
        for _ in range(10): call()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    outs[f32] = [o.cpu().numpy().astype(np.float64) for o in out]
    print(f"kkt_f32={f32}: {ms:.4f} ms per {B} x {N - 1} knots, finite {all(np.isfinite(o).all() for o in outs[f32])}")
for nm, a64, a32, apk in zip("GCgc", outs[0], outs[1], outs[2]):
    d = np.abs(a64 - a32)
    print(f"  {nm}: max |f32 - f64| = {d.max():.3e}  (max |{nm}| = {np.abs(a64).max():.3e}; relative {d.max() / np.abs(a64).max():.2e}; rms {np.sqrt((d ** 2).mean()):.2e})"
          f"   one-knot float vs packed float: max {np.abs(apk - a32).max():.3e}, identical {np.array_equal(apk, a32)}")
# against the float64 host restatement on a few windows
import iiwa_ref
M = iiwa_ref.Model()
for b in (0, 511, 1023):
    want = iiwa_ref.generate_kkt(M, xu[b].astype(np.float32).astype(np.float64), ee[b].astype(np.float32).astype(np.float64), xs[b].astype(np.float32).astype(np.float64), N)
    for f32 in (0, 1, 2):
        errs = [float(np.abs(outs[f32][i][b] - want[i]).max() / max(1.0, np.abs(want[i]).max())) for i in range(4)]
        print(f"  window {b} kkt_f32={f32}: error vs host float64 restatement / max(1, |ref|):  G {errs[0]:.2e}  C {errs[1]:.2e}  g {errs[2]:.2e}  c {errs[3]:.2e}")
