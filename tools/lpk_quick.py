"""Timing of the lane-pair-per-knot kernel (pcg_lpk.hip.h) against the lane-per-block kernel it succeeds (GPU).  Also prints the
largest difference between the two solutions after the fixed iteration count (same mathematics, different summation order).
(Correctness against the oracle: tests/test_gpu_lpk.py.)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpcgpu_amd._lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth

dev = torch.device("cuda")


def timeit(sol, S, P, g, B, N, cfg, pc, reps=7):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, pc); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:])), int(it.sum().item()), lam


shapes = [(128, 1024), (128, 256), (128, 1), (96, 1024), (64, 2048), (64, 1), (48, 2048), (40, 2048)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for N, B in shapes:
    k = synth.make_kkt(N, min(B, 64), 1)
    S0, P0, g0 = synth.form_schur(k)
    rep = (B + S0.shape[0] - 1) // S0.shape[0]
    S = torch.from_numpy(np.tile(S0, (rep, 1))[:B]).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B]).to(dev)
    g = torch.from_numpy(np.tile(g0, (rep, 1))[:B]).to(dev)
    for pc in ("ss", "jacobi"):
        cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
        res, lams = {}, {}
        for name, opt in (("lpk", "pcg_lpk"),):
            sol = PcgSolver(N, max_batch=B)
            sol.set_option(opt, 1)
            ms, its, lam = timeit(sol, S, P, g, B, N, cfg, pc)
            lams[name] = lam
            res[name] = {"ms": round(ms, 4), "Mit_s": round(its / ms / 1e3, 1), "family": sol.get_option("last_kernel_family")}
        cfg2 = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=10)
        d = []
        for name, opt in (("lpk", "pcg_lpk"),):
            sol = PcgSolver(N, max_batch=B)
            sol.set_option(opt, 1)
            lam = torch.zeros(B, 14 * N, device=dev)
            sol.solve(S, P, g, lam, cfg2, pc)
            d.append(lam)
        rel = float(((d[0] - d[1]).abs().amax() / d[1].abs().amax()).item())
        print("time", N, B, pc, json.dumps(res), "rel diff after 10 it: %.2e" % rel, "finite", bool(torch.isfinite(lams["lpk"]).all().item()), flush=True)
