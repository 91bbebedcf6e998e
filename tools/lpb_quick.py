"""Timing of the lane-per-block kernel against the single-workgroup kernels (GPU).  (Correctness: tests/test_gpu_lpb.py.)"""
import os, sys, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpcgpu_amd._lib as _L
if os.environ.get("AB_LIB"):                      # A/B against another build of the library (tools/_prof/ab/)
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth

dev = torch.device("cuda")
out = {}
# correctness
def timeit(sol, S, P, g, B, N, cfg, reps=5):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), int(it.sum().item())

for N, B in ((128, 1024), (128, 256), (128, 1), (96, 1024), (64, 1024), (64, 2048), (32, 2048), (32, 1)):
    k = synth.make_kkt(N, min(B, 64), 1)
    S0, P0, g0 = synth.form_schur(k)
    rep = (B + S0.shape[0] - 1) // S0.shape[0]
    S = torch.from_numpy(np.tile(S0, (rep, 1))[:B]).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B]).to(dev)
    g = torch.from_numpy(np.tile(g0, (rep, 1))[:B]).to(dev)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
    res = {}
    for name, lpb in (("lpb", -1), ("traj", 0)):
        sol = PcgSolver(N, max_batch=B)
        sol.set_option("pcg_lpb", lpb)
        sol.set_option("pcg_rpl", 0)               # (this tool compares the lane-per-block and row-pair kernels; tools/rpl_quick.py the row-per-lane one)
        ms, its = timeit(sol, S, P, g, B, N, cfg)
        res[name] = {"ms": ms, "Mit_s": its / ms / 1e3, "family": sol.get_option("last_kernel_family"), "waves": sol.get_option("last_kernel_waves")}
    print("time", N, B, json.dumps(res), flush=True)
