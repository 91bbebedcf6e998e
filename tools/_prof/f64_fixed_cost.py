"""linsys_t = double: fixed cost of a solve (bringing S and Pinv on chip, setup, write-back) against its per-iteration cost:
f64_fixed_cost.py [N] [batches] [option=value ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BATCHES = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 1]
opts = [a.split("=") for a in sys.argv[3:]]
for B in BATCHES:
    sol = PcgSolver(N, max_batch=B)
    for k, v in opts: sol.set_option(k, int(v))
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    dS, dP, dg = torch.nan_to_num(dS).double(), torch.nan_to_num(dP).double(), dg.double()
    lam = torch.zeros(B, 14 * N, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    ts = {}
    for K in (1, 21, 41, 81):
        cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
        def go():
            lam.zero_(); sol.solve_f64(dS, dP, dg, lam, cfg, iters=it, exits=ex)
        go()
        ts[K] = bench.timed(go, 5, warm=2) - bench.timed(lambda: lam.zero_(), 5, warm=2)
    per_it = (ts[81] - ts[41]) / 40
    print(f"N={N} batch={B}: " + "  ".join(f"K={k}: {v*1e3:.1f} us" for k, v in ts.items()) +
          f"   per iteration {per_it*1e3:.2f} us = {B / per_it / 1e3:.1f} M it/s, fixed {1e3*(ts[1]-per_it):.1f} us (family {sol.get_option('last_kernel_family')})", flush=True)
