#!/usr/bin/env python3
"""What the matrix load of the lane-quad kernel costs when the blocks come from HBM and when they sit in the Infinity Cache:
256 trajectories of 128 knots (108 MB of blocks) solved again and again (resident in the 256 MB cache) vs four such sets in rotation (431 MB: every call streams
from HBM), at 0 and 167 iterations; and the full batch of 1024 in one launch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
N = 128
k = synth.make_kkt(N, 8, 1)
S0, P0, g0 = synth.form_schur(k)
def mk(B):
    return tuple(torch.from_numpy(np.tile(a, ((B + 7) // 8, 1))[:B]).to(dev) for a in (S0, P0, g0))
sets = [mk(256) for _ in range(4)]
full = mk(1024)
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(ts))
for K in (0, 1, 167):
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=1024)
    sol.set_option("assume_symmetric", 1)
    lam = torch.zeros(1024, 14 * N, device=dev)
    def warm():
        for _ in range(4): sol.solve(*sets[0], lam[:256], cfg, "ss")
    def cold():
        for s in sets: sol.solve(*s, lam[:256], cfg, "ss")
    def one():
        sol.solve(*full, lam, cfg, "ss")
    tw, tc, t1 = timed(warm, 10), timed(cold, 10), timed(one, 10)
    print(f"K={K:3d}: 4 x 256 same set (cache-resident) {tw:8.1f} us | 4 x 256 rotating sets (HBM) {tc:8.1f} us | 1 x 1024 {t1:8.1f} us   (family {sol.get_option('last_kernel_family')})", flush=True)
