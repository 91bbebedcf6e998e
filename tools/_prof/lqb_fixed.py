#!/usr/bin/env python3
"""Fixed cost of a solve (max_iter = 0: matrix load, set-up passes, write-back, launch) — one trajectory and a full batch, default policy vs "pcg_lqb" = 0."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
for N in (128, 64, 40):
    k = synth.make_kkt(N, 8, 1)
    S0, P0, g0 = synth.form_schur(k)
    for B in (1, 1024):
        S, P, g = (torch.from_numpy(np.tile(a, ((B + 7) // 8, 1))[:B]).to(dev) for a in (S0, P0, g0))
        out = []
        for lqb in (-1, 0):
            sol = PcgSolver(N, max_batch=B)
            sol.set_option("assume_symmetric", 1); sol.set_option("pcg_lqb", lqb)
            res = {}
            for K in (0, 1, 2):
                cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
                lam = torch.zeros(B, 14 * N, device=dev)
                for _ in range(20):
                    sol.solve(S, P, g, lam, cfg, "ss")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    sol.solve(S, P, g, lam, cfg, "ss")
                e1.record(); torch.cuda.synchronize()
                res[K] = e0.elapsed_time(e1) / 100 * 1e3
            out.append(f"{'auto' if lqb else 'lqb off'} (family {sol.get_option('last_kernel_family')}): K=0 {res[0]:.1f} us, K=1 {res[1]:.1f}, K=2 {res[2]:.1f}")
        print(f"N={N} B={B}: " + " | ".join(out), flush=True)
