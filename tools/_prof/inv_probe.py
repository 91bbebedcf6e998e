"""Calibration of tests/test_gpu_invariants.py: prints every measured quantity over its rounding-model scale (run on the GPU box)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import test_gpu_invariants as T
from mpcgpu_amd import PcgSolver, synth
n = 14
worst = {}
for N in (32, 128, 512):
    for pc in ("ss", "jacobi"):
        for seed in range(4):
            k = synth.make_kkt(N, 1, 9000 + N + 17 * seed)
            S, P, g = synth.form_schur(k, precond=pc); S, P, g = S[0], P[0], g[0]
            rng = np.random.default_rng(N + seed); lam0 = (0.1 * rng.standard_normal(n * N)).astype(np.float32)
            sol = PcgSolver(N, max_batch=1); dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
            for K in (2, 10, 50, 167):
                q = T.step_quantities(S, P, g, lam0, T.state(sol, dS, dP, dg, lam0, K - 1), T.state(sol, dS, dP, dg, lam0, K), N, K)
                for key, val in q.items():
                    worst[key] = max(worst.get(key, 0.0), val)
                print(f"N={N} {pc} s{seed} K={K}: " + " ".join(f"{a} {b:.2g}" for a, b in q.items()))
print("WORST (each measured quantity / its scale):", {a: float(f"{b:.3g}") for a, b in worst.items()})
