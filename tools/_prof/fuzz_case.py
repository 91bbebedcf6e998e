#!/usr/bin/env python3
"""One case of tests/fuzz_cases.py looked at closely: replays the generator of `seed` up to `case`, then compares the default kernel, the CPU float32
restatement (the oracle in float) and the float64 iterate per trajectory.   python tools/_prof/fuzz_case.py <seed> <case>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import fp32_band, relinf
seed, case = int(sys.argv[1]), int(sys.argv[2])
orc.build()
rng = np.random.default_rng(seed)
for ci in range(case + 1):
    N = int(rng.choice([rng.integers(2, 33), rng.integers(33, 129), rng.integers(129, 400)], p=[0.4, 0.4, 0.2]))
    B = int(rng.integers(1, 7))
    pc = str(rng.choice(["ss", "jacobi"]))
    K = int(rng.integers(1, min(40, 14 * N)))
    kseed = int(rng.integers(1 << 30))
    warm = rng.random() < 0.5
    lam0 = (0.1 * rng.standard_normal((B, 14 * N))).astype(np.float32) if warm else np.zeros((B, 14 * N), np.float32)
    force = N <= 32 and rng.random() < 0.25
print(f"case {case}: N={N} B={B} {pc} K={K} warm={warm} forced lane-pair={force}")
k = synth.make_kkt(N, B, kseed)
S, P, g = synth.form_schur(k, precond=pc, dtype=np.float32, poison_unused=True)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for KK in sorted({1, 2, 3, max(1, K - 2), K, K + 2}):
    sol = PcgSolver(N, max_batch=B)
    if force: sol.set_option("pcg_lpk", 1)
    lam = dev(lam0.copy())
    it, ex = sol.solve(dev(S), dev(P), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=KK), pc)
    torch.cuda.synchronize()
    lam_h = lam.cpu().numpy()
    rows = []
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
        ref = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, KK, 0.0, pc)["lam"]
        c32 = orc.pcg(Sz, Pz, g[b], lam0[b], N, KK, 0.0, pc)["lam"]
        band = fp32_band(orc, Sz, Pz, g[b], lam0[b], N, KK, pc, ref)
        rows.append(f"b{b}: gpu {relinf(lam_h[b], ref):.2e} cpu-f32 {relinf(c32, ref):.2e} band {band:.2e}")
    print(f"K={KK} family {sol.get_option('last_kernel_family')}: " + " | ".join(rows))
