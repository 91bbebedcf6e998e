#!/usr/bin/env python3
"""Randomised differential run of the double-precision PCG kernels (row-per-lane up to N = 32, its clustered form up to N = 256, the streaming
kernel beyond / with "cluster" = 0) against the oracle's float64 iterate:  fuzz_f64.py [cases] [seed]   (run on the GPU box)"""
import os, sys, time, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import relinf
orc.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
fam, worst, bad, fix = collections.Counter(), 0.0, 0, 0
t0 = time.time()
for ci in range(cases):
    N = int(rng.choice([rng.integers(2, 33), rng.integers(33, 65), rng.integers(65, 129), rng.integers(129, 257), rng.integers(257, 513)], p=[0.15, 0.2, 0.3, 0.2, 0.15]))
    B = int(rng.integers(1, 8))
    pc = str(rng.choice(["ss", "jacobi"]))
    K = int(rng.integers(1, min(40, 14 * N)))
    k = synth.make_kkt(N, B, int(rng.integers(1 << 30)))
    S, P, g = synth.form_schur(k, precond=pc, dtype=np.float64, poison_unused=True)
    lam0 = 0.1 * rng.standard_normal((B, 14 * N)) if rng.random() < 0.5 else np.zeros((B, 14 * N))
    sol = PcgSolver(N, max_batch=B)
    u = rng.random()
    stream = u < 0.12 and N <= 340                        # the streaming kernel (its iterate vectors fit LDS up to N = 350)
    if stream:
        sol.set_option("cluster", 0)
    elif u < 0.3 and N <= 256:
        sol.set_option("pcg_lqk", 0)                      # the clustered row-per-lane kernel instead of the lane-quad kernels
    elif u < 0.36 and N <= 32:
        sol.set_option("pcg_lqk", 1)                      # the lane-quad kernel forced on a short horizon
    Sd, Pd = dev(S), dev(P)                               # (blocks (0, left) and (N-1, right) stay NaN: no kernel, no symmetry check may read them)
    lam = dev(lam0.copy())
    it, ex = sol.solve_f64(Sd, Pd, dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
    torch.cuda.synchronize()
    f = sol.get_option("last_kernel_family")
    fam[f] += 1
    fix += sol.get_option("cluster_fixups")
    lamh = lam.cpu().numpy()
    ok = (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
        ref = orc.pcg(Sz, Pz, g[b], lam0[b], N, K, 0.0, pc)["lam"]
        # what rounding-level noise does to float64 CG on THIS system after K iterations (small systems iterated past convergence are chaotic:
        # eta is not monotone, tools/_prof/small_f64.py: the oracle's own spread reaches 1e-3 at N = 5..9, K = 40): the oracle on inputs moved by one ulp, eight trials
        pert = lambda a_: a_ * (1 + 1.1e-16 * rng.standard_normal(a_.shape))
        band = max(relinf(orc.pcg(pert(Sz), Pz, pert(g[b]), pert(lam0[b]), N, K, 0.0, pc)["lam"], ref) for _ in range(8))
        e = relinf(lamh[b], ref)
        tol = max(1e-9, 20 * band)
        worst = max(worst, e / tol)
        ok = ok and e <= tol
    if not ok:
        bad += 1
        print(f"MISMATCH case {ci}: N={N} B={B} {pc} K={K} family {f} warm {bool(np.abs(lam0).max() > 0)}", flush=True)
print(f"{cases} random double-precision cases in {time.time()-t0:.0f} s: kernel families {dict(sorted(fam.items()))}, worst distance from the float64 oracle iterate / tolerance (max(1e-9, 20 x the oracle's own 1-ulp band)) {worst:.3f}, "
      f"trajectories left to the fix-up {fix}, mismatches {bad}")
sys.exit(1 if bad else 0)
