"""Replay ONE case of tests/fuzz_cases.py (seed, case index) on every float kernel that can serve it: is an exceedance of the heuristic float32 band a
property of the system (all kernels and the CPU float32 restatement scatter alike) or of one kernel?   fuzz_replay.py <seed> <case>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import fp32_band, relinf
seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for ci in range(case + 1):
    N = int(rng.choice([rng.integers(2, 33), rng.integers(33, 129), rng.integers(129, 400)], p=[0.4, 0.4, 0.2]))
    B = int(rng.integers(1, 7)); pc = str(rng.choice(["ss", "jacobi"])); K = int(rng.integers(1, min(40, 14 * N)))
    seed = int(rng.integers(1 << 30)); warm = rng.random() < 0.5
    lam0 = (0.1 * rng.standard_normal((B, 14 * N))).astype(np.float32) if warm else np.zeros((B, 14 * N), np.float32)
    forced = N <= 32 and rng.random() < 0.25
print(f"case {case} of seed {seed0}: N={N} B={B} {pc} K={K} warm={warm}")
k = synth.make_kkt(N, B, seed)
S, P, g = synth.form_schur(k, precond=pc, dtype=np.float32)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
refs, tols, cpu = [], [], []
for b in range(B):
    r64 = orc.pcg(S[b].astype(np.float64), P[b].astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, K, 0.0, pc)["lam"]
    refs.append(r64); tols.append(max(2e-5 if K <= 3 else 1e-3, 4 * fp32_band(orc, S[b], P[b], g[b], lam0[b], N, K, pc, r64)))
    cpu.append(relinf(orc.pcg(S[b], P[b], g[b], lam0[b], N, K, 0.0, pc)["lam"], r64) / tols[-1])
print("CPU float32 restatement, error / tolerance per trajectory:", " ".join(f"{c:.2f}" for c in cpu))
for name, opts in (("default", {}), ("row-per-lane 4 waves", {"pcg_rpl": 1, "rpl_waves": 4}), ("row-per-lane 8 waves", {"pcg_rpl": 1, "rpl_waves": 8}),
                   ("row-per-lane 16 waves", {"pcg_rpl": 1, "rpl_waves": 16}), ("lane-pair", {"pcg_lpk": 1}), ("row-pair 8 waves", {"pcg_rpl": 0, "pcg_lpk": 0, "pcg_waves": 8})):
    try:
        sol = PcgSolver(N, max_batch=B)
        for o, v in opts.items(): sol.set_option(o, v)
        lam = dev(lam0.copy())
        sol.solve(dev(S), dev(P), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        e = [relinf(lam.cpu().numpy()[b], refs[b]) / tols[b] for b in range(B)]
        print(f"{name} (family {sol.get_option('last_kernel_family')} x {sol.get_option('last_kernel_waves')}): error / tolerance " + " ".join(f"{x:.2f}" for x in e))
    except Exception as ex:
        print(name, "-", str(ex)[:100])
# the recurrence invariants (tests/test_gpu_invariants.py) of the default kernel at every iteration of the flagged trajectories: each update within a few
# float32 ulps of its float64 evaluation = nothing wrong with the arithmetic, whatever CG makes of the differences
import test_gpu_invariants as T
for b in range(B):
    sol = PcgSolver(N, max_batch=1)
    dS, dP, dg = (dev(a[b]) for a in (S, P, g))
    worst = {}
    for KK in range(2, K + 1):
        q = T.step_quantities(S[b], P[b], g[b], lam0[b], T.state(sol, dS, dP, dg, lam0[b], KK - 1), T.state(sol, dS, dP, dg, lam0[b], KK), N, KK)
        for key, val in q.items():
            worst[key] = max(worst.get(key, 0.0), val)
    print(f"trajectory {b}: invariants of iterations 2..{K} (family {sol.get_option('last_kernel_family')}), worst measured / limit: " +
          " ".join(f"{a} {v:.2g}/{T.LIMITS[a]:g}" for a, v in worst.items()))
