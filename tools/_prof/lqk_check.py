"""lane-quad double kernel against the oracle (fixed K, warm start, tolerance exit) + rate."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from mpcgpu_amd import PcgSolver, pcg_config, synth
import oracle as orc
n = 14
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def relinf(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
worst = 0.0
NS_ = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [64, 33, 40, 57, 32, 16, 5]
for N in NS_:
    for pc in ("ss", "jacobi"):
        B, K = 3, 30
        k = synth.make_kkt(N, B, 6100 + N)
        S, Pinv, g = synth.form_schur(k, precond=pc, dtype=np.float64)
        dS, dP, dg = dev(S), dev(Pinv), dev(g)
        sol = PcgSolver(N, max_batch=B)
        if N <= 32: sol.set_option("pcg_lqk", 1)
        rng = np.random.default_rng(N)
        for lam0 in (np.zeros((B, n * N)), 0.1 * rng.standard_normal((B, n * N))):
            lam = dev(lam0.copy())
            r = torch.zeros(B, n * N, dtype=torch.float64, device="cuda"); p = torch.zeros_like(r)
            it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
            torch.cuda.synchronize()
            fam = sol.get_option("last_kernel_family")
            e = 0.0
            for b in range(B):
                ref = orc.pcg(S[b], Pinv[b], g[b], lam0[b], N, K, 0.0, pc)
                e = max(e, relinf(lam.cpu().numpy()[b], ref["lam"]))
            worst = max(worst, e)
            print(f"N={N} {pc} fam={fam} G={sol.get_option('last_kernel_cluster')} fixups={sol.get_option('cluster_fixups')} iters={it.cpu().numpy()} exits={ex.cpu().numpy()} relinf={e:.2e}", flush=True)
print("worst", worst)
