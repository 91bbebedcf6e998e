"""Violation history of six full-step SQP iterations on perturbed IIWA windows (the chain of tests/test_gpu_kkt.py), several seeds:
how much the end state moves with the build of the library (AB_LIB) — i.e. with 1e-7-level differences in the KKT blocks."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, Plant, iiwa, pcg_config, synth
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
N, B = 32, 8
plant = Plant()
for seed in (11, 12, 13, 14):
    xu, goals, xs = iiwa.random_windows(N, B, seed)
    sol = PcgSolver(N, max_batch=B)
    d_goals, d_xs, d_xu = dev(goals.reshape(B, -1)), dev(xs), dev(xu)
    lam = torch.zeros(B, 14 * N, device="cuda")
    viol = []
    for it in range(6):
        G, C, g, c = sol.generate_kkt(plant, d_goals, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
        viol.append(c.abs().amax(dim=1).cpu().numpy().astype(np.float64))
        S, Pinv, gam = sol.form_schur(G, C, g, c, synth.RHO_INIT, "ss")
        sol.solve(S, Pinv, gam, lam, pcg_config(pcg_exit_tol=1e-7, pcg_max_iter=3000), "ss")
        d_xu = d_xu - sol.compute_dz(G, C, g, lam)
    viol = np.array(viol)
    print(seed, "median viol per iteration:", " ".join(f"{np.median(v):.4f}" for v in viol), " ratio[5]/[0] %.3f ratio[2]/[0] %.3f" % (np.median(viol[5]) / np.median(viol[0]), np.median(viol[2]) / np.median(viol[0])))
