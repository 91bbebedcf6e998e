import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import golden, rel_residual, fp32_iters_band
N = 32
G = golden(N)
rng = np.random.default_rng(0)
for pc in ("jacobi", "ss"):
    print(pc, "cpu band", fp32_iters_band(orc, G["S"], G["Pinv"], G["gamma"], np.zeros(14 * N), N, 5000, 1e-4, pc), "f64 want", int(G[f"iters_tol_{pc}"]))
    for trial in range(6):
        S = G["S"].astype(np.float64); g = G["gamma"].astype(np.float64)
        if trial:
            S = S * (1 + 6e-8 * rng.standard_normal(S.shape)); g = g * (1 + 6e-8 * rng.standard_normal(g.shape))
        S = S.astype(np.float32); g = g.astype(np.float32)
        row = []
        for lpb in (-1, 0):
            sol = PcgSolver(N, max_batch=1)
            sol.set_option("pcg_lpb", lpb)
            lam = torch.zeros(1, 14 * N, device="cuda")
            it, ex = sol.solve(torch.from_numpy(S).cuda().view(1, -1), torch.from_numpy(G["Pinv"]).cuda().view(1, -1), torch.from_numpy(g).cuda().view(1, -1), lam,
                               pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=5000), pc)
            torch.cuda.synchronize()
            row.append((sol.get_option("last_kernel_family"), int(it.item()), float(rel_residual(S, g, lam.cpu().numpy()[0], N))))
        c = orc.pcg(S, G["Pinv"], g, np.zeros(14 * N, np.float32), N, 5000, 1e-4, pc)
        print("  trial", trial, "lpb", row[0], "traj", row[1], "cpu32", c["iters"], float(rel_residual(S, g, c["lam"], N)))
