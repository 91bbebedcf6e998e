"""Which PCG kernel family wins where (policy of launch_pcg): lane-pair (lpk), lane-per-block (lpb), row-per-lane (rpl), automatic."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")


def timeit(sol, S, P, g, B, N, cfg, pc, reps=7):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, pc); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:])), int(it.sum().item())


for N in (24, 32, 36, 40, 48, 64):
    for B in (1, 256, 2048):
        k = synth.make_kkt(N, min(B, 64), 1)
        S0, P0, g0 = synth.form_schur(k)
        rep = (B + S0.shape[0] - 1) // S0.shape[0]
        S = torch.from_numpy(np.tile(S0, (rep, 1))[:B]).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B]).to(dev)
        g = torch.from_numpy(np.tile(g0, (rep, 1))[:B]).to(dev)
        for pc in ("ss", "jacobi"):
            cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
            res = {}
            for name, opts in (("auto", {}), ("lpk", {"pcg_lpk": 1}), ("rpl", {"pcg_rpl": 1}), ("traj", {"pcg_lpk": 0, "pcg_lpb": 0, "pcg_rpl": 0})):
                sol = PcgSolver(N, max_batch=B)
                for k_, v_ in opts.items():
                    sol.set_option(k_, v_)
                ms, its = timeit(sol, S, P, g, B, N, cfg, pc)
                res[name] = (round(ms, 4), round(its / ms / 1e3, 1), sol.get_option("last_kernel_family"))
            print("N", N, "B", B, pc, json.dumps(res), flush=True)
