import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
for N in (36, 40, 44, 48, 56, 64):
    B = 2048
    k = synth.make_kkt(N, 64, 1)
    S0, P0, g0 = synth.form_schur(k)
    S = torch.from_numpy(np.tile(S0, (B // 64, 1))).to(dev); P = torch.from_numpy(np.tile(P0, (B // 64, 1))).to(dev); g = torch.from_numpy(np.tile(g0, (B // 64, 1))).to(dev)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
    res = {}
    for name, opts in (("default", {}), ("lpb", {"pcg_lpb": 1}), ("traj", {"pcg_lpb": 0, "pcg_rpl": 0}), ("rpl", {"pcg_rpl": 1})):
        sol = PcgSolver(N, max_batch=B)
        for kk, v in opts.items(): sol.set_option(kk, v)
        lam = torch.zeros(B, 14 * N, device=dev); ts = []
        for i in range(5):
            lam.zero_(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, "ss"); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[1:]))
        res[name] = (round(it.sum().item() / ms / 1e3, 1), sol.get_option("last_kernel_family"))
    print(N, res, flush=True)
