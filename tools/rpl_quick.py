"""Row-per-lane kernel (N <= 64) against the default kernels: oracle check + timing (GPU)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import fp32_band, relinf
dev = torch.device("cuda")
for N in (2, 5, 16, 32, 33, 48, 64):
    B, K = 3, 30
    k = synth.make_kkt(N, B, 7000 + N)
    S, P, g = synth.form_schur(k, poison_unused=True)
    for pc in ("ss", "jacobi"):
        for waves in (0, 16):
            sol = PcgSolver(N, max_batch=B)
            sol.set_option("pcg_rpl", 1); sol.set_option("rpl_waves", waves)
            lam = torch.zeros(B, 14 * N, device=dev)
            it, ex = sol.solve(torch.from_numpy(S).to(dev), torch.from_numpy(P).to(dev), torch.from_numpy(g).to(dev), lam,
                               pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
            torch.cuda.synchronize()
            errs = []
            for b in range(B):
                Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
                r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), np.zeros(14 * N), N, K, 0.0, pc)
                band = fp32_band(orc, Sz, Pz, g[b], np.zeros(14 * N), N, K, pc, r64["lam"])
                errs.append((relinf(lam[b].cpu().numpy(), r64["lam"]), band))
            print("check", N, pc, "family", sol.get_option("last_kernel_family"), "waves", sol.get_option("last_kernel_waves"), "slots", sol.get_option("last_kernel_reg_rows"),
                  "iters", it.cpu().tolist(), "err/band", [(f"{e:.1e}", f"{bd:.1e}") for e, bd in errs], flush=True)

def timeit(sol, S, P, g, B, N, cfg, pc, reps=6):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, pc); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), int(it.sum().item())

for N, B, pc in ((32, 1, "jacobi"), (32, 1, "ss"), (32, 2048, "ss"), (32, 2048, "jacobi"), (64, 1, "ss"), (64, 2048, "ss"), (48, 2048, "ss"), (16, 4096, "ss")):
    k = synth.make_kkt(N, min(B, 64), 1)
    S0, P0, g0 = synth.form_schur(k)
    rep = (B + S0.shape[0] - 1) // S0.shape[0]
    S = torch.from_numpy(np.tile(S0, (rep, 1))[:B]).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B]).to(dev)
    g = torch.from_numpy(np.tile(g0, (rep, 1))[:B]).to(dev)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
    res = {}
    for name, opts in (("default", {}), ("rpl", {"pcg_rpl": 1}), ("rpl4", {"pcg_rpl": 1, "rpl_waves": 4}), ("rpl8", {"pcg_rpl": 1, "rpl_waves": 8}), ("rpl16", {"pcg_rpl": 1, "rpl_waves": 16})):
        sol = PcgSolver(N, max_batch=B)
        try:
            for kk, v in opts.items(): sol.set_option(kk, v)
            ms, its = timeit(sol, S, P, g, B, N, cfg, pc)
        except Exception as e:
            res[name] = "n/a"; continue
        res[name] = {"ms": round(ms, 4), "Mit_s": round(its / ms / 1e3, 1), "us_it": round(ms * 1e3 / (its / B), 3), "fam": sol.get_option("last_kernel_family"),
                     "w": sol.get_option("last_kernel_waves"), "slots": sol.get_option("last_kernel_reg_rows")}
    print("time", N, B, pc, json.dumps(res), flush=True)
