#!/usr/bin/env python3
"""The float lane-quad kernel (pcg_lqb.hip.h, "pcg_lqb" = 1) against the lane-pair kernel it replaces: agreement at fixed iteration counts
(both against the float64 oracle iterate) and time per solve — batch 1024 / 256 / 1, N = 128 / 64 / 32, both preconditioners.
   python tools/_prof/lqb_quick.py [--check-only] [--time-only]      AB_LIB=path: another build of the library"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpcgpu_amd import PcgSolver, pcg_config, synth, _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
dev = torch.device("cuda")
relinf = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())


def check():
    import oracle
    bad = 0
    for N in (128, 100, 64, 47, 32, 20, 2, 5):
        for pc in ("ss", "jacobi"):
            B = 5
            k = synth.make_kkt(N, B, 300 + N)
            S, P, g = synth.form_schur(k, precond=pc, poison_unused=True)
            lam0 = np.random.default_rng(N).normal(0, 0.2, (B, 14 * N)).astype(np.float32)
            dS, dP, dg = (torch.from_numpy(x).to(dev) for x in (S, P, g))
            for K in (0, 1, 2, 10, 40):
                out = {}
                for name, v in (("lpk", 0), ("lqb", 1)):
                    sol = PcgSolver(N, max_batch=B)
                    sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", v); sol.set_option("assume_symmetric", 1)
                    lam = torch.from_numpy(lam0.copy()).to(dev)
                    r = torch.zeros(B, 14 * N, device=dev); p = torch.zeros(B, 14 * N, device=dev)
                    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
                    torch.cuda.synchronize()
                    out[name] = (lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy(), sol.get_option("last_kernel_family"))
                e = []
                for t in range(B):
                    ref = oracle.pcg(np.nan_to_num(S[t]).astype(np.float64), np.nan_to_num(P[t]).astype(np.float64), g[t].astype(np.float64),
                                     lam0[t].astype(np.float64), N, K, 0.0, pc)["lam"]
                    e.append((relinf(out["lpk"][0][t], ref), relinf(out["lqb"][0][t], ref)))
                e = np.array(e)
                ok = out["lqb"][3] == 11 and (out["lqb"][1] == K).all() and (out["lqb"][2] == (1 if K else 1)).all() and e[:, 1].max() <= max(3 * e[:, 0].max(), 3e-6)
                bad += not ok
                print(f"check N={N:3d} {pc:6s} K={K:3d} family {out['lqb'][3]} iters {out['lqb'][1][:2]} exit {out['lqb'][2][:2]}  err vs f64 oracle: lpk {e[:, 0].max():.2e}  lqb {e[:, 1].max():.2e}  {'ok' if ok else 'BAD'}", flush=True)
    # tolerance exits: same counts (+-2), d_r / d_p through the reference-shaped entry
    print("bad:", bad)
    return bad


def timeit(sol, S, P, g, B, N, cfg, pc, reps=9):
    lam = torch.zeros(B, 14 * N, device=dev)
    for _ in range(3):
        lam.zero_(); sol.solve(S, P, g, lam, cfg, pc)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, pc); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), it


def times():
    for N in (128, 64, 32, 96):
        k = synth.make_kkt(N, 32, 1)
        for pc in ("ss", "jacobi"):
            S0, P0, g0 = synth.form_schur(k, precond=pc)
            for B in (1024, 2048, 256, 1):
                rep = (B + 31) // 32
                S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).to(dev) for a in (S0, P0, g0))
                res = {}
                for K in (synth.pcg_max_iter(N), 20):
                    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
                    for name, v in (("lpk", 0), ("lqb", 1)):
                        sol = PcgSolver(N, max_batch=B)
                        sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", v); sol.set_option("assume_symmetric", 1)
                        ms, it = timeit(sol, S, P, g, B, N, cfg, pc)
                        res[f"{name}@{K}"] = round(ms, 4)
                Kc = synth.pcg_max_iter(N)
                per = {n_: (res[f"{n_}@{Kc}"] - res[f"{n_}@20"]) / (Kc - 20) * 1e3 for n_ in ("lpk", "lqb")}       # us per iteration of the batch
                print(f"time N={N:3d} {pc:6s} B={B:4d}: " + json.dumps(res) + f"  us/it lpk {per['lpk']:.3f} lqb {per['lqb']:.3f}  "
                      f"Mit/s@cap lpk {B * Kc / res[f'lpk@{Kc}'] / 1e3:.1f} lqb {B * Kc / res[f'lqb@{Kc}'] / 1e3:.1f}", flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    if "--check-only" not in sys.argv:
        times()
    sys.exit(1 if rc else 0)
