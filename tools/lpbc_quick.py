"""Timing of the clustered lane-per-block kernel against the row-triple cluster kernel for long horizons (GPU)."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpcgpu_amd._lib as _L
if os.environ.get("AB_LIB"):                      # A/B against another build of the library (tools/_prof/ab/)
    _L.LIB_PATH = os.environ["AB_LIB"]
from mpcgpu_amd import PcgSolver, pcg_config, synth

dev = torch.device("cuda")

def timeit(sol, S, P, g, B, N, cfg, reps=6):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    assert int(ex.max().item()) <= 1, "a cluster timed out"
    return float(np.median(ts[1:])), int(it.sum().item())

cases = [(256, 1024), (256, 128), (256, 1), (512, 1024), (512, 64), (512, 1), (192, 1024), (192, 1)]
if len(sys.argv) > 1:
    cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for N, B in cases:
    k = synth.make_kkt(N, min(B, 16), 1)
    S0, P0, g0 = synth.form_schur(k)
    rep = (B + S0.shape[0] - 1) // S0.shape[0]
    S = torch.from_numpy(np.tile(S0, (rep, 1))[:B]).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B]).to(dev)
    g = torch.from_numpy(np.tile(g0, (rep, 1))[:B]).to(dev)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
    res = {}
    for name, opts in (("lpbc8", {"cluster_lpb": 1, "cluster_waves": 8}), ("lpbc8_wt", {"cluster_lpb": 1, "cluster_waves": 8, "cluster_l2": 0}), ("lpbc4", {"cluster_lpb": 1, "cluster_waves": 4}),
                       ("triple", {"cluster_lpb": 0}), ("single", {"cluster": 0})):
        sol = PcgSolver(N, max_batch=B)
        for kk, v in opts.items(): sol.set_option(kk, v)
        ms, its = timeit(sol, S, P, g, B, N, cfg)
        res[name] = {"ms": round(ms, 4), "Mit_s": round(its / ms / 1e3, 2), "us_it": round(ms * 1e3 / (its / B), 3), "family": sol.get_option("last_kernel_family"),
                     "waves": sol.get_option("last_kernel_waves"), "G": sol.get_option("last_kernel_cluster")}
    print("time", N, B, json.dumps(res), flush=True)

# mixed iteration counts (MPC-style warm starts): lambda0 = converged solution + noise of log-uniform relative amplitude, tolerance exit
if os.environ.get("LPBC_MIXED", "1") == "1":
    for N, B in ((256, 1024), (512, 1024)):
        k = synth.make_kkt(N, 16, 1)
        S0, P0, g0 = synth.form_schur(k)
        rep = B // 16
        S = torch.from_numpy(np.tile(S0, (rep, 1))).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))).to(dev)
        g = torch.from_numpy(np.tile(g0, (rep, 1))).to(dev)
        sol = PcgSolver(N, max_batch=B)
        lam = torch.zeros(B, 14 * N, device=dev)
        sol.solve(S, P, g, lam, pcg_config(pcg_exit_tol=1e-7, pcg_max_iter=3000), "ss")
        amp = torch.exp(torch.empty(B, 1, device=dev).uniform_(np.log(1e-4), np.log(1e-1)))
        gen = torch.Generator(device=dev); gen.manual_seed(3)
        lam0 = lam + amp * lam.abs().mean() * torch.randn(lam.shape, device=dev, generator=gen)
        cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
        l = torch.empty_like(lam0)
        ts = []
        for i in range(6):
            l.copy_(lam0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); it, ex = sol.solve(S, P, g, l, cfg, "ss"); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        itn = it.cpu().numpy()
        ms = float(np.median(ts[1:]))
        print("mixed", N, B, json.dumps({"ms": round(ms, 4), "mean_it": float(itn.mean()), "min_it": int(itn.min()), "max_it": int(itn.max()),
                                         "Mit_s": round(itn.sum() / ms / 1e3, 2), "linsolves_per_s": round(B / ms * 1e3), "family": sol.get_option("last_kernel_family")}), flush=True)
