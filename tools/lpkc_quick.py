"""Timing of the clustered lane-pair kernel (pcg_lpk_cluster.hip.h, family 7): fixed iteration counts = the reference's caps, batch 1024 / 64 / 1,
and the mixed-iteration (warm-start) batch.  (Correctness: tests/test_gpu_cluster.py.)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, pcg_config, synth, _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
dev = torch.device("cuda")


def timeit(sol, S, P, g, lam0, B, N, cfg, reps=7):
    lam = torch.zeros(B, 14 * N, device=dev)
    ts = []
    for i in range(reps):
        lam.copy_(lam0) if lam0 is not None else lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(S, P, g, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[2:])), it, lam


for N in (256, 512, 192, 384):
    k = synth.make_kkt(N, 32, 1)
    S0, P0, g0 = synth.form_schur(k)
    for B in (1024, 64, 1):
        rep = (B + 31) // 32
        S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).to(dev) for a in (S0, P0, g0))
        cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
        res, lams = {}, {}
        for name, v in (("lpkc", -1),):
            sol = PcgSolver(N, max_batch=B)
            ms, it, lam = timeit(sol, S, P, g, None, B, N, cfg)
            its = int(it.sum().item())
            lams[name] = lam
            res[name] = {"ms": round(ms, 4), "Mit_s": round(its / ms / 1e3, 2), "us_it": round(ms * 1e3 / synth.pcg_max_iter(N), 3), "family": sol.get_option("last_kernel_family"),
                         "G": sol.get_option("last_kernel_cluster"), "fixups": sol.get_option("cluster_fixups")}
        print("time", N, B, json.dumps(res), flush=True)
    if N in (256, 512):
        # warm-started batch: solves leave at different iterations (persistent clusters draw the next trajectory)
        B = 1024
        rep = (B + 31) // 32
        S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).to(dev) for a in (S0, P0, g0))
        sol = PcgSolver(N, max_batch=B)
        star = torch.zeros(B, 14 * N, device=dev)
        sol.solve(S, P, g, star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000), "ss")
        gen = torch.Generator(device=dev).manual_seed(7)
        amp = torch.logspace(-4, -1, B, device=dev)[torch.randperm(B, device=dev, generator=gen)].unsqueeze(1)
        lam0 = star + amp * star.abs().amax(dim=1, keepdim=True) * torch.randn(B, 14 * N, device=dev, generator=gen)
        cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
        res = {}
        for name, v in (("lpkc", -1),):
            sol = PcgSolver(N, max_batch=B)
            ms, it, _ = timeit(sol, S, P, g, lam0, B, N, cfg)
            ms -= 0.0
            res[name] = {"ms": round(ms, 4), "mean_it": float(it.float().mean().item()), "Mit_s": round(int(it.sum().item()) / ms / 1e3, 2), "linsolves_per_s": int(B / ms * 1e3)}
        print("mixed", N, B, json.dumps(res), flush=True)
