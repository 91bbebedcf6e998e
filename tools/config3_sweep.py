#!/usr/bin/env python3
"""BASELINE config 3 / 4 companion: IIWA-14 N=128, SS preconditioner, exit-tolerance sweep
(tolerances of examples/track_iiwa_pcg.cu:62-68; max_iter 167 of settings.cuh:131, and 2000 to see where the
solves actually converge), cold start (lambda0 = 0) and warm start (lambda0 = previous solution of a perturbed
system, as the MPC loop provides: include/mpcsim.cuh:186,267,337).  For every case: iterations, exit-on-max-iter
rate, TRUE relative residual ||gamma - S lambda|| / ||gamma|| (float64 on the host for a sample), throughput.
   python tools/config3_sweep.py [--knots 128] [--batch 1024] > profiles/r01_config3_tol_sweep.json"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--knots", type=int, default=128)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--sample", type=int, default=4)
args = ap.parse_args()
N, B = args.knots, args.batch
dev = torch.device("cuda", 0)
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", dev)
Sh, gh = dS[: args.sample].cpu().numpy(), dg[: args.sample].cpu().numpy()
Sd = [synth.bd_to_dense(np.nan_to_num(Sh[b]), N) for b in range(args.sample)]


def true_res(lam, g):
    lam = lam[: args.sample].cpu().numpy().astype(np.float64)
    return [float(np.linalg.norm(g[b] - Sd[b] @ lam[b]) / np.linalg.norm(g[b])) for b in range(args.sample)]


# warm start: the converged solution for gamma, used as lambda0 for gamma' = gamma + 2 % noise
lam_ref = torch.zeros(B, 14 * N, device=dev)
sol.solve(dS, dP, dg, lam_ref, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000))
gen = torch.Generator(device=dev); gen.manual_seed(7)
dg2 = dg * (1 + 0.02 * torch.randn(dg.shape, device=dev, generator=gen))
gh2 = dg2[: args.sample].cpu().numpy()
out = {"workload": f"IIWA-14 N={N}, SS preconditioner, batch {B}, rho={synth.RHO_INIT}", "sweeps": []}
for start in ("cold", "warm"):
    for max_it in (synth.pcg_max_iter(N), 2000):
        for tol in (1e-5, 5e-5, 1e-4, 5e-4, 1e-3):
            cfg = pcg_config(pcg_exit_tol=tol, pcg_max_iter=max_it)
            rhs, rh = (dg, gh) if start == "cold" else (dg2, gh2)
            lam = torch.zeros(B, 14 * N, device=dev)
            ts = []
            for rep in range(3):
                if start == "cold": lam.zero_()
                else: lam.copy_(lam_ref)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); it, ex = sol.solve(dS, dP, rhs, lam, cfg); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            itn, exn = it.cpu().numpy(), ex.cpu().numpy()
            ms = float(np.median(ts))
            out["sweeps"].append({"start": start, "max_iter": max_it, "exit_tol": tol, "mean_iters": float(itn.mean()),
                                  "min_iters": int(itn.min()), "max_iters": int(itn.max()), "max_iter_exit_rate": float(exn.mean()),
                                  "ms_per_batch": ms, "pcg_iterations_per_sec": float(itn.sum() / ms * 1e3),
                                  "us_per_linsolve_throughput": ms * 1e3 / B,
                                  "true_rel_residual_sample": true_res(lam, rh)})
print(json.dumps(out, indent=1))
