#!/usr/bin/env python3
"""BASELINE config 5: IIWA-14 N=512 long horizon, fp32 vs fp16 MATRIX STORAGE, exit-tolerance sweep
(tolerances of examples/track_iiwa_pcg.cu:62-68, max_iter 67 of settings.cuh:135).  For every tolerance:
iterations, exit-on-max-iter rate, TRUE relative residual ||gamma - S lambda|| / ||gamma|| against the
ORIGINAL fp32 system (float64 evaluation on the host for a sample), throughput.  Prints one JSON object.
   python tools/config5_sweep.py [--knots 512] [--batch 256] [--warm]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--knots", type=int, default=512)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--max-iter", type=int, default=0)
ap.add_argument("--sample", type=int, default=4)
args = ap.parse_args()
N, B = args.knots, args.batch
max_iter = args.max_iter or synth.pcg_max_iter(N)
dev = torch.device("cuda", 0)
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", dev)
S16, P16 = sol.to_f16(dS), sol.to_f16(dP)
Sh, gh = dS[: args.sample].cpu().numpy(), dg[: args.sample].cpu().numpy()
Sd = [synth.bd_to_dense(np.nan_to_num(Sh[b]), N) for b in range(args.sample)]


def true_res(lam):
    lam = lam[: args.sample].cpu().numpy().astype(np.float64)
    return [float(np.linalg.norm(gh[b] - Sd[b] @ lam[b]) / np.linalg.norm(gh[b])) for b in range(args.sample)]


out = {"workload": f"IIWA-14 N={N}, SS preconditioner, batch {B}, lambda0=0", "sweeps": []}
cases = [(synth.RHO_INIT, mi, tol) for mi in (max_iter, 2000) for tol in (1e-5, 5e-5, 1e-4, 5e-4, 1e-3)]
cases += [(rho, 2000, 1e-4) for rho in (1e-2, 1e-1, 1.0, 10.0)]      # rho adapts in [1e-3, 10] (settings.cuh:185-196)
cur_rho = synth.RHO_INIT
for rho, max_it, tol in cases:
    if rho != cur_rho:
        dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", dev, rho=rho)
        S16, P16 = sol.to_f16(dS), sol.to_f16(dP)
        Sh, gh = dS[: args.sample].cpu().numpy(), dg[: args.sample].cpu().numpy()
        Sd = [synth.bd_to_dense(np.nan_to_num(Sh[b]), N) for b in range(args.sample)]
        cur_rho = rho
    if True:
        row = {"rho": rho, "max_iter": max_it, "exit_tol": tol}
        for name in ("f32", "f16"):
            cfg = pcg_config(pcg_exit_tol=tol, pcg_max_iter=max_it)
            lam = torch.zeros(B, 14 * N, device=dev)
            ts = []
            for rep in range(3):
                lam.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                it, ex = (sol.solve(dS, dP, dg, lam, cfg) if name == "f32" else sol.solve_f16(S16, P16, dg, lam, cfg))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            itn = it.cpu().numpy()
            row[name] = {"mean_iters": float(itn.mean()), "max_iter_exit_rate": float(ex.float().mean().item()),
                         "true_rel_residual_sample": true_res(lam), "ms": float(np.median(ts)),
                         "pcg_iters_per_sec": float(itn.sum() / (np.median(ts) * 1e-3))}
        out["sweeps"].append(row)
print(json.dumps(out))
