#!/usr/bin/env python3
"""Time the default PCG launch of the bench workload under different values of ONE handle option:
   python tools/tune_opt.py --knots 128 --batch 1024 --opt stream_cached=-1,0,1,2 [--set pcg_waves=8 --set pcg_reg_rows=3]"""
import argparse, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--knots", type=int, default=128)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--opt", required=True)
ap.add_argument("--set", action="append", default=[])
args = ap.parse_args()
N, B = args.knots, args.batch
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
lam = torch.zeros(B, 14 * N, device="cuda")
cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
for kv in args.set:
    k, v = kv.split("="); sol.set_option(k, int(v))
key, vals = args.opt.split("=")
for v in [int(x) for x in vals.split(",")]:
    sol.set_option(key, v)
    ts = []
    for i in range(args.steps + 1):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    its = int(it.sum().item()); ms = float(np.median(ts))
    print(f"N={N} B={B} {key}={v:3d}  {ms:8.3f} ms  {its / ms / 1e3:7.3f} Miter/s  checksum {float(lam.double().abs().sum()):.6e}", flush=True)
