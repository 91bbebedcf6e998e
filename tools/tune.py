#!/usr/bin/env python3
"""Sweep the launch knobs of the persistent PCG kernel on the bench workload (run on the GPU box):
   python tools/tune.py [--batch 1024] [--knots 128] > gpurun_out/tune.txt"""
import argparse, itertools, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--knots", type=int, default=128)
ap.add_argument("--precond", default="ss")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--waves", default="4,8,16")
ap.add_argument("--wgs", default="0,1,2,3,4")
ap.add_argument("--nts", default="0,1")
ap.add_argument("--regs", default="0", help="comma list of waves:reg_rows[:lds_rows] variants, e.g. 8:6:-1,4:12:-1")
args = ap.parse_args()
N, B = args.knots, args.batch
sol0 = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol0, N, B, 0, args.precond, torch.device("cuda", 0))
lam = torch.zeros(B, 14 * N, device="cuda")
cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
bytes_it = synth.algorithmic_bytes(N, precond=args.precond)["pcg_iter"]
sol = PcgSolver(N, max_batch=B)
combos = []
if args.regs != "0":
    for spec in args.regs.split(","):
        f = [int(x) for x in spec.split(":")]
        for nt in [int(x) for x in args.nts.split(",")]:
            combos.append((f[0], 0, nt, f[1], f[2] if len(f) > 2 else -1))
else:
    for waves, wg, nt in itertools.product([int(x) for x in args.waves.split(",")], [int(x) for x in args.wgs.split(",")],
                                           [int(x) for x in args.nts.split(",")]):
        combos.append((waves, wg, nt, 0, -1))
for waves, wg, nt, rr, rl in combos:
    sol.set_option("pcg_waves", waves); sol.set_option("pcg_max_wg_per_cu", wg); sol.set_option("nt_loads", nt)
    sol.set_option("pcg_reg_rows", rr); sol.set_option("pcg_lds_rows", rl)
    try:
        res = sol.checkPcgOccupancy()
    except Exception as e:
        print(f"waves={waves} wg/cu={wg} nt={nt}: {e}"); continue
    ts = []
    for i in range(args.steps + 1):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg, args.precond); e1.record()
        torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    its = int(it.sum().item())
    ms = float(np.median(ts))
    print(f"waves={waves:2d} wg/cu={wg} nt={nt} regrows={rr:2d} ldsrows={rl:2d} resident={res:5d}  {ms:8.3f} ms  {its/ms/1e3:7.3f} Miter/s  "
          f"{its*bytes_it/ms/1e6:8.1f} GB/s algorithmic", flush=True)
