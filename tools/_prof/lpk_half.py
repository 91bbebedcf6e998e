"""float, N <= 32: the lane-pair kernel's HALF build (one wavefront per matrix, four workgroups per CU; "pcg_lpk" = 1) against the default policy's
kernel — per-iteration and fixed cost at several batches, and a parity check against the float64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench, oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config
from util import fp32_band, relinf
NS_ = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32, 24, 20]
BS_ = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512, 1024, 2048, 4096]
for N in NS_:
    for B in BS_:
        for forced in (0, 1):
            sol = PcgSolver(N, max_batch=B)
            if forced:
                sol.set_option("pcg_lpk", 1)
            dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
            lam = torch.zeros(B, 14 * N, device="cuda")
            ts = {}
            for K in (41, 81):
                cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
                def go():
                    lam.zero_(); sol.solve(dS, dP, dg, lam, cfg)
                go(); go()
                ts[K] = bench.timed(go, 5, warm=2) - bench.timed(lambda: lam.zero_(), 5, warm=2)
            per = (ts[81] - ts[41]) / 40
            fam, wv = sol.get_option("last_kernel_family"), sol.get_option("last_kernel_waves")
            if forced and B == BS_[0]:
                S, P, g = (t[:2].cpu().numpy() for t in (torch.nan_to_num(dS), torch.nan_to_num(dP), dg))
                lam2 = torch.zeros(2, 14 * N, device="cuda")
                s2 = PcgSolver(N, max_batch=2); s2.set_option("pcg_lpk", 1)
                s2.solve(dS[:2].contiguous(), dP[:2].contiguous(), dg[:2].contiguous(), lam2, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=25))
                torch.cuda.synchronize()
                for b in range(2):
                    r64 = orc.pcg(S[b].astype(np.float64), P[b].astype(np.float64), g[b].astype(np.float64), np.zeros(14 * N), N, 25, 0.0, "ss")["lam"]
                    band = fp32_band(orc, S[b], P[b], g[b], np.zeros(14 * N, np.float32), N, 25, "ss", r64, trials=2)
                    print(f"   parity: {relinf(lam2.cpu().numpy()[b], r64):.2e} (float32 band {band:.2e}), family {s2.get_option('last_kernel_family')} x {s2.get_option('last_kernel_waves')} waves")
            print(f"N={N} B={B} forced={forced} family {fam} x {wv} waves: K=41 {ts[41]*1e3:.1f} us, per iteration {per*1e3:.2f} us = {B/per/1e3:.0f} M it/s, fixed {1e3*(ts[41]-41*per):.1f} us", flush=True)
