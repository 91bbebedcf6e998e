#!/usr/bin/env python3
"""Direct sweep as the starting point of PCG: mpcg_block_solve, then mpcg_pcg_solve from that lambda to the reference's
exit tolerances — iterations needed, time, true residual (N=128, SS, batch 1024, the bench systems)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
N, B = 128, 1024
dev = torch.device("cuda", 0)
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", dev)
ns = 4
Sd = [synth.bd_to_dense(np.nan_to_num(dS[b].cpu().numpy()), N) for b in range(ns)]
gh = dg[:ns].cpu().numpy()
def res(lam):
    l = lam[:ns].cpu().numpy().astype(np.float64)
    return max(float(np.linalg.norm(gh[b] - Sd[b] @ l[b]) / np.linalg.norm(gh[b])) for b in range(ns))
lam = torch.empty(B, 14 * N, device=dev)
def t(fn, reps=5):
    ts = []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), out
ms, _ = t(lambda: sol.block_solve(dS, dg, lam))
print(f"block_solve alone: {ms:.3f} ms, true rel residual {res(lam):.2e}")
for tol in (1e-3, 1e-4, 1e-5, 1e-6):
    cfg = pcg_config(pcg_exit_tol=tol, pcg_max_iter=2000)
    def hybrid():
        sol.block_solve(dS, dg, lam)
        return sol.solve(dS, dP, dg, lam, cfg)
    ms, (it, ex) = t(hybrid)
    itn = it.cpu().numpy()
    def cold():
        lam.zero_()
        return sol.solve(dS, dP, dg, lam, cfg)
    r_h = res(lam) if False else None
    hybrid(); torch.cuda.synchronize(); r_h = res(lam)
    ms_c, (itc, exc) = t(cold)
    print(f"tol {tol:g}: direct+PCG {ms:.3f} ms, PCG iterations mean {itn.mean():.1f} max {itn.max()}, residual {r_h:.2e} | "
          f"PCG from zero {ms_c:.3f} ms, iterations mean {itc.cpu().numpy().mean():.1f}, residual {res(lam):.2e}")
