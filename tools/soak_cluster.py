#!/usr/bin/env python3
"""Soak test of the cluster kernel's cross-workgroup hand-offs: many launches, every trajectory must finish with a
valid iteration count (0xFFFFFFFF = a bounded spin timed out) and bit-identical results from launch to launch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# optional 2nd argument: "wt" = write-through hand-offs instead of L2-resident ones
mode = sys.argv[2] if len(sys.argv) > 2 else "l2"
for N, B in ((128, 100), (256, 64), (512, 32), (512, 256), (256, 512), (192, 300), (640, 51)):
    sol = PcgSolver(N, max_batch=B)
    if mode == "wt": sol.set_option("cluster_l2", 0)
    if N <= 128: sol.set_option("cluster", 2)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    lam = torch.zeros(B, 14 * N, device="cuda")
    ref = None; bad = 0; t0 = time.time()
    for i in range(reps):
        lam.zero_()
        it, ex = sol.solve(dS, dP, dg, lam, cfg)
        if i % 10 == 9 or i == reps - 1:
            torch.cuda.synchronize()
            itn = it.cpu().numpy().astype(np.int64)
            bad += int((itn >= 1 << 30).sum())
            cur = lam.cpu().numpy()
            if ref is None: ref = cur
            elif not np.array_equal(ref, cur): bad += 1000000
    print(f"N={N} batch={B}: {reps} launches, kernel family {sol.get_option('last_kernel_family')} x {sol.get_option('last_kernel_cluster')} members, "
          f"failures={bad}, fix-up launches that re-solved a trajectory={sol.get_option('cluster_fixups')} ({time.time()-t0:.1f} s)", flush=True)
