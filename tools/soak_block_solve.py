import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from mpcgpu_amd import PcgSolver
for N, B in ((128, 300), (512, 64), (32, 3000)):
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    ref = None; bad = 0
    for wide in (0, 1):
        sol.set_option("block_solve_wide", wide)
        for i in range(200):
            lam = sol.block_solve(dS, dg)
            if i % 20 == 19:
                cur = lam.cpu().numpy()
                if ref is None: ref = cur
                elif not np.array_equal(ref, cur): bad += 1
    print(N, B, "mismatches", bad, "finite", bool(np.isfinite(ref).all()))
