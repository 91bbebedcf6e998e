import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
for N,B,G in ((128,1,2),(256,1,4),(512,1,8),(256,64,4),(512,32,8),(128,128,2)):
    sol = PcgSolver(N, max_batch=B); sol.set_option("cluster", G)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    lam = torch.zeros(B, 14 * N, device="cuda")
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    for rnd in range(2):
      for adj in (0,1):
        sol.set_option("cluster_adj", adj)
        ts=[]
        for i in range(25):
            lam.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"N={N} B={B} G={G} adj={adj}: {np.median(ts[5:])*1e3:8.1f} us  min {min(ts)*1e3:8.1f}", flush=True)
