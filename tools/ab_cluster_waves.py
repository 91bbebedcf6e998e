import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
for N,B in ((256,64),(256,512),(512,32),(512,256),(512,1024),(192,512)):
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0))
    lam = torch.zeros(B, 14 * N, device="cuda")
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    ref=None
    for cw in (8,4,-1):
        sol.set_option("cluster_waves", cw)
        ts=[]
        for i in range(8):
            lam.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        itn=it.cpu().numpy().astype(np.int64); ok = (itn < 1<<30).all()
        ms=float(np.median(ts[2:]))
        cur=lam.cpu().numpy()
        d = 0.0 if ref is None else float(np.abs(cur-ref).max()/np.abs(ref).max())
        if ref is None: ref=cur
        print(f"N={N} B={B} cluster_waves={cw:2d}: {ms:8.3f} ms  {itn.sum()/ms/1e3:7.3f} Miter/s  {'ok' if ok else 'TIMEOUT'}  rel diff vs 8-wave {d:.1e}", flush=True)
