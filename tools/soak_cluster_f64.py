#!/usr/bin/env python3
"""The clustered double kernels (lane-quad clusters, pcg_lqk_cluster_f64.hip.h; row-per-lane clusters with "pcg_lqk" = 0, pcg_rpl_cluster_f64.hip.h) next to a memory-streaming kernel on another stream (some CUs busy, L2
under pressure, members possibly not co-resident): every call must return valid results — bit-identical to the undisturbed run where the clusters
produced them, to round-off where the streaming fix-up had to step in; and no trajectory abandoned when nothing disturbs the call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
Ns, Bs = 128, 2048
spm = PcgSolver(Ns, max_batch=Bs)
S0, P0, g0 = bench.build_inputs(spm, Ns, 256, 0, "ss", dev)
Sb = S0.repeat(Bs // 256, 1).contiguous(); del S0, P0
xb = torch.randn(Bs, 14 * Ns, device=dev); yb = torch.empty_like(xb)
for N, B, lqk in ((128, 300, -1), (256, 80, -1), (512, 40, -1), (64, 200, 0), (128, 64, 0), (128, 300, 0), (256, 40, 0)):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("pcg_lqk", lqk)
    dS, dP, dg = (t.double() for t in bench.build_inputs(sol, N, B, 0, "ss", dev))
    dS, dP = torch.nan_to_num(dS), torch.nan_to_num(dP)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=40)
    lam = torch.zeros(B, 14 * N, dtype=torch.float64, device=dev)
    for quiet in range(20):
        lam.zero_(); sol.solve_f64(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == (10 if lqk else 8) and sol.get_option("cluster_fixups") == 0, "fix-ups in an undisturbed run"
    ref = lam.clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    same = close = bad = 0
    t0 = time.time()
    for i in range(reps):
        with torch.cuda.stream(s1):
            for _ in range(3): spm.bt_spmv(Sb, xb, yb)
        with torch.cuda.stream(s2):
            lam.zero_()
            it, ex = sol.solve_f64(dS, dP, dg, lam, cfg)
        torch.cuda.synchronize()
        itn = it.cpu().numpy().astype(np.int64)
        if (itn != 40).any() or (ex.cpu().numpy() > 1).any(): bad += 1
        elif torch.equal(lam, ref): same += 1
        elif float((lam - ref).abs().max() / ref.abs().max()) < 1e-9: close += 1
        else: bad += 1
    print(f"double N={N} batch={B} kernel family {sol.get_option('last_kernel_family')}: {reps} calls next to a 630 MB SpMV stream: bit-identical {same}, fix-up within round-off {close}, BAD {bad}, "
          f"trajectories left to the fix-up {sol.get_option('cluster_fixups')} ({time.time()-t0:.1f} s)", flush=True)
    assert bad == 0
