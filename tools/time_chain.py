#!/usr/bin/env python3
"""One SQP linear-system step for a batch, on the device, as a hipGraph: KKT blocks -> mpcg_form_schur -> solver ->
mpcg_compute_dz (include/pcg/sqp.cuh:207-259), with PCG (the reference's iteration cap, cold) or the direct sweep."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, pcg_config, synth
n, m = 14, 7
for N, B in ((32, 1024), (128, 128), (128, 1024)):
    sol = PcgSolver(N, max_batch=B)
    parts = [synth.pack_kkt_dense(synth.make_kkt(N, min(B, 128), 50 + i), np.float32) for i in range((B + 127) // 128)]
    G, C, g, c = (torch.from_numpy(np.concatenate([p[i] for p in parts])[:B]).cuda() for i in range(4))
    dG = torch.empty_like(G)
    S = torch.empty(B, 3 * n * n * N, device="cuda"); P = torch.empty_like(S)
    gam = torch.empty(B, n * N, device="cuda"); lam = torch.empty(B, n * N, device="cuda")
    dz = torch.empty(B, (n + m) * N - m, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
    def chain(direct):
        dG.copy_(G)
        sol.form_schur(dG, C, g, c, 1e-3, "none" if direct else "ss", S=S, Pinv=P, gamma=gam)
        if direct: sol.block_solve(S, gam, lam)
        else:
            lam.zero_(); sol.solve(S, P, gam, lam, cfg, "ss", iters=it, exits=ex)
        sol.compute_dz(dG, C, g, lam, dz=dz)
    for direct in (False, True):
        chain(direct); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr): chain(direct)
        ts = []
        for i in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[2:]))
        print(f"N={N:3d} batch={B:4d} {'direct sweep' if direct else 'PCG (cap, cold)':16s}: {ms:7.3f} ms per batch step = {ms * 1e3 / B:6.2f} us per trajectory-step "
              f"({B / ms * 1e3:9.0f} linear-system steps/s)", flush=True)
