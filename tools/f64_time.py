import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
for N, B in ((32, 1), (32, 1024), (128, 1), (128, 256)):
    k = synth.make_kkt(N, min(B, 16), 1)
    S0, P0, g0 = synth.form_schur(k)
    rep = (B + S0.shape[0] - 1) // S0.shape[0]
    S = torch.from_numpy(np.tile(S0, (rep, 1))[:B].astype(np.float64)).to(dev); P = torch.from_numpy(np.tile(P0, (rep, 1))[:B].astype(np.float64)).to(dev)
    g = torch.from_numpy(np.tile(g0, (rep, 1))[:B].astype(np.float64)).to(dev)
    sol = PcgSolver(N, max_batch=B)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
    lam = torch.zeros(B, 14 * N, device=dev, dtype=torch.float64)
    ts = []
    for i in range(5):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve_f64(S, P, g, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts[1:]))
    print(f"f64 N={N} B={B}: {ms:.3f} ms  {it.sum().item()/ms/1e3:.2f} M it/s  {ms*1e3/(it.sum().item()/B):.2f} us/it", flush=True)
