import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
for N in (64, 48, 36):
    k = synth.make_kkt(N, 8, 1)
    for pc in ("ss", "jacobi"):
        S0, P0, g0 = synth.form_schur(k, precond=pc)
        for B in (1, 256):
            S, P, g = (torch.from_numpy(np.tile(a, ((B + 7) // 8, 1))[:B]).to(dev) for a in (S0, P0, g0))
            row = []
            for KK in (synth.pcg_max_iter(N), 26):
                cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=KK)
                for lqb in (0, -1):
                    sol = PcgSolver(N, max_batch=B)
                    sol.set_option("assume_symmetric", 1); sol.set_option("pcg_lqb", lqb)
                    lam = torch.zeros(B, 14 * N, device=dev)
                    t0 = time.time()
                    while time.time() - t0 < 0.05:
                        for _ in range(10): sol.solve(S, P, g, lam, cfg, pc)
                        torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): sol.solve(S, P, g, lam, cfg, pc)
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 20
                    row.append(f"{'round-5 policy' if lqb == 0 else 'default'}({sol.get_option('last_kernel_family')})@{KK}: {ms:.4f}")
            print(f"N={N} {pc:6s} B={B}: " + " | ".join(row), flush=True)
