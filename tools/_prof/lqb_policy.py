#!/usr/bin/env python3
"""Policy sweep for the float lane-quad kernel: what the default policy runs today against "pcg_lpk" = 1 + "pcg_lqb" = 1, solve time at the
reference's iteration cap and at 26 iterations (the warm regime), SS and block-Jacobi."""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
Ns = [int(x) for x in sys.argv[1:]] or [20, 24, 32, 36, 40, 48, 64, 80, 96, 128]
for N in Ns:
    k = synth.make_kkt(N, 8, 1)
    for pc in ("ss", "jacobi"):
        S0, P0, g0 = synth.form_schur(k, precond=pc)
        for B in (1, 64, 256, 640, 1024, 4096):
            rep = (B + 7) // 8
            S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).to(dev) for a in (S0, P0, g0))
            row = {}
            for name in ("auto", "lqb"):
                sol = PcgSolver(N, max_batch=B)
                sol.set_option("assume_symmetric", 1)
                if name == "lqb":
                    sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", 1)
                for KK in (synth.pcg_max_iter(N), 26):
                    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=KK)
                    lam = torch.zeros(B, 14 * N, device=dev)
                    for _ in range(3):
                        lam.zero_(); sol.solve(S, P, g, lam, cfg, pc)
                    ts = []
                    for i in range(9):
                        lam.zero_()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); sol.solve(S, P, g, lam, cfg, pc); e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1))
                    row[f"{name}@{'cap' if KK > 26 else 26}"] = round(float(np.median(ts)), 4)
                row[name + "_fam"] = (sol.get_option("last_kernel_family"), sol.get_option("last_kernel_waves"))
            print(f"N={N:3d} {pc:6s} B={B:4d}: cap auto {row['auto@cap']:.4f} lqb {row['lqb@cap']:.4f} ({row['auto@cap'] / row['lqb@cap']:.3f}x) | 26 it auto {row['auto@26']:.4f} lqb {row['lqb@26']:.4f} "
                  f"({row['auto@26'] / row['lqb@26']:.3f}x)  auto family {row['auto_fam']}", flush=True)
