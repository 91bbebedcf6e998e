#!/usr/bin/env python3
"""linsys_t = double: PCG iterations per second of mpcg_pcg_solve_f64 at batch 1024 (fixed iteration count, exit_tol 0) against the HBM streaming
ceiling of kernels that re-read S and Pinv every iteration."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config, _lib as _L
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
for N in (16, 32, 64, 128, 256):
    sol = PcgSolver(N, max_batch=B)
    S, P, g = bench.build_inputs(sol, N, B, 7, "ss", dev, chunk=64)
    S64, P64, g64 = torch.nan_to_num(S).double(), torch.nan_to_num(P).double(), g.double()
    it = torch.zeros(B, dtype=torch.int32, device=dev); ex = torch.zeros(B, dtype=torch.uint8, device=dev)
    iters = 40
    cfg = pcg_config(pcg_max_iter=iters, pcg_exit_tol=0.0)
    lam = torch.zeros(B, 14 * N, dtype=torch.float64, device=dev)
    def go():
        lam.zero_(); sol.solve_f64(S64, P64, g64, lam, cfg, "ss", iters=it, exits=ex)
    ms = bench.timed(go, 5, warm=2)
    cols = 2 if (N > 32 and sol.get_option("symmetry_state") == 1) else 3      # the streaming kernel skips the right block column once the latch allows
    bytes_it = 2 * cols * 196 * N * 8
    print("N=%3d batch %d: %.3f ms per %d iterations -> %.2f M it/s; streaming model (%d block columns) %.0f GB/s (%.2f of 8 TB/s); kernel family %s" % (
        N, B, ms, iters, B * iters / ms / 1e3, cols, B * iters * bytes_it / ms / 1e6, B * iters * bytes_it / ms / 1e6 / 8000, sol.get_option("last_kernel_family") if hasattr(sol, "get_option") else "?"))
