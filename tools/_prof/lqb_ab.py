#!/usr/bin/env python3
"""Per-iteration time of the lane-quad kernel ("pcg_lqb" = 1) and the lane-pair kernel across builds of the library:
   python tools/_prof/lqb_ab.py [lib.so ...]     N = 128 / 64, SS, batch 1 / 256 / 1024: us per iteration (from solves of 167 and 20 iterations)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from mpcgpu_amd import _lib as _L
    if sys.argv[2] != "-":
        _L.LIB_PATH = sys.argv[2]
    from mpcgpu_amd import PcgSolver, pcg_config, synth
    dev = torch.device("cuda")
    out = {}
    import time
    for N in ((128, 64) if os.environ.get('LQB_AB_N64') else (128,)):
        k = synth.make_kkt(N, 8, 1)
        for pc in (("ss", "jacobi") if os.environ.get('LQB_AB_JACOBI') else ("ss",)):
            S0, P0, g0 = synth.form_schur(k, precond=pc)
            for B in (1, 256, 1024):
                rep = (B + 7) // 8
                S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).to(dev) for a in (S0, P0, g0))
                for name, v in (("lpk", 0), ("lqb", 1)):
                    t = {}
                    for K in (167, 20):
                        sol = PcgSolver(N, max_batch=B)
                        sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", v); sol.set_option("assume_symmetric", 1)
                        cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
                        lam = torch.zeros(B, 14 * N, device=dev)
                        t0 = time.time()
                        while time.time() - t0 < 0.05:                  # steady clocks: back-to-back solves in front of the measurement
                            for _ in range(10):
                                sol.solve(S, P, g, lam, cfg, pc)
                            torch.cuda.synchronize()
                        ts = []
                        for i in range(5):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(20):
                                sol.solve(S, P, g, lam, cfg, pc)
                            e1.record()
                            torch.cuda.synchronize()
                            ts.append(e0.elapsed_time(e1) / 20)
                        t[K] = float(np.median(ts))
                    out[f"{name} N={N} {pc} B={B}"] = (round((t[167] - t[20]) / 147 * 1e3, 3), round(t[167], 4))
    print(json.dumps(out))
    sys.exit(0)
libs = sys.argv[1:] or ["-"]
res = {}
for lib in libs:
    r = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
    try:
        res[lib] = json.loads(r.stdout.strip().split("\n")[-1])
    except Exception:
        print(lib, "FAILED", r.stdout[-300:], r.stderr[-600:])
keys = list(next(iter(res.values())).keys()) if res else []
print(f"{'us/it | ms@167 (back to back)':30s} " + " ".join(f"{os.path.basename(l)[:22]:>22s}" for l in res))
for k in keys:
    print(f"{k:30s} " + " ".join(f"{res[l][k][0]:10.3f} |{res[l][k][1]:9.4f}" for l in res))
