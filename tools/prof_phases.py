#!/usr/bin/env python3
"""Where does one PCG iteration of pcg_traj_kernel go?  Diagnostic build (-DMPCG_PROF) of the library in
tools/_prof/, workgroup 0 stamps s_memtime at the phase boundaries of iteration 20.
   python tools/prof_phases.py --build            (here: cross-compile)
   python tools/prof_phases.py --knots 128 --batch 1024 --cfg 4:7:-1 [--cfg 8:3:-1]   (on the GPU box)"""
import argparse, ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_SO = os.path.join(ROOT, "tools", "_prof", "libmpcg_hip_prof.so")
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--knots", type=int, default=128)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--cfg", action="append", default=[])
args = ap.parse_args()
if args.build:
    from mpcgpu_amd import build as B
    os.makedirs(os.path.dirname(PROF_SO), exist_ok=True)
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-DMPCG_PROF", *B.SOURCES, "-o", PROF_SO])
    print("built", PROF_SO)
    sys.exit(0)
import numpy as np, torch
from mpcgpu_amd import _lib
_lib.LIB_PATH = PROF_SO
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
N, Bn = args.knots, args.batch
sol = PcgSolver(N, max_batch=Bn)
dS, dP, dg = bench.build_inputs(sol, N, Bn, 0, "ss", torch.device("cuda", 0))
lam = torch.zeros(Bn, 14 * N, device="cuda")
cfg = pcg_config(pcg_exit_tol=1e-30, pcg_max_iter=synth.pcg_max_iter(N))
rd = _lib.load().mpcg_debug_read_prof
rd.restype = C.c_int
NAMES = ["S regs", "S lds", "S stream", "S fold", "S barrier", "vec update", "barrier", "(P start)",
         "P regs", "P lds", "P stream", "P fold", "P barrier", "p update", "barrier"]
IDX = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
LPK_NAMES = ["S transp", "S direct", "S merge+dot", "(to barrier)", "barrier A", "alpha+r/lam", "barrier B", "(P start)",
             "P transp", "P direct", "P merge+dot", "(to barrier)", "barrier C", "eta+p upd", "barrier D"]
for spec in args.cfg or ["4:7:-1"]:
    f = [int(x) for x in spec.split(":")] if spec not in ("lpk", "lqb") else [4 if N <= 64 else 8]
    if spec == "lqb":     # float lane-quad kernel: every wave runs both passes.  0 iteration start | 1 operand in place | 2 FMAs done | 3 published | 4 past barrier A | 5 alpha known | 6 operand | 7 FMAs done | 8 published | 9 past barrier C
        f = [8 if N > 64 else 4 if N > 32 else 2]
        NAMES = ["rebuild p", "S FMAs", "S epilogue", "barrier A", "alpha", "rebuild r", "P FMAs", "P epilogue", "barrier C", "beta"] + ["-"] * 5
        sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", 1); sol.set_option("assume_symmetric", 1)
    elif spec == "lpk":      # lane-pair-per-knot kernel (round 3 default for 36 < N <= 128): waves 0-3 S, 4-7 Pinv (stamps of the other role's pass stay 0)
        NAMES = LPK_NAMES
        sol.set_option("pcg_lpk", 1)
    else:
        sol.set_option("cluster", 0)
        sol.set_option("pcg_waves", f[0]); sol.set_option("pcg_reg_rows", f[1]); sol.set_option("pcg_lds_rows", f[2] if len(f) > 2 else -1)
    for rep in range(2):
        lam.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss"); e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = (C.c_longlong * (16 * 32))()
    assert rd(buf, 16 * 32) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(16, 32)[: f[0], :16]
    print(f"== N={N} batch={Bn} cfg={spec}  {ms:.3f} ms  {it.sum().item() / ms / 1e3:.2f} Miter/s (instrumented)   "
          f"[s_memtime ticks; one iteration of workgroup 0, per wave]")
    if spec == "lqb":
        pt = np.array(buf[:], dtype=np.int64).reshape(16, 32)[: f[0], 16:23]
        print("prologue / write-back of workgroup 600 (third round), s_memtime ticks from the wave's entry: matrices loaded | parked + barrier | vectors staged | "
              "set-up passes done (loop entry) | loop exit | written back")
        for w in range(f[0]):
            print(f"{w:4d} " + " ".join(f"{int(x - pt[w, 0]):9d}" for x in pt[w, 1:]) + f"   (entry {int(pt[w, 0] - pt[:, 0].min())} after the first wave)")
    d = np.diff(t, axis=1)
    tot = t[:, 15] - t[:, 0]
    print("raw stamps relative to the earliest stamp of the iteration (0 = not stamped by this wave):")
    t0 = t[t > 0].min()
    for w in range(f[0]):
        print(f"{w:4d} " + " ".join(f"{int(x - t0) if x > 0 else 0:7d}" for x in t[w]))
    print("wave " + " ".join(f"{n:>10s}" for n in NAMES) + "      total")
    for w in range(f[0]):
        print(f"{w:4d} " + " ".join(f"{int(x):10d}" for x in d[w]) + f" {int(tot[w]):10d}")
    print(" avg " + " ".join(f"{x:10.0f}" for x in d.mean(axis=0)) + f" {tot.mean():10.0f}")
