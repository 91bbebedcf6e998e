#!/usr/bin/env python3
"""Which CU ran which workgroup of a throughput-sized lane-quad launch, when it entered and left (diagnostic build, tools/prof_phases.py --build):
the gap between consecutive workgroups of one CU, the spread of the finish times, and the time from the launch to the first entry."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_prof", "libmpcg_hip_prof.so")
from mpcgpu_amd import PcgSolver, pcg_config, synth
dev = torch.device("cuda")
N, B = 128, int(os.environ.get("GAPS_BATCH", "1024"))
k = synth.make_kkt(N, 8, 1)
S0, P0, g0 = synth.form_schur(k)
S, P, g = (torch.from_numpy(np.tile(a, ((B + 7) // 8, 1))[:B]).to(dev) for a in (S0, P0, g0))
rd = _lib.load().mpcg_debug_read_wg_prof
rd.restype = C.c_int
for K in (0, 167):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("assume_symmetric", 1)
    lam = torch.zeros(B, 14 * N, device=dev)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    for _ in range(30):
        sol.solve(S, P, g, lam, cfg, "ss")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sol.solve(S, P, g, lam, cfg, "ss"); e1.record(); torch.cuda.synchronize()
    buf = (C.c_longlong * (B * 4))()
    assert rd(buf, B * 4) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(B, 4)
    ent, ext, hw, xcc = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    cu = (xcc & 0xF) * 256 + ((hw >> 8) & 0xFF)
    t0 = ent.min()
    us = lambda x: x / 100.0                                     # s_memrealtime: 100 MHz
    print(f"K={K}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events; {len(np.unique(cu))} CUs; first entry -> last exit {us(ext.max() - t0):.1f} us; "
          f"entries of the first round spread over {us(np.sort(ent)[min(B, 256) - 1] - t0):.1f} us")
    gaps, durs, rounds = [], [], {}
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(ent[idx])]
        for j, i in enumerate(idx):
            durs.append(us(ext[i] - ent[i]))
            rounds.setdefault(j, []).append((us(ent[i] - t0), us(ext[i] - t0)))
            if j: gaps.append(us(ent[i] - ext[idx[j - 1]]))
    print(f"   workgroup duration: median {np.median(durs):.1f} us (min {np.min(durs):.1f}, max {np.max(durs):.1f}); gap between consecutive workgroups of a CU: "
          f"median {np.median(gaps) if gaps else 0:.2f} us (min {np.min(gaps) if gaps else 0:.2f}, max {np.max(gaps) if gaps else 0:.2f})")
    if K:
        per_x = {int(x): float(np.median((ext - ent)[(xcc & 0xF) == x])) / 100.0 for x in np.unique(xcc & 0xF)}
        print("   median workgroup duration per XCD (us): " + "  ".join(f"{x}: {v:.1f}" for x, v in sorted(per_x.items())))
        tot = {}
        for c in np.unique(cu):
            tot[c] = us(ext[cu == c].max() - t0)
        v = np.array(list(tot.values()))
        print(f"   last exit per CU: min {v.min():.1f}  median {np.median(v):.1f}  max {v.max():.1f} us  (the launch ends with the slowest CU: {100 * (v.max() / np.median(v) - 1):.1f} % above the median)")
    for j in sorted(rounds):
        a = np.array(rounds[j])
        print(f"   round {j}: {len(a)} workgroups, entries {a[:, 0].min():.1f} .. {a[:, 0].max():.1f} us, exits {a[:, 1].min():.1f} .. {a[:, 1].max():.1f} us")
