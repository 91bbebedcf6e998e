"""s_memtime stamps of one iteration of the clustered lane-per-block kernel (member 0 of cluster 0, every wave): -DMPCG_PROF build.
   python tools/prof_phases.py --build ; python tools/_prof/lpbc_phases.py [N] [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpcgpu_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_prof", "libmpcg_hip_prof.so")
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", torch.device("cuda", 0), chunk=min(B, 64))
lam = torch.zeros(B, 14 * N, device="cuda")
cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
for rep in range(3):
    lam.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss"); e1.record(); torch.cuda.synchronize()
rd = _lib.load().mpcg_debug_read_prof
buf = (C.c_longlong * (16 * 32))()
assert rd(buf, 16 * 32) == 0
t = np.array(buf[:], dtype=np.int64).reshape(16, 32)[:8, :13]
names = ["S pass+publish", "tables", "poll", "sum+barrier", "r update+barrier", "P pass / lam", "tables", "poll", "sum+barrier", "p update+barrier"]
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 6), (6, 7), (7, 8), (8, 9), (9, 10), (10, 12)]
print(f"N={N} batch={B} family {sol.get_option('last_kernel_family')} G={sol.get_option('last_kernel_cluster')}  {e0.elapsed_time(e1):.3f} ms (instrumented); ticks of iteration 20, member 0")
print("wave " + " ".join(f"{n:>16s}" for n in names) + "   total")
for w in range(8):
    row = t[w]
    d = [int(row[b] - row[a]) if row[a] and row[b] else 0 for a, b in pairs]
    print(f"{w:4d} " + " ".join(f"{x:16d}" for x in d) + f"   {int(row[12] - row[0]):6d}")
