"""s_memtime stamps of one iteration of the row-per-lane kernel (workgroup 0, every wave): -DMPCG_PROF build.
   python tools/prof_phases.py --build ; python tools/_prof/rpl_phases.py [N] [batch] [ss|jacobi]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpcgpu_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_prof", "libmpcg_hip_prof.so")
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pc = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, pc, torch.device("cuda", 0), chunk=min(B, 64))
lam = torch.zeros(B, 14 * N, device="cuda")
cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(N))
for rep in range(3):
    lam.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); it, ex = sol.solve(dS, dP, dg, lam, cfg, pc); e1.record(); torch.cuda.synchronize()
rd = _lib.load().mpcg_debug_read_prof
buf = (C.c_longlong * (16 * 32))()
assert rd(buf, 16 * 32) == 0
nw = sol.get_option("last_kernel_waves")
t = np.array(buf[:], dtype=np.int64).reshape(16, 32)[:nw, :9]
names = ["S pass", "fold", "publish", "barrier", "alpha,r", "P pass", "fold+publish", "barrier", "beta,p"]
print(f"N={N} batch={B} {pc} family {sol.get_option('last_kernel_family')} waves {nw} x {sol.get_option('last_kernel_reg_rows')}  {e0.elapsed_time(e1):.3f} ms (instrumented); ticks of iteration 20")
print("wave " + " ".join(f"{n:>13s}" for n in names[:8]) + "   total(0..8)")
for w in range(nw):
    d = np.diff(t[w])
    print(f"{w:4d} " + " ".join(f"{int(x):13d}" for x in d) + f"   {int(t[w][8] - t[w][0]):6d}")
