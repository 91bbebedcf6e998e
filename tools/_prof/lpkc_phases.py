"""s_memtime stamps of one iteration of the clustered lane-pair kernel (member 0 of cluster 0, every wave): -DMPCG_PROF build.
   python tools/prof_phases.py --build ; python tools/_prof/lpkc_phases.py [N] [batch]
Stamps (S half: +0, Pinv half: +8): 0 half starts (operand rebuild) | 1 pass done, wave partial published | 2 enters the exchange |
3 (poller) all granules seen | 4 (poller) halo + sum in LDS | 5 past the barrier."""
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
t = np.array(buf[:], dtype=np.int64).reshape(16, 32)[:8, :14]
print(f"N={N} batch={B} family {sol.get_option('last_kernel_family')} G={sol.get_option('last_kernel_cluster')}  {e0.elapsed_time(e1):.3f} ms (instrumented), "
      f"{e0.elapsed_time(e1) * 1e3 / synth.pcg_max_iter(N):.3f} us per iteration; raw stamps of iteration 20, member 0, relative to the earliest (0 = not stamped)")
t0 = t[t > 0].min()
print("wave " + " ".join(f"{i:7d}" for i in range(14)))
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{int(x - t0) if x > 0 else 0:7d}" for x in t[w]))
