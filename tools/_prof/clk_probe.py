"""Shader clock and socket power while one PCG kernel family runs back to back on all CUs (are the lane-per-block / lane-pair kernels
clock-limited at full batch?).  python tools/_prof/clk_probe.py [lpk|lpb] [batch]"""
import os, subprocess, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, pcg_config, synth

which = sys.argv[1] if len(sys.argv) > 1 else "lpk"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = 128
k = synth.make_kkt(N, 32, 1)
S0, P0, g0 = synth.form_schur(k)
rep = (B + 31) // 32
S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B]).cuda() for a in (S0, P0, g0))
sol = PcgSolver(N, max_batch=B)
sol.set_option("pcg_" + which, 1)
cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=167)
lam = torch.zeros(B, 14 * N, device="cuda")
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            s = [ln.strip() for ln in out.splitlines() if "sclk" in ln or "Power" in ln or "fclk" in ln]
            samples.append(" | ".join(s))
        except Exception as e:  # noqa
            samples.append(repr(e))
        time.sleep(0.3)


th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < 4.0:
    for _ in range(50):
        sol.solve(S, P, g, lam, cfg, "ss")
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print(which, "batch", B, "ms per solve (back to back, incl. launch):", e0.elapsed_time(e1) / n)
for s in samples[:3] + samples[-4:]:
    print(s)
