#!/usr/bin/env python3
"""One trajectory through generate_kkt -> form_schur -> PCG -> compute_dz as a replayed hipGraph (bench leg batch1_sqp_step_latency), N from argv:
   rocprofv3 --kernel-trace --stats -- python tools/_prof/batch1_trace.py 32      -> kernel durations vs the measured step: what the graph's edges cost."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpcgpu_amd import PcgSolver, Plant, iiwa, pcg_config, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
plant = Plant(device=0)
f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
sol = PcgSolver(N, max_batch=1, device=0)
if os.environ.get("KKT_F32"):
    sol.set_option("kkt_f32", int(os.environ["KKT_F32"]))
xu_h, goals_h, xs_h = iiwa.random_windows(N, 1, 77 + N)
rc = iiwa.r_cost(N)
cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=synth.pcg_max_iter(N))
d_xu, d_goal, d_xs = f32(xu_h[:1]), f32(goals_h[:1].reshape(1, -1)), f32(xs_h[:1])
lam_prev = torch.zeros(1, 14 * N, device=dev)
lam = lam_prev.clone()
it = torch.zeros(1, dtype=torch.int32, device=dev)
ex = torch.zeros(1, dtype=torch.uint8, device=dev)
dz = torch.empty(1, 21 * N - 7, device=dev)
def step():
    lam.copy_(lam_prev)
    G_, C_, g_, c_ = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
    S_, P_, gam_ = sol.form_schur(G_, C_, g_, c_, synth.RHO_INIT, "ss")
    sol.solve(S_, P_, gam_, lam, cfg, "ss", iters=it, exits=ex)
    sol.compute_dz(G_, C_, g_, lam, dz=dz)
for _ in range(3):
    step(); torch.cuda.synchronize()
# warm start = the solution itself shifted a little: a handful of iterations, as in the MPC loop
step(); torch.cuda.synchronize()
lam_prev.copy_(lam)              # (warm start = the converged multipliers: a few iterations, as in the MPC loop)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
for _ in range(20): gr.replay()
torch.cuda.synchronize()
R = 300
t0 = time.perf_counter()
for _ in range(R): gr.replay()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / R * 1e6
print(f"N={N}: {us:.1f} us per replayed step, {int(it.item())} PCG iterations, family {sol.get_option('last_kernel_family')}", flush=True)
