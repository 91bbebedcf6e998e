import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import bench
from mpcgpu_amd import PcgSolver, Plant, iiwa, synth, pcg_config, _lib as _L
import os
if os.environ.get("AB_LIB"):
    _L.LIB_PATH = os.environ["AB_LIB"]
dev = torch.device("cuda:0")
plant = Plant(device=0)
f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
for N in (32, 128):
    sol = PcgSolver(N, max_batch=1)
    xu, goals, xs = iiwa.random_windows(N, 1, 77 + N)
    d_xu, d_goal, d_xs = f32(xu), f32(goals.reshape(1, -1)), f32(xs)
    rc = iiwa.r_cost(N)
    G, C, g, c = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
    G0 = G.clone()
    S, P, gam = sol.form_schur(G, C, g, c, synth.RHO_INIT, "ss")
    lam = torch.zeros(1, 14 * N, device=dev)
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=20)
    t = lambda fn: bench.timed(fn, 30, warm=5) * 1e3
    print("N=%d one trajectory: generate_kkt %.1f us | form_schur %.1f us (chunk %d) | pcg 20 it %.1f us | compute_dz %.1f us | empty-ish (lam.zero_) %.1f us" % (
        N, t(lambda: sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)),
        t(lambda: (G.copy_(G0), sol.form_schur(G, C, g, c, synth.RHO_INIT, "ss"))) - t(lambda: G.copy_(G0)), sol.get_option("last_schur_chunk"),
        t(lambda: (lam.zero_(), sol.solve(S, P, gam, lam, cfg, "ss"))) - t(lambda: lam.zero_()),
        t(lambda: sol.compute_dz(G, C, g, lam)), t(lambda: lam.zero_())))
