import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
from mpcgpu_amd import PcgSolver, Plant, iiwa
N, B = 128, 8
dev = torch.device("cuda", 0)
xu, ee, xs = iiwa.random_windows(N, B, seed=3)
plant = Plant()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
outs = {}
for f32 in (0, 1, 2):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("kkt_f32", f32)
    outs[f32] = [o.cpu().numpy().astype(np.float64) for o in sol.generate_kkt(plant, t(ee).reshape(B, -1), t(xs), t(xu), iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))]
for nm, a0, a1, a2 in zip("GCgc", *[outs[k] for k in (0, 1, 2)]):
    sc = max(1.0, np.abs(a0).max())
    print(nm, "float1 vs f64:", np.abs(a1 - a0).max() / sc, " packed vs f64:", np.abs(a2 - a0).max() / sc, " packed vs float1:", np.abs(a2 - a1).max() / sc, "equal entries:", (a1 == a2).mean())
