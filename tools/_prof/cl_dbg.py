import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mpcgpu_amd import PcgSolver, pcg_config, synth
N, B = 256, 2
cfg = pcg_config(pcg_exit_tol=1e-6, pcg_max_iter=118)
k = synth.make_kkt(N, B, 9000)
S, P, g = synth.form_schur(k)
dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
for fix in (0, 1):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_fixup", fix)
    outs = []
    for rep in range(6):
        lam = torch.zeros(B, 14 * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
        torch.cuda.synchronize()
        outs.append(lam.cpu().numpy())
        print("fixup", fix, "rep", rep, "it", it.cpu().tolist(), "ex", ex.cpu().tolist(), "fam", sol.get_option("last_kernel_family"),
              "G", sol.get_option("last_kernel_cluster"), "same as rep0:", np.array_equal(outs[0], outs[-1]), flush=True)
    if fix == 0: base = outs[0]
    else: print("fixup result == no-fixup result:", np.array_equal(base, outs[0]), np.abs(base - outs[0]).max())
