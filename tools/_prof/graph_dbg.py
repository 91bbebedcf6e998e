import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mpcgpu_amd import PcgSolver, pcg_config, synth
N, B = 256, 2
n = 14
cfg = pcg_config(pcg_exit_tol=1e-6, pcg_max_iter=118)
k = synth.make_kkt(N, B, 9000)
S0, P0, g0 = synth.form_schur(k)
S, P, gam = (torch.from_numpy(a).cuda() for a in (S0, P0, g0))
for fix in (0, 1):
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_fixup", fix)
    lam_e = torch.zeros(B, n * N, device="cuda")
    sol.solve(S, P, gam, lam_e, cfg, "ss")
    torch.cuda.synchronize()
    lam = torch.empty(B, n * N, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lam.zero_()
        sol.solve(S, P, gam, lam, cfg, "ss", iters=it, exits=ex)
    for rep in range(3):
        graph.replay(); torch.cuda.synchronize()
        import ctypes as C
        buf = (C.c_ulonglong * (4096 + 8 * 64))()
        sol.lib.mpcg_debug_read_cluster_scratch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        sol.lib.mpcg_debug_read_cluster_scratch(sol._h, buf, 4096 + 8 * 64)
        arr = np.array(buf[:], dtype=np.uint64)
        print("   flags:", arr[0], arr[16], " partial cells member0:", [hex(int(x)) for x in arr[4096 + 56:4096 + 58]], " epochs of all members' word56:", [int(arr[4096 + 64 * m + 56] >> 32) for m in range(8)])
        print("fixup", fix, "replay", rep, "equal eager:", torch.equal(lam, lam_e), float((lam - lam_e).abs().max()), it.cpu().tolist(), ex.cpu().tolist(), flush=True)
