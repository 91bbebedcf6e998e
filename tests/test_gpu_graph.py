"""GPU: every entry point is pure stream work after the first call, so the SQP linear-system chain
(form_schur_system -> pcg -> compute_dz; include/pcg/sqp.cuh:207-259) can be captured ONCE into a hipGraph and
replayed per SQP iteration — the reference re-launches three kernels and synchronises the device twice per iteration
(include/pcg/sqp.cuh:225,236).  Replays must reproduce the eager results bit for bit, for the single-workgroup
kernel (N=32) and for the cluster kernel (N=256: memset node + cluster launch)."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth

pytestmark = pytest.mark.gpu
n, m = 14, 7


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N,B", [(32, 4), (256, 2)])
def test_linsys_chain_replays_from_a_hipgraph(N, B):
    from mpcgpu_amd import PcgSolver, pcg_config
    cfg = pcg_config(pcg_exit_tol=1e-6, pcg_max_iter=synth.pcg_max_iter(N))
    sol = PcgSolver(N, max_batch=B)
    sets = [synth.pack_kkt_dense(synth.make_kkt(N, B, 9000 + s), np.float32) for s in range(3)]

    def eager(G, C, g, c):
        dG, dC, dg, dc = dev(G), dev(C), dev(g), dev(c)
        S, P, gam = sol.form_schur(dG, dC, dg, dc, 1e-3, "ss")
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(S, P, gam, lam, cfg, "ss")
        dz = sol.compute_dz(dG, dC, dg, lam)
        lam_d = sol.block_solve(S, gam)
        torch.cuda.synchronize()
        return lam.cpu().numpy(), dz.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy(), lam_d.cpu().numpy()

    want = [eager(*s) for s in sets]            # also the warm-up that sizes the library's scratch

    # static buffers the graph reads and writes
    inG, inC, ing, inc = (dev(a) for a in sets[0])
    dG = torch.empty_like(inG)
    S = torch.empty(B, 3 * n * n * N, device="cuda")
    P = torch.empty_like(S)
    gam = torch.empty(B, n * N, device="cuda")
    lam = torch.empty(B, n * N, device="cuda")
    dz = torch.empty(B, (n + m) * N - m, device="cuda")
    lam_d = torch.empty(B, n * N, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        dG.copy_(inG)                           # form_schur inverts G in place
        lam.zero_()
        sol.form_schur(dG, inC, ing, inc, 1e-3, "ss", S=S, Pinv=P, gamma=gam)
        sol.solve(S, P, gam, lam, cfg, "ss", iters=it, exits=ex)
        sol.compute_dz(dG, inC, ing, lam, dz=dz)
        sol.block_solve(S, gam, lam_d)                      # the other solver, same graph
    for rep in (0, 1, 2, 1):
        G, C, g, c = sets[rep]
        inG.copy_(dev(G)); inC.copy_(dev(C)); ing.copy_(dev(g)); inc.copy_(dev(c))
        graph.replay()
        torch.cuda.synchronize()
        wl, wz, wi, we, wd = want[rep]
        np.testing.assert_array_equal(lam_d.cpu().numpy(), wd)
        np.testing.assert_array_equal(it.cpu().numpy(), wi)
        np.testing.assert_array_equal(ex.cpu().numpy(), we)
        np.testing.assert_array_equal(lam.cpu().numpy(), wl)
        np.testing.assert_array_equal(dz.cpu().numpy(), wz)


@pytest.mark.parametrize("N", [128, 256])
def test_capture_while_the_symmetry_latch_is_still_pending(N):
    """A handle whose first lower-triangle solves are still waiting for the latch's asynchronous flag copy may be captured: the library must
    not query the latch's event during capture (not a capturable operation — it invalidated the capture, round 4), it stays on the guarded
    launches, and the replay reproduces the eager result."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B = 3
    k = synth.make_kkt(N, B, 31337 + N)
    S, Pinv, g = (dev(a) for a in synth.form_schur(k, rho=1e-2))
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=20)
    ref = PcgSolver(N, max_batch=B)
    want = torch.zeros(B, n * N, device="cuda")
    for _ in range(3):
        ref.solve(S, Pinv, g, want.zero_(), cfg, "ss")
        torch.cuda.synchronize()
    sol = PcgSolver(N, max_batch=B)                         # fresh handle: latch undecided
    lam = torch.zeros(B, n * N, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        sol.solve(S, Pinv, g, lam, cfg, "ss", iters=it, exits=ex)      # eager: starts the flag copy; NOT synchronised before the capture below
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            lam.zero_()
            sol.solve(S, Pinv, g, lam, cfg, "ss", iters=it, exits=ex)
    torch.cuda.synchronize()
    for _ in range(2):
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(lam.cpu().numpy(), want.cpu().numpy())
    assert sol.get_option("symmetry_state") in (0, 1)
