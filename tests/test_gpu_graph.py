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


def test_first_call_allocations_are_refused_inside_a_capture_and_leave_it_intact():
    """mpcg_form_schur and mpcg_block_solve allocate a handle-owned work buffer at their first call (hipMalloc: not stream work).  On a fresh
    handle whose stream is being captured they return MPCG_ERR_INVALID with a message (include/mpcg.h, GRAPH CAPTURE) instead of failing inside
    the capture; the capture stays usable — the PCG solve captured next to them replays — and after one eager call the same calls capture."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 32, 3
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=12)
    G, C, g, c = (dev(a) for a in synth.pack_kkt_dense(synth.make_kkt(N, B, 4242), np.float32))
    ref = PcgSolver(N, max_batch=B)
    z = lambda: torch.zeros(B, 3 * n * n * N, device="cuda")       # (blocks (0, left) and (N-1, right) are never written: zeros on both sides)
    S, P, gam = ref.form_schur(G.clone(), C, g, c, 1e-3, "ss", S=z(), Pinv=z(), gamma=torch.zeros(B, n * N, device="cuda"))
    lam_want = torch.zeros(B, n * N, device="cuda")
    ref.solve(S, P, gam, lam_want, cfg, "ss")
    torch.cuda.synchronize()

    sol = PcgSolver(N, max_batch=B)                     # fresh: no seam buffer, no block-solve scratch
    S2, P2, gam2 = torch.empty_like(S), torch.empty_like(P), torch.empty_like(gam)
    lam = torch.empty(B, n * N, device="cuda")
    lam_d = torch.empty(B, n * N, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    dG = G.clone()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        with pytest.raises(RuntimeError, match="outside the stream capture"):
            sol.form_schur(dG, C, g, c, 1e-3, "ss", S=S2, Pinv=P2, gamma=gam2)
        with pytest.raises(RuntimeError, match="outside the stream capture"):
            sol.block_solve(S, gam, lam_d)
        lam.zero_()
        sol.solve(S, P, gam, lam, cfg, "ss", iters=it, exits=ex)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(lam, lam_want)
    # one eager call each, then the same calls capture and replay
    sol.form_schur(G.clone(), C, g, c, 1e-3, "ss", S=S2, Pinv=P2, gamma=gam2)
    sol.block_solve(S, gam, lam_d)
    torch.cuda.synchronize()
    want_d = lam_d.clone()
    S2.zero_(); P2.zero_(); lam_d.zero_()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        dG.copy_(G)
        sol.form_schur(dG, C, g, c, 1e-3, "ss", S=S2, Pinv=P2, gamma=gam2)
        sol.block_solve(S2, gam2, lam_d)
    graph2.replay()
    torch.cuda.synchronize()
    assert torch.equal(S2, S) and torch.equal(P2, P) and torch.equal(lam_d, want_d)


def test_a_large_handles_first_double_solve_needs_reserve_f64_before_a_capture():
    """Handles with more than 8 MB of double iterates do not get the double cluster kernels' buffers from mpcg_create (a float caller would pay
    59 MB per handle at N = 128, max_batch 4096 for nothing — ADVICE r05): their first mpcg_pcg_solve_f64 allocates, so inside a capture it is
    refused with a message and leaves the capture intact; "reserve_f64" = 1 makes the buffers ahead of it, and then the captured first call
    replays the eager answer bit for bit."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 128, 4, 8
    big = 1024                                           # max_batch: 1024 x 128 x 14 x 8 B = 14.7 MB > 8 MB
    k = synth.make_kkt(N, B, 6600)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    eager = PcgSolver(N, max_batch=big)
    eager.set_option("pcg_lqk", 0)                       # (a capturing first call cannot run the latch's check: the three-column cluster kernel)
    want = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    eager.solve_f64(dS, dP, dg, want, cfg)
    torch.cuda.synchronize()
    assert eager.get_option("last_kernel_family") == 8

    sol = PcgSolver(N, max_batch=big)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    lam32 = torch.zeros(B, n * N, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    S32, P32, g32 = dS.float(), dP.float(), dg.float()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        with pytest.raises(RuntimeError, match="reserve_f64"):
            sol.solve_f64(dS, dP, dg, lam, cfg, iters=it, exits=ex)
        sol.solve(S32, P32, g32, lam32, cfg, "ss")       # the capture is intact: a float solve captured next to the refused call replays
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(lam32).all() and float(lam32.abs().max()) > 0

    sol2 = PcgSolver(N, max_batch=big)
    sol2.set_option("reserve_f64", 1)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        lam.zero_()
        sol2.solve_f64(dS, dP, dg, lam, cfg, iters=it, exits=ex)
    graph2.replay()
    torch.cuda.synchronize()
    assert sol2.get_option("last_kernel_family") == 8 and (it.cpu().numpy() == K).all()
    assert torch.equal(lam, want)
