"""GPU: the lane-quad-per-knot kernel in double (pcg_lqk_f64.hip.h, family 9) — linsys_t = double (USE_DOUBLES, include/common/settings.cuh:41-49)
for 32 < N <= 64 on ONE CU, the lower block triangle of S and Pinv in registers: against the oracle's float64 iterate (cold and warm start, both
preconditioners, ragged horizons, the forced short ones), tolerance exits, the d_r / d_p outputs, determinism and batch-composition independence,
the symmetry latch (a non-symmetric Pinv must never reach this kernel), graph capture, and the streaming / clustered kernels as cross-checks."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import relinf

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def band64(orc, S, P, g, lam0, N, K, pc, ref, rng, trials=6):
    """The oracle's own spread under one-ulp input perturbations (small over-iterated systems are chaotic in float64 too, tools/_prof/small_f64.py)."""
    pert = lambda a_: a_ * (1 + 1.1e-16 * rng.standard_normal(a_.shape))
    return max(relinf(orc.pcg(pert(S), P, pert(g), pert(lam0), N, K, 0.0, pc)["lam"], ref) for _ in range(trials))


@pytest.mark.parametrize("N", [33, 40, 57, 64])
@pytest.mark.parametrize("precond", ["ss", "jacobi"])
def test_lane_quad_kernel_vs_oracle(orc, N, precond):
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 3, 30
    k = synth.make_kkt(N, B, 7100 + N)
    S, Pinv, g = synth.form_schur(k, precond=precond, dtype=np.float64, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    rng = np.random.default_rng(N)
    for lam0 in (np.zeros((B, n * N)), 0.1 * rng.standard_normal((B, n * N))):
        lam = dev(lam0.copy())
        it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), precond)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == 9 and sol.get_option("symmetry_state") == 1
        assert sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes_f64(14, N)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
        for b in range(B):
            ref = orc.pcg(np.nan_to_num(S[b]), np.nan_to_num(Pinv[b]), g[b], lam0[b], N, K, 0.0, precond)
            assert relinf(lam.cpu().numpy()[b], ref["lam"]) < 1e-9, (b, relinf(lam.cpu().numpy()[b], ref["lam"]))
        # the reference kernel's argument list (one trajectory; include/pcg/sqp.cuh:137-150): d_r / d_p = the residual and the search direction of
        # the last completed update, compared on the scale of r0 = gamma - S lambda0 (both are 1e-6 of it by then)
        one = [dev(a_[:1].copy()) for a_ in (np.nan_to_num(S), np.nan_to_num(Pinv), g, lam0)]
        d_r, d_p, scr = (torch.zeros(n * N, dtype=torch.float64, device="cuda") for _ in range(3))
        d_it = torch.zeros(1, dtype=torch.int32, device="cuda"); d_ex = torch.zeros(1, dtype=torch.uint8, device="cuda")
        if precond == "ss":                                # (the reference's call site has no preconditioner argument: the handle's default, SS)
            sol.solve_ref_f64(one[0], one[1], one[2], one[3], d_r, d_p, scr, scr, d_it, d_ex, K, 0.0)
            torch.cuda.synchronize()
            ref = orc.pcg(np.nan_to_num(S[0]), np.nan_to_num(Pinv[0]), g[0], lam0[0], N, K, 0.0, "ss")
            scale = np.abs(g[0] - synth.bd_to_dense(np.nan_to_num(S[0]), N) @ lam0[0]).max()
            assert sol.get_option("last_kernel_family") == 9 and int(d_it.item()) == K
            assert relinf(one[3].cpu().numpy()[0], ref["lam"]) < 1e-9
            assert np.abs(d_r.cpu().numpy() - ref["r"]).max() < 1e-9 * scale
            assert np.abs(d_p.cpu().numpy() - ref["p"]).max() < 1e-9 * max(scale, np.abs(ref["p"]).max())
        lam2 = dev(lam0.copy())
        sol.solve_f64(dS, dP, dg, lam2, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), precond)
        torch.cuda.synchronize()
        assert torch.equal(lam, lam2)                    # run-to-run determinism
    # tolerance exit: the oracle's count (+- the plateau), flag 0; then already converged: no update, lambda untouched
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-10, pcg_max_iter=5000), precond)
    torch.cuda.synchronize()
    itn = it.cpu().numpy()
    assert (ex.cpu().numpy() == 0).all()
    for b in range(B):
        ref = orc.pcg(np.nan_to_num(S[b]), np.nan_to_num(Pinv[b]), g[b], np.zeros(n * N), N, 5000, 1e-10, precond)
        assert abs(int(itn[b]) - ref["iters"]) <= max(2, ref["iters"] // 8), (b, int(itn[b]), ref["iters"])
    lamh = lam.cpu().numpy().copy()
    it2, ex2 = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-8, pcg_max_iter=5000), precond)
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == 0).all() and (ex2.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(lam.cpu().numpy(), lamh)


@pytest.mark.parametrize("N", [2, 3, 5, 16, 31, 32])
def test_forced_on_short_horizons(orc, N):
    """ "pcg_lqk" = 1: the four-wavefront build (32 knots) on horizons the row-per-lane kernel serves by default — the edges of the mapping: a
    horizon of two knots, one that ends inside a wavefront, a full one."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B = 4
    K = min(25, 4 * N)
    k = synth.make_kkt(N, B, 7300 + N)
    rng = np.random.default_rng(N)
    lam0 = 0.1 * rng.standard_normal((B, n * N))
    for pc in ("ss", "jacobi"):
        S, Pinv, g = synth.form_schur(k, precond=pc, dtype=np.float64)
        sol = PcgSolver(N, max_batch=B)
        sol.set_option("pcg_lqk", 1)
        lam = dev(lam0.copy())
        it, ex = sol.solve_f64(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == 9 and sol.get_option("last_kernel_waves") == 4 and (it.cpu().numpy() == K).all()
        for b in range(B):
            ref = orc.pcg(S[b], Pinv[b], g[b], lam0[b], N, K, 0.0, pc)["lam"]
            e = relinf(lam.cpu().numpy()[b], ref)
            assert e < max(1e-9, 20 * band64(orc, S[b], Pinv[b], g[b], lam0[b], N, K, pc, ref, rng) if e >= 1e-9 else 0.0), (pc, b, e)


def test_full_batch_is_composition_independent_and_agrees_with_the_other_double_kernels(orc):
    """N = 64, 700 trajectories (more than the chip holds workgroups): copies of five systems solve to the same bits wherever they land, a
    sub-batch gives the same bits, and the clustered row-per-lane kernel ("pcg_lqk" = 0) and the streaming kernel ("cluster" = 0) agree to round-off."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 64, 700, 25
    k = synth.make_kkt(N, 5, 7400)
    S5, P5, g5 = synth.form_schur(k, dtype=np.float64)
    rep = (B + 4) // 5
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S5, P5, g5))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 9 and (it.cpu().numpy() == K).all()
    lamh = lam.cpu().numpy()
    for b in range(5, B):
        np.testing.assert_array_equal(lamh[b], lamh[b % 5])
    lam_s = torch.zeros(7, n * N, dtype=torch.float64, device="cuda")
    sol.solve_f64(dS[:7], dP[:7], dg[:7], lam_s, cfg)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lam_s.cpu().numpy(), lamh[:7])
    for opt, fam in (("pcg_lqk", 8), ("cluster", 3)):
        sol.set_option(opt, 0)
        lam_o = torch.zeros(5, n * N, dtype=torch.float64, device="cuda")
        sol.solve_f64(dS[:5], dP[:5], dg[:5], lam_o, cfg)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == fam
        assert relinf(lam_o.cpu().numpy(), lamh[:5]) < 1e-9


def test_a_non_symmetric_pinv_never_reaches_the_lower_triangle_kernel(orc):
    """The kernel reads only the left + diagonal block columns (include/mpcg.h, BLOCK SYMMETRY).  A caller-made Pinv whose right blocks are not
    the transposes of the next row's left blocks: the handle's one blocking check on the first call finds it, the call — and every later one —
    runs a kernel that reads all three columns, and the result is the oracle's for THOSE matrices."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 48, 2, 20
    k = synth.make_kkt(N, B, 7500)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    Pa = Pinv.copy().reshape(B, N, 3, n * n)
    Pa[:, :-1, 2, :] *= 0.5                                   # right blocks halved: a valid (if poor) preconditioner apply, not symmetric
    Pa = Pa.reshape(B, -1)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dev(S), dev(Pa), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 2 and sol.get_option("last_kernel_family") == 8
    assert "block-symmetric" in sol.last_error()
    for b in range(B):
        ref = orc.pcg(S[b], Pa[b], g[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        assert relinf(lam.cpu().numpy()[b], ref) < 1e-9
    # "assume_symmetric" = 1 on a fresh handle: the caller vouches, no check — with the symmetric matrices the lane-quad kernel at once
    sol2 = PcgSolver(N, max_batch=B)
    sol2.set_option("assume_symmetric", 1)
    lam2 = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    sol2.solve_f64(dev(S), dev(Pinv), dev(g), lam2, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol2.get_option("last_kernel_family") == 9
    for b in range(B):
        assert relinf(lam2.cpu().numpy()[b], orc.pcg(S[b], Pinv[b], g[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]) < 1e-9


def test_graph_capture_before_and_after_the_latch_knows():
    """A fresh handle whose first double solve is captured cannot run its blocking symmetry check: the captured call is the clustered
    row-per-lane kernel (all three columns).  After one eager call the latch knows and a capture holds the lane-quad kernel; both graphs
    replay to the eager results."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 64, 5, 20
    k = synth.make_kkt(N, B, 7600)
    S, Pinv, g = (dev(a) for a in synth.form_schur(k, dtype=np.float64))
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")

    def capture():
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            lam.zero_()
            sol.solve_f64(S, Pinv, g, lam, cfg, iters=it, exits=ex)
        return graph

    g1 = capture()
    assert sol.get_option("last_kernel_family") == 8 and sol.get_option("symmetry_state") == 0
    g1.replay(); torch.cuda.synchronize()
    first = lam.clone()
    lam_e = torch.zeros_like(lam)
    sol.solve_f64(S, Pinv, g, lam_e, cfg)                      # eager: the one blocking check, then the lane-quad kernel
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 9 and sol.get_option("symmetry_state") == 1
    g2 = capture()
    assert sol.get_option("last_kernel_family") == 9
    for _ in range(2):
        g2.replay(); torch.cuda.synchronize()
        assert torch.equal(lam, lam_e) and (it.cpu().numpy() == K).all()
    assert relinf(first.cpu().numpy(), lam_e.cpu().numpy()) < 1e-9


# ---- the same kernel across G = ceil(N / 64) CUs of one XCD (pcg_lqk_cluster_f64.hip.h, family 10) ----

@pytest.mark.parametrize("N", [65, 100, 128, 200, 256, 300, 512])
@pytest.mark.parametrize("precond", ["ss", "jacobi"])
@pytest.mark.parametrize("l2", [1, 0])
def test_clustered_lane_quad_kernel_vs_oracle(orc, N, precond, l2):
    """64 < N <= 512, ragged member sizes (65 -> 32 + 33 knots, 300 -> five members of 60), L2-resident and write-through hand-offs: fixed
    iteration counts against the oracle's float64 iterate (cold and warm start), nothing left to the fix-up, the tolerance exit, determinism."""
    from mpcgpu_amd import PcgSolver, pcg_config
    if l2 == 0 and N not in (100, 256):
        pytest.skip("write-through hand-offs: two horizons")
    B, K = 3, 25
    k = synth.make_kkt(N, B, 7700 + N)
    S, Pinv, g = synth.form_schur(k, precond=precond, dtype=np.float64, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_l2", l2)
    rng = np.random.default_rng(N)
    G = (N + 63) // 64
    for lam0 in (np.zeros((B, n * N)), 0.1 * rng.standard_normal((B, n * N))):
        lam = dev(lam0.copy())
        it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), precond)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == 10 and sol.get_option("last_kernel_cluster") == G and sol.get_option("cluster_fixups") == 0
        assert sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes_f64(14, N)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
        for b in range(B):
            ref = orc.pcg(np.nan_to_num(S[b]), np.nan_to_num(Pinv[b]), g[b], lam0[b], N, K, 0.0, precond)["lam"]
            assert relinf(lam.cpu().numpy()[b], ref) < 1e-9, (b, relinf(lam.cpu().numpy()[b], ref))
        lam2 = dev(lam0.copy())
        sol.solve_f64(dS, dP, dg, lam2, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), precond)
        torch.cuda.synchronize()
        assert torch.equal(lam, lam2)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-10, pcg_max_iter=5000), precond)
    torch.cuda.synchronize()
    assert (ex.cpu().numpy() == 0).all()
    ref = orc.pcg(np.nan_to_num(S[0]), np.nan_to_num(Pinv[0]), g[0], np.zeros(n * N), N, 5000, 1e-10, precond)
    assert abs(int(it.cpu().numpy()[0]) - ref["iters"]) <= max(2, ref["iters"] // 8)
    lamh = lam.cpu().numpy().copy()
    it2, ex2 = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-8, pcg_max_iter=5000), precond)
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == 0).all() and (ex2.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(lam.cpu().numpy(), lamh)


def test_clustered_lane_quad_kernel_queue_outputs_and_cross_checks(orc):
    """N = 128, 300 trajectories (128 clusters of two CUs fit the chip: the queue is in use): copies of five systems solve to the same bits wherever
    they land; the reference kernel's argument list gives d_r / d_p; the clustered row-per-lane kernel ("pcg_lqk" = 0) and the streaming kernel
    ("cluster" = 0) agree to round-off."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 128, 300, 20
    k = synth.make_kkt(N, 5, 7800)
    S5, P5, g5 = synth.form_schur(k, dtype=np.float64)
    rep = (B + 4) // 5
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S5, P5, g5))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 10 and (it.cpu().numpy() == K).all() and sol.get_option("cluster_fixups") == 0
    lamh = lam.cpu().numpy()
    for b in range(5, B):
        np.testing.assert_array_equal(lamh[b], lamh[b % 5])
    lam0 = 0.1 * np.random.default_rng(1).standard_normal(n * N)
    d_lam = dev(lam0.copy())
    d_r, d_p, scr = (torch.zeros(n * N, dtype=torch.float64, device="cuda") for _ in range(3))
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda"); d_ex = torch.zeros(1, dtype=torch.uint8, device="cuda")
    sol.solve_ref_f64(dS[:1], dP[:1], dg[:1], d_lam, d_r, d_p, scr, scr, d_it, d_ex, K, 0.0)
    torch.cuda.synchronize()
    ref = orc.pcg(S[0], Pinv[0], g[0], lam0, N, K, 0.0, "ss")
    scale = np.abs(g[0] - synth.bd_to_dense(S[0], N) @ lam0).max()
    assert sol.get_option("last_kernel_family") == 10 and int(d_it.item()) == K
    assert relinf(d_lam.cpu().numpy(), ref["lam"]) < 1e-9
    assert np.abs(d_r.cpu().numpy() - ref["r"]).max() < 1e-9 * scale
    assert np.abs(d_p.cpu().numpy() - ref["p"]).max() < 1e-9 * max(scale, np.abs(ref["p"]).max())
    for opt, fam in (("pcg_lqk", 8), ("cluster", 3)):
        sol.set_option(opt, 0)
        lam_o = torch.zeros(5, n * N, dtype=torch.float64, device="cuda")
        sol.solve_f64(dS[:5], dP[:5], dg[:5], lam_o, cfg)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == fam
        assert relinf(lam_o.cpu().numpy(), lamh[:5]) < 1e-9


@pytest.mark.parametrize("N,G", [(33, 2), (40, 8), (64, 3), (128, 4), (128, 8), (200, 7), (512, 8)])
def test_forced_member_counts(orc, N, G):
    """ "cluster" = G: more members than the horizon needs — members of 5 knots (N = 40 over eight CUs: most wavefronts of a member hold no knot),
    ragged splits (200 over seven), a single-CU horizon on three CUs — against the oracle and, bit for bit, against nothing else: every member count
    sums the inner products in its own order."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 3, 20
    k = synth.make_kkt(N, B, 7900 + N + G)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    lam0 = 0.1 * np.random.default_rng(N + G).standard_normal((B, n * N))
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", G)
    lam = dev(lam0.copy())
    it, ex = sol.solve_f64(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 10 and sol.get_option("last_kernel_cluster") == G and sol.get_option("cluster_fixups") == 0
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    for b in range(B):
        assert relinf(lam.cpu().numpy()[b], orc.pcg(S[b], Pinv[b], g[b], lam0[b], N, K, 0.0, "ss")["lam"]) < 1e-9


def test_short_horizons_take_the_lane_quad_kernel_only_for_throughput_sized_batches(orc):
    """16 < N <= 32: a call with at least four trajectories per CU runs two 32-knot lane-quad workgroups per CU (family 9), anything smaller the
    row-per-lane kernel (family 5) — a sub-batch of the same systems therefore comes from another kernel and agrees to round-off, not to the bit."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, K = 24, 20
    sol = PcgSolver(N, max_batch=4096)
    B = 4 * sol.get_option("num_cus")
    k = synth.make_kkt(N, 4, 8100)
    S4, P4, g4 = synth.form_schur(k, dtype=np.float64)
    S, Pinv, g = (np.tile(a_, (B // 4, 1)) for a_ in (S4, P4, g4))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 9 and sol.get_option("last_kernel_waves") == 4 and (it.cpu().numpy() == K).all()
    lamh = lam.cpu().numpy()
    for b in range(4, B):
        np.testing.assert_array_equal(lamh[b], lamh[b % 4])
    for b in range(4):
        assert relinf(lamh[b], orc.pcg(S4[b], P4[b], g4[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]) < 1e-9
    lam_s = torch.zeros(B - 1, n * N, dtype=torch.float64, device="cuda")
    sol.solve_f64(dS[:B - 1], dP[:B - 1], dg[:B - 1], lam_s, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 5
    assert relinf(lam_s.cpu().numpy()[:4], lamh[:4]) < 1e-9
    sol16 = PcgSolver(16, max_batch=B)                       # N <= 16: the row-per-lane kernel at every batch (twice the lane-quad kernel's rate)
    k16 = synth.make_kkt(16, 2, 8101)
    S2, P2, g2 = (np.tile(a_, (B // 2, 1)) for a_ in synth.form_schur(k16, dtype=np.float64))
    sol16.solve_f64(dev(S2), dev(P2), dev(g2), torch.zeros(B, n * 16, dtype=torch.float64, device="cuda"), cfg)
    torch.cuda.synchronize()
    assert sol16.get_option("last_kernel_family") == 5
