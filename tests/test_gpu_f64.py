"""GPU: the double-precision PCG entry (linsys_t = double, USE_DOUBLES=1 of include/common/settings.cuh:41-49) against
the float64 oracle: same iteration counts, iterates equal up to the summation order of the inner products."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import relinf

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", [2, 9, 32, 128, 300])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_f64_fixed_iterations_vs_oracle(orc, N, pc):
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 3, 40
    k = synth.make_kkt(N, B, 8800 + N)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64, poison_unused=True)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    lam = lam.cpu().numpy()
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(Pinv[b])
        ref = orc.pcg(Sz, Pz, g[b], np.zeros(n * N), N, K, 0.0, pc)
        # what one ulp on the right-hand side does to the same float64 iteration (CG loses orthogonality on the small,
        # clustered systems long before K: N=9 drifts 7e-4 by iteration 40, N=128 stays at 3e-14)
        band = max(relinf(orc.pcg(Sz, Pz, np.nextafter(g[b], s), np.zeros(n * N), N, K, 0.0, pc)["lam"], ref["lam"])
                   for s in (np.inf, -np.inf))
        assert relinf(lam[b], ref["lam"]) <= max(1e-10, 20 * band), (relinf(lam[b], ref["lam"]), band)


def test_f64_tolerance_exit_warm_start_and_flags(orc):
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 64, 4
    k = synth.make_kkt(N, B, 31)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-10, pcg_max_iter=5000))
    torch.cuda.synchronize()
    itn = it.cpu().numpy()
    assert (ex.cpu().numpy() == 0).all() and (itn > 0).all()
    lamh = lam.cpu().numpy()
    for b in range(B):
        ref = orc.pcg(S[b], Pinv[b], g[b], np.zeros(n * N), N, 5000, 1e-10, "ss")
        assert abs(int(itn[b]) - ref["iters"]) <= max(2, ref["iters"] // 8)      # (the tail of the convergence is a plateau)
        assert relinf(lamh[b], orc.direct_solve(S[b], g[b], N)) < 1e-3      # (|eta| < 1e-10 is not a residual bound at cond ~1e5)
    it2, ex2 = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-8, pcg_max_iter=5000))   # already converged
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == 0).all() and (ex2.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(lam.cpu().numpy(), lamh)
    it3, ex3 = sol.solve_f64(dS, dP, dg, torch.zeros_like(lam), pcg_config(pcg_exit_tol=1e-30, pcg_max_iter=3))
    torch.cuda.synchronize()
    assert (it3.cpu().numpy() == 3).all() and (ex3.cpu().numpy() == 1).all()


@pytest.mark.parametrize("N", [2, 3, 8, 33, 128])
@pytest.mark.parametrize("precond", ["ss", "jacobi"])
@pytest.mark.parametrize("formation", ["auto", "walk16", "walk5", "walk1", "lds"])
def test_f64_form_schur_and_dz_bit_exact_vs_oracle(orc, N, precond, formation):
    """linsys_t = double types the whole linear-system step of the reference (include/common/settings.cuh:41-49), not only the PCG:
    mpcg_form_schur_f64 / mpcg_compute_dz_f64 against the oracle's double instantiation, bit for bit (same operation order, contraction
    off on both sides), including which bd slots are left unwritten; then the double-precision chain KKT blocks -> Schur -> PCG -> dz
    solves the KKT system to 1e-9.  Round 5: the register-resident walking formation in double (schur_walk_f64.hip.h) at every chunk length
    — auto (L = 1 here), 16, 5 (ragged last chunk), 1 — and the four-knots-per-wavefront dz kernel; "lds" = the round-1 LDS kernels
    ("schur_dpp" = 0, "dz_dpp" = 0)."""
    from mpcgpu_amd import PcgSolver, pcg_config
    m, B = 7, 5                                       # (a wavefront of four chunks straddles trajectories)
    k = synth.make_kkt(N, B, 555 + N)
    G, C, g, c = synth.pack_kkt_dense(k, np.float64)
    sol = PcgSolver(N, max_batch=B)
    if formation.startswith("walk"):
        sol.set_option("schur_chunk", int(formation[4:]))
    elif formation == "lds":
        sol.set_option("schur_dpp", 0)
        sol.set_option("dz_dpp", 0)
    dG, dC, dg, dc = dev(G), dev(C), dev(g), dev(c)
    S = torch.full((B, 3 * n * n * N), float("nan"), device="cuda", dtype=torch.float64)
    P = torch.full((B, 3 * n * n * N), float("nan"), device="cuda", dtype=torch.float64)
    gam = torch.full((B, n * N), float("nan"), device="cuda", dtype=torch.float64)
    sol.form_schur(dG, dC, dg, dc, 1e-3, precond, S=S, Pinv=P, gamma=gam)
    torch.cuda.synchronize()
    assert sol.get_option("last_schur_chunk") == {"auto": 1, "lds": 0}.get(formation, int(formation[4:]) if formation.startswith("walk") else -1)
    Sh, Ph, gh, Gih = S.cpu().numpy(), P.cpu().numpy(), gam.cpu().numpy(), dG.cpu().numpy()
    for b in range(B):
        So, Po, go, Go = orc.form_schur(G[b], C[b], g[b], c[b], N, np.float64(1e-3), ss=(precond == "ss"))
        np.testing.assert_array_equal(Sh[b], So)
        np.testing.assert_array_equal(Ph[b], Po)
        np.testing.assert_array_equal(gh[b], go)
        np.testing.assert_array_equal(Gih[b], Go)
    # solve in double precision and recover dz; compare dz with the oracle's on the same multipliers, bit for bit
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    Sz, Pz = torch.nan_to_num(S), torch.nan_to_num(P)
    it, ex = sol.solve_f64(Sz, Pz, gam, lam, pcg_config(pcg_exit_tol=1e-22, pcg_max_iter=4000), precond)
    dz = sol.compute_dz(dG, dC, dg, lam)
    torch.cuda.synchronize()
    lamh, dzh = lam.cpu().numpy(), dz.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(dzh[b], orc.compute_dz(Gih[b], C[b], g[b], lamh[b], N))
        # the multipliers solve S lambda = gamma: true residual in float64
        r = orc.bt_spmv(np.nan_to_num(Sh[b]), lamh[b], N) - gh[b]
        assert np.abs(r).max() <= 1e-9 * max(1.0, np.abs(gh[b]).max()), (b, np.abs(r).max())


def test_f64_formations_agree_at_throughput_sized_batches():
    """The automatic chunk length of a throughput-sized double call (L = 4 at 700 x 64 knots: the grid-stride loop wraps, the last wavefront is
    ragged, the seam buffer is the one sized at the handle's first — float — call) against one row per chunk and the LDS kernels: the same
    bits in S, Pinv, gamma, G^-1; and both dz kernels on the result."""
    from mpcgpu_amd import PcgSolver
    N, B = 64, 700
    k = synth.make_kkt(N, 50, 4242)
    G, C, g, c = (np.tile(a, (B // 50, 1)) for a in synth.pack_kkt_dense(k, np.float64))
    sol = PcgSolver(N, max_batch=B)
    f32 = [dev(a[:8].astype(np.float32)) for a in (G, C, g, c)]
    sol.form_schur(*f32, 1e-3, "ss")                                  # (a float call first: the seam buffer must serve the double calls too)
    outs, chunks = [], []
    lam = torch.randn(B, n * N, dtype=torch.float64, device="cuda")
    for mode in ("auto", "one", "lds"):
        sol.set_option("schur_dpp", 0 if mode == "lds" else 1)
        sol.set_option("dz_dpp", 0 if mode == "lds" else 1)
        sol.set_option("schur_chunk", 1 if mode == "one" else 0)
        dG = dev(G)
        S = torch.full((B, 3 * n * n * N), float("nan"), device="cuda", dtype=torch.float64)
        P = torch.full((B, 3 * n * n * N), float("nan"), device="cuda", dtype=torch.float64)
        gam = torch.full((B, n * N), float("nan"), device="cuda", dtype=torch.float64)
        sol.form_schur(dG, dev(C), dev(g), dev(c), 1e-3, "ss", S=S, Pinv=P, gamma=gam)
        dz = sol.compute_dz(dG, dev(C), dev(g), lam)
        torch.cuda.synchronize()
        chunks.append(sol.get_option("last_schur_chunk"))
        outs.append([t.cpu().numpy() for t in (S, P, gam, dG, dz)])
    assert chunks[0] > 1 and chunks[1] == 1 and chunks[2] == 0, chunks
    for other in outs[1:]:
        for a0, a1 in zip(outs[0], other):
            np.testing.assert_array_equal(a0, a1)


def test_f64_wrappers_reject_mixed_precision():
    from mpcgpu_amd import PcgSolver
    N, B = 4, 1
    sol = PcgSolver(N, max_batch=B)
    k = synth.make_kkt(N, B, 1)
    G, C, g, c = synth.pack_kkt_dense(k, np.float64)
    with pytest.raises(ValueError):
        sol.form_schur(dev(G.astype(np.float32)), dev(C), dev(g), dev(c), 1e-3)


@pytest.mark.parametrize("N", [64, 128])
def test_f64_streaming_kernel_reads_the_lower_triangle_once_the_latch_says_symmetric(orc, N):
    """linsys_t = double beyond N = 32: block (k, right) is taken from block (k+1, left) once the handle knows the matrices are block-symmetric
    (one blocking check on the first call).  Same products in the same order: against a solve that reads all three columns (a CAPTURED
    solve on a fresh handle does: the check cannot run during capture) bit-identical with the block-Jacobi preconditioner — S's right blocks
    ARE the transposed left ones — and equal to the round-off of the symmetric-stair Pinv, whose two halves are the same triple product
    associated two ways; with NaN in every right block and "assume_symmetric" the same bits as the latched solve; a structurally asymmetric
    Pinv is solved as given (three columns, oracle) and latches the handle the other way."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 3, 30
    k = synth.make_kkt(N, B, 4100 + N)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", 0)                       # (the streaming kernel: since round 5 the default beyond N = 32 is the clustered row-per-lane kernel)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    sol.solve_f64(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 1 and sol.get_option("last_kernel_family") == 3
    # all three columns: a fresh handle whose first solve is captured
    sol3 = PcgSolver(N, max_batch=B)
    sol3.set_option("cluster", 0)
    lam3 = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it3 = torch.zeros(B, dtype=torch.int32, device="cuda"); ex3 = torch.zeros(B, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            sol3.solve_f64(dS, dP, dg, lam3, cfg, iters=it3, exits=ex3)
    graph.replay()
    torch.cuda.synchronize()
    assert sol3.get_option("symmetry_state") == 0
    assert relinf(lam.cpu().numpy(), lam3.cpu().numpy()) < 1e-9
    lamj, lamj3 = torch.zeros_like(lam), torch.zeros_like(lam)
    sol.solve_f64(dS, dP, dg, lamj, cfg, "jacobi")
    with torch.cuda.stream(side):
        graphj = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graphj, stream=side):
            sol3.solve_f64(dS, dP, dg, lamj3, cfg, "jacobi", iters=it3, exits=ex3)
    graphj.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lamj.cpu().numpy(), lamj3.cpu().numpy())
    # right blocks unread
    Sp, Pp = S.copy().reshape(B, N, 3, 196), Pinv.copy().reshape(B, N, 3, 196)
    Sp[:, :, 2, :] = np.nan; Pp[:, :, 2, :] = np.nan
    solp = PcgSolver(N, max_batch=B)
    solp.set_option("cluster", 0)
    solp.set_option("assume_symmetric", 1)
    lamp = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    solp.solve_f64(dev(Sp.reshape(B, -1)), dev(Pp.reshape(B, -1)), dg, lamp, cfg)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lam.cpu().numpy(), lamp.cpu().numpy())
    # asymmetric Pinv
    Pa = Pinv.copy().reshape(B, N, 3, 196)
    Pa[:, :-1, 2, :] *= 1.25
    Pa = Pa.reshape(B, -1)
    sola = PcgSolver(N, max_batch=B)
    sola.set_option("cluster", 0)
    lama = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    sola.solve_f64(dS, dev(Pa), dg, lama, cfg)
    torch.cuda.synchronize()
    assert sola.get_option("symmetry_state") == 2
    for b in range(B):
        ref = orc.pcg(S[b], Pa[b], g[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        assert relinf(lama[b].cpu().numpy(), ref) < 1e-8


@pytest.mark.parametrize("N", [33, 64, 100, 128, 200, 256])
@pytest.mark.parametrize("precond", ["ss", "jacobi"])
def test_f64_clustered_row_per_lane_kernel_vs_oracle(orc, N, precond):
    """linsys_t = double beyond N = 32 (round 5): the row-per-lane kernel across ceil(N / 32) CUs of one XCD (pcg_rpl_cluster_f64.hip.h) —
    S and Pinv stay in registers, members exchange their boundary knots' entries and the wave partials once per pass.  Fixed iteration counts
    against the oracle's float64 iterate (cold and warm start), the tolerance exit, and the semantics of every other kernel: flags, counts,
    lambda untouched when converged, run-to-run determinism."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 3, 30
    k = synth.make_kkt(N, B, 6100 + N)
    S, Pinv, g = synth.form_schur(k, precond=precond, dtype=np.float64, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("pcg_lqk", 0)                         # (N <= 64 would run the lane-quad kernel of one CU, tests/test_gpu_lqk64.py)
    rng = np.random.default_rng(N)
    G = (N + 31) // 32
    for lam0 in (np.zeros((B, n * N)), 0.1 * rng.standard_normal((B, n * N))):
        lam = dev(lam0.copy())
        it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), precond)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == 8 and sol.get_option("last_kernel_cluster") == G
        assert sol.get_option("cluster_fixups") == 0          # (solved by the clusters themselves, not by the streaming fix-up behind them)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
        lamh = lam.cpu().numpy()
        for b in range(B):
            ref = orc.pcg(np.nan_to_num(S[b]), np.nan_to_num(Pinv[b]), g[b], lam0[b], N, K, 0.0, precond)["lam"]
            assert relinf(lamh[b], ref) < 1e-9, (b, relinf(lamh[b], ref))
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


@pytest.mark.parametrize("N,B", [(64, 300), (128, 150)])
def test_f64_clustered_kernel_draws_trajectories_from_the_queue(orc, N, B):
    """More trajectories than resident clusters (128 at N = 64, 64 at N = 128): the persistent clusters draw from the queue; copies of one system
    solve to the same bits wherever they land; against the streaming kernel ("cluster" = 0) to round-off; no trajectory left to the fix-up."""
    from mpcgpu_amd import PcgSolver, pcg_config
    K = 25
    k = synth.make_kkt(N, 5, 6200 + N)
    S5, P5, g5 = synth.form_schur(k, dtype=np.float64)
    rep = (B + 4) // 5
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S5, P5, g5))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("pcg_lqk", 0)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 8 and (it.cpu().numpy() == K).all() and sol.get_option("cluster_fixups") == 0
    lamh = lam.cpu().numpy()
    for b in range(5, B):
        np.testing.assert_array_equal(lamh[b], lamh[b % 5])
    sol.set_option("cluster", 0)
    lam_s = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    sol.solve_f64(dS, dP, dg, lam_s, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 3
    assert relinf(lamh[:5], lam_s.cpu().numpy()[:5]) < 1e-9


@pytest.mark.parametrize("family", [10, 8])
def test_f64_clustered_kernel_without_fixup_reports_abandoned_trajectories_only_when_there_are_any(family):
    """(family 10: the lane-quad clusters of two CUs, the default at N = 128; 8: the row-per-lane clusters of four, "pcg_lqk" = 0.)
    "cluster_fixup" = 0: no streaming launch behind the clusters — an undisturbed call still solves everything itself (iteration counts,
    not the 0xFFFFFFFF / flag 2 of an abandoned trajectory), which is what says the clusters did the work."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 128, 70, 12
    k = synth.make_kkt(N, B, 6300)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_fixup", 0)
    sol.set_option("pcg_lqk", -1 if family == 10 else 0)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == family
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all() and torch.isfinite(lam).all()


@pytest.mark.parametrize("family", [10, 8])
def test_f64_clusters_that_cannot_get_their_cus_fall_to_the_streaming_fixup(orc, family):
    """Members of a cluster must be co-resident.  With another stream holding the chip (4,000-iteration float solves on every CU) a 4-member double
    cluster may not find its CUs: its members give up after the bounded spin, leave lambda alone, and the streaming kernel behind the launch solves
    exactly the trajectories whose completion count is short ("cluster_fixups" counts them).  Either way the caller gets the oracle's answer."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 128, 4, 10
    k = synth.make_kkt(N, B, 6400)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("pcg_lqk", -1 if family == 10 else 0)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    if family == 10:                                     # (the latch's one blocking check happens on an undisturbed first call)
        sol.solve_f64(dS, dP, dg, torch.zeros(B, n * N, dtype=torch.float64, device="cuda"), pcg_config(pcg_exit_tol=0.0, pcg_max_iter=1))
    blocker = torch.cuda.Stream()
    ncu = sol.get_option("num_cus")
    big = PcgSolver(128, max_batch=4 * ncu)
    kb = synth.make_kkt(128, 2, 3)
    Sb, Pb, gb = synth.form_schur(kb, precond="ss")
    bS, bP, bg = (dev(a).repeat(2 * ncu, 1).contiguous() for a in (Sb, Pb, gb))
    bl = torch.zeros(4 * ncu, n * 128, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(blocker):
        for _ in range(3):
            big.solve(bS, bP, bg, bl, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=4000), "ss")
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    fixed = sol.get_option("cluster_fixups")
    assert sol.get_option("last_kernel_family") == family and 0 <= fixed <= B
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    lamh = lam.cpu().numpy()
    for b in range(B):
        ref = orc.pcg(S[b], Pinv[b], g[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        assert relinf(lamh[b], ref) < 1e-9, (b, fixed, relinf(lamh[b], ref))


def test_f64_clustered_kernel_in_a_graph_captured_on_a_fresh_handle():
    """The first double solve of a fresh handle may be a captured one: the clustered kernel's scratch is allocated by mpcg_create, the launch is
    three kernel nodes (fill, clusters, gated streaming fix-up); replays reproduce the eager solve bit for bit."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 128, 6, 20
    k = synth.make_kkt(N, B, 6500)
    S, Pinv, g = synth.form_schur(k, dtype=np.float64)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda"); ex = torch.zeros(B, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            lam.zero_()
            sol.solve_f64(dS, dP, dg, lam, cfg, iters=it, exits=ex)
    outs = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        outs.append(lam.clone())
    assert sol.get_option("last_kernel_family") == 8 and sol.get_option("cluster_fixups") == 0 and (it.cpu().numpy() == K).all()
    eager = PcgSolver(N, max_batch=B)
    eager.set_option("pcg_lqk", 0)                          # (the captured call could not run the latch's check: the kernel that reads all three columns)
    lam_e = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    eager.solve_f64(dS, dP, dg, lam_e, cfg)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, lam_e)


@pytest.mark.parametrize("N,rho", [(16, 1e-1), (64, 1e-2), (128, 1e-3)])
def test_f64_kkt_to_step_pipeline_vs_dense_kkt_solve(N, rho):
    """The whole linear-system step in double against the DEFINITION — numpy's dense solve of the regularised KKT system
    [G C^T; C 0][dz; lam] = [g; c] (sign conventions of include/common/dz.cuh / linsys_setup.cuh), no restatement of the Schur algebra on the
    comparison side: KKT blocks -> mpcg_form_schur_f64 (walking kernel) -> mpcg_pcg_solve_f64 (row-per-lane kernel, lane-quad kernel to N = 64, clusters beyond) ->
    mpcg_compute_dz_f64, to 1e-7 of the dense solution (cond up to 1e7 at rho = 1e-3) and C dz = c to 1e-9."""
    from mpcgpu_amd import PcgSolver, pcg_config
    m, B = 7, 2
    k = synth.make_kkt(N, B, 31337 + N)
    G, C, g, c = synth.pack_kkt_dense(k, np.float64)
    sol = PcgSolver(N, max_batch=B)
    dG, dC, dg, dc = dev(G), dev(C), dev(g), dev(c)
    S, P, gam = sol.form_schur(dG, dC, dg, dc, rho, "ss")
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    it, ex = sol.solve_f64(S, P, gam, lam, pcg_config(pcg_exit_tol=1e-26, pcg_max_iter=20000), "ss")
    dz = sol.compute_dz(dG, dC, dg, lam)
    torch.cuda.synchronize()
    assert (ex.cpu().numpy() == 0).all() and sol.get_option("cluster_fixups") == 0
    assert sol.get_option("last_kernel_family") == (5 if N <= 32 else 9 if N <= 64 else 10)
    dz, lam = dz.cpu().numpy(), lam.cpu().numpy()
    nz = (n + m) * N - m
    for b in range(B):
        Cm, Gm, gz = np.zeros((n * N, nz)), np.zeros((nz, nz)), np.zeros(nz)
        Cm[:n, :n] = np.eye(n)
        for kk in range(N):
            o = kk * (n + m)
            Gm[o:o + n, o:o + n] = k.Q[b, kk] + rho * np.eye(n)
            gz[o:o + n] = k.q[b, kk]
            if kk < N - 1:
                Gm[o + n:o + n + m, o + n:o + n + m] = k.R[b, kk] + rho * np.eye(m)
                gz[o + n:o + n + m] = k.r[b, kk]
            if kk > 0:
                po = (kk - 1) * (n + m)
                Cm[kk * n:(kk + 1) * n, po:po + n] = -k.A[b, kk - 1]
                Cm[kk * n:(kk + 1) * n, po + n:po + n + m] = -k.Bm[b, kk - 1]
                Cm[kk * n:(kk + 1) * n, o:o + n] = np.eye(n)
        K = np.block([[Gm, Cm.T], [Cm, np.zeros((n * N, n * N))]])
        sol64 = np.linalg.solve(K, np.concatenate([gz, k.c[b].reshape(-1)]))
        assert relinf(lam[b], sol64[nz:]) < 1e-7, (b, relinf(lam[b], sol64[nz:]))
        assert relinf(dz[b], sol64[:nz]) < 1e-7, (b, relinf(dz[b], sol64[:nz]))
        assert np.abs(Cm @ dz[b] - k.c[b].reshape(-1)).max() < 1e-9 * max(1.0, np.abs(dz[b]).max())


def test_f64_seeded_fuzz_of_the_double_kernels(orc):
    """80 random (N in 2..256, batch, preconditioner, warm start, iteration cap) double solves on the default policy — row-per-lane kernel up to
    N = 32, the lane-quad kernel up to 64, its clustered form beyond (ragged member sizes: N = 33 -> 16 + 17 knots, 97 -> 24 + 24 + 24 + 25, ...) — against the oracle's float64
    iterate; nothing left to the fix-up."""
    from mpcgpu_amd import PcgSolver, pcg_config
    rng = np.random.default_rng(20250930)
    fam = {}
    for case in range(80):
        N = int(rng.choice([rng.integers(2, 33), rng.integers(33, 65), rng.integers(65, 129), rng.integers(129, 257)], p=[0.2, 0.25, 0.3, 0.25]))
        B = int(rng.integers(1, 6))
        pc = str(rng.choice(["ss", "jacobi"]))
        K = int(rng.integers(1, min(35, 14 * N)))
        k = synth.make_kkt(N, B, int(rng.integers(1 << 30)))
        S, P, g = synth.form_schur(k, precond=pc, dtype=np.float64, poison_unused=True)
        lam0 = 0.1 * rng.standard_normal((B, n * N)) if rng.random() < 0.5 else np.zeros((B, n * N))
        sol = PcgSolver(N, max_batch=B)
        lam = dev(lam0.copy())
        it, ex = sol.solve_f64(dev(S), dev(P), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        f = sol.get_option("last_kernel_family")
        fam[f] = fam.get(f, 0) + 1
        assert f == (5 if N <= 32 else 9 if N <= 64 else 10), (N, f)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all() and sol.get_option("cluster_fixups") == 0
        lamh = lam.cpu().numpy()
        for b in range(B):
            Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
            ref = orc.pcg(Sz, Pz, g[b], lam0[b], N, K, 0.0, pc)["lam"]
            e = relinf(lamh[b], ref)
            if e >= 1e-9:     # small systems iterated past convergence are chaotic in float64 too (tools/_prof/small_f64.py): the oracle's own 1-ulp spread
                pert = lambda a_: a_ * (1 + 1.1e-16 * rng.standard_normal(a_.shape))
                band = max(relinf(orc.pcg(pert(Sz), Pz, pert(g[b]), pert(lam0[b]), N, K, 0.0, pc)["lam"], ref) for _ in range(8))
                assert e <= 20 * band, (case, N, B, pc, K, b, e, band)
    assert fam.get(5, 0) >= 8 and fam.get(9, 0) >= 10 and fam.get(10, 0) >= 25, fam
