"""GPU: round-3 contract items of the C ABI.
 * "check_symmetry": the lower-triangle kernels' precondition (include/mpcg.h, BLOCK SYMMETRY) verified per call; a caller whose
   Pinv is not block-symmetric is served by a three-column kernel and gets the three-column answer;
 * "cluster_fixups": fix-up launches report how many trajectories they re-solved;
 * cluster residency is sized per XCD: horizons whose member count does not divide an XCD's 32 CUs (G = 3, 5) at FULL batch need
   no fix-up (sized chip-wide they over-subscribed some XCDs)."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", [48, 128, 256])
def test_check_symmetry_passes_on_reference_matrices_and_catches_asymmetric_pinv(orc, N):
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 2, 12
    k = synth.make_kkt(N, B, 4400 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    if N <= 64:
        sol.set_option("pcg_rpl", 0)
    lam = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(S), dev(Pinv), dev(g), lam, cfg, "ss")
    fam_default = sol.get_option("last_kernel_family")
    assert fam_default in (6, 7, 11)                             # a lower-triangle kernel serves this call
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 1             # ... and the handle has latched "block-symmetric" (the asynchronous copy has landed)
    sol.set_option("check_symmetry", 1)
    lam1 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(S), dev(Pinv), dev(g), lam1, cfg, "ss")
    assert sol.get_option("last_symmetry_violations") == 0 and sol.get_option("last_kernel_family") == fam_default
    assert torch.equal(lam, lam1)
    # a caller-made, NOT block-symmetric preconditioner: right blocks scaled by 0.9 (still a valid, if worse, preconditioner
    # for a kernel that reads all three columns as the reference's does)
    Pa = np.array(Pinv, np.float32).reshape(B, N, 3, 196).copy()
    Pa[:, :, 2] *= 0.9
    Pa = Pa.reshape(B, -1)
    lam2 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(S), dev(Pa), dev(g), lam2, cfg, "ss")
    assert sol.get_option("last_symmetry_violations") == B * (N - 1)
    assert sol.get_option("last_kernel_family") in (0, 5)          # three-column kernels
    for b in range(B):
        ref = orc.pcg(S[b].astype(np.float64), Pa[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        band = fp32_band(orc, S[b], Pa[b], g[b], np.zeros(n * N), N, K, "ss", ref)
        assert relinf(lam2[b].cpu().numpy(), ref) <= max(1e-3, 4 * band)
    # with the debug check off, a handle that has LATCHED "symmetric" keeps its lower-triangle kernels: the same call silently gets the
    # lower-triangle solve — the documented limit of a per-handle latch (a caller that changes structure uses a fresh handle)
    sol.set_option("check_symmetry", 0)
    lam3 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(S), dev(Pa), dev(g), lam3, cfg, "ss")
    assert sol.get_option("last_kernel_family") == fam_default
    assert relinf(lam3.cpu().numpy(), lam.cpu().numpy()) < 1e-6 < relinf(lam3.cpu().numpy(), lam2.cpu().numpy())


@pytest.mark.parametrize("N,B", [(128, 3), (256, 2), (48, 300)])
def test_symmetry_latch_gives_an_asymmetric_pinv_the_three_column_solve_from_the_first_call(orc, N, B):
    """the symmetry contract by default.  A fresh handle, NO option touched, the reference-shaped 12-argument entry for the
    first trajectory (and the batched one for all): a caller-made Pinv whose right blocks are not the transposes of the left ones must get
    the PCG of its three block columns (oracle, precond_cols = 3) — from the very first solve, which runs guarded; after the asynchronous
    copy of the flag has landed the handle is on three-column kernels for good and mpcg_last_error() says why."""
    from mpcgpu_amd import PcgSolver, pcg_config
    K = 12
    k = synth.make_kkt(N, min(B, 4), 4500 + N)
    S, Pinv, g = (np.tile(a, ((B + 3) // 4, 1))[:B] for a in synth.form_schur(k, precond="ss"))
    Pa = np.array(Pinv, np.float32).reshape(B, N, 3, 196).copy()
    Pa[:, :, 2] *= 0.9
    Pa = Pa.reshape(B, -1)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    dS, dPa, dg = dev(S), dev(Pa), dev(g)
    refs = {}
    for b in (0, B - 1):
        refs[b] = orc.pcg(S[b].astype(np.float64), Pa[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]

    def ok(lam_row, b):
        band = fp32_band(orc, S[b], Pa[b], g[b], np.zeros(n * N), N, K, "ss", refs[b])
        return relinf(lam_row, refs[b]) <= max(1e-3, 4 * band)
    # 1. batched entry, first call of a fresh handle
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dPa, dg, lam, cfg, "ss")
    assert sol.get_option("last_kernel_family") in (6, 7, 11)        # the guarded launch of the lower-triangle family ...
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == K).all()
    assert ok(lam[0].cpu().numpy(), 0) and ok(lam[B - 1].cpu().numpy(), B - 1)      # ... whose result is the three-column solve
    assert sol.get_option("symmetry_state") == 2 and "block-symmetric" in sol.last_error()
    lam2 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dS, dPa, dg, lam2, cfg, "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") in (0, 5)            # latched: three-column kernels, no check kernel any more
    assert ok(lam2[0].cpu().numpy(), 0) and ok(lam2[B - 1].cpu().numpy(), B - 1)
    # 2. the reference's 12-argument entry on a fresh handle (one trajectory)
    if N > 64:
        sol1 = PcgSolver(N, max_batch=1)
        l1 = torch.zeros(n * N, device="cuda")
        r1, p1 = torch.empty(n * N, device="cuda"), torch.empty(n * N, device="cuda")
        v1, e1 = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        i1, x1 = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.uint8, device="cuda")
        sol1.solve_ref(dS[0], dPa[0], dg[0], l1, r1, p1, v1, e1, i1, x1, K, 0.0)
        torch.cuda.synchronize()
        assert int(i1.item()) == K and ok(l1.cpu().numpy(), 0)


@pytest.mark.parametrize("N,B", [(128, 3), (96, 2)])
def test_forced_cluster_on_a_short_horizon_keeps_the_symmetry_guarantee(orc, N, B):
    """"cluster" = 2 forced at N <= 128 on a FRESH handle with an asymmetric caller-made Pinv.  The guarded clustered
    launch gate-exits; its fix-up launch must then be a three-column kernel (it used to be the lower-triangle lane-pair kernel: the wrong system
    for the first solves) and gated exits must not be counted as abandoned trajectories."""
    from mpcgpu_amd import PcgSolver, pcg_config
    K = 12
    k = synth.make_kkt(N, B, 4700 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    Pa = np.array(Pinv, np.float32).reshape(B, N, 3, 196).copy()
    Pa[:, :, 2] *= 0.9
    Pa = Pa.reshape(B, -1)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", 2)
    sol.set_option("pcg_lpk", 0)
    sol.set_option("pcg_rpl", 0)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dev(S), dev(Pa), dev(g), lam, cfg, "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == 2
    assert (it.cpu().numpy() == K).all() and sol.get_option("cluster_fixups") == 0
    for b in range(B):
        ref = orc.pcg(S[b].astype(np.float64), Pa[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        band = fp32_band(orc, S[b], Pa[b], g[b], np.zeros(n * N), N, K, "ss", ref)
        assert relinf(lam[b].cpu().numpy(), ref) <= max(1e-3, 4 * band), (b, relinf(lam[b].cpu().numpy(), ref), band)
    assert sol.get_option("symmetry_state") == 2
    # and on the reference's own (symmetric) matrices the forced cluster still runs the clustered kernel, bit-identical before and after the latch
    sol2 = PcgSolver(N, max_batch=B)
    sol2.set_option("cluster", 2); sol2.set_option("pcg_lpk", 0); sol2.set_option("pcg_rpl", 0)
    la, lb = torch.zeros(B, n * N, device="cuda"), torch.zeros(B, n * N, device="cuda")
    sol2.solve(dev(S), dev(Pinv), dev(g), la, cfg, "ss")
    torch.cuda.synchronize()
    assert sol2.get_option("symmetry_state") == 1
    sol2.solve(dev(S), dev(Pinv), dev(g), lb, cfg, "ss")
    torch.cuda.synchronize()
    assert torch.equal(la, lb) and sol2.get_option("last_kernel_family") == 7 and sol2.get_option("cluster_fixups") == 0


@pytest.mark.parametrize("N,B", [(128, 3), (256, 2)])
def test_symmetry_latch_on_reference_matrices_costs_nothing_after_the_first_calls(N, B):
    """The reference's own (block-symmetric) matrices: the guarded first solve and the plain solves after the latch give the same bits,
    and NaN-poisoned right blocks — which a three-column kernel would choke on — are still fine once the handle knows."""
    from mpcgpu_amd import PcgSolver, pcg_config
    K = 15
    k = synth.make_kkt(N, B, 4600 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    sol = PcgSolver(N, max_batch=B)
    assert sol.get_option("symmetry_state") == 0
    lam = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(S), dev(Pinv), dev(g), lam, cfg, "ss")             # guarded
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 1 and sol.get_option("cluster_fixups") == 0
    Sp, Pp, _ = synth.form_schur(k, precond="ss", poison_unused=True)
    Sp = Sp.reshape(B, N, 3, 196).copy(); Pp = Pp.reshape(B, N, 3, 196).copy()
    Sp[:, :, 2] = np.nan; Pp[:, :, 2] = np.nan                       # the right block column is never read by the latched handle
    lam2 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dev(Sp.reshape(B, -1)), dev(Pp.reshape(B, -1)), dev(g), lam2, cfg, "ss")
    torch.cuda.synchronize()
    assert torch.equal(lam, lam2) and sol.get_option("last_kernel_family") in (6, 7, 11)


@pytest.mark.parametrize("N,G", [(384, 3), (640, 5), (256, 2)])
def test_full_batch_ragged_member_counts_need_no_fixup(N, G):
    """Every cluster of a persistent launch must be resident on ITS XCD: full-chip batches at G = 3 and 5."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 300, 6
    k = synth.make_kkt(N, 4, 5100 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    rep = (B + 3) // 4
    dS, dP, dg = (dev(a).repeat(rep, 1)[:B].contiguous() for a in (S, Pinv, g))
    sol = PcgSolver(N, max_batch=B)
    before = sol.get_option("cluster_fixups")
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == G
    assert sol.get_option("cluster_fixups") == before, "a cluster gave up on an otherwise idle GPU"
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    l = lam.cpu().numpy()
    np.testing.assert_array_equal(l[:4], l[4:8])            # the same four systems, whichever cluster drew them


def test_fixup_counter_counts_abandoned_trajectories():
    """A 2-member cluster whose peers cannot start (another stream holds the CUs) gives up; the fix-up launch re-solves and counts."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 256, 2, 8
    k = synth.make_kkt(N, B, 777)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    lam_ref = torch.zeros(B, n * N, device="cuda")
    sol.solve(dS, dP, dg, lam_ref, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("cluster_fixups") == 0
    # occupy the chip on a second stream with spinning workgroups holding most of every CU's LDS, then solve on the default stream
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
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), "ss")
    torch.cuda.synchronize()
    fixed = sol.get_option("cluster_fixups")
    assert 0 <= fixed <= B
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all() and np.isfinite(lam.cpu().numpy()).all()
    if fixed == 0:       # the clusters found their CUs anyway: same bits as the undisturbed solve
        assert torch.equal(lam, lam_ref)


def test_check_pcg_occupancy_mirrors_the_launch_policy():
    """checkPcgOccupancy (reference examples/track_iiwa_pcg.cu:24) reports the residency of the kernel a throughput-sized call really
    launches: row-per-lane kernel for N <= 32, lane-pair kernel up to 128 (one workgroup per CU at N > 64, two below),
    clusters per XCD beyond."""
    from mpcgpu_amd import PcgSolver
    ncu = PcgSolver(32, max_batch=1).get_option("num_cus")
    assert PcgSolver(32, max_batch=4096).checkPcgOccupancy() >= 2 * ncu          # 4 waves x 2 slots: at least two workgroups per CU
    assert PcgSolver(128, max_batch=1024).checkPcgOccupancy() == ncu             # 8 waves x 254 registers: one trajectory per CU
    assert PcgSolver(64, max_batch=2048).checkPcgOccupancy() == 2 * ncu
    assert PcgSolver(256, max_batch=1024).checkPcgOccupancy() == ncu // 2        # 2-member clusters
    assert PcgSolver(384, max_batch=1024).checkPcgOccupancy() == 8 * ((ncu // 8) // 3)
    assert PcgSolver(512, max_batch=1024).checkPcgOccupancy() == ncu // 4


@pytest.mark.parametrize("N", [32, 64])
def test_kernel_family_depends_on_batch_and_can_be_pinned(orc, N):
    """include/mpcg.h: which family serves a call depends on knot_points AND batch; families sum the inner products in different orders, so a
    trajectory solved alone and inside a large batch agrees to the fp32 band, not bit for bit — unless the family is pinned."""
    from mpcgpu_amd import PcgSolver, pcg_config
    K, Bbig = 25, 600
    k = synth.make_kkt(N, 4, 77 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    rep = Bbig // 4
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)

    def run(B, opts):
        sol = PcgSolver(N, max_batch=B)
        for k_, v_ in opts.items():
            sol.set_option(k_, v_)
        lam = torch.zeros(B, n * N, device="cuda")
        r = B // 4
        sol.solve(dev(np.tile(S, (r, 1))), dev(np.tile(Pinv, (r, 1))), dev(np.tile(g, (r, 1))), lam, cfg, "ss")
        torch.cuda.synchronize()
        return lam.cpu().numpy()[:4], sol.get_option("last_kernel_family"), sol.get_option("last_kernel_waves")
    small, fam_s, w_s = run(4, {})
    big, fam_b, w_b = run(Bbig, {})
    if N == 32:
        assert (fam_s, w_s) != (fam_b, w_b)                # 8 x 1 vs 4 x 2 row-per-lane shapes
    else:
        assert (fam_s, w_s) == (fam_b, w_b) == (11, 4)     # round 6: from 33 knots the lane-quad kernel takes every call (until then: row-per-lane vs lane-pair kernel)
        np.testing.assert_array_equal(small, big)
    for t in range(4):
        ref = orc.pcg(S[t].astype(np.float64), Pinv[t].astype(np.float64), g[t].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        band = fp32_band(orc, S[t], Pinv[t], g[t], np.zeros(n * N), N, K, "ss", ref, trials=2)
        assert relinf(small[t], ref) <= max(1e-3, 4 * band) and relinf(big[t], ref) <= max(1e-3, 4 * band)
    pin = {"pcg_rpl": 1, "rpl_waves": 8} if N == 32 else {"pcg_lpk": 1}
    a_, fa, _ = run(4, pin)
    b_, fb, _ = run(Bbig, pin)
    assert fa == fb
    np.testing.assert_array_equal(a_, b_)                  # pinned: bit-identical whatever the batch


@pytest.mark.parametrize("N,B", [(128, 600), (256, 300), (64, 1100), (32, 1300)])
def test_dispatch_order_hint_changes_no_result(N, B):
    """Option "sched_hint" (default on): calls with more trajectories than CUs dispatch them longest-expected-first, the expectation being
    the previous call's iteration counts (sched_order_kernel).  A scheduling matter only: with warm starts of very different quality
    (0 .. cap iterations) the hinted second and third calls return the bits of the un-hinted call, for every trajectory; a call with
    another batch size in between (no valid prediction) and a capped-at-zero call (all counts equal) do too."""
    from mpcgpu_amd import PcgSolver, pcg_config, synth
    import bench
    dev_ = torch.device("cuda", 0)
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = bench.build_inputs(sol, N, B, 3, "ss", dev_)
    cap = synth.pcg_max_iter(N)
    lam_star = torch.zeros(B, 14 * N, device=dev_)
    sol.solve(dS, dP, dg, lam_star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=3000))
    gen = torch.Generator(device=dev_); gen.manual_seed(5)
    amp = 10 ** (-7.0 + 6.0 * torch.rand(B, 1, device=dev_, generator=gen))
    lam0 = lam_star + amp * lam_star.abs().amax(dim=1, keepdim=True) * torch.randn(B, 14 * N, device=dev_, generator=gen)
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=cap)

    def run(batch=B, c=cfg):
        lam = lam0[:batch].clone()
        it, ex = sol.solve(dS[:batch], dP[:batch], dg[:batch], lam, c)
        torch.cuda.synchronize()
        return lam.cpu().numpy(), it.cpu().numpy().copy(), ex.cpu().numpy().copy()

    sol.set_option("sched_hint", 0)
    ref = run()
    assert ref[1].min() < 0.3 * cap and ref[1].max() >= 0.9 * cap          # the batch really is mixed
    sol.set_option("sched_hint", 1)
    assert sol.get_option("sched_hint") == 1
    fam = None
    for _ in range(3):                                                       # 1st: no prediction yet; 2nd, 3rd: ordered by the previous counts
        got = run()
        fam = sol.get_option("last_kernel_family")
        for a0, a1 in zip(ref, got):
            np.testing.assert_array_equal(a0, a1)
    assert fam in (5, 6, 7, 11)
    part = run(batch=B - 37)                                                 # another batch size: the stored order does not apply
    for a0, a1 in zip(ref, part):
        np.testing.assert_array_equal(a0[:B - 37], a1)
    zero = run(c=pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=0))             # all counts equal (0): any order
    assert (zero[1] == 0).all()
    got = run()                                                              # ... and the order it left behind is still a permutation
    for a0, a1 in zip(ref, got):
        np.testing.assert_array_equal(a0, a1)


def test_hbm_read_probe_contract():
    """mpcg_probe_hbm_read (measurement aid next to mpcg_bt_spmv): reads, writes nothing for ordinary data, rejects misaligned arguments."""
    from mpcgpu_amd import PcgSolver
    sol = PcgSolver(8, max_batch=1)
    src = torch.randn(1 << 20, device="cuda")
    sink = torch.full((1,), 7.0, device="cuda")
    sol.probe_hbm_read(src, sink)
    torch.cuda.synchronize()
    assert sink.item() == 7.0
    with pytest.raises(RuntimeError):
        sol.probe_hbm_read(src[1:], sink)                   # 4-byte aligned only


@pytest.mark.parametrize("precision", ["float", "double"])
def test_a_latch_that_resolved_on_block_jacobi_calls_reopens_at_the_first_ss_call(orc, precision):
    """block-Jacobi calls never show the handle a Pinv with off-diagonal blocks.  A handle whose latch resolved to "symmetric" on such
    calls must not hand a later SS call with a caller-made, structurally asymmetric Pinv to a lower-triangle kernel unchecked: the first SS
    call re-opens the latch, gets the three-column solve, and the handle ends up "violated".  "assume_symmetric" = 0 then forgets the violation
    (the sticky device flag is cleared by the next solve): the same handle latches symmetric again on the reference's own matrices."""
    from mpcgpu_amd import PcgSolver, pcg_config
    dbl = precision == "double"
    N, B, K = 128, 3, 12
    dt = np.float64 if dbl else np.float32
    k = synth.make_kkt(N, B, 4700)
    S, Pinv, g = synth.form_schur(k, precond="ss", dtype=dt)
    Pa = np.array(Pinv, dt).reshape(B, N, 3, 196).copy()
    Pa[:, :, 2] *= 0.9
    Pa = Pa.reshape(B, -1)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    dS, dP, dPa, dg = dev(S), dev(Pinv), dev(Pa), dev(g)
    sol = PcgSolver(N, max_batch=B)
    solve = sol.solve_f64 if dbl else sol.solve
    z = lambda: torch.zeros(B, n * N, dtype=torch.float64 if dbl else torch.float32, device="cuda")
    for _ in range(3):                                    # block-Jacobi calls: the latch resolves on S alone
        solve(dS, dP, dg, z(), cfg, "jacobi")
        torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 1
    lam = z()
    it, ex = solve(dS, dPa, dg, lam, cfg, "ss")
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == K).all()
    for b in (0, B - 1):
        ref = orc.pcg(S[b].astype(np.float64), Pa[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        tol = 1e-9 if dbl else max(1e-3, 4 * fp32_band(orc, S[b], Pa[b], g[b], np.zeros(n * N), N, K, "ss", ref))
        assert relinf(lam[b].cpu().numpy(), ref) <= tol, (b, relinf(lam[b].cpu().numpy(), ref), tol)
    solve(dS, dPa, dg, z(), cfg, "ss")
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 2 and "block-symmetric" in sol.last_error()
    # forget it: back to "unknown", the device flag is cleared by the next solve; the reference's matrices latch symmetric again
    sol.set_option("assume_symmetric", 0)
    assert sol.get_option("symmetry_state") == 0
    lam_ok = z()
    for _ in range(3):
        lam_ok.zero_()
        solve(dS, dP, dg, lam_ok, cfg, "ss")
        torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 1
    assert sol.get_option("last_kernel_family") == (10 if dbl else 11)
    ref = orc.pcg(S[0].astype(np.float64), Pinv[0].astype(np.float64), g[0].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
    assert relinf(lam_ok[0].cpu().numpy(), ref) <= (1e-9 if dbl else max(1e-3, 4 * fp32_band(orc, S[0], Pinv[0], g[0], np.zeros(n * N), N, K, "ss", ref)))


def test_double_symmetry_check_uses_a_double_tolerance(orc):
    """the float path's 1 % tolerance let a double S that is asymmetric at the 1e-3 level through to a lower-triangle kernel, which then
    solves a different system than the caller's.  The double check looks at 1e-6 of the pair's largest entry."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 64, 2, 10
    k = synth.make_kkt(N, B, 4800)
    S, Pinv, g = synth.form_schur(k, precond="ss", dtype=np.float64)
    Pa = np.array(Pinv).reshape(B, N, 3, 196).copy()
    Pa[:, :, 2] *= 1.0 + 1e-3
    Pa = Pa.reshape(B, -1)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, dtype=torch.float64, device="cuda")
    sol.solve_f64(dev(S), dev(Pa), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 2 and sol.get_option("last_kernel_family") != 9
    for b in range(B):
        ref = orc.pcg(S[b], Pa[b], g[b], np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        assert relinf(lam[b].cpu().numpy(), ref) <= 1e-9
