"""GPU: the clustered lane-pair kernel (pcg_lpk_cluster.hip.h, family 7) — fp32 horizons beyond 128 knots, G = ceil(N / 128) workgroups
on G CUs of one XCD per trajectory, one hand-off per matrix pass: forced member counts against the single-workgroup kernel and the oracle,
the automatic policy on ragged horizons, batches larger than the chip holds clusters, determinism and batch-composition independence,
warm starts, and the fix-up path when a member cannot become resident."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf, rel_residual

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", [129, 131, 255, 257, 300, 512, 640])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
@pytest.mark.parametrize("l2", [1, 0])
def test_auto_policy_ragged_horizons_vs_oracle(orc, N, pc, l2):
    """Default handle, no knobs: N > 128 runs family 7 with G = ceil(N / 128) members of floor/ceil(N / G) knots."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B = 2
    k = synth.make_kkt(N, B, 9100 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam0 = np.random.default_rng(N).normal(0, 0.2, (B, n * N)).astype(np.float32)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_l2", l2)               # 1 (default): L2-resident hand-offs inside an XCD; 0: write-through hand-offs
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    for K in (2, 25):
        lam = dev(lam0)
        it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == (N + 127) // 128
        assert sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes(14, N)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
        for t in range(B):
            Sz, Pz = np.nan_to_num(S[t]), np.nan_to_num(Pinv[t])
            r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[t].astype(np.float64), lam0[t].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, Sz, Pz, g[t], lam0[t], N, K, pc, r64["lam"], trials=2)
            assert relinf(lam.cpu().numpy()[t], r64["lam"]) <= max(2e-5 if K == 2 else 1e-3, 4 * band)


def test_full_batch_in_several_launches_is_deterministic_and_composition_independent(orc):
    """N = 256, 300 trajectories: 128 clusters fit the chip, so the call is three launches (128 + 128 + 44), each followed by its
    fix-up launch.  Same answer twice, the same answer for a sub-batch, tolerance exits and counts like the CPU restatement."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 256, 300
    k = synth.make_kkt(N, B, 77)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=60)
    runs = []
    for _ in range(2):
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
        torch.cuda.synchronize()
        runs.append((lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy()))
    assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == 2
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    assert (runs[0][2] <= 1).all() and (runs[0][1] <= 60).all()          # no cluster gave up (flag 2 / 0xFFFFFFFF)
    sub = [0, 127, 128, 255, 256, 299]
    lam_s = torch.zeros(len(sub), n * N, device="cuda")
    it_s, _ = sol.solve(dev(S[sub]), dev(Pinv[sub]), dev(g[sub]), lam_s, cfg, "ss")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lam_s.cpu().numpy(), runs[0][0][sub])
    np.testing.assert_array_equal(it_s.cpu().numpy(), runs[0][1][sub])
    for t in (0, 128, 299):
        Sz, Pz = np.nan_to_num(S[t]), np.nan_to_num(Pinv[t])
        r32 = orc.pcg(Sz, Pz, g[t], np.zeros(n * N, np.float32), N, 60, 1e-4, "ss")
        assert abs(int(runs[0][1][t]) - int(r32["iters"])) <= max(3, 0.12 * r32["iters"]), (t, runs[0][1][t], r32["iters"])
        assert rel_residual(S[t], g[t], runs[0][0][t], N) <= 2 * rel_residual(S[t], g[t], r32["lam"], N) + 1e-6


def test_warm_start_and_zero_iterations():
    """lambda in/out: a second call starting from the converged lambda exits at once with 0 iterations and leaves lambda alone."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 384, 3
    k = synth.make_kkt(N, B, 5)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-2, pcg_max_iter=2000), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 7 and (ex.cpu().numpy() == 0).all() and (it.cpu().numpy() > 0).all()
    before = lam.clone()
    it2, ex2 = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-2, pcg_max_iter=2000), "ss")
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == 0).all() and (ex2.cpu().numpy() == 0).all()
    assert torch.equal(lam, before)


def test_options_round_trip_and_selection():
    """"cluster": -1 auto / 0 off / G forced; the retired kernels' options are gone (MPCG_ERR_INVALID, handle still usable); L2-resident and
    write-through hand-offs give the same bits."""
    from mpcgpu_amd import PcgSolver, pcg_config
    from mpcgpu_amd._lib import MpcgError
    N = 256
    k = synth.make_kkt(N, 1, 2)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=1)
    assert sol.get_option("cluster") == -1 and sol.get_option("cluster_fixup") == 1
    for gone in ("cluster_lpb", "cluster_lpk", "cluster_waves", "cluster_adj", "pcg_lpb"):
        with pytest.raises(MpcgError):
            sol.set_option(gone, 1)
    with pytest.raises(MpcgError):
        sol.set_option("cluster", 33)
    fam = {}
    for v in (-1, 0, 2):
        sol.set_option("cluster", v)
        lam = torch.zeros(1, n * N, device="cuda")
        it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=7), "ss")
        torch.cuda.synchronize()
        fam[v] = (sol.get_option("last_kernel_family"), int(it.item()), lam.cpu().numpy())
    assert fam[-1][0] == 7 and fam[2][0] == 7 and fam[0][0] == 0 and all(f[1] == 7 for f in fam.values())
    np.testing.assert_array_equal(fam[-1][2], fam[2][2])
    # hand-offs through the XCD's L2 (default, when the members of a cluster share an XCD) or write-through: same arithmetic, same bits
    assert sol.get_option("cluster_l2") == 1
    sol.set_option("cluster_l2", 0)
    lam = torch.zeros(1, n * N, device="cuda")
    sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=7), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 7
    np.testing.assert_array_equal(lam.cpu().numpy(), fam[2][2])
    assert relinf(fam[0][2][0], fam[2][2][0]) <= 2e-3          # two kernels, same PCG, 7 iterations: fp32 round-off of the products and inner products


@pytest.mark.parametrize("N,G,B", [(128, 2, 3), (200, 3, 2), (256, 4, 2), (512, 8, 2), (64, 2, 1), (512, 4, 40), (256, 2, 70)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_forced_member_counts_match_oracle_and_single_workgroup(orc, N, G, B, pc):
    """"cluster" = G on horizons one CU could hold as well: same PCG, inner products summed per member then across members, so iterates match
    the single-workgroup kernel / the float64 oracle within the fp32 band; flags, counts and lambda conventions are identical; results are
    deterministic."""
    from mpcgpu_amd import PcgSolver, pcg_config
    k = synth.make_kkt(N, B, 7700 + N + G)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam0 = np.random.default_rng(N).normal(0, 0.2, (B, n * N)).astype(np.float32)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    out = {}
    for mode in ("cluster", "single"):
        sol = PcgSolver(N, max_batch=B)
        sol.set_option("cluster", G if mode == "cluster" else 0)
        for K, tol in ((3, 0.0), (30, 0.0), (400, 1e-3)):
            lam = dev(lam0)
            it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=tol, pcg_max_iter=K), pc)
            torch.cuda.synchronize()
            out[(mode, K)] = (lam.cpu().numpy(), it.cpu().numpy().astype(np.int64), ex.cpu().numpy())
            if mode == "cluster":                      # the kernel under test really ran
                assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == G and sol.get_option("last_kernel_waves") == 8
        if mode == "cluster":      # deterministic: bitwise identical on a second run
            lam = dev(lam0)
            sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-3, pcg_max_iter=400), pc)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(lam.cpu().numpy(), out[("cluster", 400)][0])
    for K in (3, 30):
        c, s1 = out[("cluster", K)], out[("single", K)]
        assert (c[1] == K).all() and (c[2] == 1).all(), (c[1], c[2])          # no timeout (would be 0xFFFFFFFF / 2)
        for t in (range(B) if B <= 3 else (0, B // 2, B - 1)):
            r64 = orc.pcg(np.nan_to_num(S[t]).astype(np.float64), np.nan_to_num(Pinv[t]).astype(np.float64),
                          g[t].astype(np.float64), lam0[t].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, S[t], Pinv[t], g[t], lam0[t], N, K, pc, r64["lam"], trials=2)
            tol = max(2e-5 if K == 3 else 1e-3, 4 * band)
            assert relinf(c[0][t], r64["lam"]) <= tol and relinf(c[0][t], s1[0][t]) <= tol
    c, s1 = out[("cluster", 400)], out[("single", 400)]
    assert (c[2] == s1[2]).all() or (np.abs(c[1] - s1[1]) > 0).any()
    assert (np.abs(c[1] - s1[1]) <= np.maximum(3, 0.12 * s1[1])).all(), (c[1], s1[1])


def test_cluster_does_not_apply_beyond_eight_members():
    """G > 8 members (or more members than knots allow): the single-workgroup kernel runs."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 32, 20
    k = synth.make_kkt(N, B, 3)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", 16)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=5))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") != 7
    assert (it.cpu().numpy() == 5).all() and np.isfinite(lam.cpu().numpy()).all()


def test_cluster_falls_back_when_peers_are_not_resident(orc):
    """The cluster kernel (N > 128) needs all members of a trajectory resident.  Keep 255 of the 256 CUs busy with a
    long solve on another stream, then run a batch-1 N=256 solve (2 members of the clustered lane-pair kernel): the members that do get a CU give up
    after the bounded spin, and the fix-up launch re-solves the trajectory with the single-workgroup kernel —
    the caller gets a normal result (no flag 2, no 0xFFFFFFFF)."""
    from mpcgpu_amd import PcgSolver, pcg_config
    # the blocker: N=128 lane-pair kernel, one workgroup per CU, 255 trajectories, ~20 ms
    Nb, Bb = 128, 255
    kb = synth.make_kkt(Nb, 8, 5)
    Sb, Pb, gb = synth.form_schur(kb)
    rep = (Bb + 7) // 8
    dSb, dPb, dgb = (dev(np.tile(a, (rep, 1))[:Bb]) for a in (Sb, Pb, gb))
    blocker = PcgSolver(Nb, max_batch=Bb)
    lam_b = torch.zeros(Bb, n * Nb, device="cuda")
    N, K = 256, 25
    k = synth.make_kkt(N, 1, 6)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=1)
    assert sol.get_option("num_cus") == 256
    lam = torch.zeros(1, n * N, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        blocker.solve(dSb, dPb, dgb, lam_b, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=12000))
    with torch.cuda.stream(s2):
        it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 7 and sol.get_option("last_kernel_cluster") == 2
    assert int(it.item()) == K and int(ex.item()) == 1, (it, ex)
    Sz, Pz = np.nan_to_num(S[0]), np.nan_to_num(Pinv[0])
    r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[0].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")
    band = fp32_band(orc, Sz, Pz, g[0], np.zeros(n * N), N, K, "ss", r64["lam"])
    assert relinf(lam.cpu().numpy()[0], r64["lam"]) <= max(1e-3, 4 * band)
    # and undisturbed, the same call runs the cluster kernel to the same answer (fp32 round-off of the inner products)
    lam2 = torch.zeros(1, n * N, device="cuda")
    it2, ex2 = sol.solve(dS, dP, dg, lam2, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K))
    torch.cuda.synchronize()
    assert int(it2.item()) == K and relinf(lam2.cpu().numpy()[0], r64["lam"]) <= max(1e-3, 4 * band)


@pytest.mark.parametrize("precision", ["float", "float-128-lpk", "float-128-lqb", "double", "double-rpl"])
def test_fixup_starts_from_the_callers_lambda_even_if_some_members_finished(orc, precision):
    """A member that gives up AFTER its peers passed their last hand-off ("cluster_test_fail": the last member of cluster 0, at the write-back of
    its first trajectory): the peers have written their knots of lambda, the trajectory's completion count is short of G, the fix-up launch
    re-solves it — from the handle's copy of lambda0 (PcgArgs::lam0), not from the half-written array.  Warm start, few iterations: a wrong
    start would be far outside the band.  The peers then wait for the member that left, give up on their next trajectory: two fix-ups.
    "float-128-*": a forced "cluster" = 2 at 128 knots, whose fix-up launch is the lane-pair or (round 6) the lane-quad kernel reading the redo flags."""
    from mpcgpu_amd import PcgSolver, pcg_config
    dbl = precision.startswith("double")
    short = precision.startswith("float-128")
    N, B, K = (128, 140 if precision == "double" else 70, 6) if dbl else (128 if short else 256, 140, 6)      # more trajectories than resident clusters (128 / 64 / 128): the queue is in use
    k = synth.make_kkt(N, 3, 8800 + N)
    S3, P3, g3 = synth.form_schur(k, dtype=np.float64 if dbl else np.float32)
    rep = (B + 2) // 3
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S3, P3, g3))
    lam0 = np.random.default_rng(5).normal(0, 0.3, (B, n * N)).astype(S.dtype)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    if precision == "double-rpl":
        sol.set_option("pcg_lqk", 0)                      # the row-per-lane clusters of four CUs instead of the lane-quad clusters of two
    if short:
        sol.set_option("cluster", 2)
        sol.set_option("pcg_lqb", int(precision.endswith("lqb")))
        sol.solve(dS, dP, dg, dev(lam0.copy()), pcg_config(pcg_exit_tol=0.0, pcg_max_iter=1))      # (the symmetry latch: the first call's launches are guarded)
        torch.cuda.synchronize()
        assert sol.get_option("symmetry_state") == 1
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    solve = sol.solve_f64 if dbl else sol.solve
    lam_ok = dev(lam0.copy())
    solve(dS, dP, dg, lam_ok, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == (7 if short else {"float": 7, "double": 10, "double-rpl": 8}[precision]) and sol.get_option("cluster_fixups") == 0
    sol.set_option("cluster_test_fail", 1)
    lam = dev(lam0.copy())
    it, ex = solve(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    fixed = sol.get_option("cluster_fixups")
    assert 1 <= fixed <= 2, fixed
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    lamh, okh = lam.cpu().numpy(), lam_ok.cpu().numpy()
    changed = [b for b in range(B) if not np.array_equal(lamh[b], okh[b])]
    assert 0 in changed and len(changed) <= fixed, changed     # (re-solved by another kernel: other bits; everything else untouched)
    for b in changed:
        r64 = orc.pcg(S[b].astype(np.float64), Pinv[b].astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, K, 0.0, "ss")
        tol = 1e-9 if dbl else max(2e-5, 4 * fp32_band(orc, S[b], Pinv[b], g[b], lam0[b], N, K, "ss", r64["lam"], trials=2))
        assert relinf(lamh[b], r64["lam"]) <= tol, (b, relinf(lamh[b], r64["lam"]), tol)
    sol.set_option("cluster_test_fail", 0)
    lam2 = dev(lam0.copy())
    solve(dS, dP, dg, lam2, cfg)
    torch.cuda.synchronize()
    assert torch.equal(lam2, lam_ok) and sol.get_option("cluster_fixups") == fixed


@pytest.mark.parametrize("precision,N", [("float", 256), ("double", 128), ("double-rpl", 128), ("double", 384)])
def test_without_a_fixup_launch_an_abandoned_trajectory_is_always_reported(precision, N):
    """No fix-up launch behind the clusters ("cluster_fixup" = 0; double N > 350: the streaming kernel cannot hold the horizon, so there is none
    either): a trajectory whose completion count is short of G must come back as d_iters = 0xFFFFFFFF / d_max_iter_exit = 2 — also when member 0
    passed its last hand-off and stored a valid-looking count (the member that "fails" here does so at its write-back, after every hand-off), and
    also when the cluster that would have drawn it has stopped drawing.  Everything else is solved and bit-identical to the
    undisturbed call."""
    from mpcgpu_amd import PcgSolver, pcg_config
    dbl = precision != "float"
    B, K = 140 if N < 384 else 48, 6
    k = synth.make_kkt(N, 3, 8900 + N)
    S3, P3, g3 = synth.form_schur(k, dtype=np.float64 if dbl else np.float32)
    rep = (B + 2) // 3
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S3, P3, g3))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    if precision == "double-rpl":
        sol.set_option("pcg_lqk", 0)
    if N < 384:
        sol.set_option("cluster_fixup", 0)                # (N = 384 in double: on, but no fix-up kernel exists for that horizon)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    solve = sol.solve_f64 if dbl else sol.solve
    dt = torch.float64 if dbl else torch.float32
    lam_ok = torch.zeros(B, n * N, dtype=dt, device="cuda")
    it0, ex0 = solve(dS, dP, dg, lam_ok, cfg)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == {"float": 7, "double": 10, "double-rpl": 8}[precision]
    assert (it0.cpu().numpy() == K).all() and (ex0.cpu().numpy() == 1).all()
    sol.set_option("cluster_test_fail", 1)
    lam = torch.zeros(B, n * N, dtype=dt, device="cuda")
    it, ex = solve(dS, dP, dg, lam, cfg)
    torch.cuda.synchronize()
    ith, exh = it.cpu().numpy().view(np.uint32), ex.cpu().numpy()          # (d_iters is uint32_t: 0xFFFFFFFF, not -1)
    gone = np.flatnonzero(ith == np.uint32(0xFFFFFFFF))
    assert 1 <= len(gone) <= 2 and 0 in gone, gone         # the trajectory the member left + the one its peers then wait on
    assert (exh[gone] == 2).all() and (np.delete(exh, gone) == 1).all() and (np.delete(ith, gone) == K).all()
    keep = np.setdiff1d(np.arange(B), gone)
    assert torch.equal(lam[keep], lam_ok[keep])
    assert sol.get_option("cluster_fixups") == 0
    sol.set_option("cluster_test_fail", 0)
    lam2 = torch.zeros(B, n * N, dtype=dt, device="cuda")
    it2, _ = solve(dS, dP, dg, lam2, cfg)
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == K).all() and torch.equal(lam2, lam_ok)
