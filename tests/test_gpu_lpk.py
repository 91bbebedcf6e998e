"""GPU (MI355X): the two lower-triangle single-CU PCG kernels — the lane-pair-per-knot kernel (mpcgpu_amd/csrc/pcg_lpk.hip.h, round 3, family 6)
and the lane-quad kernel with both matrices in every wavefront (pcg_lqb.hip.h, round 6, family 11: the default for SS up to 128 knots and for
both preconditioners up to 64) — through the C ABI, against the CPU oracle, the golden vectors and the single-workgroup kernels.  Every test
of this file runs once per kernel (fixture `P`).

It reads only the block lower triangle (left + diagonal blocks) of S and Pinv: S[k,right] is the bitwise transpose of
S[k+1,left] in the reference's construction (include/pcg/linsys_setup.cuh:536-557) and the symmetric-stair Pinv
(:97-136) is symmetric up to the rounding of two independently formed products (~1e-7), which moves a fixed-K
iterate by 1e-6..3e-5 — inside the stated fp32 tolerance (tests/test_gpu_parity.py docstring).  Every test here
poisons ALL right blocks with NaN.
"""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import exit_iter_bounds, fp32_band, fp32_iters_band, golden, relinf, rel_residual

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class _Kern(tuple):
    """(PcgSolver, pcg_config) + which of the two lower-triangle single-CU kernels the test runs: "lpk" = the lane-pair kernel (family 6),
    "lqb" = round 6's lane-quad kernel with both matrices in every wavefront (family 11, pcg_lqb.hip.h) — same contract, same tests."""
    kern = "lpk"

    @property
    def family(self):
        return 11 if self.kern == "lqb" else 6

    def pin(self, sol):
        """Force this kernel on a handle (the automatic policy picks by horizon, batch and preconditioner)."""
        sol.set_option("pcg_lpk", 1)
        sol.set_option("pcg_lqb", 1 if self.kern == "lqb" else 0)

    def default_policy(self, sol):
        """Leave the automatic policy alone (it picks the lane-quad kernel for SS) or take the lane-quad kernel out of it."""
        if self.kern == "lpk":
            sol.set_option("pcg_lqb", 0)


@pytest.fixture(scope="module", params=["lpk", "lqb"])
def P(request):
    from mpcgpu_amd import PcgSolver, pcg_config, _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.load()
    k = _Kern((PcgSolver, pcg_config))
    k.kern = request.param
    return k


def poison_right(M, N, jacobi=False):
    """NaN in every block the kernel must not read: all right blocks, (0,left); block-Jacobi: every off-diagonal."""
    M = np.array(M, np.float32).reshape(-1, N, 3, 196).copy()
    M[:, :, 2] = np.nan
    M[:, 0, 0] = np.nan
    if jacobi:
        M[:, :, 0] = np.nan
    return M.reshape(M.shape[0], -1)


def lpk_solve(P, N, S, Pinv, g, lam0, max_iter, tol, pc="ss"):
    PcgSolver, pcg_config = P
    B = S.shape[0]
    sol = PcgSolver(N, max_batch=B)
    P.pin(sol)
    # these tests hand over matrices whose right block column is NaN (it must never be read): a lower-triangle-only caller says so,
    # otherwise the handle's first solves would CHECK the right blocks against the left ones (tests/test_gpu_contract.py: the symmetry latch)
    sol.set_option("assume_symmetric", 1)
    lam = dev(np.asarray(lam0, np.float32))
    it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=tol, pcg_max_iter=max_iter), pc)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == P.family and sol.get_option("last_kernel_waves") == (2 if N <= 32 else 4 if N <= 64 else 8)
    assert P.kern != "lqb" or N <= 32 or sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes(14, N)      # (pcgSharedMemSize mirrors the default policy)
    return lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy()


@pytest.mark.parametrize("N", [2, 3, 17, 32, 33, 63, 64, 65, 100, 127, 128])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_lpk_fixed_iterations_vs_oracle(P, orc, N, pc):
    """Ragged / odd / maximum horizons, 3 trajectories each, 30 fixed iterations vs the float64 oracle inside the
    fp32 band; bitwise run-to-run determinism."""
    B, K = 3, 30
    k = synth.make_kkt(N, B, 8100 + N)
    S, Pinv, g = synth.form_schur(k, precond="ss")
    Sp, Pp = poison_right(S, N), poison_right(Pinv, N, jacobi=(pc == "jacobi"))
    lam0 = np.zeros((B, n * N), np.float32)
    lam, it, ex = lpk_solve(P, N, Sp, Pp, g, lam0, K, 0.0, pc)
    lam2, _, _ = lpk_solve(P, N, Sp, Pp, g, lam0, K, 0.0, pc)
    np.testing.assert_array_equal(lam, lam2)
    assert (it == K).all() and (ex == 1).all() and np.isfinite(lam).all()
    for b in range(B):
        r64 = orc.pcg(S[b].astype(np.float64), Pinv[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, pc)
        band = fp32_band(orc, S[b], Pinv[b], g[b], np.zeros(n * N), N, K, pc, r64["lam"])
        assert relinf(lam[b], r64["lam"]) <= max(1e-3, 4 * band), (N, pc, b, relinf(lam[b], r64["lam"]), band)


@pytest.mark.parametrize("N", [8, 32])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_lpk_vs_golden(P, orc, N, pc):
    G = golden(N)
    Sp = poison_right(G["S"], N)
    Pp = poison_right(G["Pinv"], N, jacobi=(pc == "jacobi"))
    for K in (5, 20, 50):
        lam, it, ex = lpk_solve(P, N, Sp, Pp, G["gamma"].reshape(1, -1), np.zeros((1, n * N)), K, 0.0, pc)
        want = G[f"lam_{pc}_K{K}"]
        band = fp32_band(orc, G["S"], G["Pinv"], G["gamma"], np.zeros(n * N), N, K, pc, want)
        assert it[0] == K and ex[0] == 1
        assert relinf(lam[0], want) <= max(1e-3, 4 * band), (K, relinf(lam[0], want), band)
    # tolerance exit
    want_it = int(G[f"iters_tol_{pc}"])
    lam, it, ex = lpk_solve(P, N, Sp, Pp, G["gamma"].reshape(1, -1), np.zeros((1, n * N)), 5000, 1e-4, pc)
    assert ex[0] == 0
    lo, hi = exit_iter_bounds(orc, G["S"], G["Pinv"], G["gamma"], np.zeros(n * N), N, 5000, 1e-4, pc)
    assert lo <= int(it[0]) <= hi and lo <= want_it <= hi, (int(it[0]), lo, hi, want_it)
    cpu32 = orc.pcg(G["S"], G["Pinv"], G["gamma"], np.zeros(n * N, np.float32), N, 5000, 1e-4, pc)
    assert rel_residual(G["S"], G["gamma"], lam[0], N) <= 2 * rel_residual(G["S"], G["gamma"], cpu32["lam"], N) + 1e-6


@pytest.mark.parametrize("N", [64, 128])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_lpk_first_iterations_tight(P, orc, N, pc):
    """K = 1, 2, 3 from a random warm start: rounding has not been amplified yet — a wrong block, transpose,
    sign or reduction would show as O(1); the bound is a few 1e-5."""
    k = synth.make_kkt(N, 2, 531 + N)
    S, Pinv, g = synth.form_schur(k)
    lam0 = np.random.default_rng(N).normal(0, 0.5, (2, n * N)).astype(np.float32)
    for K in (1, 2, 3):
        lam, it, ex = lpk_solve(P, N, poison_right(S, N), poison_right(Pinv, N, pc == "jacobi"), g, lam0, K, 0.0, pc)
        for b in range(2):
            r64 = orc.pcg(S[b].astype(np.float64), Pinv[b].astype(np.float64), g[b].astype(np.float64),
                          lam0[b].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, S[b], Pinv[b], g[b], lam0[b], N, K, pc, r64["lam"])
            assert band < 5e-4
            assert relinf(lam[b], r64["lam"]) <= max(2e-5, 4 * band), (K, b, relinf(lam[b], r64["lam"]), band)


def test_lpk_flags_warm_start_and_r_p_outputs(P, orc):
    G = golden(8)
    N = 8
    args = (poison_right(G["S"], N), poison_right(G["Pinv"], N), G["gamma"].reshape(1, -1))
    lam, it, ex = lpk_solve(P, N, *args, np.zeros((1, n * N)), 7, 1e-4)
    assert it[0] == 7 and ex[0] == 1
    lam, it, ex = lpk_solve(P, N, *args, G["lam_warm"].reshape(1, -1), 20, 0.0)
    band = fp32_band(orc, G["S"], G["Pinv"], G["gamma"], G["lam_warm"], N, 20, "ss", G["lam_warm_ss_K20"])
    assert relinf(lam[0], G["lam_warm_ss_K20"]) <= max(1e-3, 4 * band)
    lam0 = G["lam_direct"].astype(np.float32).reshape(1, -1)
    lam, it, ex = lpk_solve(P, N, *args, lam0, 50, 1e-2)          # already converged: lambda untouched bit for bit
    assert it[0] == 0 and ex[0] == 0
    np.testing.assert_array_equal(lam, lam0)
    lam, it, ex = lpk_solve(P, N, *args, np.zeros((1, n * N)), 0, 1e-9)
    assert it[0] == 0 and ex[0] == 1 and (lam == 0).all()
    # the reference's 12-argument entry: d_r / d_p receive the final residual and search direction
    G = golden(32)
    N = 32
    sol = P[0](N)
    P.pin(sol)
    sol.set_option("assume_symmetric", 1)      # (right blocks are NaN here: a lower-triangle-only caller)
    d_lambda = torch.zeros(n * N, device="cuda")
    d_r = torch.full((n * N,), 7.0, device="cuda")
    d_p = torch.full((n * N,), 7.0, device="cuda")
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_ex = torch.zeros(1, dtype=torch.bool, device="cuda")
    sol.solve_ref(dev(poison_right(G["S"], N)[0]), dev(poison_right(G["Pinv"], N)[0]), dev(G["gamma"]), d_lambda, d_r, d_p,
                  torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), d_it, d_ex, 20, 0.0)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == P.family
    r64 = orc.pcg(G["S"].astype(np.float64), G["Pinv"].astype(np.float64), G["gamma"].astype(np.float64), np.zeros(n * N), N, 20, 0.0, "ss")
    assert relinf(d_r.cpu().numpy(), r64["r"]) < 5e-2 and relinf(d_p.cpu().numpy(), r64["p"]) < 5e-2


def test_lpk_reference_style_pinv_is_inside_the_band(P, orc):
    """Pinv formed the reference's way (oracle restatement of include/pcg/linsys_setup.cuh:97-136, bit-exact twin of
    mpcg_form_schur): its right blocks are NOT bitwise transposes of the next row's left blocks (two independent
    fp32 products), S's are.  The kernel reads the left blocks only; the result stays inside the fp32 band of the
    solve with the blocks as given."""
    N, K = 128, 40
    k = synth.make_kkt(N, 2, 99)
    Gd, Cd, gd, cd = synth.pack_kkt_dense(k, np.float32)
    for b in range(2):
        S, Pinv, g = orc.form_schur(Gd[b].copy(), Cd[b], gd[b], cd[b], N, synth.RHO_INIT, ss=True)[:3]
        S, Pinv = np.nan_to_num(S), np.nan_to_num(Pinv)
        S4, P4 = S.reshape(N, 3, 14, 14), Pinv.reshape(N, 3, 14, 14)
        assert all(np.array_equal(S4[j, 2], S4[j + 1, 0].T) for j in range(N - 1))          # bitwise symmetric
        asym = max(np.abs(P4[j, 2] - P4[j + 1, 0].T).max() for j in range(N - 1)) / np.abs(Pinv).max()
        assert 0 < asym < 1e-6
        lam, it, ex = lpk_solve(P, N, poison_right(S, N), poison_right(Pinv, N), g.reshape(1, -1), np.zeros((1, n * N)), K, 0.0)
        r64 = orc.pcg(S.astype(np.float64), Pinv.astype(np.float64), g.astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")
        band = fp32_band(orc, S, Pinv, g, np.zeros(n * N), N, K, "ss", r64["lam"])
        assert relinf(lam[0], r64["lam"]) <= max(1e-3, 4 * band), (relinf(lam[0], r64["lam"]), band)


def test_lpk_agrees_with_single_workgroup_kernel(P):
    """Same systems through the lane-per-block kernel and the streaming single-workgroup kernel <16,0,2>: different
    summation orders, same solve — iteration counts within 5 %, solutions within fp32 CG drift."""
    PcgSolver, pcg_config = P
    N, B = 96, 40
    k = synth.make_kkt(N, B, 77)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=1e-5, pcg_max_iter=400)
    res = []
    for lpb in (True, False):
        sol = PcgSolver(N, max_batch=B)
        if not lpb:
            sol.set_option("pcg_waves", 16); sol.set_option("pcg_reg_rows", 0); sol.set_option("pcg_lds_rows", 0)
        else:
            P.default_policy(sol)
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == (P.family if lpb else 0)
        res.append((lam.cpu().numpy(), it.cpu().numpy().astype(int), ex.cpu().numpy()))
    assert np.abs(res[0][1] - res[1][1]).max() <= max(3, int(0.05 * res[1][1].max()))
    assert (res[0][2] == res[1][2]).all()
    r0 = [rel_residual(S[b], g[b], res[0][0][b], N) for b in (0, 13, 39)]
    r1 = [rel_residual(S[b], g[b], res[1][0][b], N) for b in (0, 13, 39)]
    assert all(a <= 2 * c + 1e-6 for a, c in zip(r0, r1))


def test_lpk_config4_workload_n128_batch1024(P, orc):
    """BASELINE config 4 on one GPU = the workload bench.py times: N=128, SS, batch 1024, lambda0 = 0, max_iter 167,
    exit_tol 1e-4, default configuration.  Oracle band on 8 sampled trajectories, determinism, residual decrease
    for all 1024 (size-independent property, evaluated with the library's own SpMV)."""
    PcgSolver, pcg_config = P
    N, B, MI, TOL = 128, 1024, 167, 1e-4
    k = synth.make_kkt(N, B, 4000)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=TOL, pcg_max_iter=MI)
    outs = []
    for rep in range(2):
        sol = PcgSolver(N, max_batch=B)
        P.default_policy(sol)
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == P.family and sol.get_option("last_kernel_waves") == 8
        outs.append((lam, it.cpu().numpy().astype(int), ex.cpu().numpy()))
    assert torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] == outs[1][1]).all()
    lam, it, ex = outs[0]
    assert ((it == MI) == (ex == 1)).all() and (it <= MI).all() and (it > 0).all()
    # residual decrease for every trajectory: |gamma - S lam| < |gamma - S 0|
    res = dg - sol.bt_spmv(dS, lam)
    assert torch.isfinite(lam).all()
    rr = (res.double().norm(dim=1) / dg.double().norm(dim=1)).cpu().numpy()
    assert (rr < 1.0).all(), rr.max()
    lam_h = lam.cpu().numpy()
    for b in (0, 1, 255, 256, 511, 640, 1000, 1023):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(Pinv[b])
        r32 = orc.pcg(Sz, Pz, g[b], np.zeros(n * N, np.float32), N, MI, TOL, "ss")
        lo, hi = exit_iter_bounds(orc, Sz, Pz, g[b], np.zeros(n * N, np.float32), N, MI, TOL, "ss", trials=4)
        assert lo <= it[b] <= hi, (b, it[b], lo, hi)
        if it[b] == MI and r32["iters"] == MI:        # same fixed count: compare the iterates through the band
            r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, MI, 0.0, "ss")
            band = fp32_band(orc, Sz, Pz, g[b], np.zeros(n * N), N, MI, "ss", r64["lam"], trials=2)
            assert relinf(lam_h[b], r64["lam"]) <= max(1e-3, 4 * band), (b, relinf(lam_h[b], r64["lam"]), band)
        assert rel_residual(S[b], g[b], lam_h[b], N) <= 2 * rel_residual(S[b], g[b], r32["lam"], N) + 1e-6


def test_batch_composition_independence(P):
    PcgSolver, pcg_config = P
    N, B = 128, 300
    k = synth.make_kkt(N, B, 31)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    cfg = (60, 1e-5)
    lam, it, ex = lpk_solve(P, N, S, Pinv, g, np.zeros((B, n * N)), *cfg)
    sub = [5, 257, 299]
    lam_s, it_s, _ = lpk_solve(P, N, S[sub], Pinv[sub], g[sub], np.zeros((3, n * N)), *cfg)
    np.testing.assert_array_equal(lam_s, lam[sub])
    np.testing.assert_array_equal(it_s, it[sub])


def test_half_build_policy_on_short_horizons(P, orc):
    """16 < N <= 32 (round 5): calls of at least 2.5 trajectories per CU run the lane-pair kernel's HALF build (family 6, two wavefronts, four
    workgroups per CU; since round 6, by default, the lane-quad kernel's 32-knot build: family 11, two wavefronts, four workgroups per CU),
    smaller ones and N <= 16 the row-per-lane kernel (family 5).  Copies of four systems solve to the same bits wherever
    they land; a sub-batch below the threshold comes from the other kernel and agrees inside the float32 band."""
    PcgSolver, pcg_config = P
    N, K = 24, 20
    sol = PcgSolver(N, max_batch=4096)
    P.default_policy(sol)
    ncu = sol.get_option("num_cus")
    B = (5 * ncu + 1) // 2
    k = synth.make_kkt(N, 4, 515)
    S4, P4, g4 = synth.form_schur(k)
    rep = (B + 3) // 4
    S, Pinv, g = (np.tile(a_, (rep, 1))[:B] for a_ in (S4, P4, g4))
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == P.family and sol.get_option("last_kernel_waves") == 2 and (it.cpu().numpy() == K).all()
    lamh = lam.cpu().numpy()
    for b in range(4, B):
        np.testing.assert_array_equal(lamh[b], lamh[b % 4])
    for b in range(4):
        r64 = orc.pcg(S4[b].astype(np.float64), P4[b].astype(np.float64), g4[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")["lam"]
        band = fp32_band(orc, S4[b], P4[b], g4[b], np.zeros(n * N, np.float32), N, K, "ss", r64)
        assert relinf(lamh[b], r64) <= max(1e-3, 4 * band)
    Bs = (5 * ncu + 1) // 2 - 1                              # one below 2.5 trajectories per CU
    lam_s = torch.zeros(Bs, n * N, device="cuda")
    sol.solve(dS[:Bs], dP[:Bs], dg[:Bs], lam_s, cfg, "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 5
    assert relinf(lam_s.cpu().numpy()[:4], lamh[:4]) < 2e-3
    sol16 = PcgSolver(16, max_batch=B)
    k16 = synth.make_kkt(16, 2, 516)
    S2, P2, g2 = (np.tile(a_, ((B + 1) // 2, 1))[:B] for a_ in synth.form_schur(k16))
    sol16.solve(dev(S2), dev(P2), dev(g2), torch.zeros(B, n * 16, device="cuda"), cfg, "ss")
    torch.cuda.synchronize()
    assert sol16.get_option("last_kernel_family") == 5
