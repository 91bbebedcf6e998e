"""GPU: the row-per-lane kernel (pcg_rpl.hip.h) — short horizons (N <= 64; the reference's real-time case N = 32, one trajectory):
a DPP row per knot, a lane per matrix row, vectors in registers, two barriers per iteration.  Default for N <= 32 and for
latency-sized calls up to N = 64.  Same PCG as every other kernel: fixed-count iterates inside the float64 oracle's fp32 band,
tolerance exits / counts / flags / warm starts like the CPU restatement, never-written blocks never read, deterministic."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf, rel_residual, exit_iter_bounds

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def solve(N, S, Pinv, g, lam0, max_iter, tol, pc="ss", waves=0, want=None):
    from mpcgpu_amd import PcgSolver, pcg_config
    B = S.shape[0]
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("pcg_rpl", 1)
    sol.set_option("rpl_waves", waves)
    lam = dev(np.asarray(lam0, np.float32))
    it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=tol, pcg_max_iter=max_iter), pc)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 5
    if want is not None:
        assert (sol.get_option("last_kernel_waves"), sol.get_option("last_kernel_reg_rows")) == want
    return lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy()


@pytest.mark.parametrize("N,waves,shape", [(2, 0, (4, 1)), (3, 0, (4, 1)), (15, 0, (4, 1)), (16, 8, (8, 1)), (17, 0, (8, 1)), (32, 0, (8, 1)), (32, 4, (4, 2)),
                                           (32, 16, (16, 1)), (33, 0, (8, 2)), (47, 16, (16, 1)), (63, 0, (8, 2)), (64, 0, (8, 2)), (64, 16, (16, 1))])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_rpl_fixed_iterations_vs_oracle(orc, N, waves, shape, pc):
    """Ragged horizons and every compiled (waves, slots) shape, 3 trajectories, 30 fixed iterations vs the float64 oracle inside the fp32
    band; NaN in the two never-written blocks (and in every off-diagonal Pinv block for block-Jacobi); bitwise run-to-run."""
    B, K = 3, 30
    k = synth.make_kkt(N, B, 8800 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    if pc == "jacobi":
        Pp = np.array(Pinv).reshape(B, N, 3, 196).copy()
        Pp[:, :, 0] = np.nan
        Pp[:, :, 2] = np.nan
        Pinv_in = Pp.reshape(B, -1)
    else:
        Pinv_in = Pinv
    lam0 = np.zeros((B, n * N), np.float32)
    lam, it, ex = solve(N, S, Pinv_in, g, lam0, K, 0.0, pc, waves, shape)
    lam2, _, _ = solve(N, S, Pinv_in, g, lam0, K, 0.0, pc, waves, shape)
    np.testing.assert_array_equal(lam, lam2)
    assert (it == K).all() and (ex == 1).all() and np.isfinite(lam).all()
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(Pinv[b])
        r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, pc)
        band = fp32_band(orc, Sz, Pz, g[b], np.zeros(n * N), N, K, pc, r64["lam"])
        assert relinf(lam[b], r64["lam"]) <= max(1e-3, 4 * band), (N, pc, b, relinf(lam[b], r64["lam"]), band)


@pytest.mark.parametrize("N", [8, 32, 64])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_rpl_first_iterations_tight(orc, N, pc):
    """K = 1, 2, 3 from a random warm start: a wrong block, row, neighbour or reduction would show as O(1)."""
    k = synth.make_kkt(N, 2, 631 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam0 = np.random.default_rng(N).normal(0, 0.5, (2, n * N)).astype(np.float32)
    for K in (1, 2, 3):
        lam, it, ex = solve(N, S, Pinv, g, lam0, K, 0.0, pc)
        for b in range(2):
            Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(Pinv[b])
            r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, Sz, Pz, g[b], lam0[b], N, K, pc, r64["lam"])
            assert relinf(lam[b], r64["lam"]) <= max(2e-5, 4 * band), (K, b, relinf(lam[b], r64["lam"]), band)


def test_rpl_tolerance_exit_counts_flags_and_outputs(orc):
    """Tolerance exit on the reference's config-2 shape (N = 32, block-Jacobi, cap 173, tol 5e-6): iteration counts inside the CPU
    restatement's band, flags, r / p outputs consistent with the returned lambda, warm start = 0 iterations."""
    N, B = 32, 6
    k = synth.make_kkt(N, B, 55)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam, it, ex = solve(N, S, Pinv, g, np.zeros((B, n * N)), 173, 5e-6, "jacobi")
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(Pinv[b])
        lo, hi = exit_iter_bounds(orc, Sz, Pz, g[b], np.zeros(n * N, np.float32), N, 173, 5e-6, "jacobi", trials=4)
        assert lo <= it[b] <= hi, (b, it[b], lo, hi)
        assert ex[b] == (1 if it[b] == 173 else 0)
        r32 = orc.pcg(Sz, Pz, g[b], np.zeros(n * N, np.float32), N, 173, 5e-6, "jacobi")
        assert rel_residual(S[b], g[b], lam[b], N) <= 2 * rel_residual(S[b], g[b], r32["lam"], N) + 1e-6
    # the reference's own argument list (one trajectory, d_r / d_p filled on exit): r and p of the float64 oracle after the same 20 iterations
    from mpcgpu_amd import PcgSolver
    sol = PcgSolver(N, max_batch=1)
    assert sol.get_option("pcg_rpl") == -1                      # default policy: N = 32 -> this kernel
    d_lambda = torch.zeros(n * N, device="cuda")
    d_r = torch.full((n * N,), 7.0, device="cuda")
    d_p = torch.full((n * N,), 7.0, device="cuda")
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_ex = torch.zeros(1, dtype=torch.bool, device="cuda")
    sol.solve_ref(dev(S[0]), dev(Pinv[0]), dev(g[0]), d_lambda, d_r, d_p, torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), d_it, d_ex, 20, 0.0)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 5 and int(d_it.item()) == 20
    Sz, Pz = np.nan_to_num(S[0]), np.nan_to_num(Pinv[0])
    r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[0].astype(np.float64), np.zeros(n * N), N, 20, 0.0, "ss")
    assert relinf(d_r.cpu().numpy(), r64["r"]) < 5e-2 and relinf(d_p.cpu().numpy(), r64["p"]) < 5e-2
    # warm start from the converged lambda of a loose tolerance: exits at once, lambda untouched
    lam1, it1, ex1 = solve(N, S, Pinv, g, np.zeros((B, n * N)), 4000, 1e-2, "jacobi")
    assert (ex1 == 0).all()
    lam2, it2, ex2 = solve(N, S, Pinv, g, lam1, 4000, 1e-2, "jacobi")
    assert (it2 == 0).all() and (ex2 == 0).all()
    np.testing.assert_array_equal(lam1, lam2)


def test_rpl_default_policy_and_batch_composition():
    """No knobs: N <= 32 runs this kernel at any batch (8 waves x 1 slot up to one trajectory per CU, 4 x 2 beyond); 32 < N <= 64
    was its range for latency-sized calls until round 6 — from 33 knots the lane-quad kernel (family 11) now takes every call: it beats this kernel's one-trajectory latency there.  A trajectory's result does not depend on its neighbours in the batch."""
    from mpcgpu_amd import PcgSolver, pcg_config
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=60)
    for N, B, fam, shape in ((32, 1, 5, (8, 1)), (32, 600, 5, (4, 2)), (16, 600, 5, (4, 1)), (64, 3, 11, None), (64, 600, 11, None), (48, 600, 11, None), (36, 600, 11, None)):
        k = synth.make_kkt(N, min(B, 8), 3)
        S, Pinv, g = synth.form_schur(k, poison_unused=True)
        rep = (B + S.shape[0] - 1) // S.shape[0]
        S, Pinv, g = (np.tile(a, (rep, 1))[:B] for a in (S, Pinv, g))
        sol = PcgSolver(N, max_batch=B)
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, cfg, "ss")
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == fam, (N, B, sol.get_option("last_kernel_family"))
        if shape:
            assert (sol.get_option("last_kernel_waves"), sol.get_option("last_kernel_reg_rows")) == shape
        lam = lam.cpu().numpy()
        for b in range(8, B, 97):          # tiled inputs: trajectory b must equal trajectory b mod 8, bit for bit
            np.testing.assert_array_equal(lam[b], lam[b % 8])


@pytest.mark.parametrize("N,waves", [(5, 4), (16, 4), (17, 8), (32, 8)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_rpl_double_precision(orc, N, waves, pc):
    """linsys_t = double (USE_DOUBLES of the reference) at N <= 32 runs the same kernel in double (v_fmac_f64_dpp): against the float64
    oracle the iterate differs by summation order only."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 2, 25
    k = synth.make_kkt(N, B, 4100 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    sol = PcgSolver(N, max_batch=B)
    lam = torch.zeros(B, n * N, device="cuda", dtype=torch.float64)
    it, ex = sol.solve_f64(dev(S.astype(np.float64)), dev(Pinv.astype(np.float64)), dev(g.astype(np.float64)), lam,
                           pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 5 and sol.get_option("last_kernel_waves") == waves
    assert sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes_f64(14, N)
    assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]).astype(np.float64), np.nan_to_num(Pinv[b]).astype(np.float64)
        r64 = orc.pcg(Sz, Pz, g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, pc)
        assert relinf(lam.cpu().numpy()[b], r64["lam"]) <= 1e-8, (N, pc, b, relinf(lam.cpu().numpy()[b], r64["lam"]))
