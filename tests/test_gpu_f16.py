"""GPU: half-precision MATRIX STORAGE (BASELINE config 5).  The kernel computes in fp32 on matrices that
were rounded to fp16 once; so its contract is: identical to the fp32 kernel / the oracle run on the ROUNDED
matrices (within the same fp32 band), and the conversion is exact round-to-nearest-even."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf, rel_residual

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_conversion_is_round_to_nearest_even():
    from mpcgpu_amd import PcgSolver
    sol = PcgSolver(8)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 100, 100003), [0.0, -0.0, 65504.0, 1e-8, 2049.0, 2051.0, 1 + 2 ** -11]]).astype(np.float32)
    x = np.resize(x, (x.size // 4) * 4)
    y = sol.to_f16(dev(x)).cpu().numpy()
    np.testing.assert_array_equal(y.view(np.uint16), x.astype(np.float16).view(np.uint16))


@pytest.mark.parametrize("N", [8, 33, 128, 256, 512])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_f16_storage_equals_fp32_solve_of_rounded_matrices(orc, N, pc):
    from mpcgpu_amd import PcgSolver, pcg_config
    B, K = 2, 25
    k = synth.make_kkt(N, B, 808 + N)
    S, Pinv, g = synth.form_schur(k, rho=1e-2, poison_unused=True)
    sol = PcgSolver(N, max_batch=B)
    dS, dP = dev(S), dev(Pinv)
    S16, P16 = sol.to_f16(dS), sol.to_f16(dP)
    Sr, Pr = S16.float(), P16.float()                       # the rounded matrices, back in fp32
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    lam16 = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve_f16(S16, P16, dev(g), lam16, cfg, pc)
    lam32 = torch.zeros(B, n * N, device="cuda")
    fam16 = sol.get_option("last_kernel_family")
    sol.set_option("pcg_lqb", 0)                           # (fp16 storage lives in the lane-pair kernels; the fp32 default up to 128 knots is the lane-quad kernel, another summation order)
    sol.solve(Sr.contiguous(), Pr.contiguous(), dev(g), lam32, cfg, pc)
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == K).all()
    assert N <= 36 or fam16 == sol.get_option("last_kernel_family")
    lam16, lam32 = lam16.cpu().numpy(), lam32.cpu().numpy()
    if N > 36:
        # the register-resident kernels (lane-pair N <= 128, clustered above) convert the blocks once, at the load: from there on the solve IS the
        # fp32 solve of the rounded matrices — same kernel, same registers, same bits
        assert sol.get_option("last_kernel_family") == (6 if N <= 128 else 7)
        np.testing.assert_array_equal(lam16, lam32)
    Srh, Prh = Sr.cpu().numpy(), Pr.cpu().numpy()
    for b in range(B):
        r64 = orc.pcg(np.nan_to_num(Srh[b]).astype(np.float64), np.nan_to_num(Prh[b]).astype(np.float64),
                      g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, pc)
        band = fp32_band(orc, Srh[b], Prh[b], g[b], np.zeros(n * N), N, K, pc, r64["lam"])
        assert relinf(lam16[b], r64["lam"]) <= max(1e-3, 4 * band)
        assert relinf(lam16[b], lam32[b]) <= max(1e-3, 4 * band)


def test_f16_storage_true_residual_floor(orc):
    """What the rounding costs against the ORIGINAL system: converged fp16-storage solves sit at a relative
    residual of order 2^-11 * cond-ish, far above fp32 storage — the number the bench's sweep reports."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 64, 4
    k = synth.make_kkt(N, B, 99)
    S, Pinv, g = synth.form_schur(k, rho=1e-1)
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    cfg = pcg_config(pcg_exit_tol=1e-10, pcg_max_iter=3000)
    lam32 = torch.zeros(B, n * N, device="cuda")
    sol.solve(dS, dP, dg, lam32, cfg)
    lam16 = torch.zeros(B, n * N, device="cuda")
    sol.solve_f16(sol.to_f16(dS), sol.to_f16(dP), dg, lam16, cfg)
    torch.cuda.synchronize()
    for b in range(B):
        r32 = rel_residual(S[b], g[b], lam32[b].cpu().numpy(), N)
        r16 = rel_residual(S[b], g[b], lam16[b].cpu().numpy(), N)
        assert r32 < 1e-4 and 1e-5 < r16 < 5e-2 and r16 > 3 * r32


def test_f16_storage_asymmetric_matrices_take_the_three_column_kernel(orc):
    """The lower-triangle kernels' symmetry latch also guards fp16 storage: a Pinv whose right blocks are not the transposes of the left ones is
    solved as given (three block columns), like the fp32 path (tests/test_gpu_contract.py)."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, K = 64, 3, 12
    k = synth.make_kkt(N, B, 4242)
    S, Pinv, g = synth.form_schur(k, rho=1e-2)
    Pa = Pinv.copy().reshape(B, N, 3, 196)
    Pa[:, :-1, 2, :] *= 1.25                                # right blocks scaled: no longer L_{k+1}^T
    Pa = Pa.reshape(B, -1)
    sol = PcgSolver(N, max_batch=B)
    S16, P16 = sol.to_f16(dev(S)), sol.to_f16(dev(Pa))
    cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
    lam = torch.zeros(B, n * N, device="cuda")
    sol.solve_f16(S16, P16, dev(g), lam, cfg, "ss")
    torch.cuda.synchronize()
    Sr, Pr = S16.float().cpu().numpy(), P16.float().cpu().numpy()
    for b in range(B):
        r64 = orc.pcg(Sr[b].astype(np.float64), Pr[b].astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, "ss")
        band = fp32_band(orc, Sr[b], Pr[b], g[b], np.zeros(n * N), N, K, "ss", r64["lam"])
        assert relinf(lam[b].cpu().numpy(), r64["lam"]) <= max(1e-3, 4 * band)
    for _ in range(3):                                      # the latch lands on a later call
        sol.solve_f16(S16, P16, dev(g), lam.zero_(), cfg, "ss")
        torch.cuda.synchronize()
    assert sol.get_option("symmetry_state") == 2
