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
    sol.solve(Sr.contiguous(), Pr.contiguous(), dev(g), lam32, cfg, pc)
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == K).all()
    lam16, lam32 = lam16.cpu().numpy(), lam32.cpu().numpy()
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
