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
