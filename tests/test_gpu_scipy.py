"""GPU: the PCG kernels against THIRD-PARTY code directly — scipy.sparse.linalg.cg (float64, M = the preconditioner, every iterate captured by the
callback) — with no same-author restatement in between (the oracle and the goldens are same-author; tests/test_oracle.py pins the
oracle on scipy, this file pins the kernels on it).  S and Pinv are stored negated (negative definite): scipy gets A = -S, M = -Pinv, b = -gamma,
the same system and the same recurrences (include/pcg/sqp.cuh:137-150 semantics).
  * linsys_t = double — the row-per-lane kernel (N = 32), its clustered form (N = 64, 128: two / four CUs per trajectory) and the streaming kernel:
    the iterate after K = 1 .. 40 iterations within 1e-10 of scipy's (cond ~1e5 amplifies the last bits of two float64 summation orders);
  * float — every kernel family of the default policy at K = 1, 2, 3, before CG has amplified float32 rounding: within 2e-4."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import default_family, relinf

pytestmark = pytest.mark.gpu
n = 14


def scipy_iterates(S, P, g, lam0, N, pc, KM):
    import scipy.sparse.linalg as sla
    Sd = synth.bd_to_dense(np.nan_to_num(S).astype(np.float64), N)
    Pd = synth.bd_to_dense(np.nan_to_num(P).astype(np.float64), N)
    if pc == "jacobi":
        Pd = Pd * np.kron(np.eye(N), np.ones((n, n)))
    xs = []
    sla.cg(sla.aslinearoperator(-Sd), -g.astype(np.float64), x0=lam0.astype(np.float64).copy(), rtol=0.0, atol=0.0, maxiter=KM, M=sla.aslinearoperator(-Pd),
           callback=lambda xk: xs.append(xk.copy()))
    assert len(xs) == KM
    return xs


@pytest.mark.parametrize("N,family,cluster", [(32, 5, -1), (64, 9, -1), (64, 8, 2), (128, 10, -1), (128, 8, 4), (64, 3, 0)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_double_kernels_against_scipy_cg_at_every_iteration(N, family, cluster, pc):
    from mpcgpu_amd import PcgSolver, pcg_config
    k = synth.make_kkt(N, 1, 8000 + N)
    S, P, g = (a[0] for a in synth.form_schur(k, precond=pc, dtype=np.float64))
    rng = np.random.default_rng(N)
    lam0 = 0.1 * rng.standard_normal(n * N)
    KM = 40
    xs = scipy_iterates(S, P, g, lam0, N, pc, KM)
    sol = PcgSolver(N, max_batch=1)
    sol.set_option("cluster", cluster)
    if family == 8:
        sol.set_option("pcg_lqk", 0)                      # (the clustered row-per-lane kernel instead of the lane-quad kernels)
    dS, dP, dg = (torch.from_numpy(a.reshape(1, -1).copy()).cuda() for a in (S, P, g))
    worst = 0.0
    for K in range(1, KM + 1):
        lam = torch.from_numpy(lam0.reshape(1, -1).copy()).cuda()
        it, ex = sol.solve_f64(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == family and int(it.item()) == K
        e = relinf(lam.cpu().numpy()[0], xs[K - 1])
        worst = max(worst, e)
        assert e < 1e-10, (N, pc, K, e)            # (measured worst: 2e-12)
    print(f"double N={N} {pc} family {family}: worst distance from scipy's iterate over K = 1..{KM}: {worst:.2e}")


@pytest.mark.parametrize("N", [8, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_float_kernels_against_scipy_cg_first_iterations(N, pc):
    from mpcgpu_amd import PcgSolver, pcg_config
    family = default_family(N, pc)
    k = synth.make_kkt(N, 1, 8100 + N)
    S, P, g = (a[0] for a in synth.form_schur(k, precond=pc))
    rng = np.random.default_rng(N)
    lam0 = (0.3 * rng.standard_normal(n * N)).astype(np.float32)
    xs = scipy_iterates(S, P, g, lam0, N, pc, 3)
    sol = PcgSolver(N, max_batch=1)
    dS, dP, dg = (torch.from_numpy(a.reshape(1, -1).copy()).cuda() for a in (S, P, g))
    for K in (1, 2, 3):
        lam = torch.from_numpy(lam0.reshape(1, -1).copy()).cuda()
        it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == family and int(it.item()) == K
        e = relinf(lam.cpu().numpy()[0], xs[K - 1])
        print(f"float N={N} {pc} K={K}: {e:.2e}")
        assert e < 2e-4, (N, pc, K, e)             # (measured worst: 5.4e-5 — float32 rounding through the cancellation of alpha on a random warm start; a wrong block, row, sign or reduction shows as O(1))


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_float_kernels_against_scipy_cg_inside_a_band_scipy_itself_sets(N, pc):
    """Deeper into the iteration float32 CG drifts whatever the summation order; how far is a property of the SYSTEM, and scipy measures it
    without any code of ours: its float64 iterates on inputs perturbed by one float32 ulp (relative 6e-8 gaussian, four trials) move by `band`.
    A correct float32 kernel lands within a small multiple of it (measured: <= 2.8 x over 32 (system, K) pairs, tools/_prof/band_probe.py); the
    limit is 8 x, floor 2e-5.  K = 10, 25, 50 from lambda0 = 0."""
    family = default_family(N, pc)
    from mpcgpu_amd import PcgSolver, pcg_config
    k = synth.make_kkt(N, 1, 8200 + N)
    S, P, g = (a[0] for a in synth.form_schur(k, precond=pc))
    lam0 = np.zeros(n * N, np.float32)
    KM = 50
    xs = scipy_iterates(S, P, g, lam0, N, pc, KM)
    rng = np.random.default_rng(1)
    pert = []
    for _ in range(4):
        Sp = S.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(S.shape))
        gp = g.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(g.shape))
        pert.append(scipy_iterates(Sp, P, gp, lam0, N, pc, KM))
    sol = PcgSolver(N, max_batch=1)
    dS, dP, dg = (torch.from_numpy(a.reshape(1, -1).copy()).cuda() for a in (S, P, g))
    for K in (10, 25, 50):
        lam = torch.zeros(1, n * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == family and int(it.item()) == K
        e = relinf(lam.cpu().numpy()[0], xs[K - 1])
        band = max(relinf(p_[K - 1], xs[K - 1]) for p_ in pert)
        assert e <= max(2e-5, 8 * band), (N, pc, K, e, band)
