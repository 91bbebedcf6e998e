"""CPU: the product's selectable CPU solver (LINSYS_SOLVE == 0 twin; mpcg_ldl_* in libmpcg_hip.so, mpcgpu_amd/csrc/ldl_host.hpp)
against the oracle: CSR pattern (include/utils/csr.cuh:40-73), the oracle's own QDLDL-style restatement, and the float64
block-tridiagonal ground truth.  Pure host code: runs without a GPU."""
import numpy as np
import pytest

from mpcgpu_amd import synth


@pytest.mark.parametrize("N", [2, 5, 32])
def test_pattern_matches_reference_csr(hiplib, orc, N):
    from mpcgpu_amd import QdldlSolver
    q = QdldlSolver(N)
    cp, ri = orc.prep_csr(N)
    assert q.nnz == (N - 1) * 196 + N * 105                  # include/qdldl/sqp.cuh:148
    np.testing.assert_array_equal(q.col_ptr, cp)
    np.testing.assert_array_equal(q.row_ind, ri)
    assert q.sum_lnz > 0


@pytest.mark.parametrize("N", [2, 8, 32, 128])
def test_ldl_solve_vs_oracle_and_ground_truth(hiplib, orc, N):
    from mpcgpu_amd import QdldlSolver
    B = 3
    k = synth.make_kkt(N, B, 600 + N)
    S, _, g = synth.form_schur(k)
    q = QdldlSolver(N)
    L = orc.LdlSolver(N, np.float32)
    for b in range(B):
        val = orc.bd_to_csr_lowertri(S[b], N)
        lam = q.solve_host(val, g[b])
        ref32 = L.solve(val, g[b])
        exact = orc.direct_solve(S[b], g[b], N)
        scale = np.abs(exact).max()
        # same algorithm, same precision as the oracle's restatement: agreement to float round-off of the factorisation
        assert np.abs(lam - ref32).max() / scale < 2e-3
        # and both are as far from the float64 solution as a float LDL^T at cond ~1e5 gets
        e_prod, e_orc = np.abs(lam - exact).max() / scale, np.abs(ref32 - exact).max() / scale
        assert e_prod < max(5e-2, 3 * e_orc), (e_prod, e_orc)
    # well-conditioned system: float LDL^T is accurate
    kw = synth.make_kkt(N, 1, 5)
    Sw, _, gw = synth.form_schur(kw, rho=1.0)
    lam = q.solve_host(orc.bd_to_csr_lowertri(Sw[0], N), gw[0])
    exact = orc.direct_solve(Sw[0], gw[0], N)
    assert np.abs(lam - exact).max() / np.abs(exact).max() < 5e-4


def test_ldl_rejects_zero_pivot(hiplib):
    from mpcgpu_amd import QdldlSolver, _lib
    q = QdldlSolver(4)
    with pytest.raises(_lib.MpcgError):
        q.solve_host(np.zeros(q.nnz, np.float32), np.ones(14 * 4, np.float32))
