"""GPU: state sizes other than the tuned n = 14 (SURVEY.md §8b "n=14 specialisation (+ generic fallback)"): the PCG entry
points run the generic streaming kernel (pcg_generic_kernel, float and double), same bd layout with n x n blocks, same
semantics; checked against the oracle, whose PCG restatement takes n as a parameter.  Everything else says UNSUPPORTED."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import relinf

pytestmark = pytest.mark.gpu


def random_system(n, N, seed, dtype):
    """Negated SPD block-tridiagonal S in the bd layout (column-major n x n blocks), symmetric-stair Pinv
    (include/pcg/linsys_setup.cuh:97-136: D^-1 - D^-1 O D^-1), gamma."""
    rng = np.random.default_rng(seed)
    D = [np.eye(n) * (3.0 + rng.random()) + 0.2 * (lambda a: a + a.T)(rng.standard_normal((n, n))) for _ in range(N)]
    O = [0.3 * rng.standard_normal((n, n)) for _ in range(N - 1)]            # block (k+1, k)
    S = np.full((N, 3, n * n), np.nan)
    P = np.full((N, 3, n * n), np.nan)
    Di = [np.linalg.inv(d) for d in D]
    for k in range(N):
        S[k, 1] = (-D[k]).T.reshape(-1)
        P[k, 1] = (-Di[k]).T.reshape(-1)
        if k > 0:
            S[k, 0] = (-O[k - 1]).T.reshape(-1)
            P[k, 0] = (Di[k] @ O[k - 1] @ Di[k - 1]).T.reshape(-1)           # -Pinv_kk S_k,k-1 Pinv_k-1,k-1 with the stored signs
        if k < N - 1:
            S[k, 2] = (-O[k].T).T.reshape(-1)
            P[k, 2] = (Di[k] @ O[k].T @ Di[k + 1]).T.reshape(-1)
    g = rng.standard_normal(n * N)
    return S.reshape(-1).astype(dtype), P.reshape(-1).astype(dtype), g.astype(dtype)


@pytest.mark.parametrize("n,N", [(6, 40), (20, 17), (1, 5)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_generic_state_size_vs_oracle(orc, n, N, pc):
    from mpcgpu_amd import PcgSolver, pcg_config, _lib
    B, K = 3, min(25, n * N - 2)        # (fixed-count runs must stop before CG has converged exactly: eta -> 0)
    sol = PcgSolver(N, max_batch=B, state_size=n)
    systems = [random_system(n, N, 100 * n + b, np.float32) for b in range(B)]
    S, P, g = (np.stack([s[i] for s in systems]) for i in range(3))
    dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == 3 and (it.cpu().numpy() == K).all()
    for b in range(B):
        Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
        r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), np.zeros(n * N), N, K, 0.0, pc, n=n)
        r32 = orc.pcg(Sz, Pz, g[b], np.zeros(n * N, np.float32), N, K, 0.0, pc, n=n)
        band = relinf(r32["lam"], r64["lam"])
        assert relinf(lam[b].cpu().numpy(), r64["lam"]) <= max(1e-4, 4 * band), (n, N, pc, b)
    # tolerance exit + double precision through the same kernel
    S64, P64, g64 = (np.stack([random_system(n, N, 100 * n + b, np.float64)[i] for b in range(B)]) for i in range(3))
    lam64 = torch.zeros(B, n * N, device="cuda", dtype=torch.float64)
    it, ex = sol.solve_f64(torch.from_numpy(S64).cuda(), torch.from_numpy(P64).cuda(), torch.from_numpy(g64).cuda(), lam64,
                           pcg_config(pcg_exit_tol=1e-8, pcg_max_iter=500), pc)
    torch.cuda.synchronize()
    assert (ex.cpu().numpy() == 0).all()
    for b in range(B):
        ref = orc.pcg(np.nan_to_num(S64[b]), np.nan_to_num(P64[b]), g64[b], np.zeros(n * N), N, 500, 1e-8, pc, n=n)
        assert abs(int(it[b]) - ref["iters"]) <= max(2, 0.03 * ref["iters"]) and relinf(lam64[b].cpu().numpy(), ref["lam"]) < 1e-5
    # the entry points that exist for n = 14 only say so
    with pytest.raises(_lib.MpcgError) as e:
        sol.bt_spmv(dS, dg)
    assert e.value.code == _lib.MPCG_ERR_UNSUPPORTED
    assert sol.checkPcgOccupancy() >= sol.get_option("num_cus")
    assert sol.lib.mpcg_pcg_lds_bytes(n, N) == 4 * (2 * (N + 2) * n + 2 * N * n + 16)
