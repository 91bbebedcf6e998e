"""GPU: the PCG recurrences checked on the kernel's OWN outputs, in float64 on the host — no oracle, no same-author restatement on the
comparison side (VERDICT r04 #2b).  The 12-argument entry (include/pcg/sqp.cuh:137-150) returns lambda, d_r and d_p of the last completed
update; the kernels are deterministic, so a run capped at K - 1 iterations IS the state the K-iteration run passed through.  From the two
states, with S / Pinv / gamma promoted to float64 and nothing but textbook preconditioned CG (what pcg<> computes):

  step      alpha = (r' Pinv r) / (p' S p) from state K-1;   lambda_K = lambda_{K-1} + alpha p_{K-1};   r_K = r_{K-1} - alpha S p_{K-1};
            beta = (r_K' Pinv r_K) / (r_{K-1}' Pinv r_{K-1});   p_K = Pinv r_K + beta p_{K-1}
            — ONE iteration of float32 arithmetic against its float64 evaluation: rounding has not been amplified by CG, so the tolerance is
            a few 1e-5 of each update's size (a 1e-4 arithmetic slip fails; the K-iterate comparisons of test_gpu_parity.py, with their
            max(1e-3, 4 x band), would pass it);
  gap       || d_r - (gamma - S lambda_K) || <= 16 K eps32 ||S||_inf max(||lambda||) (the classical bound on the drift of the updated residual);
  conjugacy | p_K' S p_{K-1} | / sqrt((p_K' S p_K)(p_{K-1}' S p_{K-1})) small (local conjugacy survives finite precision);
  orthogonality  | r_K' Pinv r_{K-1} | / sqrt(eta_K eta_{K-1}) small.

K = 10, 50, 167 for N = 32 (row-per-lane kernel), 128 (lane-pair kernel), 512 (clustered kernel); symmetric-stair and block-Jacobi Pinv."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth

pytestmark = pytest.mark.gpu
n = 14
EPS32 = 2.0 ** -24


def bt_matvec(M, x, N):
    """Block-tridiagonal product from the bd layout in float64 (numpy): y_k = M[k,0] x_{k-1} + M[k,1] x_k + M[k,2] x_{k+1}, blocks column-major."""
    B = np.nan_to_num(np.asarray(M, np.float64)).reshape(N, 3, n, n).transpose(0, 1, 3, 2)       # [k][col] as row-major matrices
    X = np.asarray(x, np.float64).reshape(N, n)
    y = np.einsum("kij,kj->ki", B[:, 1], X)
    y[1:] += np.einsum("kij,kj->ki", B[1:, 0], X[:-1])
    y[:-1] += np.einsum("kij,kj->ki", B[:-1, 2], X[1:])
    return y.reshape(-1)


def state(sol, dS, dP, dg, lam0, K):
    N = sol.N
    d_lam = torch.from_numpy(lam0.copy()).cuda()
    d_r = torch.full((n * N,), float("nan"), device="cuda")
    d_p = torch.full((n * N,), float("nan"), device="cuda")
    scratch = torch.zeros(N, device="cuda")
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_ex = torch.zeros(1, dtype=torch.bool, device="cuda")
    sol.solve_ref(dS, dP, dg, d_lam, d_r, d_p, scratch, scratch, d_it, d_ex, K, 0.0)
    torch.cuda.synchronize()
    assert int(d_it.item()) == K and bool(d_ex.item()) is True
    out = [t.cpu().numpy().astype(np.float64) for t in (d_lam, d_r, d_p)]
    assert all(np.isfinite(v).all() for v in out)
    return out


@pytest.mark.parametrize("N,family", [(32, 5), (128, 6), (512, 7)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_one_step_of_the_recurrences_and_the_cg_invariants(N, family, pc):
    from mpcgpu_amd import PcgSolver
    k = synth.make_kkt(N, 1, 9000 + N)
    S, P, g = synth.form_schur(k, precond=pc)                # (block-Jacobi: the off-diagonal blocks of Pinv are zeros; the entry applies three columns)
    S, P, g = S[0], P[0], g[0]
    rng = np.random.default_rng(N)
    lam0 = (0.1 * rng.standard_normal(n * N)).astype(np.float32)
    sol = PcgSolver(N, max_batch=1)
    dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
    g64 = g.astype(np.float64)
    Snorm = np.abs(np.nan_to_num(S).astype(np.float64).reshape(N, 3, n, n)).sum(axis=(1, 2)).max()      # ||S||_inf (rows of a block row: sum over its three blocks' columns)
    report = []
    for K in (10, 50, 167):
        lam_a, r_a, p_a = state(sol, dS, dP, dg, lam0, K - 1)
        assert sol.get_option("last_kernel_family") == family
        lam_b, r_b, p_b = state(sol, dS, dP, dg, lam0, K)
        Sp = bt_matvec(S, p_a, N)
        z_a, z_b = bt_matvec(P, r_a, N), bt_matvec(P, r_b, N)
        eta_a, eta_b, v = r_a @ z_a, r_b @ z_b, p_a @ Sp
        alpha, beta = eta_a / v, eta_b / eta_a
        assert eta_a < 0 and eta_b < 0 and v < 0             # S and Pinv are stored negated (include/pcg/linsys_setup.cuh:15-19): eta = r' Pinv r < 0
        # ---- one step, float32 on the device against float64 here ----
        inf = lambda x: np.abs(x).max()
        e_lam = inf(lam_b - (lam_a + alpha * p_a)) / (abs(alpha) * inf(p_a))
        e_r = inf(r_b - (r_a - alpha * Sp)) / max(abs(alpha) * inf(Sp), inf(r_a))
        e_p = inf(p_b - (z_b + beta * p_a)) / max(inf(z_b), abs(beta) * inf(p_a))
        # lambda is rounded to float32 at every update: its own half-ulp is part of e_lam when the update is small against lambda itself
        lam_ulp = EPS32 * inf(lam_b) / (abs(alpha) * inf(p_a))
        assert e_lam <= 3e-5 + 2 * lam_ulp, (N, pc, K, "lambda", e_lam, lam_ulp)
        assert e_r <= 3e-5, (N, pc, K, "r", e_r)
        assert e_p <= 3e-5, (N, pc, K, "p", e_p)
        # ---- drift of the updated residual from the true one ----
        true_r = g64 - bt_matvec(S, lam_b, N)
        gap = np.linalg.norm(r_b - true_r)
        bound = 16 * K * EPS32 * Snorm * max(np.linalg.norm(lam_b), np.linalg.norm(lam0))
        assert gap <= bound, (N, pc, K, "gap", gap, bound)
        # ---- local conjugacy / orthogonality ----
        conj = abs(p_b @ Sp) / np.sqrt((p_b @ bt_matvec(S, p_b, N)) * v)
        orth = abs(r_b @ z_a) / np.sqrt(eta_a * eta_b)
        assert conj <= 2e-3, (N, pc, K, "conjugacy", conj)
        assert orth <= 2e-3, (N, pc, K, "orthogonality", orth)
        report.append((K, e_lam, e_r, e_p, gap / np.linalg.norm(g64), bound / np.linalg.norm(g64), conj, orth))
    for row in report:
        print("N=%d %s K=%d: step lambda %.1e r %.1e p %.1e | gap/|gamma| %.1e (bound %.1e) | conjugacy %.1e orthogonality %.1e" % ((N, pc) + row))
