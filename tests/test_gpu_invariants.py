"""GPU: the PCG recurrences checked on the kernel's OWN outputs, in float64 on the host — no oracle, no same-author restatement on the
comparison side (VERDICT r04 #2b).  The 12-argument entry (include/pcg/sqp.cuh:137-150) returns lambda, d_r and d_p of the last completed
update; the kernels are deterministic, so a run capped at K - 1 iterations IS the state the K-iteration run passed through.  From the two
states, with S / Pinv / gamma promoted to float64 and nothing but textbook preconditioned CG (what pcg<> computes):

  step      lambda_K = lambda_{K-1} + alpha p_{K-1};   r_K = r_{K-1} - alpha S p_{K-1};   p_K = Pinv r_K + beta p_{K-1}, with the alpha / beta the kernel
            itself used (recovered from its updates), each within a few float32 ulps (u = 2^-24) of the SIZE OF ITS TERMS (|S||p|, |Pinv||r|:
            cancellation inside the three-block sums is accounted for, so the limits are O(1) x u whatever N, K or the conditioning);
  scalars   that alpha = (r' Pinv r) / (p' S p) and beta = (r_K' Pinv r_K) / (r_{K-1}' Pinv r_{K-1}) of the float64 evaluation, within a
            few u x the cancellation of the inner products
            — ONE iteration of float32 arithmetic against its float64 evaluation: rounding has not been amplified by CG, so an error of a few
            ulps in one update fails (the K-iterate comparisons of test_gpu_parity.py, with max(1e-3, 4 x band), would pass a 1e-4 slip);
  gap       || d_r - (gamma - S lambda_K) || <= 2 K u ||S||_inf max ||lambda|| (the classical bound on the drift of the updated residual);
  conjugacy | p_K' S p_{K-1} | / sqrt((p_K' S p_K)(p_{K-1}' S p_{K-1})) <= 2e-3 (local conjugacy survives finite precision);
  orthogonality  | r_K' Pinv r_{K-1} | / sqrt(eta_K eta_{K-1}) <= 2e-3.

K = 10, 50, 167 for N = 32 (row-per-lane kernel), 128 (lane-pair kernel), 512 (clustered kernel); symmetric-stair and block-Jacobi Pinv."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth

pytestmark = pytest.mark.gpu
n = 14
EPS32 = 2.0 ** -24


HP = [np.float64]          # the host's evaluation precision: float64 for the float kernels, the x87 80-bit long double for the double kernels


def bt_matvec(M, x, N):
    """Block-tridiagonal product from the bd layout in HP[0] precision (numpy): y_k = M[k,0] x_{k-1} + M[k,1] x_k + M[k,2] x_{k+1}, blocks column-major."""
    B = np.nan_to_num(np.asarray(M, HP[0])).reshape(N, 3, n, n).transpose(0, 1, 3, 2)       # [k][col] as row-major matrices
    X = np.asarray(x, HP[0]).reshape(N, n)
    y = np.einsum("kij,kj->ki", B[:, 1], X)
    y[1:] += np.einsum("kij,kj->ki", B[1:, 0], X[:-1])
    y[:-1] += np.einsum("kij,kj->ki", B[:-1, 2], X[1:])
    return y.reshape(-1)


def state(sol, dS, dP, dg, lam0, K):
    N = sol.N
    dt = dS.dtype                                          # float32, or float64 = linsys_t double (the same checks at u = 2^-53)
    d_lam = torch.from_numpy(lam0.copy()).cuda()
    d_r = torch.full((n * N,), float("nan"), device="cuda", dtype=dt)
    d_p = torch.full((n * N,), float("nan"), device="cuda", dtype=dt)
    scratch = torch.zeros(N, device="cuda", dtype=dt)
    d_it = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_ex = torch.zeros(1, dtype=torch.bool, device="cuda")
    (sol.solve_ref if dt == torch.float32 else sol.solve_ref_f64)(dS, dP, dg, d_lam, d_r, d_p, scratch, scratch, d_it, d_ex, K, 0.0)
    torch.cuda.synchronize()
    assert int(d_it.item()) == K and bool(d_ex.item()) is True
    out = [t.cpu().numpy().astype(HP[0]) for t in (d_lam, d_r, d_p)]
    assert all(np.isfinite(v).all() for v in out)
    return out


def abs_matvec(M, x, N):
    """|M| |x|: the size of the terms a float32 evaluation of M x rounds (cancellation inside the three-block sums shows here, not in |M x|)."""
    return bt_matvec(np.abs(np.nan_to_num(M)), np.abs(x), N)


def step_quantities(S, P, g, lam0, st_a, st_b, N, K, EPS32=EPS32):
    """Every checked quantity DIVIDED BY the scale its float32 rounding model gives it (u = 2^-24): a correct kernel lands at O(1..30)
    whatever N, K, the preconditioner or the conditioning; tools/_prof/inv_probe.py prints them for 96 (system, K) pairs."""
    (lam_a, r_a, p_a), (lam_b, r_b, p_b) = st_a, st_b
    inf = lambda x: np.abs(x).max()
    Sp, z_a, z_b = bt_matvec(S, p_a, N), bt_matvec(P, r_a, N), bt_matvec(P, r_b, N)
    eta_a, eta_b, v = r_a @ z_a, r_b @ z_b, p_a @ Sp
    assert eta_a < 0 and eta_b < 0 and v < 0             # S and Pinv are stored negated (include/pcg/linsys_setup.cuh:15-19): eta = r' Pinv r < 0
    alpha, beta = eta_a / v, eta_b / eta_a
    # what the kernel itself used, recovered from its own updates (least squares along p_{K-1})
    alpha_hat = ((lam_b - lam_a) @ p_a) / (p_a @ p_a)
    beta_hat = ((p_b - z_b) @ p_a) / (p_a @ p_a)
    # cancellation of the three inner products: sum |terms| / |sum|
    c_eta_a = (np.abs(r_a) @ abs_matvec(P, r_a, N)) / abs(eta_a)
    c_eta_b = (np.abs(r_b) @ abs_matvec(P, r_b, N)) / abs(eta_b)
    c_v = (np.abs(p_a) @ abs_matvec(S, p_a, N)) / abs(v)
    q = {
        # the update directions, with the kernel's own scalars: pure axpy / matvec rounding
        "lam": inf(lam_b - (lam_a + alpha_hat * p_a)) / (EPS32 * (inf(lam_b) + abs(alpha_hat) * inf(p_a))),
        "r": inf(r_b - (r_a - alpha_hat * Sp)) / (EPS32 * (inf(r_a) + abs(alpha_hat) * inf(abs_matvec(S, p_a, N)))),
        "p": inf(p_b - (z_b + beta_hat * p_a)) / (EPS32 * (inf(abs_matvec(P, r_b, N)) + abs(beta_hat) * inf(p_a))),
        # the scalars: float32 inner products of ~14 N terms against their float64 values, scaled by the cancellation of the sums
        "alpha": abs(alpha_hat / alpha - 1) / (EPS32 * (c_eta_a + c_v)),
        "beta": abs(beta_hat / beta - 1) / (EPS32 * (c_eta_a + c_eta_b)),
    }
    Snorm = float(np.abs(np.nan_to_num(S).astype(np.float64).reshape(N, 3, n, n)).sum(axis=(1, 2)).max())      # ||S||_inf (a block row's three blocks, summed along rows)
    true_r = g.astype(HP[0]) - bt_matvec(S, lam_b, N)
    nrm = lambda x: float(np.sqrt(float(x @ x)))
    q["gap"] = nrm(r_b - true_r) / (K * EPS32 * Snorm * max(nrm(lam_b), nrm(np.asarray(lam0, HP[0]))))
    q["conjugacy"] = abs(p_b @ Sp) / np.sqrt((p_b @ bt_matvec(S, p_b, N)) * v)
    q["orthogonality"] = abs(r_b @ z_a) / np.sqrt(eta_a * eta_b)
    return {k_: float(v_) for k_, v_ in q.items()}


# Worst over 96 (system, K) pairs on an MI355X (tools/_prof/inv_probe.py -> profiles/r05_invariants.txt): lam 0.89 u, r 1.67 u, p 0.95 u of their update's
# terms; alpha 0.57, beta 1.26 u x the cancellation of their inner products; gap 0.30 of K u ||S|| ||lambda||; conjugacy 1.7e-4, orthogonality 3.4e-4.
# The limits leave a factor 4-6: an error of a few float32 ulps in ONE update of ONE iteration fails.
LIMITS = {"lam": 4.0, "r": 8.0, "p": 8.0, "alpha": 8.0, "beta": 8.0, "gap": 2.0, "conjugacy": 2e-3, "orthogonality": 2e-3}


@pytest.mark.parametrize("N,family", [(32, 5), (64, 11), (128, 11), (128, 6), (512, 7)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_one_step_of_the_recurrences_and_the_cg_invariants(N, family, pc):
    from mpcgpu_amd import PcgSolver
    k = synth.make_kkt(N, 1, 9000 + N)
    S, P, g = synth.form_schur(k, precond=pc)                # (block-Jacobi: the off-diagonal blocks of Pinv are zeros; the entry applies three columns)
    S, P, g = S[0], P[0], g[0]
    rng = np.random.default_rng(N)
    lam0 = (0.1 * rng.standard_normal(n * N)).astype(np.float32)
    sol = PcgSolver(N, max_batch=1)
    if family in (6, 11):                                      # (the lane-pair / the lane-quad kernel, pinned: the policy picks by horizon and preconditioner)
        sol.set_option("pcg_lpk", 1); sol.set_option("pcg_lqb", int(family == 11))
    dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
    for K in (10, 50, 167):
        st_a = state(sol, dS, dP, dg, lam0, K - 1)
        assert sol.get_option("last_kernel_family") == family
        st_b = state(sol, dS, dP, dg, lam0, K)
        q = step_quantities(S, P, g, lam0, st_a, st_b, N, K)
        print(f"N={N} {pc} K={K}: " + " ".join(f"{a} {b:.2g}" for a, b in q.items()))
        for key, lim in LIMITS.items():
            assert q[key] <= lim, (N, pc, K, key, q[key], lim)


@pytest.mark.parametrize("N,family", [(32, 5), (64, 9), (64, 8), (128, 10), (128, 8)])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
def test_the_same_invariants_in_double(N, family, pc):
    """linsys_t = double: the row-per-lane kernel (N = 32), the lane-quad kernel (N = 64) and the clustered row-per-lane kernel across 2 / 4 CUs
    (N = 64 / 128) against the same
    recurrences at u = 2^-53, the host side evaluated in the x87 80-bit long double (u = 2^-64; numpy's longdouble on x86-64) so that it is
    again the more precise side; same limits as the float test."""
    from mpcgpu_amd import PcgSolver
    k = synth.make_kkt(N, 1, 9100 + N)
    S, P, g = synth.form_schur(k, precond=pc, dtype=np.float64)
    S, P, g = S[0], P[0], g[0]
    rng = np.random.default_rng(N)
    lam0 = 0.1 * rng.standard_normal(n * N)
    sol = PcgSolver(N, max_batch=1)
    if family == 8:
        sol.set_option("pcg_lqk", 0)                      # (the clustered row-per-lane kernel instead of the lane-quad kernel)
    dS, dP, dg = (torch.from_numpy(a).cuda() for a in (S, P, g))
    U64 = 2.0 ** -53
    if np.finfo(np.longdouble).eps > 2.0 ** -60:
        pytest.skip("no extended-precision long double on this host")
    HP[0] = np.longdouble
    try:
        _double_invariants(sol, dS, dP, dg, S, P, g, lam0, N, family, pc, U64)
    finally:
        HP[0] = np.float64


def _double_invariants(sol, dS, dP, dg, S, P, g, lam0, N, family, pc, U64):
    for K in (10, 50, 167):
        st_a = state(sol, dS, dP, dg, lam0, K - 1)
        assert sol.get_option("last_kernel_family") == family and sol.get_option("cluster_fixups") == 0
        st_b = state(sol, dS, dP, dg, lam0, K)
        q = step_quantities(S, P, g, lam0, st_a, st_b, N, K, EPS32=U64)
        print(f"double N={N} {pc} K={K}: " + " ".join(f"{a} {b:.2g}" for a, b in q.items()))
        for key, lim in LIMITS.items():
            scale = 1.0 if key in ("lam", "r", "p", "alpha", "beta", "gap") else 2.0 ** -29      # (conjugacy / orthogonality scale with u: float32's limit x 2^-29)
            assert q[key] <= lim * scale, (N, pc, K, key, q[key], lim * scale)
