import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(N):
    d = np.load(os.path.join(GOLDEN, f"pcg_n14_N{N}.npz"))
    return {k: d[k] for k in d.files}


def relinf(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rel_residual(S_bd, gamma, lam, N):
    """||gamma - S lam||_2 / ||gamma||_2 in float64 on the dense matrix."""
    from mpcgpu_amd import synth
    Sd = synth.bd_to_dense(np.nan_to_num(S_bd), N)
    g = np.asarray(gamma, np.float64)
    return np.linalg.norm(g - Sd @ np.asarray(lam, np.float64)) / np.linalg.norm(g)


def fp32_band(orc, S, Pinv, g, lam0, N, K, pc, ref64, trials=4):
    """Self-calibrating fp32 tolerance for a FIXED iteration count K: the largest distance from the
    float64 iterate reached by the CPU float32 restatement on (a) the same inputs and (b) inputs
    perturbed by one float32 ulp (relative 6e-8 gaussian) — i.e. what rounding-level noise does to
    fp32 CG on this system (cond ~1e5).  Any correct fp32 implementation with a different summation
    order lands inside a small multiple of this band."""
    S = np.nan_to_num(np.asarray(S, np.float32))
    P = np.nan_to_num(np.asarray(Pinv, np.float32))
    g = np.asarray(g, np.float32)
    lam0 = np.asarray(lam0, np.float32)
    band = relinf(orc.pcg(S, P, g, lam0, N, K, 0.0, pc)["lam"], ref64)
    rng = np.random.default_rng(K * 1000 + N)
    for _ in range(trials):
        Sp = (S.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(S.shape))).astype(np.float32)
        gp = (g.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(g.shape))).astype(np.float32)
        band = max(band, relinf(orc.pcg(Sp, P, gp, lam0, N, K, 0.0, pc)["lam"], ref64))
    return band


def fp32_iters_band(orc, S, Pinv, g, lam0, N, max_iter, tol, pc, trials=8):
    """(lo, hi) of the iteration count the CPU float32 restatement needs on the same and on
    1-ulp-perturbed inputs.  With a tolerance exit the crossing iteration of fp32 CG is erratic
    (measured on the N=32 golden system: 174..201 against 172 in float64), so a fixed +-10 % around
    one run would be a coin toss; a correct fp32 kernel must land within this band (+-7 %)."""
    S = np.nan_to_num(np.asarray(S, np.float32))
    P = np.nan_to_num(np.asarray(Pinv, np.float32))
    g = np.asarray(g, np.float32)
    lam0 = np.asarray(lam0, np.float32)
    its = [orc.pcg(S, P, g, lam0, N, max_iter, tol, pc)["iters"]]
    rng = np.random.default_rng(N + max_iter)
    for _ in range(trials):
        Sp = (S.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(S.shape))).astype(np.float32)
        gp = (g.astype(np.float64) * (1 + 6e-8 * rng.standard_normal(g.shape))).astype(np.float32)
        its.append(orc.pcg(Sp, P, gp, lam0, N, max_iter, tol, pc)["iters"])
    return min(its), max(its)


def exit_iter_bounds(orc, S, Pinv, g, lam0, N, max_iter, tol, pc, trials=8):
    """Acceptable range of the iteration at which an fp32 PCG leaves on |eta| < tol.  Lower end: 7 % below the
    earliest exit of the CPU float32 restatement (same / 1-ulp-perturbed inputs) and of the float64 restatement.
    Upper end: 7 % above the latest of those AND of the first DECISIVE crossing of the float64 eta history
    (|eta| < tol/4): near the threshold eta hovers (N=32 golden system, block-Jacobi: float64 dips to 0.55 tol at
    iteration 343 for three iterations, climbs back to 2 tol and only falls below tol/4 at 382), so whether an fp32
    implementation catches a marginal dip is a coin toss on its summation order — the same final residual either way."""
    lo, hi = fp32_iters_band(orc, S, Pinv, g, lam0, N, max_iter, tol, pc, trials)
    S64 = np.nan_to_num(np.asarray(S, np.float64))
    P64 = np.nan_to_num(np.asarray(Pinv, np.float64))
    h = np.abs(orc.pcg(S64, P64, np.asarray(g, np.float64), np.asarray(lam0, np.float64), N, max_iter, tol * 1e-3, pc, hist=True)["eta_hist"])
    first = int(np.argmax(h < tol)) if (h < tol).any() else max_iter
    decisive = int(np.argmax(h < tol / 4)) if (h < tol / 4).any() else max_iter
    return int(0.93 * min(lo, first) - 2), int(min(max_iter, 1.07 * max(hi, decisive) + 2))


def default_family(N, pc="ss", batch_per_cu=0.0):
    """"last_kernel_family" of a default fp32 solve (state_size 14; include/mpcg.h, options): 5 row-per-lane up to 32 knots (16 < N <= 32 from 2.5
    trajectories per CU: the lane-quad kernel's 32-knot build), 11 lane-quad to 128 knots (both preconditioners); 7 clustered
    lane-pair beyond.  (6, the lane-pair kernel: fp16 storage and "pcg_lqb" = 0.)"""
    if N <= 32:
        return 11 if N > 16 and batch_per_cu >= 2.5 else 5
    return 11 if N <= 128 else 7
