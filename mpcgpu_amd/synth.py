"""Synthetic IIWA-14-shaped KKT problems and their Schur systems (host side, numpy).

The reference obtains (G, C, g, c) from GRiD rigid-body dynamics of the KUKA IIWA-14
(`include/common/kkt.cuh:22-163`), which is out of scope.  For
tests and benchmarks we draw KKT blocks with the *structure* of that problem
(SURVEY.md §8d "Synthetic inputs") and turn them into the PCG inputs `(S, Pinv, gamma)`
with the math of `include/pcg/linsys_setup.cuh:139-562` (block rows) and `:9-137`
(symmetric-stair completion), vectorised over (trajectory, knot).

This module is an *input producer*: nothing here is on the solve path.  It is
independent of `oracle/` (which restates the same formulas in C one knot at a time and is
used by the tests to check this builder).

Layouts (all identical to the reference's device buffers):
  G_dense : [Q_0 | R_0 | Q_1 | R_1 | ... | Q_{N-1}]   column-major blocks  (`include/pcg/sqp.cuh:43`)
  C_dense : [-A_0 | -B_0 | ... | -A_{N-2} | -B_{N-2}] column-major, ALREADY negated (`include/common/kkt.cuh:115-116`)
  g       : [q_0 | r_0 | ... | q_{N-1}]
  c       : [c_0 | ... | c_{N-1}]
  S, Pinv : "bd" layout [N][3][n*n], block (k, col) column-major at k*3n^2 + col*n^2;
            col 0 = left off-diagonal, 1 = diagonal, 2 = right off-diagonal; stored NEGATED
            (`include/pcg/linsys_setup.cuh:15-19, 490-507`).  Blocks (0, col 0) and
            (N-1, col 2) are never written by the reference (`:97,118`): we fill them with
            NaN on request so that a kernel that reads them is caught.
  gamma   : [N][n], stored negated (`:272-276, 528-532`).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

STATE_SIZE = 14      # include/common/settings.cuh:10-12
CONTROL_SIZE = 7
TIMESTEP = 1.0 / 64  # examples/track_iiwa_pcg.cu:19
RHO_INIT = 1e-3      # include/mpcsim.cuh:219
QD_COST = 1e-4       # include/common/settings.cuh:92-94
R_COST = 1e-4        # include/common/settings.cuh:84-90

# PCG iteration caps "found using experiments", include/common/settings.cuh:123-139
PCG_MAX_ITER = {32: 173, 64: 167, 128: 167, 256: 118, 512: 67}


def pcg_max_iter(knot_points: int) -> int:
    return PCG_MAX_ITER.get(int(knot_points), 200)


@dataclass
class KKT:
    """Per-trajectory KKT blocks, batched: leading dim = trajectory."""
    Q: np.ndarray   # [B, N, n, n]
    R: np.ndarray   # [B, N-1, m, m]
    A: np.ndarray   # [B, N-1, n, n]   (x_{k+1} = A_k x_k + B_k u_k + ...)
    Bm: np.ndarray  # [B, N-1, n, m]
    q: np.ndarray   # [B, N, n]
    r: np.ndarray   # [B, N-1, m]
    c: np.ndarray   # [B, N, n]

    @property
    def batch(self):
        return self.Q.shape[0]

    @property
    def knot_points(self):
        return self.Q.shape[1]


def make_kkt(knot_points: int, batch: int, seed: int, *, n: int = STATE_SIZE, m: int = CONTROL_SIZE,
             dt: float = TIMESTEP, stiffness: float = 3.0) -> KKT:
    """Draw `batch` independent trajectories' KKT blocks (float64).

    Structure follows the IIWA tracking problem: Q = blkdiag(q_pos q_pos^T, QD_COST*I) with
    q_pos the position-error gradient (`include/dynamics/iiwa/iiwa_eepos_plant.cuh:329-368`),
    R = R_COST*I, A = I + dt*[[0, I], [dqdd/dq, dqdd/dqd]] and B = dt*[0; Minv]
    (`include/common/integrator.cuh:67-79`).  Trajectory b depends only on (seed, b).
    """
    N = int(knot_points)
    h = n // 2
    assert n == 2 * h and m == h
    Q = np.zeros((batch, N, n, n))
    R = np.zeros((batch, N - 1, m, m))
    A = np.zeros((batch, N - 1, n, n))
    Bm = np.zeros((batch, N - 1, n, m))
    q = np.zeros((batch, N, n))
    r = np.zeros((batch, N - 1, m))
    c = np.zeros((batch, N, n))
    eye_h = np.eye(h)
    for b in range(batch):
        rng = np.random.default_rng([int(seed), b])
        gq = rng.normal(0.0, 0.3, size=(N, h))            # q_pos = J^T err; Hessian = q_pos q_pos^T (rank 1)
        Q[b, :, :h, :h] = np.einsum("ki,kj->kij", gq, gq)
        Q[b, :, h:, h:] = QD_COST * eye_h
        R[b] = R_COST * np.eye(m)
        dq = rng.normal(0.0, stiffness, size=(N - 1, h, h))
        dqd = rng.normal(0.0, 0.3 * stiffness, size=(N - 1, h, h))
        A[b, :, :h, :h] = eye_h
        A[b, :, :h, h:] = dt * eye_h
        A[b, :, h:, :h] = dt * dq
        A[b, :, h:, h:] = eye_h + dt * dqd
        W = rng.normal(0.0, 1.0, size=(N - 1, h, h))
        Minv = np.einsum("kia,kja->kij", W, W) / h + eye_h
        Bm[b, :, h:, :] = dt * Minv
        q[b, :, :h] = gq
        q[b, :, h:] = QD_COST * rng.normal(0.0, 1.0, size=(N, h))      # QD_cost * qd
        r[b] = R_COST * rng.normal(0.0, 5.0, size=(N - 1, m))         # R_cost * u
        c[b, 1:] = rng.normal(0.0, 1e-2, size=(N - 1, n))   # c_0 = x_0 - x_s = 0 (kkt.cuh:106-108)
    return KKT(Q, R, A, Bm, q, r, c)


def pack_kkt_dense(k: KKT, dtype=np.float32):
    """Flatten to the reference's (G_dense, C_dense, g, c) device layouts; C is stored negated."""
    B, N, n, _ = k.Q.shape
    m = k.R.shape[-1]
    G = np.zeros((B, (n * n + m * m) * N - m * m), dtype)
    C = np.zeros((B, (n * n + n * m) * (N - 1)), dtype)
    g = np.zeros((B, (n + m) * N - m), dtype)
    gs, cs, vs = n * n + m * m, n * n + n * m, n + m
    for kk in range(N):
        G[:, kk * gs: kk * gs + n * n] = k.Q[:, kk].transpose(0, 2, 1).reshape(B, -1)
        g[:, kk * vs: kk * vs + n] = k.q[:, kk]
        if kk < N - 1:
            G[:, kk * gs + n * n: (kk + 1) * gs] = k.R[:, kk].transpose(0, 2, 1).reshape(B, -1)
            C[:, kk * cs: kk * cs + n * n] = -k.A[:, kk].transpose(0, 2, 1).reshape(B, -1)
            C[:, kk * cs + n * n: (kk + 1) * cs] = -k.Bm[:, kk].transpose(0, 2, 1).reshape(B, -1)
            g[:, kk * vs + n: (kk + 1) * vs] = k.r[:, kk]
    c = k.c.reshape(B, -1).astype(dtype)
    return G, C, g, c


def _bd(blocks):
    """[..., i, j] matrix blocks -> column-major storage order [..., j, i]."""
    return np.swapaxes(blocks, -1, -2)


def form_schur(k: KKT, rho: float = RHO_INIT, precond: str = "ss", dtype=np.float32,
               poison_unused: bool = False):
    """(Q,R,A,B,q,r,c,rho) -> (S, Pinv, gamma) in bd layout, as the reference's
    `form_schur_system` (`include/pcg/linsys_setup.cuh:620-656`) leaves them on the device.

    Math in float64, result cast to `dtype`.  `precond` in {"jacobi", "ss"}:
    "jacobi" keeps only Pinv[k,1] (the off-diagonal blocks are written as zeros).
    """
    B, N, n, _ = k.Q.shape
    m = k.R.shape[-1]
    In, Im = np.eye(n), np.eye(m)
    Qi = np.linalg.inv(k.Q + rho * In)                 # [B,N,n,n]   (:180-181, 329-331, 214-217, 356-368)
    Ri = np.linalg.inv(k.R + rho * Im)                 # [B,N-1,m,m]
    Ab, Bb = -k.A, -k.Bm                               # as stored in C_dense
    phi = Ab @ Qi[:, :-1]                              # phi_k = Abar Qi_{k-1}           (:397-398)
    BR = Bb @ Ri                                       # Bbar Ri                          (:405-406)
    theta = phi @ _bd(Ab) + Qi[:, 1:] + BR @ _bd(Bb)   # (:446-488)
    gam = (np.einsum("bkij,bkj->bki", Qi[:, 1:], k.q[:, 1:]) - k.c[:, 1:]
           + np.einsum("bkij,bkj->bki", phi, k.q[:, :-1])
           + np.einsum("bkij,bkj->bki", BR, k.r))      # (:410-444)

    S = np.zeros((B, N, 3, n, n))
    P = np.zeros((B, N, 3, n, n))
    gamma = np.zeros((B, N, n))
    S[:, 0, 1] = _bd(-Qi[:, 0])                        # (:248-255)
    P[:, 0, 1] = _bd(-(k.Q[:, 0] + rho * In))          # (:201-210)
    gamma[:, 0] = -np.einsum("bij,bj->bi", Qi[:, 0], k.q[:, 0])   # (:259-275)
    S[:, 1:, 0] = _bd(-phi)                            # (:490-497)
    S[:, 1:, 1] = _bd(-theta)                          # (:500-507)
    S[:, :-1, 2] = -phi                                # phi^T, column-major == phi row-major (:536-557)
    thetaInv = np.linalg.inv(theta)
    P[:, 1:, 1] = _bd(-thetaInv)                       # (:510-524)
    gamma[:, 1:] = -gam                                # (:528-532)

    if precond == "ss":                                # complete_SS_Pinv_blockrow (:9-137)
        Pd = _bd(P[:, :, 1])                           # Pinv[k,1] as matrices
        Sl = _bd(S[:, 1:, 0])                          # S[k,0], k>=1
        left = -(Pd[:, 1:] @ Sl @ Pd[:, :-1])          # (:97-115)
        right = -(Pd[:, :-1] @ _bd(Sl) @ Pd[:, 1:])    # (:118-136)  S[k+1,0]^T
        P[:, 1:, 0] = _bd(left)
        P[:, :-1, 2] = _bd(right)
    elif precond != "jacobi":
        raise ValueError("precond must be 'jacobi' or 'ss'")

    if poison_unused:
        for M in (S, P):
            M[:, 0, 0] = np.nan
            M[:, -1, 2] = np.nan
    return (S.reshape(B, N * 3 * n * n).astype(dtype),
            P.reshape(B, N * 3 * n * n).astype(dtype),
            gamma.reshape(B, N * n).astype(dtype))


def bd_to_dense(S_bd: np.ndarray, N: int, n: int = STATE_SIZE) -> np.ndarray:
    """One trajectory's bd-layout matrix -> dense (nN x nN) float64 (tests/diagnostics)."""
    S = np.asarray(S_bd, dtype=np.float64).reshape(N, 3, n, n)
    D = np.zeros((N * n, N * n))
    for kk in range(N):
        for col in range(3):
            kc = kk + col - 1
            if 0 <= kc < N:
                D[kk * n:(kk + 1) * n, kc * n:(kc + 1) * n] = S[kk, col].T
    return D


def algorithmic_bytes(N: int, n: int = STATE_SIZE, precond: str = "ss", elem: int = 4) -> dict:
    """SURVEY.md §8(d) / BASELINE.md §2 byte model per trajectory."""
    spmv = elem * ((3 * N - 2) * n * n + 2 * n * N)
    if precond == "ss":
        it = elem * (2 * (3 * N - 2) * n * n + 6 * n * N)
    else:
        it = elem * ((3 * N - 2) * n * n + N * n * n + 6 * n * N)
    return {"spmv": spmv, "pcg_iter": it}
