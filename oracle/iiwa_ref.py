"""CPU restatement (numpy, float64) of the PRODUCER of the hot path's inputs on the real robot — TEST INFRASTRUCTURE ONLY (oracle/).

IIWA-14 tracking problem: rigid-body dynamics, end-effector kinematics and the KKT block assembly of MPCGPU's SQP iteration
 — the checker of the HIP twin mpcg_generate_kkt (mpcgpu_amd/csrc/kkt_plant.hip.h).  Only tests/, bench.py's
checking legs and tests/make_iiwa_golden.py import this module; nothing under mpcgpu_amd/ does (tests/test_abi.py greps for it).

Restates:
  include/common/kkt.cuh:22-163          generate_kkt_submatrices: per knot k < N-1  A_k, B_k, integrator defect c_{k+1},
                                          cost Hessian / gradient Q_k, R_k, q_k, r_k (+ Q_{N-1}, q_{N-1} in the last block);
                                          C stores -A, -B (:115-116, 158-159); c_0 = x_0 - x_s (:106-108)
  include/common/integrator.cuh:56-104    Euler (INTEGRATOR_TYPE 0, the default the call site uses, include/pcg/sqp.cuh:190):
                                          A = I + dt [[0, I], [dqdd/dq, dqdd/dqd]],  B = dt [0; dqdd/du]
  include/dynamics/iiwa/iiwa_eepos_plant.cuh:127-155   forwardDynamicsAndGradient: qdd = Minv (u - c(q, qd)) with GRAVITY = 0 (:53),
                                          dqdd/d(q,qd) = -Minv d(ID)/d(q,qd) at that qdd, dqdd/du = Minv
  include/dynamics/iiwa/iiwa_eepos_plant.cuh:307-390   trackingCostGradientAndHessian: g = J_ee^T (ee(q) - goal), q_k = [g; QD qd],
                                          Q_k = blkdiag(g g^T, QD I) (rank-one Gauss-Newton block), R_k = R_COST I, r_k = R_COST u
  :392-411                                _lastblock: Q_{N-1}, q_{N-1} are evaluated AT x_{N-2} against goal N-1 (the reference passes
                                          s_xux, not the next state; restated as is)
The rigid-body algorithms are the textbook ones GRiD generates code for (RNEA; mass matrix column by column); the robot
itself is DATA: mpcgpu_amd/data/iiwa14_model.json = the spatial transforms, spatial inertias and homogeneous transforms of
the KUKA LBR iiwa 14 as tabulated in include/dynamics/iiwa/iiwa_eepos_grid.cuh (extracted by tests/make_iiwa_golden.py).
Derivatives of the inverse dynamics are central differences in float64 (|error| ~1e-9), not GRiD's analytic recursion — an independent
check of the analytic recursion the HIP kernel runs.

PINNED ON REFERENCE-HELD DATA (round 4): the reference's precomputed trajectory examples/trajfiles/0_0_traj.csv was produced by its own
dynamics — consecutive rows satisfy  q[t+1] = q[t] + dt qd[t],  qd[t+1] = qd[t] + dt FD(q, qd, u)[t]  with dt = 1/64 — and the paired
0_0_eepos.traj by its own kinematics.  `euler_defect` of this module reproduces the 656 in-segment transitions of the 666-row file to
3e-7 (csv print precision) and all 666 end-effector rows to 3e-6: tests/test_iiwa_plant.py.  (The PCG itself stays unpinned: mpcg_oracle.c.)
"""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_PATH = os.path.join(os.path.dirname(_HERE), "mpcgpu_amd", "data", "iiwa14_model.json")
NJ = 7
QD_COST = 1e-4          # include/common/settings.cuh:92-94
TIMESTEP = 1.0 / 64     # examples/track_iiwa_pcg.cu:19


def r_cost(knot_points: int) -> float:
    """include/common/settings.cuh:84-90"""
    return 1e-3 if knot_points == 64 else 1e-4

class Model:
    def __init__(self, path: str = MODEL_PATH):
        d = json.load(open(path))
        self.X_const = np.array(d["X_const"], np.float64)          # 7 x 36, column-major 6x6
        self.X_trig = [(int(i), float(c), int(j)) for i, c, j in d["X_trig"]]
        self.I = np.array(d["I"], np.float64).reshape(NJ, 6, 6).transpose(0, 2, 1)
        self.Xhom_const = np.array(d["Xhom_const"], np.float64)    # 7 x 16, column-major 4x4
        self.Xhom_trig = [(int(i), float(c), int(j)) for i, c, j in d["Xhom_trig"]]

    def X(self, q):
        """Spatial transforms parent -> link k (load_update_XImats_helpers)."""
        t = np.concatenate([np.sin(q), np.cos(q)])
        x = self.X_const.copy()
        for i, c, j in self.X_trig:
            x[i] = c * t[j]
        X = x.reshape(NJ, 6, 6).transpose(0, 2, 1).copy()
        X[:, 3:, 3:] = X[:, :3, :3]                                 # rotation block repeated bottom right
        return X

    def Xhom(self, q):
        t = np.concatenate([np.sin(q), np.cos(q)])
        x = self.Xhom_const.copy()
        for i, c, j in self.Xhom_trig:
            x[i] = c * t[j]
        return x.reshape(NJ, 4, 4).transpose(0, 2, 1)

    # ---- kinematics ----
    def ee_pos(self, q):
        """Position of the origin of link 7 in the base frame (end_effector_positions_inner: translation of Xhom_0 ... Xhom_6)."""
        T = np.eye(4)
        for Xh in self.Xhom(q):
            T = T @ Xh
        return T[:3, 3].copy()

    def ee_jac(self, q, h=1e-6):
        J = np.zeros((3, NJ))
        for j in range(NJ):
            e = np.zeros(NJ)
            e[j] = h
            J[:, j] = (self.ee_pos(q + e) - self.ee_pos(q - e)) / (2 * h)
        return J

    # ---- dynamics ----
    @staticmethod
    def _crm(v):
        w, u = v[:3], v[3:]
        sk = lambda a: np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        M = np.zeros((6, 6))
        M[:3, :3] = sk(w)
        M[3:, :3] = sk(u)
        M[3:, 3:] = sk(w)
        return M

    def rnea(self, q, qd, qdd, X=None):
        """Inverse dynamics tau = ID(q, qd, qdd), no gravity (gato_plant::GRAVITY = 0)."""
        X = self.X(q) if X is None else X
        S = np.zeros(6)
        S[2] = 1.0
        v = np.zeros((NJ, 6))
        a = np.zeros((NJ, 6))
        f = np.zeros((NJ, 6))
        vp, ap = np.zeros(6), np.zeros(6)
        for k in range(NJ):
            v[k] = X[k] @ vp + S * qd[k]
            a[k] = X[k] @ ap + S * qdd[k] + self._crm(v[k]) @ (S * qd[k])
            f[k] = self.I[k] @ a[k] - self._crm(v[k]).T @ (self.I[k] @ v[k])
            vp, ap = v[k], a[k]
        tau = np.zeros(NJ)
        for k in range(NJ - 1, -1, -1):
            tau[k] = f[k][2]
            if k > 0:
                f[k - 1] += X[k].T @ f[k]
        return tau

    def mass_matrix(self, q):
        X = self.X(q)
        z = np.zeros(NJ)
        M = np.zeros((NJ, NJ))
        for j in range(NJ):
            e = np.zeros(NJ)
            e[j] = 1.0
            M[:, j] = self.rnea(q, z, e, X)
        return 0.5 * (M + M.T)

    def forward_dynamics_and_gradient(self, q, qd, u, h=1e-6):
        """(qdd, dqdd/dq, dqdd/dqd, dqdd/du) as forwardDynamicsAndGradient computes them."""
        Minv = np.linalg.inv(self.mass_matrix(q))
        qdd = Minv @ (u - self.rnea(q, qd, np.zeros(NJ)))
        dq = np.zeros((NJ, NJ))
        dqd = np.zeros((NJ, NJ))
        for j in range(NJ):
            e = np.zeros(NJ)
            e[j] = h
            dq[:, j] = (self.rnea(q + e, qd, qdd) - self.rnea(q - e, qd, qdd)) / (2 * h)
            dqd[:, j] = (self.rnea(q, qd + e, qdd) - self.rnea(q, qd - e, qdd)) / (2 * h)
        return qdd, -Minv @ dq, -Minv @ dqd, Minv


def generate_kkt(model: Model, xu, ee_goals, xs, knot_points: int, dt: float = TIMESTEP):
    """generate_kkt_submatrices (include/common/kkt.cuh:22-163) for ONE trajectory.
    xu [(n+m)N - m] = x_0,u_0,...,x_{N-1};  ee_goals [N][6] (only xyz used);  xs [n].
    Returns float64 (G_dense, C_dense, g, c) in the reference's dense layouts (column-major blocks, C = -A, -B)."""
    n, m, N = 2 * NJ, NJ, knot_points
    R = r_cost(N)
    G = np.zeros((n * n + m * m) * N - m * m)
    C = np.zeros((n * n + n * m) * (N - 1))
    g = np.zeros((n + m) * N - m)
    c = np.zeros(n * N)

    def cost(x, goal):
        q, qd = x[:NJ], x[NJ:]
        gq = model.ee_jac(q).T @ (model.ee_pos(q) - goal[:3])
        Q = np.zeros((n, n))
        Q[:NJ, :NJ] = np.outer(gq, gq)
        Q[NJ:, NJ:] = QD_COST * np.eye(NJ)
        return Q, np.concatenate([gq, QD_COST * qd])

    c[:n] = xu[:n] - xs
    for k in range(N - 1):
        x = xu[k * (n + m):k * (n + m) + n]
        u = xu[k * (n + m) + n:(k + 1) * (n + m)]
        xn = xu[(k + 1) * (n + m):(k + 1) * (n + m) + n]
        q, qd = x[:NJ], x[NJ:]
        qdd, dq, dqd, du = model.forward_dynamics_and_gradient(q, qd, u)
        A = np.eye(n)
        A[:NJ, NJ:] += dt * np.eye(NJ)
        A[NJ:, :NJ] += dt * dq
        A[NJ:, NJ:] += dt * dqd
        B = np.zeros((n, m))
        B[NJ:, :] = dt * du
        c[(k + 1) * n:(k + 2) * n] = xn - np.concatenate([q + dt * qd, qd + dt * qdd])
        Q, qk = cost(x, ee_goals[k])
        o = (n * n + m * m) * k
        G[o:o + n * n] = Q.T.reshape(-1)
        G[o + n * n:o + n * n + m * m] = (R * np.eye(m)).reshape(-1)
        g[(n + m) * k:(n + m) * k + n] = qk
        g[(n + m) * k + n:(n + m) * (k + 1)] = R * u
        oc = (n * n + n * m) * k
        C[oc:oc + n * n] = (-A).T.reshape(-1)
        C[oc + n * n:oc + n * n + n * m] = (-B).T.reshape(-1)
        if k == N - 2:                                   # last block: cost of knot N-1 evaluated at x_{N-2} (reference quirk)
            Q1, q1 = cost(x, ee_goals[k + 1])
            G[(n * n + m * m) * (k + 1):(n * n + m * m) * (k + 1) + n * n] = Q1.T.reshape(-1)
            g[(n + m) * (k + 1):(n + m) * (k + 1) + n] = q1
    return G, C, g, c


def euler_defect(model: Model, x, u, x_next, dt: float = TIMESTEP):
    """The integrator defect generate_kkt_submatrices stores as c_{k+1} (include/common/kkt.cuh:117,160; Euler, integrator.cuh:56-104):
    x_next - [q + dt qd; qd + dt FD(q, qd, u)]."""
    q, qd = x[:NJ], x[NJ:]
    Minv = np.linalg.inv(model.mass_matrix(q))
    qdd = Minv @ (u - model.rnea(q, qd, np.zeros(NJ)))
    return x_next - np.concatenate([q + dt * qd, qd + dt * qdd])


# The 0_0 trajectory is five point-to-point segments: row `s` of SEGMENT_STARTS is the first row of a segment (a waypoint with zero torque
# and a start-up velocity perturbation of +-0.01..0.05 rad/s that was not integrated: the step out of it leaves q unchanged — q defect
# exactly -dt * qd[s], qd defect <= 6e-5), row s - 1 the last row of the previous one (the step across the seam is a jump to the next
# waypoint, not dynamics).  All other 656 transitions are Euler steps of the reference's forward dynamics.
SEGMENT_STARTS = (0, 148, 294, 424, 542)


def in_segment_transitions(rows: int = 666):
    """Indices t for which rows t -> t + 1 of 0_0_traj.csv are one Euler step of the reference's forward dynamics."""
    bad = set(SEGMENT_STARTS) | {s - 1 for s in SEGMENT_STARTS if s > 0}
    return [t for t in range(rows - 1) if t not in bad]

