"""Host-side data of the IIWA-14 tracking problem for mpcg_generate_kkt: the robot's model tables, the
reference's constants and a generator of tracking windows from the reference's precomputed trajectory.  No dynamics here: the numpy
restatement that CHECKS the HIP kernel lives in oracle/iiwa_ref.py (test infrastructure), the kernel in csrc/kkt_plant.hip.h.

The robot is DATA: mpcgpu_amd/data/iiwa14_model.json = the spatial transforms, spatial inertias and homogeneous transforms of the
KUKA LBR iiwa 14 as tabulated in include/dynamics/iiwa/iiwa_eepos_grid.cuh:909-1904 (extracted by tests/make_iiwa_golden.py).
"""
from __future__ import annotations

import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_PATH = os.path.join(_HERE, "data", "iiwa14_model.json")
NJ = 7
QD_COST = 1e-4          # include/common/settings.cuh:92-94
TIMESTEP = 1.0 / 64     # examples/track_iiwa_pcg.cu:19


def r_cost(knot_points: int) -> float:
    """include/common/settings.cuh:84-90"""
    return 1e-3 if knot_points == 64 else 1e-4


class Model:
    """The model tables mpcg_plant_create takes (solver.Plant): constant parts of the spatial / homogeneous transforms, the entries that
    are coef * sin(q_j) or coef * cos(q_j), spatial inertias."""

    def __init__(self, path: str = MODEL_PATH):
        d = json.load(open(path))
        self.X_const = np.array(d["X_const"], np.float64)          # 7 x 36, column-major 6x6
        self.X_trig = [(int(i), float(c), int(j)) for i, c, j in d["X_trig"]]
        self.I = np.array(d["I"], np.float64).reshape(NJ, 6, 6).transpose(0, 2, 1)
        self.Xhom_const = np.array(d["Xhom_const"], np.float64)    # 7 x 16, column-major 4x4
        self.Xhom_trig = [(int(i), float(c), int(j)) for i, c, j in d["Xhom_trig"]]


def read_csv(path):
    """readCSVToVecVec (include/utils/experiment.cuh:145-169)."""
    return np.array([[float(v) for v in line.strip().split(",") if v != ""] for line in open(path) if line.strip()])


TRAJ_FIXTURE = os.path.join(_HERE, "data", "iiwa_traj_0_0.npz")


def random_windows(knot_points: int, batch: int, seed: int, max_noise: float = 0.05):
    """`batch` tracking problems cut from the reference's precomputed trajectory (mpcgpu_amd/data/iiwa_traj_0_0.npz = the
    first 400 rows of examples/trajfiles/0_0_traj.csv and 0_0_eepos.traj): random window offset, goals 0..8 steps ahead,
    measured state x_s and iterate perturbed by gaussian noise of random amplitude <= max_noise (rad, rad/s) — the
    'random-init trajectories' of BASELINE config 4 on the real robot.  Returns float64 (xu [B, (n+m)N-m], goals [B, N, 6], xs [B, n])."""
    d = np.load(TRAJ_FIXTURE)
    traj, eep = d["xu"].astype(np.float64), d["eepos"].astype(np.float64)
    n, m, N = 2 * NJ, NJ, knot_points
    rng = np.random.default_rng(seed)
    xu = np.zeros((batch, (n + m) * N - m))
    goals = np.zeros((batch, N, 6))
    xs = np.zeros((batch, n))
    for b in range(batch):
        t0 = int(rng.integers(0, traj.shape[0] - N - 8))
        sh = int(rng.integers(0, 9))
        w = traj[t0:t0 + N].reshape(-1)[:(n + m) * N - m].copy()
        amp = max_noise * rng.random()
        xs[b] = w[:n] + amp * rng.standard_normal(n)
        w += 0.3 * amp * rng.standard_normal(w.shape)
        w[:n] = xs[b]
        xu[b], goals[b] = w, eep[t0 + sh:t0 + sh + N]
    return xu, goals, xs
