#!/usr/bin/env python3
"""Runs in the BUILD container only (reads /root/reference): stores ALL 666 rows of the reference's own precomputed trajectory pair
    examples/trajfiles/0_0_traj.csv   (x = [q(7), qd(7)], u(7) per row — the file the reference's examples track, experiment.cuh:145-169)
    examples/trajfiles/0_0_eepos.traj (end-effector pose per row)
as tests/golden/iiwa_traj_0_0_full.npz (float64, the parsed numbers as they are).  The file is DATA the reference holds — inputs and
outputs of its own dynamics: consecutive rows are one Euler step of its forward dynamics (dt = 1/64) apart, which is what
tests/test_iiwa_plant.py::test_forward_dynamics_reproduce_the_reference_trajectory and tests/test_gpu_kkt.py pin the restated plant on."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/examples/trajfiles"


def read_csv(path):
    return np.array([[float(v) for v in line.strip().split(",") if v != ""] for line in open(path) if line.strip()])


if __name__ == "__main__":
    xu, eep = read_csv(os.path.join(REF, "0_0_traj.csv")), read_csv(os.path.join(REF, "0_0_eepos.traj"))
    assert xu.shape == (666, 21) and eep.shape == (666, 6)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "iiwa_traj_0_0_full.npz"), xu=xu, eepos=eep)
    print("wrote", xu.shape, eep.shape)
