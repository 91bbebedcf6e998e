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
