"""Generates tests/golden/*.npz.  Run in the build container: `python tests/make_golden.py`.

The reference has no golden vectors for the PCG boundary ("parity unpinned"),
and its PCG source (submodule GBD-PCG) is absent, so these fixtures pin the ORACLE and the HIP
path to mathematics instead: inputs are float32 (S, Pinv, gamma) in the reference's bd layout
(produced by mpcgpu_amd.synth from seeded IIWA-shaped KKT blocks); expected outputs come from an
independent dense numpy/float64 implementation written here (no code shared with oracle/ or the
HIP kernels):
  lam_direct   : numpy.linalg.solve on the dense matrix
  lam_K        : PCG iterate after exactly K iterations, float64, for K in KS (both preconditioners)
  eta_hist     : |eta| after setup and after every iteration (float64)
  iters_tol    : iterations until |eta| < TOL (float64)
  spmv_x / spmv_y : y = S x for a seeded x
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpcgpu_amd import synth  # noqa: E402

KS = (5, 20, 50)
TOL = 1e-4


def dense_pcg(Sd, Pd, g, lam0, iters, tol=0.0):
    lam = lam0.copy()
    r = g - Sd @ lam
    rt = Pd @ r
    p = rt.copy()
    eta = r @ rt
    hist = [abs(eta)]
    snaps = {}
    done = 0
    for it in range(1, iters + 1):
        u = Sd @ p
        alpha = eta / (p @ u)
        lam = lam + alpha * p
        r = r - alpha * u
        rt = Pd @ r
        en = r @ rt
        hist.append(abs(en))
        done = it
        snaps[it] = lam.copy()
        if abs(en) < tol:
            break
        p = rt + (en / eta) * p
        eta = en
    return lam, done, np.array(hist), snaps


def make(N, seed):
    k = synth.make_kkt(N, 1, seed)
    S, P, g = synth.form_schur(k, precond="ss", dtype=np.float32)
    S, P, g = S[0], P[0], g[0]
    Sd = synth.bd_to_dense(S, N)
    Pss = synth.bd_to_dense(P, N)
    Pj = np.zeros_like(Pss)
    n = synth.STATE_SIZE
    for kk in range(N):
        Pj[kk * n:(kk + 1) * n, kk * n:(kk + 1) * n] = Pss[kk * n:(kk + 1) * n, kk * n:(kk + 1) * n]
    g64 = g.astype(np.float64)
    rng = np.random.default_rng(seed + 99)
    lam0 = np.zeros(n * N)
    lam_w = rng.normal(0, 1.0, n * N).astype(np.float32)     # a warm start
    out = dict(N=N, seed=seed, S=S, Pinv=P, gamma=g, lam_warm=lam_w,
               lam_direct=np.linalg.solve(Sd, g64))
    for name, Pd in (("ss", Pss), ("jacobi", Pj)):
        _, _, hist, snaps = dense_pcg(Sd, Pd, g64, lam0, max(KS))
        for K in KS:
            out[f"lam_{name}_K{K}"] = snaps[K]
        out[f"eta_hist_{name}"] = hist
        _, it_tol, _, _ = dense_pcg(Sd, Pd, g64, lam0, 5000, TOL)
        out[f"iters_tol_{name}"] = it_tol
        _, _, _, snaps_w = dense_pcg(Sd, Pd, g64, lam_w.astype(np.float64), 20)
        out[f"lam_warm_{name}_K20"] = snaps_w[20]
    x = rng.normal(0, 1.0, n * N).astype(np.float32)
    out["spmv_x"] = x
    out["spmv_y"] = Sd @ x.astype(np.float64)
    out["precond_y"] = Pss @ x.astype(np.float64)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    for N, seed in ((8, 11), (32, 12)):
        d = make(N, seed)
        path = os.path.join(ROOT, "tests", "golden", f"pcg_n14_N{N}.npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path), "bytes; iters_tol ss/jacobi:", d["iters_tol_ss"], d["iters_tol_jacobi"])
