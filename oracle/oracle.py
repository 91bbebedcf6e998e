"""ctypes binding of the CPU oracle (oracle/libmpcg_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from
mpcgpu_amd/.  PARITY UNPINNED (see mpcg_oracle.c).  Build: `make -C oracle`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmpcg_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("mpcg_oracle.c", "mpcg_oracle_impl.inc", "mpcg_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmpcg_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


_CT = {np.dtype(np.float32): (C.c_float, "f32"), np.dtype(np.float64): (C.c_double, "f64")}


def _real(dtype):
    return _CT[np.dtype(dtype)]


def form_schur(G, Cd, g, c, N, rho, ss=True, n=14, m=7):
    """One trajectory.  Returns (S, Pinv, gamma, Ginv); untouched bd slots are NaN."""
    ct, suf = _real(G.dtype)
    G = np.array(G, copy=True)
    S = np.full(3 * n * n * N, np.nan, G.dtype)
    P = np.full(3 * n * n * N, np.nan, G.dtype)
    gam = np.zeros(n * N, G.dtype)
    f = getattr(lib(), f"orc_form_schur_{suf}")
    f.restype = C.c_int
    rc = f(n, m, N, _p(G, ct), _p(np.ascontiguousarray(Cd), ct), _p(np.ascontiguousarray(g), ct),
           _p(np.ascontiguousarray(c), ct), _p(S, ct), _p(P, ct), _p(gam, ct), ct(rho), int(bool(ss)))
    assert rc == 0
    return S, P, gam, G


def bt_spmv(M, x, N, cols=3, n=14):
    ct, suf = _real(M.dtype)
    x = np.ascontiguousarray(x, M.dtype)
    y = np.zeros(n * N, M.dtype)
    getattr(lib(), f"orc_bt_spmv_{suf}")(n, N, _p(np.ascontiguousarray(M), ct), _p(x, ct), _p(y, ct), cols)
    return y


def pcg(S, Pinv, gamma, lam0, N, max_iter, exit_tol, precond="ss", n=14, hist=False):
    """One trajectory.  Returns dict(lambda, iters, max_iter_exit, r, p[, eta_hist])."""
    ct, suf = _real(S.dtype)
    dt = S.dtype
    lam = np.array(lam0, dtype=dt, copy=True)
    r = np.zeros(n * N, dt)
    p = np.zeros(n * N, dt)
    it = C.c_uint32(0)
    ex = C.c_uint8(0)
    eh = np.zeros(max_iter + 1, dt) if hist else None
    f = getattr(lib(), f"orc_pcg_{suf}")
    f.restype = C.c_int
    rc = f(n, N, _p(np.ascontiguousarray(S), ct), _p(np.ascontiguousarray(Pinv), ct),
           _p(np.ascontiguousarray(gamma, dt), ct), _p(lam, ct), _p(r, ct), _p(p, ct),
           int(max_iter), ct(exit_tol), 3 if precond == "ss" else 1, C.byref(it), C.byref(ex),
           _p(eh, ct) if hist else None)
    assert rc == 0
    out = dict(lam=lam, iters=int(it.value), max_iter_exit=bool(ex.value), r=r, p=p)
    if hist:
        out["eta_hist"] = eh[: it.value + 1]
    return out


def compute_dz(Ginv, Cd, g, lam, N, n=14, m=7):
    ct, suf = _real(Ginv.dtype)
    dz = np.zeros((n + m) * N - m, Ginv.dtype)
    getattr(lib(), f"orc_compute_dz_{suf}")(n, m, N, _p(np.ascontiguousarray(Ginv), ct),
                                            _p(np.ascontiguousarray(Cd), ct), _p(np.ascontiguousarray(g), ct),
                                            _p(np.ascontiguousarray(lam, Ginv.dtype), ct), _p(dz, ct))
    return dz


def direct_solve(S, b, N, n=14):
    """fp64 ground truth for S x = b (S bd layout, any float dtype)."""
    S64 = np.ascontiguousarray(np.nan_to_num(np.asarray(S, np.float64)))
    b64 = np.ascontiguousarray(b, np.float64)
    x = np.zeros(n * N)
    f = lib().orc_bt_direct_solve_f64
    f.restype = C.c_int
    rc = f(n, N, _p(S64, C.c_double), _p(b64, C.c_double), _p(x, C.c_double))
    if rc != 0:
        raise FloatingPointError(f"pivot block {-rc - 1} of -S not positive definite")
    return x


def block_solve(S, gamma, N, n=14):
    """Block-tridiagonal direct solve in the dtype of S (the operation order of mpcgpu_amd/csrc/block_solve.hip.h)."""
    ct, suf = _real(S.dtype)
    S = np.ascontiguousarray(np.nan_to_num(S))
    gamma = np.ascontiguousarray(gamma, S.dtype)
    lam = np.zeros(n * N, S.dtype)
    work = np.zeros(N * (n * n + n), S.dtype)
    f = getattr(lib(), f"orc_bt_block_solve_{suf}")
    f.restype = C.c_int
    rc = f(n, N, _p(S, ct), _p(gamma, ct), _p(lam, ct), _p(work, ct))
    assert rc == 0
    return lam


def prep_csr(N, n=14):
    nnz = (N - 1) * n * n + N * (n * (n + 1)) // 2      # include/qdldl/sqp.cuh:148
    col_ptr = np.zeros(n * N + 1, np.int32)
    row_ind = np.zeros(nnz, np.int32)
    lib().orc_prep_csr(n, N, _p(col_ptr, C.c_int), _p(row_ind, C.c_int))
    return col_ptr, row_ind


def bd_to_csr_lowertri(S, N, mult=1.0, n=14):
    ct, suf = _real(S.dtype)
    nnz = (N - 1) * n * n + N * (n * (n + 1)) // 2
    val = np.zeros(nnz, S.dtype)
    getattr(lib(), f"orc_bd_to_csr_lowertri_{suf}")(n, N, _p(np.ascontiguousarray(S), ct), _p(val, ct), ct(mult))
    return val


class LdlSolver:
    """QDLDL-style solver for one sparsity pattern; mirrors the reference's workspace protocol
    (include/qdldl/sqp.cuh:148-198): symbolic once, numeric factor + solve per call."""

    def __init__(self, N, dtype=np.float32, n=14):
        self.n, self.N, self.An = n, N, n * N
        self.dtype = np.dtype(dtype)
        self.ct, self.suf = _real(dtype)
        self.Ap, self.Ai = prep_csr(N, n)
        An = self.An
        self.etree = np.zeros(An, np.int32)
        self.Lnz = np.zeros(An, np.int32)
        self.iwork = np.zeros(3 * An, np.int32)
        f = lib().orc_ldl_etree
        f.restype = C.c_int
        self.sumLnz = f(An, _p(self.Ap, C.c_int), _p(self.Ai, C.c_int), _p(self.iwork, C.c_int),
                        _p(self.Lnz, C.c_int), _p(self.etree, C.c_int))
        assert self.sumLnz >= 0
        self.Lp = np.zeros(An + 1, np.int32)
        self.Li = np.zeros(max(self.sumLnz, 1), np.int32)
        self.Lx = np.zeros(max(self.sumLnz, 1), dtype)
        self.D = np.zeros(An, dtype)
        self.Dinv = np.zeros(An, dtype)
        self.bwork = np.zeros(An, np.uint8)
        self.fwork = np.zeros(An, dtype)
        self._f = getattr(lib(), f"orc_ldl_solve_schur_{self.suf}")
        self._f.restype = C.c_int

    def solve(self, val, b):
        ct = self.ct
        val = np.ascontiguousarray(val, self.dtype)
        b = np.ascontiguousarray(b, self.dtype)
        x = np.zeros(self.An, self.dtype)
        rc = self._f(self.An, _p(self.Ap, C.c_int), _p(self.Ai, C.c_int), _p(val, ct), _p(b, ct), _p(x, ct),
                     _p(self.Lp, C.c_int), _p(self.Li, C.c_int), _p(self.Lx, ct), _p(self.D, ct),
                     _p(self.Dinv, ct), _p(self.Lnz, C.c_int), _p(self.etree, C.c_int),
                     _p(self.bwork, C.c_ubyte), _p(self.iwork, C.c_int), _p(self.fwork, ct))
        if rc < 0:
            raise FloatingPointError("zero pivot in LDL^T")
        self.positive_D = rc
        return x

    def throughput(self, vals, bs, nthreads, seconds):
        """(solves, elapsed s) of `nthreads` POSIX threads doing factor+solve over the given systems
        (float32 only; bench.py cpu_baseline all-cores leg)."""
        assert self.dtype == np.float32
        vals = np.ascontiguousarray(vals, np.float32)
        bs = np.ascontiguousarray(bs, np.float32)
        f = lib().orc_ldl_throughput_f32
        f.restype = C.c_long
        el = C.c_double(0)
        cnt = f(self.An, _p(self.Ap, C.c_int), _p(self.Ai, C.c_int), _p(self.Lnz, C.c_int), _p(self.etree, C.c_int),
                C.c_int(int(self.sumLnz)), _p(vals, C.c_float), _p(bs, C.c_float), C.c_int(vals.shape[0]),
                C.c_int(int(nthreads)), C.c_double(float(seconds)), C.byref(el))
        if cnt < 0:
            raise RuntimeError("orc_ldl_throughput_f32 failed")
        return int(cnt), el.value
