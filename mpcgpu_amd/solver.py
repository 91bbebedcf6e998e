"""Host-side mirror of the reference's PCG interface, over the C ABI (include/mpcg.h).

Names follow the reference (GBD-PCG as used by include/pcg/sqp.cuh and include/mpcsim.cuh):
`pcg_config` (fields pcg_block, pcg_exit_tol, pcg_max_iter — include/mpcsim.cuh:213-216),
`pcgSharedMemSize` (include/pcg/sqp.cuh:151), `checkPcgOccupancy`
(examples/track_iiwa_pcg.cu:24).  Tensors are torch CUDA tensors used purely as device memory;
all arithmetic happens in libmpcg_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from .synth import CONTROL_SIZE, STATE_SIZE, pcg_max_iter

PCG_NUM_THREADS = 128   # include/common/settings.cuh:111-113 (reference launch shape; informational)


@dataclass
class pcg_config:
    """include/mpcsim.cuh:213-216."""
    pcg_block: int = PCG_NUM_THREADS
    pcg_exit_tol: float = 1e-4
    pcg_max_iter: int = 167


def pcgSharedMemSize(state_size: int, knot_points: int) -> int:
    """Dynamic LDS bytes of one trajectory's workgroup (0 = unsupported shape)."""
    return int(_lib.load().mpcg_pcg_lds_bytes(state_size, knot_points))


def _ptr(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plant:
    """Device-side robot model for mpcg_generate_kkt: the reference's d_dynMem_const (GRiD robotModel,
    gato_plant::initializeDynamicsConstMem, include/dynamics/iiwa/iiwa_eepos_plant.cuh:63-66) as data."""

    def __init__(self, model=None, device: int | None = None):
        import numpy as np
        from . import iiwa
        self.lib = _lib.load()
        self.model = model or iiwa.Model()
        m = self.model
        dev = torch.cuda.current_device() if device is None else int(device)
        xi = np.array([t[0] for t in m.X_trig], np.int32); xc = np.array([t[1] for t in m.X_trig], np.float64); xj = np.array([t[2] for t in m.X_trig], np.int32)
        hi = np.array([t[0] for t in m.Xhom_trig], np.int32); hc = np.array([t[1] for t in m.Xhom_trig], np.float64); hj = np.array([t[2] for t in m.Xhom_trig], np.int32)
        Icol = np.ascontiguousarray(m.I.transpose(0, 2, 1).reshape(-1))          # back to the column-major table
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        Xc, Hc = np.ascontiguousarray(m.X_const), np.ascontiguousarray(m.Xhom_const)
        h = C.c_void_p()
        rc = self.lib.mpcg_plant_create(C.byref(h), dev, iiwa.NJ, p(Xc), p(Icol), p(Hc), p(xi), p(xc), p(xj), len(xi), p(hi), p(hc), p(hj), len(hi))
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, self.lib.mpcg_last_error(None).decode())
        self._p = h

    def close(self):
        if getattr(self, "_p", None):
            self.lib.mpcg_plant_destroy(self._p)
            self._p = None

    __del__ = close


class QdldlSolver:
    """The reference's LINSYS_SOLVE == 0 path (include/qdldl/sqp.cuh) as a selectable solver: CSR lower-triangle pattern
    (include/utils/csr.cuh:40-73) + elimination tree once, then numeric LDL^T factor + solve per call on the HOST
    (libmpcg_hip's own implementation of the QDLDL algorithm; pure host code, works without a GPU)."""

    def __init__(self, knot_points: int, state_size: int = STATE_SIZE):
        import numpy as np
        self.lib = _lib.load()
        self.n, self.N = int(state_size), int(knot_points)
        h = C.c_void_p()
        rc = self.lib.mpcg_ldl_create(C.byref(h), self.n, self.N)
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, "mpcg_ldl_create")
        self._l = h
        cp, ri = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        nnz, slnz = C.c_uint32(), C.c_uint32()
        self.lib.mpcg_ldl_pattern(self._l, C.byref(cp), C.byref(ri), C.byref(nnz), C.byref(slnz))
        self.nnz, self.sum_lnz = nnz.value, slnz.value
        self.col_ptr = np.ctypeslib.as_array(cp, shape=(self.n * self.N + 1,)).copy()
        self.row_ind = np.ctypeslib.as_array(ri, shape=(self.nnz,)).copy()

    def close(self):
        if getattr(self, "_l", None):
            self.lib.mpcg_ldl_destroy(self._l)
            self._l = None

    __del__ = close

    def solve_host(self, val, gamma):
        """qdldl_solve_schur (include/qdldl/sqp.cuh:22-49) on host arrays: val [nnz] float32, gamma [nN] float32."""
        import numpy as np
        val = np.ascontiguousarray(val, np.float32)
        gamma = np.ascontiguousarray(gamma, np.float32)
        assert val.size == self.nnz and gamma.size == self.n * self.N
        lam = np.empty_like(gamma)
        rc = self.lib.mpcg_ldl_solve(self._l, val.ctypes.data_as(C.c_void_p), gamma.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p))
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, "mpcg_ldl_solve: zero pivot")
        return lam

    def solve_schur(self, sol: "PcgSolver", d_val, d_gamma, d_lambda):
        """The reference's timed region (include/qdldl/sqp.cuh:268-273): D2H(values, gamma), factor + solve, H2D(lambda)."""
        rc = self.lib.mpcg_qdldl_solve_schur(sol._h, self._l, _ptr(d_val), _ptr(d_gamma), _ptr(d_lambda), _stream())
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, self.lib.mpcg_last_error(sol._h).decode())
        return d_lambda


class PcgSolver:
    """One handle per (device, state_size, knot_points).  `solve` is the batched hot path,
    `solve_ref` the reference's single-trajectory 12-argument launch."""

    def __init__(self, knot_points: int, max_batch: int = 1, state_size: int = STATE_SIZE, device: int | None = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("mpcgpu_amd.PcgSolver needs a HIP device (no CPU fallback)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.n, self.N, self.max_batch = int(state_size), int(knot_points), int(max_batch)
        h = C.c_void_p()
        rc = self.lib.mpcg_create(C.byref(h), self.device, self.n, self.N, self.max_batch)
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, self.lib.mpcg_last_error(None).decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mpcg_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != _lib.MPCG_OK:
            raise _lib.MpcgError(rc, self.lib.mpcg_last_error(self._h).decode())

    def set_option(self, key: str, value: int):
        self._check(self.lib.mpcg_set_option(self._h, key.encode(), int(value)))

    def last_error(self) -> str:
        """mpcg_last_error: the text of the handle's last error — or warning (the symmetry latch leaves one)."""
        return self.lib.mpcg_last_error(self._h).decode()

    def get_option(self, key: str) -> int:
        v = C.c_int()
        self._check(self.lib.mpcg_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def checkPcgOccupancy(self) -> int:
        """Number of trajectories resident on the GPU at once (never aborts: no grid sync here)."""
        v = C.c_uint32()
        self._check(self.lib.mpcg_check_pcg_occupancy(self._h, C.byref(v)))
        return v.value

    def _chk(self, t, numel, dtype, name):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype and t.numel() == numel):
            raise ValueError(f"{name}: need contiguous cuda {dtype} tensor with {numel} elements, "
                             f"got {tuple(t.shape)} {t.dtype} {t.device}")

    def solve(self, S, Pinv, gamma, lam, config: pcg_config | None = None, precond: str = "ss",
              iters: torch.Tensor | None = None, exits: torch.Tensor | None = None):
        """In-place batched solve.  S, Pinv: [B, 3*n*n*N]; gamma, lam: [B, n*N] (lam in/out).
        Returns (iters uint32-as-int32 [B], max_iter_exit uint8 [B]) device tensors; no sync."""
        cfg = config or pcg_config(pcg_max_iter=pcg_max_iter(self.N))
        B = lam.shape[0] if lam.dim() > 1 else 1
        n, N = self.n, self.N
        self._chk(S, B * 3 * n * n * N, torch.float32, "S")
        self._chk(Pinv, B * 3 * n * n * N, torch.float32, "Pinv")
        self._chk(gamma, B * n * N, torch.float32, "gamma")
        self._chk(lam, B * n * N, torch.float32, "lambda")
        if iters is None:
            iters = torch.empty(B, dtype=torch.int32, device=lam.device)
        if exits is None:
            exits = torch.empty(B, dtype=torch.uint8, device=lam.device)
        pc = _lib.MPCG_PRECOND_SS if precond == "ss" else _lib.MPCG_PRECOND_JACOBI
        if precond not in ("ss", "jacobi"):
            raise ValueError("precond must be 'ss' or 'jacobi'")
        self._check(self.lib.mpcg_pcg_solve(self._h, _ptr(S), _ptr(Pinv), _ptr(gamma), _ptr(lam), B,
                                            int(cfg.pcg_max_iter), float(cfg.pcg_exit_tol), pc,
                                            _ptr(iters), _ptr(exits), _stream()))
        return iters, exits

    def to_f16(self, M: torch.Tensor) -> torch.Tensor:
        """Round a bd-layout fp32 matrix buffer to half precision on the device (mpcg_convert_f32_to_f16)."""
        self._chk(M, M.numel(), torch.float32, "M")
        out = torch.empty(M.shape, dtype=torch.float16, device=M.device)
        self._check(self.lib.mpcg_convert_f32_to_f16(self._h, _ptr(M), _ptr(out), M.numel(), _stream()))
        return out

    def solve_f16(self, S16, Pinv16, gamma, lam, config: pcg_config | None = None, precond: str = "ss",
                  iters: torch.Tensor | None = None, exits: torch.Tensor | None = None):
        """`solve` with S / Pinv stored in half precision (arithmetic stays fp32)."""
        cfg = config or pcg_config(pcg_max_iter=pcg_max_iter(self.N))
        B = lam.shape[0] if lam.dim() > 1 else 1
        n, N = self.n, self.N
        self._chk(S16, B * 3 * n * n * N, torch.float16, "S16")
        self._chk(Pinv16, B * 3 * n * n * N, torch.float16, "Pinv16")
        self._chk(gamma, B * n * N, torch.float32, "gamma")
        self._chk(lam, B * n * N, torch.float32, "lambda")
        if iters is None:
            iters = torch.empty(B, dtype=torch.int32, device=lam.device)
        if exits is None:
            exits = torch.empty(B, dtype=torch.uint8, device=lam.device)
        if precond not in ("ss", "jacobi"):
            raise ValueError("precond must be 'ss' or 'jacobi'")
        pc = _lib.MPCG_PRECOND_SS if precond == "ss" else _lib.MPCG_PRECOND_JACOBI
        self._check(self.lib.mpcg_pcg_solve_f16(self._h, _ptr(S16), _ptr(Pinv16), _ptr(gamma), _ptr(lam), B,
                                                int(cfg.pcg_max_iter), float(cfg.pcg_exit_tol), pc,
                                                _ptr(iters), _ptr(exits), _stream()))
        return iters, exits

    def solve_f64(self, S, Pinv, gamma, lam, config: pcg_config | None = None, precond: str = "ss",
                  iters: torch.Tensor | None = None, exits: torch.Tensor | None = None):
        """`solve` in double precision (linsys_t = double, USE_DOUBLES=1 of include/common/settings.cuh:41-49)."""
        cfg = config or pcg_config(pcg_max_iter=pcg_max_iter(self.N))
        B = lam.shape[0] if lam.dim() > 1 else 1
        n, N = self.n, self.N
        for t, k, nm in ((S, 3 * n * n, "S"), (Pinv, 3 * n * n, "Pinv"), (gamma, n, "gamma"), (lam, n, "lambda")):
            self._chk(t, B * k * N, torch.float64, nm)
        if iters is None:
            iters = torch.empty(B, dtype=torch.int32, device=lam.device)
        if exits is None:
            exits = torch.empty(B, dtype=torch.uint8, device=lam.device)
        if precond not in ("ss", "jacobi"):
            raise ValueError("precond must be 'ss' or 'jacobi'")
        pc = _lib.MPCG_PRECOND_SS if precond == "ss" else _lib.MPCG_PRECOND_JACOBI
        self._check(self.lib.mpcg_pcg_solve_f64(self._h, _ptr(S), _ptr(Pinv), _ptr(gamma), _ptr(lam), B,
                                                int(cfg.pcg_max_iter), float(cfg.pcg_exit_tol), pc,
                                                _ptr(iters), _ptr(exits), _stream()))
        return iters, exits

    def solve_ref(self, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, d_v_temp, d_eta_new_temp,
                  d_pcg_iters, d_pcg_exit, pcg_max_iter: int, pcg_exit_tol: float):
        """The reference kernel's argument list, in order (include/pcg/sqp.cuh:137-150)."""
        self._check(self.lib.mpcg_pcg_solve_ref(self._h, _ptr(d_S), _ptr(d_Pinv), _ptr(d_gamma), _ptr(d_lambda),
                                                _ptr(d_r), _ptr(d_p), _ptr(d_v_temp), _ptr(d_eta_new_temp),
                                                _ptr(d_pcg_iters), _ptr(d_pcg_exit),
                                                int(pcg_max_iter), float(pcg_exit_tol), _stream()))

    def solve_ref_f64(self, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, d_v_temp, d_eta_new_temp,
                      d_pcg_iters, d_pcg_exit, pcg_max_iter: int, pcg_exit_tol: float):
        """The reference kernel's argument list with linsys_t = double (pcg<double, n, N>; include/pcg/sqp.cuh:137-150)."""
        self._check(self.lib.mpcg_pcg_solve_ref_f64(self._h, _ptr(d_S), _ptr(d_Pinv), _ptr(d_gamma), _ptr(d_lambda),
                                                    _ptr(d_r), _ptr(d_p), _ptr(d_v_temp), _ptr(d_eta_new_temp),
                                                    _ptr(d_pcg_iters), _ptr(d_pcg_exit),
                                                    int(pcg_max_iter), float(pcg_exit_tol), _stream()))

    def block_solve(self, S, gamma, lam=None):
        """Batched block-tridiagonal direct solve (the GPU counterpart of qdldl_solve_schur,
        include/qdldl/sqp.cuh:22-49).  S: [B, 3*n*n*N], gamma: [B, n*N]; returns lambda [B, n*N]."""
        B = gamma.shape[0] if gamma.dim() > 1 else 1
        n, N = self.n, self.N
        self._chk(S, B * 3 * n * n * N, torch.float32, "S")
        self._chk(gamma, B * n * N, torch.float32, "gamma")
        if lam is None:
            lam = torch.empty(B, n * N, device=gamma.device)
        self._chk(lam, B * n * N, torch.float32, "lambda")
        self._check(self.lib.mpcg_block_solve(self._h, _ptr(S), _ptr(gamma), _ptr(lam), B, _stream()))
        return lam

    def form_schur(self, G_dense, C_dense, g, c, rho: float, precond: str = "ss", S=None, Pinv=None, gamma=None,
                   control_size: int = CONTROL_SIZE):
        """form_schur_system (include/pcg/linsys_setup.cuh:620-656), batched.  G_dense is overwritten by
        its block inverses (the reference's side effect).  Returns (S, Pinv, gamma) device tensors."""
        B = c.shape[0] if c.dim() > 1 else 1
        n, m, N = self.n, control_size, self.N
        dt = c.dtype                                           # float32, or float64 = linsys_t double (mpcg_form_schur_f64)
        if dt not in (torch.float32, torch.float64):
            raise TypeError("form_schur: float32 or float64 tensors")
        self._chk(G_dense, B * ((n * n + m * m) * N - m * m), dt, "G_dense")
        self._chk(C_dense, B * (n * n + n * m) * (N - 1), dt, "C_dense")
        self._chk(g, B * ((n + m) * N - m), dt, "g")
        self._chk(c, B * n * N, dt, "c")
        dev = c.device
        S = torch.empty(B, 3 * n * n * N, device=dev, dtype=dt) if S is None else S
        Pinv = torch.empty(B, 3 * n * n * N, device=dev, dtype=dt) if Pinv is None else Pinv
        gamma = torch.empty(B, n * N, device=dev, dtype=dt) if gamma is None else gamma
        self._chk(S, B * 3 * n * n * N, dt, "S")
        self._chk(Pinv, B * 3 * n * n * N, dt, "Pinv")
        self._chk(gamma, B * n * N, dt, "gamma")
        pc = {"ss": _lib.MPCG_PRECOND_SS, "jacobi": _lib.MPCG_PRECOND_JACOBI, "none": _lib.MPCG_PRECOND_NONE}[precond]
        fn = self.lib.mpcg_form_schur if dt == torch.float32 else self.lib.mpcg_form_schur_f64
        self._check(fn(self._h, m, _ptr(G_dense), _ptr(C_dense), _ptr(g), _ptr(c), _ptr(S), _ptr(Pinv), _ptr(gamma), float(rho), B, pc, _stream()))
        return S, Pinv, gamma

    def generate_kkt(self, plant: "Plant", eePos_traj, xs, xu, timestep: float, qd_cost: float, r_cost: float,
                     control_size: int = CONTROL_SIZE):
        """generate_kkt_submatrices (include/common/kkt.cuh:22-163), batched: eePos_traj [B, 6N], xs [B, n], xu [B, (n+m)N - m]
        -> (G_dense, C_dense, g, c) device tensors in the layouts form_schur consumes."""
        B = xu.shape[0] if xu.dim() > 1 else 1
        n, m, N = self.n, control_size, self.N
        self._chk(eePos_traj, B * 6 * N, torch.float32, "eePos_traj")
        self._chk(xs, B * n, torch.float32, "xs")
        self._chk(xu, B * ((n + m) * N - m), torch.float32, "xu")
        dev = xu.device
        G = torch.empty(B, (n * n + m * m) * N - m * m, device=dev)
        Cd = torch.empty(B, (n * n + n * m) * (N - 1), device=dev)
        g = torch.empty(B, (n + m) * N - m, device=dev)
        c = torch.empty(B, n * N, device=dev)
        self._check(self.lib.mpcg_generate_kkt(self._h, plant._p, m, float(timestep), _ptr(eePos_traj), _ptr(xs), _ptr(xu), float(qd_cost),
                                               float(r_cost), _ptr(G), _ptr(Cd), _ptr(g), _ptr(c), B, _stream()))
        return G, Cd, g, c

    def compute_dz(self, Ginv_dense, C_dense, g, lam, dz=None, control_size: int = CONTROL_SIZE):
        """compute_dz (include/common/dz.cuh:124-136), batched."""
        B = lam.shape[0] if lam.dim() > 1 else 1
        n, m, N = self.n, control_size, self.N
        dt = lam.dtype                                         # float32, or float64 = linsys_t double (mpcg_compute_dz_f64)
        if dt not in (torch.float32, torch.float64):
            raise TypeError("compute_dz: float32 or float64 tensors")
        self._chk(Ginv_dense, B * ((n * n + m * m) * N - m * m), dt, "Ginv_dense")
        self._chk(C_dense, B * (n * n + n * m) * (N - 1), dt, "C_dense")
        self._chk(g, B * ((n + m) * N - m), dt, "g")
        self._chk(lam, B * n * N, dt, "lam")
        if dz is None:
            dz = torch.empty(B, (n + m) * N - m, device=lam.device, dtype=dt)
        self._chk(dz, B * ((n + m) * N - m), dt, "dz")
        fn = self.lib.mpcg_compute_dz if dt == torch.float32 else self.lib.mpcg_compute_dz_f64
        self._check(fn(self._h, m, _ptr(Ginv_dense), _ptr(C_dense), _ptr(g), _ptr(lam), _ptr(dz), B, _stream()))
        return dz

    def csr_nnz(self) -> int:
        """nnz of the lower triangle (include/qdldl/sqp.cuh:148)."""
        n, N = self.n, self.N
        return (N - 1) * n * n + N * (n * (n + 1)) // 2

    def prep_csr(self):
        """prep_csr (include/utils/csr.cuh:40-73): (col_ptr int32 [nN+1], row_ind int32 [nnz]) on the device."""
        dev = torch.device("cuda", self.device)
        col_ptr = torch.empty(self.n * self.N + 1, dtype=torch.int32, device=dev)
        row_ind = torch.empty(self.csr_nnz(), dtype=torch.int32, device=dev)
        self._check(self.lib.mpcg_prep_csr(self._h, _ptr(col_ptr), _ptr(row_ind), _stream()))
        return col_ptr, row_ind

    def bd_to_csr_lowertri(self, S, mult: float = 1.0):
        """Values of the QDLDL path's d_val from a bd-layout S, batched: [B, nnz]."""
        B = S.shape[0] if S.dim() > 1 else 1
        self._chk(S, B * 3 * self.n * self.n * self.N, torch.float32, "S")
        val = torch.empty(B, self.csr_nnz(), device=S.device)
        self._check(self.lib.mpcg_bd_to_csr_lowertri(self._h, _ptr(S), _ptr(val), float(mult), B, _stream()))
        return val

    def probe_hbm_read(self, src, sink=None):
        """Pure read of `src` (measurement aid: the device's HBM read ceiling, mpcg_probe_hbm_read)."""
        if sink is None:
            sink = torch.zeros(1, dtype=torch.float32, device=src.device)
        nbytes = src.numel() * src.element_size()
        self._check(self.lib.mpcg_probe_hbm_read(self._h, _ptr(src), nbytes - nbytes % 16, _ptr(sink), _stream()))
        return sink

    def bt_spmv(self, M, x, y=None, cols: int = 3):
        B = x.shape[0] if x.dim() > 1 else 1
        n, N = self.n, self.N
        self._chk(M, B * 3 * n * n * N, torch.float32, "M")
        self._chk(x, B * n * N, torch.float32, "x")
        if y is None:
            y = torch.empty_like(x)
        self._check(self.lib.mpcg_bt_spmv(self._h, _ptr(M), _ptr(x), _ptr(y), B, int(cols), _stream()))
        return y
