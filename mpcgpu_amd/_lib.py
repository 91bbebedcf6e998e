"""ctypes binding of the C ABI declared in include/mpcg.h.

The library is the product: if it is missing or does not load, importing fails loudly —
there is no Python/CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpcg_hip.so")

MPCG_OK = 0
MPCG_ERR_INVALID = -1
MPCG_ERR_UNSUPPORTED = -2
MPCG_ERR_HIP = -3
MPCG_ERR_NOMEM = -4
MPCG_PRECOND_NONE = 0
MPCG_PRECOND_JACOBI = 1
MPCG_PRECOND_SS = 3

# every symbol include/mpcg.h declares: (name, restype, argtypes)
_f32p = C.c_void_p   # device pointers travel as integers
SYMBOLS = {
    "mpcg_abi_version": (C.c_int, []),
    "mpcg_build_info": (C.c_char_p, []),
    "mpcg_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mpcg_destroy": (C.c_int, [C.c_void_p]),
    "mpcg_last_error": (C.c_char_p, [C.c_void_p]),
    "mpcg_pcg_lds_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "mpcg_pcg_lds_bytes_f64": (C.c_size_t, [C.c_uint32, C.c_uint32]),
    "mpcg_check_pcg_occupancy": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "mpcg_pcg_solve": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_float,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_pcg_solve_ref": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p,
                                     C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]),
    "mpcg_bt_spmv": (C.c_int, [C.c_void_p, _f32p, _f32p, _f32p, C.c_uint32, C.c_int, C.c_void_p]),
    "mpcg_probe_hbm_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, _f32p, C.c_void_p]),
    "mpcg_convert_f32_to_f16": (C.c_int, [C.c_void_p, _f32p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mpcg_pcg_solve_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_float,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_form_schur": (C.c_int, [C.c_void_p, C.c_uint32, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float,
                                  C.c_uint32, C.c_int, C.c_void_p]),
    "mpcg_compute_dz": (C.c_int, [C.c_void_p, C.c_uint32, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_void_p]),
    "mpcg_form_schur_f64": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_double, C.c_uint32, C.c_int, C.c_void_p]),
    "mpcg_compute_dz_f64": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mpcg_prep_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_bd_to_csr_lowertri": (C.c_int, [C.c_void_p, _f32p, _f32p, C.c_float, C.c_uint32, C.c_void_p]),
    "mpcg_pcg_solve_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_pcg_solve_ref_f64": (C.c_int, [C.c_void_p] * 11 + [C.c_uint32, C.c_double, C.c_void_p]),
    "mpcg_block_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mpcg_plant_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "mpcg_plant_create_iiwa14": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "mpcg_plant_destroy": (C.c_int, [C.c_void_p]),
    "mpcg_generate_kkt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "mpcg_ldl_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32]),
    "mpcg_ldl_destroy": (C.c_int, [C.c_void_p]),
    "mpcg_ldl_pattern": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32)),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "mpcg_ldl_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_qdldl_solve_schur": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mpcg_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mpcg_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m mpcgpu_amd.build` "
                "(hipcc --offload-arch=gfx950).  mpcgpu_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class MpcgError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"mpcg error {code}: {text}")
        self.code = code
