"""CPU: the C-ABI library loads and exports every symbol include/mpcg.h declares; argument
validation that needs no device works; without a GPU the compute path fails loudly (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mpcg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hiplib):
    from mpcgpu_amd import _lib
    names = _declared()
    assert len(names) >= 12
    for nm in names:
        assert hasattr(hiplib, nm), f"{nm} declared in include/mpcg.h but not exported"
        assert nm in _lib.SYMBOLS, f"{nm} has no ctypes signature in mpcgpu_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == names


def test_exports_are_plain_c(hiplib):
    from mpcgpu_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(_declared()) <= exported
    # no torch / python symbols pulled into the product library
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "torch" not in und.lower() and "Py_" not in und


def test_version_and_lds_size(hiplib):
    assert hiplib.mpcg_abi_version() == 2
    assert b"gfx950" in hiplib.mpcg_build_info()
    # = the dynamic LDS of the launch a default batch-1 SS solve makes.
    r4 = lambda x: (x + 3) & ~3
    # N <= 32: the row-per-lane kernel of a batch-1 call — six vectors padded by a knot either side + 2 partials per wave (4 waves
    # for N <= 16, else 8), DESIGN.md §3.1
    for N in (2, 16, 32):
        assert hiplib.mpcg_pcg_lds_bytes(14, N) == 4 * (6 * r4((N + 2) * 14) + r4(2 * (4 if N <= 16 else 8)))
    # 32 < N <= 128 (round 6): the lane-quad kernel, layout compile-time per knots-per-workgroup (64 or 128): seven pair-major vectors of
    # 7 x (NMAX + 5) float2 padded by 10 floats (T / Z bank offset; round 6 first had NMAX + 4 and 24: the same size), two partials per wave, and a second 6,272-byte load tile per wave whose
    # tail holds the parked matrix pairs
    lqb = lambda nmax: 4 * (7 * (7 * (nmax + 4) * 2 + 24) + 2 * (nmax // 16) + (nmax // 16) * 8 * 196)
    for N in (33, 48, 64):
        assert hiplib.mpcg_pcg_lds_bytes(14, N) == lqb(64) == 52448
    for N in (65, 128):
        assert hiplib.mpcg_pcg_lds_bytes(14, N) == lqb(128) == 102656
    # N > 128: a member of the clustered lane-pair kernel — the same seven vectors, broadcast cell, hand-off tables (5 x 64), parked pairs (§3.1g)
    for N in (129, 256, 512):
        assert hiplib.mpcg_pcg_lds_bytes(14, N) == 4 * (7 * 7 * 132 * 2 + 4 + 5 * 64 + 3 * 2 * 8 * 64) == 65328
    assert hiplib.mpcg_pcg_lds_bytes_f64(14, 32) == 8 * (6 * r4((32 + 2) * 14) + 16)          # N <= 32: the row-per-lane kernel in double
    # 32 < N <= 64 (round 5): the lane-quad kernel — seven pair-major vectors of 64 + 2 knot slots, eight wave partials, five parked values per lane
    for N in (33, 64):
        assert hiplib.mpcg_pcg_lds_bytes_f64(14, N) == 8 * (7 * 7 * 66 * 2 + 8 + 5 * 512) == 72288
    # 64 < N <= 512 (round 5): a member of the clustered lane-quad kernel — the same seven vectors, broadcast cell, three hand-off tables of 64 ints, ten parked values per lane
    for N in (65, 128, 256, 300, 512):
        assert hiplib.mpcg_pcg_lds_bytes_f64(14, N) == 8 * (7 * 7 * 66 * 2 + 4 + 3 * 32 + 10 * 512) == 93504
    assert hiplib.mpcg_pcg_lds_bytes_f64(14, 513) == 0          # beyond: the streaming kernel's vectors would not fit 160 KiB of LDS in double
    assert hiplib.mpcg_pcg_lds_bytes(12, 128) == 4 * (2 * 130 * 12 + 2 * 128 * 12 + 16)   # n != 14: the generic kernel's vectors
    assert hiplib.mpcg_pcg_lds_bytes(65, 8) == 0            # state sizes beyond 64 are not served
    assert hiplib.mpcg_pcg_lds_bytes(14, 1024) == 0         # vectors would not fit 160 KiB LDS


def test_create_argument_errors(hiplib):
    from mpcgpu_amd import _lib
    h = C.c_void_p()
    assert hiplib.mpcg_create(None, 0, 14, 32, 1) == _lib.MPCG_ERR_INVALID
    assert hiplib.mpcg_create(C.byref(h), 0, 65, 32, 1) == _lib.MPCG_ERR_UNSUPPORTED
    assert b"state_size" in hiplib.mpcg_last_error(None)
    assert hiplib.mpcg_create(C.byref(h), 0, 14, 32, 0) == _lib.MPCG_ERR_INVALID
    assert hiplib.mpcg_destroy(None) == _lib.MPCG_OK


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(hiplib):
    from mpcgpu_amd import PcgSolver, _lib
    h = C.c_void_p()
    assert hiplib.mpcg_create(C.byref(h), 0, 14, 32, 1) == _lib.MPCG_ERR_HIP
    assert not h.value
    with pytest.raises(RuntimeError):
        PcgSolver(32)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under mpcgpu_amd/ or include/ may reference it."""
    for base in ("mpcgpu_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp", ".cuh", ".inc")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "libmpcg_oracle" not in txt and "import oracle" not in txt and "orc_" not in txt, f
                    assert "import iiwa_ref" not in txt and "from iiwa_ref" not in txt, f      # (the numpy plant restatement: oracle/iiwa_ref.py)
