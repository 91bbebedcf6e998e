"""Build libmpcg_hip.so (gfx950) and the C++ call-site programs in-tree: a thin front end of the top-level Makefile
(`make lib`, `make examples`) — a C++ maintainer builds the boundary with `make`, Python is not needed for it.
`python -m mpcgpu_amd.build [--force]`."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libmpcg_hip.so")
CSRC = os.path.join(_HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
MAKEFILE = os.path.join(_ROOT, "Makefile")


def _ex(name):
    return os.path.join(_ROOT, "examples", name)


EXAMPLE_BIN, EXAMPLE_BIN64 = _ex("sqp_pcg_callsite"), _ex("sqp_pcg_callsite_f64")
EXAMPLE_BIN64_N128 = _ex("sqp_pcg_callsite_f64_n128")
CHAIN_BIN, CHAIN_BIN64 = _ex("sqp_linsys_chain"), _ex("sqp_linsys_chain_f64")
DEMO_BINS = {1: _ex("mpcsim_shim_demo_pcg"), 0: _ex("mpcsim_shim_demo_qdldl")}
IIWA_DEMO_BINS = {1: _ex("mpcsim_iiwa_demo_pcg"), 0: _ex("mpcsim_iiwa_demo_qdldl")}
MULTI_BIN = _ex("multi_gpu_pcg")
UTILS_BIN = _ex("bd_utils_probe")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sources() + [p for pat in ("*.h", "*.hpp", "*.inc") for p in glob.glob(os.path.join(CSRC, pat))] + [os.path.join(_ROOT, "include", "mpcg.h")]


# (tools/prof_phases.py builds a -DMPCG_PROF twin of the library in one hipcc call: the Makefile's flags + -shared)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function"]
SOURCES = sources()


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in deps())


def make(*targets: str, force: bool = False, verbose: bool = False, extra=()):
    """Run the top-level Makefile for `targets` (paths relative to the repo root), translation units in parallel."""
    jobs = str(min(8, os.cpu_count() or 1))
    cmd = ["make", "-C", _ROOT, "-f", MAKEFILE, "-j", jobs, f"HIPCC={HIPCC}", *extra, *targets]
    if force:
        cmd.insert(1, "-B")
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


def _rel(p):
    return os.path.relpath(p, _ROOT)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        make("lib", force=force, verbose=verbose)
    return LIB_PATH


def _build_bins(bins, force, verbose):
    build(False, verbose)
    # (-B would also rebuild the library these depend on: remove the binaries instead)
    if force:
        for b in bins:
            if os.path.exists(b):
                os.remove(b)
    make(*[_rel(b) for b in bins], verbose=verbose)


def build_example(force: bool = False, verbose: bool = False) -> str:
    """C++ host program: the reference's PCG call site over the shim headers + the C ABI."""
    _build_bins([EXAMPLE_BIN], force, verbose)
    return EXAMPLE_BIN


def build_example_f64(force: bool = False, verbose: bool = False) -> str:
    """The same call site compiled with -DUSE_DOUBLES (linsys_t = double): pcg<double, n, N> over the shim."""
    _build_bins([EXAMPLE_BIN64], force, verbose)
    return EXAMPLE_BIN64


def build_example_f64_n128(force: bool = False, verbose: bool = False) -> str:
    """The call site with -DUSE_DOUBLES -DKNOT_POINTS=128: pcg<double, 14, 128> — the clustered row-per-lane kernel behind the shim."""
    _build_bins([EXAMPLE_BIN64_N128], force, verbose)
    return EXAMPLE_BIN64_N128


def build_chain_example(force: bool = False, verbose: bool = False) -> str:
    """C++ host program: form_schur_system -> pcg -> compute_dz as include/pcg/sqp.cuh:207-259 writes them."""
    _build_bins([CHAIN_BIN], force, verbose)
    return CHAIN_BIN


def build_chain_example_f64(force: bool = False, verbose: bool = False) -> str:
    """The same chain compiled with -DUSE_DOUBLES (linsys_t = double)."""
    _build_bins([CHAIN_BIN64], force, verbose)
    return CHAIN_BIN64


def build_mpcsim_demo(force: bool = False, verbose: bool = False):
    """simulateMPC / sqpSolvePcg / sqpSolveQdldl shims, -DLINSYS_SOLVE=1 and =0 (include/mpcsim.cuh:21-25)."""
    _build_bins(list(DEMO_BINS.values()), force, verbose)
    return DEMO_BINS


def build_iiwa_demo(force: bool = False, verbose: bool = False):
    """simulateMPC over the shim headers on a real window of the reference trajectory, KKT stage = mpcg_generate_kkt."""
    _build_bins(list(IIWA_DEMO_BINS.values()), force, verbose)
    return IIWA_DEMO_BINS


def build_multi_gpu(force: bool = False, verbose: bool = False) -> str:
    """The native multi-device driver: C++ host threads (one per GPU) over the C ABI + one RCCL all-gather."""
    _build_bins([MULTI_BIN], force, verbose)
    return MULTI_BIN


def build_utils_probe(force: bool = False, verbose: bool = False) -> str:
    """Instantiates store_block_bd / load_block_bd / gato_memcpy of include/gbd_pcg_compat/utils.cuh in a kernel."""
    _build_bins([UTILS_BIN], force, verbose)
    return UTILS_BIN


def build_all(force: bool = False, verbose: bool = False) -> str:
    """`make all`: the library and every call-site program."""
    make("all", force=force, verbose=verbose)
    return LIB_PATH


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
