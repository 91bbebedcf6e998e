"""Build libmpcg_hip.so (gfx950) in-tree with hipcc.  `python -m mpcgpu_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libmpcg_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", "mpcg_capi.hip")]
DEPS = SOURCES + [os.path.join(_HERE, "csrc", f) for f in ("pcg_kernels.hip.h", "pcg_lpk.hip.h", "pcg_lpk_cluster.hip.h", "pcg_rpl.hip.h", "schur_kernels.hip.h", "dpp_rows.hip.h", "schur_walk.hip.h", "block_solve.hip.h", "pcg_f64.hip.h", "ldl_host.hpp", "kkt_plant.hip.h", "iiwa14_model.inc")] + [
    os.path.join(_ROOT, "include", "mpcg.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function"]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        cmd = [HIPCC, *FLAGS, *SOURCES, "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


EXAMPLE_SRC = os.path.join(_ROOT, "examples", "sqp_pcg_callsite.cpp")
EXAMPLE_BIN = os.path.join(_ROOT, "examples", "sqp_pcg_callsite")


CHAIN_SRC = os.path.join(_ROOT, "examples", "sqp_linsys_chain.cpp")
CHAIN_BIN = os.path.join(_ROOT, "examples", "sqp_linsys_chain")


def build_chain_example(force: bool = False, verbose: bool = False) -> str:
    """C++ host program: form_schur_system -> pcg -> compute_dz as include/pcg/sqp.cuh:207-259 writes them."""
    deps = [CHAIN_SRC, LIB_PATH, os.path.join(_ROOT, "include", "gbd_pcg_compat", "gpu_pcg.cuh"),
            os.path.join(_ROOT, "include", "mpcgpu_compat", "linsys_steps.cuh")]
    if force or not os.path.exists(CHAIN_BIN) or any(os.path.getmtime(d) > os.path.getmtime(CHAIN_BIN) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(_ROOT, "include", "gbd_pcg_compat"),
               "-I" + os.path.join(_ROOT, "include", "mpcgpu_compat"), CHAIN_SRC, "-L" + _HERE, "-lmpcg_hip",
               "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", CHAIN_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CHAIN_BIN


CHAIN_BIN64 = os.path.join(_ROOT, "examples", "sqp_linsys_chain_f64")


def build_chain_example_f64(force: bool = False, verbose: bool = False) -> str:
    """The same chain compiled with -DUSE_DOUBLES (linsys_t = double): form_schur_system<double> -> pcg<double, n, N> -> compute_dz<double>."""
    deps = [CHAIN_SRC, LIB_PATH, os.path.join(_ROOT, "include", "gbd_pcg_compat", "gpu_pcg.cuh"),
            os.path.join(_ROOT, "include", "mpcgpu_compat", "linsys_steps.cuh")]
    if force or not os.path.exists(CHAIN_BIN64) or any(os.path.getmtime(d) > os.path.getmtime(CHAIN_BIN64) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-DUSE_DOUBLES", "-I" + os.path.join(_ROOT, "include", "gbd_pcg_compat"),
               "-I" + os.path.join(_ROOT, "include", "mpcgpu_compat"), CHAIN_SRC, "-L" + _HERE, "-lmpcg_hip",
               "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", CHAIN_BIN64]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CHAIN_BIN64


EXAMPLE_BIN64 = os.path.join(_ROOT, "examples", "sqp_pcg_callsite_f64")


def build_example_f64(force: bool = False, verbose: bool = False) -> str:
    """The same call site compiled with -DUSE_DOUBLES (linsys_t = double): pcg<double, n, N> over the shim."""
    deps = [EXAMPLE_SRC, LIB_PATH, os.path.join(_ROOT, "include", "gbd_pcg_compat", "gpu_pcg.cuh")]
    if force or not os.path.exists(EXAMPLE_BIN64) or any(os.path.getmtime(d) > os.path.getmtime(EXAMPLE_BIN64) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-DUSE_DOUBLES", "-I" + os.path.join(_ROOT, "include", "gbd_pcg_compat"),
               EXAMPLE_SRC, "-L" + _HERE, "-lmpcg_hip", "-lpthread", "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", EXAMPLE_BIN64]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return EXAMPLE_BIN64


def build_example(force: bool = False, verbose: bool = False) -> str:
    """C++ host program: the reference's PCG call site over the shim headers + the C ABI."""
    deps = [EXAMPLE_SRC, LIB_PATH, os.path.join(_ROOT, "include", "gbd_pcg_compat", "gpu_pcg.cuh")]
    if force or not os.path.exists(EXAMPLE_BIN) or any(os.path.getmtime(d) > os.path.getmtime(EXAMPLE_BIN) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(_ROOT, "include", "gbd_pcg_compat"),
               EXAMPLE_SRC, "-L" + _HERE, "-lmpcg_hip", "-lpthread", "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", EXAMPLE_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return EXAMPLE_BIN


DEMO_SRC = os.path.join(_ROOT, "examples", "mpcsim_shim_demo.cpp")
DEMO_BINS = {1: os.path.join(_ROOT, "examples", "mpcsim_shim_demo_pcg"), 0: os.path.join(_ROOT, "examples", "mpcsim_shim_demo_qdldl")}


def build_mpcsim_demo(force: bool = False, verbose: bool = False):
    """simulateMPC -> sqpSolvePcg | sqpSolveQdldl over this repo's include/mpcsim.cuh, include/pcg/sqp.cuh, include/qdldl/sqp.cuh:
    the same source compiled with -DLINSYS_SOLVE=1 and =0 (the reference's compile-time solver switch, include/mpcsim.cuh:21-25)."""
    inc = os.path.join(_ROOT, "include")
    deps = [DEMO_SRC, LIB_PATH] + [os.path.join(inc, f) for f in ("mpcsim.cuh", "pcg/sqp.cuh", "qdldl/sqp.cuh", "mpcgpu_compat/sqp_stages.cuh",
                                                                  "mpcgpu_compat/linsys_steps.cuh", "gbd_pcg_compat/gpu_pcg.cuh")]
    for sel, exe in DEMO_BINS.items():
        if force or not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-DLINSYS_SOLVE={sel}", "-I" + inc, DEMO_SRC, "-L" + _HERE, "-lmpcg_hip",
                   "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", exe]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return DEMO_BINS


IIWA_DEMO_SRC = os.path.join(_ROOT, "examples", "mpcsim_iiwa_demo.cpp")
IIWA_DEMO_BINS = {1: os.path.join(_ROOT, "examples", "mpcsim_iiwa_demo_pcg"), 0: os.path.join(_ROOT, "examples", "mpcsim_iiwa_demo_qdldl")}


def build_iiwa_demo(force: bool = False, verbose: bool = False):
    """simulateMPC over the shim headers on a real window of the reference trajectory, KKT stage = the library's mpcg_generate_kkt;
    -DLINSYS_SOLVE=1 and =0."""
    inc = os.path.join(_ROOT, "include")
    deps = [IIWA_DEMO_SRC, LIB_PATH] + [os.path.join(inc, f) for f in ("mpcsim.cuh", "pcg/sqp.cuh", "qdldl/sqp.cuh", "mpcgpu_compat/sqp_stages.cuh")]
    for sel, exe in IIWA_DEMO_BINS.items():
        if force or not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-DLINSYS_SOLVE={sel}", "-I" + inc, IIWA_DEMO_SRC, "-L" + _HERE, "-lmpcg_hip",
                   "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", exe]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return IIWA_DEMO_BINS


MULTI_SRC = os.path.join(_ROOT, "examples", "multi_gpu_pcg.cpp")
MULTI_BIN = os.path.join(_ROOT, "examples", "multi_gpu_pcg")


def build_multi_gpu(force: bool = False, verbose: bool = False) -> str:
    """The native multi-device driver: C++ host threads (one per GPU) over the C ABI + one RCCL all-gather."""
    deps = [MULTI_SRC, LIB_PATH, os.path.join(_ROOT, "include", "mpcg.h")]
    if force or not os.path.exists(MULTI_BIN) or any(os.path.getmtime(d) > os.path.getmtime(MULTI_BIN) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", MULTI_SRC, "-L" + _HERE, "-lmpcg_hip", "-lrccl", "-lpthread",
               "-Wl,-rpath,$ORIGIN/../mpcgpu_amd", "-o", MULTI_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return MULTI_BIN


UTILS_SRC = os.path.join(_ROOT, "examples", "bd_utils_probe.cpp")
UTILS_BIN = os.path.join(_ROOT, "examples", "bd_utils_probe")


def build_utils_probe(force: bool = False, verbose: bool = False) -> str:
    """Instantiates store_block_bd / load_block_bd / gato_memcpy of include/gbd_pcg_compat/utils.cuh in a kernel."""
    deps = [UTILS_SRC, os.path.join(_ROOT, "include", "gbd_pcg_compat", "utils.cuh")]
    if force or not os.path.exists(UTILS_BIN) or any(os.path.getmtime(d) > os.path.getmtime(UTILS_BIN) for d in deps):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(_ROOT, "include", "gbd_pcg_compat"), UTILS_SRC, "-o", UTILS_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return UTILS_BIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_example(force="--force" in sys.argv, verbose=True))
    print(build_chain_example(force="--force" in sys.argv, verbose=True))
    print(build_mpcsim_demo(force="--force" in sys.argv, verbose=True))
    print(build_utils_probe(force="--force" in sys.argv, verbose=True))
    print(build_multi_gpu(force="--force" in sys.argv, verbose=True))
    print(build_iiwa_demo(force="--force" in sys.argv, verbose=True))
