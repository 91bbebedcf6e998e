"""Build libmpcg_hip.so (gfx950) in-tree with hipcc.  `python -m mpcgpu_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libmpcg_hip.so")
SOURCES = [os.path.join(_HERE, "csrc", "mpcg_capi.hip")]
DEPS = SOURCES + [os.path.join(_HERE, "csrc", "pcg_kernels.hip.h"), os.path.join(_ROOT, "include", "mpcg.h")]
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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
