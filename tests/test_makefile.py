"""The boundary builds without Python (the reference: one `make`, /root/reference/Makefile:1-20): the top-level Makefile
names the library and every C++ call-site program, compiles each translation unit of csrc/ for gfx950 with the flags INTEGRATION.md §2
states, and mpcgpu_amd/build.py is only its front end."""
import glob
import os
import subprocess

from conftest import ROOT


def _dry(*targets):
    r = subprocess.run(["make", "-C", ROOT, "-n", "-B", *targets], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_make_all_names_the_library_and_every_example():
    out = _dry("all")
    units = sorted(glob.glob(os.path.join(ROOT, "mpcgpu_amd", "csrc", "*.hip")))
    assert units
    for u in units:
        rel = os.path.relpath(u, ROOT)
        line = [ln for ln in out.splitlines() if f"-c {rel}" in ln]
        assert len(line) == 1, rel
        for flag in ("hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"):
            assert flag in line[0], (rel, flag)
    link = [ln for ln in out.splitlines() if "-shared" in ln and "-o mpcgpu_amd/libmpcg_hip.so" in ln]
    assert len(link) == 1
    from mpcgpu_amd import build
    bins = [build.EXAMPLE_BIN, build.EXAMPLE_BIN64, build.EXAMPLE_BIN64_N128, build.CHAIN_BIN, build.CHAIN_BIN64, *build.DEMO_BINS.values(), *build.IIWA_DEMO_BINS.values(),
            build.MULTI_BIN, build.UTILS_BIN]
    for b in bins:
        assert f"-o {os.path.relpath(b, ROOT)}" in out, b
    assert "-DLINSYS_SOLVE=0" in out and "-DLINSYS_SOLVE=1" in out and "-DUSE_DOUBLES" in out and "-lrccl" in out


def test_build_py_is_a_front_end_of_the_makefile():
    src = open(os.path.join(ROOT, "mpcgpu_amd", "build.py")).read()
    assert '"make"' in src and "check_call([HIPCC" not in src
    out = _dry("oracle")
    assert "libmpcg_oracle.so" in out
