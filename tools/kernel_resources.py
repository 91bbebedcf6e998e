#!/usr/bin/env python3
"""Register / LDS / scratch usage of every gfx950 kernel in the built library (from the code objects' metadata notes).
    python tools/kernel_resources.py [substring ...]      (default: every kernel; a substring filters the demangled names)"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, tmp):
    so = os.path.join(tmp, "lib.so")
    shutil.copy(lib, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def kernels(lib):
    tmp = tempfile.mkdtemp(prefix="kres_")
    out = []
    try:
        for co in code_objects(lib, tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                f = dict(re.findall(r"\.(\w+):\s+(\S+)", "  - .agpr_count:" + blk.split("\n  - .agpr_count:")[0]))
                if "name" not in f:
                    continue
                out.append(f)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, nm in zip(out, names):
        k["demangled"] = re.sub(r"\(.*", "", nm)
    return out


if __name__ == "__main__":
    lib = os.path.join(ROOT, "mpcgpu_amd", "libmpcg_hip.so")
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for a in sys.argv[1:]:
        if a.endswith(".so"):
            lib = a
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s} {'spill_v':>7s}")
    for k in kernels(lib):
        if pats and not any(p in k["demangled"] for p in pats):
            continue
        print(f"{k['demangled'][:70]:70s} {k.get('vgpr_count', '?'):>5s} {k.get('agpr_count', '?'):>5s} {k.get('sgpr_count', '?'):>5s} "
              f"{k.get('group_segment_fixed_size', '?'):>7s} {k.get('private_segment_fixed_size', '?'):>8s} {k.get('vgpr_spill_count', '?'):>7s}")
