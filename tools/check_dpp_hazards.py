#!/usr/bin/env python3
"""Static check of the built library's gfx950 code for the DPP read-after-VALU-write hazard.

A DPP instruction reads its src0 through the cross-lane network; a VGPR written by a VALU instruction needs TWO wait states before a DPP
instruction may read it (otherwise the read returns the old value — silently).  hipcc's hazard recogniser inserts the s_nop itself for
the instructions it emits, but it cannot see inside inline assembly, and several kernels here issue `v_fmac_f32_dpp` / `v_fmac_f64_dpp`
from assembler text (pcg_rpl.hip.h, schur_dpp_body.inc).  This script disassembles every kernel of libmpcg_hip.so and reports each
DPP instruction whose src0 register is written by a VALU instruction fewer than two wait states earlier in straight-line code
(s_nop N counts N + 1 wait states, every other instruction one; a label resets the window — a hazard across a branch is not looked for).
    python tools/check_dpp_hazards.py [path/to/libmpcg_hip.so]      exit code 1 if anything is found
"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
DPP_CTRL = re.compile(r"\b(row_newbcast|row_shl|row_shr|row_ror|row_share|row_xmask|row_mirror|row_half_mirror|row_bcast|quad_perm|wave_shl|wave_shr|wave_rol|wave_ror)\b")


def regs(op):
    """VGPR numbers named by an operand like v7, v[4:5], -v3, |v2|."""
    m = re.search(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", op)
    return {int(m.group(1))} if m else set()


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="dpphaz_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=tmp)
        cos = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        if not cos:
            raise RuntimeError("no amdgcn code object found in " + lib)
        # one code object per translation unit of the library (mpcg_pcg / mpcg_producers / mpcg_plant / mpcg_ldl): all of them
        return "\n".join(subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout for co in sorted(cos))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(text):
    findings, n_dpp, n_kern = [], 0, 0
    func = None
    window = []                      # [(wait_states_ago_start, written_vgprs, text)] of recent VALU writes
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            window = []
            n_kern += 1
            continue
        ins = line.split("//")[0].strip()
        if not ins or ins.endswith(":"):
            if ins.endswith(":"):
                window = []
            continue
        parts = ins.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        ws = 1
        if op == "s_nop":
            ws = int(rest.strip(), 0) + 1
        if op.startswith("v_") and DPP_CTRL.search(rest):
            n_dpp += 1
            ops = [o.strip() for o in DPP_CTRL.split(rest)[0].split(",")]
            src0 = regs(ops[1]) if len(ops) > 1 else set()
            for age, written, wtxt in window:
                if age < 2 and written & src0:
                    findings.append((func, wtxt, ins))
        # age the window by this instruction's wait states, then record what it writes
        window = [(age + ws, w, t) for age, w, t in window if age + ws < 2]
        if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
            dst = rest.split(",")[0]
            w = regs(dst)
            if w:
                window.append((0, w, ins))
    return findings, n_dpp, n_kern


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mpcgpu_amd", "libmpcg_hip.so")
    findings, n_dpp, n_kern = check(disassemble(lib))
    print(f"{lib}: {n_kern} functions, {n_dpp} DPP instructions, {len(findings)} read-after-write hazards")
    for f, w, d in findings[:40]:
        print(f"  {f}\n      {w}\n      {d}")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
