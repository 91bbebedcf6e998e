#!/usr/bin/env python3
"""Disassembly of one kernel of the built library + instruction statistics of its hottest loop (the longest backward branch span).
   python tools/_prof/disasm_kernel.py <substring of the demangled name> [lib.so] [--dump file]"""
import os, re, shutil, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1]
lib = next((a for a in sys.argv[2:] if a.endswith(".so")), os.path.join(ROOT, "mpcgpu_amd", "libmpcg_hip.so"))
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
tmp = tempfile.mkdtemp(prefix="dis_")
try:
    so = os.path.join(tmp, "lib.so"); shutil.copy(lib, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
    text = ""
    for co in sorted(f for f in os.listdir(tmp) if "amdgcn" in f):
        text += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--demangle", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
finally:
    shutil.rmtree(tmp, ignore_errors=True)
funcs = re.split(r"\n(?=[0-9a-f]+ <)", text)
for fn in funcs:
    head = fn.split("\n", 1)[0]
    if pat not in head or ">:" not in head:
        continue
    lines = [l for l in fn.split("\n")[1:] if l.strip()]
    ins = []
    for l in lines:
        m = re.match(r"\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    print(head, len(ins), "instructions")
    if dump:
        open(dump, "w").write(fn)
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    # backward branches: loops
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            # the operand is a signed 16-bit dword offset from the next instruction
            m = re.match(r"(\d+)", args)
            if m:
                off = int(m.group(1))
                off = off - 65536 if off >= 32768 else off
                tgt = a + 4 + 4 * off
                if tgt in addr_index and addr_index[tgt] < i:
                    loops.append((addr_index[tgt], i))
    loops.sort(key=lambda t: t[0] - t[1])
    for (s, e) in loops[:3]:
        body = ins[s:e + 1]
        c = collections.Counter()
        for _, op, args in body:
            if op.startswith("v_pk_fma") or op.startswith("v_pk_mul"): c["pk_fma/mul"] += 1
            elif op.startswith("v_pk_"): c["pk_other"] += 1
            elif op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mul_f32") or op.startswith("v_add_f32") or op.startswith("v_sub_f32"): c["v_fp32" + ("_dpp" if "dpp" in op or "quad_perm" in args or "row_" in args else "")] += 1
            elif op.startswith("v_mov") and ("quad_perm" in args or "row_" in args): c["v_mov_dpp"] += 1
            elif op.startswith("v_cndmask"): c["v_cndmask"] += 1
            elif op.startswith("v_"): c["v_other"] += 1
            elif op.startswith("ds_read") or op.startswith("ds_load"): c["ds_read"] += 1
            elif op.startswith("ds_write") or op.startswith("ds_store"): c["ds_write"] += 1
            elif op.startswith("scratch_") or (op.startswith("buffer_") and "off" in args): c[op.split("_")[0] + "_" + op.split("_")[1]] += 1
            elif op.startswith("s_nop"): c["s_nop"] += 1
            elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
            elif op.startswith("s_barrier"): c["s_barrier"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            else: c[op] += 1
        print(f"  loop [{s}..{e}] {e - s + 1} instructions:", dict(sorted(c.items(), key=lambda kv: -kv[1])))
