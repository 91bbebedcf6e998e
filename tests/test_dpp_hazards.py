"""CPU: static check of the built gfx950 code for the DPP read-after-VALU-write hazard (tools/check_dpp_hazards.py).
Several kernels issue v_fmac_f32_dpp / v_fmac_f64_dpp from inline assembly, which the compiler's hazard recogniser cannot see into:
a DPP source written fewer than two wait states earlier is read stale — silently.  The whole library is disassembled and every DPP
instruction checked; the checker itself is checked on hand-written snippets first."""
import importlib.util
import os

import pytest

from conftest import ROOT

spec = importlib.util.spec_from_file_location("check_dpp_hazards", os.path.join(ROOT, "tools", "check_dpp_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def _snippet(body):
    return "0000000000001000 <kernel>:\n" + "\n".join("\t" + l for l in body) + "\n"


@pytest.mark.parametrize("body,bad", [
    (["v_mul_f32_e32 v1, v2, v3", "v_fmac_f32_dpp v4, v1, v5 row_newbcast:0 row_mask:0xf bank_mask:0xf"], True),
    (["v_mul_f32_e32 v1, v2, v3", "v_add_f32_e32 v9, v2, v3", "v_fmac_f32_dpp v4, v1, v5 row_newbcast:0 row_mask:0xf bank_mask:0xf"], True),
    (["v_mul_f32_e32 v1, v2, v3", "s_nop 0", "v_mov_b32_dpp v4, v1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"], True),
    (["v_mul_f32_e32 v1, v2, v3", "s_nop 1", "v_fmac_f32_dpp v4, v1, v5 row_newbcast:0 row_mask:0xf bank_mask:0xf"], False),
    (["v_mul_f32_e32 v1, v2, v3", "v_add_f32_e32 v9, v2, v3", "v_add_f32_e32 v8, v2, v3", "v_mul_f32_dpp v4, v1, v5 row_newbcast:3 row_mask:0xf bank_mask:0xf"], False),
    (["v_mul_f32_e32 v1, v2, v3", "v_fmac_f32_dpp v4, v6, v1 row_newbcast:0 row_mask:0xf bank_mask:0xf"], False),        # src1 is not read through DPP
    (["v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]", "v_fmac_f64_dpp v[0:1], v[10:11], v[4:5] row_newbcast:2 row_mask:0xf bank_mask:0xf"], True),
    (["global_load_dword v1, v[2:3], off", "v_fmac_f32_dpp v4, v1, v5 row_newbcast:0 row_mask:0xf bank_mask:0xf"], False),  # (memory results wait on vmcnt, not on wait states)
])
def test_checker_on_snippets(body, bad):
    findings, n_dpp, _ = chk.check(_snippet(body))
    assert n_dpp == 1 and bool(findings) == bad, findings


@pytest.mark.skipif(not os.path.exists(chk.OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_library_has_no_dpp_read_after_write_hazard():
    from mpcgpu_amd import _lib
    findings, n_dpp, n_kern = chk.check(chk.disassemble(_lib.LIB_PATH))
    assert n_kern > 30 and n_dpp > 5000          # (every kernel family was looked at)
    assert not findings, findings[:10]
