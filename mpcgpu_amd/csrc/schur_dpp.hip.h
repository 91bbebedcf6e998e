// schur_dpp.hip.h — register-resident version of the Schur formation, gfx950.
//
// Same arithmetic, operation for operation, as form_schur_kernel / complete_ss_kernel in schur_kernels.hip.h
// (which restate include/pcg/linsys_setup.cuh) and therefore the same bits as the C oracle — but nothing
// lives in LDS.  FOUR knots per wavefront: a knot owns one 16-lane DPP row, lane r < 14 of the row holds ROW r
// of every 14x14 (14x7, 7x7) operand in registers.  The value another row needs from row t of an operand
// is a DPP `row_newbcast:t` source modifier on the multiply (verified on the chip: tools/_prof/dpp_probe.hip),
// so a 14x14x14 product is 196 x (v_mul_f32_dpp + v_add_f32) per wave for four knots, a Gauss-Jordan pivot
// step 28 x (mul + mul_dpp + sub + select): no LDS round trips, no barriers, ~3x fewer instructions per knot
// than the LDS version and all of them independent across the 14 columns.
//
// Contraction is OFF in this header: every a*b+c below is a rounded multiply followed by a rounded add,
// in the oracle's order (sequential over the contracted index, accumulator starting at +0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "schur_kernels.hip.h"

namespace mpcg {

// Two instantiations of the same source (schur_dpp_body.inc):
//   contraction OFF — the default: every a*b+c is a rounded multiply followed by a rounded add, in the oracle's order: the oracle's bits;
//   contraction ON  — option "schur_fma" = 1 (off by default): the compiler fuses them into v_fmac_f32 (+ DPP): half the arithmetic
//                     instructions in the products and the Gauss-Jordan updates.  Which of the two the reference's own build computes is not
//                     knowable here (its products live in the absent GLASS submodule; nvcc's default -fmad=true contracts a*b+c wherever
//                     the source has that shape, e.g. include/utils/matrix.cuh:146) — parity of this variant is tolerance-based
//                     (tests/test_gpu_schur.py), the bit-exact one stays the default.
#define SDPP_NS sdpp
#define SDPP_KERNEL(name) name
#define SDPP_FMA 0
#pragma clang fp contract(off)
#include "schur_dpp_body.inc"
#undef SDPP_NS
#undef SDPP_KERNEL
#undef SDPP_FMA
#define SDPP_NS sdpp_fma
#define SDPP_KERNEL(name) name##_fma
#define SDPP_FMA 1
#pragma clang fp contract(fast)
#include "schur_dpp_body.inc"
#undef SDPP_NS
#undef SDPP_KERNEL
#undef SDPP_FMA

}  // namespace mpcg
