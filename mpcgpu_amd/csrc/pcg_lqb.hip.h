// pcg_lqb.hip.h — "lane QUAD per knot, BOTH matrices in every wavefront" PCG kernel for gfx950 (round 6), fp32, knot_points <= 128.
//
// What the lane-pair kernel (pcg_lpk.hip.h) leaves on the table.  It gives a matrix to a wavefront: waves 0-3 hold S, waves 4-7 Pinv, one of
// each per SIMD — and classic PCG runs the two passes of an iteration one after the other, so at any time ONE wavefront per SIMD is working
// and the other waits at a barrier with half of the CU's register file.  Every configuration of that kernel (128 knots x 1, 64 x 2, 32 x 4
// trajectories per CU) ends up there: 15-21 G knot-iterations/s, VALU pipe 52 % busy (profiles/r05e_pmc.json), and with one wavefront issuing,
// every instruction — a DPP move, an add, a packed FMA — costs the same ~4.6 clocks (profiles/r04_dpp_rate.txt).
//
// Here every wavefront holds a QUARTER of both matrices for its knots, so all eight wavefronts run the S pass, then all eight run the Pinv pass:
// two working wavefronts per SIMD, which fill each other's LDS / DPP / dependent-issue gaps and co-issue the non-FMA instructions.
//
// Mapping.  Knot k owns FOUR adjacent lanes (h, g) of one wavefront (16 knots per wavefront, 128 knots = 8 wavefronts = the CU's 512 KB register
// file, as before).  The 14 indices of a knot are split into two PIECES of seven: piece 0 = entries 0..5 and 6, piece 1 = entries 8..13 and 7 —
// as float2 row pairs: pairs 4s, 4s+1, 4s+2 and element s of pair 3.  Lane (h, g) holds, for BOTH S and Pinv, the 7 x 7 sub-blocks
// D_k[piece g, piece h] and L_k[piece g, piece h] (the lower block triangle: D_k = M[k, diag], L_k = M[k, left]): 4 x 49 = 196 VGPRs.  A 7 x 7
// sub-block as packed operands: three ROW pairs x seven columns (21 float2), the seventh row as three COLUMN pairs (3 float2) + one scalar.
//   direct      ypart[piece g] = D[g,h] x_k[piece h] + L[g,h] x_{k-1}[piece h]        42 + 6 packed FMAs + 2 FMAs
//   transposed  zpart[piece h] = L[g,h]^T x_k[piece g]                                 21 + 3 packed FMAs + 1 FMA (+ 7 adds)
// 75 FMA-class instructions per lane and pass where the lane-pair kernel has 147 — in twice as many lanes.  The column-piece partial sums of y
// are merged across h, the row-piece partial sums of z across g — and a lane needs only ONE of the two merged results: lanes h == g publish
// y_k[piece g] (to US / RT), lanes h != g publish z[piece h] (L_k^T x_k, which belongs to knot k-1; to ZS / ZP): one quad_perm DPP add per value.
//
// "The reader rebuilds" (pcg_lpk.hip.h) carries over, with the quad sharing the work: lane (h, g) CARRIES piece h of knot k - g of p and of r in
// registers (7 + 7), fetches the two published vectors for that piece only (8 LDS loads), forms the updated piece with the very operations every
// other holder uses, and the quad hands the three pieces a pass needs — x_k[piece h], x_{k-1}[piece h], x_k[piece g] — around by DPP quad_perm.
// p and r never exist in LDS inside the loop; lambda (piece h of knot k - g) lives in registers too.  Two barriers per iteration.
// Inner product without the assembled vector, from the partial sums themselves (no weights, no duplicates): x^T M x = sum over lanes of
// x_k[piece g] . (ypart + its L part) — the coupling term x_{k-1}^T (L^T x_k) taken as x_k^T (L x_{k-1}) from the direct product.
// Block-Jacobi (Pinv without off-diagonal blocks) is a build of its own, PC3 = false: no L sub-block of Pinv in registers, no z, a shorter epilogue.
// Measured (steady clocks, tools/_prof/lqb_ab.py; lane-pair kernel in brackets): N = 128 SS 1.58 (1.62) us per iteration of one trajectory, 6.31 (6.54) per
// 1024; block-Jacobi 1.29 (1.34); N = 64, two independent workgroups per CU: 1.03 (1.44), 3.02 (3.71); DESIGN.md §3.2b for where an iteration goes.
//
// Reads only the left + diagonal block columns (include/mpcg.h, BLOCK SYMMETRY), like the lane-pair kernel; same PCG, same exit rule, same
// outputs (include/pcg/sqp.cuh:137-150); another summation order, so results agree with it to the fp32 band, not bit for bit.
#pragma once
#include "pcg_lpk.hip.h"
#include "pcg_rpl.hip.h"

namespace mpcg {

// LDS layout of one vector: pair-major as LpkLds (V[q][slot] = entries (2q, 2q+1) of knot slot - 1), with KN = 5 (mod 8) knot slots per row pair and
// vectors padded to VS = 16 (mod 32) floats, for the two access shapes of the loop (MI355X_MICROARCH.md, LDS):
//   * operand loads, ds_read_b64: 32-lane groups over 64 banks — a group is eight knots x (g, h): the h = 0 lanes read 2 (kl - g) + const (18 banks, the two g
//     lanes of neighbouring knots share addresses), the h = 1 lanes 4 row pairs = 8 KN = 40 (mod 64) dwords further: disjoint;
//   * publishing stores, ds_write_b64: 16-lane groups over a 32-BANK modulus — four knots x four lanes, to TWO vectors at once (lanes h == g to T, lanes
//     h != g to Z = T + VS): T pairs rp at 2 kl, T pairs 4 + rp at 8 + 2 kl, Z pairs rp at 16 + 2 kl, Z pairs 4 + rp at 24 + 2 kl — the 32 banks exactly once.
//     (The first layout, KN = 4 (mod 8) as in the lane-pair kernel, put the two halves of each vector on the SAME banks for a store: two-way conflicts in
//     every publishing store, 20 % of the kernel's LDS cycles — profiles/r06d_pmc.json, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)
template <int NMAXQ> struct LqbLds {
    static constexpr int NMAX = NMAXQ, NW = NMAXQ / 16;
    static_assert(NMAXQ == 32 || NMAXQ == 64 || NMAXQ == 128, "knots per workgroup");
    static constexpr int KN = NMAX + 5;
    static_assert(KN % 8 == 5, "row pairs q and q + 4: 40 (mod 64) dwords apart for the loads, 8 (mod 32) for the stores");
    static constexpr int VS = 7 * KN * 2 + 10;
    static_assert(VS % 32 == 16 && VS % 2 == 0, "T and Z = T + VS must sit 16 banks apart (8-byte aligned)");
    // P0, R0: staging of lambda0 / gamma, at the end p and r for d_p / d_r | US, ZS: what the S pass publishes | RT, ZP: the Pinv pass |
    // LAM: lambda at the end | RED: wave partials of the two inner products | TILE2: second load tile of every wavefront (the first aliases the vectors)
    static constexpr int P0 = 0, R0 = VS, US = 2 * VS, ZS = 3 * VS, RT = 4 * VS, ZP = 5 * VS, LAM = 6 * VS, RED = 7 * VS, TILE2 = RED + 2 * NW,
                         TOTAL = TILE2 + NW * LPK_TILE_FLOATS;
    static_assert(NW * LPK_TILE_FLOATS <= RED, "the first load tiles alias the iterate vectors");
    static_assert(TILE2 % 4 == 0, "tiles are 16-byte aligned");
    __host__ __device__ static constexpr int at(int k, int i) { return 2 * ((i >> 1) * KN + k + 1) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lqb_lds_floats(int nmax) {
    return nmax == 32 ? (size_t)LqbLds<32>::TOTAL : nmax == 64 ? (size_t)LqbLds<64>::TOTAL : (size_t)LqbLds<128>::TOTAL;
}

// a 7 x 7 sub-block in packed form: A[rp][j] = rows (2 (4g + rp), + 1) of column c_j; B[cp] = row 6 + g of columns (c_2cp, c_2cp+1); e = row
// 6 + g of column c_6 (c_j = 8 h + j for j < 6, c_6 = 6 + h)
struct LqbSub { f2 A[3][7]; f2 B[3]; float e; };
// seven entries of a vector: a piece (pairs 4s .. 4s + 2 and element s of pair 3)
struct LqbPiece { f2 p[3]; float s; };

typedef __attribute__((address_space(3))) const volatile float lds_cv_f1;
__device__ __forceinline__ float lds_ld32(const float* p) { return *(lds_cv_f1*)(p); }

template <int QP> __device__ __forceinline__ float lqb_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), QP, 0xF, 0xF, true));
}
template <int QP> __device__ __forceinline__ LqbPiece lqb_quad(const LqbPiece& v) {
    LqbPiece o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.p[i] = f2{lqb_quad<QP>(v.p[i].x), lqb_quad<QP>(v.p[i].y)};
        // (an opaque 64-bit value from here on: left as two scalars, a broadcast of the ODD element is compiled as a copy into an even register +
        //  a low-half broadcast — three v_mov per piece; as the high half of a register pair it is the instruction's op_sel modifier)
        asm("" : "+v"(o.p[i]));
    }
    o.s = lqb_quad<QP>(v.s);
    return o;
}
// lane index inside the quad: q = 2 g + h
constexpr int LQB_QP_XH = 0x44;     // quad_perm [0,1,0,1]: from the g = 0 lane with the same h — x_k[piece h]
constexpr int LQB_QP_XM = 0xEE;     // quad_perm [2,3,2,3]: from the g = 1 lane with the same h — x_{k-1}[piece h]
constexpr int LQB_QP_XG = 0x50;     // quad_perm [0,0,1,1]: from the g = 0 lane whose h is this lane's g — x_k[piece g]
constexpr int LQB_QP_MERGE = 0x8D;  // quad_perm [1,3,0,2]: the lane that holds the other partial sum of what this lane publishes

// ---- matrix registers through the LDS stage of pcg_lpk.hip.h (lpk_load_blocks_lds): the wavefront streams whole blocks with buffer_load ... lds
// into two 6,272-byte tiles (eight knots of one block column each, two rounds in flight) and the lanes pick their sub-blocks out of LDS.
// Round r = 0..7: matrix r >> 2 (S, Pinv), block column (r >> 1) & 1 (L, D), knots kfirst + 8 (r & 1) .. + 7 — gathered by the 32 lanes of that half.
__device__ __forceinline__ void lqb_load_blocks_lds(rsrc_t MS, rsrc_t MP, int kfirst, int kend, bool p3, int lane, float* tileA, float* tileB,
                                                    LqbSub& SD, LqbSub& SL, LqbSub& PD, LqbSub& PL) {
    constexpr uint32_t PB = 16u, BLKB = (uint32_t)(NS * NS) * 4u, ROWB = (uint32_t)ROWF * 4u;
    typedef __attribute__((address_space(3))) char lds_c;
    const int h = lane & 1, g = (lane >> 1) & 1, kt = (lane >> 2) & 7, myhalf = lane >> 5;
    uint32_t goff[7];
    int pk[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int p = lane + 64 * i;
        pk[i] = p / 49;
        goff[i] = (uint32_t)pk[i] * ROWB + (uint32_t)(p - 49 * pk[i]) * PB;
    }
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto issue = [&](int r) {
        const int mat = r >> 2, blk = (r >> 1) & 1;
        const int kr = kfirst + 8 * (r & 1);
        const int lo = (blk == 0 && kr == 0) ? 1 : 0;
        int hi = kend - kr;
        hi = hi < 0 ? 0 : (hi > 8 ? 8 : hi);
        if (blk == 0 && mat == 1 && !p3) hi = 0;               // block-Jacobi: Pinv has no off-diagonal blocks
        const uint32_t sbase = (uint32_t)kr * ROWB + (uint32_t)blk * BLKB;
        const bool all = lo == 0 && hi == 8;
        lds_c* tile = (lds_c*)((r & 1) ? tileB : tileA);
        const rsrc_t M = mat ? MP : MS;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const uint32_t vo = (all || (pk[i] >= lo && pk[i] < hi)) ? goff[i] : OOB_OFF;
            if (i < 6 || lane < 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(M, (__attribute__((address_space(3))) void*)(tile + 64 * i * 16), 16, (int)vo, (int)sbase, 0, 0);
        }
    };
    auto gather = [&](LqbSub& X, int r) {
        if (myhalf != (r & 1)) return;
        const int mat = r >> 2, blk = (r >> 1) & 1;
        const float* tile = (r & 1) ? tileB : tileA;
        const int gk = kfirst + 8 * (r & 1) + kt;
        // (a knot outside the horizon / the absent L_0 / block-Jacobi's L: zeros — an LDS-destination load leaves LDS untouched for such a lane)
        if (!(gk < kend && (blk == 1 || (gk > 0 && (mat == 0 || p3))))) {
#pragma unroll
            for (int rp = 0; rp < 3; ++rp)
#pragma unroll
                for (int j = 0; j < 7; ++j) X.A[rp][j] = f2{0.f, 0.f};
#pragma unroll
            for (int cp = 0; cp < 3; ++cp) X.B[cp] = f2{0.f, 0.f};
            X.e = 0.f;
            return;
        }
        const float* blkp = tile + kt * (NS * NS);
        const int cb = NS * 8 * h, c6 = NS * (6 + h);           // float offsets of columns c_0 and c_6
#pragma unroll
        for (int rp = 0; rp < 3; ++rp)
#pragma unroll
            for (int j = 0; j < 7; ++j) X.A[rp][j] = lds_ld64(blkp + (j < 6 ? cb + NS * j : c6) + 2 * (4 * g + rp));
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) X.B[cp] = f2{lds_ld32(blkp + cb + NS * (2 * cp) + 6 + g), lds_ld32(blkp + cb + NS * (2 * cp + 1) + 6 + g)};
        X.e = lds_ld32(blkp + c6 + 6 + g);
    };
    // (Two rounds in flight.  A third tile per wavefront — three in flight, 50 KB more LDS at 128 knots — changes nothing where the chip loads in lock-step
    //  (a full batch of equal iteration counts: the stream is bound by HBM then, 4.3-4.9 TB/s) and costs 1 % at 64 knots; streaming L_k and D_k of the same
    //  eight knots in consecutive rounds (1,568-byte instead of 784-byte runs in flight together) changes nothing either: tools/_prof/lqb_fixed.py, lqb_ab.py.)
    issue(0);
    issue(1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r < 7) __builtin_amdgcn_s_waitcnt(0x0F77);      // vmcnt(7): everything but the younger round has landed
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        wave_sync();
        gather(r < 2 ? SL : r < 4 ? SD : r < 6 ? PL : PD, r);
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the tile has been read before the next round is sent into it
        wave_sync();
        if (r + 2 < 8) issue(r + 2);
    }
}

#ifndef LQB_NPARK
#define LQB_NPARK 6      // pairs of each matrix's diagonal sub-block parked in LDS (lane-private slots inside the wavefront's second load tile, idle after the load)
#endif
#ifndef LQB_NPARK_J
#define LQB_NPARK_J 0    // the same for the block-Jacobi build: none — it carries no off-diagonal Pinv sub-block (214 VGPRs); 0 / 3 / 6 parked: 1.287 / 1.295 / 1.319 us per iteration at N = 128
#endif
static_assert(LQB_NPARK >= 0 && LQB_NPARK <= 6 && LQB_NPARK_J >= 0 && LQB_NPARK_J <= 6, "2 x NPARK x 512 B must fit the wavefront's 6,272-byte tile");
// x + y of a pair as ONE scalar add (left to the compiler, neighbouring horizontal sums are "vectorised": three moves + a packed add per two)
__device__ __forceinline__ float lqb_hsum(f2 v) {
    float r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
    return r;
}
// The epilogue of a pass as ONE block of assembler text, because its two dependent chains have to be interleaved by hand (the compiler runs them
// one after the other and fills the DPP wait states with s_nop):
//   * the wavefront sum of `part` (valid in LANE 63 afterwards): four row_shr steps (an inclusive scan inside each 16-lane row: lane 15 of a row
//     ends with the row's sum), then row_bcast:15 into rows 1, 3 and row_bcast:31 into rows 2, 3 — written as `v_add_f32_dpp x, x, x` with a
//     row mask: the lanes of a masked row keep the destination, which IS x, so a step is one instruction (the compiler needs a zeroed
//     temporary, a move and an add for it) and there is no v_readlane -> SGPR -> VALU round trip (pcg_lpk.hip.h's fold: 145 clocks per pass here);
//   * which partial sum this lane publishes: keep = diag ? y : z, send = diag ? z : y (v_cndmask on the 64-bit lane mask `diag`), two selects
//     between consecutive steps of the sum — exactly the two wait states a DPP read of a just-written VGPR needs;
//   * fin = keep + (send of quad lane [1,3,0,2]): the other partial sum of what this lane publishes.
// 27 VALU instructions, no s_nop.  (DPP read-after-VALU-write hazards inside assembler text are invisible to the compiler:
// tools/check_dpp_hazards.py checks the built code — every DPP source here was written at least two instructions earlier.)
__device__ __forceinline__ void lqb_epilogue(float& part, LqbPiece& fin, const LqbPiece& y, const LqbPiece& z, unsigned long long diag) {
    float k0, k1, k2, k3, k4, k5, k6, s0, s1, s2, s3, s4, s5, s6;
    asm volatile(
        "v_cndmask_b32 %8, %22, %15, %29\n\t"      // keep0 = diag ? y0 : z0
        "v_cndmask_b32 %1, %15, %22, %29\n\t"      // send0 = diag ? z0 : y0
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32 %9, %23, %16, %29\n\t"
        "v_cndmask_b32 %2, %16, %23, %29\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32 %10, %24, %17, %29\n\t"
        "v_cndmask_b32 %3, %17, %24, %29\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32 %11, %25, %18, %29\n\t"
        "v_cndmask_b32 %4, %18, %25, %29\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_cndmask_b32 %12, %26, %19, %29\n\t"
        "v_cndmask_b32 %5, %19, %26, %29\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_cndmask_b32 %13, %27, %20, %29\n\t"
        "v_cndmask_b32 %6, %20, %27, %29\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_cndmask_b32 %14, %28, %21, %29\n\t"
        "v_cndmask_b32 %7, %21, %28, %29\n\t"
        "v_add_f32_dpp %8, %1, %8 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %9, %2, %9 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %10, %3, %10 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %11, %4, %11 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %12, %5, %12 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %13, %6, %13 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %14, %7, %14 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(part),                                                                                    // 0
          "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(s4), "=&v"(s5), "=&v"(s6),                  // 1..7
          "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3), "=&v"(k4), "=&v"(k5), "=&v"(k6)                   // 8..14
        : "v"(y.p[0].x), "v"(y.p[0].y), "v"(y.p[1].x), "v"(y.p[1].y), "v"(y.p[2].x), "v"(y.p[2].y), "v"(y.s),      // 15..21
          "v"(z.p[0].x), "v"(z.p[0].y), "v"(z.p[1].x), "v"(z.p[1].y), "v"(z.p[2].x), "v"(z.p[2].y), "v"(z.s),      // 22..28
          "s"(diag));                                                                                    // 29
    fin.p[0] = f2{k0, k1}; fin.p[1] = f2{k2, k3}; fin.p[2] = f2{k4, k5}; fin.s = k6;
}
// The same for a block-Jacobi Pinv pass (no off-diagonal blocks: nothing for knot k-1, z stays zero): the wavefront sum interleaved with the seven merge
// adds of y — the lanes h == g publish, the others have nothing to select or send.  16 instructions instead of 27.
__device__ __forceinline__ void lqb_epilogue_y(float& part, LqbPiece& fin, const LqbPiece& y) {
    float k0, k1, k2, k3, k4, k5, k6;
    asm volatile(
        "v_add_f32_dpp %1, %8, %8 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %9, %9 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %10, %10 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %4, %11, %11 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %5, %12, %12 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %6, %13, %13 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %7, %14, %14 quad_perm:[1,3,0,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(part), "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3), "=&v"(k4), "=&v"(k5), "=&v"(k6)
        : "v"(y.p[0].x), "v"(y.p[0].y), "v"(y.p[1].x), "v"(y.p[1].y), "v"(y.p[2].x), "v"(y.p[2].y), "v"(y.s));
    fin.p[0] = f2{k0, k1}; fin.p[1] = f2{k2, k3}; fin.p[2] = f2{k4, k5}; fin.s = k6;
}

// (alpha = eta / v and beta = eta' / eta stay IEEE divisions: v_rcp_f32 + one residual correction behind a wave-uniform range test — v_rcp_f32 has
//  no denormal support, pcg_lpk.hip.h — measured SLOWER here too: 1.64-1.67 against 1.60 us per iteration of one N = 128 trajectory; the second
//  code path costs two registers and the branch more than the seven dependent instructions it saves.)
// (Nor does taking beta's reciprocal off the critical path pay — 1 / eta formed half an iteration early, eta' * (1 / eta) behind the barrier, the division
//  outside [2^-100, 2^100]: two more live registers, N = 64 one trajectory 1.07 -> 1.14 us per iteration, N = 128 unchanged.)
// PC3: the preconditioner has off-diagonal blocks (SS); false = block-Jacobi — a build of its own, so that the SS build carries nothing of it (as a
// wave-uniform branch inside one build the second epilogue cost the SS path 4-7 %: tools/_prof/lqb_ab.py).
template <int NMAXQ, bool PC3>
__global__ __launch_bounds__(NMAXQ * 4, 2) void pcg_lqb_kernel(PcgArgs a) {
    typedef LqbLds<NMAXQ> L;
    constexpr int NW = L::NW, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = sched_pick(a.order, (int)blockIdx.x, a.order_tag);
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* red_v = lds + L::RED;
    float* red_e = red_v + NW;
#ifdef MPCG_PROF
    // prologue / write-back stamps of one workgroup of the third round of a throughput-sized launch (slots 16.. of the wave's row: tools/prof_phases.py --cfg lqb)
    const bool pro_on = (int)blockIdx.x == ((int)gridDim.x > 600 ? 600 : 0);
#define LQB_PSTAMP(i) do { if (pro_on) { long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) g_pcg_prof[w * 32 + 16 + (i)] = t_; } } while (0)
#else
#define LQB_PSTAMP(i) do {} while (0)
#endif
    LQB_PSTAMP(0);
    MPCG_WG_STAMP(0);

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- this lane: knot k, column piece h, row piece g; the knot whose piece h it carries: kk = k - g ----
    const int h = lane & 1, g = (lane >> 1) & 1;
    const int k = 16 * w + (lane >> 2);
    const bool diag = h == g;                                  // publishes y (else z)
    const unsigned long long diag_mask = __builtin_amdgcn_ballot_w64(diag);      // (an SGPR pair: the v_cndmask operand of lqb_epilogue)
    constexpr bool p3 = PC3;                                   // (the launcher picks the build by a.pcols)
    constexpr int KN = L::KN, K2 = 2 * KN;
    // float offsets inside a vector: pairs 4 h + rp of knot j at 2 (j + 1) + (4 h + rp) K2; entry 6 + h at 2 (j + 1) + 3 K2 + h
    const int fb = 2 * (k - g + 1) + 4 * h * K2, fs = 2 * (k - g + 1) + 3 * K2 + h;       // the carried piece (knot k - g)
    const int ob = 2 * (k + 1) + 4 * h * K2 + (diag ? 0 : L::VS), os = 2 * (k + 1) + 3 * K2 + h + (diag ? 0 : L::VS);   // what this lane publishes (knot k's slot; Z = T + VS)

    // ---- lambda0 and gamma: requested in front of the matrix stream (in-order returns: the counted waits of the block load cover them), staged in LDS behind it —
    //      their round trip (2 us under a full batch's load) hides behind the eight rounds of the matrices ----
    constexpr int NVEC = (NMAXQ * NS + NTHR - 1) / NTHR;
    float lam_in[NVEC], gam_in[NVEC];
    {
        const float* lam_src = a.lam0 ? a.lam0 + (size_t)b * vstride : lam_g;      // (a fix-up launch behind a forced cluster: PcgArgs::lam0)
#pragma unroll
        for (int i = 0; i < NVEC; ++i) {
            const int e = tid + i * NTHR;
            lam_in[i] = e < N * NS ? lam_src[e] : 0.f;
            gam_in[i] = e < N * NS ? gam[e] : 0.f;
        }
    }

    // ---- matrix registers ----
    LqbSub SD, SL, PD, PL;
    {
        const rsrc_t MS = make_rsrc(static_cast<const char*>(a.S) + (size_t)b * mstride * 4, (uint32_t)(mstride * 4));
        const rsrc_t MP = make_rsrc(static_cast<const char*>(a.Pinv) + (size_t)b * mstride * 4, (uint32_t)(mstride * 4));
        lqb_load_blocks_lds(MS, MP, 16 * w, N, p3, lane, lds + w * LPK_TILE_FLOATS, lds + L::TILE2 + w * LPK_TILE_FLOATS, SD, SL, PD, PL);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    LQB_PSTAMP(1);
    // park the pairs the passes use last (the diagonal sub-blocks' last columns) in lane-private LDS slots: the register file holds 196 matrix
    // registers + the working set of a pass only just, and what the compiler spills goes to SCRATCH, reloaded in every pass (pcg_lpk.hip.h).
    // The slots live in this wavefront's own second load tile, idle from here on.
    f2* const parkS = reinterpret_cast<f2*>(lds + L::TILE2 + w * LPK_TILE_FLOATS) + lane;
    constexpr int NPK = PC3 ? LQB_NPARK : LQB_NPARK_J;
    f2* const parkP = parkS + NPK * 64;
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        parkS[i * 64] = SD.A[i % 3][6 - i / 3];
        parkP[i * 64] = PD.A[i % 3][6 - i / 3];
    }
    lds_barrier();                                         // every wavefront is done with its load tiles: the vectors' region may be written
    LQB_PSTAMP(2);

    // ---- stage vectors: P0 <- lambda0, R0 <- gamma, everything else (pads included) <- 0 ----
    for (int e = tid; e < L::RED + 2 * NW; e += NTHR) lds[e] = 0.f;
    lds_barrier();
#pragma unroll
    for (int i = 0; i < NVEC; ++i) {
        const int e = tid + i * NTHR;
        if (e < N * NS) {
            const int kk = e / NS, ii = e - kk * NS;
            lds[L::P0 + L::at(kk, ii)] = lam_in[i];
            lds[L::R0 + L::at(kk, ii)] = gam_in[i];
        }
    }
    lds_barrier();
    LQB_PSTAMP(3);

#ifdef MPCG_PROF
    bool prof_on = false;
#endif
    // the NW wave partials of an inner product: requested FIRST after a barrier (volatile: in program order, ahead of the operand loads), summed in the
    // same order in every thread — deterministic; packed adds on the loaded register pairs: a tree of depth three
    typedef __attribute__((address_space(3))) const volatile f4 lds_cv_f4;
    struct Red { f4 u, v; };
    auto load_red = [&](const float* red) -> Red {
        Red r;
        if constexpr (NW >= 4) r.u = *(lds_cv_f4*)(red);
        else { const f2 t = lds_ld64(red); r.u = f4{t.x, t.y, 0.f, 0.f}; }
        if constexpr (NW == 8) r.v = *(lds_cv_f4*)(red + 4);
        return r;
    };
    auto sum_red = [&](const Red& r) -> float {
        if constexpr (NW == 8) return lqb_hsum((f2{r.u.x, r.u.y} + f2{r.u.z, r.u.w}) + (f2{r.v.x, r.v.y} + f2{r.v.z, r.v.w}));
        else if constexpr (NW == 4) return lqb_hsum(f2{r.u.x, r.u.y} + f2{r.u.z, r.u.w});
        else return lqb_hsum(f2{r.u.x, r.u.y});
    };
    // the carried piece of the vector at float offset X (knot shift dk: +1 = the slot z of the next knot sits in)
    auto load_piece = [&](int X, int dk) -> LqbPiece {
        const float* x = lds + X + 2 * dk;
        LqbPiece o;
#pragma unroll
        for (int rp = 0; rp < 3; ++rp) o.p[rp] = lds_ld64(x + fb + K2 * rp);
        o.s = lds_ld32(x + fs);
        return o;
    };
    struct Fetch { LqbPiece t, z; };
    auto fetch = [&](int T) -> Fetch { return Fetch{load_piece(T, 0), load_piece(T + L::VS, 1)}; };
    auto bc = [](const LqbPiece& x, auto jt) -> f2 {
        constexpr int J = decltype(jt)::value;
        const float v = J == 6 ? x.s : ((J & 1) ? x.p[J >> 1].y : x.p[J >> 1].x);
        return f2{v, v};
    };

    // One pass of matrix (D, L) over the operand whose carried piece is `mine`: publishes y / z to TOUT / TOUT + VS, the wave's share of x^T M x
    // to red[w].  hasL: wave-uniform (block-Jacobi's Pinv has no off-diagonal blocks).  `park`: this matrix's pairs parked in LDS.
    // Order (register budget): transposed product in two column groups -> direct L (x_{k-1}) -> direct D (x_k), the parked pairs last.
    // The coupling term of the inner product comes from the DIRECT product: x_{k-1}^T (L^T x_k) = x_k^T (L x_{k-1}), so that x_{k-1}'s piece is
    // dead once the L columns are done: x^T M x = sum x_k[piece g] . (ypart + L-part of ypart).
    auto pass = [&](auto hasl_tag, const LqbSub& D, const LqbSub& Lb, const LqbPiece& mine, int TOUT, float* red, const f2* park, int pb) {
        constexpr bool hasL = decltype(hasl_tag)::value;
        const LqbPiece xg = lqb_quad<LQB_QP_XG>(mine);
        MPCG_STAMP(pb + 1);
        LqbPiece ypart, zpart;
        f2 acc[3], dd;
        float y6, ds;
        if constexpr (hasL) {
            // transposed: zpart[piece h] = L^T x_k[piece g]
            const f2 xs = f2{xg.s, xg.s};
            {
                f2 zt[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) zt[j] = Lb.A[0][j] * xg.p[0];
#pragma unroll
                for (int rp = 1; rp < 3; ++rp)
#pragma unroll
                    for (int j = 0; j < 4; ++j) zt[j] = __builtin_elementwise_fma(Lb.A[rp][j], xg.p[rp], zt[j]);
                zpart.p[0] = __builtin_elementwise_fma(Lb.B[0], xs, f2{lqb_hsum(zt[0]), lqb_hsum(zt[1])});
                zpart.p[1] = __builtin_elementwise_fma(Lb.B[1], xs, f2{lqb_hsum(zt[2]), lqb_hsum(zt[3])});
            }
            {
                f2 zt[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) zt[j] = Lb.A[0][4 + j] * xg.p[0];
#pragma unroll
                for (int rp = 1; rp < 3; ++rp)
#pragma unroll
                    for (int j = 0; j < 3; ++j) zt[j] = __builtin_elementwise_fma(Lb.A[rp][4 + j], xg.p[rp], zt[j]);
                zpart.p[2] = __builtin_elementwise_fma(Lb.B[2], xs, f2{lqb_hsum(zt[0]), lqb_hsum(zt[1])});
                zpart.s = fmaf(Lb.e, xg.s, lqb_hsum(zt[2]));
            }
            // direct, off-diagonal block: L x_{k-1}[piece h]
            const LqbPiece xm = lqb_quad<LQB_QP_XM>(mine);
#pragma unroll
            for (int rp = 0; rp < 3; ++rp) acc[rp] = Lb.A[rp][0] * bc(xm, std::integral_constant<int, 0>{});
            SFor14<8>::run([&](auto jt) {                       // j = 1 .. 6
                constexpr int J = decltype(jt)::value - 7;
                const f2 xb = bc(xm, std::integral_constant<int, J>{});
#pragma unroll
                for (int rp = 0; rp < 3; ++rp) acc[rp] = __builtin_elementwise_fma(Lb.A[rp][J], xb, acc[rp]);
            });
            f2 a6 = Lb.B[0] * xm.p[0];
            a6 = __builtin_elementwise_fma(Lb.B[1], xm.p[1], a6);
            a6 = __builtin_elementwise_fma(Lb.B[2], xm.p[2], a6);
            y6 = fmaf(Lb.e, xm.s, lqb_hsum(a6));
            // the coupling term: x_k[piece g] . (L x_{k-1})
            dd = acc[0] * xg.p[0];
            dd = __builtin_elementwise_fma(acc[1], xg.p[1], dd);
            dd = __builtin_elementwise_fma(acc[2], xg.p[2], dd);
            ds = y6 * xg.s;
        } else {
#pragma unroll
            for (int cp = 0; cp < 3; ++cp) { acc[cp] = f2{0.f, 0.f}; zpart.p[cp] = f2{0.f, 0.f}; }      // (compile-time: folded away)
            y6 = 0.f; ds = 0.f; dd = f2{0.f, 0.f}; zpart.s = 0.f;
        }
        // direct, diagonal block: D x_k[piece h]; the pairs parked in LDS are requested here (volatile: in program order) and used last
        const LqbPiece xh = lqb_quad<LQB_QP_XH>(mine);
        f2 pk_[NPK > 0 ? NPK : 1];
#pragma unroll
        for (int i = 0; i < NPK; ++i) pk_[i] = lds_ld64(reinterpret_cast<const float*>(park + i * 64));
        f2 a6 = D.B[0] * xh.p[0];
        a6 = __builtin_elementwise_fma(D.B[1], xh.p[1], a6);
        a6 = __builtin_elementwise_fma(D.B[2], xh.p[2], a6);
        SFor14<7>::run([&](auto jt) {                           // j = 0 .. 6; parked pair i is (rp, j) = (i % 3, 6 - i / 3)
            constexpr int J = decltype(jt)::value - 7;
            const f2 xb = bc(xh, std::integral_constant<int, J>{});
#pragma unroll
            for (int rp = 0; rp < 3; ++rp) {
                const int pi = 3 * (6 - J) + rp;
                acc[rp] = __builtin_elementwise_fma(pi < NPK ? pk_[pi < NPK ? pi : 0] : D.A[rp][J], xb, acc[rp]);
            }
        });
#pragma unroll
        for (int rp = 0; rp < 3; ++rp) ypart.p[rp] = acc[rp];
        ypart.s = (fmaf(D.e, xh.s, lqb_hsum(a6))) + y6;
        // inner product share: x_k[piece g] . ypart (+ the coupling term above)
        dd = __builtin_elementwise_fma(ypart.p[0], xg.p[0], dd);
        dd = __builtin_elementwise_fma(ypart.p[1], xg.p[1], dd);
        dd = __builtin_elementwise_fma(ypart.p[2], xg.p[2], dd);
        MPCG_STAMP(pb + 2);
        float part = fmaf(ypart.s, xg.s, ds) + lqb_hsum(dd);
        LqbPiece fin;
        if constexpr (hasL) lqb_epilogue(part, fin, ypart, zpart, diag_mask);
        else lqb_epilogue_y(part, fin, ypart);
        if (lane == 63) red[w] = part;
        if (hasL || diag) {                                  // (block-Jacobi: the z vector stays at its zeros)
            float* out = lds + TOUT;
#pragma unroll
            for (int rp = 0; rp < 3; ++rp) *reinterpret_cast<f2*>(out + ob + K2 * rp) = fin.p[rp];
            out[os] = fin.s;
        }
        MPCG_STAMP(pb + 3);
    };
    // the carried piece after an update: MODE 1: old - c (T + Z<<1) (r, c = alpha); MODE 2: (T + Z<<1) + c old (p, c = beta)
    auto rebuild = [&](auto mode_tag, const LqbPiece& old, const Fetch& f, float c) -> LqbPiece {
        constexpr int MODE = decltype(mode_tag)::value;
        LqbPiece o;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const f2 u = f.t.p[i] + f.z.p[i]; o.p[i] = MODE == 1 ? old.p[i] - c * u : u + c * old.p[i]; }
        { const float u = f.t.s + f.z.s; o.s = MODE == 1 ? old.s - c * u : u + c * old.s; }
        return o;
    };

    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; eta = r . r~   (p = r~ is formed by the first S half: beta = 0) ----
    LqbPiece lam = load_piece(L::P0, 0);
    LqbPiece rv = load_piece(L::R0, 0);
    LqbPiece pv;
#pragma unroll
    for (int i = 0; i < 3; ++i) pv.p[i] = f2{0.f, 0.f};
    pv.s = 0.f;
    pass(std::true_type{}, SD, SL, lam, L::US, red_v, parkS, 16);
    lds_barrier();
    Fetch f = fetch(L::US);
    rv = rebuild(std::integral_constant<int, 1>{}, rv, f, 1.f);
    pass(std::integral_constant<bool, PC3>{}, PD, PL, rv, L::RT, red_e, parkP, 16);
    lds_barrier();
    Red rd = load_red(red_e);
    f = fetch(L::RT);
    float eta = uniform(sum_red(rd));
    LQB_PSTAMP(4);
    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    float beta = 0.f;                                          // scalar of the NEXT p update (the S half applies it)
    bool p_pending = true;                                     // that update has not been applied to p (write-back of d_p does it)
    if (fabsf(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
            prof_on = b == 0 && it == 20;
#endif
            MPCG_STAMP(0);
            // p = r~ + beta p ; upsilon = S p ; v = p . upsilon
            pv = rebuild(std::integral_constant<int, 2>{}, pv, f, beta);
            pass(std::true_type{}, SD, SL, pv, L::US, red_v, parkS, 0);
            lds_barrier();
            MPCG_STAMP(4);
            rd = load_red(red_v);
            f = fetch(L::US);                                   // (the operand loads fly during the scalar chain)
            // alpha = eta / v ; lambda += alpha p ; r -= alpha upsilon ; r~ = Pinv r ; eta' = r . r~
            const float alpha = uniform(eta / sum_red(rd));    // (IEEE division: 75 clocks of the chain, measured by ablation; the reciprocal forms were slower, see above)
#pragma unroll
            for (int i = 0; i < 3; ++i) lam.p[i] = lam.p[i] + alpha * pv.p[i];
            lam.s = lam.s + alpha * pv.s;
            rv = rebuild(std::integral_constant<int, 1>{}, rv, f, alpha);
            MPCG_STAMP(5);
            pass(std::integral_constant<bool, PC3>{}, PD, PL, rv, L::RT, red_e, parkP, 5);
            lds_barrier();
            MPCG_STAMP(9);
            rd = load_red(red_e);
            f = fetch(L::RT);
            // eta' ; exit test ; beta
            const float eta_new = uniform(sum_red(rd));
            iters = (uint32_t)(it + 1);
            if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; p_pending = false; break; }     // (the reference leaves p as it is on this exit)
            beta = uniform(eta_new / eta);
            eta = eta_new;
            MPCG_STAMP(10);
        }
    }
    LQB_PSTAMP(5);
    // ---- lambda, p, r of knot k from the g = 0 lanes into the staging vectors (free since the setup), then out ----
    if (g == 0) {
        const int sb = 2 * (k + 1) + 4 * h * K2, ss = 2 * (k + 1) + 3 * K2 + h;
#pragma unroll
        for (int rp = 0; rp < 3; ++rp) {
            *reinterpret_cast<f2*>(lds + L::LAM + sb + K2 * rp) = lam.p[rp];
            *reinterpret_cast<f2*>(lds + L::P0 + sb + K2 * rp) = pv.p[rp];
            *reinterpret_cast<f2*>(lds + L::R0 + sb + K2 * rp) = rv.p[rp];
        }
        lds[L::LAM + ss] = lam.s;
        lds[L::P0 + ss] = pv.s;
        lds[L::R0 + ss] = rv.s;
    }
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        lam_g[e] = lds[L::LAM + L::at(kk, i)];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = lds[L::R0 + L::at(kk, i)];
        if (a.p_out) {
            // p of the last completed update; when the loop ended without a tolerance exit that update is still pending: p = r~ + beta p
            float pvv = lds[L::P0 + L::at(kk, i)];
            if (p_pending) pvv = (lds[L::RT + L::at(kk, i)] + lds[L::ZP + L::at(kk + 1, i)]) + beta * pvv;
            a.p_out[(size_t)b * vstride + e] = pvv;
        }
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
    LQB_PSTAMP(6);
    MPCG_WG_STAMP(1);
#undef LQB_PSTAMP
}

}  // namespace mpcg
