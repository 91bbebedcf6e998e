// pcg_lpk.hip.h — "lane pair per knot" PCG kernel for gfx950 (round 3): the register-resident successor of the
// lane-per-block kernel (round 2; retired in round 4, HISTORY.md §3.1c) for fp32, knot_points <= 128.  TWO barriers per PCG iteration.
//
// What the lane-per-block kernel left on the table (HISTORY.md §3.1c, profiles/r02_lpb_ablate.txt): of 5,812 cycles per
// iteration only 46 % were FMA chains; the off-diagonal waves carried 196 packed FMAs per pass against 98 on the diagonal
// waves (two of four SIMDs idle half of every pass); every pass wrote three PART vectors (yD, yL, yT) to LDS that an
// element-wise phase of ALL waves read back, summed and wrote again — FOUR "LDS write -> barrier -> LDS read" stages per
// iteration, ~500 cycles of pure latency each, which is what bounds a kernel with one active wavefront per SIMD.
//
// Mapping.  A knot k owns TWO ADJACENT LANES of one wavefront, per matrix; lane h (0/1) of the pair holds seven columns of BOTH
// blocks of block row k that the lower triangle keeps — D_k = M[k,diag] and L_k = M[k,left] — as 2 x 49 register pairs (196 VGPRs,
// as before: a trajectory of 128 knots fills 2 x 256 lanes = the 512 KB register file).  Lane 0: columns 0..6, lane 1: 8..13, 7.
//   direct      acc[rows]  = sum_{c in half} D_k[:,c] x_k[c] + L_k[:,c] x_{k-1}[c]        98 v_pk_fma_f32, ONE accumulator set
//   transposed  z[c]       = sum_rows L_k[row,c] x_k[row],  c in half                       49 v_pk_fma_f32 + 7 adds
// => 147 packed FMAs in EVERY lane of EVERY wave of the pass (was 196 / 98): four S waves and four Pinv waves, one of each per
// SIMD.  The two column-half partial sums of a knot are merged with DPP (quad_perm [1,0,3,2]: the partner lane): a lane pair ends
// the pass with the complete (D x_k + L x_{k-1}) of its knot IN REGISTERS, split by rows — lane 0 owns row pairs P0..P3 (entries
// 0..7), lane 1 owns P4..P6 (entries 8..13) — and publishes just that (4 x 8 bytes) plus z, the coupling L_k^T x_k that belongs
// to knot k-1 (7 floats).
// Uniform instruction stream for both lanes of a pair: lane 1 keeps its row pairs in the order P4 P5 P6 P3 P0 P1 P2 (lane 0:
// P0 .. P6), so "own pairs" are register slots 0..3 in both, the partner's copy of own slot s is its slot (4, 5, 6, 3)[s], and —
// because the column order follows the same pattern — the lane's own COLUMNS of a vector are its own row-pair slots 0..2 plus one
// half of slot 3: no separate per-column operand loads.  (Lane 1's slot 3 duplicates lane 0's: same bits, written to the same place.)
//
// Two barriers per iteration ("the reader rebuilds").  A pass does not wait for a separate vector-update phase: the lanes of the
// NEXT pass form their operand themselves, from what the previous pass published and the scalar that pass's inner product gives,
//     Pinv pass:  r_new = r_old - alpha (US + ZS<<1)         S pass:  p_new = (RT + ZP<<1) + beta p_old
// (US / RT = the merged row halves, ZS / ZP = the z vectors, <<1 = knot k+1's) — for the 14 entries of knot k and the own 8 of knot
// k-1 — with the very operations, hence the very bits, every other holder of that entry uses.  The iterate vectors p and r therefore
// never exist in LDS inside the loop: every lane carries its 8 + 8 entries (knot k, knot k-1) in registers from one half to the next,
// fetches only US / ZS (RT / ZP) — 16 ds_read_b64 — and gets the remaining 6 entries of knot k from the partner lane by DPP.
// lambda += alpha p is done by the S lanes while the Pinv pass runs.  Per iteration: S half | barrier | Pinv half | barrier.
// Inner product without the assembled vector: x^T M x = sum_k x_k^T (D_k x_k + L_k x_{k-1}) + x_{k-1}^T (L_k^T x_k); the lane has
// both factors of both terms in registers.
//
// Reads only the left + diagonal block columns (include/mpcg.h, BLOCK SYMMETRY), like the lane-per-block kernel.
#pragma once
#include "pcg_kernels.hip.h"

#ifndef LPK_STAGED_LOAD
#define LPK_STAGED_LOAD 1      // matrix registers filled through the LDS stage (lpk_load_blocks_lds); 0: lane-private loads (lpk_load_blocks)
#endif

namespace mpcg {

// LDS layout of one vector: PAIR-MAJOR, V[q][slot] = entries (2q, 2q+1) of knot slot - 1 as one float2, q = 0..6,
// slot = 0..KN-1 (one zero knot in front: knot k lives in slot k + 1; zero knots behind).  Why: every access of the kernel is then
// bank-conflict-free (MI355X_MICROARCH.md §LDS: ds_read_b64 is served in two groups of 32 lanes over 64 banks) —
//   * a lane pair reads row pairs q and q + 4 of the SAME knot (slot orders P0.. / P4..): KN = 4 (mod 8) puts them 32 banks apart;
//   * consecutive knots of one pair are consecutive float2: 16 knots x 2 lanes of a group cover the 64 banks exactly once.
// The first version used knot-major [knot][16 floats]: 8- to 16-way conflicts on every access made the LDS pipe the bottleneck
// (profiles/r03_lpk_phases.txt).
template <int NWR> struct LpkLds {
    static constexpr int NMAX = NWR ? 64 * NWR : 32, NW = NWR ? 4 * NWR : 2;      // (NWR = 0: the HALF build — one wavefront per matrix, 32 knots, four workgroups per CU)
    static constexpr int KN = NMAX + 4;                        // knot slots per row pair: NMAX + 2 rounded up to 4 (mod 8)
    static_assert(KN % 8 == 4, "row pairs q and q + 4 must sit 32 banks apart");
    static constexpr int VS = 7 * KN * 2;                      // floats per vector
    // P0, R0: staging of lambda0 / gamma, at the end p and r for d_p / d_r (inside the loop p and r live in registers only) |
    // US, ZS: what the S pass publishes | RT, ZP: the Pinv pass | lambda | wave partials
    static constexpr int P0 = 0, R0 = VS, US = 2 * VS, ZS = 3 * VS, RT = 4 * VS, ZP = 5 * VS, LAM = 6 * VS, RED = 7 * VS, MX = RED + NW, NPARK = 3, TILE2 = MX + NPARK * 2 * NW * 64,
                         TOTAL = TILE2 + (LPK_STAGED_LOAD ? NW * 8 * 14 * 14 : 0);      // (TILE2: the second load tile of every wavefront, lpk_load_blocks_lds)
    // MX: NPARK matrix register pairs per lane parked in LDS (lane-private float2 slots, [pair][thread]: conflict-free): the register file
    // holds 196 matrix registers + the working set of a half-iteration only just; left to the compiler the overflow goes to SCRATCH, whose
    // reloads (global-memory latency, three per pass) cost more than the whole FMA stream
    // float index of entry i of knot k inside a vector
    __host__ __device__ static constexpr int at(int k, int i) { return 2 * ((i >> 1) * KN + k + 1) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lpk_lds_floats(int NW) {
    return NW == 2 ? (size_t)LpkLds<0>::TOTAL : NW == 4 ? (size_t)LpkLds<1>::TOTAL : (size_t)LpkLds<2>::TOTAL;
}

__device__ __forceinline__ f2 buf_load2(rsrc_t r, uint32_t voff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
    return __builtin_bit_cast(f2, v);
}
// Operand loads are VOLATILE float2 reads: plain ones get merged into ds_read2_b64, which is served in 16-lane groups over a 32-bank
// modulus (MI355X_MICROARCH.md §LDS) — the layout above is conflict-free for ds_read_b64 (32-lane groups, 64 banks) and 2-way conflicted
// for the merged form, at 8 instead of 2 LDS cycles to begin with: 16 cycles per pair of loads instead of 4, which made the operand fetch
// of a half-iteration its longest phase (profiles/r03_lpk_phases.txt).  Volatile accesses are neither merged nor reordered.
typedef __attribute__((address_space(3))) const volatile f2 lds_cv_f2;
__device__ __forceinline__ f2 lds_ld64(const float* p) { return *(lds_cv_f2*)(p); }     // (explicit LDS address space: a volatile access through a generic pointer is a flat load)
// a value every lane holds identically, moved to a scalar register
__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// (alpha and beta are plain IEEE divisions.  v_rcp_f32 + multiply has no denormal support — an over-iterated, exactly converged system
//  turns lambda into inf / NaN where the correctly rounded quotient stays finite, tests/test_gpu_lpk.py N = 2 — and guarded by a range
//  test it measured SLOWER than the 13-instruction division, whose latency is covered by the operand loads issued in front of it:
//  one trajectory 0.318 vs 0.301 ms.)
// value of `v` in the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_partner(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// ---- the matrix registers of a lane: seven columns of D_k and L_k (lane 0 of the pair: columns 0..6; lane 1: 8..13, 7), row pairs in the
// lane's slot order (slot s holds row pair P_q: lane 0: q = s; lane 1: q = (4, 5, 6, 3, 0, 1, 2)[s]).  Shared by pcg_lpk_kernel and
// pcg_lpkc_kernel.  `M` = the trajectory's matrix (bd layout), element size ES bytes: 4 = float, 2 = _Float16 storage (mpcg_pcg_solve_f16:
// converted to fp32 ONCE here — the blocks stay in registers for the whole solve, so fp16 storage costs the resident kernels nothing per
// iteration and every product is the fp32 product of the rounded entry).
//   * Column-major issue order: the seven row pairs of one column of a block are contiguous (56 / 28 bytes), so the loads of a column stay inside
//     one or two 128-byte lines (slot-major order touched 64 lines per load and came back to each 13 loads later: the 16 KB L1 had long
//     evicted it, every line crossed L2 -> L1 seven times).
//   * Byte of (pair q, column c) inside a block = (14 c + 2 q) ES: one lane-variable base per slot, the column as the instruction's immediate
//     offset (columns 8h + j for j < 6, 6 + h for j = 6).
//   * Row pairs (q, q + 1) of one column are contiguous, and in BOTH lanes' slot orders the slot pairs (0, 1) and (4, 5) hold consecutive row
//     pairs (lane 0: P0 P1 / P4 P5, lane 1: P4 P5 / P0 P1): those four slots come in as two double-width loads, slots 2, 3, 6 as single ones —
//     five instead of seven load instructions per column.  The load phase is bound by the L1's tag rate (every lane of a load touches its
//     own cache line: 64 lookups per instruction, ~21 us per 600 KB trajectory at seven loads per column), not by bytes.
__device__ __forceinline__ f2 lpk_h2f(uint32_t w) {
    const h2 v = __builtin_bit_cast(h2, w);
    return f2{(float)v.x, (float)v.y};
}
template <int ES>
__device__ __forceinline__ void lpk_load_blocks(rsrc_t M, int k, int h, bool okD, bool okL, f2 (&Md)[7][7], f2 (&Ml)[7][7]) {
    static_assert(ES == 4 || ES == 2, "float or _Float16 storage");
    constexpr uint32_t PB = 2u * ES, CB = (uint32_t)NS * ES, BLKB = (uint32_t)(NS * NS) * ES;     // bytes of a row pair, a column, a block
    const uint32_t rowb = (uint32_t)k * ((uint32_t)ROWF * ES);
    uint32_t bL[7], bD[7], bL6[7], bD6[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int q1 = s < 3 ? s + 4 : (s == 3 ? 3 : s - 4);
        const uint32_t bs = rowb + PB * (uint32_t)(h ? q1 : s);
        bL[s] = okL ? bs + CB * 8u * (uint32_t)h : OOB_OFF;
        bD[s] = okD ? bs + CB * 8u * (uint32_t)h + BLKB : OOB_OFF;
        bL6[s] = okL ? bs + CB * (uint32_t)(6 + h) : OOB_OFF;
        bD6[s] = okD ? bs + CB * (uint32_t)(6 + h) + BLKB : OOB_OFF;
    }
    auto raw = [](uint32_t w) -> float { return __builtin_bit_cast(float, w); };
    auto load_col = [&](f2 (&Mx)[7][7], const uint32_t (&bs)[7], int j, uint32_t coff) {
        if constexpr (ES == 4) {
            const f4 a01 = buf_load4<false>(M, bs[0] + coff), a45 = buf_load4<false>(M, bs[4] + coff);
            Mx[0][j] = f2{a01.x, a01.y}; Mx[1][j] = f2{a01.z, a01.w};
            Mx[4][j] = f2{a45.x, a45.y}; Mx[5][j] = f2{a45.z, a45.w};
            Mx[2][j] = buf_load2(M, bs[2] + coff);
            Mx[3][j] = buf_load2(M, bs[3] + coff);
            Mx[6][j] = buf_load2(M, bs[6] + coff);
        } else {
            // the two halves of a row pair arrive as ONE dword, parked in .x until every load is under way (lpk_load_blocks' tail converts in
            // place: a conversion here would wait for its load and serialise the latencies of the 28 columns)
            const f2 a01 = buf_load2(M, bs[0] + coff), a45 = buf_load2(M, bs[4] + coff);
            Mx[0][j].x = a01.x; Mx[1][j].x = a01.y;
            Mx[4][j].x = a45.x; Mx[5][j].x = a45.y;
            Mx[2][j].x = raw((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(M, (int)(bs[2] + coff), 0, 0));
            Mx[3][j].x = raw((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(M, (int)(bs[3] + coff), 0, 0));
            Mx[6][j].x = raw((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(M, (int)(bs[6] + coff), 0, 0));
        }
    };
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        load_col(Ml, bL, j, CB * (uint32_t)j);
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler would regroup the loads by base register = slot-major)
    }
    load_col(Ml, bL6, 6, 0u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        load_col(Md, bD, j, CB * (uint32_t)j);
        __builtin_amdgcn_sched_barrier(0);
    }
    load_col(Md, bD6, 6, 0u);
    if constexpr (ES == 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int s = 0; s < 7; ++s) Ml[s][j] = lpk_h2f(__builtin_bit_cast(uint32_t, Ml[s][j].x));
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int s = 0; s < 7; ++s) Md[s][j] = lpk_h2f(__builtin_bit_cast(uint32_t, Md[s][j].x));
    }
}

// ---- the same registers filled through an LDS stage (round 4; default).  lpk_load_blocks above gives every lane its own 16-byte pieces: 70
// load instructions per lane, each touching 64 different cache lines — the load of a 600 KB trajectory was bound by the L1's tag rate at
// ~15 us, a fifth of a warm-started single-trajectory SQP step.  But a lane's share of a block is CONTIGUOUS in memory — lane 0 of a knot's
// pair owns columns 0..6 = the first 392 bytes (196 for fp16) of each block, lane 1 the second half — so the wavefront reads whole blocks
// with full-width loads STRAIGHT INTO LDS (`buffer_load_dwordx4 ... lds`: no staging registers) and the lanes pick their halves out of it:
//   * ROUND r = 0..7 (block column L: r < 4, D: r >= 4): the blocks of the eight knots of lanes 16 (r & 3) .. + 15, 8 x 784 contiguous bytes
//     per knot, as 392 pieces of 16 bytes (8 for fp16) = 6.1 load instructions of 64 lanes, ~10 cache lines each instead of 64;
//   * the pieces land in memory order in one of the wavefront's two 6,272-byte tiles (one aliases the iterate vectors' region, not yet in
//     use; two rounds are in flight), and the 16 lanes of the round read their 49 row pairs as 8-byte LDS loads: knot stride 196 dwords = 4
//     banks, halves 98 dwords = 34 banks apart: conflict-free;
//   * knots outside the horizon / the absent L_0: requested at the out-of-bounds offset (no traffic), their lanes take zeros.
// The registers end up bit-identical to lpk_load_blocks' (tests run both).  `kfirst` = global knot of lanes 0, 1; valid knots are < kend.
// (fp32 storage only: the LDS-destination loads come in 1, 2, 4, 12 and 16 bytes per lane, and a half-precision block is 24.5 x 16 bytes;
//  fp16 storage — experimental — keeps the lane-private loads.)
constexpr int LPK_TILE_FLOATS = 8 * NS * NS;               // one round of one wavefront
__device__ __forceinline__ void lpk_load_blocks_lds(rsrc_t M, int kfirst, int kend, bool hasL, int lane, float* tileA, float* tileB, f2 (&Md)[7][7], f2 (&Ml)[7][7]) {
    constexpr int ES = 4;
    constexpr uint32_t PB = 16u;                           // bytes of a piece: 4 elements
    constexpr uint32_t RPB = 2u * ES, CB = (uint32_t)NS * ES, BLKB = (uint32_t)(NS * NS) * ES, ROWB = (uint32_t)ROWF * ES;
    typedef __attribute__((address_space(3))) char lds_c;
    const int h = lane & 1, kl = (lane >> 1) & 7, myround = lane >> 4;
    // piece i of this lane in a round: p = lane + 64 i < 392: knot p / 49 of the round, piece p % 49 of its block
    uint32_t goff[7];                                      // byte offset from (first knot of the round, block column 0)
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
    auto issue = [&](int r) {                              // round r -> tile r & 1
        const int blk = r >> 2;
        const int kr = kfirst + 8 * (r & 3);               // first knot of the round (wave-uniform)
        const int lo = (blk == 0 && kr == 0) ? 1 : 0;      // knots lo <= pk < hi of the round exist in this block column
        int hi = kend - kr;
        hi = hi < 0 ? 0 : (hi > 8 ? 8 : hi);
        if (blk == 0 && !hasL) hi = 0;
        const uint32_t sbase = (uint32_t)kr * ROWB + (uint32_t)blk * BLKB;
        const bool all = lo == 0 && hi == 8;               // (wave-uniform: the common case needs no per-piece test)
        lds_c* tile = (lds_c*)((r & 1) ? tileB : tileA);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const uint32_t vo = (all || (pk[i] >= lo && pk[i] < hi)) ? goff[i] : OOB_OFF;
            if (i < 6 || lane < 8)                         // (piece 6 exists for lanes 0..7 only: 392 = 6 x 64 + 8; the others must not write past the tile)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(M, (__attribute__((address_space(3))) void*)(tile + 64 * i * 16), 16, (int)vo, (int)sbase, 0, 0);
        }
    };
    auto gather = [&](f2 (&Mx)[7][7], int r) {
        if (myround != (r & 3)) return;
        lds_c* tile = (lds_c*)((r & 1) ? tileB : tileA);
        // a knot outside the horizon / the absent L_0: zeros.  (Its pieces were requested at the out-of-bounds offset, and an LDS-destination
        // load leaves LDS UNTOUCHED for such a lane — the tile still holds an earlier round there.)
        const int gk = kfirst + (lane >> 1);
        if (!(gk < kend && (r >= 4 || (gk > 0 && hasL)))) {
#pragma unroll
            for (int s = 0; s < 7; ++s)
#pragma unroll
                for (int j = 0; j < 7; ++j) Mx[s][j] = f2{0.f, 0.f};
            return;
        }
        // slot s holds row pair q(h, s); columns 8h + j for j < 6 by immediate offset, column 6 + h apart
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int q1 = s < 3 ? s + 4 : (s == 3 ? 3 : s - 4);
            const uint32_t b = (uint32_t)kl * BLKB + RPB * (uint32_t)(h ? q1 : s);
            const uint32_t sb = b + CB * 8u * (uint32_t)h, sb6 = b + CB * (uint32_t)(6 + h);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const uint32_t ad = (j < 6 ? sb + CB * (uint32_t)j : sb6);
                if constexpr (ES == 4) Mx[s][j] = *(__attribute__((address_space(3))) const volatile f2*)(tile + ad);
                else Mx[s][j] = lpk_h2f(*(__attribute__((address_space(3))) const volatile uint32_t*)(tile + ad));
            }
        }
    };
    issue(0);
    issue(1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (r < 7) __builtin_amdgcn_s_waitcnt(0x0F77);      // vmcnt(7): everything but the younger round has landed
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        wave_sync();
        if (r < 4) gather(Ml, r); else gather(Md, r);
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): the tile has been read before the next round is sent into it
        wave_sync();
        if (r + 2 < 8) issue(r + 2);
    }
}

template <int NWR>
__global__ __launch_bounds__(NWR ? NWR * 256 : 128, 2) void pcg_lpk_kernel(PcgArgs a) {
    typedef LpkLds<NWR> L;
    constexpr int NW = L::NW, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = sched_pick(a.order, (int)blockIdx.x, a.order_tag);     // (dispatch order: longest-expected first, sched_order_kernel)
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* red_v = lds + L::RED;
    float* red_e = red_v + NW / 2;

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- role of this wave, knot and column half of this lane ----
    const bool isP = w >= NW / 2;                          // wave-uniform
    const int wl = w - (isP ? NW / 2 : 0);                 // wave of its matrix: 0 .. NW / 2 - 1
    const int li = 64 * wl + lane;
    const int k = li >> 1, h = li & 1;
    const bool p3 = a.pcols == 3;
    const bool hasL = !isP || p3;                          // wave-uniform: block-Jacobi has no off-diagonal Pinv blocks
    const bool valid = k < N;
    // float indices inside a vector (add the vector's offset; K2 = floats between consecutive row pairs):
    //   register slots 0..2 -> pairs 4h + s: bA + K2 s | slot 3 -> pair 3: b0 + 3 K2   (slots 4..6 = the partner lane's pairs: by DPP, never from LDS)
    //   knot k - 1: subtract 2; knot k + 1: add 2
    constexpr int KN = L::KN, K2 = 2 * KN;
    const int b0 = 2 * (k + 1);
    const int bA = b0 + (h ? 4 * K2 : 0);

    // ---- matrix registers: seven columns of D_k and L_k (lane 0: columns 0..6; lane 1: 8..13, 7), row pairs in this lane's slot order ----
    f2 Md[7][7], Ml[7][7];                                 // [slot][j]
    {
        const size_t es = a.esz == 2 ? 2 : 4;
        const rsrc_t M = make_rsrc(static_cast<const char*>(isP ? a.Pinv : a.S) + (size_t)b * mstride * es, (uint32_t)(mstride * es));
#if LPK_STAGED_LOAD
        // (through the wavefront's two LDS tiles: one inside the iterate vectors' region, idle until the barrier below, one behind the parking area)
        static_assert(NW * LPK_TILE_FLOATS <= L::RED, "the first load tiles alias the iterate vectors");
        float* tileA = lds + w * LPK_TILE_FLOATS;
        float* tileB = lds + L::TILE2 + w * LPK_TILE_FLOATS;
        const int kfirst = 32 * wl;
        if (a.esz == 2) lpk_load_blocks<2>(M, k, h, valid, valid && k > 0 && hasL, Md, Ml);
        else lpk_load_blocks_lds(M, kfirst, N, hasL, lane, tileA, tileB, Md, Ml);
#else
        const bool okD = valid, okL = valid && k > 0 && hasL;
        if (a.esz == 2) lpk_load_blocks<2>(M, k, h, okD, okL, Md, Ml);
        else lpk_load_blocks<4>(M, k, h, okD, okL, Md, Ml);
#endif
    }

    // park the pairs the pass uses last (rows 4..6 of the diagonal block's seventh column) in LDS; they are fetched back inside the pass
    f2* const park = reinterpret_cast<f2*>(lds + L::MX) + tid;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
#pragma unroll
    for (int i = 0; i < L::NPARK; ++i) park[i * NTHR] = Md[4 + i][6];
#if LPK_STAGED_LOAD
    lds_barrier();                                         // (every wavefront is done with its load tile: the vectors' region may be written)
#endif

    // ---- stage vectors: P0 <- lambda0 (operand of the setup product), lambda <- lambda0, R0 <- gamma, everything else (pads included) <- 0 ----
    for (int e = tid; e < L::RED; e += NTHR) lds[e] = 0.f;
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        const float l0 = a.lam0 ? a.lam0[(size_t)b * vstride + e] : lam_g[e];      // (a fix-up launch behind a forced cluster: PcgArgs::lam0)
        lds[L::P0 + L::at(kk, i)] = l0;
        lds[L::LAM + L::at(kk, i)] = l0;
        lds[L::R0 + L::at(kk, i)] = gam[e];
    }
    lds_barrier();

#ifdef MPCG_PROF
    bool prof_on = false;
#endif
    auto wave_fold = [&](float part) -> float {
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1"
            : "+v"(part));
        const int pb = __builtin_bit_cast(int, part);
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 48));
        return ((part + r1) + r2) + r3;
    };
    // sum of the NW/2 wave partials of one inner product (the waves of one matrix), same order in every thread: deterministic
    auto sum_red = [&](const float* red) -> float {
        if constexpr (NW == 8) {
            const f4 v = *reinterpret_cast<const f4*>(red);
            return ((v.x + v.y) + v.z) + v.w;
        } else if constexpr (NW == 4) {
            const f2 v = *reinterpret_cast<const f2*>(red);
            return v.x + v.y;
        } else {
            return red[0];
        }
    };
    struct Own { f2 v[4]; };
    // own entries (register slots 0..3) of knot k + dk of the vector at float offset X
    auto load_own = [&](int X, int dk) -> Own {
        const float* x = lds + X + 2 * dk;
        Own o;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.v[s] = *reinterpret_cast<const f2*>(x + bA + K2 * s);
        o.v[3] = *reinterpret_cast<const f2*>(x + b0 + 3 * K2);
        return o;
    };
    // (no `k < N` predicate: a lane beyond the horizon holds all-zero blocks — its products, its z and its copies of the vectors are exact
    //  zeros, and its knot slots exist (NMAX + 4 of them) — so it may write them; three exec-mask branches less per half-iteration)
    auto store_own = [&](int X, const Own& o) {
        float* x = lds + X;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<f2*>(x + bA + K2 * s) = o.v[s];
        *reinterpret_cast<f2*>(x + b0 + 3 * K2) = o.v[3];
    };

    // The operand loads of a half-iteration, requested as soon as the barrier in front of it is passed — BEFORE the scalar of the update
    // (alpha / beta: LDS read of the partials, three adds, a 13-instruction IEEE division, the exit test) is worked out: that chain is
    // ~300 cycles of dependent latency and the loads do not depend on it.  Own row pairs (register slots 0..3) of knot k and of knot k-1
    // of the two published vectors the operand is formed from: 16 volatile ds_read_b64.
    struct Fetch { f2 t[4], z[4], gt[4], gz[4]; };
    auto fetch = [&](int T, int Z) -> Fetch {
        const float* xt = lds + T;
        const float* xz = lds + Z + 2;
        Fetch f;
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.t[s] = lds_ld64(xt + bA + K2 * s); f.z[s] = lds_ld64(xz + bA + K2 * s); }
        f.t[3] = lds_ld64(xt + b0 + 3 * K2); f.z[3] = lds_ld64(xz + b0 + 3 * K2);
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.gt[s] = lds_ld64(xt - 2 + bA + K2 * s); f.gz[s] = lds_ld64(xz - 2 + bA + K2 * s); }
        f.gt[3] = lds_ld64(xt - 2 + b0 + 3 * K2); f.gz[3] = lds_ld64(xz - 2 + b0 + 3 * K2);
        return f;
    };
    struct Vec { Own k, m; };                                  // a lane's copy of a vector: its own row pairs of knot k and of knot k-1

    // One half-iteration of this wave's matrix.  `old` = the lane's register copy of the vector being updated (r or p).
    //   MODE 0: the operand is `old` as it stands (setup product S lambda0);
    //   MODE 1: operand = old - c (T + Z<<1)          (Pinv half: r_new, c = alpha; setup: c = 1)
    //   MODE 2: operand = (T + Z<<1) + c old          (S half: p_new, c = beta; first iteration: c = 0)
    // (block-Jacobi: the Pinv waves never write ZP, it stays zero.)  The lane forms its OWN row pairs of the operand for knot k and for
    // knot k-1, takes the other three pairs of knot k from the partner lane's registers (DPP), runs the pass, publishes the merged own
    // row pairs to TOUT and z to ZOUT, the wave's share of x^T M x to red[wl].  Returns the operand (= the updated vector) for the next half.
    auto half = [&](auto mode_tag, const Fetch& f, const Vec& old, float c, int TOUT, int ZOUT, float* red, int pb) -> Vec {
        constexpr int MODE = decltype(mode_tag)::value;
        MPCG_STAMP(pb + 0);
        f2 xk[7];
        Own om;                                                  // knot k-1, own entries
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { xk[s] = old.k.v[s]; om.v[s] = old.m.v[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) { const f2 u = f.t[s] + f.z[s]; xk[s] = MODE == 1 ? old.k.v[s] - c * u : u + c * old.k.v[s]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) { const f2 u = f.gt[s] + f.gz[s]; om.v[s] = MODE == 1 ? old.m.v[s] - c * u : u + c * old.m.v[s]; }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) xk[4 + s] = f2{dpp_partner(xk[s].x), dpp_partner(xk[s].y)};
        Own me;
#pragma unroll
        for (int s = 0; s < 4; ++s) me.v[s] = xk[s];
        MPCG_STAMP(pb + 1);
        f2 acc[7];
        float cterm = 0.f;
        const float xk6 = h ? xk[3].y : xk[3].x;
#if defined(MPCG_ABLATE_LPK) && (MPCG_ABLATE_LPK & 1)     // (timing experiments only, tools/lpk_ablate.sh: results are wrong with any bit set)
        if (false) {
#else
        if (hasL) {
#endif
            // transposed: z[j] = sum over row pairs of L[pair][column j] (.) x_k[pair]; four + three independent chains (register budget)
            f2 z2[3];
            float z6;
            {
                f2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = Ml[0][j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = __builtin_elementwise_fma(Ml[s][j], xk[s], t[j]);
                z2[0] = f2{t[0].x + t[0].y, t[1].x + t[1].y};
                z2[1] = f2{t[2].x + t[2].y, t[3].x + t[3].y};
            }
            {
                f2 t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) t[j] = Ml[0][4 + j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] = __builtin_elementwise_fma(Ml[s][4 + j], xk[s], t[j]);
                z2[2] = f2{t[0].x + t[0].y, t[1].x + t[1].y};
                z6 = t[2].x + t[2].y;
            }
            const float xm6 = h ? om.v[3].y : om.v[3].x;        // x_{k-1} at this lane's seventh column (entry 6 + h)
            {                                                    // z belongs to knot k-1's vector: entries of this lane's columns
                float* zo = lds + ZOUT;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<f2*>(zo + bA + K2 * s) = z2[s];
                zo[b0 + 3 * K2 + h] = z6;
            }
            // second copy of the coupling term of the inner product: x_{k-1}^T (L_k^T x_k), own columns
            f2 ct = z2[0] * om.v[0];
            ct = __builtin_elementwise_fma(z2[1], om.v[1], ct);
            ct = __builtin_elementwise_fma(z2[2], om.v[2], ct);
            cterm = fmaf(z6, xm6, ct.x + ct.y);
            MPCG_STAMP(pb + 2);
            // direct, off-diagonal columns: acc = L[:, c_j] x_{k-1}[c_j]
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Ml[s][0] * f2{om.v[0].x, om.v[0].x};
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                const float xs = (j & 1) ? om.v[j >> 1].y : om.v[j >> 1].x;
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Ml[s][j], f2{xs, xs}, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Ml[s][6], f2{xm6, xm6}, acc[s]);
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][0], f2{xk[0].x, xk[0].x}, acc[s]);
        } else {
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Md[s][0] * f2{xk[0].x, xk[0].x};
        }
        // (the parked pairs are requested here, volatile = in program order, and consumed by the last FMAs of the pass)
        f2 pk_[L::NPARK];
#pragma unroll
        for (int i = 0; i < L::NPARK; ++i) pk_[i] = lds_ld64(reinterpret_cast<const float*>(park + i * NTHR));
        // direct, diagonal columns
#if !(defined(MPCG_ABLATE_LPK) && (MPCG_ABLATE_LPK & 2))
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            const float xs = (j & 1) ? xk[j >> 1].y : xk[j >> 1].x;
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][j], f2{xs, xs}, acc[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(Md[s][6], f2{xk6, xk6}, acc[s]);
#pragma unroll
        for (int i = 0; i < L::NPARK; ++i) acc[4 + i] = __builtin_elementwise_fma(pk_[i], f2{xk6, xk6}, acc[4 + i]);
#endif
        MPCG_STAMP(pb + 3);
        // merge the two column halves: own slot s + the partner's slot (4, 5, 6, 3)[s]
        Own o;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f2 oth = acc[s < 3 ? s + 4 : 3];
            o.v[s] = f2{acc[s].x + dpp_partner(oth.x), acc[s].y + dpp_partner(oth.y)};
        }
        store_own(TOUT, o);
        // inner product share: x_k . (D x_k + L x_{k-1}) over this lane's OWN rows (lane 1's slot 3 duplicates lane 0's: weight 0) + the coupling copy
        f2 d0 = o.v[0] * me.v[0], d1 = o.v[1] * me.v[1];
        d0 = __builtin_elementwise_fma(o.v[2], me.v[2], d0);
        const f2 d3 = o.v[3] * me.v[3];
        const f2 dd = d0 + d1;
#if defined(MPCG_ABLATE_LPK) && (MPCG_ABLATE_LPK & 4)
        const float part = ((dd.x + dd.y) + (h ? 0.f : d3.x + d3.y)) + cterm;
#else
        const float part = wave_fold(((dd.x + dd.y) + (h ? 0.f : d3.x + d3.y)) + cterm);
#endif
        if (lane == 0) red[wl] = part;
        MPCG_STAMP(pb + 4);
        return Vec{me, om};
    };

    // The S waves and the Pinv waves run the same barrier sequence through two SEPARATE code paths (the role is wave-uniform).
    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    float beta = 0.f;                                          // scalar of the NEXT p update (the S half applies it)
    bool p_pending = true;                                     // that update has not been applied to p (write-back of d_p does it)
    auto run_role = [&](auto role_tag) {
        constexpr bool P = decltype(role_tag)::value;
        // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; eta = r . r~   (p = r~ is formed by the first S half: beta = 0) ----
        Fetch f;
        Vec x;                                                   // S waves: p;  Pinv waves: r
        x.k = load_own(P ? L::R0 : L::P0, 0);
        x.m = load_own(P ? L::R0 : L::P0, -1);
        if constexpr (!P) (void)half(std::integral_constant<int, 0>{}, f, x, 0.f, L::US, L::ZS, red_v, 0);
        lds_barrier();
        if constexpr (P) {
            f = fetch(L::US, L::ZS);
            x = half(std::integral_constant<int, 1>{}, f, x, 1.f, L::RT, L::ZP, red_e, 8);
        }
        lds_barrier();
        if constexpr (!P) f = fetch(L::RT, L::ZP);
        float eta = uniform(sum_red(red_e));
        // every matrix load has been consumed on the waves that ran a setup pass; say so, or the compiler keeps vmcnt waits inside the loop
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        if (fabsf(eta) < a.exit_tol) {
            max_iter_exit = 0;
        } else {
            for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
                prof_on = b == 0 && it == 20;
#endif
                if constexpr (!P) {
                    // p = r~ + beta p ; upsilon = S p ; v = p . upsilon
                    x = half(std::integral_constant<int, 2>{}, f, x, beta, L::US, L::ZS, red_v, 0);
                    MPCG_STAMP(5);
                    lds_barrier();
                    MPCG_STAMP(6);
                    // alpha = eta / v ; lambda += alpha p (own entries) — while the Pinv half runs
                    const Own cur = load_own(L::LAM, 0);
                    const float alpha = uniform(eta / sum_red(red_v));
                    Own nw;
#pragma unroll
                    for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] + alpha * x.k.v[s];
                    store_own(L::LAM, nw);
                    MPCG_STAMP(7);
                    lds_barrier();
                    MPCG_STAMP(15);
                    f = fetch(L::RT, L::ZP);                    // the next S half's operand loads fly during the scalar chain below
                } else {
                    MPCG_STAMP(5);
                    lds_barrier();
                    MPCG_STAMP(6);
                    // alpha = eta / v ; r -= alpha upsilon ; r~ = Pinv r ; eta' = r . r~
                    f = fetch(L::US, L::ZS);
                    const float alpha = uniform(eta / sum_red(red_v));
                    x = half(std::integral_constant<int, 1>{}, f, x, alpha, L::RT, L::ZP, red_e, 8);
                    MPCG_STAMP(14);
                    lds_barrier();
                    MPCG_STAMP(15);
                }
                // eta' ; exit test ; beta
                const float eta_new = uniform(sum_red(red_e));
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; p_pending = false; break; }     // (the reference leaves p as it is on this exit)
                beta = uniform(eta_new / eta);
                eta = eta_new;
            }
        }
        // the vector this role carries, for d_p / d_r (the staging buffers are free since the setup)
        store_own(P ? L::R0 : L::P0, x.k);
    };
    if (isP) run_role(std::true_type{}); else run_role(std::false_type{});

    // ---- write back ----
    lds_barrier();
    constexpr int XPf = L::P0, XRf = L::R0;                    // (the reference leaves p and r of the last completed update in d_p / d_r)
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        lam_g[e] = lds[L::LAM + L::at(kk, i)];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = lds[XRf + L::at(kk, i)];
        if (a.p_out) {
            // p of the last completed update; when the loop ended without a tolerance exit that update is still pending: p = r~ + beta p
            float pv = lds[XPf + L::at(kk, i)];
            if (p_pending) pv = (lds[L::RT + L::at(kk, i)] + (p3 ? lds[L::ZP + L::at(kk + 1, i)] : 0.f)) + beta * pv;
            a.p_out[(size_t)b * vstride + e] = pv;
        }
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
}

}  // namespace mpcg
