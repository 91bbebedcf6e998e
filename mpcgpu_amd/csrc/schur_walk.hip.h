// schur_walk.hip.h — round 4: the Schur + preconditioner formation as ONE pass over the KKT blocks, and the
// dz recovery on the same rows-in-lanes primitives (row 3).  gfx950.
//
// What it replaces, and why.  Round 3 ran the formation as three kernels (invert G in place | block rows from the inverses | symmetric-stair
// completion): 12.5 KB of HBM traffic per knot against 8.0 KB algorithmic (every inverse and every theta^-1 went out to memory and came
// back two or three times) and ~3,850 VALU instructions per four knots.  Here a 16-lane DPP row WALKS a chunk of consecutive block rows of
// one trajectory and carries what the next row needs in registers — (Q_{k-1} + rho I)^-1 and theta_{k-1}^-1 — so that
//   * every Q and every R is inverted once, every theta once; G^-1 is written once, in place (no staging copy);
//   * the symmetric-stair couplings Pinv[k,0] and Pinv[k-1,2] are formed from the registers that hold theta_k^-1, phi_k, theta_{k-1}^-1
//     (the round-3 completion kernel re-read three Pinv and two S blocks per knot);
//   * what is left for a second, small kernel (schur_seam_kernel) is the seam between two chunks: the two coupling blocks that need the
//     theta^-1 of both sides, and the one Q^-1 per chunk that cannot be stored in place because the neighbouring chunk still reads the raw Q.
// Same arithmetic, operation for operation, as linsys_setup.cuh:139-562 and :9-137 (every product a rounded multiply followed by a rounded
// add, sequential over the contracted index, accumulators starting at +0; Gauss-Jordan without pivoting, matrix.cuh:120-238) — the C
// oracle's bits; tests/test_gpu_schur.py compares bits.
//
// Instruction diet of the primitives (the kernels are VALU-issue bound, not HBM bound, once the traffic is down):
//   * operands live as PAIRS of neighbouring columns (float2 in an even-aligned register pair): the rounded add of a product term is one
//     v_pk_add_f32 for two columns (the multiply keeps its DPP row-broadcast source, which packed instructions do not have):
//     1.5 instead of 2 VALU instructions per multiply-add;
//   * Gauss-Jordan: the pivot row is scaled IN PLACE under an EXEC mask (one lane per row active: 7 v_pk_mul_f32, and the IEEE division is
//     inside the mask too), every other row then adds (-pcol) x (pivot row entry); the pivot lane runs the same instruction with +0.0 as
//     its multiplier, and x + (+0 * x) has the bits of x for every finite x including both zeros — so there is no per-element select:
//     2 instead of 4 VALU instructions per element and pivot step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "schur_kernels.hip.h"

namespace mpcg {
namespace sw {

#pragma clang fp contract(off)

typedef float f2 __attribute__((ext_vector_type(2)));

// A/B switches of the round-4 experiments (tools/_prof/ab_walk.sh builds the variants)
#ifndef SW_PREFETCH
#define SW_PREFETCH 1      // operands of the next block row requested one row ahead
#endif
#ifndef SW_PAIR
#define SW_PAIR 0          // Q and R inverted together (two dependency chains)
#endif
#ifndef SW_ABLATE
#define SW_ABLATE 0        // timing experiments only: 1 no stores (values kept alive), 2 no loads (operands made up from the offset), 4 no Gauss-Jordan, 8 no products, 16 stores confined to 64 KB per array
#endif
#ifndef SW_STAGE
#define SW_STAGE 1         // stores leave through the LDS stage in 16-byte pieces
#endif
#ifndef SW_NT_STORE
#define SW_NT_STORE 0      // cache policy of the output stores (buffer aux bits: 2 = nt)
#endif
#ifndef SW_MASKED
#define SW_MASKED 0        // 1: pivot row scaled under an EXEC mask with the compiler's IEEE division inside the mask
#endif

// value held by lane L of this lane's 16-lane row
template <int L>
__device__ __forceinline__ float rbc(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + L, 0xf, 0xf, true));
}

template <int I, int E>
struct SFor {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, I>{});
        SFor<I + 1, E>::run(f);
    }
};
template <int E>
struct SFor<E, E> {
    template <class F>
    static __device__ __forceinline__ void run(F&&) {}
};

constexpr int np(int nc) { return (nc + 1) / 2; }
// phase boundary: the scheduler may not move instructions across it (left alone it interleaves the independent inversions and products of
// a block row for latency, which needs more than 256 registers)
#define SW_FENCE() __builtin_amdgcn_sched_barrier(0)
// column i of an operand kept as column pairs
#define SW_EL(X, i) (((i) & 1) ? (X)[(i) >> 1].y : (X)[(i) >> 1].x)

// acc(pair) += (t0, t1); the empty asm keeps the two multiplies scalar (v_mul_f32_dpp): left alone, the vectoriser turns them into
// 2 x v_mov_b32_dpp + v_pk_mul_f32.
__device__ __forceinline__ void acc2(f2& acc, float t0, float t1) {
    asm("" : "+v"(t0), "+v"(t1));
    acc = acc + f2{t0, t1};
}

// A private copy of a broadcast operand that the optimiser cannot identify with the original.  Two products that broadcast the same
// entries of the same operand (Dk L and Dm L^T below) otherwise share ONE v_mov_b32_dpp per entry, kept alive from the first product to
// the second — 196 registers; a broadcast with a single use folds into its multiply (v_mul_f32_dpp) and costs no register at all.
template <int NPAIR>
__device__ __forceinline__ void launder(f2 (&D)[NPAIR], const f2 (&Src)[NPAIR]) {
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) { D[j] = Src[j]; asm volatile("" : "+v"(D[j])); }
}

// An ordering point for the scheduler: everything that produces `acc` is finished before anything that consumes `a` starts.  Left
// alone, the scheduler (which sees a 256-register budget) likes to issue ALL 196 multiplies of a product first and the adds afterwards —
// 196 live products, spilled — whenever the code around the product changes a little; which products it did that to changed from build to
// build.
template <int NPAIR>
__device__ __forceinline__ void pin(float& a, f2 (&acc)[NPAIR]) {
    if constexpr (NPAIR == 7) asm volatile("" : "+v"(a), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
    else if constexpr (NPAIR == 4) asm volatile("" : "+v"(a), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    else static_assert(NPAIR == 7 || NPAIR == 4, "7 or 4 column pairs");
}

// C[r][c] = sum_t A[r][t] * B[t][c]      A: NI columns per lane; B: rows in lanes 0..NI-1, NC columns
template <int NI, int NC>
__device__ __forceinline__ void gemm_nn(const f2 (&A)[np(NI)], const f2 (&Bsrc)[np(NC)], f2 (&Cm)[np(NC)]) {
    f2 B[np(NC)];
    launder(B, Bsrc);
#if SW_ABLATE & 8
    for (int j = 0; j < np(NC); ++j) Cm[j] = B[j] + A[j % np(NI)];
    return;
#endif
#pragma unroll
    for (int j = 0; j < np(NC); ++j) Cm[j] = f2{0.f, 0.f};
    SFor<0, NI>::run([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        float a = SW_EL(A, T);
        pin(a, Cm);                  // the products of term T start after the sums of term T-1: at most NC products in flight
#pragma unroll
        for (int j = 0; j < np(NC); ++j) {
            const float t0 = a * rbc<T>(B[j].x);
            const float t1 = (2 * j + 1 < NC) ? a * rbc<T>(B[j].y) : 0.f;
            acc2(Cm[j], t0, t1);
        }
    });
}
// C[r][c] = sum_t A[r][t] * Bt[c][t]     Bt: row c in lane c (NC rows), NI columns
template <int NI, int NC>
__device__ __forceinline__ void gemm_nt(const f2 (&A)[np(NI)], const f2 (&Btsrc)[np(NI)], f2 (&Cm)[np(NC)]) {
    f2 Bt[np(NI)];
    launder(Bt, Btsrc);
#if SW_ABLATE & 8
    for (int j = 0; j < np(NC); ++j) Cm[j] = Bt[j % np(NI)] + A[j % np(NI)];
    return;
#endif
    SFor<0, np(NC)>::run([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f2 acc{0.f, 0.f};
        if constexpr (J > 0) {       // column pair J starts after pair J-1 is finished: 2 NI products in flight at most
            float dummy = Cm[J - 1].x;
            pin(dummy, Bt);
            Cm[J - 1].x = dummy;
        }
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const float a = SW_EL(A, t), b = SW_EL(Bt, t);
            const float t0 = a * rbc<2 * J>(b);
            const float t1 = (2 * J + 1 < NC) ? a * rbc<(2 * J + 1 < 16 ? 2 * J + 1 : 0)>(b) : 0.f;
            acc2(acc, t0, t1);
        }
        Cm[J] = acc;
    });
}
// out[r] = sum_c M[r][c] * v[c]          v: element c in lane c
template <int NC>
__device__ __forceinline__ float matvec(const f2 (&M)[np(NC)], float v) {
    float acc = 0.f;
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        const float p = SW_EL(M, Cc) * rbc<Cc>(v);
        acc = acc + p;
    });
    return acc;
}
// two of them at once: (M1 v1, M2 v2), one packed add per column
template <int NC>
__device__ __forceinline__ f2 matvec2(const f2 (&M1)[np(NC)], float v1, const f2 (&M2)[np(NC)], float v2) {
    f2 acc{0.f, 0.f};
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        acc2(acc, SW_EL(M1, Cc) * rbc<Cc>(v1), SW_EL(M2, Cc) * rbc<Cc>(v2));
    });
    return acc;
}

// 1 / x with the bits of the IEEE division (the oracle divides).  v_rcp_f32 + ONE Newton step reproduces the correctly rounded quotient
// for EVERY float with 2^-100 <= |x| <= 2^100 — checked exhaustively on the chip, all 2^32 bit patterns, tools/_prof/rcp_exhaustive.hip
// (profiles/r04_rcp_exhaustive.txt: 0 mismatches in that range; outside it — denormal quotients, overflow — the short form differs) — 3
// dependent instructions instead of the 11 of the compiler's division sequence, at the head of every pivot step's dependency chain.
// Pivots outside the range (a singular or wildly scaled block) take the division.
__device__ __forceinline__ float recip_ieee(float x, bool wanted) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    float r = __builtin_fmaf(e, r0, r0);
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(wanted && !(ax >= 0x1p-100f && ax <= 0x1p100f), 0)) r = 1.0f / x;
    return r;
}

// One pivot step P of the Gauss-Jordan elimination of [A | I] (include/utils/matrix.cuh:120-238; rows in lanes 0..NN-1, lr = lane index
// inside the 16-lane row).  Half of the 2 NN columns are structurally inert at every pivot step and are skipped (columns of A at or left of
// the pivot: finished, never read again; columns of I right of the pivot: still unit columns, the reference's update leaves them as they
// are) — round 3, no bit changed.  Round 4: the pivot row is scaled by a per-lane multiplier (1 / pivot in the pivot lane, exactly 1.0 in
// every other lane: x * 1.0 is x), then every lane adds (-pcol) * (pivot row entry) with +0.0 as the pivot lane's multiplier.
template <int NN, int P>
__device__ __forceinline__ void gj_step(f2 (&A)[np(NN)], f2 (&I)[np(NN)], int lr) {
    constexpr int NP = np(NN);
    constexpr int JP = P / 2;                    // the pair that holds column P
    constexpr bool ODD = P & 1;
    const float app = SW_EL(A, P);               // pivot column entry of this row (the pivot itself in lane P)
    const bool is_p = lr == P;
    const float nmul = is_p ? 0.f : -app;        // x + (+0)(x) keeps the bits of x: the pivot lane needs no select
#if SW_MASKED
    if (is_p) {
        const float mrow = 1.0f / app;
        if (!ODD && 2 * JP + 1 < NN) A[JP].y = A[JP].y * mrow;
#pragma unroll
        for (int j = JP + 1; j < NP; ++j) A[j] = A[j] * f2{mrow, mrow};
#pragma unroll
        for (int j = 0; j < JP; ++j) I[j] = I[j] * f2{mrow, mrow};
        if (ODD) I[JP] = I[JP] * f2{mrow, mrow};
        else I[JP].x = I[JP].x * mrow;
    }
#else
    const float pinv = recip_ieee(app, is_p);    // (matrix.cuh:146)
    const float mrow = is_p ? pinv : 1.0f;
    // pair JP of A: column P is finished; its neighbour P+1 (P even) is live.  Pair JP of I: column P is live, its neighbour P+1 (P even) is
    // still a unit column and must keep its +0 in the pivot lane.
    if (!ODD && 2 * JP + 1 < NN) A[JP].y = A[JP].y * mrow;
#pragma unroll
    for (int j = JP + 1; j < NP; ++j) A[j] = A[j] * f2{mrow, mrow};
#pragma unroll
    for (int j = 0; j < JP; ++j) I[j] = I[j] * f2{mrow, mrow};
    if (ODD) I[JP] = I[JP] * f2{mrow, mrow};
    else I[JP].x = I[JP].x * mrow;
#endif
    // every row: x += (-pcol) * (pivot row entry)
    if (!ODD && 2 * JP + 1 < NN) A[JP].y = A[JP].y + nmul * rbc<P>(A[JP].y);
#pragma unroll
    for (int j = JP + 1; j < NP; ++j) {
        const float t0 = nmul * rbc<P>(A[j].x);
        const float t1 = (2 * j + 1 < NN) ? nmul * rbc<P>(A[j].y) : 0.f;
        acc2(A[j], t0, t1);
    }
#pragma unroll
    for (int j = 0; j < JP; ++j) acc2(I[j], nmul * rbc<P>(I[j].x), nmul * rbc<P>(I[j].y));
    if (ODD) acc2(I[JP], nmul * rbc<P>(I[JP].x), nmul * rbc<P>(I[JP].y));
    else I[JP].x = I[JP].x + nmul * rbc<P>(I[JP].x);
}
template <int NN>
__device__ __forceinline__ void unit_rows(f2 (&I)[np(NN)], int lr) {
#pragma unroll
    for (int j = 0; j < np(NN); ++j) I[j] = f2{lr == 2 * j ? 1.f : 0.f, lr == 2 * j + 1 ? 1.f : 0.f};
}
// A destroyed, I becomes A^-1
template <int NN>
__device__ __forceinline__ void invert(f2 (&A)[np(NN)], f2 (&I)[np(NN)], int lr) {
    unit_rows<NN>(I, lr);
#if SW_ABLATE & 4
    for (int j = 0; j < np(NN); ++j) I[j] = I[j] + A[j];
    return;
#endif
    SFor<0, NN>::run([&](auto pc) { gj_step<NN, decltype(pc)::value>(A, I, lr); });
}
// two independent inversions advanced together (pivot P of both in the same step): two dependency chains for the scheduler to interleave
template <int N1, int N2>
__device__ __forceinline__ void invert_pair(f2 (&A1)[np(N1)], f2 (&I1)[np(N1)], f2 (&A2)[np(N2)], f2 (&I2)[np(N2)], int lr) {
    static_assert(N2 <= N1, "the shorter one second");
    unit_rows<N1>(I1, lr);
    unit_rows<N2>(I2, lr);
    SFor<0, N1>::run([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        gj_step<N1, P>(A1, I1, lr);
        if constexpr (P < N2) gj_step<N2, P>(A2, I2, lr);
    });
}

// ---- memory: every array goes through a buffer resource (base in SGPRs, ONE 32-bit byte offset per lane and block, column offsets as
// instruction immediates).  With per-lane 64-bit pointers the walking kernel kept ~20 register pairs of addresses alive across its loop and
// spilled.  A lane that must not store is sent to OOB_OFF (the hardware drops the access): no branches around the stores. ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t SW_OOB = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, uint32_t off) {
#if SW_ABLATE & 2
    return 1.0f + (float)(off & 1023u) * 0x1p-12f;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
#endif
}
__device__ __forceinline__ void bst(rsrc_t r, uint32_t off, float v) {
#if SW_ABLATE & 1
    asm volatile("" :: "v"(v), "v"(off));
#else
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, SW_NT_STORE);
#endif
}
// row lr of a column-major rows x COLS matrix that starts at byte `off`, as column pairs.  Rows beyond the matrix repeat its last row:
// what those lanes compute stays in those lanes (every broadcast reads a lane below `rows`) and is never stored.
template <int COLS>
__device__ __forceinline__ void load_rows(f2 (&M)[np(COLS)], rsrc_t r, uint32_t off, int rows, int lr) {
    const uint32_t v = off + 4u * (uint32_t)(lr < rows ? lr : rows - 1);
#pragma unroll
    for (int j = 0; j < np(COLS); ++j) {
        M[j].x = bld(r, v + 4u * (uint32_t)(2 * j * rows));
        M[j].y = (2 * j + 1 < COLS) ? bld(r, v + 4u * (uint32_t)((2 * j + 1) * rows)) : 0.f;
    }
}
// the TRANSPOSE of a column-major ROWS x COLS block (ROWS = leading dimension): lane lr gets column lr of the block (4 ROWS contiguous
// bytes), its first COLS... entries 0..NE-1
template <int NE>
__device__ __forceinline__ void load_rows_t(f2 (&M)[np(NE)], rsrc_t r, uint32_t off, int ncols, int lr) {
    const uint32_t v = off + 4u * (uint32_t)NE * (uint32_t)(lr < ncols ? lr : ncols - 1);
#pragma unroll
    for (int j = 0; j < np(NE); ++j) {
        M[j].x = bld(r, v + 8u * j);
        M[j].y = (2 * j + 1 < NE) ? bld(r, v + 8u * j + 4u) : 0.f;
    }
}
template <int COLS>
__device__ __forceinline__ void store_rows(const f2 (&M)[np(COLS)], rsrc_t r, uint32_t off, int rows, int lr, bool on, float mult) {
    const uint32_t v = on ? off + 4u * (uint32_t)lr : SW_OOB;
#pragma unroll
    for (int j = 0; j < np(COLS); ++j) {
        bst(r, v + 4u * (uint32_t)(2 * j * rows), M[j].x * mult);
        if (2 * j + 1 < COLS) bst(r, v + 4u * (uint32_t)((2 * j + 1) * rows), M[j].y * mult);
    }
}
// row lr of M becomes COLUMN lr of the destination block (a transposed store: 4 COLS contiguous bytes per lane)
template <int COLS>
__device__ __forceinline__ void store_rows_t(const f2 (&M)[np(COLS)], rsrc_t r, uint32_t off, int lr, bool on, float mult) {
    const uint32_t v = on ? off + 4u * (uint32_t)COLS * (uint32_t)lr : SW_OOB;
#pragma unroll
    for (int j = 0; j < np(COLS); ++j) {
        bst(r, v + 8u * j, M[j].x * mult);
        if (2 * j + 1 < COLS) bst(r, v + 8u * j + 4u, M[j].y * mult);
    }
}
// ---- stores through an LDS stage (round 4b).  A block held rows-in-lanes leaves the wavefront as 14 dword stores whose 64 lanes touch four
// 56-byte segments each — 150 store instructions per block row, and the walking kernel turned out to be bound by exactly that: the memory
// pipeline takes them one instruction at a time, later wavefronts queue behind earlier ones, and the next row's loads wait behind them all
// (tools/_prof/ab_walk.sh: with the stores compiled out 354 -> 269 us, with all I/O out 265).  So a group writes its block into its slot of
// the wavefront's LDS stage in memory order (14 ds_write_b32, conflict-free), and the WHOLE wavefront then copies each group's slot out in
// 16-byte pieces: 49 lanes x dwordx4 per 14x14 block — 4 store instructions per block (one per group) instead of 14, every one a run of
// 784 contiguous bytes.  The group's destination offset travels to the other lanes as an SGPR (v_readlane -> the store's soffset).
// No barrier: one wavefront per workgroup, LDS operations of a wavefront execute in
// order; the wavefront-scope fences only keep the COMPILER from reordering a lane's reads against the other lanes' writes. ----
typedef __attribute__((address_space(3))) float lds_f;
constexpr int SW_SLOT = 392;                 // floats per group in the stage: two 14x14 blocks
__device__ __forceinline__ void stage_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// rows-in-lanes block -> column-major ROWS x COLS at float offset o of this group's slot
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_put(lds_f* slot, int o, const f2 (&M)[np(COLS)], int lr, float mult) {
    if (lr < ROWS) {
#pragma unroll
        for (int j = 0; j < np(COLS); ++j) {
            slot[o + 2 * j * ROWS + lr] = M[j].x * mult;
            if (2 * j + 1 < COLS) slot[o + (2 * j + 1) * ROWS + lr] = M[j].y * mult;
        }
    }
}
// the same block TRANSPOSED: row lr becomes column lr of the staged COLS x COLS block
template <int COLS>
__device__ __forceinline__ void stage_put_t(lds_f* slot, int o, const f2 (&M)[np(COLS)], int lr, float mult) {
    if (lr < COLS) {
#pragma unroll
        for (int j = 0; j < np(COLS); ++j) {
            slot[o + lr * COLS + 2 * j] = M[j].x * mult;
            if (2 * j + 1 < COLS) slot[o + lr * COLS + 2 * j + 1] = M[j].y * mult;
        }
    }
}
// copy the first NFL floats of every group's slot to that group's destination: dst = byte offset in r (uniform inside a 16-lane group),
// SW_OOB = this group stores nothing.  GB = bytes per lane and store: 16 (destination 16-byte aligned... or not: the hardware takes
// dword-aligned 16-byte stores), or 4.
template <int NFL, int GB>
__device__ __forceinline__ void stage_flush(lds_f* stage, rsrc_t r, uint32_t dst, int lane) {
    constexpr int NG = NFL * 4 / GB;                     // pieces per group
    constexpr int NR = (NG + 63) / 64;                   // store instructions per group
    static_assert(NFL * 4 % GB == 0, "whole pieces");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u4 lds_u4;
    stage_fence();
    // all LDS reads first (one round trip), then the stores.  The soffset operand is not bounds-checked (only voffset is): a disabled
    // group's lanes get the out-of-range voffset, like the lanes beyond the last piece.
    u4 v16[4][NR];
    float v4[4][NR];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            lds_f* src = stage + g * SW_SLOT + (q * 64 * GB) / 4;
            if constexpr (GB == 16) v16[g][q] = *(lds_u4*)(src + lane * 4);
            else v4[g][q] = src[lane];
        }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t sd = (uint32_t)__builtin_amdgcn_readlane((int)dst, 16 * g);
        const bool en = sd != SW_OOB;                    // (uniform)
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int idx = q * 64 + lane;
            const uint32_t vo = (en && (q * 64 + 64 <= NG || idx < NG)) ? (uint32_t)(idx * GB) : SW_OOB;
#if SW_ABLATE & 1
            if constexpr (GB == 16) asm volatile("" :: "v"(v16[g][q]), "v"(vo)); else asm volatile("" :: "v"(v4[g][q]), "v"(vo));
#else
#if SW_ABLATE & 16      // every store lands in the first 64 KB of its array (L2-resident): the issue side of the stores without their HBM side
            const uint32_t sd_ = sd & 0xFFF0u;
#else
            const uint32_t sd_ = sd;
#endif
            if constexpr (GB == 16) __builtin_amdgcn_raw_buffer_store_b128(v16[g][q], r, (int)vo, (int)(en ? sd_ : 0u), SW_NT_STORE);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v4[g][q]), r, (int)vo, (int)(en ? sd_ : 0u), SW_NT_STORE);
#endif
        }
    }
    stage_fence();
}

template <int NN>
__device__ __forceinline__ void add_rho(f2 (&M)[np(NN)], int lr, float rho) {
#pragma unroll
    for (int j = 0; j < np(NN); ++j)         // x + (-0.0) has the bits of x for every x: one packed add per pair, no branches
        M[j] = M[j] + f2{lr == 2 * j ? rho : -0.f, lr == 2 * j + 1 ? rho : -0.f};
}

struct WalkArgs {
    SchurArgs s;
    float* seam_qinv;       // [batch][chunks][196]: (Q_{k0-1} + rho I)^-1 of chunks j >= 1 (schur_seam_kernel stores them into G)
    int L;                  // block rows per chunk
    int chunks;             // ceil((N - 1) / L) >= 1
};

// One 16-lane row = one chunk: block rows k0 = 1 + j L ... k1 - 1 of trajectory b; chunk 0 also emits block row 0 (linsys_setup.cuh:152-277),
// which needs nothing but Q_0.  Four chunks per wavefront, in lock-step.  (Host: every array below 2^31 bytes.)
#ifndef SW_WAVES
#define SW_WAVES 2         // launch bound: wavefronts per SIMD the register allocation aims at
#endif
__global__ __launch_bounds__(64, SW_WAVES) void schur_walk_kernel(WalkArgs w) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m, nm = n * m;
    constexpr uint32_t Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const SchurArgs& a = w.s;
    const int N = a.N, L = w.L, chunks = w.chunks;
    const uint32_t Gsz = Gset * (uint32_t)N - mm, Csz = Cset * (uint32_t)(N - 1), gsz = gset * (uint32_t)N - m;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv_out, (size_t)B * Gsz * 4), rC = make_rsrc(a.C, (size_t)B * Csz * 4), rg = make_rsrc(a.g, (size_t)B * gsz * 4),
                 rc = make_rsrc(a.c, (size_t)B * n * N * 4), rS = make_rsrc(a.S, (size_t)B * 3 * nn * N * 4),
                 rP = make_rsrc(a.Pinv, a.pinv ? (size_t)B * 3 * nn * N * 4 : 0), rgam = make_rsrc(a.gamma, (size_t)B * n * N * 4),
                 rQ = make_rsrc(w.seam_qinv, (size_t)B * chunks * nn * 4);
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const bool r14 = lr < n, r7 = lr < m;
    const uint32_t l14 = 4u * (r14 ? lr : n - 1), l7 = 4u * (r7 ? lr : m - 1);
    __shared__ float sStage[4 * SW_SLOT];                   // the store stage (6,272 B per wavefront)
    lds_f* stage = (lds_f*)sStage;
    const unsigned items = B * (unsigned)chunks;
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;        // dead rows redo the last item and store nothing
        const uint32_t b = it / (unsigned)chunks, j = it % (unsigned)chunks;
        const int k0 = 1 + (int)j * L;
        const int k1 = (k0 + L < N) ? k0 + L : N;
        // byte offsets of trajectory b
        const uint32_t oG = b * Gsz * 4u, oC = b * Csz * 4u, og = b * gsz * 4u, oc = b * (uint32_t)(n * N) * 4u, oS = b * (3u * nn * (uint32_t)N) * 4u;
        const bool st14 = live && r14;

        // ---- prologue: (Q_{k0-1} + rho I)^-1; for chunk 0 that is block row 0 ----
        f2 Qi[np(n)];          // carried: (Q_{k-1} + rho I)^-1
        f2 Tm[np(n)];          // carried: theta_{k-1}^-1 (un-negated; for k-1 = 0: Q_0 + rho I, i.e. -Pinv[0,1])
        {
            f2 Qa[np(n)];
            load_rows<n>(Qa, rG, oG + (uint32_t)(k0 - 1) * Gset * 4u, n, lr);
            add_rho<n>(Qa, lr, a.rho);
#pragma unroll
            for (int q = 0; q < np(n); ++q) Tm[q] = Qa[q];
            const bool first = j == 0;
            store_rows<n>(Qa, rP, oS + nn * 4u, n, lr, st14 && first, -1.f);                      // Pinv[0,1] = -(Q0 + rho I)   :201-210  (no Pinv: zero-size resource)
            invert<n>(Qa, Qi, lr);                                                                // :356-368
            const float q0 = bld(rg, og + l14);
            const float g0 = matvec<n>(Qi, q0);                                                   // :259-264
            store_rows<n>(Qi, rS, oS + nn * 4u, n, lr, st14 && first, -1.f);                      // S[0,1] = -Q0^-1             :248-255
            bst(rgam, (st14 && first) ? oc + 4u * lr : SW_OOB, -g0);                              // :272-276
            // G <- G^-1 (:371-380): Q_0 in place (nobody else reads it); the other chunks' first inverse belongs to the left neighbour's
            // last knot, which that neighbour still reads raw — it goes to the seam buffer
            store_rows<n>(Qi, rG, oG, n, lr, st14 && first, 1.f);
            store_rows<n>(Qi, rQ, (b * (uint32_t)chunks + j) * nn * 4u, n, lr, st14 && !first, 1.f);
        }
        bool have_tm = j == 0;
        // The operands of a block row are requested one row AHEAD, each set as soon as the registers of the previous set are dead (the raw
        // Q / R right after their inversion, A / B after the last product that reads them): half a row of arithmetic hides the HBM latency
        // without a register of its own.  (Past the chunk's end the requests repeat its last row.)
        f2 Ak[np(n)], Bk[np(m)], Rk[np(m)], Qp[np(n)];                                            // linsys_setup.cuh:318-325
        float qk, rk, qp, ck;
        auto row_of = [&](int s_) -> uint32_t { const int kk_ = k0 + s_; return (uint32_t)(kk_ < k1 ? kk_ : k1 - 1); };
        auto load_QR = [&](uint32_t k_) {
            const uint32_t oGk_ = oG + (k_ - 1) * Gset * 4u;
            load_rows<m>(Rk, rG, oGk_ + nn * 4u, m, lr);
            load_rows<n>(Qp, rG, oGk_ + Gset * 4u, n, lr);
        };
        auto load_AB = [&](uint32_t k_) {
            const uint32_t oCk_ = oC + (k_ - 1) * Cset * 4u;
            load_rows<n>(Ak, rC, oCk_, n, lr);
            load_rows<m>(Bk, rC, oCk_ + nn * 4u, n, lr);
        };
        auto load_vec = [&](uint32_t k_) {
            qk = bld(rg, og + (k_ - 1) * gset * 4u + l14); rk = bld(rg, og + (k_ - 1) * gset * 4u + n * 4u + l7);
            qp = bld(rg, og + k_ * gset * 4u + l14); ck = bld(rc, oc + k_ * n * 4u + l14);
        };
        load_QR(row_of(0)); load_AB(row_of(0)); load_vec(row_of(0));
        // The first row's operands are waited for HERE, all of them.  Without this the compiler's wait-count bookkeeping merges two loop
        // entries — this one, where the requests above are the youngest memory operations, and the back edge, where the same registers were
        // requested BEFORE the previous row's 24 output stores — into the stricter of the two: vmcnt(5) ... vmcnt(0) at the top of every
        // row, i.e. every row also waited for the previous row's stores.  (With it: vmcnt(61) ... — loads only.  No measurable difference
        // in time, profiles/r04_walk_store_side.txt: the stores' cost is on the memory side.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        for (int s = 0; s < L; ++s) {
            const int kk = k0 + s;
            const bool rowl = kk < k1;
            const uint32_t k = row_of(s), kn = row_of(s + 1);    // rows past the chunk's end redo its last row and store nothing
            const bool on14 = st14 && rowl;
            [[maybe_unused]] const bool on7 = live && r7 && rowl;
            const uint32_t oGk = oG + (k - 1) * Gset * 4u, oSk = oS + k * 3u * nn * 4u;
#if SW_STAGE
            // the stage's lane constants (piece offsets, slot addresses) are re-derived from an opaque copy of the lane number in every
            // trip: as loop invariants the compiler keeps two dozen of them in registers across the loop and spills
            int lane_s = lane;
            asm volatile("" : "+v"(lane_s));
            const int lr_s = lane_s & 15;
            lds_f* slot = stage + (lane_s >> 4) * SW_SLOT;
#endif
            add_rho<n>(Qp, lr, a.rho);
            add_rho<m>(Rk, lr, a.rho);
            f2 Qpi[np(n)], Rki[np(m)];
            SW_FENCE();
#if SW_PAIR
            invert_pair<n, m>(Qp, Qpi, Rk, Rki, lr);                                              // :356-368
#else
            invert<n>(Qp, Qpi, lr);
            SW_FENCE();
            invert<m>(Rk, Rki, lr);
#endif
            SW_FENCE();
            // G <- G^-1 (:371-380): R_{k-1} and Q_k are this chunk's own — except the chunk's LAST Q when a right neighbour exists (that one
            // reads it raw in its prologue and hands its inverse to the seam kernel)
#if SW_STAGE
            {
                const bool gl = live && rowl;                                                     // (uniform inside the group)
                stage_put<m, m>(slot, 0, Rki, lr_s, 1.f);
                stage_flush<mm, 4>(stage, rG, gl ? oGk + nn * 4u : SW_OOB, lane_s);
                stage_put<n, n>(slot, 0, Qpi, lr_s, 1.f);
                stage_flush<nn, 16>(stage, rG, (gl && (kk < k1 - 1 || k1 == N)) ? oGk + Gset * 4u : SW_OOB, lane_s);
            }
#else
            store_rows<m>(Rki, rG, oGk + nn * 4u, m, lr, on7, 1.f);
            store_rows<n>(Qpi, rG, oGk + Gset * 4u, n, lr, on14 && (kk < k1 - 1 || k1 == N), 1.f);
#endif
#if SW_PREFETCH
            load_QR(kn);                                                                          // (next row; after this row's in-place stores)
#endif
            f2 phi[np(n)], BR[np(m)];
            gemm_nn<n, n>(Ak, Qi, phi);                                                           // phi = Abar Qi      :397-398
            SW_FENCE();
            gemm_nn<m, m>(Bk, Rki, BR);                                                           // Bbar Ri            :405-406
            SW_FENCE();
            const f2 gv = matvec2<n>(Qpi, qp, phi, qk);                                           // :410-415, 421-426
            float gam = gv.x - ck;                                                                // :416-418
            const float v2 = matvec<m>(BR, rk);                                                   // :431-436
            gam += v2 + gv.y;                                                                     // :441-443
            bst(rgam, on14 ? oc + k * n * 4u + 4u * lr : SW_OOB, -gam);                           // :528-532
#if SW_PREFETCH
            load_vec(kn);
#endif
            f2 theta[np(n)];
            {
                f2 t1[np(n)];
                gemm_nt<n, n>(phi, Ak, theta);                                                    // phi Abar^T         :446-455
                SW_FENCE();
                gemm_nt<m, n>(BR, Bk, t1);                                                        // (Bbar Ri) Bbar^T   :472-481
#pragma unroll
                for (int q = 0; q < np(n); ++q) { theta[q] = theta[q] + Qpi[q]; theta[q] = theta[q] + t1[q]; }   // :466-468, 485-487
            }
            SW_FENCE();
#if SW_PREFETCH
            load_AB(kn);
#endif
#if SW_STAGE
            {
                const bool gl = live && rowl;
                stage_put<n, n>(slot, 0, phi, lr_s, -1.f);                                          // S[k,0]             :490-497
                stage_put<n, n>(slot, nn, theta, lr_s, -1.f);                                       // S[k,1]             :500-507
                stage_flush<2 * nn, 16>(stage, rS, gl ? oSk : SW_OOB, lane_s);
                stage_put_t<n>(slot, 0, phi, lr_s, -1.f);                                           // S[k-1,2] = -phi^T  :536-557
                stage_flush<nn, 16>(stage, rS, gl ? oSk - nn * 4u : SW_OOB, lane_s);
            }
#else
            store_rows<n>(phi, rS, oSk, n, lr, on14, -1.f);                                       // S[k,0]             :490-497
            store_rows<n>(theta, rS, oSk + nn * 4u, n, lr, on14, -1.f);                           // S[k,1]             :500-507
            store_rows_t<n>(phi, rS, oSk - nn * 4u, lr, on14, -1.f);                              // S[k-1,2] = -phi^T  :536-557
#endif
#pragma unroll
            for (int q = 0; q < np(n); ++q) Qi[q] = Qpi[q];
            if (a.pinv) {                                                                         // (uniform)
                f2 Ti[np(n)];
                SW_FENCE();
                invert<n>(theta, Ti, lr);                                                         // :510-514
                SW_FENCE();
#if SW_STAGE
                stage_put<n, n>(slot, 0, Ti, lr_s, -1.f);                                           // Pinv[k,1] = -theta^-1   :517-524
                stage_flush<nn, 16>(stage, rP, (live && rowl) ? oSk + nn * 4u : SW_OOB, lane_s);
#else
                store_rows<n>(Ti, rP, oSk + nn * 4u, n, lr, on14, -1.f);                          // Pinv[k,1] = -theta^-1   :517-524
#endif
                if (a.ss) {                                                                       // (uniform)  :9-137
                    // stored blocks are D = -theta^-1, L = -phi; the reference forms -(D_k L_k) D_{k-1} and -(D_{k-1} L_k^T) D_k from the stored
                    // (negated) blocks: the three sign flips cancel exactly, so the un-negated operands give the stored values directly
                    f2 t1[np(n)], t2[np(n)];
                    gemm_nn<n, n>(Ti, phi, t1);                                                   // Dk L            :100
                    SW_FENCE();
                    gemm_nn<n, n>(t1, Tm, t2);                                                    // (Dk L) Dm       :102
                    SW_FENCE();
#if SW_STAGE
                    stage_put<n, n>(slot, 0, t2, lr_s, 1.f);                                        // Pinv[k,0]       :106-113
                    stage_flush<nn, 16>(stage, rP, (live && rowl && have_tm) ? oSk : SW_OOB, lane_s);
#else
                    store_rows<n>(t2, rP, oSk, n, lr, on14 && have_tm, 1.f);                      // Pinv[k,0]       :106-113
#endif
                    gemm_nt<n, n>(Tm, phi, t1);                                                   // Dm phi^T        :121
                    SW_FENCE();
                    gemm_nn<n, n>(t1, Ti, t2);                                                    // (Dm phi^T) Dk   :123
                    SW_FENCE();
#if SW_STAGE
                    stage_put<n, n>(slot, 0, t2, lr_s, 1.f);                                        // Pinv[k-1,2]     :127-134
                    stage_flush<nn, 16>(stage, rP, (live && rowl && have_tm) ? oSk - nn * 4u : SW_OOB, lane_s);
#else
                    store_rows<n>(t2, rP, oSk - nn * 4u, n, lr, on14 && have_tm, 1.f);            // Pinv[k-1,2]     :127-134
#endif
                }
#pragma unroll
                for (int q = 0; q < np(n); ++q) Tm[q] = Ti[q];
            }
            have_tm = true;
#if !SW_PREFETCH
            load_QR(kn); load_AB(kn); load_vec(kn);
#endif
        }
    }
}

// The seams: for chunk j >= 1 of every trajectory, with k0 = 1 + j L — G[k0-1].Q <- the inverse the chunk's prologue computed, and (SS) the
// two coupling blocks across the seam, Pinv[k0,0] and Pinv[k0-1,2], from the stored blocks (linsys_setup.cuh:97-136).
__global__ __launch_bounds__(64, 2) void schur_seam_kernel(WalkArgs w) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m;
    constexpr uint32_t Gset = nn + mm;
    const SchurArgs& a = w.s;
    const int N = a.N, L = w.L, chunks = w.chunks;
    const uint32_t Gsz = Gset * (uint32_t)N - mm;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv_out, (size_t)B * Gsz * 4), rS = make_rsrc(a.S, (size_t)B * 3 * nn * N * 4),
                 rP = make_rsrc(a.Pinv, a.pinv ? (size_t)B * 3 * nn * N * 4 : 0), rQ = make_rsrc(w.seam_qinv, (size_t)B * chunks * nn * 4);
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const bool r14 = lr < n;
    const unsigned per = (unsigned)(chunks - 1);
    const unsigned items = B * per;
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;
        const uint32_t b = it / per, j = 1 + it % per;
        const uint32_t k0 = 1 + j * (uint32_t)L;
        const uint32_t oS = b * (3u * nn * (uint32_t)N) * 4u, oSk = oS + k0 * 3u * nn * 4u;
        {
            const uint32_t src = (b * (uint32_t)chunks + j) * nn * 4u, dst = b * Gsz * 4u + (k0 - 1) * Gset * 4u;
            for (uint32_t e = lr; e < nn; e += 16) bst(rG, live ? dst + 4u * e : SW_OOB, bld(rQ, src + 4u * e));
        }
        if (!a.ss) continue;
        // stored blocks: Dk = Pinv[k0,1], Dm = Pinv[k0-1,1], Lk = S[k0,0].  Pinv[k0,0] = -((Dk Lk) Dm), Pinv[k0-1,2] = -((Dm Lk^T) Dk).
        // Second operands are loaded transposed and used through gemm_nt: the sums of A B in the same order.
        f2 Dk[np(n)], Dm[np(n)], t1[np(n)], t2[np(n)];
        load_rows<n>(Dk, rP, oSk + nn * 4u, n, lr);
        load_rows<n>(Dm, rP, oSk - 2u * nn * 4u, n, lr);
        {
            f2 LT[np(n)], DmT[np(n)];
            load_rows_t<n>(LT, rS, oSk, n, lr);
            load_rows_t<n>(DmT, rP, oSk - 2u * nn * 4u, n, lr);
            gemm_nt<n, n>(Dk, LT, t1);                                                            // Dk L            :100
            SW_FENCE();
            gemm_nt<n, n>(t1, DmT, t2);                                                           // (Dk L) Dm       :102
            store_rows<n>(t2, rP, oSk, n, lr, live && r14, -1.f);                                 // Pinv[k0,0]      :106-113
        }
        SW_FENCE();
        {
            f2 Lk[np(n)], DkT[np(n)];
            load_rows<n>(Lk, rS, oSk, n, lr);
            load_rows_t<n>(DkT, rP, oSk + nn * 4u, n, lr);
            gemm_nt<n, n>(Dm, Lk, t1);                                                            // Dm phi^T        :121
            SW_FENCE();
            gemm_nt<n, n>(t1, DkT, t2);                                                           // (Dm phi^T) Dk   :123
            store_rows<n>(t2, rP, oSk - nn * 4u, n, lr, live && r14, -1.f);                       // Pinv[k0-1,2]    :127-134
        }
    }
}

// dz = G^-1 (g - C^T lambda)  (include/common/dz.cuh:3-121), four knots per wavefront, a 16-lane row per knot:
//   dz_x = Qi (q - (lambda_k + Abar^T lambda_{k+1})), dz_u = Ri (r - Bbar^T lambda_{k+1});  the last knot has neither the A/B terms nor a dz_u.
// Abar^T lambda: lane t takes row t of Abar^T = column t of Abar (56 contiguous bytes), gato_ATx's sum order (matrix.cuh:10-25).
// Round 3 ran one 64-thread workgroup per knot with two block barriers: 128 us per 1024 x 128 knots; this one is a plain HBM stream.
// (Host: N >= 2, every array below 2^31 bytes.)
__global__ __launch_bounds__(64, 4) void compute_dz_dpp_kernel(DzArgs a) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m, nm = n * m;
    constexpr uint32_t Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const int N = a.N;
    const uint32_t Gsz = Gset * (uint32_t)N - mm, Csz = Cset * (uint32_t)(N - 1), gsz = gset * (uint32_t)N - m;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv, (size_t)B * Gsz * 4), rC = make_rsrc(a.C, (size_t)B * Csz * 4), rg = make_rsrc(a.g, (size_t)B * gsz * 4),
                 rl = make_rsrc(a.lambda, (size_t)B * n * N * 4), rz = make_rsrc(a.dz, (size_t)B * gsz * 4);
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const bool r14 = lr < n, r7 = lr < m;
    const uint32_t l14 = 4u * (r14 ? lr : n - 1), l7 = 4u * (r7 ? lr : m - 1);
    const unsigned items = B * (unsigned)N;
    // (one group of four knots per workgroup and trip.  Tried in round 4: spans of 4 / 8 / 16 consecutive groups per wavefront, so that its
    //  84-byte pieces of dz meet in one L2 as whole lines — what gained 25 % in bt_spmv_kernel: 60-63 us against 56.6 here, the serial
    //  trips cost more memory-level parallelism than the partial lines cost.)
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;
        const uint32_t b = it / (unsigned)N, k = it % (unsigned)N;
        const bool last = k == (uint32_t)(N - 1);
        const uint32_t kc = last ? k - 1 : k;                         // the last knot re-reads its neighbour's C / R (results dropped)
        const uint32_t oG = b * Gsz * 4u + k * Gset * 4u, oR = b * Gsz * 4u + kc * Gset * 4u + nn * 4u, oC = b * Csz * 4u + kc * Cset * 4u;
        const uint32_t og = b * gsz * 4u + k * gset * 4u, ol = b * (uint32_t)(n * N) * 4u + k * n * 4u;
        f2 At[np(n)], Bt[np(n)], Qi[np(n)], Ri[np(m)];
        load_rows_t<n>(At, rC, oC, n, lr);                            // lane t: Abar[0..13][t] = Ck[t n + i]
        load_rows_t<n>(Bt, rC, oC + nn * 4u, m, lr);                  // lane j < 7: Bbar[0..13][j]
        load_rows<n>(Qi, rG, oG, n, lr);
        load_rows<m>(Ri, rG, oR, m, lr);
        const float lk = bld(rl, ol + l14);
        const float ln = bld(rl, ol + (last ? 0u : n * 4u) + l14);
        const float gx = bld(rg, og + l14), gu = bld(rg, og + (last ? l14 : n * 4u + l7));
        const f2 acc = matvec2<n>(At, ln, Bt, ln);                    // Abar^T lambda_{k+1} | Bbar^T lambda_{k+1}
        const float ax = last ? 0.f : acc.x;
        const float tx = gx - (lk + ax);
        const float tu = gu - acc.y;
        const float dx = matvec<n>(Qi, tx);
        const float du = matvec<m>(Ri, tu);
        bst(rz, (live && r14) ? og + 4u * lr : SW_OOB, dx);
        bst(rz, (live && r7 && !last) ? og + n * 4u + 4u * lr : SW_OOB, du);
    }
}

#pragma clang fp contract(fast)

}  // namespace sw
}  // namespace mpcg
