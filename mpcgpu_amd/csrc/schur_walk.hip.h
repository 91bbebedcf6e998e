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

// C[r][c] = sum_t A[r][t] * B[t][c]      A: NI columns per lane; B: rows in lanes 0..NI-1, NC columns
template <int NI, int NC>
__device__ __forceinline__ void gemm_nn(const f2 (&A)[np(NI)], const f2 (&Bsrc)[np(NC)], f2 (&Cm)[np(NC)]) {
    f2 B[np(NC)];
    launder(B, Bsrc);
#pragma unroll
    for (int j = 0; j < np(NC); ++j) Cm[j] = f2{0.f, 0.f};
    SFor<0, NI>::run([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        const float a = SW_EL(A, T);
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
    SFor<0, np(NC)>::run([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f2 acc{0.f, 0.f};
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

// Gauss-Jordan on [A | I] without pivoting (include/utils/matrix.cuh:120-238), rows in lanes 0..NN-1: A destroyed, I becomes A^-1.
// lr = lane index inside the 16-lane row.  Half of the 2 NN columns are structurally inert at every pivot step and are skipped (columns of
// A at or left of the pivot: finished, never read again; columns of I right of the pivot: still unit columns, the reference's update leaves
// them as they are) — round 3, no bit changed.  Round 4: the shape described in the file header.
template <int NN>
__device__ __forceinline__ void invert(f2 (&A)[np(NN)], f2 (&I)[np(NN)], int lr) {
    constexpr int NP = np(NN);
#pragma unroll
    for (int j = 0; j < NP; ++j) I[j] = f2{lr == 2 * j ? 1.f : 0.f, lr == 2 * j + 1 ? 1.f : 0.f};
    SFor<0, NN>::run([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        constexpr int JP = P / 2;                    // the pair that holds column P
        constexpr bool ODD = P & 1;
        const float app = SW_EL(A, P);               // pivot column entry of this row (the pivot itself in lane P)
        const bool is_p = lr == P;
        const float nmul = is_p ? 0.f : -app;        // x + (+0)(x) keeps the bits of x: the pivot lane needs no select
        if (is_p) {                                  // EXEC-masked: the pivot row becomes row / pivot   (matrix.cuh:146)
            const float pinv = 1.0f / app;
            // pair JP of A: column P is finished; its neighbour P+1 (P even) is live.  Pair JP of I: column P is live, its neighbour P+1 (P
            // even) is still a unit column and must keep its +0.
            if (!ODD) { if (2 * JP + 1 < NN) A[JP].y = A[JP].y * pinv; }
#pragma unroll
            for (int j = JP + 1; j < NP; ++j) A[j] = A[j] * f2{pinv, pinv};
#pragma unroll
            for (int j = 0; j < JP; ++j) I[j] = I[j] * f2{pinv, pinv};
            if (ODD) I[JP] = I[JP] * f2{pinv, pinv};
            else I[JP].x = I[JP].x * pinv;
        }
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
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 0);
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
template <int MINW>
__global__ __launch_bounds__(64, MINW) void schur_walk_kernel(WalkArgs w) {
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
        for (int s = 0; s < L; ++s) {
            const int kk = k0 + s;
            const bool rowl = kk < k1;
            const uint32_t k = (uint32_t)(rowl ? kk : k1 - 1);   // rows past the chunk's end redo its last row and store nothing
            const bool on14 = st14 && rowl, on7 = live && r7 && rowl;
            const uint32_t oCk = oC + (k - 1) * Cset * 4u, oGk = oG + (k - 1) * Gset * 4u, oSk = oS + k * 3u * nn * 4u;
            f2 Ak[np(n)], Bk[np(m)], Rk[np(m)], Qp[np(n)];                                        // linsys_setup.cuh:318-325
            load_rows<n>(Ak, rC, oCk, n, lr);
            load_rows<m>(Bk, rC, oCk + nn * 4u, n, lr);
            load_rows<m>(Rk, rG, oGk + nn * 4u, m, lr);
            load_rows<n>(Qp, rG, oGk + Gset * 4u, n, lr);
            const float qk = bld(rg, og + (k - 1) * gset * 4u + l14), rk = bld(rg, og + (k - 1) * gset * 4u + n * 4u + l7);
            const float qp = bld(rg, og + k * gset * 4u + l14), ck = bld(rc, oc + k * n * 4u + l14);
            add_rho<n>(Qp, lr, a.rho);
            add_rho<m>(Rk, lr, a.rho);
            f2 Qpi[np(n)], Rki[np(m)];
            SW_FENCE();
            invert<n>(Qp, Qpi, lr);                                                               // :356-368
            SW_FENCE();
            invert<m>(Rk, Rki, lr);
            SW_FENCE();
            // G <- G^-1 (:371-380): R_{k-1} and Q_k are this chunk's own — except the chunk's LAST Q when a right neighbour exists (that one
            // reads it raw in its prologue and hands its inverse to the seam kernel)
            store_rows<m>(Rki, rG, oGk + nn * 4u, m, lr, on7, 1.f);
            store_rows<n>(Qpi, rG, oGk + Gset * 4u, n, lr, on14 && (kk < k1 - 1 || k1 == N), 1.f);
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
            store_rows<n>(phi, rS, oSk, n, lr, on14, -1.f);                                       // S[k,0]             :490-497
            store_rows<n>(theta, rS, oSk + nn * 4u, n, lr, on14, -1.f);                           // S[k,1]             :500-507
            store_rows_t<n>(phi, rS, oSk - nn * 4u, lr, on14, -1.f);                              // S[k-1,2] = -phi^T  :536-557
#pragma unroll
            for (int q = 0; q < np(n); ++q) Qi[q] = Qpi[q];
            if (a.pinv) {                                                                         // (uniform)
                f2 Ti[np(n)];
                SW_FENCE();
                invert<n>(theta, Ti, lr);                                                         // :510-514
                SW_FENCE();
                store_rows<n>(Ti, rP, oSk + nn * 4u, n, lr, on14, -1.f);                          // Pinv[k,1] = -theta^-1   :517-524
                if (a.ss) {                                                                       // (uniform)  :9-137
                    // stored blocks are D = -theta^-1, L = -phi; the reference forms -(D_k L_k) D_{k-1} and -(D_{k-1} L_k^T) D_k from the stored
                    // (negated) blocks: the three sign flips cancel exactly, so the un-negated operands give the stored values directly
                    f2 t1[np(n)], t2[np(n)];
                    gemm_nn<n, n>(Ti, phi, t1);                                                   // Dk L            :100
                    SW_FENCE();
                    gemm_nn<n, n>(t1, Tm, t2);                                                    // (Dk L) Dm       :102
                    SW_FENCE();
                    store_rows<n>(t2, rP, oSk, n, lr, on14 && have_tm, 1.f);                      // Pinv[k,0]       :106-113
                    gemm_nt<n, n>(Tm, phi, t1);                                                   // Dm phi^T        :121
                    SW_FENCE();
                    gemm_nn<n, n>(t1, Ti, t2);                                                    // (Dm phi^T) Dk   :123
                    SW_FENCE();
                    store_rows<n>(t2, rP, oSk - nn * 4u, n, lr, on14 && have_tm, 1.f);            // Pinv[k-1,2]     :127-134
                }
#pragma unroll
                for (int q = 0; q < np(n); ++q) Tm[q] = Ti[q];
            }
            have_tm = true;
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
