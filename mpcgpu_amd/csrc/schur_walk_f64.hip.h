// schur_walk_f64.hip.h — round 5: linsys_t = double (USE_DOUBLES = 1, include/common/settings.cuh:41-49) twins of the walking Schur +
// preconditioner formation (include/pcg/linsys_setup.cuh:139-656) and of the rows-in-lanes dz recovery (include/common/dz.cuh:3-136).  gfx950.
//
// Same design as schur_walk.hip.h (a 16-lane DPP row walks a chunk of consecutive block rows of one trajectory and carries
// (Q_{k-1} + rho I)^-1 and theta_{k-1}^-1 in registers; a seam kernel closes the chunk boundaries), same operations in the same order — every
// product a rounded multiply followed by a rounded add, sequential over the contracted index, accumulators starting at +0, Gauss-Jordan without
// pivoting — so the results are the bits of the C oracle's double instantiation (tests/test_gpu_f64.py compares bits).  What differs from float:
//   * a 64-bit operand is broadcast inside its row by `v_mov_b64_dpp ... row_newbcast:t` (the only DPP control 64-bit operations have; the
//     compiler does not produce it — it splits a 64-bit broadcast into two v_mov_b32_dpp — so it is written as assembler text, and the DPP read-after-VALU-write hazard, two wait states, is ours to keep: SW64_SETTLE() puts one s_nop 1
//     between the writers of a to-be-broadcast operand and its DPP readers; tools/check_dpp_hazards.py verifies the built code);
//   * no DPP source modifier on the 64-bit multiply and no packed 64-bit add: 3 VALU instructions per multiply-add (float: 1.5);
//   * operands take two registers each: one wavefront per SIMD (launch bound 1: up to 512 registers, the upper 256 as AGPR spill space).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "schur_walk.hip.h"

namespace mpcg {
namespace sw64 {

#pragma clang fp contract(off)

using sw::make_rsrc;
using sw::rsrc_t;
using sw::SFor;
using sw::SW_OOB;

// a * (b held by lane L of this lane's 16-lane row): v_mov_b64_dpp + a rounded multiply.  (gfx950 has DPP forms of v_mov_b64 and v_fmac_f64
// only — v_mul_f64 / v_add_f64 are VOP3 — and the fused multiply-add is not what the oracle computes.)
template <int L>
__device__ __forceinline__ double mulbc(double a, double b) {
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "n"(L));
    return a * r;
}
#define SW64_FENCE() __builtin_amdgcn_sched_barrier(0)
// Two wait states between the last VALU write of an operand and its first DPP read: nothing is scheduled across this point, and the s_nop 1
// stands between whatever wrote the operands before it and the DPP multiplies behind it (verified on the built code: tools/check_dpp_hazards.py).
#define SW64_SETTLE() do { SW64_FENCE(); asm volatile("s_nop 1"); SW64_FENCE(); } while (0)

// A private copy of a broadcast operand that the optimiser cannot identify with the original (as sw::launder): two products that broadcast
// the same entries of the same operand (Dk L and Dm L^T) would otherwise share ONE v_mov_b64_dpp per entry, kept alive from the first product
// to the second — 392 registers.
template <int NC>
__device__ __forceinline__ void launder(double (&D)[NC], const double (&Src)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { D[c] = Src[c]; asm volatile("" : "+v"(D[c])); }
}

// C[r][c] = sum_t A[r][t] * B[t][c]      A: NI columns per lane; B: rows in lanes 0..NI-1, NC columns
template <int NI, int NC>
__device__ __forceinline__ void gemm_nn(const double (&A)[NI], const double (&Bsrc)[NC], double (&Cm)[NC]) {
    double B[NC];
    launder(B, Bsrc);
    SW64_SETTLE();
#pragma unroll
    for (int c = 0; c < NC; ++c) Cm[c] = 0.0;
    SFor<0, NI>::run([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if constexpr (T > 0) SW64_FENCE();          // the products of term T start after the sums of term T-1: at most NC products in flight
#pragma unroll
        for (int c = 0; c < NC; ++c) Cm[c] = Cm[c] + mulbc<T>(A[T], B[c]);
    });
}
// C[r][c] = sum_t A[r][t] * Bt[c][t]     Bt: row c in lane c (NC rows), NI columns
template <int NI, int NC>
__device__ __forceinline__ void gemm_nt(const double (&A)[NI], const double (&Btsrc)[NI], double (&Cm)[NC]) {
    double Bt[NI];
    launder(Bt, Btsrc);
    SW64_SETTLE();
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        if constexpr (Cc > 0 && Cc % 2 == 0) SW64_FENCE();      // two columns' chains interleave; the next pair starts after them
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < NI; ++t) acc = acc + mulbc<Cc>(A[t], Bt[t]);
        Cm[Cc] = acc;
    });
}
// out[r] = sum_c M[r][c] * v[c]          v: element c in lane c
template <int NC>
__device__ __forceinline__ double matvec(const double (&M)[NC], double v) {
    SW64_SETTLE();
    double acc = 0.0;
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        acc = acc + mulbc<Cc>(M[Cc], v);
    });
    return acc;
}

// One pivot step of the Gauss-Jordan elimination of [A | I] (include/utils/matrix.cuh:120-238), rows in lanes, as sw::gj_step: columns of A at
// or left of the pivot and columns of I right of it are inert and skipped; the pivot row is scaled by a per-lane multiplier (1 / pivot in the
// pivot lane, exactly 1.0 elsewhere), then every lane adds (-pcol) x (pivot row entry) with +0.0 as the pivot lane's multiplier.
template <int NN, int P>
__device__ __forceinline__ void gj_step(double (&A)[NN], double (&I)[NN], int lr) {
    const double app = A[P];
    const bool is_p = lr == P;
    const double nmul = is_p ? 0.0 : -app;
    const double mrow = is_p ? 1.0 / app : 1.0;            // (matrix.cuh:146: the IEEE quotient)
#pragma unroll
    for (int c = P + 1; c < NN; ++c) A[c] = A[c] * mrow;
#pragma unroll
    for (int c = 0; c <= P; ++c) I[c] = I[c] * mrow;
    SW64_SETTLE();
#pragma unroll
    for (int c = P + 1; c < NN; ++c) A[c] = A[c] + mulbc<P>(nmul, A[c]);
#pragma unroll
    for (int c = 0; c <= P; ++c) I[c] = I[c] + mulbc<P>(nmul, I[c]);
    SW64_FENCE();
}
// A destroyed, I becomes A^-1
template <int NN>
__device__ __forceinline__ void invert(double (&A)[NN], double (&I)[NN], int lr) {
#pragma unroll
    for (int c = 0; c < NN; ++c) I[c] = lr == c ? 1.0 : 0.0;
    SFor<0, NN>::run([&](auto pc) { gj_step<NN, decltype(pc)::value>(A, I, lr); });
}

// ---- memory: buffer resources, one 32-bit byte offset per lane (as schur_walk.hip.h); SW_OOB drops a store ----
__device__ __forceinline__ double bld(rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, uint32_t off, double v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (int)off, 0, 0);
}
// row lr of a column-major rows x COLS matrix that starts at byte `off`; rows beyond the matrix repeat its last row (never stored)
template <int COLS>
__device__ __forceinline__ void load_rows(double (&M)[COLS], rsrc_t r, uint32_t off, int rows, int lr) {
    const uint32_t v = off + 8u * (uint32_t)(lr < rows ? lr : rows - 1);
#pragma unroll
    for (int c = 0; c < COLS; ++c) M[c] = bld(r, v + 8u * (uint32_t)(c * rows));
}
// the TRANSPOSE of a column-major block with leading dimension NE: lane lr gets column lr (8 NE contiguous bytes)
template <int NE>
__device__ __forceinline__ void load_rows_t(double (&M)[NE], rsrc_t r, uint32_t off, int ncols, int lr) {
    const uint32_t v = off + 8u * (uint32_t)NE * (uint32_t)(lr < ncols ? lr : ncols - 1);
#pragma unroll
    for (int c = 0; c < NE; ++c) M[c] = bld(r, v + 8u * c);
}
template <int COLS>
__device__ __forceinline__ void store_rows(const double (&M)[COLS], rsrc_t r, uint32_t off, int rows, int lr, bool on, double mult) {
    const uint32_t v = on ? off + 8u * (uint32_t)lr : SW_OOB;
#pragma unroll
    for (int c = 0; c < COLS; ++c) bst(r, v + 8u * (uint32_t)(c * rows), M[c] * mult);
}
// row lr of M becomes COLUMN lr of the destination block (8 COLS contiguous bytes per lane)
template <int COLS>
__device__ __forceinline__ void store_rows_t(const double (&M)[COLS], rsrc_t r, uint32_t off, int lr, bool on, double mult) {
    const uint32_t v = on ? off + 8u * (uint32_t)COLS * (uint32_t)lr : SW_OOB;
#pragma unroll
    for (int c = 0; c < COLS; ++c) bst(r, v + 8u * c, M[c] * mult);
}
template <int NN>
__device__ __forceinline__ void add_rho(double (&M)[NN], int lr, double rho) {
#pragma unroll
    for (int c = 0; c < NN; ++c) M[c] = M[c] + (lr == c ? rho : -0.0);      // x + (-0.0) has the bits of x
}

struct WalkArgs64 {
    SchurArgsT<double> s;
    double* seam_qinv;      // [batch][chunks][196]
    int L;                  // block rows per chunk
    int chunks;             // ceil((N - 1) / L) >= 1
};

// One 16-lane row = one chunk: block rows k0 = 1 + j L ... k1 - 1 of trajectory b; chunk 0 also emits block row 0.  Four chunks per
// wavefront, in lock-step.  (Host: every array below 2^31 bytes.)
__global__ __launch_bounds__(64, 1) void schur_walk_f64_kernel(WalkArgs64 w) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m, nm = n * m, E = 8;
    constexpr uint32_t Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const SchurArgsT<double>& a = w.s;
    const int N = a.N, L = w.L, chunks = w.chunks;
    const uint32_t Gsz = Gset * (uint32_t)N - mm, Csz = Cset * (uint32_t)(N - 1), gsz = gset * (uint32_t)N - m;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv_out, (size_t)B * Gsz * E), rC = make_rsrc(a.C, (size_t)B * Csz * E), rg = make_rsrc(a.g, (size_t)B * gsz * E),
                 rc = make_rsrc(a.c, (size_t)B * n * N * E), rS = make_rsrc(a.S, (size_t)B * 3 * nn * N * E),
                 rP = make_rsrc(a.Pinv, a.pinv ? (size_t)B * 3 * nn * N * E : 0), rgam = make_rsrc(a.gamma, (size_t)B * n * N * E),
                 rQ = make_rsrc(w.seam_qinv, (size_t)B * chunks * nn * E);
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const bool r14 = lr < n, r7 = lr < m;
    const uint32_t l14 = E * (r14 ? lr : n - 1), l7 = E * (r7 ? lr : m - 1);
    const unsigned items = B * (unsigned)chunks;
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;        // dead rows redo the last item and store nothing
        const uint32_t b = it / (unsigned)chunks, j = it % (unsigned)chunks;
        const int k0 = 1 + (int)j * L;
        const int k1 = (k0 + L < N) ? k0 + L : N;
        const uint32_t oG = b * Gsz * E, oC = b * Csz * E, og = b * gsz * E, oc = b * (uint32_t)(n * N) * E, oS = b * (3u * nn * (uint32_t)N) * E;
        const bool st14 = live && r14;

        // ---- prologue: (Q_{k0-1} + rho I)^-1; for chunk 0 that is block row 0 (linsys_setup.cuh:152-277) ----
        double Qi[n];          // carried: (Q_{k-1} + rho I)^-1
        double Tm[n];          // carried: theta_{k-1}^-1 (un-negated; for k-1 = 0: Q_0 + rho I, i.e. -Pinv[0,1])
        {
            double Qa[n];
            load_rows<n>(Qa, rG, oG + (uint32_t)(k0 - 1) * Gset * E, n, lr);
            add_rho<n>(Qa, lr, a.rho);
#pragma unroll
            for (int q = 0; q < n; ++q) Tm[q] = Qa[q];
            const bool first = j == 0;
            store_rows<n>(Qa, rP, oS + nn * E, n, lr, st14 && first, -1.0);                       // Pinv[0,1] = -(Q0 + rho I)   :201-210
            invert<n>(Qa, Qi, lr);                                                                // :356-368
            const double q0 = bld(rg, og + l14);
            const double g0 = matvec<n>(Qi, q0);                                                  // :259-264
            store_rows<n>(Qi, rS, oS + nn * E, n, lr, st14 && first, -1.0);                       // S[0,1] = -Q0^-1             :248-255
            bst(rgam, (st14 && first) ? oc + E * lr : SW_OOB, -g0);                               // :272-276
            store_rows<n>(Qi, rG, oG, n, lr, st14 && first, 1.0);                                 // G <- G^-1 (:371-380): Q_0 in place
            store_rows<n>(Qi, rQ, (b * (uint32_t)chunks + j) * nn * E, n, lr, st14 && !first, 1.0);   // the other chunks': to the seam buffer
        }
        bool have_tm = j == 0;
        double Ak[n], Bk[m], Rk[m], Qp[n];                                                        // linsys_setup.cuh:318-325
        double qk, rk, qp, ck;
        auto row_of = [&](int s_) -> uint32_t { const int kk_ = k0 + s_; return (uint32_t)(kk_ < k1 ? kk_ : k1 - 1); };
        auto load_QR = [&](uint32_t k_) {
            const uint32_t oGk_ = oG + (k_ - 1) * Gset * E;
            load_rows<m>(Rk, rG, oGk_ + nn * E, m, lr);
            load_rows<n>(Qp, rG, oGk_ + Gset * E, n, lr);
        };
        auto load_AB = [&](uint32_t k_) {
            const uint32_t oCk_ = oC + (k_ - 1) * Cset * E;
            load_rows<n>(Ak, rC, oCk_, n, lr);
            load_rows<m>(Bk, rC, oCk_ + nn * E, n, lr);
        };
        auto load_vec = [&](uint32_t k_) {
            qk = bld(rg, og + (k_ - 1) * gset * E + l14); rk = bld(rg, og + (k_ - 1) * gset * E + n * E + l7);
            qp = bld(rg, og + k_ * gset * E + l14); ck = bld(rc, oc + k_ * n * E + l14);
        };
        load_QR(row_of(0)); load_AB(row_of(0)); load_vec(row_of(0));
        for (int s = 0; s < L; ++s) {
            const int kk = k0 + s;
            const bool rowl = kk < k1;
            const uint32_t k = row_of(s), kn = row_of(s + 1);    // rows past the chunk's end redo its last row and store nothing
            const bool on14 = st14 && rowl;
            const bool on7 = live && r7 && rowl;
            const uint32_t oGk = oG + (k - 1) * Gset * E, oSk = oS + k * 3u * nn * E;
            add_rho<n>(Qp, lr, a.rho);
            add_rho<m>(Rk, lr, a.rho);
            double Qpi[n], Rki[m];
            SW64_FENCE();
            invert<n>(Qp, Qpi, lr);                                                               // :356-368
            SW64_FENCE();
            invert<m>(Rk, Rki, lr);
            SW64_FENCE();
            // G <- G^-1 (:371-380): R_{k-1} and Q_k are this chunk's own — except the chunk's LAST Q when a right neighbour exists
            store_rows<m>(Rki, rG, oGk + nn * E, m, lr, on7, 1.0);
            store_rows<n>(Qpi, rG, oGk + Gset * E, n, lr, on14 && (kk < k1 - 1 || k1 == N), 1.0);
            load_QR(kn);                                                                          // (next row; after this row's in-place stores)
            double phi[n], BR[m];
            gemm_nn<n, n>(Ak, Qi, phi);                                                           // phi = Abar Qi      :397-398
            SW64_FENCE();
            gemm_nn<m, m>(Bk, Rki, BR);                                                           // Bbar Ri            :405-406
            SW64_FENCE();
            const double gx = matvec<n>(Qpi, qp), gy = matvec<n>(phi, qk);                        // :410-415, 421-426
            double gam = gx - ck;                                                                 // :416-418
            const double v2 = matvec<m>(BR, rk);                                                  // :431-436
            gam += v2 + gy;                                                                       // :441-443
            bst(rgam, on14 ? oc + k * n * E + E * lr : SW_OOB, -gam);                             // :528-532
            load_vec(kn);
            double theta[n];
            {
                double t1[n];
                gemm_nt<n, n>(phi, Ak, theta);                                                    // phi Abar^T         :446-455
                SW64_FENCE();
                gemm_nt<m, n>(BR, Bk, t1);                                                        // (Bbar Ri) Bbar^T   :472-481
#pragma unroll
                for (int q = 0; q < n; ++q) { theta[q] = theta[q] + Qpi[q]; theta[q] = theta[q] + t1[q]; }   // :466-468, 485-487
            }
            SW64_FENCE();
            store_rows<n>(phi, rS, oSk, n, lr, on14, -1.0);                                       // S[k,0]             :490-497
            store_rows<n>(theta, rS, oSk + nn * E, n, lr, on14, -1.0);                            // S[k,1]             :500-507
            store_rows_t<n>(phi, rS, oSk - nn * E, lr, on14, -1.0);                               // S[k-1,2] = -phi^T  :536-557
#pragma unroll
            for (int q = 0; q < n; ++q) Qi[q] = Qpi[q];
            if (a.pinv) {                                                                         // (uniform)
                double Ti[n];
                SW64_FENCE();
                invert<n>(theta, Ti, lr);                                                         // :510-514
                SW64_FENCE();
                store_rows<n>(Ti, rP, oSk + nn * E, n, lr, on14, -1.0);                           // Pinv[k,1] = -theta^-1   :517-524
                if (a.ss) {                                                                       // (uniform)  :9-137
                    // stored blocks are D = -theta^-1, L = -phi; the three sign flips of the reference's -(D_k L_k) D_{k-1} cancel exactly
                    double t1[n], t2[n];
                    gemm_nn<n, n>(Ti, phi, t1);                                                   // Dk L            :100
                    SW64_FENCE();
                    gemm_nn<n, n>(t1, Tm, t2);                                                    // (Dk L) Dm       :102
                    SW64_FENCE();
                    store_rows<n>(t2, rP, oSk, n, lr, on14 && have_tm, 1.0);                      // Pinv[k,0]       :106-113
                    gemm_nt<n, n>(Tm, phi, t1);                                                   // Dm phi^T        :121
                    SW64_FENCE();
                    gemm_nn<n, n>(t1, Ti, t2);                                                    // (Dm phi^T) Dk   :123
                    SW64_FENCE();
                    store_rows<n>(t2, rP, oSk - nn * E, n, lr, on14 && have_tm, 1.0);             // Pinv[k-1,2]     :127-134
                }
#pragma unroll
                for (int q = 0; q < n; ++q) Tm[q] = Ti[q];
            }
            have_tm = true;
            // next row's A / B only now: requested any earlier they would be live across the symmetric-stair products, the register peak of a row
            // (130 doubles); the two inversions at the head of the next row hide their latency
            load_AB(kn);
        }
    }
}

// The seams (as sw::schur_seam_kernel): G[k0-1].Q <- the inverse the chunk's prologue computed, and (SS) the two coupling blocks across the
// seam from the stored blocks (linsys_setup.cuh:97-136).
__global__ __launch_bounds__(64, 1) void schur_seam_f64_kernel(WalkArgs64 w) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m, E = 8;
    constexpr uint32_t Gset = nn + mm;
    const SchurArgsT<double>& a = w.s;
    const int N = a.N, L = w.L, chunks = w.chunks;
    const uint32_t Gsz = Gset * (uint32_t)N - mm;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv_out, (size_t)B * Gsz * E), rS = make_rsrc(a.S, (size_t)B * 3 * nn * N * E),
                 rP = make_rsrc(a.Pinv, a.pinv ? (size_t)B * 3 * nn * N * E : 0), rQ = make_rsrc(w.seam_qinv, (size_t)B * chunks * nn * E);
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
        const uint32_t oS = b * (3u * nn * (uint32_t)N) * E, oSk = oS + k0 * 3u * nn * E;
        {
            const uint32_t src = (b * (uint32_t)chunks + j) * nn * E, dst = b * Gsz * E + (k0 - 1) * Gset * E;
            for (uint32_t e = lr; e < nn; e += 16) bst(rG, live ? dst + E * e : SW_OOB, bld(rQ, src + E * e));
        }
        if (!a.ss) continue;
        // stored blocks: Dk = Pinv[k0,1], Dm = Pinv[k0-1,1], Lk = S[k0,0].  Pinv[k0,0] = -((Dk Lk) Dm), Pinv[k0-1,2] = -((Dm Lk^T) Dk).
        double Dk[n], Dm[n], t1[n], t2[n];
        load_rows<n>(Dk, rP, oSk + nn * E, n, lr);
        load_rows<n>(Dm, rP, oSk - 2u * nn * E, n, lr);
        {
            double LT[n], DmT[n];
            load_rows_t<n>(LT, rS, oSk, n, lr);
            load_rows_t<n>(DmT, rP, oSk - 2u * nn * E, n, lr);
            gemm_nt<n, n>(Dk, LT, t1);                                                            // Dk L            :100
            SW64_FENCE();
            gemm_nt<n, n>(t1, DmT, t2);                                                           // (Dk L) Dm       :102
            store_rows<n>(t2, rP, oSk, n, lr, live && r14, -1.0);                                 // Pinv[k0,0]      :106-113
        }
        SW64_FENCE();
        {
            double Lk[n], DkT[n];
            load_rows<n>(Lk, rS, oSk, n, lr);
            load_rows_t<n>(DkT, rP, oSk + nn * E, n, lr);
            gemm_nt<n, n>(Dm, Lk, t1);                                                            // Dm phi^T        :121
            SW64_FENCE();
            gemm_nt<n, n>(t1, DkT, t2);                                                           // (Dm phi^T) Dk   :123
            store_rows<n>(t2, rP, oSk - nn * E, n, lr, live && r14, -1.0);                        // Pinv[k0-1,2]    :127-134
        }
    }
}

// dz = G^-1 (g - C^T lambda) in double (as sw::compute_dz_dpp_kernel): four knots per wavefront, a 16-lane row per knot.
__global__ __launch_bounds__(64, 2) void compute_dz_dpp_f64_kernel(DzArgsT<double> a) {
    constexpr int n = 14, m = 7;
    constexpr uint32_t nn = n * n, mm = m * m, nm = n * m, E = 8;
    constexpr uint32_t Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const int N = a.N;
    const uint32_t Gsz = Gset * (uint32_t)N - mm, Csz = Cset * (uint32_t)(N - 1), gsz = gset * (uint32_t)N - m;
    const uint32_t B = (uint32_t)a.batch;
    const rsrc_t rG = make_rsrc(a.Ginv, (size_t)B * Gsz * E), rC = make_rsrc(a.C, (size_t)B * Csz * E), rg = make_rsrc(a.g, (size_t)B * gsz * E),
                 rl = make_rsrc(a.lambda, (size_t)B * n * N * E), rz = make_rsrc(a.dz, (size_t)B * gsz * E);
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const bool r14 = lr < n, r7 = lr < m;
    const uint32_t l14 = E * (r14 ? lr : n - 1), l7 = E * (r7 ? lr : m - 1);
    const unsigned items = B * (unsigned)N;
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;
        const uint32_t b = it / (unsigned)N, k = it % (unsigned)N;
        const bool last = k == (uint32_t)(N - 1);
        const uint32_t kc = last ? k - 1 : k;                         // the last knot re-reads its neighbour's C / R (results dropped)
        const uint32_t oG = b * Gsz * E + k * Gset * E, oR = b * Gsz * E + kc * Gset * E + nn * E, oC = b * Csz * E + kc * Cset * E;
        const uint32_t og = b * gsz * E + k * gset * E, ol = b * (uint32_t)(n * N) * E + k * n * E;
        double At[n], Bt[n], Qi[n], Ri[m];
        load_rows_t<n>(At, rC, oC, n, lr);                            // lane t: Abar[0..13][t]
        load_rows_t<n>(Bt, rC, oC + nn * E, m, lr);                   // lane j < 7: Bbar[0..13][j]
        load_rows<n>(Qi, rG, oG, n, lr);
        load_rows<m>(Ri, rG, oR, m, lr);
        const double lk = bld(rl, ol + l14);
        const double ln = bld(rl, ol + (last ? 0u : n * E) + l14);
        const double gx = bld(rg, og + l14), gu = bld(rg, og + (last ? l14 : n * E + l7));
        const double accx = matvec<n>(At, ln);                        // Abar^T lambda_{k+1}
        const double accy = matvec<n>(Bt, ln);                        // Bbar^T lambda_{k+1}
        const double ax = last ? 0.0 : accx;
        const double tx = gx - (lk + ax);
        const double tu = gu - accy;
        const double dx = matvec<n>(Qi, tx);
        const double du = matvec<m>(Ri, tu);
        bst(rz, (live && r14) ? og + E * lr : SW_OOB, dx);
        bst(rz, (live && r7 && !last) ? og + n * E + E * lr : SW_OOB, du);
    }
}

#pragma clang fp contract(fast)

}  // namespace sw64
}  // namespace mpcg
