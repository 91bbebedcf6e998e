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

#pragma clang fp contract(off)

namespace sdpp {

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

// C[r][c] = sum_t A[r][t] * B[t][c]      A: NI columns per lane, B: rows in lanes 0..NI-1, NC columns
// PIN: B comes straight from memory (complete_ss): instruction selection otherwise emits all NI*NC broadcasts
// of all products of the kernel first and parks them in scratch (1.8 KB per lane, 5x slower than the LDS
// version); an empty volatile asm on the broadcast source keeps each one next to its multiply.
template <int NI, int NC, bool PIN = false>
__device__ __forceinline__ void gemm_nn(const float (&A)[NI], const float (&B)[NC], float (&Cm)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) Cm[c] = 0.f;
    SFor<0, NI>::run([&](auto tc) {
        constexpr int T = decltype(tc)::value;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float bsrc = B[c];
            if constexpr (PIN) asm volatile("" : "+v"(bsrc));
            const float p = A[T] * rbc<T>(bsrc);
            Cm[c] = Cm[c] + p;
        }
    });
}
// C[r][c] = sum_t A[r][t] * B[c][t]      (B transposed: B's row c sits in lane c), NC output columns
// SERIAL (operands straight from memory, complete_ss): an empty volatile asm ties B to the previous column's
// result, so that a column's broadcasts cannot be issued before the previous column is finished — otherwise all
// NI*NC broadcasts of all products are emitted first and parked in scratch.
template <int NI, int NC, bool SERIAL = false>
__device__ __forceinline__ void gemm_nt(const float (&A)[NI], float (&B)[NI], float (&Cm)[NC]) {
    float prev = 0.f;
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        if constexpr (SERIAL) {
            static_assert(!SERIAL || NI == 14, "SERIAL is written for 14 columns");
            asm volatile("" : "+v"(prev), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]),
                              "+v"(B[7]), "+v"(B[8]), "+v"(B[9]), "+v"(B[10]), "+v"(B[11]), "+v"(B[12]), "+v"(B[NI - 1]));
        }
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const float p = A[t] * rbc<Cc>(B[t]);
            acc = acc + p;
        }
        Cm[Cc] = acc;
        prev = acc;
    });
}
// out[r] = sum_c M[r][c] * v[c]          v: element c in lane c
template <int NC>
__device__ __forceinline__ float matvec(const float (&M)[NC], float v) {
    float acc = 0.f;
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        const float p = M[Cc] * rbc<Cc>(v);
        acc = acc + p;
    });
    return acc;
}
// Gauss-Jordan on [A | I] without pivoting (include/utils/matrix.cuh:120-238), rows in lanes 0..NN-1:
// A destroyed, I becomes A^-1.  lr = lane index inside the 16-lane row.
// A pivot step updates every column of [A | I] independently of the others (only the pivot COLUMN, read before the step, couples
// them), and without pivoting half of the 2 NN columns are structurally inert at every step: columns of A at or left of the pivot are
// finished (nobody reads them again) and columns of I right of the pivot are still unit columns — the pivot row holds 0 there, so
// the reference's update leaves them as they are (x - pcol * 0).  Skipping both halves the work (round 3) and changes no bit of A^-1
// for finite inputs.
template <int NN>
__device__ __forceinline__ void invert(float (&A)[NN], float (&I)[NN], int lr) {
#pragma unroll
    for (int c = 0; c < NN; ++c) I[c] = (lr == c) ? 1.f : 0.f;
    SFor<0, NN>::run([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const float pinv = 1.0f / rbc<P>(A[P]);
        const float pcol = A[P];
        const bool is_p = lr == P;
#pragma unroll
        for (int c = P + 1; c < NN; ++c) {
            const float pa = A[c] * pinv;               // the pivot row's entry (meaningful in lane P)
            const float ta = pcol * rbc<P>(pa);
            const float na = A[c] - ta;
            A[c] = is_p ? pa : na;
        }
#pragma unroll
        for (int c = 0; c <= P; ++c) {
            const float pi = I[c] * pinv;
            const float ti = pcol * rbc<P>(pi);
            const float ni = I[c] - ti;
            I[c] = is_p ? pi : ni;
        }
    });
}

// Gauss-Jordan elimination of [A | R] -> [I | A^-1 R] without pivoting, rows in lanes 0..NN-1, NR right-hand
// columns.  Columns of A at or left of the pivot are not touched (they are unit columns afterwards by construction
// and nobody reads them): 4 (NN - 1 - p + NR) instructions for pivot p.  A is destroyed.
template <int NN, int NR>
__device__ __forceinline__ void solve_aug(float (&A)[NN], float (&R)[NR], int lr) {
    SFor<0, NN>::run([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        const float pinv = 1.0f / rbc<P>(A[P]);
        const float pcol = A[P];
        const bool is_p = lr == P;
#pragma unroll
        for (int c = P + 1; c < NN; ++c) {
            const float pa = A[c] * pinv;
            const float ta = pcol * rbc<P>(pa);
            const float na = A[c] - ta;
            A[c] = is_p ? pa : na;
        }
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const float pr = R[c] * pinv;
            const float tr = pcol * rbc<P>(pr);
            const float nr = R[c] - tr;
            R[c] = is_p ? pr : nr;
        }
    });
}

// row lr of a column-major rows x cols matrix
// (always loads — from a clamped, in-bounds row — and selects afterwards: a per-row condition around the loads
//  turns into divergent branches with the whole operand array parked in scratch)
template <int COLS>
__device__ __forceinline__ void load_rows(float (&M)[COLS], const float* base, int rows, int lr, bool on) {
    const int lrc = lr < rows ? lr : rows - 1;
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const float v = base[lrc + c * rows];
        M[c] = on ? v : 0.f;
    }
}
template <int COLS>
__device__ __forceinline__ void store_rows(const float (&M)[COLS], float* base, int rows, int lr, bool on, float mult) {
    if (on) {
#pragma unroll
        for (int c = 0; c < COLS; ++c) base[lr + c * rows] = M[c] * mult;
    }
}

}  // namespace sdpp

// Block rows k >= 1 of every trajectory (the k = 1 item also emits block row 0):
// S[k,0], S[k,1], S[k-1,2], Pinv[k,1], gamma[k]; inverses of Q_{k-1}, R_{k-1} (and Q_{N-1}) -> scratch.
__global__ __launch_bounds__(64, 2) void form_schur_dpp_kernel(SchurArgs a) {
    using namespace sdpp;
    constexpr int n = 14, m = 7;
    constexpr int nn = n * n, mm = m * m, nm = n * m;
    constexpr int Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const int N = a.N;
    const size_t Gsz = (size_t)Gset * N - mm, Csz = (size_t)Cset * (N - 1), gsz = (size_t)gset * N - m;
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const unsigned items = (unsigned)a.batch * (unsigned)(N - 1);      // (host: batch * N < 2^31)
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;        // dead rows redo the last item and store nothing
        const int b = (int)(it / (unsigned)(N - 1)), k = 1 + (int)(it % (unsigned)(N - 1));
        const bool r14 = lr < n, r7 = lr < m;
        const float* G = a.G + (size_t)b * Gsz;
        const float* C = a.C + (size_t)b * Csz;
        const float* g = a.g + (size_t)b * gsz;
        const float* c = a.c + (size_t)b * n * N;
        float* S = a.S + (size_t)b * 3 * nn * N;
        float* P = a.Pinv + (size_t)b * 3 * nn * N;
        float* gamma = a.gamma + (size_t)b * n * N;
        float* Gs = a.Ginv_scratch + (size_t)b * Gsz;

        float Ak[n], Bk[m], Qk[n], Qp[n], Rk[m];                                   // linsys_setup.cuh:318-325
        load_rows(Ak, C + (size_t)(k - 1) * Cset, n, lr, r14);
        load_rows(Bk, C + (size_t)(k - 1) * Cset + nn, n, lr, r14);
        load_rows(Qk, G + (size_t)(k - 1) * Gset, n, lr, r14);
        load_rows(Rk, G + (size_t)(k - 1) * Gset + nn, m, lr, r7);
        load_rows(Qp, G + (size_t)k * Gset, n, lr, r14);
        const int l14 = r14 ? lr : n - 1, l7 = r7 ? lr : m - 1;
        const float qk_ = g[(size_t)(k - 1) * gset + l14], rk_ = g[(size_t)(k - 1) * gset + n + l7];
        const float qp_ = g[(size_t)k * gset + l14], ck_ = c[(size_t)k * n + l14];
        const float qk = r14 ? qk_ : 0.f, rk = r7 ? rk_ : 0.f, qp = r14 ? qp_ : 0.f, ck = r14 ? ck_ : 0.f;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            if (lr == i) { Qk[i] += a.rho; Qp[i] += a.rho; }
        }
#pragma unroll
        for (int i = 0; i < m; ++i) {
            if (lr == i) Rk[i] += a.rho;
        }
        const bool st14 = live && r14, st7 = live && r7;
        // block row 0 (linsys_setup.cuh:152-277) needs nothing but Q_0 + rho I, its inverse and q_0 — all of which
        // the k = 1 item has in hand: Pinv[0,1] = -(Q0 + rho I), S[0,1] = -Q0^-1, gamma_0 = -Q0^-1 q_0
        if (k == 1 && a.pinv) store_rows(Qk, P + nn, n, lr, st14, -1.f);              // :201-210
        float Qki[n], Qpi[n], Rki[m];
        invert(Qk, Qki, lr);                                                          // :356-368
        invert(Qp, Qpi, lr);
        invert(Rk, Rki, lr);
        {
            const float g0 = matvec<n>(Qki, qk);                                      // :259-264
            if (k == 1) {
                store_rows(Qki, S + nn, n, lr, st14, -1.f);                           // :248-255
                if (st14) gamma[lr] = -g0;                                            // :272-276
            }
        }
        float phi[n], BR[m];
        gemm_nn<n, n>(Ak, Qki, phi);                                                  // phi = Abar Qi      :397-398
        gemm_nn<m, m>(Bk, Rki, BR);                                                   // Bbar Ri            :405-406
        float gam = matvec<n>(Qpi, qp);                                               // :410-415
        gam -= ck;                                                                    // :416-418
        const float v1 = matvec<n>(phi, qk);                                          // :421-426
        const float v2 = matvec<m>(BR, rk);                                           // :431-436
        gam += v2 + v1;                                                               // :441-443
        float theta[n], t1[n];
        gemm_nt<n, n>(phi, Ak, theta);                                                // phi Abar^T         :446-455
        gemm_nt<m, n>(BR, Bk, t1);                                                    // (Bbar Ri) Bbar^T   :472-481
#pragma unroll
        for (int cc = 0; cc < n; ++cc) { theta[cc] += Qpi[cc]; theta[cc] += t1[cc]; } // :466-468, 485-487
        store_rows(phi, S + (size_t)k * 3 * nn, n, lr, st14, -1.f);                   // S[k,0]             :490-497
        store_rows(theta, S + (size_t)k * 3 * nn + nn, n, lr, st14, -1.f);            // S[k,1]             :500-507
        if (st14) {                                                                   // S[k-1,2] = -phi^T  :536-557
            float* dst = S + (size_t)(k - 1) * 3 * nn + 2 * nn;
#pragma unroll
            for (int cc = 0; cc < n; ++cc) dst[cc + lr * n] = phi[cc] * -1.f;
        }
        if (a.pinv) {                                                                 // (uniform)
            float thetaInv[n];
            invert(theta, thetaInv, lr);                                              // :510-514
            store_rows(thetaInv, P + (size_t)k * 3 * nn + nn, n, lr, st14, -1.f);     // Pinv[k,1]          :517-524
        }
        if (st14) gamma[(size_t)k * n + lr] = -gam;                                   // :528-532
        store_rows(Qki, Gs + (size_t)(k - 1) * Gset, n, lr, st14, 1.f);               // G <- G^-1 (via scratch) :371-380
        store_rows(Rki, Gs + (size_t)(k - 1) * Gset + nn, m, lr, st7, 1.f);
        if (k == N - 1) store_rows(Qpi, Gs + (size_t)k * Gset, n, lr, st14, 1.f);
    }
}

// symmetric-stair completion (linsys_setup.cuh:9-137) + publication of G^-1, four knots per wave
__global__ __launch_bounds__(64, 2) void complete_ss_dpp_kernel(SchurArgs a) {
    using namespace sdpp;
    constexpr int n = 14, m = 7, nn = n * n, mm = m * m;
    constexpr int Gset = nn + mm;
    const int N = a.N;
    const size_t Gsz = (size_t)Gset * N - mm;
    const int lane = threadIdx.x;
    const int lr = lane & 15;
    const unsigned items = (unsigned)a.batch * (unsigned)N;
    for (unsigned base = blockIdx.x * 4u; base < items; base += gridDim.x * 4u) {
        const unsigned item = base + (unsigned)(lane >> 4);
        const bool live = item < items;
        const unsigned it = live ? item : items - 1;
        const int b = (int)(it / (unsigned)N), k = (int)(it % (unsigned)N);
        const bool r14 = lr < n;
        const float* S = a.S + (size_t)b * 3 * nn * N;
        float* P = a.Pinv + (size_t)b * 3 * nn * N;
        if (live) {                                                        // G <- G^-1: this knot's Q^-1 (and R^-1)
            const int cnt = (k < N - 1) ? Gset : nn;
            const float* src = a.Ginv_scratch + (size_t)b * Gsz + (size_t)k * Gset;
            float* dst = a.Ginv_out + (size_t)b * Gsz + (size_t)k * Gset;
            for (int e = lr; e < cnt; e += 16) dst[e] = src[e];
        }
        if (!a.ss) continue;
        // The second operand of every product is loaded TRANSPOSED (row lr of B^T = column lr of B: 56 contiguous
        // bytes) and the product taken as A * (B^T)^T with gemm_nt — same sums in the same order as the reference's
        // A * B; with gemm_nn on operands that come straight from memory the compiler emits all 4 x 196 broadcasts
        // first and parks them in scratch (1.8 KB per lane, 5x slower than the LDS version).
        // No zeroing of the operands of rows that store nothing (k = 0 has no left block, k = N-1 no right one,
        // lanes 14, 15 repeat row 13): whatever they compute stays inside their own 16-lane row and is dropped.
        const int lc = r14 ? lr : n - 1;
        auto load_t = [&](float (&M)[n], const float* base) {           // M = (block at base)^T, row lc
#pragma unroll
            for (int cc = 0; cc < n; ++cc) M[cc] = base[cc + lc * n];
        };
        float Dk[n];
        const bool has_l = k > 0, has_r = k < N - 1;
        load_rows(Dk, P + (size_t)k * 3 * nn + nn, n, lr, true);
        {
            float DmT[n], LT[n], t1[n], t2[n];
            load_t(LT, S + (size_t)(has_l ? k : k + 1) * 3 * nn);
            load_t(DmT, P + (size_t)(has_l ? k - 1 : k) * 3 * nn + nn);
            gemm_nt<n, n, true>(Dk, LT, t1);                                     // Dk L            :100
            gemm_nt<n, n, true>(t1, DmT, t2);                                    // (Dk L) Dm       :102
            store_rows(t2, P + (size_t)k * 3 * nn, n, lr, live && r14 && has_l, -1.f);          // Pinv[k,0]  :106-113
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            float DpT[n], Sn[n], t1[n], t2[n];
            load_t(DpT, P + (size_t)(has_r ? k + 1 : k) * 3 * nn + nn);
            // the reference multiplies by phi_{k+1}^T, "transposed on load" (:36-43): its transpose is the block itself
            load_rows(Sn, S + (size_t)(has_r ? k + 1 : k) * 3 * nn, n, lr, true);
            gemm_nt<n, n, true>(Dk, Sn, t1);                                     // Dk phi^T        :121
            gemm_nt<n, n, true>(t1, DpT, t2);                                    // (Dk phi^T) Dp   :123
            store_rows(t2, P + (size_t)k * 3 * nn + 2 * nn, n, lr, live && r14 && has_r, -1.f); // Pinv[k,2]  :127-134
        }
    }
}

}  // namespace mpcg
