// dpp_rows.hip.h — rows-in-lanes primitives on 16-lane DPP rows (gfx950): four independent problems per wavefront, lane r < 14 of a
// row holds ROW r of every 14x14 operand in registers, and what another row needs from row t of an operand is a `row_newbcast:t` DPP source
// modifier on the multiply (verified on the chip: tools/_prof/dpp_probe.hip).  Used by the block-tridiagonal direct solver
// (block_solve.hip.h).  (Rounds 2-3 ran the Schur formation on these as well; round 4's formation, schur_walk.hip.h, has its own
// column-pair versions.)
//
// Contraction is OFF: every a*b+c is a rounded multiply followed by a rounded add, sequential over the contracted index, accumulators
// starting at +0 — the C oracle's bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mpcg {

#pragma clang fp contract(off)

namespace sdpp {

// value held by lane L of this lane's 16-lane row
template <int L>
__device__ __forceinline__ float rbc(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + L, 0xf, 0xf, true));
}

// acc += a * (b held by lane L of this 16-lane row): rounded multiply through the DPP source modifier, rounded add
template <int L>
__device__ __forceinline__ void mac_bc(float& acc, float a, float b) {
    const float p = a * rbc<L>(b);
    acc = acc + p;
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
template <int NI, int NC>
__device__ __forceinline__ void gemm_nn(const float (&A)[NI], const float (&B)[NC], float (&Cm)[NC]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) Cm[c] = 0.f;
    SFor<0, NI>::run([&](auto tc) {
        constexpr int T = decltype(tc)::value;
#pragma unroll
        for (int c = 0; c < NC; ++c) mac_bc<T>(Cm[c], A[T], B[c]);
    });
}
// out[r] = sum_c M[r][c] * v[c]          v: element c in lane c
template <int NC>
__device__ __forceinline__ float matvec(const float (&M)[NC], float v) {
    float acc = 0.f;
    SFor<0, NC>::run([&](auto cc) {
        constexpr int Cc = decltype(cc)::value;
        mac_bc<Cc>(acc, M[Cc], v);
    });
    return acc;
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

#pragma clang fp contract(fast)

}  // namespace mpcg
