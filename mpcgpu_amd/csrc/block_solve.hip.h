// block_solve.hip.h — batched block-tridiagonal DIRECT solve of S lambda = gamma on the GPU.
//
// The reference's second linear-system path (LINSYS_SOLVE == 0) ships the Schur matrix to the host and calls QDLDL
// (qdldl_solve_schur, include/qdldl/sqp.cuh:22-49: sparse LDL^T factor + solve per SQP iteration, one trajectory,
// one CPU thread).  The GPU-native counterpart keeps the system where mpcg_form_schur left it (bd layout, stored
// negated) and runs a block LU sweep per trajectory on the register/DPP primitives of schur_dpp.hip.h:
//   Delta_0 = D_0, y_0 = gamma_0;  k >= 1:  Delta_k = D_k - L_k W_{k-1},  y_k = gamma_k - L_k z_{k-1};
//   z_k = Delta_k^-1 y_k,  W_k = Delta_k^-1 U_k;   lambda_{N-1} = z_{N-1},  lambda_k = z_k - W_k lambda_{k+1}
// (D_k = S[k,1], L_k = S[k,0], U_k = S[k,2]; W_k and z_k come out of ONE pivot-free Gauss-Jordan elimination of
// [Delta_k | U_k y_k], the reference's elimination scheme — include/utils/matrix.cuh:120-238 — with the right-hand
// sides in place of the identity and the already-eliminated columns skipped).
// FOUR trajectories per wavefront — one per 16-lane DPP row, lane r < 14 holds row r of every 14x14 operand — all
// four sweeping k = 0..N-1 in lock-step; W_k and z_k go through a global scratch of N x 210 floats per trajectory
// for the back substitution.  ~1.7 k instructions per knot and wave, no LDS, no barriers; the sweep is serial in k
// (that is what PCG avoids for ONE trajectory), so this is the throughput solver for batches: 1/50 of the flops of
// 167 PCG iterations.  The test oracle restates the same operation order on the CPU: results are bit-identical in float (tested).
#pragma once
#include "schur_dpp.hip.h"

namespace mpcg {

#pragma clang fp contract(off)

struct BlockSolveArgs {
    const float* S; const float* gamma; float* lambda; float* work;   // work: [batch][N][14*14 + 14]
    int N; int batch;
};

__global__ __launch_bounds__(64, 2) void bt_block_solve_kernel(BlockSolveArgs a) {
    using namespace sdpp;
    constexpr int n = 14, nn = n * n, WS = nn + n;
    const int N = a.N;
    const int lane = threadIdx.x, lr = lane & 15;
    const bool r14 = lr < n;
    const int lc = r14 ? lr : n - 1;                       // lanes 14, 15 repeat row 13 and store nothing
    const unsigned traj = blockIdx.x * 4u + (unsigned)(lane >> 4);
    const bool live = traj < (unsigned)a.batch;
    const size_t b = live ? traj : (unsigned)a.batch - 1;  // dead rows redo the last trajectory and store nothing
    const float* S = a.S + b * 3 * nn * N;
    const float* gamma = a.gamma + b * n * N;
    float* lambda = a.lambda + b * n * N;
    float* work = a.work + b * (size_t)N * WS;
    const bool st = live && r14;

    float W[n];                                            // W_{k-1}, rows in lanes
    float zp = 0.f;                                        // z_{k-1}, element lr
#pragma unroll
    for (int c = 0; c < n; ++c) W[c] = 0.f;
    // operands of the NEXT knot are requested before the current knot is eliminated: the sweep is one dependent
    // chain per trajectory, an un-prefetched HBM round trip per knot would double its length
    float Dn[n], Ln[n], Un[n], yn;
    auto fetch = [&](int k) {
        load_rows(Dn, S + (size_t)k * 3 * nn + nn, n, lr, true);
        load_rows(Ln, S + (size_t)k * 3 * nn, n, lr, true);            // (k = 0: the never-written block, never used)
        load_rows(Un, S + (size_t)k * 3 * nn + 2 * nn, n, lr, true);   // (k = N-1: likewise)
        yn = gamma[(size_t)k * n + lc];
    };
    fetch(0);
    for (int k = 0; k < N; ++k) {
        float D[n], L[n], R[n + 1];
#pragma unroll
        for (int c = 0; c < n; ++c) { D[c] = Dn[c]; L[c] = Ln[c]; R[c] = (k < N - 1) ? Un[c] : 0.f; }
        float y = yn;
        if (k + 1 < N) fetch(k + 1);
        if (k > 0) {
            float t[n];
            gemm_nn<n, n>(L, W, t);                        // L_k W_{k-1}
#pragma unroll
            for (int c = 0; c < n; ++c) D[c] = D[c] - t[c];
            const float v = matvec<n>(L, zp);              // L_k z_{k-1}
            y = y - v;
        }
        // [Delta_k | U_k y_k] -> [I | W_k z_k] in one elimination (the last block row carries y only)
        R[n] = y;
        solve_aug<n, n + 1>(D, R, lr);
#pragma unroll
        for (int c = 0; c < n; ++c) W[c] = R[c];
        const float z = R[n];
        if (st) work[(size_t)k * WS + nn + lr] = z;
        if (k < N - 1) store_rows(W, work + (size_t)k * WS, n, lr, st, 1.f);
        zp = z;
    }
    float lam = zp;                                        // lambda_{N-1} = z_{N-1}
    if (st) lambda[(size_t)(N - 1) * n + lr] = lam;
    float Wn[n], zn = 0.f;
    auto fetch_b = [&](int k) {
        load_rows(Wn, work + (size_t)k * WS, n, lr, true);
        zn = work[(size_t)k * WS + nn + lc];
    };
    if (N >= 2) fetch_b(N - 2);
    for (int k = N - 2; k >= 0; --k) {
        float Wk[n];
#pragma unroll
        for (int c = 0; c < n; ++c) Wk[c] = Wn[c];
        const float zk = zn;
        if (k > 0) fetch_b(k - 1);
        const float v = matvec<n>(Wk, lam);
        lam = zk - v;
        if (st) lambda[(size_t)k * n + lr] = lam;
    }
}

}  // namespace mpcg
