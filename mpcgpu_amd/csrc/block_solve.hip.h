// block_solve.hip.h — batched block-tridiagonal DIRECT solve of S lambda = gamma on the GPU.
//
// The reference's second linear-system path (LINSYS_SOLVE == 0) ships the Schur matrix to the host and calls QDLDL
// (qdldl_solve_schur, include/qdldl/sqp.cuh:22-49: sparse LDL^T factor + solve per SQP iteration, one trajectory,
// one CPU thread).  The GPU-native counterpart keeps the system where mpcg_form_schur left it (bd layout, stored
// negated) and runs a block LU sweep per trajectory on the register/DPP primitives of schur_dpp.hip.h:
//   Delta_0 = D_0, y_0 = gamma_0;  k >= 1:  Delta_k = D_k - L_k W_{k-1},  y_k = gamma_k - L_k z_{k-1};
//   z_k = Delta_k^-1 y_k,  W_k = Delta_k^-1 U_k;   lambda_{N-1} = z_{N-1},  lambda_k = z_k - W_k lambda_{k+1}
// (D_k = S[k,1], L_k = S[k,0], U_k = S[k,2]; pivot blocks inverted by the reference's Gauss-Jordan).
// FOUR trajectories per wavefront — one per 16-lane DPP row, lane r < 14 holds row r of every 14x14 operand — all
// four sweeping k = 0..N-1 in lock-step; W_k and z_k go through a global scratch of N x 210 floats per trajectory
// for the back substitution.  ~2.5 k instructions per knot and wave, no LDS, no barriers; the sweep is serial in k
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
    for (int k = 0; k < N; ++k) {
        float D[n];
        load_rows(D, S + (size_t)k * 3 * nn + nn, n, lr, true);
        float y = gamma[(size_t)k * n + lc];
        if (k > 0) {
            float L[n], t[n];
            load_rows(L, S + (size_t)k * 3 * nn, n, lr, true);
            gemm_nn<n, n>(L, W, t);                        // L_k W_{k-1}
#pragma unroll
            for (int c = 0; c < n; ++c) D[c] = D[c] - t[c];
            const float v = matvec<n>(L, zp);              // L_k z_{k-1}
            y = y - v;
        }
        float Dinv[n];
        invert(D, Dinv, lr);
        const float z = matvec<n>(Dinv, y);
        if (st) work[(size_t)k * WS + nn + lr] = z;
        if (k < N - 1) {
            float UT[n];                                   // U_k^T, row lc = column lc of U_k (56 contiguous bytes)
            const float* U = S + (size_t)k * 3 * nn + 2 * nn;
#pragma unroll
            for (int c = 0; c < n; ++c) UT[c] = U[c + lc * n];
            gemm_nt<n, n, true>(Dinv, UT, W);              // W_k = Delta_k^-1 U_k  (as Dinv (U^T)^T: same sums, same order)
            store_rows(W, work + (size_t)k * WS, n, lr, st, 1.f);
        }
        zp = z;
    }
    float lam = zp;                                        // lambda_{N-1} = z_{N-1}
    if (st) lambda[(size_t)(N - 1) * n + lr] = lam;
    for (int k = N - 2; k >= 0; --k) {
        float Wk[n];
        load_rows(Wk, work + (size_t)k * WS, n, lr, true);
        const float zk = work[(size_t)k * WS + nn + lc];
        const float v = matvec<n>(Wk, lam);
        lam = zk - v;
        if (st) lambda[(size_t)k * n + lr] = lam;
    }
}

}  // namespace mpcg
