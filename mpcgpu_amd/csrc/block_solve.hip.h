// block_solve.hip.h — batched block-tridiagonal DIRECT solve of S lambda = gamma on the GPU.
//
// The reference's second linear-system path (LINSYS_SOLVE == 0) ships the Schur matrix to the host and calls QDLDL
// (qdldl_solve_schur, include/qdldl/sqp.cuh:22-49: sparse LDL^T factor + solve per SQP iteration, one trajectory,
// one CPU thread).  The GPU-native counterpart keeps the system where mpcg_form_schur left it (bd layout, stored
// negated) and runs a block LU sweep per trajectory on the register/DPP primitives of dpp_rows.hip.h:
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
#include "dpp_rows.hip.h"

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

// ---- the same sweep for FEW trajectories: one trajectory per wavefront, the 29 columns of [Delta | U y] dealt
// round-robin to the four 16-lane DPP rows (lane = 16 g + r holds row r of columns c = 4 j + g, j = 0..3, of Delta and
// of U; g = 3 also carries y / z).  Per pivot one ds_bpermute moves the pivot column from its owner row to the other
// three; everything else stays inside a row (row_newbcast).  Every entry goes through exactly the operations of the
// narrow kernel above, so the results are bit-identical; ~0.9 k instead of ~2.3 k instructions per knot on the
// critical path.  The back substitution sums over all 14 columns in order, so it runs in the narrow layout
// (redundantly in the four rows). ----
__global__ __launch_bounds__(64, 2) void bt_block_solve_wide_kernel(BlockSolveArgs a) {
    using namespace sdpp;
    constexpr int n = 14, nn = n * n, WS = nn + n, NSL = 4;
    const int N = a.N;
    const int lane = threadIdx.x, lr = lane & 15, g = lane >> 4;
    const bool r14 = lr < n;
    const int lc = r14 ? lr : n - 1;
    const size_t b = blockIdx.x;
    const float* S = a.S + b * 3 * nn * N;
    const float* gamma = a.gamma + b * n * N;
    float* lambda = a.lambda + b * n * N;
    float* work = a.work + b * (size_t)N * WS;
    // this lane's columns: c_j = 4 j + g (clamped for addressing; slots with c_j >= 14 compute on duplicates, store nothing)
    int cj[NSL];
    bool cv[NSL];
#pragma unroll
    for (int j = 0; j < NSL; ++j) { cv[j] = 4 * j + g < n; cj[j] = cv[j] ? 4 * j + g : n - 1; }

    float W[NSL];                                          // this lane's columns of W_{k-1}
    float zp = 0.f;                                        // z_{k-1}[lr] (meaningful in row g = 3)
#pragma unroll
    for (int j = 0; j < NSL; ++j) W[j] = 0.f;
    float Dn[NSL], Un[NSL], Ln[n], yn;
    auto fetch = [&](int k) {
        const float* blk = S + (size_t)k * 3 * nn;
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            Dn[j] = blk[nn + lc + cj[j] * n];
            Un[j] = blk[2 * nn + lc + cj[j] * n];
        }
#pragma unroll
        for (int c = 0; c < n; ++c) Ln[c] = blk[lc + c * n];
        yn = gamma[(size_t)k * n + lc];
    };
    fetch(0);
    for (int k = 0; k < N; ++k) {
        float D[NSL], U[NSL], L[n];
#pragma unroll
        for (int j = 0; j < NSL; ++j) { D[j] = Dn[j]; U[j] = (k < N - 1) ? Un[j] : 0.f; }
#pragma unroll
        for (int c = 0; c < n; ++c) L[c] = Ln[c];
        float y = yn;
        if (k + 1 < N) fetch(k + 1);
        if (k > 0) {
            // Delta = D - L W_{k-1} (own columns), y -= L z_{k-1}: sums over t = 0..13 in order, W / z from lane t of the row
            float t[NSL];
#pragma unroll
            for (int j = 0; j < NSL; ++j) t[j] = 0.f;
            float v = 0.f;
            SFor<0, n>::run([&](auto tc) {
                constexpr int T = decltype(tc)::value;
#pragma unroll
                for (int j = 0; j < NSL; ++j) {
                    const float p = L[T] * rbc<T>(W[j]);
                    t[j] = t[j] + p;
                }
                const float pv = L[T] * rbc<T>(zp);
                v = v + pv;
            });
#pragma unroll
            for (int j = 0; j < NSL; ++j) D[j] = D[j] - t[j];
            y = y - v;
        }
        // Gauss-Jordan on [Delta | U y], columns dealt over the four rows
        SFor<0, n>::run([&](auto pc_) {
            constexpr int P = decltype(pc_)::value;
            constexpr int GP = P % 4, JP = P / 4;
            // column P of Delta, from its owner row to every row (same lr)
            const float pcol = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((GP * 16 + lr) * 4, __builtin_bit_cast(int, D[JP])));
            const float pinv = 1.0f / rbc<P>(pcol);
            const bool is_p = lr == P;
#pragma unroll
            for (int j = JP; j < NSL; ++j) {               // columns right of the pivot (slot JP: only in rows g > GP; the
                const float pa = D[j] * pinv;              //  others recompute dead columns, which nobody reads again)
                const float ta = pcol * rbc<P>(pa);
                const float na = D[j] - ta;
                D[j] = is_p ? pa : na;
            }
#pragma unroll
            for (int j = 0; j < NSL; ++j) {
                const float pu = U[j] * pinv;
                const float tu = pcol * rbc<P>(pu);
                const float nu = U[j] - tu;
                U[j] = is_p ? pu : nu;
            }
            const float py = y * pinv;
            const float ty = pcol * rbc<P>(py);
            const float ny = y - ty;
            y = is_p ? py : ny;
        });
        // U now holds W_k (own columns), y holds z_k (row g = 3)
#pragma unroll
        for (int j = 0; j < NSL; ++j) W[j] = U[j];
        if (g == 3 && r14) work[(size_t)k * WS + nn + lr] = y;
        if (k < N - 1 && r14) {
#pragma unroll
            for (int j = 0; j < NSL; ++j)
                if (cv[j]) work[(size_t)k * WS + lr + cj[j] * n] = W[j];
        }
        zp = y;
    }
    // make the rows' stores visible to each other's loads (same wave, different lanes: order through the memory system)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float lam = work[(size_t)(N - 1) * WS + nn + lc];      // lambda_{N-1} = z_{N-1}
    if (g == 0 && r14) lambda[(size_t)(N - 1) * n + lr] = lam;
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
        if (g == 0 && r14) lambda[(size_t)k * n + lr] = lam;
    }
}

}  // namespace mpcg

#pragma clang fp contract(fast)     // (hipcc's default for device code: what the headers included after this one are written for)
