// pcg_f64.hip.h — the PCG solve in double precision (linsys_t = double: USE_DOUBLES=1 in the reference,
// include/common/settings.cuh:41-49).  Functional counterpart of pcg_traj_kernel, not a tuned one: one workgroup
// per trajectory, iterate vectors in LDS, S and Pinv streamed from memory every iteration (they are twice the bytes
// and would need twice the registers to stay resident), one thread per output row of the block-tridiagonal products,
// deterministic fixed-order reductions.  Same semantics as the fp32 solver: |eta'| < exit_tol exit, iters = completed
// lambda updates, flag = ran out of iterations, 0 iterations if already converged, unwritten boundary blocks never read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

struct PcgArgs64 {
    const double* S; const double* Pinv; const double* gamma; double* lambda;
    double* r_out; double* p_out;                      // optional [batch][N][n]
    uint32_t* iters; uint8_t* max_iter_exit;
    int N; int max_iter; double exit_tol; int pcols;   // pcols: 3 = SS, 1 = block-Jacobi
};

constexpr int F64_THREADS = 256;
__host__ __device__ constexpr size_t pcg_f64_lds_doubles(int N) { return 2 * (size_t)(N + 2) * 14 + 2 * (size_t)N * 14 + 8; }

__global__ __launch_bounds__(F64_THREADS) void pcg_f64_kernel(PcgArgs64 a) {
    constexpr int n = 14, nn = n * n, NT = F64_THREADS;
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    const int N = a.N, tid = threadIdx.x, b = blockIdx.x;
    double* xp = lds64;                                // p, knot j at (j+1)*n, zero knot either side
    double* xr = xp + (size_t)(N + 2) * n;             // r likewise
    double* lam = xr + (size_t)(N + 2) * n;
    double* tmp = lam + (size_t)N * n;
    double* red = tmp + (size_t)N * n;                 // [NT/64]
    const double* S = a.S + (size_t)b * 3 * nn * N;
    const double* P = a.Pinv + (size_t)b * 3 * nn * N;
    const double* gam = a.gamma + (size_t)b * n * N;
    double* lam_g = a.lambda + (size_t)b * n * N;

    // y = M x (x padded), returns this thread's part of d . y
    auto pass = [&](const double* M, int cols, const double* x, const double* d) -> double {
        double part = 0.0;
        for (int row = tid; row < N * n; row += NT) {
            const int k = row / n, i = row - k * n;
            double acc = 0.0;
            for (int s = (cols == 3 ? 0 : 1); s < (cols == 3 ? 3 : 2); ++s) {
                if ((s == 0 && k == 0) || (s == 2 && k == N - 1)) continue;
                const double* blk = M + ((size_t)k * 3 + s) * nn;
                const double* xk = x + (size_t)(k + s) * n;
#pragma unroll
                for (int c = 0; c < n; ++c) acc = fma(blk[i + c * n], xk[c], acc);
            }
            tmp[row] = acc;
            part = fma(d[(size_t)(k + 1) * n + i], acc, part);
        }
        return part;
    };
    auto block_sum = [&](double part) -> double {      // fixed order: lanes, then waves
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        return s;
    };

    for (int e = tid; e < (N + 2) * n; e += NT) { xp[e] = 0.0; xr[e] = 0.0; }
    __syncthreads();
    for (int e = tid; e < N * n; e += NT) {
        const double l0 = lam_g[e];
        xp[n + e] = l0; lam[e] = l0; xr[n + e] = gam[e];
    }
    __syncthreads();
    (void)pass(S, 3, xp, xp);                          // r = gamma - S lambda0
    __syncthreads();
    for (int e = tid; e < N * n; e += NT) xr[n + e] -= tmp[e];
    __syncthreads();
    double eta = block_sum(pass(P, a.pcols, xr, xr));  // r~ = Pinv r ; eta = r . r~
    for (int e = tid; e < N * n; e += NT) xp[n + e] = tmp[e];
    __syncthreads();

    uint32_t iters = 0, flag = 1;
    if (fabs(eta) < a.exit_tol) {
        flag = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
            const double v = block_sum(pass(S, 3, xp, xp));
            const double alpha = eta / v;
            for (int e = tid; e < N * n; e += NT) {
                lam[e] += alpha * xp[n + e];
                xr[n + e] -= alpha * tmp[e];
            }
            __syncthreads();
            const double eta_new = block_sum(pass(P, a.pcols, xr, xr));
            iters = (uint32_t)(it + 1);
            if (fabs(eta_new) < a.exit_tol) { flag = 0; break; }
            const double beta = eta_new / eta;
            for (int e = tid; e < N * n; e += NT) xp[n + e] = tmp[e] + beta * xp[n + e];
            eta = eta_new;
            __syncthreads();
        }
    }
    for (int e = tid; e < N * n; e += NT) {
        lam_g[e] = lam[e];
        if (a.r_out) a.r_out[(size_t)b * n * N + e] = xr[n + e];
        if (a.p_out) a.p_out[(size_t)b * n * N + e] = xp[n + e];
    }
    if (tid == 0) { a.iters[b] = iters; a.max_iter_exit[b] = (uint8_t)flag; }
}

}  // namespace mpcg
