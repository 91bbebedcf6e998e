// pcg_f64.hip.h — the GENERIC PCG kernel: any element type (float / double) and any state size n.  Serves
//   * double precision (linsys_t = double: USE_DOUBLES=1 in the reference, include/common/settings.cuh:41-49), n = 14 compiled in;
//   * state sizes other than the tuned n = 14 specialisation (SURVEY.md §8b "n=14 specialisation (+ generic fallback)"),
//     float and double, n at run time.
// Functional counterpart of the tuned kernels, not a tuned one: one workgroup
// per trajectory, iterate vectors in LDS, S and Pinv streamed from memory every iteration (they are twice the bytes
// and would need twice the registers to stay resident), one thread per output row of the block-tridiagonal products,
// deterministic fixed-order reductions.  Same semantics as the fp32 solver: |eta'| < exit_tol exit, iters = completed
// lambda updates, flag = ran out of iterations, 0 iterations if already converged, unwritten boundary blocks never read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

// Contraction is OFF in this header (as in the Schur headers): the vector updates are a rounded multiply followed by a rounded add,
// like the C oracle's; only the products spelled fma_t are fused.  (Until round 2 this was an accident of the include order — the
// pragma of dpp_rows.hip.h reached this file; the double-precision iteration counts the tests pin depend on it.)
#pragma clang fp contract(off)

__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fabs_t(double a) { return fabs(a); }
__device__ __forceinline__ float fabs_t(float a) { return fabsf(a); }

template <typename T>
struct PcgArgsG {
    const T* S; const T* Pinv; const T* gamma; T* lambda;
    T* r_out; T* p_out;                                // optional [batch][N][n]
    uint32_t* iters; uint8_t* max_iter_exit;
    int N; int max_iter; T exit_tol; int pcols;        // pcols: 3 = SS, 1 = block-Jacobi
    int n = 14;                                        // state size (used when the kernel is instantiated with NFIX = 0)
    int lower = 0;                                     // 1: never read the right block column (S and Pinv are block-symmetric: the handle's latch says so)
    // fix-up launches (behind the clustered double kernel): trajectory b is skipped when redo_flags[b * redo_stride] == redo_skip
    const unsigned long long* redo_flags = nullptr;
    unsigned long long redo_skip = 0;
    int redo_stride = 0;
    unsigned long long* redo_count = nullptr;         // incremented once per trajectory a fix-up launch actually solves ("cluster_fixups")
    const T* lam0 = nullptr;                           // fix-up launches: the warm start ([batch][N][n], the handle's copy of lambda made in front of the cluster launch); nullptr: `lambda`
};
typedef PcgArgsG<double> PcgArgs64;

constexpr int F64_THREADS = 256;                   // threads per trajectory: short systems ...
constexpr int F64_THREADS_WIDE = 1024;             // ... and from 64 x 14 rows upwards (more loads in flight per CU: double N=128 7.0 -> 8.1 M it/s)
__host__ __device__ constexpr int pcg_generic_threads(int N, int n) { return N * n >= 64 * 14 ? F64_THREADS_WIDE : F64_THREADS; }
__host__ __device__ constexpr size_t pcg_generic_lds_elems(int N, int n) { return 2 * (size_t)(N + 2) * n + 2 * (size_t)N * n + 16; }
__host__ __device__ constexpr size_t pcg_f64_lds_doubles(int N) { return pcg_generic_lds_elems(N, 14); }

// NFIX > 0: state size compiled in (inner products unrolled); NFIX = 0: a.n at run time.
template <typename T, int NFIX, int NTHR>
__global__ __launch_bounds__(NTHR) void pcg_generic_kernel(PcgArgsG<T> a) {
    typedef T real;
    const int n = NFIX ? NFIX : a.n, nn = n * n;
    constexpr int NT = NTHR;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    real* lds64 = reinterpret_cast<real*>(lds_raw);
    const int N = a.N, tid = threadIdx.x, b = blockIdx.x;
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    real* xp = lds64;                                  // p, knot j at (j+1)*n, zero knot either side
    real* xr = xp + (size_t)(N + 2) * n;             // r likewise
    real* lam = xr + (size_t)(N + 2) * n;
    real* tmp = lam + (size_t)N * n;
    real* red = tmp + (size_t)N * n;                 // [NT/64]
    const real* S = a.S + (size_t)b * 3 * nn * N;
    const real* P = a.Pinv + (size_t)b * 3 * nn * N;
    const real* gam = a.gamma + (size_t)b * n * N;
    real* lam_g = a.lambda + (size_t)b * n * N;

    // y = M x (x padded), returns this thread's part of d . y
    auto pass = [&](const real* M, int cols, const real* x, const real* d) -> real {
        real part = real(0);
        for (int row = tid; row < N * n; row += NT) {
            const int k = row / n, i = row - k * n;
            real acc = real(0);
            for (int s = (cols == 3 ? 0 : 1); s < (cols == 3 ? 3 : 2); ++s) {
                if ((s == 0 && k == 0) || (s == 2 && k == N - 1)) continue;
                const real* xk = x + (size_t)(k + s) * n;
                if (s == 2 && a.lower) {
                    // block (k, right) = block (k+1, left)^T (mpcg.h, BLOCK SYMMETRY): row i of it is COLUMN i of the left block of row k + 1 —
                    // n contiguous elements, and a block the threads of row k + 1 read in this very pass.  The same products in the same order as
                    // from the right block itself (bit-identical on symmetric matrices); a third of the HBM bytes of a pass gone: double N=64
                    // 11.3 -> 15.4 M it/s, N=128 5.1 -> 7.0 M (tools/_prof/f64_rate.py; 17.3 / 8.1 M with the wide workgroup).
                    const real* blt = M + ((size_t)(k + 1) * 3) * nn + (size_t)i * n;
                    if constexpr (NFIX > 0) {
#pragma unroll
                        for (int c = 0; c < NFIX; ++c) acc = fma_t(blt[c], xk[c], acc);
                    } else {
                        for (int c = 0; c < n; ++c) acc = fma_t(blt[c], xk[c], acc);
                    }
                    continue;
                }
                const real* blk = M + ((size_t)k * 3 + s) * nn;
if constexpr (NFIX > 0) {
#pragma unroll
                    for (int c = 0; c < NFIX; ++c) acc = fma_t(blk[i + c * n], xk[c], acc);
                } else {
                    for (int c = 0; c < n; ++c) acc = fma_t(blk[i + c * n], xk[c], acc);
                }
            }
            tmp[row] = acc;
            part = fma_t(d[(size_t)(k + 1) * n + i], acc, part);
        }
        return part;
    };
    auto block_sum = [&](real part) -> real {      // fixed order: lanes, then waves
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        real s = real(0);
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        return s;
    };

    for (int e = tid; e < (N + 2) * n; e += NT) { xp[e] = real(0); xr[e] = real(0); }
    __syncthreads();
    for (int e = tid; e < N * n; e += NT) {
        const real l0 = a.lam0 ? a.lam0[(size_t)b * n * N + e] : lam_g[e];
        xp[n + e] = l0; lam[e] = l0; xr[n + e] = gam[e];
    }
    __syncthreads();
    (void)pass(S, 3, xp, xp);                          // r = gamma - S lambda0
    __syncthreads();
    for (int e = tid; e < N * n; e += NT) xr[n + e] -= tmp[e];
    __syncthreads();
    real eta = block_sum(pass(P, a.pcols, xr, xr));  // r~ = Pinv r ; eta = r . r~
    for (int e = tid; e < N * n; e += NT) xp[n + e] = tmp[e];
    __syncthreads();

    uint32_t iters = 0, flag = 1;
    if (fabs_t(eta) < a.exit_tol) {
        flag = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
            const real v = block_sum(pass(S, 3, xp, xp));
            const real alpha = eta / v;
            for (int e = tid; e < N * n; e += NT) {
                lam[e] += alpha * xp[n + e];
                xr[n + e] -= alpha * tmp[e];
            }
            __syncthreads();
            const real eta_new = block_sum(pass(P, a.pcols, xr, xr));
            iters = (uint32_t)(it + 1);
            if (fabs_t(eta_new) < a.exit_tol) { flag = 0; break; }
            const real beta = eta_new / eta;
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

#pragma clang fp contract(fast)     // (hipcc's default for device code: what the headers included after this one are written for)

}  // namespace mpcg
