// schur_kernels.hip.h — the steps either side of the PCG, gfx950 HIP:
//   form_schur_kernel  + complete_ss_kernel : (G, C, g, c, rho) -> (S, Pinv, gamma), G <- G^-1
//        replaces form_S_gamma_Pinv_kernel / form_schur_system (include/pcg/linsys_setup.cuh:565-656)
//   compute_dz_kernel : dz = G^-1 (g - C^T lambda)      replaces compute_dz (include/common/dz.cuh:3-136)
//
// One wavefront (64 threads) per knot point; batch x N workgroups.  The reference runs both halves
// of the Schur formation in one cooperative kernel with a grid sync between them (:600); a kernel
// boundary (~1.5 us on this chip) is cheaper than any grid barrier, so they are two launches here.
// The reference also overwrites G with its block inverses while other blocks may still be reading
// the raw blocks (row k writes slot k-1 which row k-1 reads, :321 vs :372 — a benign race there);
// here the first kernel writes the inverses to a scratch buffer and the second copies them into G.
//
// Arithmetic follows the reference operation for operation and is written with contraction OFF and
// sequential inner products, so that it is BIT-IDENTICAL to the C oracle (oracle/mpcg_oracle_impl.inc,
// built with -ffp-contract=off) — the parity test compares bits, not tolerances.  These kernels move
// ~2.5 KB and ~20 kflop per knot: they are latency/launch bound and nowhere near any roofline.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

constexpr int SCH_THREADS = 64;

#pragma clang fp contract(off)

// C[m x k] = A[m x n] * B[n x k] (column-major); transB: B stored k x n.  Sequential over n.
template <typename T>
__device__ __forceinline__ void w_gemm(int m, int n, int k, const T* A, const T* B, T* C, bool transB) {
    for (int e = threadIdx.x; e < m * k; e += SCH_THREADS) {
        const int row = e % m, col = e / m;
        T acc = 0;
        for (int t = 0; t < n; ++t) acc += A[row + t * m] * (transB ? B[col + t * k] : B[t + col * n]);
        C[e] = acc;
    }
}
template <typename T>
__device__ __forceinline__ void w_matvec(int rows, int cols, const T* M, const T* v, T* out) {
    for (int r = threadIdx.x; r < rows; r += SCH_THREADS) {
        T acc = 0;
        for (int c = 0; c < cols; ++c) acc += M[r + c * rows] * v[c];
        out[r] = acc;
    }
}
// Gauss-Jordan on [A | I] without pivoting (include/utils/matrix.cuh:120-238): A destroyed, Ainv out.
// scr: 3n floats (pivot row of A, pivot row of Ainv, pivot column).
template <typename T>
__device__ __forceinline__ void w_invert(int n, T* A, T* Ainv, T* scr) {
    for (int e = threadIdx.x; e < n * n; e += SCH_THREADS) Ainv[e] = (T)((e % n) == (e / n));
    __syncthreads();
    T* prowA = scr;
    T* prowI = scr + n;
    T* pcol = scr + 2 * n;
    for (int piv = 0; piv < n; ++piv) {
        const T pinv = (T)1 / A[piv + piv * n];
        for (int c = threadIdx.x; c < n; c += SCH_THREADS) {
            prowA[c] = A[piv + c * n] * pinv;
            prowI[c] = Ainv[piv + c * n] * pinv;
            pcol[c] = A[c + piv * n];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += SCH_THREADS) {
            const int r = e % n, c = e / n;
            if (r == piv) { A[e] = prowA[c]; Ainv[e] = prowI[c]; }
            else { A[e] -= pcol[r] * prowA[c]; Ainv[e] -= pcol[r] * prowI[c]; }
        }
        __syncthreads();
    }
}
// Three independent Gauss-Jordan inversions advanced in lock-step (the reference inverts Q_k, Q_{k+1}, R_k
// together too: invertMatrix<T>(dimA, dimB, dimC, ...), include/utils/matrix.cuh): same arithmetic per element
// as three w_invert calls, a third of the barriers.  n3 <= n1 == n2.  scr: 3*(n1+n2+n3) floats.
template <typename T>
__device__ __forceinline__ void w_invert3(int n1, T* A1, T* I1, int n2, T* A2, T* I2, int n3, T* A3, T* I3, T* scr) {
    const int e1 = n1 * n1, e2 = n2 * n2, e3 = n3 * n3;
    for (int e = threadIdx.x; e < e1 + e2 + e3; e += SCH_THREADS) {
        if (e < e1) I1[e] = (T)((e % n1) == (e / n1));
        else if (e < e1 + e2) { const int f = e - e1; I2[f] = (T)((f % n2) == (f / n2)); }
        else { const int f = e - e1 - e2; I3[f] = (T)((f % n3) == (f / n3)); }
    }
    __syncthreads();
    T* s1 = scr;
    T* s2 = s1 + 3 * n1;
    T* s3 = s2 + 3 * n2;
    const int nmax = n1 > n2 ? n1 : n2;
    for (int piv = 0; piv < nmax; ++piv) {
        for (int c = threadIdx.x; c < n1 + n2 + n3; c += SCH_THREADS) {
            int n, cc; T *A, *I, *sc;
            if (c < n1) { n = n1; cc = c; A = A1; I = I1; sc = s1; }
            else if (c < n1 + n2) { n = n2; cc = c - n1; A = A2; I = I2; sc = s2; }
            else { n = n3; cc = c - n1 - n2; A = A3; I = I3; sc = s3; }
            if (piv < n) {
                const T pinv = (T)1 / A[piv + piv * n];
                sc[cc] = A[piv + cc * n] * pinv;
                sc[n + cc] = I[piv + cc * n] * pinv;
                sc[2 * n + cc] = A[cc + piv * n];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < e1 + e2 + e3; e += SCH_THREADS) {
            int n, f; T *A, *I, *sc;
            if (e < e1) { n = n1; f = e; A = A1; I = I1; sc = s1; }
            else if (e < e1 + e2) { n = n2; f = e - e1; A = A2; I = I2; sc = s2; }
            else { n = n3; f = e - e1 - e2; A = A3; I = I3; sc = s3; }
            if (piv < n) {
                const int r = f % n, c = f / n;
                if (r == piv) { A[f] = sc[c]; I[f] = sc[n + c]; }
                else { A[f] -= sc[2 * n + r] * sc[c]; I[f] -= sc[2 * n + r] * sc[n + c]; }
            }
        }
        __syncthreads();
    }
}
template <typename T>
__device__ __forceinline__ void w_copy(int cnt, const T* src, T* dst, T mult = (T)1) {
    for (int e = threadIdx.x; e < cnt; e += SCH_THREADS) dst[e] = src[e] * mult;
}

template <typename T>
struct SchurArgsT {
    const T* G; const T* C; const T* g; const T* c;
    T* S; T* Pinv; T* gamma; T* Ginv_scratch; T* Ginv_out;
    T rho; int n; int m; int N; int batch; int ss;
    int pinv;                 // 0: S and gamma only (no Pinv block is computed or written)
    int k0_only;              // form_schur_kernel: block row 0 of every trajectory only (the rest: schur_dpp.hip.h)
};
typedef SchurArgsT<float> SchurArgs;

// block row k of trajectory b: S[k,0], S[k,1], S[k-1,2], Pinv[k,1], gamma[k]; inverses -> scratch
// (n, m are compile-time: the small loops unroll and the index arithmetic folds — 3x fewer instructions
//  than the runtime-dimension version, which was issue-bound at ~16k instructions per knot)
template <int NN_, int MM_, typename T = float>
__global__ __launch_bounds__(SCH_THREADS) void form_schur_kernel(SchurArgsT<T> a) {
    __shared__ T sm[12 * 196 + 2 * 49 + 98 + 8 * 14 + 112];
    constexpr int n = NN_, m = MM_;
    const int N = a.N;
    constexpr int nn = n * n, mm = m * m, nm = n * m;
    constexpr int Gset = nn + mm, Cset = nn + nm, gset = n + m;
    const size_t Gsz = (size_t)Gset * N - mm, Csz = (size_t)Cset * (N - 1), gsz = (size_t)gset * N - m;
    T *Qk = sm, *Qki = Qk + nn, *Qp = Qki + nn, *Qpi = Qp + nn, *Ak = Qpi + nn, *phi = Ak + nn, *theta = phi + nn,
          *thetaInv = theta + nn, *t1 = thetaInv + nn, *t2 = t1 + nn, *phiT = t2 + nn, *BR = phiT + nn /* n x m */,
          *Bk = BR + nn, *Rk = Bk + nm, *Rki = Rk + mm, *gam = Rki + mm, *v1 = gam + n, *v2 = v1 + n, *qk = v2 + n,
          *qp = qk + n, *rk = qp + n, *scr = rk + n;

    const long total = a.k0_only ? (long)a.batch : (long)a.batch * N;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const int b = a.k0_only ? (int)item : (int)(item / N), k = a.k0_only ? 0 : (int)(item % N);
        const T* G = a.G + (size_t)b * Gsz;
        const T* C = a.C + (size_t)b * Csz;
        const T* g = a.g + (size_t)b * gsz;
        const T* c = a.c + (size_t)b * n * N;
        T* S = a.S + (size_t)b * 3 * nn * N;
        T* P = a.Pinv + (size_t)b * 3 * nn * N;
        T* gamma = a.gamma + (size_t)b * n * N;
        T* Gs = a.Ginv_scratch + (size_t)b * Gsz;
        __syncthreads();
        if (k == 0) {                                                    // linsys_setup.cuh:152-277
            w_copy(nn, G, Qk);
            w_copy(n, g, qk);
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += SCH_THREADS) Qk[i + i * n] += a.rho;
            __syncthreads();
            if (a.pinv) w_copy(nn, Qk, P + nn, (T)-1);                    // Pinv[0,1] = -(Q0 + rho I)
            __syncthreads();
            w_invert(n, Qk, Qki, scr);
            w_copy(nn, Qki, S + nn, (T)-1);                               // S[0,1] = -Q0^-1
            w_matvec(n, n, Qki, qk, v1);
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += SCH_THREADS) gamma[i] = -v1[i];
            continue;
        }
        w_copy(nn, C + (size_t)(k - 1) * Cset, Ak);                      // :318-325
        w_copy(nm, C + (size_t)(k - 1) * Cset + nn, Bk);
        w_copy(nn, G + (size_t)(k - 1) * Gset, Qk);
        w_copy(mm, G + (size_t)(k - 1) * Gset + nn, Rk);
        w_copy(nn, G + (size_t)k * Gset, Qp);
        w_copy(n, g + (size_t)(k - 1) * gset, qk);
        w_copy(m, g + (size_t)(k - 1) * gset + n, rk);
        w_copy(n, g + (size_t)k * gset, qp);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SCH_THREADS) { Qk[i + i * n] += a.rho; Qp[i + i * n] += a.rho; }
        for (int i = threadIdx.x; i < m; i += SCH_THREADS) Rk[i + i * m] += a.rho;
        __syncthreads();
        w_invert3(n, Qk, Qki, n, Qp, Qpi, m, Rk, Rki, scr);              // :356-368
        w_gemm(n, n, n, Ak, Qki, phi, false);                            // phi = Abar Qi          :397-398
        w_gemm(n, m, m, Bk, Rki, BR, false);                             // Bbar Ri                :405-406
        w_matvec(n, n, Qpi, qp, gam);                                    // :410-415
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SCH_THREADS) gam[i] -= c[(size_t)k * n + i];   // :416-418
        w_matvec(n, n, phi, qk, v1);                                     // :421-426
        w_matvec(n, m, BR, rk, v2);                                      // :431-436
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SCH_THREADS) gam[i] += v2[i] + v1[i];          // :441-443
        w_gemm(n, n, n, phi, Ak, theta, true);                           // phi Abar^T             :446-455
        w_gemm(n, m, n, BR, Bk, t1, true);                               // (Bbar Ri) Bbar^T       :472-481
        __syncthreads();
        for (int e = threadIdx.x; e < nn; e += SCH_THREADS) { theta[e] += Qpi[e]; theta[e] += t1[e]; }   // :466-468, 485-487
        __syncthreads();
        w_copy(nn, phi, S + (size_t)k * 3 * nn, (T)-1);                   // S[k,0]                 :490-497
        w_copy(nn, theta, S + (size_t)k * 3 * nn + nn, (T)-1);            // S[k,1]                 :500-507
        w_copy(nn, theta, t2);
        for (int e = threadIdx.x; e < nn; e += SCH_THREADS) { const int i = e % n, j = e / n; phiT[i + j * n] = phi[j + i * n]; }
        __syncthreads();
        w_copy(nn, phiT, S + (size_t)(k - 1) * 3 * nn + 2 * nn, (T)-1);   // S[k-1,2] = -phi^T      :536-557
        if (a.pinv) {
            w_invert(n, t2, thetaInv, scr);                              // :510-514
            w_copy(nn, thetaInv, P + (size_t)k * 3 * nn + nn, (T)-1);     // Pinv[k,1]              :517-524
        }
        for (int i = threadIdx.x; i < n; i += SCH_THREADS) gamma[(size_t)k * n + i] = -gam[i];   // :528-532
        w_copy(nn, Qki, Gs + (size_t)(k - 1) * Gset);                    // G <- G^-1 (via scratch) :371-380
        w_copy(mm, Rki, Gs + (size_t)(k - 1) * Gset + nn);
        if (k == N - 1) w_copy(nn, Qpi, Gs + (size_t)k * Gset);
    }
}

// symmetric-stair completion (linsys_setup.cuh:9-137) + publication of G^-1
template <int NN_, int MM_, typename T = float>
__global__ __launch_bounds__(SCH_THREADS) void complete_ss_kernel(SchurArgsT<T> a) {
    __shared__ T sm[7 * 196];
    constexpr int n = NN_, m = MM_, nn = n * n, mm = m * m;
    const int N = a.N;
    constexpr int Gset = nn + mm;
    const size_t Gsz = (size_t)Gset * N - mm;
    T *Dk = sm, *Dm = Dk + nn, *Dp = Dm + nn, *L = Dp + nn, *Rt = L + nn, *t1 = Rt + nn, *t2 = t1 + nn;
    for (long item = blockIdx.x; item < (long)a.batch * N; item += gridDim.x) {
        const int b = (int)(item / N), k = (int)(item % N);
        const T* S = a.S + (size_t)b * 3 * nn * N;
        T* P = a.Pinv + (size_t)b * 3 * nn * N;
        const int cnt = (k < N - 1) ? Gset : nn;
        w_copy(cnt, a.Ginv_scratch + (size_t)b * Gsz + (size_t)k * Gset, a.Ginv_out + (size_t)b * Gsz + (size_t)k * Gset);
        if (!a.ss) continue;
        __syncthreads();
        w_copy(nn, P + (size_t)k * 3 * nn + nn, Dk);
        if (k > 0) {
            w_copy(nn, S + (size_t)k * 3 * nn, L);
            w_copy(nn, P + (size_t)(k - 1) * 3 * nn + nn, Dm);
        }
        if (k < N - 1) {
            const T* Sn = S + (size_t)(k + 1) * 3 * nn;                 // phi_{k+1}, transposed on load (:36-43)
            for (int e = threadIdx.x; e < nn; e += SCH_THREADS) { const int i = e % n, j = e / n; Rt[j + i * n] = Sn[e]; }
            w_copy(nn, P + (size_t)(k + 1) * 3 * nn + nn, Dp);
        }
        __syncthreads();
        if (k > 0) {
            w_gemm(n, n, n, Dk, L, t1, false);                             // :100
            __syncthreads();
            w_gemm(n, n, n, t1, Dm, t2, false);                            // :102
            __syncthreads();
            w_copy(nn, t2, P + (size_t)k * 3 * nn, (T)-1);                  // Pinv[k,0]  :106-113
            __syncthreads();
        }
        if (k < N - 1) {
            w_gemm(n, n, n, Dk, Rt, t1, false);                            // :121
            __syncthreads();
            w_gemm(n, n, n, t1, Dp, t2, false);                            // :123
            __syncthreads();
            w_copy(nn, t2, P + (size_t)k * 3 * nn + 2 * nn, (T)-1);         // Pinv[k,2]  :127-134
        }
    }
}

template <typename T>
struct DzArgsT { const T* Ginv; const T* C; const T* g; const T* lambda; T* dz; int n; int m; int N; int batch; };
typedef DzArgsT<float> DzArgs;

// include/common/dz.cuh:3-121: dz_x = Qi (q - lambda_k - Abar^T lambda_{k+1}), dz_u = Ri (r - Bbar^T lambda_{k+1})
template <int NN_, int MM_, typename T = float>
__global__ __launch_bounds__(SCH_THREADS) void compute_dz_kernel(DzArgsT<T> a) {
    __shared__ T sm[64];
    constexpr int n = NN_, m = MM_, nn = n * n, mm = m * m, nm = n * m;
    const int N = a.N;
    const size_t Gsz = (size_t)(nn + mm) * N - mm, Csz = (size_t)(nn + nm) * (N - 1), gsz = (size_t)(n + m) * N - m;
    T* tx = sm;          // n
    T* tu = sm + 16;     // m
    for (long item = blockIdx.x; item < (long)a.batch * N; item += gridDim.x) {
        const int b = (int)(item / N), k = (int)(item % N);
        const T* Qi = a.Ginv + (size_t)b * Gsz + (size_t)k * (nn + mm);
        const T* Ck = a.C + (size_t)b * Csz + (size_t)k * (nn + nm);
        const T* gk = a.g + (size_t)b * gsz + (size_t)k * (n + m);
        const T* lam = a.lambda + (size_t)b * n * N;
        T* dz = a.dz + (size_t)b * gsz + (size_t)k * (n + m);
        __syncthreads();
        const int t = threadIdx.x;
        if (t < n) {                                                      // gato_ATx, matrix.cuh:10-25
            T acc = 0;
            if (k != N - 1)
                for (int i = 0; i < n; ++i) acc += Ck[t * n + i] * lam[(size_t)(k + 1) * n + i];
            tx[t] = gk[t] - (lam[(size_t)k * n + t] + acc);
        } else if (t >= 32 && t < 32 + m && k != N - 1) {
            const int j = t - 32;
            T acc = 0;
            for (int i = 0; i < n; ++i) acc += Ck[nn + j * n + i] * lam[(size_t)(k + 1) * n + i];
            tu[j] = gk[n + j] - acc;
        }
        __syncthreads();
        if (t < n) {
            T acc = 0;
            for (int c = 0; c < n; ++c) acc += Qi[t + c * n] * tx[c];
            dz[t] = acc;
        } else if (t >= 32 && t < 32 + m && k != N - 1) {
            const int j = t - 32;
            T acc = 0;
            for (int c = 0; c < m; ++c) acc += Qi[nn + j + c * m] * tu[c];
            dz[n + j] = acc;
        }
    }
}

// ---- CSR side of the QDLDL twin ----
// prep_csr_kernel: pattern of the lower triangle of a symmetric block-tridiagonal matrix, exactly
// include/utils/csr.cuh:40-73 (row (k,i) holds (k>0)*n + i+1 entries, first column (k>0)*(k-1)*n).
__global__ __launch_bounds__(SCH_THREADS) void prep_csr_kernel(int n, int N, int* col_ptr, int* row_ind) {
    const int brow = n * n + (n * (n + 1)) / 2;
    for (int k = blockIdx.x; k < N; k += gridDim.x)
        for (int row = threadIdx.x; row < n; row += SCH_THREADS) {
            if (k == 0 && row == 0) col_ptr[0] = 0;
            const int tri = ((row + 1) * row) / 2;
            const int off = (k > 0) * ((n + 1) * n) / 2 + (k > 0) * (k - 1) * brow + (k > 0) * row * n + tri;
            const int len = (k > 0) * n + row + 1;
            col_ptr[k * n + row + 1] = off + len;
            for (int c = 0; c < len; ++c) row_ind[off + c] = (k > 0) * (k - 1) * n + c;
        }
}
// values: what form_schur_qdl_kernel leaves in d_val (include/qdldl/linsys_setup.cuh:12-336 via
// store_block_csr_lowertri, include/utils/csr.cuh:9-36), gathered from the bd-layout S that
// mpcg_form_schur produced: left block S[k,0] then the lower triangle of S[k,1], scaled by mult.
struct CsrArgs { const float* S; float* val; float mult; int n; int N; int batch; };
__global__ __launch_bounds__(SCH_THREADS) void bd_to_csr_kernel(CsrArgs a) {
    const int n = a.n, N = a.N, nn = n * n;
    const int brow = nn + (n * (n + 1)) / 2;
    const size_t nnz = (size_t)(N - 1) * nn + (size_t)N * ((n * (n + 1)) / 2);
    for (long item = blockIdx.x; item < (long)a.batch * N; item += gridDim.x) {
        const int b = (int)(item / N), k = (int)(item % N);
        const float* Sk = a.S + ((size_t)b * N + k) * 3 * nn;
        float* val = a.val + (size_t)b * nnz;
        const int per_row_max = n + n;
        for (int e = threadIdx.x; e < n * per_row_max; e += SCH_THREADS) {
            const int row = e / per_row_max, c = e % per_row_max;
            const int tri = ((row + 1) * row) / 2;
            const int off = (k > 0) * ((n + 1) * n) / 2 + (k > 0) * (k - 1) * brow + (k > 0) * row * n + tri;
            if (k > 0 && c < n) val[off + c] = a.mult * Sk[row + c * n];                               // left block
            else if (c >= n && c - n <= row) val[off + (k > 0) * n + (c - n)] = a.mult * Sk[nn + row + (c - n) * n];   // diagonal block
        }
    }
}

#pragma clang fp contract(fast)

}  // namespace mpcg
