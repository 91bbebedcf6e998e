// ldl_host.hpp — host-side sparse LDL^T of the (negated) Schur matrix in the CSR-lower-triangle form the reference's
// QDLDL path uses: the SELECTABLE CPU solver behind LINSYS_SOLVE == 0 (include/mpcsim.cuh:21-25), in the product.
//
// What the reference does (include/qdldl/sqp.cuh): pattern once per SQP call — prep_csr (:164, include/utils/csr.cuh:40-73)
// and QDLDL_etree (:193) — then per SQP iteration D2H(values, gamma), qdldl_solve_schur = QDLDL_factor + QDLDL_solve
// (:22-49), H2D(lambda) (:261-282).  QDLDL itself is the un-vendored submodule osqp/qdldl (float / int32 build,
// reference Makefile:16); this file implements the same published algorithm from scratch: elimination tree of the
// upper-triangular CSC pattern (= our lower-triangular CSR rows), up-looking numeric factorisation that computes one row
// of L per column of A by a sparse triangular solve along the tree, no pivoting (the matrix is definite: all D < 0
// here), then L y = b, D z = y, L^T x = z.
// This is NOT a fallback of the GPU path: nothing in libmpcg_hip calls it unless the caller selects this solver
// (mpcg_ldl_* / mpcg_qdldl_solve_schur).
#pragma once
#include <cstdint>
#include <vector>

namespace mpcg_ldl_host {

struct Ldl {
    int n = 0;                                  // state_size
    int N = 0;                                  // knot_points
    int An = 0;                                 // matrix dimension n N
    std::vector<int32_t> Ap, Ai;                // column pointers / row indices: column j holds rows <= j (upper triangle, CSC)
    std::vector<int32_t> etree, Lnz, Lp, Li;    // elimination tree, nonzeros per column of L, L pattern (filled by factor)
    std::vector<float> Lx, D, Dinv;
    std::vector<int32_t> iwork;                 // 3 An
    std::vector<uint8_t> mark;                  // An
    std::vector<float> ywork;                   // An
    std::vector<float> val, rhs, sol;           // host staging of mpcg_qdldl_solve_schur
    long sumLnz = 0;
};

// Pattern of the lower triangle of the block-tridiagonal matrix, row by row (include/utils/csr.cuh:40-73): row (k, r) holds
// the n entries of the left block (k > 0) followed by r + 1 entries of the diagonal block; read as CSC of the upper triangle.
inline void build_pattern(Ldl& w) {
    const int n = w.n, N = w.N;
    w.An = n * N;
    w.Ap.assign(w.An + 1, 0);
    const long nnz = (long)(N - 1) * n * n + (long)N * (n * (n + 1) / 2);
    w.Ai.resize(nnz);
    long p = 0;
    for (int k = 0; k < N; ++k)
        for (int r = 0; r < n; ++r) {
            const int first = k > 0 ? (k - 1) * n : 0;
            const int len = (k > 0 ? n : 0) + r + 1;
            for (int c = 0; c < len; ++c) w.Ai[p++] = first + c;
            w.Ap[k * n + r + 1] = (int32_t)p;
        }
}

// Elimination tree and column counts of L.  Returns sum(Lnz) or -1 for a malformed pattern.
inline long etree(Ldl& w) {
    const int An = w.An;
    w.etree.assign(An, -1);
    w.Lnz.assign(An, 0);
    std::vector<int32_t>& flag = w.iwork;
    flag.assign(3 * (size_t)An, 0);
    for (int j = 0; j < An; ++j) {
        flag[j] = j;
        for (int32_t p = w.Ap[j]; p < w.Ap[j + 1]; ++p) {
            int i = w.Ai[p];
            if (i > j) return -1;                       // not upper triangular
            while (flag[i] != j) {                      // climb from row i until a node already visited for column j
                if (w.etree[i] == -1) w.etree[i] = j;
                ++w.Lnz[i];
                flag[i] = j;
                i = w.etree[i];
            }
        }
    }
    long s = 0;
    for (int j = 0; j < An; ++j) s += w.Lnz[j];
    return s;
}

inline int setup(Ldl& w, int n, int N) {
    w.n = n; w.N = N;
    build_pattern(w);
    w.sumLnz = etree(w);
    if (w.sumLnz < 0) return -1;
    w.Lp.assign(w.An + 1, 0);
    w.Li.assign((size_t)w.sumLnz, 0);
    w.Lx.assign((size_t)w.sumLnz, 0.f);
    w.D.assign(w.An, 0.f);
    w.Dinv.assign(w.An, 0.f);
    w.mark.assign(w.An, 0);
    w.ywork.assign(w.An, 0.f);
    w.val.resize(w.Ai.size());
    w.rhs.resize(w.An);
    w.sol.resize(w.An);
    return 0;
}

// Numeric factorisation A = L D L^T (unit lower L stored by columns).  Returns the number of positive entries of D,
// or -1 when a pivot is exactly zero.
inline int factor(Ldl& w, const float* Ax) {
    const int An = w.An;
    int32_t* next = w.iwork.data();                 // next free slot of every column of L
    int32_t* stack = next + An;                     // nodes of the current row, in elimination order
    int32_t* path = stack + An;
    float* y = w.ywork.data();
    w.Lp[0] = 0;
    for (int j = 0; j < An; ++j) {
        w.Lp[j + 1] = w.Lp[j] + w.Lnz[j];
        next[j] = w.Lp[j];
        w.mark[j] = 0;
        y[j] = 0.f;
    }
    int positive = 0;
    for (int k = 0; k < An; ++k) {
        // scatter column k of A (rows <= k) and collect the nodes its off-diagonal entries reach along the tree
        int top = 0;
        float dk = 0.f;
        for (int32_t p = w.Ap[k]; p < w.Ap[k + 1]; ++p) {
            const int i = w.Ai[p];
            if (i == k) { dk = Ax[p]; continue; }
            y[i] = Ax[p];
            int len = 0;
            for (int t = i; t != -1 && t < k && !w.mark[t]; t = w.etree[t]) {
                path[len++] = t;
                w.mark[t] = 1;
            }
            while (len > 0) stack[An - 1 - top++] = path[--len];   // keeps topological order: parents after children
        }
        // the collected nodes were pushed path by path from the back: visit them front to back of that region
        for (int s = An - top; s < An; ++s) {
            const int c = stack[s];
            w.mark[c] = 0;
            const float yc = y[c];
            y[c] = 0.f;
            const int32_t end = next[c];
            for (int32_t q = w.Lp[c]; q < end; ++q) y[w.Li[q]] -= w.Lx[q] * yc;
            const float l = yc * w.Dinv[c];
            dk -= yc * l;
            w.Li[end] = k;
            w.Lx[end] = l;
            ++next[c];
        }
        if (dk == 0.f) return -1;
        w.D[k] = dk;
        w.Dinv[k] = 1.f / dk;
        positive += dk > 0.f;
    }
    return positive;
}

inline void solve(const Ldl& w, float* x) {
    const int An = w.An;
    for (int j = 0; j < An; ++j) {                     // L y = b
        const float xj = x[j];
        for (int32_t q = w.Lp[j]; q < w.Lp[j + 1]; ++q) x[w.Li[q]] -= w.Lx[q] * xj;
    }
    for (int j = 0; j < An; ++j) x[j] *= w.Dinv[j];    // D z = y
    for (int j = An - 1; j >= 0; --j) {                // L^T x = z
        float xj = x[j];
        for (int32_t q = w.Lp[j]; q < w.Lp[j + 1]; ++q) xj -= w.Lx[q] * x[w.Li[q]];
        x[j] = xj;
    }
}

}  // namespace mpcg_ldl_host
