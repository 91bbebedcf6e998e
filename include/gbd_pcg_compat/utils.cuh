// utils.cuh — HIP-native stand-in for GBD-PCG's bd-layout helpers as the reference calls them
// (SURVEY.md §8a row F1): block (blockrow, col) of a block-tridiagonal matrix lives at
// blockrow*3*n*n + col*n*n, column-major n x n.  Block-cooperative: every thread of the block calls.
//   store_block_bd call sites: include/pcg/linsys_setup.cuh:106-113, 202-210, 249-255, 491-507, 518-524, 551-557
//   load_block_bd  call sites: include/pcg/linsys_setup.cuh:36-92
//   gato_memcpy    call sites: include/qdldl/linsys_setup.cuh:67-70 (dst, src, n)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

template <typename T>
__device__ void store_block_bd(uint32_t b_dim, uint32_t m_dim, const T* src, T* dst, unsigned col, unsigned blockrow, int multiplier = 1) {
    (void)m_dim;
    T* d = dst + (size_t)blockrow * 3 * b_dim * b_dim + (size_t)col * b_dim * b_dim;
    for (unsigned e = threadIdx.x; e < b_dim * b_dim; e += blockDim.x) d[e] = src[e] * static_cast<T>(multiplier);
}

template <typename T>
__device__ void load_block_bd(uint32_t b_dim, uint32_t m_dim, const T* src, T* dst, unsigned col, unsigned blockrow, bool transpose = false) {
    (void)m_dim;
    const T* s = src + (size_t)blockrow * 3 * b_dim * b_dim + (size_t)col * b_dim * b_dim;
    for (unsigned e = threadIdx.x; e < b_dim * b_dim; e += blockDim.x) {
        const unsigned i = e % b_dim, j = e / b_dim;
        dst[transpose ? (j + i * b_dim) : e] = s[e];
    }
}

template <typename T>
__device__ void gato_memcpy(T* dst, const T* src, unsigned n) {
    for (unsigned e = threadIdx.x; e < n; e += blockDim.x) dst[e] = src[e];
}
