// bd_utils_probe.cpp — instantiates the device templates of include/gbd_pcg_compat/utils.cuh (store_block_bd, load_block_bd,
// gato_memcpy: SURVEY.md §8a row F1) in a kernel the way the reference's Schur formation calls them
// (include/pcg/linsys_setup.cuh:36-57, 202-210, 491-507; include/qdldl/linsys_setup.cuh:67-70) and prints the resulting
// buffers as JSON; tests/test_gpu_schur.py compares them with the oracle's orc_store_block_bd / orc_load_block_bd.
#include <cstdio>
#include <vector>

#include "utils.cuh"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename T>
__global__ void probe_load(const T* src_blocks, const T* bd, T* loaded, T* copied, unsigned n, unsigned N) {
    // (second launch: block (k-1, right) is written by ANOTHER workgroup — the reference separates the two phases with a
    //  grid-wide sync, include/pcg/linsys_setup.cuh:600)
    const unsigned k = blockIdx.x;
    // read back: plain for the diagonal, transposed for the left neighbour's right block (:36-57 loads S[k-1, 2] etc.)
    load_block_bd<T>(n, N, bd, loaded + (size_t)(2 * k) * n * n, 1, k, false);
    if (k > 0) load_block_bd<T>(n, N, bd, loaded + (size_t)(2 * k + 1) * n * n, 2, k - 1, true);
    gato_memcpy<T>(copied + (size_t)k * n, src_blocks + (size_t)k * 3 * n * n, n);
}

template <typename T>
__global__ void probe(const T* src_blocks, T* bd, unsigned n, unsigned N) {
    // one workgroup per block row, as form_S_gamma_Pinv_kernel runs (grid = knot_points)
    extern __shared__ unsigned char smem_raw[];
    T* s_blk = reinterpret_cast<T*>(smem_raw);
    const unsigned k = blockIdx.x;
    for (unsigned col = 0; col < 3; ++col) {
        if ((k == 0 && col == 0) || (k == N - 1 && col == 2)) continue;      // the two slots the reference never writes
        gato_memcpy<T>(s_blk, src_blocks + (size_t)(k * 3 + col) * n * n, n * n);
        __syncthreads();
        store_block_bd<T>(n, N, s_blk, bd, col, k, col == 1 ? -1 : 1);       // diagonal negated like S (:500-507)
        __syncthreads();
    }
}

int main() {
    typedef float T;
    const unsigned n = 14, N = 5;
    const size_t blk = (size_t)n * n, total = 3 * blk * N;
    std::vector<T> src(total);
    for (size_t e = 0; e < total; ++e) src[e] = (T)(0.25 * (double)((e * 2654435761u) % 1009) - 100.0);
    T *d_src, *d_bd, *d_loaded, *d_copied;
    CHECK(hipMalloc(&d_src, total * sizeof(T)));
    CHECK(hipMalloc(&d_bd, total * sizeof(T)));
    CHECK(hipMalloc(&d_loaded, 2 * blk * N * sizeof(T)));
    CHECK(hipMalloc(&d_copied, n * N * sizeof(T)));
    CHECK(hipMemcpy(d_src, src.data(), total * sizeof(T), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_bd, 0xff, total * sizeof(T)));                            // NaN pattern: untouched slots stay recognisable
    CHECK(hipMemset(d_loaded, 0, 2 * blk * N * sizeof(T)));
    hipLaunchKernelGGL(probe<T>, dim3(N), dim3(128), blk * sizeof(T), 0, d_src, d_bd, n, N);
    hipLaunchKernelGGL(probe_load<T>, dim3(N), dim3(128), 0, 0, d_src, d_bd, d_loaded, d_copied, n, N);
    CHECK(hipDeviceSynchronize());
    std::vector<T> bd(total), loaded(2 * blk * N), copied(n * N);
    CHECK(hipMemcpy(bd.data(), d_bd, total * sizeof(T), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(loaded.data(), d_loaded, loaded.size() * sizeof(T), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(copied.data(), d_copied, copied.size() * sizeof(T), hipMemcpyDeviceToHost));
    auto dump = [](const char* name, const std::vector<T>& v, bool last) {
        printf("\"%s\": [", name);
        for (size_t i = 0; i < v.size(); ++i) { if (v[i] != v[i]) printf("%snull", i ? "," : ""); else printf("%s%.9g", i ? "," : "", (double)v[i]); }
        printf("]%s", last ? "" : ", ");
    };
    printf("{\"n\": %u, \"N\": %u, ", n, N);
    dump("src", src, false); dump("bd", bd, false); dump("loaded", loaded, false); dump("copied", copied, true);
    printf("}\n");
    return 0;
}
