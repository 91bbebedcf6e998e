// Pure read ceiling of the chip for a 1,285 MB array (the SpMV leg's working set): 16-byte loads, plain and nontemporal, several grid sizes.
// hipcc --offload-arch=gfx950 -O3 tools/_prof/read_rate.hip -o tools/_prof/read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, float* out, size_t n4) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * stride) : in[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
int main() {
    const size_t bytes = (size_t)1285 << 20, n4 = bytes / 16;
    f4* in; float* out;
    (void)hipMalloc(&in, bytes); (void)hipMalloc(&out, 4); (void)hipMemset(in, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int nt = 0; nt < 2; ++nt)
        for (int wgs : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL((k<true, 4>), dim3(wgs), dim3(256), 0, 0, in, out, n4);
                else hipLaunchKernelGGL((k<false, 4>), dim3(wgs), dim3(256), 0, 0, in, out, n4);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%s loads, %5d workgroups x 256: %.3f ms  %.0f GB/s\n", nt ? "nontemporal" : "plain      ", wgs, best, bytes / best / 1e6);
        }
    return 0;
}
