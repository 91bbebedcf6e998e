// Is the SpMV's load SHAPE what holds it at ~5 TB/s?  Same 1,285 MB, no arithmetic:
//   mode 0  bt_spmv_kernel's shape: a wavefront per 2352-byte block row, three loads of 49 lanes x 16 B (784 B each), rows strided over the grid
//   mode 1  the same rows, a wavefront per row, but read as ceil(2352/1024) = 3 loads of 64 lanes x 16 B from the row's start (last one partly masked)
//   mode 2  a workgroup (4 wavefronts) per tile of 8 rows = 18,816 B read as 256-lane x 16 B pieces (LDS-staged design's load side)
//   mode 3  mode 0's loads in the round-4 kernel's order: a wavefront per SPAN of 16 consecutive rows
// hipcc --offload-arch=gfx950 -O3 tools/_prof/read_pattern.hip -o tools/_prof/read_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ROWB = 2352;
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ in, float* out, long rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 || MODE == 1) {
        const long gw = (long)blockIdx.x * 4 + w, GW = (long)gridDim.x * 4;
        for (long q = gw; q < rows; q += 2 * GW) {
            f4 v[6];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const long qq = q + u * GW < rows ? q + u * GW : q;
                const char* r = in + qq * ROWB;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (MODE == 0) v[3 * u + s] = lane < 49 ? __builtin_nontemporal_load((const f4*)(r + s * 784 + lane * 16)) : f4{0, 0, 0, 0};
                    else v[3 * u + s] = s * 1024 + lane * 16 < ROWB ? __builtin_nontemporal_load((const f4*)(r + s * 1024 + lane * 16)) : f4{0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) acc += v[i];
        }
    } else if (MODE == 3) {
        // bt_spmv_kernel's round-4 ORDER: a wavefront walks a span of 16 consecutive rows (two in flight), spans strided over the grid
        const long gw = (long)blockIdx.x * 4 + w, GW = (long)gridDim.x * 4, spans = rows / 16;
        for (long sp = gw; sp < spans; sp += GW) {
            const char* r0 = in + sp * 16 * ROWB;
#pragma unroll 1
            for (int r = 0; r < 16; r += 2) {
                f4 v[6];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int s = 0; s < 3; ++s) v[3 * u + s] = lane < 49 ? __builtin_nontemporal_load((const f4*)(r0 + (r + u) * ROWB + s * 784 + lane * 16)) : f4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 6; ++i) acc += v[i];
            }
        }
    } else {
        constexpr int R = 8, TB = R * ROWB, NP = (TB + 4095) / 4096;     // pieces of 256 lanes x 16 B
        const long tiles = rows / R;
        for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
            const char* r = in + t * TB;
            f4 v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) v[p] = p * 4096 + (int)threadIdx.x * 16 < TB ? __builtin_nontemporal_load((const f4*)(r + p * 4096 + threadIdx.x * 16)) : f4{0, 0, 0, 0};
#pragma unroll
            for (int p = 0; p < NP; ++p) acc += v[p];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
int main() {
    const long rows = 4096L * 128;
    const size_t bytes = (size_t)rows * ROWB;
    char* in; float* out;
    (void)hipMalloc(&in, bytes + 4096); (void)hipMalloc(&out, 4); (void)hipMemset(in, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int wgs : {256 * 2, 256 * 3, 256 * 4, 256 * 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("mode %d, %5d workgroups x 256: %.3f ms  %.0f GB/s\n", mode, wgs, best, bytes / best / 1e6);
        }
    return 0;
}
