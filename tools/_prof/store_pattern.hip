// What does the MEMORY side charge for form_schur's output pattern?  Pure store kernels, no arithmetic: 2048 wavefronts x 4 groups, every group
// walks 16 block rows of a [1024][128][...] array set and writes, per row, the bytes the walking kernel writes (S row 2352 B, Pinv row 2352 B,
// Ginv 980 B, gamma 56 B) as 16-byte pieces from 16 lanes... in three shapes:
//   mode 0  one contiguous stream per wavefront (reference: what the HBM takes for 752 MB)
//   mode 1  the kernel's pattern: per row 784-byte runs (S left / S diag / previous row's S right; the same for Pinv), 980 B of Ginv, 56 B of gamma
//   mode 2  whole 2352-byte rows of S and Pinv in one go (what buffering the right blocks for one row would give)
// hipcc --offload-arch=gfx950 -O3 tools/_prof/store_pattern.hip -o tools/_prof/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int N = 128, B = 1024, L = 16, ROWB = 2352, BLKB = 784, GB = 980, CH = N / L;
__device__ __forceinline__ void run(char* p, int bytes, int lane, f4 v) {      // `bytes` contiguous bytes by one wavefront, 16 per lane
    for (int o = lane * 16; o + 16 <= bytes; o += 1024) *reinterpret_cast<f4*>(p + o) = v;
}
template <int MODE>
__global__ __launch_bounds__(64) void k(char* S, char* P, char* G, char* gam, float seed) {
    const int lane = threadIdx.x;
    const f4 v = {seed, seed + 1.f, seed + 2.f, seed + lane};
    if (MODE == 0) {
        const size_t per = (size_t)4 * L * (2 * ROWB + GB + 64);
        char* base = S + (size_t)blockIdx.x * per;
        for (size_t o = (size_t)lane * 16; o + 16 <= per; o += 1024) *reinterpret_cast<f4*>(base + o) = v;
        return;
    }
    for (int s = 0; s < L; ++s)
        for (int g = 0; g < 4; ++g) {                      // the four groups of a wavefront: four consecutive chunks (wave-uniform here: the runs are issued one after the other)
            const int item = blockIdx.x * 4 + g, b = item / CH, j = item % CH, kk = j * L + s;
            const size_t row = ((size_t)b * N + kk);
            char* Sr = S + row * ROWB; char* Pr = P + row * ROWB;
            run(G + row * 1024, GB - 4, lane, v);
            if (lane < 4) *reinterpret_cast<f4*>(gam + row * 64 + lane * 16) = v;
            if (MODE == 1) {
                run(Sr, BLKB, lane, v); run(Sr + BLKB, BLKB, lane, v); if (kk) run(Sr - ROWB + 2 * BLKB, BLKB, lane, v);
                run(Pr, BLKB, lane, v); run(Pr + BLKB, BLKB, lane, v); if (kk) run(Pr - ROWB + 2 * BLKB, BLKB, lane, v);
            } else {
                run(Sr, ROWB, lane, v); run(Pr, ROWB, lane, v);
            }
        }
}
int main() {
    const size_t szS = (size_t)B * N * ROWB, szG = (size_t)B * N * 1024, szg = (size_t)B * N * 64;
    char *S, *P, *G, *g;
    (void)hipMalloc(&S, (size_t)2048 * 4 * L * (2 * ROWB + GB + 64) + (1 << 20)); (void)hipMalloc(&P, szS); (void)hipMalloc(&G, szG); (void)hipMalloc(&g, szg);
    const double bytes = (double)B * N * (2.0 * ROWB + GB + 56);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(64), 0, 0, S, P, G, g, 1.f * rep);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(64), 0, 0, S, P, G, g, 1.f * rep);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(64), 0, 0, S, P, G, g, 1.f * rep);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.3f ms  %.0f GB/s for the %.0f MB of form_schur's outputs\n", mode, ms, bytes / ms / 1e6, bytes / 1e6);
        }
    return 0;
}
