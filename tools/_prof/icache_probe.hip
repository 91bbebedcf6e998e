// Does a loop body of tens of KB run slower than a small one on gfx950 (instruction-cache reach)?  UNR copies of gemm_nn<14,14> (2.9 KB of
// code each) per loop trip; ticks per gemm per wave at 2 waves per SIMD, 1-wave workgroups (like the Schur walking kernel) spread over all CUs.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mpcgpu_amd/csrc tools/_prof/icache_probe.hip -o tools/_prof/icache_probe
#include "schur_walk.hip.h"
#include <stdio.h>
using namespace mpcg::sw;
template <int UNR>
__global__ __launch_bounds__(64, 2) void k(float* out, long long* t, const float* in, int iters) {
    const int lr = threadIdx.x & 15;
    f2 A[7], B[7], C[7];
    for (int j = 0; j < 7; ++j) { A[j] = f2{in[lr + 16 * j], in[lr + 16 * j + 8]}; B[j] = f2{in[lr + 3 * j + 1], in[lr + 5 * j + 2]}; }
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        SFor<0, UNR>::run([&](auto) {
            gemm_nn<14, 14>(A, B, C);
            for (int j = 0; j < 7; ++j) A[j] = C[j] * f2{0.25f, 0.25f};
            asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]));
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int j = 0; j < 7; ++j) s += A[j].x + A[j].y + B[j].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int UNR> void run(float* out, long long* t, const float* in, int wps) {
    const int total = 96;                    // gemms per measurement
    const int iters = total * 40 / UNR;
    long long h[8];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) { (void)hipEventRecord(e0, 0); hipLaunchKernelGGL(k<UNR>, dim3(1024 * wps), dim3(64), 0, 0, out, t, in, iters); (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1); }
    (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("loop body of %2d gemm_nn (~%5.1f KB): %d waves/SIMD: %8.1f ticks per gemm per wave; kernel %.1f us => %.0f ticks per us (wave 0: %lld ticks)\n", UNR, UNR * 2.95, wps,
           (double)h[0] / (iters * UNR), ms * 1e3, (double)h[0] / (ms * 1e3), h[0]);
}
int main() {
    float *out, *in; long long* t; (void)hipMalloc(&out, 16 * 1024 * 1024); (void)hipMalloc(&t, 8 * 8192); (void)hipMalloc(&in, 8192);
    float hin[2048]; for (int i = 0; i < 2048; ++i) hin[i] = 0.01f * ((i * 7) % 23) + 0.1f;
    (void)hipMemcpy(in, hin, 8192, hipMemcpyHostToDevice);
    for (int wps : {1, 2}) { run<1>(out, t, in, wps); run<4>(out, t, in, wps); run<8>(out, t, in, wps); run<12>(out, t, in, wps); run<16>(out, t, in, wps); run<24>(out, t, in, wps); }
    return 0;
}
