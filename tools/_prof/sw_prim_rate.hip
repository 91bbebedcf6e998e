// What do the rows-in-lanes primitives of schur_walk.hip.h cost in the shader's own clock?  One gemm_nn<14,14>, one gemm_nt<14,14> and one
// invert<14> per loop trip, operands in registers, 1..4 wavefronts per SIMD (blocks of 256 threads; 256 x w blocks).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mpcgpu_amd/csrc tools/_prof/sw_prim_rate.hip -o tools/_prof/sw_prim_rate
#include "schur_walk.hip.h"
#include <stdio.h>
using namespace mpcg::sw;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* t, const float* in, int iters) {
    const int lr = threadIdx.x & 15;
    f2 A[7], B[7], C[7];
    for (int j = 0; j < 7; ++j) { A[j] = f2{in[lr + 16 * j], in[lr + 16 * j + 8]}; B[j] = f2{in[lr + 3 * j + 1], in[lr + 5 * j + 2]}; }
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { gemm_nn<14, 14>(A, B, C); for (int j = 0; j < 7; ++j) A[j] = C[j] * f2{0.25f, 0.25f}; }
        else if (MODE == 1) { gemm_nt<14, 14>(A, B, C); for (int j = 0; j < 7; ++j) A[j] = C[j] * f2{0.25f, 0.25f}; }
        else { for (int j = 0; j < 7; ++j) C[j] = A[j]; add_rho<14>(C, lr, 3.0f); invert<14>(C, B, lr); for (int j = 0; j < 7; ++j) A[j] = A[j] + B[j] * f2{1e-3f, 1e-3f}; }
        asm volatile("" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int j = 0; j < 7; ++j) s += A[j].x + A[j].y + B[j].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float *out, *in; long long* t; (void)hipMalloc(&out, 16 * 1024 * 1024); (void)hipMalloc(&t, 8 * 4 * 4096); (void)hipMalloc(&in, 8192);
    float hin[2048]; for (int i = 0; i < 2048; ++i) hin[i] = 0.01f * ((i * 7) % 23) + 0.1f;
    (void)hipMemcpy(in, hin, 8192, hipMemcpyHostToDevice);
    long long h[4];
    const int iters = 400;
    const char* nm[3] = {"gemm_nn<14,14> (196 mul_dpp + 98 pk_add + 7 pk_mul)", "gemm_nt<14,14> (same mix)", "invert<14> (Gauss-Jordan, ~520 VALU)"};
    for (int wps : {1, 2, 3, 4}) for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-56s %d waves/SIMD: %8.1f ticks per call per wave, %8.1f per SIMD\n", nm[mode], wps, (double)h[0] / iters, (double)h[0] / iters / wps);
    }
    return 0;
}
