// Calibration: s_memtime ticks vs hipEvent wall time, and VALU issue rate vs waves per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void k(float* out, long long* t, float a0, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i; p[i] = f2{a0 + i, a0 - i}; }
    float m = a0 * 1.0001f; f2 mp = {m, m * 1.1f};
    __syncthreads();
    long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));)
        } else {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(mp));)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float* out; long long* t; (void)hipMalloc(&out, 4 * 1024 * 1024); (void)hipMalloc(&t, 8 * 16 * 1024);
    long long h[16];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {64, 256, 512, 1024}) for (int wgs : {256, 512}) for (int mode = 0; mode < 2; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(threads), 0, 0, out, t, 1.0f, iters);
            else hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(threads), 0, 0, out, t, 1.0f, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)iters * 16 * 8;
        const double waves_per_simd = (double)threads / 64 * wgs / 256 / 4;
        const double flops = n * 64 * (mode ? 4 : 2) * (threads / 64) * wgs;
        printf("%-13s %2d waves/WG x %3d WGs (%.2f waves/SIMD): %6.2f ticks/instr/wave, kernel %.3f ms -> tick = %.3f ns (%.2f GHz), %.1f TFLOP/s\n",
               mode ? "v_pk_fma_f32" : "v_fma_f32", threads / 64, wgs, waves_per_simd, h[0] / n, ms, ms * 1e6 / h[0], h[0] / (ms * 1e6), flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
