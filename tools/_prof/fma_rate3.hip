// Issue cost of v_pk_fma_f32 / v_fma_f32 with REALISTIC operands (distinct matrix registers per instruction, as in
// pcg_lpb_kernel): does the register-file read traffic (3 x 64-bit sources) slow the packed form down?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, const float* in, int iters) {
    f2 m[56]; f2 acc[7]; f2 x[7];
    const float* inl = in + (threadIdx.x & 63) * 3;
    for (int i = 0; i < 56; ++i) m[i] = f2{inl[i], inl[i + 56]};
    for (int i = 0; i < 7; ++i) { acc[i] = f2{0.f, 0.f}; x[i] = f2{inl[i + 3], inl[i + 9]}; }
    __syncthreads();
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                if (MODE == 0) {          // packed, x broadcast (direct product)
                    const float xs = (u & 1) ? x[(u >> 1) % 7].y : x[(u >> 1) % 7].x;
                    acc[i] = __builtin_elementwise_fma(m[u * 7 + i], f2{xs, xs}, acc[i]);
                } else if (MODE == 1) {   // packed, x pair (transposed product)
                    acc[(u * 7 + i) % 4] = __builtin_elementwise_fma(m[u * 7 + i], x[i], acc[(u * 7 + i) % 4]);
                } else {                  // scalar FMAs, same flops as MODE 0
                    const float xs = (u & 1) ? x[(u >> 1) % 7].y : x[(u >> 1) % 7].x;
                    acc[i].x = __builtin_fmaf(m[u * 7 + i].x, xs, acc[i].x);
                    acc[i].y = __builtin_fmaf(m[u * 7 + i].y, xs, acc[i].y);
                }
            }
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int i = 0; i < 7; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float *out, *in; long long* t; (void)hipMalloc(&out, 4 * 1024 * 1024); (void)hipMalloc(&t, 8 * 16 * 1024); (void)hipMalloc(&in, 4096);
    float hin[1024]; for (int i = 0; i < 1024; ++i) hin[i] = 0.001f * (i % 17) - 0.005f;
    (void)hipMemcpy(in, hin, 4096, hipMemcpyHostToDevice);
    long long h[16];
    const int iters = 2000;
    const char* nm[3] = {"pk_fma bcast-x (direct)", "pk_fma pair-x (transposed)", "2x v_fma_f32 (scalar)"};
    for (int threads : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, t, in, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, t, in, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, t, in, iters);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-28s %d waves/SIMD: %.2f ticks per 56-pair-FMA block -> %.2f ticks per pair-FMA (wave 0), wave %d: %.2f\n", nm[mode], threads / 256,
               (double)h[0] / iters, (double)h[0] / iters / 56, threads / 64 - 1, (double)h[threads / 64 - 1] / iters / 56);
    }
    return 0;
}
