// Issue cost of the instructions the Schur kernels are made of (round 4): v_mul_f32_dpp row_newbcast, v_pk_add_f32, v_pk_mul_f32, plain
// v_mul_f32 / v_add_f32 and the 2:1 mix (2 x mul_dpp + 1 x pk_add) of the products — 1, 2 and 3 wavefronts per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/_prof/dpp_rate.hip -o tools/_prof/dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* t, const float* in, int iters) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 1], a2 = in[threadIdx.x + 2], a3 = in[threadIdx.x + 3];
    float b0 = a1 * 3.f, b1 = a2 * 3.f, b2 = a3 * 3.f, b3 = a0 * 3.f, c0 = b0 + 1.f, c1 = b1 + 1.f, c2 = b2 + 1.f, c3 = b3 + 1.f;
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)        // 64 x v_mul_f32_dpp (independent destinations d0..d7 rotate)
            asm volatile(REP16("v_mul_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %10, %11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                               "v_mul_f32_dpp %2, %8, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %10, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        else if (MODE == 1)   // 64 x v_mul_f32
            asm volatile(REP16("v_mul_f32 %0, %8, %9\n v_mul_f32 %1, %10, %11\n v_mul_f32 %2, %8, %11\n v_mul_f32 %3, %10, %9\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        else if (MODE == 2)   // 64 x v_pk_add_f32
            asm volatile(REP16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n")
                         : "+v"(*(double*)&d0), "+v"(*(double*)&d2), "+v"(*(double*)&d4), "+v"(*(double*)&d6) : "v"(*(double*)&a0), "v"(*(double*)&b0));
        else if (MODE == 3)   // the product mix: 16 x (2 mul_dpp + pk_add) = 48 instructions
            asm volatile(REP16("v_mul_f32_dpp %4, %6, %7 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                               "v_pk_add_f32 %0, %0, %2\n")
                         : "+v"(*(double*)&d0), "+v"(*(double*)&d2), "+v"(*(double*)&d4), "+v"(*(double*)&d6), "+v"(c0), "+v"(c1) : "v"(a0), "v"(b0), "v"(b1));
        else if (MODE == 4)   // 64 x v_add_f32
            asm volatile(REP16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        else if (MODE == 5)   // 64 x v_fmac_f32_dpp
            asm volatile(REP16("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %10, %11 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %8, %11 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %10, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
        else if (MODE == 6)   // 64 x v_mov_b32_dpp
            asm volatile(REP16("v_mov_b32_dpp %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                               "v_mov_b32_dpp %2, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + c0 + c1 + c2 + c3;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float *out, *in; long long* t; (void)hipMalloc(&out, 16 * 1024 * 1024); (void)hipMalloc(&t, 8 * 16 * 4096); (void)hipMalloc(&in, 8192);
    float hin[2048]; for (int i = 0; i < 2048; ++i) hin[i] = 0.001f * (i % 17) + 0.5f;
    (void)hipMemcpy(in, hin, 8192, hipMemcpyHostToDevice);
    long long h[16];
    const int iters = 4000;
    const char* nm[7] = {"v_mul_f32_dpp row_newbcast", "v_mul_f32", "v_pk_add_f32", "2 x mul_dpp + pk_add", "v_add_f32", "v_fmac_f32_dpp", "v_mov_b32_dpp"};
    const int per[7] = {64, 64, 64, 48, 64, 64, 64};
    for (int wps : {1, 2, 3}) for (int mode = 0; mode < 7; ++mode) {
        // one workgroup of 4 waves per CU and launch 256 * wps workgroups: wps waves per SIMD
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
                default: hipLaunchKernelGGL(k<6>, dim3(256 * wps), dim3(256), 0, 0, out, t, in, iters); break;
            }
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        // s_memtime ticks at 100 MHz on this chip?  report ticks per instruction per wave and, with wps waves sharing the SIMD, per SIMD
        printf("%-28s %d waves/SIMD: %.3f ticks per instruction per wave => %.3f ticks per instruction per SIMD\n", nm[mode], wps,
               (double)h[0] / iters / per[mode], (double)h[0] / iters / per[mode] / wps);
    }
    return 0;
}
