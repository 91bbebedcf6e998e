// Micro-benchmark: issue cost (s_memtime ticks per wave instruction) of v_fma_f32 vs v_pk_fma_f32 on gfx950,
// 1 or 2 waves per SIMD, independent accumulator chains (no dependency stalls).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, float a0) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i; p[i] = f2{a0 + i, a0 - i}; }
    float m = a0 * 1.0001f; f2 mp = {m, m * 1.1f};
    __syncthreads();
    long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                         "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(mp));)
        } else {   // pk with op_sel broadcast of the low half
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %3 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %4, %4, %8, %4 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %5 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %6 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %7 op_sel_hi:[1,0,1]"
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(mp));)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float* out; long long* t; hipMalloc(&out, 4 * 1024 * 512); hipMalloc(&t, 8 * 16 * 1024);
    long long h[16];
    const char* names[3] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel bcast"};
    for (int threads : {64, 256, 512}) for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, t, 1.0f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, t, 1.0f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, t, 1.0f);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
        const double n = 64.0 * 16 * 8;
        printf("%-28s waves/WG %d (per SIMD %.2g): %.2f ticks per wave-instruction (wave 0), %.2f per SIMD-issue\n", names[mode], threads / 64,
               threads / 256.0 < 1 ? 0.25 * (threads / 64) : threads / 256.0, h[0] / n, h[0] / n / (threads >= 256 ? threads / 256.0 : 1));
    }
    return 0;
}
