// probe: what do the cross-lane VALU ops do on this chip?  prints source lane per destination lane
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int a = lane, r;
    r = __builtin_amdgcn_update_dpp(-1, a, 0x130, 0xf, 0xf, true);  out[0 * 64 + lane] = r;   // wave_shl:1
    r = __builtin_amdgcn_update_dpp(-1, a, 0x138, 0xf, 0xf, true);  out[1 * 64 + lane] = r;   // wave_shr:1
    r = __builtin_amdgcn_update_dpp(-1, a, 0x134, 0xf, 0xf, true);  out[2 * 64 + lane] = r;   // wave_rol:1
    r = __builtin_amdgcn_update_dpp(-1, a, 0x155, 0xf, 0xf, false); out[3 * 64 + lane] = r;   // row_newbcast:5
    r = __builtin_amdgcn_update_dpp(-1, a, 0x101, 0xf, 0xf, true);  out[4 * 64 + lane] = r;   // row_shl:1
    int x = lane, y = 100 + lane;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    out[5 * 64 + lane] = x; out[6 * 64 + lane] = y;
    x = lane; y = 100 + lane;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    out[7 * 64 + lane] = x; out[8 * 64 + lane] = y;
}
int main() {
    int* d; hipMalloc(&d, 9 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[9] = {"wave_shl1", "wave_shr1", "wave_rol1", "row_newbcast5", "row_shl1", "pl16swap.x", "pl16swap.y", "pl32swap.x", "pl32swap.y"};
    for (int i = 0; i < 9; ++i) { printf("%-14s", nm[i]); for (int l = 0; l < 64; ++l) printf(" %d", h[i * 64 + l]); printf("\n"); }
    return 0;
}
