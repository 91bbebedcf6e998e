// pcg_lpb.hip.h — "lane per block" PCG kernel for gfx950: the whole trajectory's S and Pinv live in the
// register file for the whole solve, ONE 14x14 BLOCK PER LANE.
//
// Why: the row-pair x block mapping of pcg_traj_kernel spends its time on LDS round trips
// and cross-lane merges (latency/issue bound, no resource above 40 %), and at N=128 a CU cannot hold
// 2 x (3N-2) blocks (602 KB) — 22 % of the matrix is re-read every iteration and costs a third of it.
// Both problems go away when
//   (1) only the block LOWER triangle is kept: S is symmetric by construction — the reference writes
//       S[k-1,right] as the transposed copy of S[k,left] (include/pcg/linsys_setup.cuh:536-557), bit for bit —
//       and so is the symmetric-stair Pinv (:97-136, up to the rounding of two independently formed products),
//       as PCG requires of both.  2 x (2N-1) blocks = 510 at N=128 = 400 KB: fits the 512 KB register file;
//   (2) a lane owns a whole block (196 VGPRs = 49 x dwordx4, loaded flat — column-major pairs of consecutive
//       rows are naturally even-aligned register pairs): block x vector AND block^T x vector are both plain
//       in-lane chains of v_pk_fma_f32 (98 each), with no cross-lane traffic at all:
//           direct      y[2i..2i+1] += (M[2i][u], M[2i+1][u]) * x[u]            (x broadcast by op_sel)
//           transposed  t[u]        += (M[2i][u], M[2i+1][u]) * (x'[2i], x'[2i+1])   then z[u] = t[u].lo + t[u].hi
//       8 waves x 64 lanes = 512 lanes hold the 510 blocks of an N=128 trajectory exactly.
// The kernel is then bound by the fp32 VALU rate of the SIMDs that hold the off-diagonal blocks.
//
// Roles (NWR waves per role, N <= 64 NWR): waves [0,NWR) S off-diagonal L_k = S[k,left], k = 1..N-1 |
//   [NWR,2NWR) S diagonal D_k | [2NWR,3NWR) Pinv off-diagonal | [3NWR,4NWR) Pinv diagonal.  Waves w and
//   w + 4 (NWR = 2) share a SIMD, so every SIMD holds one S wave and one Pinv wave of the same kind: the two
//   passes of an iteration alternate on it.
//   y_k = D_k x_k + L_k x_{k-1} + L_{k+1}^T x_{k+1}: the three parts go to three LDS vectors (yD, yL, yT) and are
//   summed by the element-wise update that follows the pass.  Inner products: every lane dots its own part with
//   the matching knot of x; x_k^T (L_k x_{k-1}) = x_{k-1}^T (L_k^T x_k), so an off-diagonal lane counts its dot twice.
//
// Blocks (k, right) are never read.  LDS: iterate vectors only (43 KB at N=128).
#pragma once
#include "pcg_kernels.hip.h"

namespace mpcg {

__host__ __device__ constexpr size_t pcg_lpb_lds_floats(int N, int NW) {
    return 2 * r4((size_t)(N + 2) * NS) + r4((size_t)N * NS) + 3 * r4((size_t)(N + 1) * NS) + r4(2 * (size_t)NW);
}

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lpb_kernel(PcgArgs a) {
    constexpr int NW = 4 * NWR, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;

    float* xp = lds;                                   // p, knot j at xp + (j+1)*NS, zero knot either side
    float* xr = xp + r4((size_t)(N + 2) * NS);         // r likewise
    float* lam = xr + r4((size_t)(N + 2) * NS);
    float* yD = lam + r4((size_t)N * NS);              // D_k x_k            knot k at yD + k*NS; knot N = dump of idle lanes
    float* yL = yD + r4((size_t)(N + 1) * NS);         // L_k x_{k-1}        (knot 0 stays zero)
    float* yT = yL + r4((size_t)(N + 1) * NS);         // L_{k+1}^T x_{k+1}  (knot N-1 stays zero)
    float* red_v = yT + r4((size_t)(N + 1) * NS);
    float* red_e = red_v + NW;

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- role of this wave, block of this lane ----
    const int role = w / NWR;                          // wave-uniform
    const bool isP = role >= 2, isL = (role & 1) == 0;
    const int k = 64 * (w - role * NWR) + lane + (isL ? 1 : 0);
    const bool wave_on = !(isP && isL && a.pcols != 3);            // block-Jacobi: no off-diagonal Pinv blocks
    const bool valid = wave_on && k < N;
    const int kk = valid ? k : N;                      // knot this lane writes (N = dump)
    const int kx = k < N ? k : N - 1;                  // knot this lane reads (clamped: its block is all-zero)

    f4 m4[BLK4];
    {
        const rsrc_t M = make_rsrc((isP ? static_cast<const float*>(a.Pinv) : static_cast<const float*>(a.S)) + (size_t)b * mstride,
                                   (uint32_t)(mstride * sizeof(float)));
        const uint32_t off = valid ? (uint32_t)(k * 3 + (isL ? 0 : 1)) * (BLK4 * 16u) : OOB_OFF;
#pragma unroll
        for (int i = 0; i < BLK4; ++i) m4[i] = buf_load4<false>(M, off + 16u * i);
    }
    // (M[2i][u], M[2i+1][u]): floats 14u + 2i, +1 of the flat block
    auto mp = [&](int u, int i) -> f2 {
        const int e = NS * u + 2 * i;
        const f4 v = m4[e >> 2];
        return (e & 2) ? f2{v.z, v.w} : f2{v.x, v.y};
    };

    // ---- stage vectors ----
    for (int e = tid; e < (N + 2) * NS; e += NTHR) { xp[e] = 0.f; xr[e] = 0.f; }
    for (int e = tid; e < (N + 1) * NS; e += NTHR) { yD[e] = 0.f; yL[e] = 0.f; yT[e] = 0.f; }
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const float l0 = lam_g[e];
        xp[NS + e] = l0;
        lam[e] = l0;
        xr[NS + e] = gam[e];
    }
    lds_barrier();

    // One pass of this wave's matrix over the padded vector xv; returns the wave's share of xv^T M xv (lane 0).
    auto pass = [&](const float* xv) -> float {
        float part;
        if (isL) {
            // direct: yL[k] = L_k x_{k-1}
            const f2* xa2 = reinterpret_cast<const f2*>(xv + kx * NS);          // knot k-1 of the padded vector
            f2 xa[7], acc[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) { xa[i] = xa2[i]; acc[i] = f2{0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const float xs = (u & 1) ? xa[u >> 1].y : xa[u >> 1].x;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc[i] = __builtin_elementwise_fma(mp(u, i), f2{xs, xs}, acc[i]);
            }
            const f2* xb2 = reinterpret_cast<const f2*>(xv + (kx + 1) * NS);    // knot k
            f2 xb[7];
            f2 dt = {0.f, 0.f};
            f2* yl2 = reinterpret_cast<f2*>(yL + kk * NS);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                xb[i] = xb2[i];
                yl2[i] = acc[i];
                dt = __builtin_elementwise_fma(acc[i], xb[i], dt);
            }
            // transposed: yT[k-1] = L_k^T x_k
            f2* yt2 = reinterpret_cast<f2*>(yT + (valid ? k - 1 : N) * NS);
#pragma unroll
            for (int u = 0; u < NS; u += 2) {
                f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    t0 = __builtin_elementwise_fma(mp(u, i), xb[i], t0);
                    t1 = __builtin_elementwise_fma(mp(u + 1, i), xb[i], t1);
                }
                yt2[u >> 1] = f2{t0.x + t0.y, t1.x + t1.y};
            }
            part = 2.f * (dt.x + dt.y);
        } else {
            const f2* xa2 = reinterpret_cast<const f2*>(xv + (kx + 1) * NS);    // knot k
            f2 xa[7], acc[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) { xa[i] = xa2[i]; acc[i] = f2{0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const float xs = (u & 1) ? xa[u >> 1].y : xa[u >> 1].x;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc[i] = __builtin_elementwise_fma(mp(u, i), f2{xs, xs}, acc[i]);
            }
            f2 dt = {0.f, 0.f};
            f2* yd2 = reinterpret_cast<f2*>(yD + kk * NS);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                yd2[i] = acc[i];
                dt = __builtin_elementwise_fma(acc[i], xa[i], dt);
            }
            part = dt.x + dt.y;
        }
        // fold the 64 lanes: four DPP adds inside each 16-lane row + three readlanes (fixed order: deterministic)
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "s_nop 1"
            : "+v"(part));
        const int pb = __builtin_bit_cast(int, part);
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 48));
        return ((part + r1) + r2) + r3;
    };
    auto block_sum = [&](const float* red) -> float {      // same order in every thread: deterministic
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i += 4) {
            const f4 v = *reinterpret_cast<const f4*>(red + i);
            s += v.x; s += v.y; s += v.z; s += v.w;
        }
        return s;
    };
    const int NV2 = N * (NS / 2);
    f2* xp2 = reinterpret_cast<f2*>(xp + NS);
    f2* xr2 = reinterpret_cast<f2*>(xr + NS);
    f2* lam2 = reinterpret_cast<f2*>(lam);
    const f2* yD2 = reinterpret_cast<const f2*>(yD);
    const f2* yL2 = reinterpret_cast<const f2*>(yL);
    const f2* yT2 = reinterpret_cast<const f2*>(yT);
    const bool p3 = a.pcols == 3;

    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
    if (!isP) (void)pass(xp);
    lds_barrier();
    for (int e = tid; e < NV2; e += NTHR) xr2[e] = xr2[e] - ((yD2[e] + yL2[e]) + yT2[e]);
    lds_barrier();
    {
        float part = 0.f;
        if (isP && wave_on) part = pass(xr);
        if (lane == 0) red_e[w] = part;
    }
    lds_barrier();
    float eta = block_sum(red_e);
    for (int e = tid; e < NV2; e += NTHR) xp2[e] = p3 ? (yD2[e] + yL2[e]) + yT2[e] : yD2[e];
    lds_barrier();

    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    if (fabsf(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
            // upsilon = S p ; v = p . upsilon
            {
                float part = 0.f;
                if (!isP) part = pass(xp);
                if (lane == 0) red_v[w] = part;
            }
            lds_barrier();
            const float alpha = eta / block_sum(red_v);
            // lambda += alpha p ; r -= alpha upsilon
            for (int e = tid; e < NV2; e += NTHR) {
                const f2 ups = (yD2[e] + yL2[e]) + yT2[e];
                lam2[e] = lam2[e] + alpha * xp2[e];
                xr2[e] = xr2[e] - alpha * ups;
            }
            lds_barrier();
            // r~ = Pinv r ; eta' = r . r~
            {
                float part = 0.f;
                if (isP && wave_on) part = pass(xr);
                if (lane == 0) red_e[w] = part;
            }
            lds_barrier();
            const float eta_new = block_sum(red_e);
            iters = (uint32_t)(it + 1);
            if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
            const float beta = eta_new / eta;
            // p = r~ + beta p
            for (int e = tid; e < NV2; e += NTHR) {
                const f2 rt = p3 ? (yD2[e] + yL2[e]) + yT2[e] : yD2[e];
                xp2[e] = rt + beta * xp2[e];
            }
            eta = eta_new;
            lds_barrier();
        }
    }

    // ---- write back ----
    for (int e = tid; e < N * NS; e += NTHR) {
        lam_g[e] = lam[e];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = xr[NS + e];
        if (a.p_out) a.p_out[(size_t)b * vstride + e] = xp[NS + e];
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
}

}  // namespace mpcg
