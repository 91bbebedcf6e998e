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

// LDS layout, compile-time (NMAX = 64 NWR knots whatever the actual horizon, so that every buffer sits at a constant
// offset from ONE per-lane knot address): p[NMAX][14] | r[NMAX][14] | lambda[NMAX][14] | three part-vectors of NMAX+1
// knots (knot N = dump of idle lanes): yD[k] = D_k x_k, yL[k] = L_k x_{k-1} (knot 0 stays zero), yT[k] = L_{k+1}^T x_{k+1}
// (knot N-1 stays zero) | 2 NW wave partials.
template <int NWR> struct LpbLds {
    static constexpr int NMAX = 64 * NWR, NW = 4 * NWR;
    static constexpr int VS = NMAX * NS;                       // floats per vector (a multiple of 4)
    static constexpr int PS = (int)r4((size_t)(NMAX + 1) * NS);
    static constexpr int XP = 0, XR = VS, LAM = 2 * VS, YD = 3 * VS, YL = YD + PS, YT = YL + PS, RED = YT + PS,
                         TOTAL = RED + (int)r4(2 * NW);
};
__host__ __device__ constexpr size_t pcg_lpb_lds_floats(int N, int NW) {
    return NW == 4 ? (size_t)LpbLds<1>::TOTAL : (size_t)LpbLds<2>::TOTAL;
}

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lpb_kernel(PcgArgs a) {
    typedef LpbLds<NWR> L;
    constexpr int NW = 4 * NWR, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = a.order ? (int)a.order[blockIdx.x] : (int)blockIdx.x;     // (dispatch order: longest-expected first, sched_order_kernel)
    // fix-up launch behind a cluster kernel: only the flagged trajectories run
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* red_v = lds + L::RED;
    float* red_e = red_v + NW;

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- role of this wave, block of this lane ----
    const int role = w / NWR;                          // wave-uniform
    const bool isP = role >= 2, isL = (role & 1) == 0;
    const int k = 64 * (w - role * NWR) + lane + (isL ? 1 : 0);
    const bool p3 = a.pcols == 3;
    const bool wave_on = !(isP && isL && !p3);         // block-Jacobi: no off-diagonal Pinv blocks
    const bool valid = wave_on && k < N;
    const int kk = valid ? k : N;                      // knot this lane writes (N = dump)
    const int kx = k < N ? k : N - 1;                  // knot this lane reads (clamped: its block is all-zero); >= 1 on off-diagonal lanes
    float* const xk = lds + kx * NS;                   // THE per-lane address: knot kx of buffer B is at xk + L::B
    float* const wk = lds + kk * NS;                   // same for the knot it writes

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

    // ---- stage vectors: p <- lambda0 (operand of the setup product), lambda <- lambda0, r <- gamma, parts <- 0 ----
    for (int e = tid; e < 3 * L::PS; e += NTHR) lds[L::YD + e] = 0.f;
    for (int e = tid; e < N * NS; e += NTHR) {
        const float l0 = lam_g[e];
        lds[L::XP + e] = l0;
        lds[L::LAM + e] = l0;
        lds[L::XR + e] = gam[e];
    }
    lds_barrier();

#ifdef MPCG_PROF
    bool prof_on = false;
    int prof_base = 0;
#endif
    // fold the 64 lanes' partials into lane 0: four DPP adds inside each 16-lane row + three readlanes (fixed order)
    auto wave_fold = [&](float part) -> float {
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
    // One pass of this wave's matrix over the vector at buffer offset X; the wave's share of x^T M x goes to red[w].
    // All LDS operands are requested up front; the products are pure VALU chains (one wave issues a v_pk_fma_f32 every
    // ~4.9 cycles = 80 % of its SIMD's fp32 peak, tools/_prof/fma_rate2.hip).
    auto pass = [&](int X, float* red) {
        MPCG_STAMP(prof_base + 0);
        f2 xa[7], xb[7], acc[7];
        {
            const f2* xa2 = reinterpret_cast<const f2*>(xk + X + (isL ? -NS : 0));   // off-diagonal: knot k-1, diagonal: knot k
            const f2* xb2 = reinterpret_cast<const f2*>(xk + X);                      // knot k
#pragma unroll
            for (int i = 0; i < 7; ++i) { xa[i] = xa2[i]; xb[i] = xb2[i]; acc[i] = f2{0.f, 0.f}; }
        }
#if !(defined(MPCG_ABLATE) && (MPCG_ABLATE & 1))     // (timing experiments only: tools/lpb_ablate.sh)
#pragma unroll
        for (int u = 0; u < NS; ++u) {                   // direct: M xa
            const float xs = (u & 1) ? xa[u >> 1].y : xa[u >> 1].x;
#pragma unroll
            for (int i = 0; i < 7; ++i) acc[i] = __builtin_elementwise_fma(mp(u, i), f2{xs, xs}, acc[i]);
        }
#endif
        // x_k . (M xa) in two chains; an off-diagonal lane counts it twice: x_{k-1} . (L_k^T x_k) is the same number
        f2 dt0 = {0.f, 0.f}, dt1 = {0.f, 0.f};
        f2* yo2 = reinterpret_cast<f2*>(wk + (isL ? L::YL : L::YD));
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            yo2[i] = acc[i];
            if (i & 1) dt1 = __builtin_elementwise_fma(acc[i], xb[i], dt1);
            else dt0 = __builtin_elementwise_fma(acc[i], xb[i], dt0);
        }
        const f2 dt = dt0 + dt1;
        MPCG_STAMP(prof_base + 1);
        // the wave partial is published BEFORE the transposed product: the fold's readlanes and the LDS write's
        // latency hide behind 98 more FMAs instead of sitting in front of the barrier
#if defined(MPCG_ABLATE) && (MPCG_ABLATE & 8)
        const float part = dt.x + dt.y;
#else
        const float part = wave_fold(isL ? 2.f * (dt.x + dt.y) : dt.x + dt.y);
#endif
        if (lane == 0) red[w] = part;
        MPCG_STAMP(prof_base + 2);
#if defined(MPCG_ABLATE) && (MPCG_ABLATE & 2)
        if (false) {
#else
        if (isL) {
#endif
            // transposed: yT[k-1] = L_k^T x_k
            // (four independent chains: a wave issues a v_pk_fma_f32 every ~5 cycles, its result is ready after ~2 issues)
            f2* yt2 = reinterpret_cast<f2*>(wk + L::YT + (valid ? -NS : 0));
#pragma unroll
            for (int u = 0; u < 12; u += 4) {
                f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f}, t2 = {0.f, 0.f}, t3 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    t0 = __builtin_elementwise_fma(mp(u, i), xb[i], t0);
                    t1 = __builtin_elementwise_fma(mp(u + 1, i), xb[i], t1);
                    t2 = __builtin_elementwise_fma(mp(u + 2, i), xb[i], t2);
                    t3 = __builtin_elementwise_fma(mp(u + 3, i), xb[i], t3);
                }
                yt2[u >> 1] = f2{t0.x + t0.y, t1.x + t1.y};
                yt2[(u >> 1) + 1] = f2{t2.x + t2.y, t3.x + t3.y};
            }
            {
                f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    t0 = __builtin_elementwise_fma(mp(12, i), xb[i], t0);
                    t1 = __builtin_elementwise_fma(mp(13, i), xb[i], t1);
                }
                yt2[6] = f2{t0.x + t0.y, t1.x + t1.y};
            }
        }
        MPCG_STAMP(prof_base + 3);
    };
    // sum of the NW wave partials, same order in every thread: deterministic
    auto sum_red = [&](const f4 (&v)[NW / 4]) -> float {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) { s += v[i].x; s += v[i].y; s += v[i].z; s += v[i].w; }
        return s;
    };
    const int NV2 = N * (NS / 2);
    f2* xp2 = reinterpret_cast<f2*>(lds + L::XP);
    f2* xr2 = reinterpret_cast<f2*>(lds + L::XR);
    f2* lam2 = reinterpret_cast<f2*>(lds + L::LAM);
    const f2* yD2 = reinterpret_cast<const f2*>(lds + L::YD);
    const f2* yL2 = reinterpret_cast<const f2*>(lds + L::YL);
    const f2* yT2 = reinterpret_cast<const f2*>(lds + L::YT);
    // element-wise phases: float2 item e of an [N][14] vector; every thread owns items tid and tid + NTHR
    // (7 N <= 2 NTHR), all LDS operands of a phase — the wave partials included — are requested before the first use
    const int e0 = tid < NV2 ? tid : 0, e1 = tid + NTHR < NV2 ? tid + NTHR : e0;
    const bool ok0 = tid < NV2, ok1 = tid + NTHR < NV2;

    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
    if (lane == 0) { red_v[w] = 0.f; red_e[w] = 0.f; }   // (waves that sit a pass out leave their slot at zero)
    if (!isP) pass(L::XP, red_v);
    lds_barrier();
    for (int e = tid; e < NV2; e += NTHR) xr2[e] = xr2[e] - ((yD2[e] + yL2[e]) + yT2[e]);
    lds_barrier();
    if (isP && wave_on) pass(L::XR, red_e);
    lds_barrier();
    float eta;
    {
        f4 rv[NW / 4];
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) rv[i] = *reinterpret_cast<const f4*>(red_e + 4 * i);
        eta = sum_red(rv);
    }
    for (int e = tid; e < NV2; e += NTHR) xp2[e] = p3 ? (yD2[e] + yL2[e]) + yT2[e] : yD2[e];
    lds_barrier();
    // The matrix loads have all been consumed by now on the waves that ran a setup pass, but not on every static path
    // (block-Jacobi leaves the Pinv off-diagonal waves idle): without this the compiler keeps an s_waitcnt vmcnt(n)
    // in front of every second FMA of the loop — 49 extra issue slots per product.
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)

    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    if (fabsf(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
            prof_on = b == 0 && it == 20;
            prof_base = 0;
#endif
            // upsilon = S p ; v = p . upsilon
            if (!isP) pass(L::XP, red_v);
            MPCG_STAMP(4);
            lds_barrier();
            MPCG_STAMP(5);
            // alpha = eta / v ; r -= alpha upsilon      (lambda += alpha p is not needed before the exit: the S waves,
            // idle during the Pinv pass, do it there)
            float alpha;
            {
                f4 rv[NW / 4];
#pragma unroll
                for (int i = 0; i < NW / 4; ++i) rv[i] = *reinterpret_cast<const f4*>(red_v + 4 * i);
                const f2 d0 = yD2[e0], l0 = yL2[e0], t0 = yT2[e0], r0 = xr2[e0];
                const f2 d1 = yD2[e1], l1 = yL2[e1], t1 = yT2[e1], r1 = xr2[e1];
                alpha = eta / sum_red(rv);
#if !(defined(MPCG_ABLATE) && (MPCG_ABLATE & 4))
                if (ok0) xr2[e0] = r0 - alpha * ((d0 + l0) + t0);
                if (ok1) xr2[e1] = r1 - alpha * ((d1 + l1) + t1);
#endif
            }
            MPCG_STAMP(6);
            lds_barrier();
            MPCG_STAMP(7);
#ifdef MPCG_PROF
            prof_base = 8;
#endif
            // r~ = Pinv r ; eta' = r . r~          | S waves: lambda += alpha p
            if (isP) {
                if (wave_on) pass(L::XR, red_e);
            } else {
                for (int e = tid; e < NV2; e += NTHR / 2) lam2[e] = lam2[e] + alpha * xp2[e];
            }
            MPCG_STAMP(12);
            lds_barrier();
            MPCG_STAMP(13);
            // eta' ; exit test ; p = r~ + (eta'/eta) p
            {
                f4 rv[NW / 4];
#pragma unroll
                for (int i = 0; i < NW / 4; ++i) rv[i] = *reinterpret_cast<const f4*>(red_e + 4 * i);
                f2 rt0 = yD2[e0], rt1 = yD2[e1];
                const f2 p0 = xp2[e0], p1 = xp2[e1];
                if (p3) {
                    const f2 l0 = yL2[e0], t0 = yT2[e0], l1 = yL2[e1], t1 = yT2[e1];
                    rt0 = (rt0 + l0) + t0;
                    rt1 = (rt1 + l1) + t1;
                }
                const float eta_new = sum_red(rv);
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
                const float beta = eta_new / eta;
#if !(defined(MPCG_ABLATE) && (MPCG_ABLATE & 4))
                if (ok0) xp2[e0] = rt0 + beta * p0;
                if (ok1) xp2[e1] = rt1 + beta * p1;
#endif
                eta = eta_new;
            }
            MPCG_STAMP(14);
            lds_barrier();
            MPCG_STAMP(15);
        }
    }

    // ---- write back ----
    for (int e = tid; e < N * NS; e += NTHR) {
        lam_g[e] = lds[L::LAM + e];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = lds[L::XR + e];
        if (a.p_out) a.p_out[(size_t)b * vstride + e] = lds[L::XP + e];
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
}

}  // namespace mpcg
