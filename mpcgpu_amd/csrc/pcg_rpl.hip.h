// pcg_rpl.hip.h — "row per lane" PCG kernel for SHORT horizons (N <= 64; the reference's real-time case is N = 32, one
// trajectory): everything that moves per iteration stays in registers and DPP rows.
//
// Why: at N = 32 an iteration of the row-pair kernel (pcg_traj_kernel<8,2,0>) is 3,400 cycles of which the arithmetic is a
// few hundred — the rest are LDS round trips (operand fetch, cross-lane merges of the three blocks of a row, the part vector,
// the element-wise phases) between four barriers (profiles/r02_rpl_phases.txt).  Here
//   * a 16-lane DPP row is one knot: lane i < 14 of the row owns ROW i of the knot's block row — left, diagonal and right block,
//     42 floats, in registers — and entry i of the knot's p, r, lambda (registers).  Four knots per wavefront; a wavefront holds
//     RHO slots of four knots: knot (j NW + w) 4 + q for slot j, wave w, DPP row q;
//   * block-row x vector is 42 `v_fmac_f32_dpp ... row_newbcast:c`: the operand x_k[c] IS a register of lane c of the same DPP row,
//     read through the DPP source modifier — no operand fetch, no cross-lane merge, no part vectors;
//   * the only LDS traffic: every lane publishes its own entry of the vector after an update (one ds_write_b32 per slot) and
//     fetches the same entry of the two neighbouring knots before a pass (two ds_read_b32 per slot; the neighbour may live in
//     another wavefront), plus the NW wave partials of an inner product;
//   * the vector updates are three FMAs per slot, in registers.
// Same PCG, same exit rule, same outputs as the other kernels; a row's 42 products are summed block by block (three chains),
// so iterates agree with the other kernels to fp32 round-off of the sums (tested against the oracle band).
// Full block rows are kept (both L_k and L_{k+1}^T): 2 x 42 floats per knot row = the register file holds N <= 64 (SS).
#pragma once
#include <type_traits>

#include "pcg_kernels.hip.h"
#include "pcg_f64.hip.h"

namespace mpcg {

// acc += x[lane c of this DPP row] * m.   (asm: hipcc does not fold the row-broadcast DPP move into the FMA; written out it is
// one instruction instead of two.  A DPP source written by the previous VALU instruction needs two wait states: every product
// sequence below starts behind an s_nop 1.)
template <int C>
__device__ __forceinline__ void fmac_bc(float& acc, float x, float m) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(C));
}
// double precision (linsys_t = double): row_newbcast is the one DPP control gfx90a+ has for 64-bit operands — exactly the one needed
template <int C>
__device__ __forceinline__ void fmac_bc(double& acc, double x, double m) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(C));
}
// sum over the 64 lanes, fixed order, result in every lane's return value of lane 0's wave-uniform use
__device__ __forceinline__ float rpl_wave_fold(float part) {
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
}
// double: 64-bit operations have no row shifts in DPP (row_newbcast only) — but a shift only MOVES data, so the two halves travel as 32-bit
// DPP moves and the add is a plain v_add_f64: four steps inside the 16-lane rows (lanes shifted in from outside a row read +0.0), then the four
// row sums through v_readlane.  (Until round 5 this was six __shfl_down round trips through the LDS crossbar: ~600 clocks per fold, twice per
// iteration.)  Fixed order, the float version's.
template <int CTRL>
__device__ __forceinline__ double rpl_dpp_shift_add(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double rpl_readlane(double v, int l) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double rpl_wave_fold(double part) {
    part = rpl_dpp_shift_add<0x108>(part);             // row_shl:8
    part = rpl_dpp_shift_add<0x104>(part);
    part = rpl_dpp_shift_add<0x102>(part);
    part = rpl_dpp_shift_add<0x101>(part);
    return ((part + rpl_readlane(part, 16)) + rpl_readlane(part, 32)) + rpl_readlane(part, 48);
}
// column c of all three blocks of a row: three independent chains, interleaved (a v_fmac's result is ready after ~2 issue slots)
template <int C, typename T>
__device__ __forceinline__ void col3_bc_t(T& aL, T& aD, T& aR, T xm, T x, T xq, const T* m) {
    fmac_bc<C>(aD, x, m[14 + C]);
    fmac_bc<C>(aL, xm, m[C]);
    fmac_bc<C>(aR, xq, m[28 + C]);
}
template <int I>
struct SFor14 {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, I>{});
        SFor14<I + 1>::run(f);
    }
};
template <>
struct SFor14<14> {
    template <class F>
    static __device__ __forceinline__ void run(F&&) {}
};

// LDS: six vectors [N + 2][14] (a zero knot either side): p and r double-buffered, r~ and upsilon | 2 NW wave partials   (elements)
__host__ __device__ constexpr size_t pcg_rpl_lds_floats(int N, int NW) { return 6 * r4((size_t)(N + 2) * NS) + r4(2 * (size_t)NW); }

// the arguments of one trajectory, in the kernel's element type
template <typename T>
struct RplTraj {
    const T* S; const T* Pinv; const T* gamma; T* lambda; T* r_out; T* p_out;     // r_out / p_out may be null
    uint32_t* iters; uint8_t* max_iter_exit;
    int N; int max_iter; T exit_tol;
};

template <typename T, int NW, int RHO, bool PC3>
__device__ __forceinline__ void pcg_rpl_body(const RplTraj<T>& a) {
    typedef T real;
    constexpr int NTHR = NW * 64, PW = PC3 ? 42 : 14;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    real* lds = reinterpret_cast<real*>(lds_raw);
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int VS = (int)r4((size_t)(N + 2) * NS);
    real* xp0 = lds;                                   // knot k at (k + 1) * NS.  p of even / odd iterations
    real* xp1 = lds + VS;
    real* xr0 = lds + 2 * VS;                          // r likewise
    real* xr1 = lds + 3 * VS;
    real* xt = lds + 4 * VS;                           // r~ of the last preconditioner pass
    real* xu = lds + 5 * VS;                           // upsilon of the last S pass
    real* red_v = lds + 6 * VS;
    real* red_e = red_v + NW;
    const real* Sg = a.S;
    const real* Pg = a.Pinv;
    const real* gam = a.gamma;
    real* lam_g = a.lambda;

    const int q = lane >> 4, i = lane & 15;
    const int ii = i < NS ? i : NS - 1;                // (lanes 14, 15 of a row shadow row 13 with zero matrices)
    int kk[RHO];                                       // knot of slot j (clamped), LDS offset of its own entry
    bool act[RHO];
    real Sm[RHO][42], Pm[RHO][PW], r[RHO], p[RHO], lam[RHO];
#pragma unroll
    for (int j = 0; j < RHO; ++j) {
        const int k = (j * NW + w) * 4 + q;
        act[j] = k < N && i < NS;
        kk[j] = k < N ? k : N - 1;
        const real* sb = Sg + (size_t)kk[j] * ROWF + ii;
        const real* pb = Pg + (size_t)kk[j] * ROWF + ii;
#pragma unroll
        for (int c = 0; c < NS; ++c) {                 // element (i, c) of block s: s * 196 + 14 c + i; blocks (0, left), (N-1, right) are never read
            Sm[j][c] = act[j] && k > 0 ? sb[NS * c] : real(0);
            Sm[j][14 + c] = act[j] ? sb[196 + NS * c] : real(0);
            Sm[j][28 + c] = act[j] && k < N - 1 ? sb[392 + NS * c] : real(0);
            if constexpr (PC3) {
                Pm[j][c] = act[j] && k > 0 ? pb[NS * c] : real(0);
                Pm[j][14 + c] = act[j] ? pb[196 + NS * c] : real(0);
                Pm[j][28 + c] = act[j] && k < N - 1 ? pb[392 + NS * c] : real(0);
            } else {
                Pm[j][c] = act[j] ? pb[196 + NS * c] : real(0);
            }
        }
        const real l0 = act[j] ? lam_g[kk[j] * NS + ii] : real(0);
        lam[j] = l0;
        p[j] = l0;                                     // operand of the set-up product
        r[j] = act[j] ? gam[kk[j] * NS + ii] : real(0);
    }
    for (int e = tid; e < 6 * VS; e += NTHR) lds[e] = real(0);
    lds_barrier();

    // every lane publishes its own entry of x
    auto publish = [&](real* buf, const real (&x)[RHO]) {
#pragma unroll
        for (int j = 0; j < RHO; ++j)
            if (act[j]) buf[(kk[j] + 1) * NS + ii] = x[j];
    };
    // y_j = (block row of slot j) . x ;  returns this lane's share of x . y.   Three-block rows (S, symmetric-stair Pinv):
    // The neighbouring knots' entries of the operand are REBUILT by the reader from what their owners published before the last
    // barrier: x_nb = fma(cb, B[nb], A[nb]) — the very operation (and bits) with which the owner updated its register copy.
    // That is what lets an iteration get by with two barriers (one per inner product) instead of four.
    auto pass3 = [&](const real (&M)[RHO][42], const real* A, const real* B, real cb, const real (&x)[RHO], real (&y)[RHO]) -> real {
        real xm[RHO], xq[RHO];
#pragma unroll
        for (int j = 0; j < RHO; ++j) {                // entry i of the neighbouring knots (zero padding outside the horizon)
            const int lo = kk[j] * NS + ii, hi = (kk[j] + 2) * NS + ii;
            xm[j] = fma_t(cb, B[lo], A[lo]);
            xq[j] = fma_t(cb, B[hi], A[hi]);
        }
        real aL[RHO], aD[RHO], aR[RHO];
#pragma unroll
        for (int j = 0; j < RHO; ++j) { aL[j] = real(0); aD[j] = real(0); aR[j] = real(0); }
        real xo[RHO];
#pragma unroll
        for (int j = 0; j < RHO; ++j) { xo[j] = x[j]; asm volatile("s_nop 1" : "+v"(xo[j]), "+v"(xm[j]), "+v"(xq[j])); }
        SFor14<0>::run([&](auto cc) {                  // 3 RHO independent chains, column by column
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < RHO; ++j) col3_bc_t<C>(aL[j], aD[j], aR[j], xm[j], xo[j], xq[j], M[j]);
        });
        real part = real(0);
#pragma unroll
        for (int j = 0; j < RHO; ++j) { y[j] = (aD[j] + aL[j]) + aR[j]; part = fma_t(y[j], x[j], part); }
        return part;
    };
    // block-Jacobi Pinv: the diagonal block only
    auto pass1 = [&](const real (&M)[RHO][14], const real (&x)[RHO], real (&y)[RHO]) -> real {
        real xo[RHO], a0[RHO], a1[RHO];
#pragma unroll
        for (int j = 0; j < RHO; ++j) { xo[j] = x[j]; a0[j] = real(0); a1[j] = real(0); asm volatile("s_nop 1" : "+v"(xo[j])); }
        SFor14<0>::run([&](auto cc) {                  // two chains per row (even / odd columns): a lone chain of 14 waits on itself
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < RHO; ++j) fmac_bc<C>((C & 1) ? a1[j] : a0[j], xo[j], M[j][C]);
        });
#pragma unroll
        for (int j = 0; j < RHO; ++j) y[j] = a0[j] + a1[j];
        real part = real(0);
#pragma unroll
        for (int j = 0; j < RHO; ++j) part = fma_t(y[j], x[j], part);
        return part;
    };
    auto passP = [&](const real* A, const real* B, real cb, const real (&x)[RHO], real (&y)[RHO]) -> real {
        if constexpr (PC3) return pass3(Pm, A, B, cb, x, y);
        else return pass1(Pm, x, y);
    };
    auto all_sum = [&](const real* red) -> real {   // the NW wave partials, the same fixed (pairwise) order in every thread
        real t[NW];
#pragma unroll
        for (int c = 0; c < NW; ++c) t[c] = red[c];
#pragma unroll
        for (int h = NW / 2; h >= 1; h /= 2)
#pragma unroll
            for (int c = 0; c < h; ++c) t[c] = t[2 * c] + t[2 * c + 1];
        return t[0];
    };
    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
    real y[RHO];
    publish(xt, p);                                    // (lambda0 as the operand of the set-up product)
    lds_barrier();
    (void)pass3(Sm, xt, xp0, real(0), p, y);
#pragma unroll
    for (int j = 0; j < RHO; ++j) r[j] -= y[j];
    publish(xr0, r);                                   // r_0: "r before the update" of iteration 0
    lds_barrier();                                     // (also: every read of lambda0 in xt is done)
    {
        const real part = rpl_wave_fold(passP(xr0, xu, real(0), r, y));      // xu is still all zero: neighbours' r_0 = xr0 + 0 * 0
        if (lane == 0) red_e[w] = part;
    }
#pragma unroll
    for (int j = 0; j < RHO; ++j) p[j] = y[j];
    publish(xt, y);                                    // r~_0; p_0 = r~_0 + 0 * p_(-1), p_(-1) = the zeros of xp0
    lds_barrier();
    real eta = all_sum(red_e);

    // Iteration it: S pass on p_it, whose neighbour entries are xt + beta * xp[it & 1] (r~ and the previous p as published);
    // preconditioner pass on r_(it+1), neighbour entries xr[it & 1] - alpha * xu.  Every lane publishes its own upsilon / r~ entry
    // with the wave partial before the one barrier of the pass, and its updated r / p entry into the other buffer of the pair.
    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    real beta = real(0);
    if (fabs_t(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
            const bool prof_on = blockIdx.x == 0 && it == 20;      // (tools/_prof/rpl_phases.py)
#endif
            MPCG_STAMP(0);
            real* xp_old = (it & 1) ? xp1 : xp0;
            real* xp_new = (it & 1) ? xp0 : xp1;
            real* xr_old = (it & 1) ? xr1 : xr0;
            real* xr_new = (it & 1) ? xr0 : xr1;
            {   // upsilon = S p ; v = p . upsilon
                const real pp = pass3(Sm, xt, xp_old, beta, p, y);
                MPCG_STAMP(1);
                const real part = rpl_wave_fold(pp);
                if (lane == 0) red_v[w] = part;
            }
            MPCG_STAMP(2);
            if constexpr (PC3) publish(xu, y);
            publish(xp_new, p);                        // p_it, for the neighbours' rebuild in iteration it + 1
            lds_barrier();
            MPCG_STAMP(3);
            const real alpha = eta / all_sum(red_v);
#pragma unroll
            for (int j = 0; j < RHO; ++j) {
                lam[j] = fma_t(alpha, p[j], lam[j]);
                r[j] = fma_t(-alpha, y[j], r[j]);
            }
            if constexpr (PC3) publish(xr_new, r);     // r_(it+1), "r before the update" of the next iteration
            MPCG_STAMP(4);
            {   // r~ = Pinv r ; eta' = r . r~
                const real pp = passP(xr_old, xu, -alpha, r, y);
                MPCG_STAMP(5);
                const real part = rpl_wave_fold(pp);
                if (lane == 0) red_e[w] = part;
            }
            publish(xt, y);
            MPCG_STAMP(6);
            lds_barrier();
            MPCG_STAMP(7);
            const real eta_new = all_sum(red_e);
            iters = (uint32_t)(it + 1);
            if (fabs_t(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
            beta = eta_new / eta;
#pragma unroll
            for (int j = 0; j < RHO; ++j) p[j] = fma_t(beta, p[j], y[j]);
            eta = eta_new;
            MPCG_STAMP(8);
        }
    }

#pragma unroll
    for (int j = 0; j < RHO; ++j) {
        if (act[j]) {
            const size_t e = (size_t)kk[j] * NS + ii;
            lam_g[e] = lam[j];
            if (a.r_out) a.r_out[e] = r[j];
            if (a.p_out) a.p_out[e] = p[j];
        }
    }
    if (tid == 0) {
        *a.iters = iters;
        *a.max_iter_exit = (uint8_t)max_iter_exit;
    }
}

// register estimate -> minimum waves per SIMD the kernel is compiled for (4 when a wave needs <= 128 registers, else 2)
template <typename T, int RHO, bool PC3>
constexpr int rpl_min_waves() { return (int)(sizeof(T) / 4) * RHO * (PC3 ? 84 : 56) + 40 <= 128 ? 4 : 2; }

template <int NW, int RHO, bool PC3>
__global__ __launch_bounds__(NW * 64, (rpl_min_waves<float, RHO, PC3>())) void pcg_rpl_kernel(PcgArgs a) {
    const int b = sched_pick(a.order, (int)blockIdx.x, a.order_tag);     // (dispatch order: longest-expected first, sched_order_kernel)
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    const size_t mstride = (size_t)a.N * ROWF, vstride = (size_t)a.N * NS;
    RplTraj<float> t;
    t.S = static_cast<const float*>(a.S) + (size_t)b * mstride;
    t.Pinv = static_cast<const float*>(a.Pinv) + (size_t)b * mstride;
    t.gamma = a.gamma + (size_t)b * vstride;
    t.lambda = a.lambda + (size_t)b * vstride;
    t.r_out = a.r_out ? a.r_out + (size_t)b * vstride : nullptr;
    t.p_out = a.p_out ? a.p_out + (size_t)b * vstride : nullptr;
    t.iters = a.iters + b; t.max_iter_exit = a.max_iter_exit + b;
    t.N = a.N; t.max_iter = a.max_iter; t.exit_tol = a.exit_tol;
    pcg_rpl_body<float, NW, RHO, PC3>(t);
}

// linsys_t = double (USE_DOUBLES of the reference): the same kernel in double precision; one slot per wavefront, N <= 32
template <int NW, int RHO, bool PC3>
__global__ __launch_bounds__(NW * 64, (rpl_min_waves<double, RHO, PC3>())) void pcg_rpl_kernel_f64(PcgArgs64 a) {
    const int b = blockIdx.x;
    const size_t mstride = (size_t)a.N * ROWF, vstride = (size_t)a.N * NS;
    RplTraj<double> t;
    t.S = a.S + (size_t)b * mstride;
    t.Pinv = a.Pinv + (size_t)b * mstride;
    t.gamma = a.gamma + (size_t)b * vstride;
    t.lambda = a.lambda + (size_t)b * vstride;
    t.r_out = a.r_out ? a.r_out + (size_t)b * vstride : nullptr;
    t.p_out = a.p_out ? a.p_out + (size_t)b * vstride : nullptr;
    t.iters = a.iters + b; t.max_iter_exit = a.max_iter_exit + b;
    t.N = a.N; t.max_iter = a.max_iter; t.exit_tol = a.exit_tol;
    pcg_rpl_body<double, NW, RHO, PC3>(t);
}



}  // namespace mpcg
