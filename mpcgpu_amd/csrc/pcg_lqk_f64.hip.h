// pcg_lqk_f64.hip.h — "lane QUAD per knot": the lane-pair kernel (pcg_lpk.hip.h) re-derived for linsys_t = double (USE_DOUBLES,
// include/common/settings.cuh:41-49).  64 knots per CU, everything register-resident, two barriers per PCG iteration.
//
// Why: full block rows of S and Pinv in double fill a CU's register file at 32 knots (pcg_rpl_kernel_f64); beyond that the row-per-lane kernel
// needs ceil(N / 32) CUs per trajectory and two cluster-wide hand-offs per iteration (pcg_rpl_cluster_f64.hip.h: N = 64 at 44 M it/s).  The
// lower block triangle (D_k, L_k) is half the data: in the lane-pair mapping a double costs what a float PAIR costs there, so a knot takes
// FOUR lanes and a CU holds 64 knots — with the float kernel's instruction stream, one v_fma_f64 where it issues one v_pk_fma_f32.
//
// Mapping.  Lane (h, g) of knot k's quad (lane = 4 k_local + 2 h + g), per matrix: h = the column half of pcg_lpk_kernel (lane h holds columns
// 0..6 / 8..13, 7 of D_k and L_k), g = WHICH ROW of each row pair: slot s of lane (h, g) is row 2 q(h, s) + g, q(0, s) = s, q(1, s) =
// (4, 5, 6, 3, 0, 1, 2)[s] — a float2 register of the float kernel is the same register pair in lanes g = 0 and g = 1 here.  So everything
// element-wise carries over unchanged (the operand rebuild, the direct products acc[row] += M[row][c] x[c], the column-half merge with the
// h-partner, the vector updates, the LDS layout: entry 2 q + g of a knot is double g of "row pair" q), and what the float kernel did INSIDE a
// float2 becomes a quad_perm move between the g lanes:
//   * x[c] for the direct products: column c = (own slot c >> 1, element c & 1) lives in the lane with g = c & 1 — broadcast to both with
//     quad_perm [0,0,2,2] / [1,1,3,3] right before use (two v_mov_b32_dpp: 64-bit DPP exists for row_newbcast only);
//   * the transposed product z[c] = sum_rows L[row][c] x[row] sums this lane's seven rows; the other seven come from the g-partner (one add).
// 147 v_fma_f64 per lane and pass + 28 broadcasts + 10 partner moves.  Four S wavefronts and four Pinv wavefronts of 16 knots each.
// Reads only the left + diagonal block columns (include/mpcg.h, BLOCK SYMMETRY): launched when the handle's latch says symmetric.
// Same PCG, same exit rule, same outputs as every other kernel; results agree with the oracle's double instantiation to the round-off of the
// different summation order (tests: 1e-9 of the iterate after fixed iteration counts).
#pragma once
#include "pcg_f64.hip.h"
#include "pcg_rpl.hip.h"

namespace mpcg {

// LDS layout of one vector (doubles): PAIR-MAJOR as in LpkLds — V[q][slot] = entries (2q, 2q+1) of knot slot - 1, q = 0..6, one zero knot in
// front, one behind.  Bank check for ds_read_b64 (two groups of 32 lanes over 64 banks): a group is 8 knots x (h, g): the g lanes of 8
// consecutive knots read 16 consecutive doubles = 32 banks, the h = 1 lanes read row pair q + 4 = 8 KN doubles further = 16 KN banks:
// KN = 2 (mod 4) puts them on the other 32 banks.
#ifndef LQK_NPARK
#define LQK_NPARK 5      // matrix values per lane parked in LDS (the last column of D_k, rows 7 - NPARK .. 6 of the lane's slots): no scratch
#endif
template <int NWR> struct LqkLds {
    static constexpr int NMAX = 32 * NWR, NW = 4 * NWR;
    static constexpr int KN = NMAX + 2;
    static_assert(KN % 4 == 2, "row pairs q and q + 4 must sit 32 banks apart");
    static constexpr int VS = 7 * KN * 2;                      // doubles per vector
    static constexpr int P0 = 0, R0 = VS, US = 2 * VS, ZS = 3 * VS, RT = 4 * VS, ZP = 5 * VS, LAM = 6 * VS, RED = 7 * VS, MX = RED + NW, NPARK = LQK_NPARK,
                         TOTAL = MX + NPARK * NW * 64;
    __host__ __device__ static constexpr int at(int k, int i) { return 2 * ((i >> 1) * KN + k + 1) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lqk_lds_doubles(int NW) { return NW == 4 ? (size_t)LqkLds<1>::TOTAL : (size_t)LqkLds<2>::TOTAL; }

// the value of `v` in another lane of the quad (64-bit: two 32-bit DPP moves)
template <int QP>
__device__ __forceinline__ double lqk_quad(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, QP, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), QP, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
constexpr int LQK_QP_G = 0xB1;      // quad_perm [1,0,3,2]: the g-partner (other row of the pair)
constexpr int LQK_QP_H = 0x4E;      // quad_perm [2,3,0,1]: the h-partner (other column half)
constexpr int LQK_QP_B0 = 0xA0;     // quad_perm [0,0,2,2]: the g = 0 lane's value to both g lanes
constexpr int LQK_QP_B1 = 0xF5;     // quad_perm [1,1,3,3]: the g = 1 lane's
constexpr int LQK_QP_B6 = 0xF0;     // quad_perm [0,0,3,3]: element h of the pair (h = 0: the g = 0 lane's, h = 1: the g = 1 lane's)
__device__ __forceinline__ double lqk_uniform(double v) { return rpl_readlane(v, 0); }
typedef __attribute__((address_space(3))) const volatile double lds_cv_d;
__device__ __forceinline__ double lqk_ld(const double* p) { return *(lds_cv_d*)(p); }

// ---- the matrix registers of lane (h, g): seven columns (h = 0: 0..6; h = 1: 8..13, 7) of D_k and L_k, rows 2 q(h, s) + g.  Element (r, c) of a
// block: byte 8 (14 c + r) (column-major).  Column-major issue order; knots outside the horizon / the absent L_0 read zeros (out-of-bounds offset).
__device__ __forceinline__ void lqk_load_blocks(rsrc_t M, int k, int h, int g, bool okD, bool okL, double (&Md)[7][7], double (&Ml)[7][7]) {
    constexpr uint32_t CB = (uint32_t)NS * 8u, BLKB = (uint32_t)(NS * NS) * 8u;
    const uint32_t rowb = (uint32_t)k * ((uint32_t)ROWF * 8u);
    uint32_t bL[7], bD[7], bL6[7], bD6[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int q1 = s < 3 ? s + 4 : (s == 3 ? 3 : s - 4);
        const uint32_t bs = rowb + 8u * (uint32_t)(2 * (h ? q1 : s) + g);
        bL[s] = okL ? bs + CB * 8u * (uint32_t)h : OOB_OFF;
        bD[s] = okD ? bs + CB * 8u * (uint32_t)h + BLKB : OOB_OFF;
        bL6[s] = okL ? bs + CB * (uint32_t)(6 + h) : OOB_OFF;
        bD6[s] = okD ? bs + CB * (uint32_t)(6 + h) + BLKB : OOB_OFF;
    }
    auto ld = [&](uint32_t off) -> double {
#ifdef LQK_ABLATE_LOAD                                    // (timing experiment only: no matrix traffic, wrong results)
        return (double)off * 1e-30;
#endif
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(M, (int)off, 0, 0);
        return __builtin_bit_cast(double, v);
    };
#pragma unroll
    for (int j = 0; j < 7; ++j) {
#pragma unroll
        for (int s = 0; s < 7; ++s) Ml[s][j] = ld(j < 6 ? bL[s] + CB * (uint32_t)j : bL6[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
#pragma unroll
        for (int s = 0; s < 7; ++s) Md[s][j] = ld(j < 6 ? bD[s] + CB * (uint32_t)j : bD6[s]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lqk_f64_kernel(PcgArgs64 a) {
    typedef LqkLds<NWR> L;
    typedef double real;
    constexpr int NW = 4 * NWR, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    real* lds = lds_d;
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    real* red_v = lds + L::RED;
    real* red_e = red_v + NW / 2;

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const real* gam = a.gamma + (size_t)b * vstride;
    real* lam_g = a.lambda + (size_t)b * vstride;
    const real* lam0_g = a.lam0 ? a.lam0 + (size_t)b * vstride : lam_g;

    // ---- role of this wave; knot, column half and row of this lane ----
    const bool isP = w >= 2 * NWR;                         // wave-uniform
    const int wl = w - (isP ? 2 * NWR : 0);                // wave of its matrix: 0 .. 2 NWR - 1
    const int li = 64 * wl + lane;
    const int k = li >> 2, h = (li >> 1) & 1, g = li & 1;
    const bool p3 = a.pcols == 3;
    const bool hasL = !isP || p3;                          // wave-uniform: block-Jacobi has no off-diagonal Pinv blocks
    const bool valid = k < N;
    // double indices inside a vector (K2 = doubles between consecutive row pairs):
    //   register slots 0..2 -> pairs 4h + s, element g: bA + K2 s | slot 3 -> pair 3: b0 + 3 K2;  knot k - 1: subtract 2, knot k + 1: add 2
    constexpr int KN = L::KN, K2 = 2 * KN;
    const int b0 = 2 * (k + 1) + g;
    const int bA = b0 + (h ? 4 * K2 : 0);

    real Md[7][7], Ml[7][7];                               // [slot][j]
    {
        const rsrc_t M = make_rsrc(static_cast<const char*>(static_cast<const void*>(isP ? a.Pinv : a.S)) + (size_t)b * mstride * 8, (uint32_t)(mstride * 8));
        lqk_load_blocks(M, k, h, g, valid, valid && k > 0 && hasL, Md, Ml);
    }
    // park the values the pass uses last (the diagonal block's seventh column, slots 7 - NPARK .. 6) in LDS; they are fetched back inside the pass
    // (the register file holds 196 matrix registers + the working set of a half-iteration only just: left to the compiler the overflow goes to
    // SCRATCH, whose reloads cost global-memory latency in every pass — pcg_lpk_kernel)
    real* const park = lds + L::MX + tid;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
#pragma unroll
    for (int i = 0; i < L::NPARK; ++i) park[i * NTHR] = Md[7 - L::NPARK + i][6];

    // ---- stage vectors: P0 <- lambda0 (operand of the setup product), lambda <- lambda0, R0 <- gamma, everything else (pads included) <- 0 ----
    for (int e = tid; e < L::RED; e += NTHR) lds[e] = real(0);
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        const real l0 = lam0_g[e];
        lds[L::P0 + L::at(kk, i)] = l0;
        lds[L::LAM + L::at(kk, i)] = l0;
        lds[L::R0 + L::at(kk, i)] = gam[e];
    }
    lds_barrier();

    // sum of the NW/2 wave partials of one inner product (the waves of one matrix), same order in every thread: deterministic
    auto sum_red = [&](const real* red) -> real {
        if constexpr (NW == 8) return ((red[0] + red[1]) + red[2]) + red[3];
        else return red[0] + red[1];
    };
    struct Own { real v[4]; };
    // own entries (register slots 0..3, this lane's row of each pair) of knot k + dk of the vector at double offset X
    auto load_own = [&](int X, int dk) -> Own {
        const real* x = lds + X + 2 * dk;
        Own o;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.v[s] = x[bA + K2 * s];
        o.v[3] = x[b0 + 3 * K2];
        return o;
    };
    // (no `k < N` predicate: a lane beyond the horizon holds all-zero blocks — its products, its z and its copies of the vectors are exact
    //  zeros, and its knot slots exist — so it may write them)
    auto store_own = [&](int X, const Own& o) {
        real* x = lds + X;
#pragma unroll
        for (int s = 0; s < 3; ++s) x[bA + K2 * s] = o.v[s];
        x[b0 + 3 * K2] = o.v[3];
    };
    // The operand loads of a half-iteration, requested as soon as the barrier in front of it is passed — before the scalar of the update is
    // worked out.  Own entries (slots 0..3) of knot k and of knot k-1 of the two published vectors the operand is formed from: 16 ds_read_b64.
    struct Fetch { real t[4], z[4], gt[4], gz[4]; };
    auto fetch = [&](int T, int Z) -> Fetch {
        const real* xt = lds + T;
        const real* xz = lds + Z + 2;
        Fetch f;
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.t[s] = lqk_ld(xt + bA + K2 * s); f.z[s] = lqk_ld(xz + bA + K2 * s); }
        f.t[3] = lqk_ld(xt + b0 + 3 * K2); f.z[3] = lqk_ld(xz + b0 + 3 * K2);
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.gt[s] = lqk_ld(xt - 2 + bA + K2 * s); f.gz[s] = lqk_ld(xz - 2 + bA + K2 * s); }
        f.gt[3] = lqk_ld(xt - 2 + b0 + 3 * K2); f.gz[3] = lqk_ld(xz - 2 + b0 + 3 * K2);
        return f;
    };
    struct Vec { Own k, m; };                                  // a lane's copy of a vector: its own entries of knot k and of knot k-1
    // entry of column j (of this lane's seven) of a vector whose own slots are v: slot j >> 1, element j & 1 — from the g lane that holds it
    auto col = [&](const real (&v)[4], auto jt) -> real {
        constexpr int J = decltype(jt)::value;
        if constexpr (J == 6) return lqk_quad<LQK_QP_B6>(v[3]);                                       // entry 6 + h: pair 3, element h
        else if constexpr ((J & 1) == 0) return lqk_quad<LQK_QP_B0>(v[J >> 1]);
        else return lqk_quad<LQK_QP_B1>(v[J >> 1]);
    };

    // One half-iteration of this wave's matrix (pcg_lpk_kernel::half).  `old` = the lane's register copy of the vector being updated.
    //   MODE 0: the operand is `old` as it stands (setup product S lambda0);
    //   MODE 1: operand = old - c (T + Z<<1)          (Pinv half: r_new, c = alpha; setup: c = 1)
    //   MODE 2: operand = (T + Z<<1) + c old          (S half: p_new, c = beta; first iteration: c = 0)
    auto half = [&](auto mode_tag, const Fetch& f, const Vec& old, real c, int TOUT, int ZOUT, real* red) -> Vec {
        constexpr int MODE = decltype(mode_tag)::value;
        real xk[7];                                              // x_k at this lane's row of all seven pairs (slots 4..6: the h-partner's own)
        Own om;                                                  // knot k-1, own entries
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { xk[s] = old.k.v[s]; om.v[s] = old.m.v[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) { const real u = f.t[s] + f.z[s]; xk[s] = MODE == 1 ? old.k.v[s] - c * u : u + c * old.k.v[s]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) { const real u = f.gt[s] + f.gz[s]; om.v[s] = MODE == 1 ? old.m.v[s] - c * u : u + c * old.m.v[s]; }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) xk[4 + s] = lqk_quad<LQK_QP_H>(xk[s]);
        Own me;
#pragma unroll
        for (int s = 0; s < 4; ++s) me.v[s] = xk[s];
        real acc[7];
        real cterm = real(0);
        const real xk6 = col(me.v, std::integral_constant<int, 6>{});
        if (hasL) {
            // transposed: z[j] = sum over rows of L[row][column j] x_k[row]; this lane's seven rows, the other seven from the g-partner
            real z[7];
            {
                real t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = Ml[0][j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = fma(Ml[s][j], xk[s], t[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = t[j] + lqk_quad<LQK_QP_G>(t[j]);
            }
            {
                real t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) t[j] = Ml[0][4 + j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] = fma(Ml[s][4 + j], xk[s], t[j]);
#pragma unroll
                for (int j = 0; j < 3; ++j) z[4 + j] = t[j] + lqk_quad<LQK_QP_G>(t[j]);
            }
            {                                                    // z belongs to knot k-1's vector: column j = (slot j >> 1, element j & 1) — this lane stores element g
                real* zo = lds + ZOUT;
#pragma unroll
                for (int s = 0; s < 3; ++s) zo[bA + K2 * s] = g ? z[2 * s + 1] : z[2 * s];
                if (g == h) zo[b0 + 3 * K2] = z[6];              // entry 6 + h = pair 3, element h
            }
            // second copy of the coupling term of the inner product, x_{k-1}^T (L_k^T x_k): this lane's own columns of parity g (+ column 6 in the lane g == h)
            real ct = (g ? z[1] : z[0]) * om.v[0];
            ct = fma(g ? z[3] : z[2], om.v[1], ct);
            ct = fma(g ? z[5] : z[4], om.v[2], ct);
            cterm = g == h ? fma(z[6], om.v[3], ct) : ct;
            // direct, off-diagonal columns: acc = L[:, c_j] x_{k-1}[c_j]
            {
                const real x0 = col(om.v, std::integral_constant<int, 0>{});
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = Ml[s][0] * x0;
            }
            SFor14<8>::run([&](auto jt) {                       // j = 1 .. 6
                constexpr int J = decltype(jt)::value - 7;
                const real xs = col(om.v, std::integral_constant<int, J>{});
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = fma(Ml[s][J], xs, acc[s]);
            });
            {
                const real x0 = col(me.v, std::integral_constant<int, 0>{});
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = fma(Md[s][0], x0, acc[s]);
            }
        } else {
            const real x0 = col(me.v, std::integral_constant<int, 0>{});
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Md[s][0] * x0;
        }
        // (the parked values are requested here, volatile = in program order, and consumed by the last FMAs of the pass)
        real pk_[L::NPARK];
#pragma unroll
        for (int i = 0; i < L::NPARK; ++i) pk_[i] = lqk_ld(park + i * NTHR);
        // direct, diagonal columns
        SFor14<9>::run([&](auto jt) {                           // j = 1 .. 5
            constexpr int J = decltype(jt)::value - 8;
            const real xs = col(me.v, std::integral_constant<int, J>{});
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = fma(Md[s][J], xs, acc[s]);
        });
#pragma unroll
        for (int s = 0; s < 7 - L::NPARK; ++s) acc[s] = fma(Md[s][6], xk6, acc[s]);
#pragma unroll
        for (int i = 0; i < L::NPARK; ++i) acc[7 - L::NPARK + i] = fma(pk_[i], xk6, acc[7 - L::NPARK + i]);
        // merge the two column halves: own slot s + the h-partner's slot (4, 5, 6, 3)[s]
        Own o;
#pragma unroll
        for (int s = 0; s < 4; ++s) o.v[s] = acc[s] + lqk_quad<LQK_QP_H>(acc[s < 3 ? s + 4 : 3]);
        store_own(TOUT, o);
        // inner product share: x_k . (D x_k + L x_{k-1}) over this lane's OWN rows (lane h = 1's slot 3 duplicates h = 0's: weight 0) + the coupling copy
        real d0 = o.v[0] * me.v[0];
        d0 = fma(o.v[1], me.v[1], d0);
        d0 = fma(o.v[2], me.v[2], d0);
        const real d3 = o.v[3] * me.v[3];
        const real part = rpl_wave_fold((d0 + (h ? real(0) : d3)) + cterm);
        if (lane == 0) red[wl] = part;
        return Vec{me, om};
    };

    // The S waves and the Pinv waves run the same barrier sequence through two SEPARATE code paths (the role is wave-uniform).
    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    real beta = real(0);                                       // scalar of the NEXT p update (the S half applies it)
    bool p_pending = true;                                     // that update has not been applied to p (write-back of d_p does it)
    auto run_role = [&](auto role_tag) {
        constexpr bool P = decltype(role_tag)::value;
        // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; eta = r . r~   (p = r~ is formed by the first S half: beta = 0) ----
        Fetch f;
        Vec x;                                                   // S waves: p;  Pinv waves: r
        x.k = load_own(P ? L::R0 : L::P0, 0);
        x.m = load_own(P ? L::R0 : L::P0, -1);
        if constexpr (!P) (void)half(std::integral_constant<int, 0>{}, f, x, real(0), L::US, L::ZS, red_v);
        lds_barrier();
        if constexpr (P) {
            f = fetch(L::US, L::ZS);
            x = half(std::integral_constant<int, 1>{}, f, x, real(1), L::RT, L::ZP, red_e);
        }
        lds_barrier();
        if constexpr (!P) f = fetch(L::RT, L::ZP);
        real eta = lqk_uniform(sum_red(red_e));
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): every matrix load has been consumed
        if (fabs(eta) < a.exit_tol) {
            max_iter_exit = 0;
        } else {
            for (int it = 0; it < a.max_iter; ++it) {
                if constexpr (!P) {
                    // p = r~ + beta p ; upsilon = S p ; v = p . upsilon
                    x = half(std::integral_constant<int, 2>{}, f, x, beta, L::US, L::ZS, red_v);
                    lds_barrier();
                    // alpha = eta / v ; lambda += alpha p (own entries) — while the Pinv half runs
                    const Own cur = load_own(L::LAM, 0);
                    const real alpha = lqk_uniform(eta / sum_red(red_v));
                    Own nw;
#pragma unroll
                    for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] + alpha * x.k.v[s];
                    store_own(L::LAM, nw);
                    lds_barrier();
                    f = fetch(L::RT, L::ZP);                    // the next S half's operand loads fly during the scalar chain below
                } else {
                    lds_barrier();
                    // alpha = eta / v ; r -= alpha upsilon ; r~ = Pinv r ; eta' = r . r~
                    f = fetch(L::US, L::ZS);
                    const real alpha = lqk_uniform(eta / sum_red(red_v));
                    x = half(std::integral_constant<int, 1>{}, f, x, alpha, L::RT, L::ZP, red_e);
                    lds_barrier();
                }
                // eta' ; exit test ; beta
                const real eta_new = lqk_uniform(sum_red(red_e));
                iters = (uint32_t)(it + 1);
                if (fabs(eta_new) < a.exit_tol) { max_iter_exit = 0; p_pending = false; break; }     // (the reference leaves p as it is on this exit)
                beta = lqk_uniform(eta_new / eta);
                eta = eta_new;
            }
        }
        // the vector this role carries, for d_p / d_r (the staging buffers are free since the setup)
        store_own(P ? L::R0 : L::P0, x.k);
    };
    if (isP) run_role(std::true_type{}); else run_role(std::false_type{});

    // ---- write back ----
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        lam_g[e] = lds[L::LAM + L::at(kk, i)];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = lds[L::R0 + L::at(kk, i)];
        if (a.p_out) {
            // p of the last completed update; when the loop ended without a tolerance exit that update is still pending: p = r~ + beta p
            real pv = lds[L::P0 + L::at(kk, i)];
            if (p_pending) pv = (lds[L::RT + L::at(kk, i)] + (p3 ? lds[L::ZP + L::at(kk + 1, i)] : real(0))) + beta * pv;
            a.p_out[(size_t)b * vstride + e] = pv;
        }
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
}

}  // namespace mpcg
