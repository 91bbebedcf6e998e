// pcg_lqk_cluster_f64.hip.h — the lane-quad-per-knot kernel in double (pcg_lqk_f64.hip.h) for horizons one CU cannot hold: G = ceil(N / 64)
// workgroups on G CUs of one XCD solve ONE trajectory, each with up to 64 consecutive knots of the lower block triangle of S and Pinv in its
// register file.  The double twin of pcg_lpkc_kernel (pcg_lpk_cluster.hip.h), whose decomposition and hand-off machinery it takes over:
//   * member g owns knots [k0, k1); LDS knot slot 0 = the replica knot k0 - 1 (its operand entries are rebuilt by the lanes of knot k0), slots
//     1..KL = own knots, slot KL + 1 = the right halo knot k1;
//   * a half-iteration needs from outside T[k0 - 1] (the LEFT member's merged rows of its last knot), Z[k1] (the RIGHT member's z = L_k1^T x_k1
//     of its first knot) and the cluster-wide inner product: ONE hand-off per pass carries them as epoch-tagged 16-byte granules
//     {value lo, value hi, tag, 0} published straight from the registers of the lanes that hold them (z mid-pass), polled by a wavefront of
//     the matrix that sits the pass out; it drops the neighbours' entries into the halo slots, folds the partials, ONE barrier ends the exchange;
//   * members pinned to one XCD (verified at start-up; write-through hand-offs otherwise), persistent clusters drawing trajectories from a
//     queue, bounded spins + completion counts + a fix-up launch that warm-starts from the handle's copy of lambda0.
// N = 128: two CUs per trajectory (the clustered row-per-lane kernel needs four, with twice the hand-off partners).  Reads only the left +
// diagonal block columns: launched when the handle's latch says block-symmetric.
#pragma once
#include "pcg_lqk_f64.hip.h"
#include "pcg_rpl_cluster_f64.hip.h"

namespace mpcg {

// Cells of one member, u64 words: two exchange slots of 64 words — [0, 8) four wave partials, [8, 36) T of the last own knot (14 granules, for
// the right neighbour), [36, 64) Z of the first own knot (for the left one) — then {sequence number, trajectory} (leader) and {1, XCC id}.
constexpr int LQKC_WG_WORDS = 144;
constexpr int LQKC_SLOT_V = 0, LQKC_SLOT_E = 64, LQKC_W_T = 8, LQKC_W_Z = 36;
constexpr int LQKC_SLOT_T = 128, LQKC_SLOT_X = 130;
constexpr int LQKC_MAX_G = 8;                       // G x 4 wave partials polled by lanes 0..31 of one wavefront

#ifndef LQKC_NPARK
#define LQKC_NPARK 10     // (the hand-off state — epoch, cell addresses, tables — takes the registers the single-CU kernel still had: more of D_k waits in LDS)
#endif
// LDS (doubles): the lane-quad kernel's seven pair-major vectors | broadcast cell (4) | hand-off tables (3 x 64 ints) | parked matrix values
template <int NWR> struct LqkcLds {
    typedef LqkLds<NWR> B;
    static constexpr int NMAX = B::NMAX, NW = B::NW, KN = B::KN, VS = B::VS;
    static constexpr int P0 = 0, R0 = VS, US = 2 * VS, ZS = 3 * VS, RT = 4 * VS, ZP = 5 * VS, LAM = 6 * VS, BC = 7 * VS, TAB = BC + 4, MX = TAB + 3 * 32,
                         NPARK = LQKC_NPARK, TOTAL = MX + NPARK * NW * 64;
    // parked matrix values: the LAST NPARK entries of D_k in the order the pass consumes them (column-major: entry 7 j + s)
    __host__ __device__ static constexpr bool parked(int s, int j) { return 7 * j + s >= 49 - NPARK; }
    __host__ __device__ static constexpr int pidx(int s, int j) { return 7 * j + s - (49 - NPARK); }
    // double index of entry i of knot slot s inside a vector
    __host__ __device__ static constexpr int at(int s, int i) { return 2 * ((i >> 1) * KN + s) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lqkc_lds_doubles() { return (size_t)LqkcLds<2>::TOTAL; }

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lqkc_f64_kernel(ClusterArgs64 ca) {
    typedef LqkcLds<NWR> L;
    typedef double real;
    constexpr int NW = 4 * NWR, NTHR = NW * 64, NWM = NW / 2;
    static_assert(NWM == 4, "the partial cells hold four wavefronts per matrix");
    const PcgArgs64& a = ca.p;
    typedef const __attribute__((address_space(4))) ClusterArgs64* kargp_t;
    const kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    real* lds = lds_d;
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = ca.G;
    // members of a cluster share an XCD (pcg_lpkc_kernel): workgroup b = 8 j + x holds member j % G of cluster 8 (j / G) + x
    const unsigned nclusters = (unsigned)ca.clusters;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int gm = jx % G;
    const int cl = (jx / G) * 8 + xcd;
    if ((unsigned)cl >= nclusters) return;
    const int k0 = (int)(((long)gm * N) / G), k1 = (int)(((long)(gm + 1) * N) / G);
    const int KL = k1 - k0;                             // own knots (launcher: 1 <= KL <= NMAX)
    real* bc = lds + L::BC;                             // [0] / [3] cluster-wide sum of even / odd hand-offs, [1] sticky timeout flag, [2] trajectory index / same-XCD (int)

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    gu64* my_words = (gu64*)ca.scratch + ((size_t)cl * G + gm) * LQKC_WG_WORDS;
    gu64* cl_words = (gu64*)ca.scratch + (size_t)cl * G * LQKC_WG_WORDS;

    // ---- role of this wave; knot (i = knot inside the member, slot i + 1), column half and row of this lane (pcg_lqk_f64_kernel's mapping) ----
    const bool isP = w >= NWM;
    const int wl = w - (isP ? NWM : 0);
    const int li = 64 * wl + lane;
    const int i = li >> 2, h = (li >> 1) & 1, g = li & 1;
    const bool p3 = a.pcols == 3;
    const bool hasL = !isP || p3;
    const bool valid = i < KL;
    constexpr int KN = L::KN, K2 = 2 * KN;
    const int b0 = 2 * (i + 1) + g;
    const int bA = b0 + (h ? 4 * K2 : 0);

    real Md[7][7], Ml[7][7];
    real* const park = lds + L::MX + tid;
    bool same_xcd = false;
    unsigned epoch = 0, seq = 0;
    bool failed = false;

    struct Own { real v[4]; };
    auto load_own = [&](int X, int dk) -> Own {
        const real* x = lds + X + 2 * dk;
        Own o;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.v[s] = x[bA + K2 * s];
        o.v[3] = x[b0 + 3 * K2];
        return o;
    };
    auto store_own = [&](int X, const Own& o) {
        if (valid) {                                           // (slot KL + 1 is the right halo: lanes beyond the member's knots must not write)
            real* x = lds + X;
#pragma unroll
            for (int s = 0; s < 3; ++s) x[bA + K2 * s] = o.v[s];
            x[b0 + 3 * K2] = o.v[3];
        }
    };
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
    struct Vec { Own k, m; };
    auto col = [&](const real (&v)[4], auto jt) -> real {
        constexpr int J = decltype(jt)::value;
        if constexpr (J == 6) return lqk_quad<LQK_QP_B6>(v[3]);
        else if constexpr ((J & 1) == 0) return lqk_quad<LQK_QP_B0>(v[J >> 1]);
        else return lqk_quad<LQK_QP_B1>(v[J >> 1]);
    };
    // 14 entries of a knot held by its quad -> the 14 granules of the group at word WORD: this lane's v0..v2 are entries 8h + 2s + g, `v3` is
    // entry e3 of the lanes for which pub3 holds (T: own slot 3 = entry 6 + g, lanes h = 0; Z: column 6 = entry 6 + h, lanes g = h)
    auto publish_quad = [&](auto word_tag, auto isz_tag, unsigned ep, real v0, real v1, real v2, real v3) {
        constexpr int WORD = decltype(word_tag)::value;
        constexpr bool ISZ = decltype(isz_tag)::value;
        // (h, g re-read from the hardware lane count: as lane constants living across the pass they were registers this kernel does not have —
        //  spilled, and reloaded from scratch right in front of the hand-off, pcg_lpkc_kernel)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int hh = (ln >> 1) & 1, gg = ln & 1;
        const int e3 = ISZ ? 6 + hh : 6 + gg;
        const bool pub3 = ISZ ? gg == hh : hh == 0;
        const int h = hh, g = gg;
        auto gran = [&](real v) -> f4 {
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
            return f4{__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32)), __builtin_bit_cast(float, ep), 0.f};
        };
        const unsigned o0 = 16u * (unsigned)(8 * h + g);
        if (same_xcd) {
            granule_store16<WORD, true>(my_words, o0, gran(v0));
            granule_store16<WORD + 4, true>(my_words, o0, gran(v1));
            granule_store16<WORD + 8, true>(my_words, o0, gran(v2));
            if (pub3) granule_store16<WORD, true>(my_words, 16u * (unsigned)e3, gran(v3));
        } else {
            granule_store16<WORD, false>(my_words, o0, gran(v0));
            granule_store16<WORD + 4, false>(my_words, o0, gran(v1));
            granule_store16<WORD + 8, false>(my_words, o0, gran(v2));
            if (pub3) granule_store16<WORD, false>(my_words, 16u * (unsigned)e3, gran(v3));
        }
    };

    // One half-iteration of this wave's matrix (pcg_lqk_f64_kernel::half) + the publishing of what the neighbours and the reduction need.
    auto half = [&](auto mode_tag, auto slot, const Fetch& f, const Vec& old, real c, int TOUT, int ZOUT) -> Vec {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int base = decltype(slot)::value;
        const unsigned ep = epoch + 1;                      // tag of the hand-off that follows this pass
        real xk[7];
        Own om;
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
            const real zg0 = g ? z[1] : z[0], zg1 = g ? z[3] : z[2], zg2 = g ? z[5] : z[4];     // this lane's columns of parity g
            if (valid) {
                real* zo = lds + ZOUT;
                zo[bA] = zg0; zo[bA + K2] = zg1; zo[bA + 2 * K2] = zg2;
                if (g == h) zo[b0 + 3 * K2] = z[6];
            }
            // z of the first own knot is the LEFT member's missing part: published now, half a pass before the hand-off
            if (valid && i == 0 && gm > 0) publish_quad(std::integral_constant<int, base + LQKC_W_Z>{}, std::true_type{}, ep, zg0, zg1, zg2, z[6]);
            real ct = zg0 * om.v[0];
            ct = fma(zg1, om.v[1], ct);
            ct = fma(zg2, om.v[2], ct);
            cterm = g == h ? fma(z[6], om.v[3], ct) : ct;
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
        real pk_[L::NPARK];
#pragma unroll
        for (int q = 0; q < L::NPARK; ++q) pk_[q] = lqk_ld(park + q * NTHR);
        SFor14<9>::run([&](auto jt) {                           // j = 1 .. 5
            constexpr int J = decltype(jt)::value - 8;
            const real xs = col(me.v, std::integral_constant<int, J>{});
            SFor14<7>::run([&](auto st) {
                constexpr int S = decltype(st)::value - 7;
                if constexpr (L::parked(S, J)) acc[S] = fma(pk_[L::pidx(S, J)], xs, acc[S]);
                else acc[S] = fma(Md[S][J], xs, acc[S]);
            });
        });
        SFor14<7>::run([&](auto st) {
            constexpr int S = decltype(st)::value - 7;
            if constexpr (L::parked(S, 6)) acc[S] = fma(pk_[L::pidx(S, 6)], xk6, acc[S]);
            else acc[S] = fma(Md[S][6], xk6, acc[S]);
        });
        Own o;
#pragma unroll
        for (int s = 0; s < 4; ++s) o.v[s] = acc[s] + lqk_quad<LQK_QP_H>(acc[s < 3 ? s + 4 : 3]);
        store_own(TOUT, o);
        // the merged rows of the last own knot are the RIGHT member's T[k0 - 1]
        if (valid && i == KL - 1 && gm < G - 1) publish_quad(std::integral_constant<int, base + LQKC_W_T>{}, std::false_type{}, ep, o.v[0], o.v[1], o.v[2], o.v[3]);
        real d0 = o.v[0] * me.v[0];
        d0 = fma(o.v[1], me.v[1], d0);
        d0 = fma(o.v[2], me.v[2], d0);
        const real d3 = o.v[3] * me.v[3];
        const real part = rpl_wave_fold((d0 + (h ? real(0) : d3)) + cterm);
        if (lane == 0) {
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, part);
            const f4 gr = {__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32)), __builtin_bit_cast(float, ep), 0.f};
            if (same_xcd) granule_store16<base, true>(my_words, 16u * (unsigned)wl, gr);
            else granule_store16<base, false>(my_words, 16u * (unsigned)wl, gr);
        }
        return Vec{me, om};
    };

    using SlotV = std::integral_constant<int, LQKC_SLOT_V>;
    using SlotE = std::integral_constant<int, LQKC_SLOT_E>;
    // The one hand-off of a half.  `poller`: this wave polls (a wave of the matrix that sits the half out — it starts while the pass still runs).
    // TV / ZV: the local vectors whose halo slots receive the neighbours' T and Z.  withZ: the pass produced z (false: block-Jacobi Pinv pass).
    auto exchange = [&](auto slot, bool poller, int TV, int ZV, bool withZ) -> real {
        constexpr int base = decltype(slot)::value;
        ++epoch;
        if (poller) {
            // lane l < G NWM: wave partial l (member l / NWM, wave l % NWM); lanes 32..45: entry l - 32 of the left member's T; lanes 48..61: entry
            // l - 48 of the right member's Z.  Byte offsets and LDS destinations: three 64-entry tables filled once per launch.
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int* tab = reinterpret_cast<const int*>(lds + L::TAB) + ln;
            const unsigned pbyte = (unsigned)tab[0];
            const unsigned vbyte = (unsigned)tab[64];
            const int dst = tab[128];
            const bool isZ = ln >= 48;
            const bool wantp = pbyte != 0xFFFFFFFFu;
            const bool wantv = vbyte != 0xFFFFFFFFu && (withZ || !isZ);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 xp = {0u, 0u, 0u, 0u}, xv = {0u, 0u, 0u, 0u};
            unsigned spins = 0;
            bool ok;
            const unsigned pb_ = wantp ? pbyte : 0u, vb_ = wantv ? vbyte : 0u;
            do {
                asm volatile("s_nop 4\n\t"
                             "global_load_dwordx4 %0, %2, %4 offset:%5 sc1\n\t"
                             "global_load_dwordx4 %1, %3, %4 offset:%5 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(xp), "=&v"(xv) : "v"(pb_), "v"(vb_), "s"(cl_words), "n"(8 * base) : "memory");
                ok = (!wantp || xp.z == epoch) && (!wantv || xv.z == epoch);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < CL_SPIN_LIMIT);
            if (wantv) (lds + (isZ ? ZV : TV))[dst] = __builtin_bit_cast(real, ((unsigned long long)xv.y << 32) | (unsigned long long)xv.x);
            const real partial = __builtin_bit_cast(real, ((unsigned long long)xp.y << 32) | (unsigned long long)xp.x);
            const real tot = rpl_wave_fold(wantp ? partial : real(0));
            if (ln == 0) { bc[(epoch & 1u) ? 3 : 0] = tot; if (spins >= CL_SPIN_LIMIT) bc[1] = 1.0; }
        }
        lds_barrier();
        if (bc[1] != 0.0) failed = true;
        // (two cells by epoch parity: the next hand-off's poller — another wavefront of this workgroup — may finish before a wavefront that takes no
        //  part in the next pass has read this value; with one cell that would be a race, however unlikely a wavefront lags a whole pass)
        return bc[(epoch & 1u) ? 3 : 0];
    };

    if (tid == 0) { bc[0] = 0.0; bc[1] = 0.0; bc[2] = 0.0; bc[3] = 0.0; }
    if (tid < 64) {
        const int l = tid;
        int* tab = reinterpret_cast<int*>(lds + L::TAB) + l;
        tab[0] = l < G * NWM ? 8 * ((l / NWM) * LQKC_WG_WORDS + 2 * (l % NWM)) : -1;
        const bool isT = l >= 32 && l < 46, isZ = l >= 48 && l < 62;
        const int e = isT ? l - 32 : l - 48;
        const bool have = isT ? gm > 0 : (isZ && gm < G - 1);
        const int src_m = isT ? gm - 1 : gm + 1;
        tab[64] = have ? 8 * (src_m * LQKC_WG_WORDS + (isT ? LQKC_W_T : LQKC_W_Z) + 2 * e) : -1;
        tab[128] = have ? L::at(isT ? 0 : KL + 1, e) : 0;
    }
    // ---- are all members of this cluster on one XCD? (pcg_lpkc_kernel) ----
    if (w == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        if (lane == 0) granule_store<LQKC_SLOT_X>(my_words, 0u, (1ull << 32) | xcc);
        unsigned long long x = 0;
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
            if (lane < G) {
                x = granule_load<LQKC_SLOT_X>(cl_words, 8u * (unsigned)(lane * LQKC_WG_WORDS));
                ok = (unsigned)(x >> 32) == 1u;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
        } while (++spins < (CL_SPIN_LIMIT >> 4));
        const bool all_same = __all(lane >= G || ((unsigned)(x >> 32) == 1u && (unsigned)x == xcc));
        if (lane == 0) reinterpret_cast<int*>(bc + 2)[0] = all_same ? 1 : 0;
    }
    lds_barrier();
    same_xcd = reinterpret_cast<const int*>(bc + 2)[0] != 0 && ca.l2_handoff != 0;
    lds_barrier();
    for (;;) {
        // ---- next trajectory of this cluster: own index first, then the leader draws from the queue ----
        ++seq;
        if (seq > 1) {
            if ((unsigned)ca.batch <= nclusters) break;
            if (w == 0) {
                int bn = 0;
                if (gm == 0) {
                    if (lane == 0) {
                        bn = (int)nclusters + (int)__hip_atomic_fetch_add(kp->queue, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (same_xcd) granule_store_l2<LQKC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                        else granule_store<LQKC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                    }
                } else {
                    unsigned long long x = 0;
                    unsigned spins = 0;
                    do {
                        x = granule_load<LQKC_SLOT_T>(cl_words, 0u);
                        if ((unsigned)(x >> 32) == seq) break;
                        __builtin_amdgcn_s_sleep(1);
                    } while (++spins < CL_SPIN_LIMIT);
                    bn = (int)(unsigned)x;
                    if (spins >= CL_SPIN_LIMIT && lane == 0) bc[1] = 1.0;
                }
                if (lane == 0) reinterpret_cast<int*>(bc + 2)[0] = bn;
            }
        } else if (tid == 0) {
            reinterpret_cast<int*>(bc + 2)[0] = cl;
        }
        lds_barrier();
        const int b = reinterpret_cast<const int*>(bc + 2)[0];
        if (bc[1] != 0.0 || b >= ca.batch) break;
        const real* gam = kp->p.gamma + (size_t)b * vstride;
        const real* lam_in = kp->p.lambda + (size_t)b * vstride;
        {
            const rsrc_t M = make_rsrc(static_cast<const char*>(static_cast<const void*>(isP ? kp->p.Pinv : kp->p.S)) + (size_t)b * mstride * 8, (uint32_t)(mstride * 8));
            // (lane constants made opaque per trajectory: hoisted out of the trajectory loop, the load addresses are 40 registers this kernel
            //  does not have — spilled once, reloaded between the loads of every trajectory, each reload waiting for all loads in flight)
            int i_st = i, h_st = h, g_st = g;
            asm volatile("" : "+v"(i_st), "+v"(h_st), "+v"(g_st));
            lqk_load_blocks(M, k0 + i_st, h_st, g_st, valid, valid && k0 + i_st > 0 && hasL, Md, Ml);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
#pragma unroll
        for (int q = 0; q < L::NPARK; ++q) park[q * NTHR] = Md[(49 - L::NPARK + q) % 7][(49 - L::NPARK + q) / 7];
        // ---- stage: everything <- 0, then P0 <- lambda0, lambda <- lambda0, R0 <- gamma for the own knots AND the replica knot k0 - 1 ----
        for (int e = tid; e < L::BC; e += NTHR) lds[e] = real(0);
        lds_barrier();
        for (int e = tid + (gm == 0 ? NS : 0); e < (KL + 1) * NS; e += NTHR) {
            const int sl = e / NS, ii = e - sl * NS;            // slot 0 = knot k0 - 1
            const int ge = (k0 - 1) * NS + e;
            const real l0 = lam_in[ge];
            lds[L::P0 + L::at(sl, ii)] = l0;
            lds[L::LAM + L::at(sl, ii)] = l0;
            lds[L::R0 + L::at(sl, ii)] = gam[ge];
        }
        lds_barrier();

        uint32_t iters = 0;
        uint32_t max_iter_exit = 1;
        real beta = real(0);
        bool p_pending = true;
        auto run_role = [&](auto role_tag) {
            constexpr bool P = decltype(role_tag)::value;
            const bool poll_s = P && w == NWM, poll_p = !P && w == 0;      // poller of the S half: first Pinv wave; of the Pinv half: wave 0
            Fetch f;
            Vec x;
            x.k = load_own(P ? L::R0 : L::P0, 0);
            x.m = load_own(P ? L::R0 : L::P0, -1);
            // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; eta = r . r~   (p = r~ is formed by the first S half: beta = 0) ----
            if constexpr (!P) (void)half(std::integral_constant<int, 0>{}, SlotV{}, f, x, real(0), L::US, L::ZS);
            (void)exchange(SlotV{}, poll_s, L::US, L::ZS, true);
            if constexpr (P) {
                f = fetch(L::US, L::ZS);
                x = half(std::integral_constant<int, 1>{}, SlotE{}, f, x, real(1), L::RT, L::ZP);
            }
            real eta = lqk_uniform(exchange(SlotE{}, poll_p, L::RT, L::ZP, p3));
            if constexpr (!P) f = fetch(L::RT, L::ZP);
            if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; return; }
            if (fabs(eta) < a.exit_tol) { max_iter_exit = 0; return; }
            for (int it = 0; it < a.max_iter; ++it) {
                real v;
                if constexpr (!P) {
                    x = half(std::integral_constant<int, 2>{}, SlotV{}, f, x, beta, L::US, L::ZS);
                    v = exchange(SlotV{}, false, L::US, L::ZS, true);
                    // alpha ; lambda += alpha p (own entries) — while the Pinv half runs
                    const Own cur = load_own(L::LAM, 0);
                    const real alpha = lqk_uniform(eta / v);
                    Own nw;
#pragma unroll
                    for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] + alpha * x.k.v[s];
                    store_own(L::LAM, nw);
                } else {
                    v = exchange(SlotV{}, poll_s, L::US, L::ZS, true);
                    f = fetch(L::US, L::ZS);
                    const real alpha = lqk_uniform(eta / v);
                    x = half(std::integral_constant<int, 1>{}, SlotE{}, f, x, alpha, L::RT, L::ZP);
                }
                const real eta_new = lqk_uniform(exchange(SlotE{}, poll_p, L::RT, L::ZP, p3));
                if constexpr (!P) f = fetch(L::RT, L::ZP);
                if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; break; }
                iters = (uint32_t)(it + 1);
                if (fabs(eta_new) < a.exit_tol) { max_iter_exit = 0; p_pending = false; break; }
                beta = lqk_uniform(eta_new / eta);
                eta = eta_new;
            }
            store_own(P ? L::R0 : L::P0, x.k);
        };
        if (isP) run_role(std::true_type{}); else run_role(std::false_type{});
        lds_barrier();

        // ---- write back own knots (a member that gave up leaves lambda alone: its trajectory's count stays short of G, the fix-up launch re-solves it
        //      from the handle's copy of lambda0) ----
        if (ca.test_fail && cl == 0 && gm == G - 1 && seq == 1) failed = true;
        if (failed) {
            if (tid == 0) { kp->p.iters[b] = 0xFFFFFFFFu; kp->p.max_iter_exit[b] = 2; }
            break;
        }
        for (int e = tid; e < KL * NS; e += NTHR) {
            const int sl = e / NS + 1, ii = e % NS;
            const size_t ge = (size_t)b * vstride + (size_t)k0 * NS + e;
            kp->p.lambda[ge] = lds[L::LAM + L::at(sl, ii)];
            if (kp->p.r_out) kp->p.r_out[ge] = lds[L::R0 + L::at(sl, ii)];
            if (kp->p.p_out) {
                real pv = lds[L::P0 + L::at(sl, ii)];
                if (p_pending) pv = (lds[L::RT + L::at(sl, ii)] + (p3 ? lds[L::ZP + L::at(sl + 1, ii)] : real(0))) + beta * pv;
                kp->p.p_out[ge] = pv;
            }
        }
        if (tid == 0) {
            if (gm == 0) { kp->p.iters[b] = iters; kp->p.max_iter_exit[b] = (uint8_t)max_iter_exit; }
            __hip_atomic_fetch_add(kp->fail_flags + (size_t)b * CL_FLAG_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();                                      // LDS is restaged for the next trajectory
    }
}

}  // namespace mpcg
