// pcg_lpk_cluster.hip.h — the lane-pair-per-knot PCG kernel (pcg_lpk.hip.h) for horizons one CU cannot hold (round 3): G workgroups on
// G CUs of one XCD solve ONE trajectory, each with up to 64 NWR consecutive knots of S and Pinv in its register file.  Successor of
// pcg_lpbc_kernel (pcg_lpb_cluster.hip.h), whose hand-off machinery — epoch-tagged granules in the XCD's L2, persistent clusters drawing
// trajectories from a queue, bounded spins + completion counts + fix-up launch — it reuses unchanged; what changes is everything between
// two hand-offs, which was 70 % of an iteration (profiles/r02_lpbc_phases.txt: passes 2 x 2,300 ticks, part-vector sums and vector updates
// 2 x 1,200, against 900 + 1,600 of polling).
//
// Decomposition.  Member g owns knots [k0, k1) with the lane-pair mapping of pcg_lpk_kernel (a knot = two adjacent lanes per matrix, 147
// packed FMAs per lane and pass, iterate vectors carried in registers, the reader rebuilds its operand).  A half-iteration of a member needs
// from outside exactly what the single-CU kernel reads from its neighbours' LDS slots:
//     T[k0 - 1]  the LEFT member's merged row pairs of its last knot   (operand rebuild of the replica knot k0 - 1, held by the lanes of knot k0)
//     Z[k1]      the RIGHT member's z = L_k1^T x_k1 of its first knot    (operand rebuild of the last own knot k1 - 1)
// plus the cluster-wide inner product.  All three exist when a pass ends and are needed when the next one starts: ONE hand-off per pass
// carries them — each as five 16-byte granules {tag, v0, v1, v2} published straight from the registers of the two lanes that hold them (z
// as soon as the transposed product is done, mid-pass), the wave partials as 8-byte granules.  The hand-off is polled by a wave of the
// matrix that sits the pass out, which therefore starts polling while the pass still runs; it drops the neighbours' vectors into the halo
// slots of the local T / Z vectors (slot 0 and slot KL + 1), folds the partials, and ONE barrier ends the exchange.  Two hand-offs and two
// barriers per PCG iteration, nothing else: no part vectors, no element-wise phases (pcg_lpbc_kernel: two hand-offs and four barriers).
#pragma once
#include "pcg_lpk.hip.h"

namespace mpcg {

// ---- hand-off cells in the cluster scratch (round 2's machinery, kept from the retired clustered lane-per-block kernel) ----
constexpr int LPBC_MAX_G = 8;              // members: G x NW wave partials are polled by the 64 lanes of one wave
constexpr int LPBC_WG_WORDS = 128;         // u64 words of hand-off cells per member (1 KB)
// Cells of one member: two alternating exchange slots, each = NW 8-byte granules {tag, wave partial} (words 0..7) and three groups of
// five 16-byte granules {tag, v0, v1, v2} — 14 values each: yD and yL of the last own knot (for the right neighbour), t (for the left one).
constexpr int LPBC_SLOT_V = 0, LPBC_SLOT_E = 40;
constexpr int LPBC_W_YD = 8, LPBC_W_YL = 18, LPBC_W_T = 28;
constexpr int LPBC_SLOT_T = 120;                     // leader only: {sequence number, trajectory index} of the cluster's current trajectory
constexpr int LPBC_SLOT_X = 121;                     // every member, once per launch: {1, XCC id} (the same-XCD check)

// Granule accesses in the "uniform 64-bit base (SGPR pair) + 32-bit lane byte offset" addressing form, spelled out: left to
// the compiler, the per-lane addresses of the two exchange slots become 64-bit VGPR pointers that are hoisted out of the PCG
// loop — four registers this kernel does not have; they spill, and the publishing wave reloads its store address from scratch
// right in front of every hand-off.  (s_nop 4: the base may just have been written by a v_readlane — an SGPR spill reload — and a
// VMEM instruction reading a VALU-written SGPR needs 5 wait states; the compiler's hazard recogniser does not look into asm.)
// sc1 = agent scope (write-through store / L2-coherent load), as __hip_atomic_*(relaxed, agent).
// WORD = compile-time word index inside the member's block of cells: the instruction's immediate offset, so that the two exchange
// slots and the hand-out word share ONE base register pair.
template <int WORD>
__device__ __forceinline__ void granule_store(gu64* sbase, unsigned byte_off, unsigned long long v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 offset:%3 sc1" : : "v"(byte_off), "v"(v), "s"(sbase), "n"(8 * WORD) : "memory");
}
// The same store WITHOUT sc1: the granule stays in this XCD's L2, where a poller on the same XCD finds it (its sc1 load bypasses
// only L1) without the round trip to the memory side that a write-through store forces on both.  Only valid when the whole
// cluster sits on one XCD, which the members verify at start-up (pcg_lpbc_kernel: `same_xcd`).
template <int WORD>
__device__ __forceinline__ void granule_store_l2(gu64* sbase, unsigned byte_off, unsigned long long v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2 offset:%3" : : "v"(byte_off), "v"(v), "s"(sbase), "n"(8 * WORD) : "memory");
}
template <int WORD>
__device__ __forceinline__ unsigned long long granule_load(const gu64* sbase, unsigned byte_off) {
    unsigned long long x;
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 offset:%3 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(byte_off), "s"(sbase), "n"(8 * WORD) : "memory");
    return x;
}

// 16-byte granules {tag, v0, v1, v2}: one dwordx4 store / load (observed untorn on gfx950, MI355X_MICROARCH.md "R2's granule"), three
// values per fabric / L2 transaction instead of one.  L2 = without sc1 (see granule_store_l2).
template <int WORD, bool L2>
__device__ __forceinline__ void granule_store16(gu64* sbase, unsigned byte_off, f4 v) {
    if constexpr (L2) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" : : "v"(byte_off), "v"(v), "s"(sbase), "n"(8 * WORD) : "memory");
    else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3 sc1\n\ts_nop 1" : : "v"(byte_off), "v"(v), "s"(sbase), "n"(8 * WORD) : "memory");
}
__device__ __forceinline__ f4 granule_load16(const gu64* sbase, unsigned byte_off) {
    f4 x;
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(byte_off), "s"(sbase) : "memory");
    return x;
}


// LDS: the lane-pair kernel's seven pair-major vectors (knot slot 0 = the left halo knot k0 - 1, slots 1..KL = own knots, slot KL + 1 = the
// right halo knot k1) | broadcast cell | hand-off tables (5 x 64 ints) | parked matrix pairs.
template <int NWR> struct LpkcLds {
    typedef LpkLds<NWR> B;
    static constexpr int NMAX = B::NMAX, NW = B::NW, KN = B::KN, VS = B::VS;
    static constexpr int P0 = 0, R0 = VS, US = 2 * VS, ZS = 3 * VS, RT = 4 * VS, ZP = 5 * VS, LAM = 6 * VS, BC = 7 * VS, TAB = BC + 4, MX = TAB + 5 * 64,
                         NPARK = 3, TOTAL = MX + NPARK * 2 * NW * 64;
    // float index of entry i of knot slot s inside a vector
    __host__ __device__ static constexpr int at(int s, int i) { return 2 * ((i >> 1) * KN + s) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lpkc_lds_floats(int NW) { return NW == 4 ? (size_t)LpkcLds<1>::TOTAL : (size_t)LpkcLds<2>::TOTAL; }

// Cells of one member (LPBC_WG_WORDS u64 words): two alternating exchange slots (LPBC_SLOT_V / LPBC_SLOT_E), each = NW/2 wave partials
// (words 0..3: {tag, value}) and two groups of five 16-byte granules: T of the last own knot (for the right neighbour), Z of the first
// (for the left one).  Granule q of a group carries entries (0,1,2) (3,4,5) (6,7,-) (8,9,10) (11,12,13) of the 14-vector: q = 0..2 come
// from lane 0 of the knot's pair, q = 3, 4 from lane 1.  Words LPBC_SLOT_T / LPBC_SLOT_X as in pcg_lpbc_kernel.
constexpr int LPKC_W_T = 8, LPKC_W_Z = 18;

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lpkc_kernel(ClusterArgs ca) {
    typedef LpkcLds<NWR> L;
    constexpr int NW = 4 * NWR, NTHR = NW * 64, NWM = NW / 2;
    const PcgArgs& a = ca.p;
    typedef const __attribute__((address_space(4))) ClusterArgs* kargp_t;
    const kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();     // `ca` itself, in the constant address space
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = ca.G;
    // members of a cluster share an XCD (see pcg_lpbc_kernel): workgroup b = 8 j + x holds member j % G of cluster 8 (j / G) + x
    const unsigned nclusters = (unsigned)ca.clusters;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int g = jx % G;
    const int cl = (jx / G) * 8 + xcd;
    if ((unsigned)cl >= nclusters) return;
    // gate of the symmetry latch (mpcg_pcg.hip: launch_guarded): every member reads the same word and leaves at once when it says the
    // matrices are not block-symmetric — no trajectory is drawn, every completion count stays 0, and the fix-up launch solves them all
    if (a.redo_flags && __hip_atomic_load(a.redo_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    const int k0 = (int)(((long)g * N) / G), k1 = (int)(((long)(g + 1) * N) / G);
    const int KL = k1 - k0;                             // own knots (launcher: 1 <= KL <= NMAX)
    float* bc = lds + L::BC;                            // [0] cluster-wide sum, [1] sticky timeout flag, [2] trajectory index (int)

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    gu64* my_words = (gu64*)ca.scratch + ((size_t)cl * G + g) * LPBC_WG_WORDS;
    gu64* cl_words = (gu64*)ca.scratch + (size_t)cl * G * LPBC_WG_WORDS;

    // ---- role of this wave, knot and column half of this lane (pcg_lpk_kernel's mapping; i = knot inside the member, slot i + 1) ----
    const bool isP = w >= NWM;
    const int wl = w - (isP ? NWM : 0);
    const int li = 64 * wl + lane;
    const int i = li >> 1, h = li & 1;
    const bool p3 = a.pcols == 3;
    const bool hasL = !isP || p3;
    const bool valid = i < KL;
    constexpr int KN = L::KN, K2 = 2 * KN;
    const int b0 = 2 * (i + 1);
    const int bA = b0 + (h ? 4 * K2 : 0);

    f2 Md[7][7], Ml[7][7];
    f2* const park = reinterpret_cast<f2*>(lds + L::MX) + tid;

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
#ifdef MPCG_PROF
    bool prof_on = false;
#endif
    bool same_xcd = false;
    unsigned epoch = 0, seq = 0;
    bool failed = false;

    struct Own { f2 v[4]; };
    auto load_own = [&](int X, int dk) -> Own {
        const float* x = lds + X + 2 * dk;
        Own o;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.v[s] = *reinterpret_cast<const f2*>(x + bA + K2 * s);
        o.v[3] = *reinterpret_cast<const f2*>(x + b0 + 3 * K2);
        return o;
    };
    auto store_own = [&](int X, const Own& o) {
        if (valid) {
            float* x = lds + X;
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<f2*>(x + bA + K2 * s) = o.v[s];
            *reinterpret_cast<f2*>(x + b0 + 3 * K2) = o.v[3];
        }
    };
    struct Fetch { f2 t[4], z[4], gt[4], gz[4]; };
    auto fetch = [&](int T, int Z) -> Fetch {
        const float* xt = lds + T;
        const float* xz = lds + Z + 2;
        Fetch f;
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.t[s] = lds_ld64(xt + bA + K2 * s); f.z[s] = lds_ld64(xz + bA + K2 * s); }
        f.t[3] = lds_ld64(xt + b0 + 3 * K2); f.z[3] = lds_ld64(xz + b0 + 3 * K2);
#pragma unroll
        for (int s = 0; s < 3; ++s) { f.gt[s] = lds_ld64(xt - 2 + bA + K2 * s); f.gz[s] = lds_ld64(xz - 2 + bA + K2 * s); }
        f.gt[3] = lds_ld64(xt - 2 + b0 + 3 * K2); f.gz[3] = lds_ld64(xz - 2 + b0 + 3 * K2);
        return f;
    };
    struct Vec { Own k, m; };
    // 14 values of a knot's lane pair as 16-byte granules of the group at word WORD: this lane's slots 0..2 -> two granules (entries
    // 0..5 / 8..13), lane 0 one more: {slot 3 (entries 6, 7), 0}.  Called by both lanes of the knot.
    auto publish_pair = [&](auto word_tag, unsigned ep, f2 s0, f2 s1, f2 s2, f2 s3) {
        constexpr int WORD = decltype(word_tag)::value;
        const float tg = __builtin_bit_cast(float, ep);
        const unsigned off = h ? 48u : 0u;                  // lane 1: granules 3, 4
        const f4 g0 = {tg, s0.x, s0.y, s1.x}, g1 = {tg, s1.y, s2.x, s2.y}, g2 = {tg, s3.x, s3.y, 0.f};
        if (same_xcd) {
            granule_store16<WORD, true>(my_words, off, g0);
            granule_store16<WORD + 2, true>(my_words, off, g1);
            if (!h) granule_store16<WORD + 4, true>(my_words, off, g2);
        } else {
            granule_store16<WORD, false>(my_words, off, g0);
            granule_store16<WORD + 2, false>(my_words, off, g1);
            if (!h) granule_store16<WORD + 4, false>(my_words, off, g2);
        }
    };

    // One half-iteration of this wave's matrix (pcg_lpk_kernel::half) + the publishing of what the neighbours and the reduction need.
    auto half = [&](auto mode_tag, auto slot, const Fetch& f, const Vec& old, float c, int TOUT, int ZOUT) -> Vec {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int base = decltype(slot)::value;
        [[maybe_unused]] constexpr int pb = base == LPBC_SLOT_V ? 0 : 8;     // (stamp numbers of the -DMPCG_PROF build)
        MPCG_STAMP(pb + 0);
        const unsigned ep = epoch + 1;                      // tag of the hand-off that follows this pass
        f2 xk[7];
        Own om;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { xk[s] = old.k.v[s]; om.v[s] = old.m.v[s]; }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) { const f2 u = f.t[s] + f.z[s]; xk[s] = MODE == 1 ? old.k.v[s] - c * u : u + c * old.k.v[s]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) { const f2 u = f.gt[s] + f.gz[s]; om.v[s] = MODE == 1 ? old.m.v[s] - c * u : u + c * old.m.v[s]; }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) xk[4 + s] = f2{dpp_partner(xk[s].x), dpp_partner(xk[s].y)};
        Own me;
#pragma unroll
        for (int s = 0; s < 4; ++s) me.v[s] = xk[s];
        f2 acc[7];
        float cterm = 0.f;
        const float xk6 = h ? xk[3].y : xk[3].x;
        if (hasL) {
            f2 z2[3];
            float z6;
            {
                f2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = Ml[0][j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = __builtin_elementwise_fma(Ml[s][j], xk[s], t[j]);
                z2[0] = f2{t[0].x + t[0].y, t[1].x + t[1].y};
                z2[1] = f2{t[2].x + t[2].y, t[3].x + t[3].y};
            }
            {
                f2 t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) t[j] = Ml[0][4 + j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] = __builtin_elementwise_fma(Ml[s][4 + j], xk[s], t[j]);
                z2[2] = f2{t[0].x + t[0].y, t[1].x + t[1].y};
                z6 = t[2].x + t[2].y;
            }
            const float xm6 = h ? om.v[3].y : om.v[3].x;
            if (valid) {
                float* zo = lds + ZOUT;
#pragma unroll
                for (int s = 0; s < 3; ++s) *reinterpret_cast<f2*>(zo + bA + K2 * s) = z2[s];
                // (slot 3, entry h: b0 + 3 K2 + h, formed from bA on the spot — as a lane constant of its own it was the one VGPR the kernel
                //  spilled: a scratch reload + wait in front of this store in every pass)
                //  (h = lane & 1, re-read from the hardware lane count so that nothing of it lives across the pass.)
                int ln6;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln6));
                zo[bA + ((ln6 & 1) ? 1 - K2 : 3 * K2)] = z6;
            }
            // z of the first own knot is the LEFT member's missing part: published now, half a pass before the hand-off
            if (valid && i == 0 && g > 0) publish_pair(std::integral_constant<int, base + LPKC_W_Z>{}, ep, z2[0], z2[1], z2[2], f2{z6, dpp_partner(z6)});
            f2 ct = z2[0] * om.v[0];
            ct = __builtin_elementwise_fma(z2[1], om.v[1], ct);
            ct = __builtin_elementwise_fma(z2[2], om.v[2], ct);
            cterm = fmaf(z6, xm6, ct.x + ct.y);
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Ml[s][0] * f2{om.v[0].x, om.v[0].x};
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                const float xs = (j & 1) ? om.v[j >> 1].y : om.v[j >> 1].x;
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Ml[s][j], f2{xs, xs}, acc[s]);
            }
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Ml[s][6], f2{xm6, xm6}, acc[s]);
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][0], f2{xk[0].x, xk[0].x}, acc[s]);
        } else {
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Md[s][0] * f2{xk[0].x, xk[0].x};
        }
        f2 pk_[L::NPARK];
#pragma unroll
        for (int q = 0; q < L::NPARK; ++q) pk_[q] = lds_ld64(reinterpret_cast<const float*>(park + q * NTHR));
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            const float xs = (j & 1) ? xk[j >> 1].y : xk[j >> 1].x;
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][j], f2{xs, xs}, acc[s]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_elementwise_fma(Md[s][6], f2{xk6, xk6}, acc[s]);
#pragma unroll
        for (int q = 0; q < L::NPARK; ++q) acc[4 + q] = __builtin_elementwise_fma(pk_[q], f2{xk6, xk6}, acc[4 + q]);
        Own o;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f2 oth = acc[s < 3 ? s + 4 : 3];
            o.v[s] = f2{acc[s].x + dpp_partner(oth.x), acc[s].y + dpp_partner(oth.y)};
        }
        store_own(TOUT, o);
        // the merged rows of the last own knot are the RIGHT member's T[k0 - 1]
        if (valid && i == KL - 1 && g < G - 1) publish_pair(std::integral_constant<int, base + LPKC_W_T>{}, ep, o.v[0], o.v[1], o.v[2], o.v[3]);
        f2 d0 = o.v[0] * me.v[0], d1 = o.v[1] * me.v[1];
        d0 = __builtin_elementwise_fma(o.v[2], me.v[2], d0);
        const f2 d3 = o.v[3] * me.v[3];
        const f2 dd = d0 + d1;
        const float part = wave_fold(((dd.x + dd.y) + (h ? 0.f : d3.x + d3.y)) + cterm);
        if (lane == 0) {
            const unsigned long long gran = ((unsigned long long)ep << 32) | __builtin_bit_cast(unsigned, part);
            if (same_xcd) granule_store_l2<base>(my_words, 8u * (unsigned)wl, gran);
            else granule_store<base>(my_words, 8u * (unsigned)wl, gran);
        }
        MPCG_STAMP(pb + 1);
        return Vec{me, om};
    };

    using SlotV = std::integral_constant<int, LPBC_SLOT_V>;
    using SlotE = std::integral_constant<int, LPBC_SLOT_E>;
    // The one hand-off of a half.  `poller`: this wave polls (a wave of the matrix that sits the half out — it starts while the pass still
    // runs).  TV / ZV: the local vectors whose halo slots receive the neighbours' T and Z.  withZ: the pass produced z (false: block-Jacobi Pinv pass).
    // Returns the cluster-wide inner product.
    auto exchange = [&](auto slot, bool poller, int TV, int ZV, bool withZ) -> float {
        constexpr int base = decltype(slot)::value;
        [[maybe_unused]] constexpr int pb = base == LPBC_SLOT_V ? 0 : 8;
        ++epoch;
        MPCG_STAMP(pb + 2);
        if (poller) {
            // lane l < G NWM polls wave partial l (member l / NWM, wave l % NWM); lanes 32..36 one 16-byte granule each of the left member's T,
            // lanes 40..44 of the right member's Z.  Byte offsets and LDS destinations: five 64-entry tables filled once per launch.
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int* tab = reinterpret_cast<const int*>(lds + L::TAB) + ln;
            const unsigned pbyte = (unsigned)tab[0];
            const unsigned vbyte = (unsigned)tab[64];
            const int d0 = tab[128], d1 = tab[192], d2 = tab[256];
            const bool isZ = ln >= 40;
#ifdef MPCG_CG_EMULATE
            // TIMING EXPERIMENT ONLY (tools/_prof/cg_emulate.py; wrong numerics): what a single-reduction (Chronopoulos-Gear) recurrence could
            // save at best without overlapping — its hand-off after the Pinv pass carries the neighbours' halo only: no partials polled, nothing folded
            const bool wantp = base != LPBC_SLOT_E && pbyte != 0xFFFFFFFFu;
#else
            const bool wantp = pbyte != 0xFFFFFFFFu;
#endif
            const bool wantv = vbyte != 0xFFFFFFFFu && (withZ || !isZ);
            unsigned long long x = 0;
            f4 xv = {0.f, 0.f, 0.f, 0.f};
            unsigned spins = 0;
            bool ok;
            // one poll = BOTH loads in flight, one wait (the two sc1 loads one after the other, each with its own vmcnt(0), made a poll
            // iteration two L2 round trips long: ~700 cycles of detection granularity on a hand-off that takes ~1,000)
            const unsigned pb_ = wantp ? pbyte : 0u, vb_ = wantv ? vbyte : 0u;
            do {
                asm volatile("s_nop 4\n\t"
                             "global_load_dwordx2 %0, %2, %4 offset:%5 sc1\n\t"
                             "global_load_dwordx4 %1, %3, %4 offset:%5 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(x), "=&v"(xv) : "v"(pb_), "v"(vb_), "s"(cl_words), "n"(8 * base) : "memory");
                ok = (!wantp || (unsigned)(x >> 32) == epoch) && (!wantv || __builtin_bit_cast(unsigned, xv.x) == epoch);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < CL_SPIN_LIMIT);
            MPCG_STAMP(pb + 3);
            if (wantv) {
                float* dv = lds + (isZ ? ZV : TV);
                dv[d0] = xv.y;
                dv[d1] = xv.z;
                if (d2 >= 0) dv[d2] = xv.w;
            }
#ifdef MPCG_CG_EMULATE
            const float tot = base == LPBC_SLOT_E ? 1.0f : wave_fold(wantp ? __builtin_bit_cast(float, (unsigned)x) : 0.f);
#else
            const float tot = wave_fold(wantp ? __builtin_bit_cast(float, (unsigned)x) : 0.f);
#endif
            if (ln == 0) { bc[(epoch & 1u) ? 3 : 0] = tot; if (spins >= CL_SPIN_LIMIT) bc[1] = 1.f; }
            MPCG_STAMP(pb + 4);
        }
        lds_barrier();
        MPCG_STAMP(pb + 5);
        if (bc[1] != 0.f) failed = true;
        // (two cells by epoch parity: the next hand-off's poller — another wavefront of this workgroup — may finish before a wavefront that takes no
        //  part in the next pass has read this value; with one cell that would be a race, however unlikely a wavefront lags a whole pass)
        return bc[(epoch & 1u) ? 3 : 0];
    };

    if (tid == 0) { bc[0] = 0.f; bc[1] = 0.f; bc[3] = 0.f; }
    if (tid < 64) {
        const int l = tid;
        int* tab = reinterpret_cast<int*>(lds + L::TAB) + l;
        tab[0] = l < G * NWM ? 8 * ((l / NWM) * LPBC_WG_WORDS + l % NWM) : -1;
        // 16-byte granule q of: lanes 32..36 T of member g - 1 (into slot 0), lanes 40..44 Z of member g + 1 (into slot KL + 1)
        const bool isT = l >= 32 && l < 37, isZ = l >= 40 && l < 45;
        const int q = isT ? l - 32 : l - 40;
        const bool have = isT ? g > 0 : (isZ && g < G - 1);
        const int src_m = isT ? g - 1 : g + 1;
        tab[64] = have ? 8 * (src_m * LPBC_WG_WORDS + (isT ? LPKC_W_T : LPKC_W_Z) + 2 * q) : -1;
        const int e0 = q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 6 : q == 3 ? 8 : 11;        // first entry of granule q
        const int sl = isT ? 0 : KL + 1;
        tab[128] = L::at(sl, e0);
        tab[192] = L::at(sl, e0 + 1);
        tab[256] = q == 2 ? -1 : L::at(sl, e0 + 2);
    }
    // ---- are all members of this cluster on one XCD? (pcg_lpbc_kernel) ----
    if (w == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        if (lane == 0) granule_store<LPBC_SLOT_X>(my_words, 0u, (1ull << 32) | xcc);
        unsigned long long x = 0;
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
            if (lane < G) {
                x = granule_load<LPBC_SLOT_X>(cl_words, 8u * (unsigned)(lane * LPBC_WG_WORDS));
                ok = (unsigned)(x >> 32) == 1u;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
        } while (++spins < (CL_SPIN_LIMIT >> 4));
        const bool all_same = __all(lane >= G || ((unsigned)(x >> 32) == 1u && (unsigned)x == xcc));
        if (lane == 0) reinterpret_cast<int*>(bc)[2] = all_same ? 1 : 0;
    }
    lds_barrier();
    same_xcd = reinterpret_cast<const int*>(bc)[2] != 0 && ca.l2_handoff != 0;
    lds_barrier();
    for (;;) {
        // ---- next trajectory of this cluster (pcg_lpbc_kernel: own index first, then the leader draws from the queue) ----
        ++seq;
        if (seq > 1) {
            if ((unsigned)ca.batch <= nclusters) break;
            if (w == 0) {
                int bn = 0;
                if (g == 0) {
                    if (lane == 0) {
                        kargp_t k_q = kp;
                        asm volatile("" : "+s"(k_q));
                        bn = (int)nclusters + (int)__hip_atomic_fetch_add(k_q->queue, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (same_xcd) granule_store_l2<LPBC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                        else granule_store<LPBC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                    }
                } else {
                    unsigned long long x = 0;
                    unsigned spins = 0;
                    do {
                        x = granule_load<LPBC_SLOT_T>(cl_words, 0u);
                        if ((unsigned)(x >> 32) == seq) break;
                        __builtin_amdgcn_s_sleep(1);
                    } while (++spins < CL_SPIN_LIMIT);
                    bn = (int)(unsigned)x;
                    if (spins >= CL_SPIN_LIMIT && lane == 0) bc[1] = 1.f;
                }
                if (lane == 0) reinterpret_cast<int*>(bc)[2] = bn;
            }
        } else if (tid == 0) {
            reinterpret_cast<int*>(bc)[2] = cl;
        }
        lds_barrier();
        const int draw = reinterpret_cast<const int*>(bc)[2];
        if (bc[1] != 0.f || draw >= ca.batch) break;
        kargp_t k_in = kp;
        asm volatile("" : "+s"(k_in));
        // (dispatch order: the q-th draw of the call solves trajectory order[q] — longest-expected first, sched_order_kernel)
        const int b = sched_pick(k_in->p.order, draw, k_in->p.order_tag);
        const float* gam = k_in->p.gamma + (size_t)b * vstride;
        const float* lam_in = k_in->p.lambda + (size_t)b * vstride;
        int t_st = tid, i_st = i, h_st = h;
        asm volatile("" : "+v"(t_st), "+v"(i_st), "+v"(h_st));
        {
            // matrix registers: seven columns of D_k and L_k, k = k0 + i (lpk_load_blocks, pcg_lpk.hip.h; fp16 storage converted once, here)
            const int esz = k_in->p.esz;
            const size_t es = esz == 2 ? 2 : 4;
            const rsrc_t M = make_rsrc(static_cast<const char*>(isP ? k_in->p.Pinv : k_in->p.S) + (size_t)b * mstride * es, (uint32_t)(mstride * es));
            const bool okD = valid, okL = valid && k0 + i_st > 0 && hasL;
            if (esz == 2) lpk_load_blocks<2>(M, k0 + i_st, h_st, okD, okL, Md, Ml);
            else lpk_load_blocks<4>(M, k0 + i_st, h_st, okD, okL, Md, Ml);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
#pragma unroll
        for (int q = 0; q < L::NPARK; ++q) park[q * NTHR] = Md[4 + q][6];
        // ---- stage: everything <- 0, then P0 <- lambda0, lambda <- lambda0, R0 <- gamma for the own knots AND the replica knot k0 - 1 ----
        for (int e = t_st; e < L::BC; e += NTHR) lds[e] = 0.f;
        lds_barrier();
        for (int e = t_st + (g == 0 ? NS : 0); e < (KL + 1) * NS; e += NTHR) {
            const int sl = e / NS, ii = e - sl * NS;            // slot 0 = knot k0 - 1
            const int ge = (k0 - 1) * NS + e;
            const float l0 = lam_in[ge];
            lds[L::P0 + L::at(sl, ii)] = l0;
            lds[L::LAM + L::at(sl, ii)] = l0;
            lds[L::R0 + L::at(sl, ii)] = gam[ge];
        }
        lds_barrier();

        uint32_t iters = 0;
        uint32_t max_iter_exit = 1;
        float beta = 0.f;
        bool p_pending = true;
        // the S waves and the Pinv waves run the same hand-off / barrier sequence through two separate code paths (pcg_lpk_kernel)
        auto run_role = [&](auto role_tag) {
            constexpr bool P = decltype(role_tag)::value;
            const bool poll_s = P && w == NWM, poll_p = !P && w == 0;      // poller of the S half: first Pinv wave; of the Pinv half: wave 0
            Fetch f;
            Vec x;
            x.k = load_own(P ? L::R0 : L::P0, 0);
            x.m = load_own(P ? L::R0 : L::P0, -1);
            // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; eta = r . r~   (p = r~ is formed by the first S half: beta = 0) ----
            if constexpr (!P) (void)half(std::integral_constant<int, 0>{}, SlotV{}, f, x, 0.f, L::US, L::ZS);
            (void)exchange(SlotV{}, poll_s, L::US, L::ZS, true);
            if constexpr (P) {
                f = fetch(L::US, L::ZS);
                x = half(std::integral_constant<int, 1>{}, SlotE{}, f, x, 1.f, L::RT, L::ZP);
            }
            float eta = uniform(exchange(SlotE{}, poll_p, L::RT, L::ZP, p3));
            if constexpr (!P) f = fetch(L::RT, L::ZP);
            if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; return; }
            if (fabsf(eta) < a.exit_tol) { max_iter_exit = 0; return; }
            for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
                prof_on = cl == 0 && g == 0 && it == 20;
#endif
                float v;
                if constexpr (!P) {
                    x = half(std::integral_constant<int, 2>{}, SlotV{}, f, x, beta, L::US, L::ZS);
                    v = exchange(SlotV{}, false, L::US, L::ZS, true);
                    // alpha ; lambda += alpha p (own entries) — while the Pinv half runs
                    const Own cur = load_own(L::LAM, 0);
                    const float alpha = uniform(eta / v);
                    Own nw;
#pragma unroll
                    for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] + alpha * x.k.v[s];
                    store_own(L::LAM, nw);
                } else {
                    v = exchange(SlotV{}, poll_s, L::US, L::ZS, true);
                    f = fetch(L::US, L::ZS);
                    const float alpha = uniform(eta / v);
                    x = half(std::integral_constant<int, 1>{}, SlotE{}, f, x, alpha, L::RT, L::ZP);
                }
                const float eta_new = uniform(exchange(SlotE{}, poll_p, L::RT, L::ZP, p3));
                if constexpr (!P) f = fetch(L::RT, L::ZP);
                if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; break; }
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; p_pending = false; break; }
                beta = uniform(eta_new / eta);
                eta = eta_new;
            }
            store_own(P ? L::R0 : L::P0, x.k);
        };
        if (isP) run_role(std::true_type{}); else run_role(std::false_type{});
        lds_barrier();

        // ---- write back own knots (a member that gave up leaves lambda alone: its trajectory's count stays short of G, the fix-up launch re-solves it) ----
        kargp_t k_out = kp;
        asm volatile("" : "+s"(k_out));
        if (ca.test_fail && cl == 0 && g == G - 1 && seq == 1) failed = true;      // ("cluster_test_fail": a member that gives up when its peers are past their last hand-off)
        if (failed) {
            if (tid == 0) { k_out->p.iters[b] = 0xFFFFFFFFu; k_out->p.max_iter_exit[b] = 2; }
            break;
        }
        {
            int t_wb = tid;
            asm volatile("" : "+v"(t_wb));
            for (int e = t_wb; e < KL * NS; e += NTHR) {
                const int sl = e / NS + 1, ii = e % NS;
                const size_t ge = (size_t)b * vstride + (size_t)k0 * NS + e;
                k_out->p.lambda[ge] = lds[L::LAM + L::at(sl, ii)];
                if (k_out->p.r_out) k_out->p.r_out[ge] = lds[L::R0 + L::at(sl, ii)];
                if (k_out->p.p_out) {
                    float pv = lds[L::P0 + L::at(sl, ii)];
                    if (p_pending) pv = (lds[L::RT + L::at(sl, ii)] + (p3 ? lds[L::ZP + L::at(sl + 1, ii)] : 0.f)) + beta * pv;
                    k_out->p.p_out[ge] = pv;
                }
            }
        }
        if (tid == 0) {
            if (g == 0) {
                k_out->p.iters[b] = iters;
                k_out->p.max_iter_exit[b] = (uint8_t)max_iter_exit;
            }
            __hip_atomic_fetch_add(k_out->fail_flags + (size_t)b * CL_FLAG_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();                                      // LDS is restaged for the next trajectory
    }
}

}  // namespace mpcg
