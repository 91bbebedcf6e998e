// pcg_lpb_cluster.hip.h — the lane-per-block PCG kernel (pcg_lpb.hip.h) for horizons one CU cannot hold: G workgroups
// on G CUs solve ONE trajectory, each with up to 64 NWR consecutive knots of S and Pinv in its register file
// (one 14x14 block of the block lower triangle per lane, N = 256: G = 2, N = 512: G = 4 with NWR = 2).
//
// Decomposition.  Member g owns knots [k0, k1), their diagonal blocks D_k and the sub-diagonal blocks
// L_k = S[k, left], k in [k0, k1) — i.e. also the block L_k0 that couples it to its left neighbour — and keeps, besides
// its own knots of p and r, ONE replica knot k0 - 1 of both.  With the replica it forms L_k0 x_{k0-1} (for its own y_k0),
// the coupling term of the inner product (counted twice, as in the single-workgroup kernel) and t = L_k0^T x_k0, the part
// of y_{k0-1} that the LEFT neighbour lacks.  What crosses a boundary per matrix pass is therefore
//     right -> left : t = L_k0^T x_k0                         (completes the left member's last knot)
//     left -> right : s = (D x)_{k0-1} + (L x)_{k0-1}         (lets the right member update its replica of knot k0 - 1:
//                                                              y_{k0-1} = s + t, the same bits the owner computes)
// and both are needed at the same point as the all-reduced inner product — after the pass, before the vector update.  So
// ONE hand-off per pass carries everything, and it is published straight from the registers that hold it, the moment it exists: every
// wave its partial of the inner product (an 8-byte {tag, value} granule), the lanes of block row k1 - 1 the diagonal and the
// sub-diagonal part of s, lane 0 of the first off-diagonal wave t (five 16-byte {tag, v0, v1, v2} granules each) — the R2 recipe of
// cdna_hip_programming.md §6 G16.  One wave polls all of them, drops the neighbours' parts into LDS and folds the partials; one
// barrier ends the exchange.  Two exposed hand-offs per PCG iteration; the row-triple cluster kernel needs four (2 all-reduces + 2
// halo fetches) and re-reads nothing either, but runs the slower row-pair arithmetic.  No operand halo is ever waited for inside a pass.
//
// Clusters are PERSISTENT: the launch holds as many clusters as fit the chip (every member resident) and each cluster draws
// trajectories from a queue (one atomic counter; the leader publishes the index to its peers as one more tagged granule)
// until the batch is done.  A call of any batch size is ONE launch, and solves that exit early on the tolerance — the MPC
// loop's warm-started solves: 25 iterations on average, a few at the cap — do not hold a whole wave of clusters back.
//
// Same fail-safe as pcg_cluster_kernel: bounded spins; every member that FINISHES a trajectory counts itself in that
// trajectory's flag word, a member that times out stops (and so, one bounded spin later, do its peers); the host follows the
// launch with the single-workgroup kernel restricted to the trajectories whose count is not G — abandoned or never drawn.
#pragma once
#include "pcg_lpb.hip.h"

namespace mpcg {

// LDS layout: six vectors of NMAX + 2 knot slots.  Slot 0 = replica of knot k0 - 1, slots 1..KL = own knots, slot NMAX + 1 = dump
// of idle lanes.  p | r | lambda | yD | yL | yT | 2 NW wave partials | broadcast cell | hand-off tables.
template <int NWR> struct LpbcLds {
    static constexpr int NMAX = 64 * NWR, NW = 4 * NWR, SLOTS = NMAX + 2, DUMP = NMAX + 1;
    static constexpr int VS = (int)r4((size_t)SLOTS * NS);
    static constexpr int XP = 0, XR = VS, LAM = 2 * VS, YD = 3 * VS, YL = 4 * VS, YT = 5 * VS, RED = 6 * VS, BC = RED + (int)r4(2 * NW),
                         TAB = BC + 4,          // 3 x 64 ints: what lane l of the publishing wave polls / publishes (filled once per launch)
                         TOTAL = TAB + 3 * 64;
};
__host__ __device__ constexpr size_t pcg_lpbc_lds_floats(int NW) { return NW == 4 ? (size_t)LpbcLds<1>::TOTAL : (size_t)LpbcLds<2>::TOTAL; }

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
// 14 values as five 16-byte granules at word WORD of the member's cells (executed by the one lane that holds them): one asm block —
// one hazard pad for the base, five stores back to back
template <int WORD, bool L2>
__device__ __forceinline__ void publish14(gu64* sbase, unsigned ep, const f2 (&v)[7]) {
    const float tg = __builtin_bit_cast(float, ep);
    const f4 g0 = {tg, v[0].x, v[0].y, v[1].x}, g1 = {tg, v[1].y, v[2].x, v[2].y}, g2 = {tg, v[3].x, v[3].y, v[4].x},
             g3 = {tg, v[4].y, v[5].x, v[5].y}, g4 = {tg, v[6].x, v[6].y, 0.f};
    const unsigned z = 0u;
    if constexpr (L2)
        asm volatile("s_nop 4\n\t"
                     "global_store_dwordx4 %0, %1, %6 offset:%7\n\t"
                     "global_store_dwordx4 %0, %2, %6 offset:%7+16\n\t"
                     "global_store_dwordx4 %0, %3, %6 offset:%7+32\n\t"
                     "global_store_dwordx4 %0, %4, %6 offset:%7+48\n\t"
                     "global_store_dwordx4 %0, %5, %6 offset:%7+64\n\t"
                     "s_nop 1"
                     : : "v"(z), "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "s"(sbase), "n"(8 * WORD) : "memory");
    else
        asm volatile("s_nop 4\n\t"
                     "global_store_dwordx4 %0, %1, %6 offset:%7 sc1\n\t"
                     "global_store_dwordx4 %0, %2, %6 offset:%7+16 sc1\n\t"
                     "global_store_dwordx4 %0, %3, %6 offset:%7+32 sc1\n\t"
                     "global_store_dwordx4 %0, %4, %6 offset:%7+48 sc1\n\t"
                     "global_store_dwordx4 %0, %5, %6 offset:%7+64 sc1\n\t"
                     "s_nop 1"
                     : : "v"(z), "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "s"(sbase), "n"(8 * WORD) : "memory");
}

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lpbc_kernel(ClusterArgs ca) {
    typedef LpbcLds<NWR> L;
    constexpr int NW = 4 * NWR, NTHR = NW * 64;
    const PcgArgs& a = ca.p;
    typedef const __attribute__((address_space(4))) ClusterArgs* kargp_t;
    const kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();     // `ca` itself, in the constant address space
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = ca.G;
    // Members of a cluster share an XCD: workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md "Contract"; a
    // hand-off inside an XCD is 0.1-0.3 us cheaper than across), so the launch is 8 lanes of workgroups b = 8 j + x, and lane x
    // holds the members j % G of clusters 8 (j / G) + x.  The grid is rounded up to whole groups of 8 clusters; the surplus ones leave.
    const unsigned nclusters = (unsigned)ca.clusters;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int g = jx % G;                               // member of its cluster
    const int cl = (jx / G) * 8 + xcd;                  // cluster of this launch
    if ((unsigned)cl >= nclusters) return;
    const int k0 = (int)(((long)g * N) / G), k1 = (int)(((long)(g + 1) * N) / G);
    const int KL = k1 - k0;                             // own knots (launcher: 1 <= KL <= NMAX)
    float* bc = lds + L::BC;                            // [0] cluster-wide sum, [1] sticky timeout flag, [2] trajectory index (int)

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    gu64* my_words = (gu64*)ca.scratch + ((size_t)cl * G + g) * LPBC_WG_WORDS;
    gu64* cl_words = (gu64*)ca.scratch + (size_t)cl * G * LPBC_WG_WORDS;

    // ---- role of this wave, block of this lane (pcg_lpb_kernel's roles; i = index of the block inside the member) ----
    const int role = w / NWR;
    const bool isP = role >= 2, isL = (role & 1) == 0;
    const int i = 64 * (w - role * NWR) + lane;         // block row k0 + i: reads slot i + 1 (and i), writes slot i + 1 (and yT of slot i)
    const bool p3 = a.pcols == 3;
    const bool wave_on = !(isP && isL && !p3);
    const bool valid = wave_on && i < KL && !(isL && k0 + i == 0);
    const int sr = (i < KL ? i : KL - 1) + 1;           // slot this lane reads as "knot k" (clamped: its block is all-zero)
    float* const xk = lds + sr * NS;
    float* const wk = lds + (valid ? i + 1 : L::DUMP) * NS;

    f4 m4[BLK4];
    auto mp = [&](int u, int r) -> f2 {
        const int e = NS * u + 2 * r;
        const f4 v = m4[e >> 2];
        return (e & 2) ? f2{v.z, v.w} : f2{v.x, v.y};
    };

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
    int prof_base = 0;
#endif
    bool same_xcd = false;                             // uniform: all members of this cluster run on one XCD (set below)
    unsigned epoch = 0, seq = 0;                       // hand-offs / trajectories of this cluster so far: tags never repeat inside a launch
    bool failed = false;                               // uniform across the workgroup (published through LDS)
    // The one hand-off of a pass.  Called by all threads after the pass; returns the cluster-wide inner product.  On return
    // yD[slot 0] holds the left neighbour's s and yT[slot KL] the right neighbour's t (where those neighbours exist).
    // parts3: the pass wrote off-diagonal parts (false for the block-Jacobi preconditioner pass: s = yD alone).
    // one pass of this wave's blocks over the vector at buffer offset X (pcg_lpb_kernel::pass with slot addressing)
    auto pass = [&](int X, auto slot) {
        constexpr int base = decltype(slot)::value;
        const unsigned ep = epoch + 1;                  // tag of the hand-off that follows this pass
        f2 xa[7], xb[7], acc[7];
        {
            const f2* xa2 = reinterpret_cast<const f2*>(xk + X + (isL ? -NS : 0));
            const f2* xb2 = reinterpret_cast<const f2*>(xk + X);
#pragma unroll
            for (int r = 0; r < 7; ++r) { xa[r] = xa2[r]; xb[r] = xb2[r]; acc[r] = f2{0.f, 0.f}; }
        }
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const float xs = (u & 1) ? xa[u >> 1].y : xa[u >> 1].x;
#pragma unroll
            for (int r = 0; r < 7; ++r) acc[r] = __builtin_elementwise_fma(mp(u, r), f2{xs, xs}, acc[r]);
        }
        f2 dt0 = {0.f, 0.f}, dt1 = {0.f, 0.f};
        f2* yo2 = reinterpret_cast<f2*>(wk + (isL ? L::YL : L::YD));
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            yo2[r] = acc[r];
            if (r & 1) dt1 = __builtin_elementwise_fma(acc[r], xb[r], dt1);
            else dt0 = __builtin_elementwise_fma(acc[r], xb[r], dt0);
        }
        const f2 dt = dt0 + dt1;
        const float part = wave_fold(isL ? 2.f * (dt.x + dt.y) : dt.x + dt.y);
        // published straight from the registers that hold them, the moment they exist: the wave partial by lane 0, the last own
        // knot's yD / yL (the right neighbour's s) by the lane of block row KL - 1
        if (lane == 0) {
            const unsigned long long gran = ((unsigned long long)ep << 32) | __builtin_bit_cast(unsigned, part);
            if (same_xcd) granule_store_l2<base>(my_words, 8u * (unsigned)w, gran);
            else granule_store<base>(my_words, 8u * (unsigned)w, gran);
        }
        if (valid && i == KL - 1 && g < G - 1) {
            if (isL) { if (same_xcd) publish14<base + LPBC_W_YL, true>(my_words, ep, acc); else publish14<base + LPBC_W_YL, false>(my_words, ep, acc); }
            else { if (same_xcd) publish14<base + LPBC_W_YD, true>(my_words, ep, acc); else publish14<base + LPBC_W_YD, false>(my_words, ep, acc); }
        }
        if (isL) {
            f2 tq[7];
            f2* yt2 = reinterpret_cast<f2*>(wk + L::YT + (valid ? -NS : 0));
#pragma unroll
            for (int u = 0; u < 12; u += 4) {
                f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f}, t2 = {0.f, 0.f}, t3 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    t0 = __builtin_elementwise_fma(mp(u, r), xb[r], t0);
                    t1 = __builtin_elementwise_fma(mp(u + 1, r), xb[r], t1);
                    t2 = __builtin_elementwise_fma(mp(u + 2, r), xb[r], t2);
                    t3 = __builtin_elementwise_fma(mp(u + 3, r), xb[r], t3);
                }
                tq[u >> 1] = f2{t0.x + t0.y, t1.x + t1.y};
                tq[(u >> 1) + 1] = f2{t2.x + t2.y, t3.x + t3.y};
                yt2[u >> 1] = tq[u >> 1];
                yt2[(u >> 1) + 1] = tq[(u >> 1) + 1];
            }
            {
                f2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 7; ++r) {
                    t0 = __builtin_elementwise_fma(mp(12, r), xb[r], t0);
                    t1 = __builtin_elementwise_fma(mp(13, r), xb[r], t1);
                }
                tq[6] = f2{t0.x + t0.y, t1.x + t1.y};
                yt2[6] = tq[6];
            }
            if (valid && i == 0 && g > 0) {             // t = L_k0^T x_k0: the left neighbour's missing part
                if (same_xcd) publish14<base + LPBC_W_T, true>(my_words, ep, tq); else publish14<base + LPBC_W_T, false>(my_words, ep, tq);
            }
        }
    };
    // a wave that sits a pass out still owes its (zero) partial to the hand-off
    auto idle_partial = [&](auto slot) {
        constexpr int base = decltype(slot)::value;
        if (lane == 0) {
            const unsigned long long gran = (unsigned long long)(epoch + 1) << 32;
            if (same_xcd) granule_store_l2<base>(my_words, 8u * (unsigned)w, gran);
            else granule_store<base>(my_words, 8u * (unsigned)w, gran);
        }
    };

    using SlotV = std::integral_constant<int, LPBC_SLOT_V>;
    using SlotE = std::integral_constant<int, LPBC_SLOT_E>;
    auto exchange = [&](auto slot, bool parts3) -> float {
        constexpr int base = decltype(slot)::value;
        MPCG_STAMP(prof_base + 0);
        ++epoch;
        if (w == 0) {
            // Lane l < G NW polls wave partial l (member l / NW, wave l % NW) — 8-byte granules; lanes 0..14 also poll one 16-byte
            // granule each: 0..4 the left neighbour's yD, 5..9 its yL, 10..14 the right neighbour's t.  Byte offsets into the cluster's
            // cells and the LDS destinations come from two 64-entry tables filled once per launch.
            int lane;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
            const int* tab = reinterpret_cast<const int*>(lds + L::TAB) + lane;
            const unsigned pbyte = (unsigned)tab[0];               // this lane's partial granule, 0xFFFFFFFF: none
            const unsigned vbyte = (unsigned)tab[64];              // this lane's 16-byte granule (without the slot base), 0xFFFFFFFF: none
            const int vdst = tab[128];                             // where its three values go in LDS
            const bool wantp = pbyte != 0xFFFFFFFFu;
            const bool wantv = vbyte != 0xFFFFFFFFu && (parts3 || lane < 5);      // block-Jacobi pass: only yD crosses
            MPCG_STAMP(prof_base + 1);
            unsigned long long x = 0;
            f4 xv = {0.f, 0.f, 0.f, 0.f};
            unsigned spins = 0;
            bool ok;
            do {
                ok = true;
                if (wantp) {
                    x = granule_load<base>(cl_words, pbyte);
                    ok = (unsigned)(x >> 32) == epoch;
                }
                if (wantv) {
                    xv = granule_load16(cl_words + base, vbyte);
                    ok = ok && __builtin_bit_cast(unsigned, xv.x) == epoch;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < CL_SPIN_LIMIT);
            MPCG_STAMP(prof_base + 2);
            if (wantv) {                                           // yD / yL of slot 0, yT of slot KL: three values per lane (the fifth granule: two)
                lds[vdst] = xv.y;
                lds[vdst + 1] = xv.z;
                if (lane % 5 != 4) lds[vdst + 2] = xv.w;
            }
            // sum of all wave partials, fixed order: the 16-lane rows by DPP, the four rows in sequence (members x waves in lane order)
            const float tot = wave_fold(wantp ? __builtin_bit_cast(float, (unsigned)x) : 0.f);
            if (lane == 0) { bc[0] = tot; if (spins >= CL_SPIN_LIMIT) bc[1] = 1.f; }
        }
        lds_barrier();                                  // parts of every wave, the neighbours' parts and the sum are in LDS
        MPCG_STAMP(prof_base + 3);
        if (bc[1] != 0.f) failed = true;
        return bc[0];
    };

    f2* xp2 = reinterpret_cast<f2*>(lds + L::XP);
    f2* xr2 = reinterpret_cast<f2*>(lds + L::XR);
    f2* lam2 = reinterpret_cast<f2*>(lds + L::LAM);
    const f2* yD2 = reinterpret_cast<const f2*>(lds + L::YD);
    const f2* yL2 = reinterpret_cast<const f2*>(lds + L::YL);
    const f2* yT2 = reinterpret_cast<const f2*>(lds + L::YT);
    // element-wise phases: float2 items of slots [0 or 1, KL]; every thread owns items e0 and e0 + NTHR (7 (KL + 1) <= 2 NTHR)
    const int e_lo = g == 0 ? NS / 2 : 0, e_hi = (NS / 2) * (KL + 1);
    const bool ok0 = e_lo + tid < e_hi, ok1 = e_lo + tid + NTHR < e_hi;
    const int e0 = ok0 ? e_lo + tid : e_lo, e1 = ok1 ? e_lo + tid + NTHR : e0;

    if (tid == 0) { bc[0] = 0.f; bc[1] = 0.f; bc[3] = 0.f; }     // bc[3]: the zero a one-part message adds
    if (tid < 64) {
        const int l = tid;
        int* tab = reinterpret_cast<int*>(lds + L::TAB) + l;
        // wave partial l: member l / NW, word l % NW of its slot
        tab[0] = l < G * NW ? 8 * ((l / NW) * LPBC_WG_WORDS + l % NW) : -1;
        // 16-byte granule j = l % 5 of: 0..4 yD of member g-1, 5..9 yL of member g-1, 10..14 t of member g+1
        const int grp = l / 5, j = l - 5 * grp;
        const bool have = grp == 2 ? g < G - 1 : (grp < 2 && g > 0);
        const int src_m = grp == 2 ? g + 1 : g - 1;
        const int src_w = grp == 0 ? LPBC_W_YD : grp == 1 ? LPBC_W_YL : LPBC_W_T;
        tab[64] = l < 15 && have ? 8 * (src_m * LPBC_WG_WORDS + src_w + 2 * j) : -1;
        tab[128] = (grp == 0 ? L::YD : grp == 1 ? L::YL : L::YT + KL * NS) + 3 * j;
    }
    // ---- are all members of this cluster on one XCD?  (They are meant to be, see above, but workgroup -> XCD placement is not a
    //      contract.)  Every member publishes its XCC id write-through; everybody compares.  If so, the hand-offs of this launch
    //      use L2-resident stores; if not — or if a peer does not answer — write-through ones (and a dead peer shows up as a
    //      timeout of the first real hand-off). ----
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
        // ---- next trajectory of this cluster.  The first one is its own index; further ones the leader draws from the queue
        //      (which therefore starts at the number of clusters) and hands to its peers.  A call that fits the chip in one go
        //      — the single-trajectory MPC case — never touches the queue. ----
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
        const int b = reinterpret_cast<const int*>(bc)[2];
        if (bc[1] != 0.f || b >= ca.batch) break;
        // Per-trajectory pointers are read from the kernel-argument segment where they are used (opaque copy of its address per
        // use site): kept in SGPRs for the whole kernel they push the scalar register file over its 102 — the compiler then
        // parks SGPRs in VGPR lanes (two fewer VGPRs for the matrix, v_readlane traffic inside the PCG loop).
        kargp_t k_in = kp;
        asm volatile("" : "+s"(k_in));
        const float* gam = k_in->p.gamma + (size_t)b * vstride;
        const float* lam_in = k_in->p.lambda + (size_t)b * vstride;
        // Per-lane addresses of the staging and write-back code are invariant across trajectories; hoisted out of this loop they
        // are live across the whole PCG loop, where there is not one free register (22 dwords spilled, reloads inside the PCG
        // loop: -6 %).  An opaque copy of the thread index per trajectory keeps them where they are used.
        int t_st = tid, i_st = i;
        asm volatile("" : "+v"(t_st), "+v"(i_st));
        {
            const rsrc_t M = make_rsrc((isP ? static_cast<const float*>(k_in->p.Pinv) : static_cast<const float*>(k_in->p.S)) + (size_t)b * mstride,
                                       (uint32_t)(mstride * sizeof(float)));
            const uint32_t off = valid ? (uint32_t)((k0 + i_st) * 3 + (isL ? 0 : 1)) * (BLK4 * 16u) : OOB_OFF;
#pragma unroll
            for (int c = 0; c < BLK4; ++c) m4[c] = buf_load4<false>(M, off + 16u * c);
        }
        // ---- stage: parts <- 0 (yL of the replica slot: -0, so that (s + yL) + t keeps the bits of s), p <- lambda0, r <- gamma
        //      for the own knots AND the replica (both complete in global memory), lambda <- lambda0 ----
        for (int e = t_st; e < 6 * L::VS; e += NTHR) lds[e] = 0.f;
        lds_barrier();
        for (int e = t_st + (g == 0 ? NS : 0); e < (KL + 1) * NS; e += NTHR) {
            const int ge = (k0 - 1) * NS + e;               // element of the global [N][14] vector
            const float l0 = lam_in[ge];
            lds[L::XP + e] = l0;
            lds[L::LAM + e] = l0;
            lds[L::XR + e] = gam[ge];
        }
        lds_barrier();

        // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
        if (!isP) pass(L::XP, SlotV{}); else idle_partial(SlotV{});
        (void)exchange(SlotV{}, true);
        if (ok0) xr2[e0] = xr2[e0] - ((yD2[e0] + yL2[e0]) + yT2[e0]);
        if (ok1) xr2[e1] = xr2[e1] - ((yD2[e1] + yL2[e1]) + yT2[e1]);
        lds_barrier();
        if (isP && wave_on) pass(L::XR, SlotE{}); else idle_partial(SlotE{});
        float eta = exchange(SlotE{}, p3);
        if (ok0) xp2[e0] = p3 ? (yD2[e0] + yL2[e0]) + yT2[e0] : yD2[e0];
        if (ok1) xp2[e1] = p3 ? (yD2[e1] + yL2[e1]) + yT2[e1] : yD2[e1];
        lds_barrier();
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): see pcg_lpb_kernel

        uint32_t iters = 0;
        uint32_t max_iter_exit = 1;
        if (failed) {
            iters = 0xFFFFFFFFu; max_iter_exit = 2;
        } else if (fabsf(eta) < a.exit_tol) {
            max_iter_exit = 0;
        } else {
            for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
                prof_on = cl == 0 && g == 0 && it == 20;
                prof_base = 1;
#endif
                MPCG_STAMP(0);
                // upsilon = S p ; v = p . upsilon
                if (!isP) pass(L::XP, SlotV{}); else idle_partial(SlotV{});
                const float alpha = eta / exchange(SlotV{}, true);
                {
                    const f2 d0 = yD2[e0], l0 = yL2[e0], t0 = yT2[e0], r0 = xr2[e0];
                    const f2 d1 = yD2[e1], l1 = yL2[e1], t1 = yT2[e1], r1 = xr2[e1];
                    if (ok0) xr2[e0] = r0 - alpha * ((d0 + l0) + t0);
                    if (ok1) xr2[e1] = r1 - alpha * ((d1 + l1) + t1);
                }
                lds_barrier();
                MPCG_STAMP(6);
#ifdef MPCG_PROF
                prof_base = 7;
#endif
                // r~ = Pinv r ; eta' = r . r~          | S waves: lambda += alpha p (own knots)
                if (isP) {
                    if (wave_on) pass(L::XR, SlotE{}); else idle_partial(SlotE{});
                } else {
                    idle_partial(SlotE{});
                    for (int e = NS / 2 + tid; e < e_hi; e += NTHR / 2) lam2[e] = lam2[e] + alpha * xp2[e];
                }
                const float eta_new = exchange(SlotE{}, p3);
                if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; break; }
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
                {
                    f2 rt0 = yD2[e0], rt1 = yD2[e1];
                    const f2 p0 = xp2[e0], p1 = xp2[e1];
                    if (p3) {
                        const f2 l0 = yL2[e0], t0 = yT2[e0], l1 = yL2[e1], t1 = yT2[e1];
                        rt0 = (rt0 + l0) + t0;
                        rt1 = (rt1 + l1) + t1;
                    }
                    const float beta = eta_new / eta;
                    if (ok0) xp2[e0] = rt0 + beta * p0;
                    if (ok1) xp2[e1] = rt1 + beta * p1;
                    eta = eta_new;
                }
                lds_barrier();
                MPCG_STAMP(12);
            }
        }

        // ---- write back own knots (a member that gave up leaves lambda alone and flags the trajectory for the fix-up launch) ----
        kargp_t k_out = kp;
        asm volatile("" : "+s"(k_out));
        if (failed) {                                      // this trajectory's count stays short of G: the fix-up launch re-solves it
            if (tid == 0) { k_out->p.iters[b] = 0xFFFFFFFFu; k_out->p.max_iter_exit[b] = 2; }      // (what the caller sees with "cluster_fixup" = 0)
            break;
        }
        {
            int t_wb = tid;
            asm volatile("" : "+v"(t_wb));
            for (int e = t_wb; e < KL * NS; e += NTHR) {
                const size_t ge = (size_t)b * vstride + (size_t)k0 * NS + e;
                k_out->p.lambda[ge] = lds[L::LAM + NS + e];
                if (k_out->p.r_out) k_out->p.r_out[ge] = lds[L::XR + NS + e];
                if (k_out->p.p_out) k_out->p.p_out[ge] = lds[L::XP + NS + e];
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
