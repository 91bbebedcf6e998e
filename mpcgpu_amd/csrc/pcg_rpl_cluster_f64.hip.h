// pcg_rpl_cluster_f64.hip.h — round 5: linsys_t = double (USE_DOUBLES, include/common/settings.cuh:41-49) beyond N = 32 WITHOUT streaming the
// matrices every iteration: the row-per-lane kernel (pcg_rpl.hip.h) across G = ceil(N / 32) CUs of one XCD per trajectory.
//
// Why this kernel and not a double twin of the lane-pair kernels: full block rows of S and Pinv in double are 2 x 42 x 8 B per matrix row —
// 32 knots fill a CU's register file (pcg_rpl_kernel_f64, N <= 32: 141 M it/s); the lower-triangle lane-pair mapping would hold 64 knots per CU
// but is 1,200 lines of float-specific code (packed FMAs, quad_perm merges).  The row-per-lane body is 200 lines, templated on the element
// type, and its data flow across a member boundary is small: what a pass needs from outside is entry i of the operand at the two neighbouring
// knots, which the reader REBUILDS from two published vectors (x_nb = fma(c, B[nb], A[nb]) — the owner's own operation, hence its bits).
// So a member publishes, after each of the two passes of an iteration, the two vectors of its first and last knot (2 x 14 doubles to each
// side) and its wave partials; ONE hand-off per pass, as in pcg_lpkc_kernel, whose machinery this kernel reuses unchanged: epoch-tagged
// 16-byte granules {value, tag} published straight from registers into the XCD's L2 (members pinned to one XCD, verified at start-up;
// write-through otherwise), two alternating exchange slots, persistent clusters drawing trajectories from a queue, bounded spins +
// completion counts + a fix-up launch (here: the streaming kernel) for a cluster that could not make progress.
// Reads all three block columns (no symmetry contract).  Same PCG, same exit rule, same outputs as every other kernel.
#pragma once
#include "pcg_rpl.hip.h"
#include "pcg_lpk_cluster.hip.h"

namespace mpcg {

// Cells of one member, u64 words: two exchange slots of 128 words — [0, 16) NW = 8 wave partials, [16, 72) the group for the LEFT neighbour
// (vector 0 / vector 1 x 14 entries of the first own knot), [72, 128) the group for the RIGHT neighbour (last own knot); every granule 16 bytes
// {value lo, value hi, tag, 0} — then {sequence number, trajectory} of the cluster's current trajectory (leader) and {1, XCC id}.
constexpr int RPLC_WG_WORDS = 272;
constexpr int RPLC_SLOT = 128, RPLC_W_L = 16, RPLC_W_R = 72;
constexpr int RPLC_SLOT_T = 256, RPLC_SLOT_X = 258;
constexpr int RPLC_NW = 8, RPLC_KMAX = 32;          // wavefronts per member; knots per member (four per wavefront)
constexpr int RPLC_MAX_G = 8;                       // G x NW <= 64 partials polled by one wavefront

struct ClusterArgs64 {
    PcgArgs64 p;
    unsigned long long* scratch;         // [clusters * G][RPLC_WG_WORDS] hand-off cells, zeroed before the launch
    unsigned long long* fail_flags;      // [batch][CL_FLAG_STRIDE], zeroed before the launch: members that finished the trajectory
    unsigned long long* queue;           // next trajectory to hand out (zeroed before the launch)
    int G, batch, clusters, l2_handoff;
    int test_fail = 0;                   // tests only ("cluster_test_fail"): the last member of cluster 0 gives up at the write-back of its first trajectory
};

// LDS (doubles): six vectors [KL + 2][14] with a halo knot either side | 64 partials of the cluster | {timeout flag, trajectory index, same-XCD}
__host__ __device__ constexpr size_t pcg_rplc_lds_doubles() { return 6 * r4((size_t)(RPLC_KMAX + 2) * NS) + 64 + 8; }

template <bool PC3>
__global__ __launch_bounds__(RPLC_NW * 64, 2) void pcg_rplc_f64_kernel(ClusterArgs64 ca) {
    typedef double real;
    constexpr int NW = RPLC_NW, NTHR = NW * 64, PW = PC3 ? 42 : 14;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    real* lds = reinterpret_cast<real*>(lds_raw);
    const PcgArgs64& a = ca.p;
    const int N = a.N, G = ca.G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // members of a cluster share an XCD (pcg_lpkc_kernel): workgroup b = 8 j + x holds member j % G of cluster 8 (j / G) + x
    const unsigned nclusters = (unsigned)ca.clusters;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int g = jx % G;
    const int cl = (jx / G) * 8 + xcd;
    if ((unsigned)cl >= nclusters) return;
    const int k0 = (int)(((long)g * N) / G), k1 = (int)(((long)(g + 1) * N) / G);
    const int KL = k1 - k0;                             // own knots (launcher: 1 <= KL <= 32)
    constexpr int VS = (int)r4((size_t)(RPLC_KMAX + 2) * NS);
    real* xp0 = lds;                                    // local knot kl at (kl + 1) * NS; slot 0 / KL + 1 = the neighbours' boundary knots
    real* xp1 = lds + VS;
    real* xr0 = lds + 2 * VS;
    real* xr1 = lds + 3 * VS;
    real* xt = lds + 4 * VS;
    real* xu = lds + 5 * VS;
    real* red = lds + 6 * VS;                           // [0], [1] the cluster-wide inner product of the current hand-off, by epoch parity
    real* bc = red + 64;                                // [0] timeout flag  [1] trajectory index (as int)  [2] same-XCD (as int)
    gu64* my_words = (gu64*)ca.scratch + ((size_t)cl * G + g) * RPLC_WG_WORDS;
    gu64* cl_words = (gu64*)ca.scratch + (size_t)cl * G * RPLC_WG_WORDS;
    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;

    const int q = lane >> 4, i = lane & 15;
    const int ii = i < NS ? i : NS - 1;                 // (lanes 14, 15 of a row shadow row 13 with zero matrices)
    const int kl = w * 4 + q;                           // local knot of this lane's row
    const bool act = kl < KL && i < NS;
    const int klc = kl < KL ? kl : KL - 1;
    const int k = k0 + klc;                             // global knot
    const bool first = act && klc == 0 && g > 0, last = act && klc == KL - 1 && g < G - 1;

    bool same_xcd = false;
    unsigned epoch = 0, seq = 0;
    bool failed = false;

    // ---- are all members of this cluster on one XCD? ----
    if (tid == 0) { bc[0] = 0.0; }
    if (w == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        if (lane == 0) granule_store<RPLC_SLOT_X>(my_words, 0u, (1ull << 32) | xcc);
        unsigned long long x = 0;
        unsigned spins = 0;
        bool ok;
        do {
            ok = true;
            if (lane < G) {
                x = granule_load<RPLC_SLOT_X>(cl_words, 8u * (unsigned)(lane * RPLC_WG_WORDS));
                ok = (unsigned)(x >> 32) == 1u;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
        } while (++spins < (CL_SPIN_LIMIT >> 4));
        const bool all_same = __all(lane >= G || ((unsigned)(x >> 32) == 1u && (unsigned)x == xcc));
        if (lane == 0) reinterpret_cast<int*>(bc)[4] = all_same ? 1 : 0;
    }
    lds_barrier();
    same_xcd = reinterpret_cast<const int*>(bc)[4] != 0 && ca.l2_handoff != 0;

    // One hand-off: every wave publishes its partial, the rows of the first / last own knot publish their entries of two vectors (va -> the
    // neighbour's halo of bufA, vb -> of bufB); wave 0 polls the cluster's partials and the neighbours' groups, drops the halo entries into the
    // local vectors and folds the partials into ONE value; ONE barrier.
    auto exchange = [&](real* bufA, real va, real* bufB, real vb, real wave_part) -> real {
        ++epoch;
        const unsigned sb = (epoch & 1u) * (unsigned)RPLC_SLOT;           // word offset of this hand-off's slot
        auto gran = [&](real v) -> f4 {
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
            return f4{__builtin_bit_cast(float, (unsigned)bits), __builtin_bit_cast(float, (unsigned)(bits >> 32)), __builtin_bit_cast(float, epoch), 0.f};
        };
        auto put = [&](unsigned word, real v) {
            if (same_xcd) granule_store16<0, true>(my_words, 8u * (sb + word), gran(v));
            else granule_store16<0, false>(my_words, 8u * (sb + word), gran(v));
        };
        if (lane == 0) put(2u * (unsigned)w, wave_part);
        if (first) { put(RPLC_W_L + 2u * (unsigned)ii, va); put(RPLC_W_L + 2u * (unsigned)(14 + ii), vb); }
        if (last) { put(RPLC_W_R + 2u * (unsigned)ii, va); put(RPLC_W_R + 2u * (unsigned)(14 + ii), vb); }
        if (w == 0) {
            // lane l < G NW: wave partial l (member l / NW, wave l % NW).  Lanes 0..27: entry l of the LEFT member's group for its right
            // neighbour; lanes 32..59: entry l - 32 of the RIGHT member's group for its left neighbour (a second load of the same lanes).
            const bool wantp = lane < G * NW;
            const unsigned pbyte = 8u * ((unsigned)(lane / NW) * RPLC_WG_WORDS + sb + 2u * (unsigned)(lane % NW));
            const int e = lane & 31;
            const bool fromL = lane < 32;
            const bool wantv = e < 28 && (fromL ? g > 0 : g < G - 1);
            const unsigned vbyte = 8u * ((unsigned)(fromL ? g - 1 : g + 1) * RPLC_WG_WORDS + sb + (fromL ? RPLC_W_R : RPLC_W_L) + 2u * (unsigned)e);
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 xp_ = {0u, 0u, 0u, 0u}, xv_ = {0u, 0u, 0u, 0u};
            unsigned spins = 0;
            const unsigned pb_ = wantp ? pbyte : 0u, vb_ = wantv ? vbyte : 0u;
            bool ok;
            do {
                asm volatile("s_nop 4\n\t"
                             "global_load_dwordx4 %0, %2, %4 sc1\n\t"
                             "global_load_dwordx4 %1, %3, %4 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(xp_), "=&v"(xv_) : "v"(pb_), "v"(vb_), "s"(cl_words) : "memory");
                ok = (!wantp || xp_.z == epoch) && (!wantv || xv_.z == epoch);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < CL_SPIN_LIMIT);
            const real halo = __builtin_bit_cast(real, ((unsigned long long)xv_.y << 32) | (unsigned long long)xv_.x);
            const real partial = __builtin_bit_cast(real, ((unsigned long long)xp_.y << 32) | (unsigned long long)xp_.x);
            if (wantv) {
                real* dst = e < 14 ? bufA : bufB;
                dst[(fromL ? 0 : KL + 1) * NS + (e < 14 ? e : e - 14)] = halo;
            }
            // the cluster-wide inner product: the G NW partials sit in lanes 0 .. G NW - 1 of this wavefront — folded here, in one fixed order that is
            // the same in every member (lane l = member l / NW, wave l % NW), one value for the workgroup
            const real tot_ = rpl_wave_fold(wantp ? partial : real(0));
            if (lane == 0) { red[epoch & 1u] = tot_; if (spins >= CL_SPIN_LIMIT) bc[0] = 1.0; }
#ifdef RPLC_DEBUG
            if (spins >= CL_SPIN_LIMIT) {
                double* d = ca.p.lambda + g * 256 + lane * 4;
                d[0] = (double)xp_.z; d[1] = (double)xv_.z;
                d[2] = (double)epoch; d[3] = (wantp ? 1.0 : 0.0) + (wantv ? 2.0 : 0.0) + (same_xcd ? 4.0 : 0.0);
            }
#endif
        }
        lds_barrier();
        if (bc[0] != 0.0) failed = true;
        // (two cells by epoch parity.  One would do: the next hand-off's poll cannot complete before every wavefront of this member has published its
        //  next partial, i.e. after it has read this value — the second cell only makes that independence of timing local to these lines)
        const real tot = red[epoch & 1u];
        return tot;
    };

    typedef const __attribute__((address_space(4))) ClusterArgs64* kargp_t;
    const kargp_t kp = (kargp_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        // ---- next trajectory of this cluster: own index first, then the leader draws from the queue (pcg_lpkc_kernel) ----
        ++seq;
        if (seq > 1) {
            if ((unsigned)ca.batch <= nclusters) break;
            if (w == 0) {
                int bn = 0;
                if (g == 0) {
                    if (lane == 0) {
                        bn = (int)nclusters + (int)__hip_atomic_fetch_add(kp->queue, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (same_xcd) granule_store_l2<RPLC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                        else granule_store<RPLC_SLOT_T>(my_words, 0u, ((unsigned long long)seq << 32) | (unsigned)bn);
                    }
                } else {
                    unsigned long long x = 0;
                    unsigned spins = 0;
                    do {
                        x = granule_load<RPLC_SLOT_T>(cl_words, 0u);
                        if ((unsigned)(x >> 32) == seq) break;
                        __builtin_amdgcn_s_sleep(1);
                    } while (++spins < CL_SPIN_LIMIT);
                    bn = (int)(unsigned)x;
                    if (spins >= CL_SPIN_LIMIT && lane == 0) bc[0] = 1.0;
                }
                if (lane == 0) reinterpret_cast<int*>(bc)[2] = bn;
            }
        } else if (tid == 0) {
            reinterpret_cast<int*>(bc)[2] = cl;
        }
        lds_barrier();
        const int b = reinterpret_cast<const int*>(bc)[2];
        if (bc[0] != 0.0 || b >= ca.batch) break;
        const real* Sg = a.S + (size_t)b * mstride;
        const real* Pg = a.Pinv + (size_t)b * mstride;
        const real* gam = a.gamma + (size_t)b * vstride;
        real* lam_g = a.lambda + (size_t)b * vstride;

        // ---- this lane's matrix row (all three blocks), its entries of lambda0 / gamma ----
        real Sm[42], Pm[PW];
        {
            const real* sb_ = Sg + (size_t)k * ROWF + ii;
            const real* pb_ = Pg + (size_t)k * ROWF + ii;
#pragma unroll
            for (int c = 0; c < NS; ++c) {              // element (i, c) of block s: s * 196 + 14 c + i; blocks (0, left), (N-1, right) are never read
                Sm[c] = act && k > 0 ? sb_[NS * c] : real(0);
                Sm[14 + c] = act ? sb_[196 + NS * c] : real(0);
                Sm[28 + c] = act && k < N - 1 ? sb_[392 + NS * c] : real(0);
                if constexpr (PC3) {
                    Pm[c] = act && k > 0 ? pb_[NS * c] : real(0);
                    Pm[14 + c] = act ? pb_[196 + NS * c] : real(0);
                    Pm[28 + c] = act && k < N - 1 ? pb_[392 + NS * c] : real(0);
                } else {
                    Pm[c] = act ? pb_[196 + NS * c] : real(0);
                }
            }
        }
        real lam = act ? lam_g[k * NS + ii] : real(0);
        real p = lam;                                    // operand of the set-up product
        real r = act ? gam[k * NS + ii] : real(0);
        for (int e = tid; e < 6 * VS; e += NTHR) lds[e] = real(0);
        lds_barrier();
        // the neighbours' lambda0 at the two halo knots comes straight from the caller's array
        if (tid < NS && g > 0) xt[tid] = lam_g[(k0 - 1) * NS + tid];
        if (tid >= 64 && tid < 64 + NS && g < G - 1) xt[(KL + 1) * NS + (tid - 64)] = lam_g[k1 * NS + (tid - 64)];

        const int own = (klc + 1) * NS + ii;             // LDS offset of this lane's own entry
        auto publish = [&](real* buf, real x) { if (act) buf[own] = x; };
        // y = (block row) . x ; returns this lane's share of x . y.  The neighbouring knots' entries of the operand are REBUILT from what their
        // owners published: x_nb = fma(cb, B[nb], A[nb]) (pcg_rpl_body::pass3).
        auto pass3 = [&](const real (&M)[42], const real* A, const real* B, real cb, real x, real& y) -> real {
            const int lo = klc * NS + ii, hi = (klc + 2) * NS + ii;
            real xm = fma_t(cb, B[lo], A[lo]);
            real xq = fma_t(cb, B[hi], A[hi]);
            real aL = real(0), aD = real(0), aR = real(0);
            real xo = x;
            asm volatile("s_nop 1" : "+v"(xo), "+v"(xm), "+v"(xq));
            SFor14<0>::run([&](auto cc) {
                constexpr int C = decltype(cc)::value;
                col3_bc_t<C>(aL, aD, aR, xm, xo, xq, M);
            });
            y = (aD + aL) + aR;
            return fma_t(y, x, real(0));
        };
        auto pass1 = [&](const real (&M)[14], real x, real& y) -> real {
            real xo = x, a0 = real(0), a1 = real(0);
            asm volatile("s_nop 1" : "+v"(xo));
            SFor14<0>::run([&](auto cc) {
                constexpr int C = decltype(cc)::value;
                fmac_bc<C>((C & 1) ? a1 : a0, xo, M[C]);
            });
            y = a0 + a1;
            return fma_t(y, x, real(0));
        };
        auto passP = [&](const real* A, const real* B, real cb, real x, real& y) -> real {
            if constexpr (PC3) return pass3(Pm, A, B, cb, x, y);
            else return pass1(Pm, x, y);
        };

        // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
        real y;
        publish(xt, p);
        lds_barrier();
        (void)pass3(Sm, xt, xp0, real(0), p, y);
        r -= y;
        publish(xr0, r);
        (void)exchange(xr0, r, xu, real(0), real(0));                       // r_0 of the neighbours' boundary knots (xu's halo stays zero)
        {
            const real part = rpl_wave_fold(passP(xr0, xu, real(0), r, y));
            p = y;
            publish(xt, y);                                                  // r~_0 ; p_0 = r~_0 + 0 * p_(-1)
            real eta = exchange(xt, y, xp0, real(0), part);
            uint32_t iters = 0;
            uint32_t max_iter_exit = 1;
            real beta = real(0);
            if (failed) {
                iters = 0xFFFFFFFFu; max_iter_exit = 2;
            } else if (fabs_t(eta) < a.exit_tol) {
                max_iter_exit = 0;
            } else {
                for (int it = 0; it < a.max_iter; ++it) {
                    real* xp_old = (it & 1) ? xp1 : xp0;
                    real* xp_new = (it & 1) ? xp0 : xp1;
                    real* xr_old = (it & 1) ? xr1 : xr0;
                    real* xr_new = (it & 1) ? xr0 : xr1;
                    // upsilon = S p ; v = p . upsilon
                    const real pv = rpl_wave_fold(pass3(Sm, xt, xp_old, beta, p, y));
                    publish(xu, y);
                    publish(xp_new, p);
                    const real v = exchange(xu, y, xp_new, p, pv);
                    if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; break; }
                    const real alpha = eta / v;
                    lam = fma_t(alpha, p, lam);
                    r = fma_t(-alpha, y, r);
                    publish(xr_new, r);
                    // r~ = Pinv r ; eta' = r . r~
                    const real pe = rpl_wave_fold(passP(xr_old, xu, -alpha, r, y));
                    publish(xt, y);
                    const real eta_new = exchange(xr_new, r, xt, y, pe);
                    if (failed) { iters = 0xFFFFFFFFu; max_iter_exit = 2; break; }
                    iters = (uint32_t)(it + 1);
                    if (fabs_t(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
                    beta = eta_new / eta;
                    p = fma_t(beta, p, y);
                    eta = eta_new;
                }
            }
            // ---- write back (a member that gave up leaves lambda alone: the trajectory's count stays short of G, the fix-up launch re-solves it
            //      from the handle's copy of lambda0 — its peers may be past their last hand-off and have written their knots) ----
            if (ca.test_fail && cl == 0 && g == G - 1 && seq == 1) failed = true;
            if (failed) {
                if (tid == 0) { kp->p.iters[b] = 0xFFFFFFFFu; kp->p.max_iter_exit[b] = 2; }
                break;
            }
            if (act) {
                const size_t e = (size_t)b * vstride + (size_t)k * NS + ii;
                kp->p.lambda[e] = lam;
                if (kp->p.r_out) kp->p.r_out[e] = r;
                if (kp->p.p_out) kp->p.p_out[e] = p;
            }
            if (tid == 0) {
                if (g == 0) { kp->p.iters[b] = iters; kp->p.max_iter_exit[b] = (uint8_t)max_iter_exit; }
                __hip_atomic_fetch_add(kp->fail_flags + (size_t)b * CL_FLAG_STRIDE, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        lds_barrier();                                      // LDS is restaged for the next trajectory
    }
}

}  // namespace mpcg
