// pcg_lpk.hip.h — "lane pair per knot" PCG kernel for gfx950 (round 3): the register-resident successor of the
// lane-per-block kernel (pcg_lpb.hip.h) for fp32, knot_points <= 128.
//
// What the lane-per-block kernel left on the table (profiles/r02_lpb_ablate.txt): of 5,812 cycles per
// iteration only 46 % were FMA chains; the off-diagonal waves carried 196 packed FMAs per pass against 98 on the diagonal
// waves (two of four SIMDs idle half of every pass); every pass wrote three PART vectors (yD, yL, yT) to LDS that an
// element-wise phase of ALL waves read back, summed and wrote again (914 cycles), between four barriers.
//
// Mapping.  A knot k owns TWO ADJACENT LANES of one wavefront, per matrix; lane h (0/1) of the pair holds the column half
// c = 7h .. 7h+6 of BOTH blocks of block row k that the lower triangle keeps — D_k = M[k,diag] and L_k = M[k,left] — as
// 2 x 49 register pairs (196 VGPRs, as before: a trajectory of 128 knots fills 2 x 256 lanes = the 512 KB register file).
//   direct      acc[rows]  = sum_{c in half} D_k[:,c] x_k[c] + L_k[:,c] x_{k-1}[c]        98 v_pk_fma_f32, ONE accumulator set
//   transposed  z[c]       = sum_rows L_k[row,c] x_k[row],  c in half                       49 v_pk_fma_f32 + 7 adds
// => 147 packed FMAs in EVERY lane of EVERY wave of the pass (was 196 / 98): four S waves and four Pinv waves, one of each
// per SIMD.  The two column-half partial sums of a knot are merged with DPP (quad_perm [1,0,3,2]: the partner lane), so a
// lane pair ends the pass with the complete (D x_k + L x_{k-1}) of its knot IN REGISTERS, split by rows: lane 0 owns row
// pairs P0..P3 (entries 0..7), lane 1 owns P4..P6 (entries 8..13).  Only z — the coupling L_k^T x_k that belongs to knot
// k-1 — goes through LDS (7 floats per lane), and it is read back together with the reduction partials after the barrier
// the inner product needs anyway.  No part vectors, no all-wave element-wise phases: after the barrier the S lanes update
// THEIR 8 entries of r (r -= alpha (acc + z_{k+1})) and publish them, the Pinv lanes likewise p.
// Inner product without the merged vector: x^T M x = sum_k x_k^T (D_k x_k + L_k x_{k-1}) + x_{k-1}^T (L_k^T x_k), and the
// lane has both factors of both terms: acc . x_k (all rows, own columns) + z . x_{k-1}[own columns].
//
// Uniform instruction stream for both lanes of a pair: lane 1 keeps its row pairs in the order P4 P5 P6 P3 P0 P1 P2
// (lane 0: P0 .. P6), so "own pairs" are register slots 0..3 in both, and the partner's copy of own slot s is its slot
// f(s) = (4, 5, 6, 3)[s].  (Lane 1's slot 3 is a duplicate of P3: same bits as lane 0's, written to the same place.)
//
// Reads only the left + diagonal block columns (include/mpcg.h, BLOCK SYMMETRY), like the lane-per-block kernel.
// LDS: four vectors (p, r, lambda, z) of 7 row pairs x (NMAX + 4) knot slots, pair-major (see LpkLds) + wave partials: 29.6 KB at NWR = 2.
// Per iteration: S pass | barrier | alpha, r update (S lanes), lambda update (Pinv lanes) | barrier | Pinv pass | barrier |
// eta', exit test, p update (Pinv lanes) | barrier.
#pragma once
#include "pcg_kernels.hip.h"

namespace mpcg {

// LDS layout of one iterate vector: PAIR-MAJOR, V[q][slot] = entries (2q, 2q+1) of knot slot - 1 as one float2, q = 0..6,
// slot = 0..KN-1 (one zero knot in front: knot k lives in slot k + 1; zero knots behind).  Why: every access of the kernel is then
// bank-conflict-free (MI355X_MICROARCH.md §LDS: ds_read_b64 is served in two groups of 32 lanes over 64 banks, ds_read_b32 over 32) —
//   * a lane pair reads row pairs q and q + 4 of the SAME knot (slot orders P0.. / P4..): KN = 4 (mod 8) puts them 32 banks apart;
//   * consecutive knots of one pair are consecutive float2: 16 knots x 2 lanes of a group cover the 64 banks exactly once;
//   * the per-column scalars x[7h + j] of the two lanes of a knot have opposite parity (7 is odd): distinct banks.
// The first version used knot-major [knot][16 floats]: 8- to 16-way conflicts on every access, the LDS pipe was the bottleneck
// (profiles/r03_lpk_phases.txt: 1,100-1,300 ticks for the 49-FMA transposed stage, 900 for the 8-entry vector updates).
template <int NWR> struct LpkLds {
    static constexpr int NMAX = 64 * NWR, NW = 4 * NWR;
    static constexpr int KN = NMAX + 4;                        // knot slots per row pair: NMAX + 2 rounded up to 4 (mod 8)
    static_assert(KN % 8 == 4, "row pairs q and q + 4 must sit 32 banks apart");
    static constexpr int VS = 7 * KN * 2;                      // floats per vector
    static constexpr int XP = 0, XR = VS, LAM = 2 * VS, Z = 3 * VS, RED = 4 * VS, TOTAL = RED + NW;     // RED: NW/2 wave partials per inner product
    // float index of entry i of knot k inside a vector
    __host__ __device__ static constexpr int at(int k, int i) { return 2 * ((i >> 1) * KN + k + 1) + (i & 1); }
};
__host__ __device__ constexpr size_t pcg_lpk_lds_floats(int NW) {
    return NW == 4 ? (size_t)LpkLds<1>::TOTAL : (size_t)LpkLds<2>::TOTAL;
}

__device__ __forceinline__ f2 buf_load2(rsrc_t r, uint32_t voff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
    return __builtin_bit_cast(f2, v);
}
// a value every lane holds identically, moved to a scalar register
__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// value of `v` in the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_partner(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int NWR>
__global__ __launch_bounds__(NWR * 256, 2) void pcg_lpk_kernel(PcgArgs a) {
    typedef LpkLds<NWR> L;
    constexpr int NW = 4 * NWR, NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip) return;
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* red_v = lds + L::RED;
    float* red_e = red_v + NW / 2;

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- role of this wave, knot and column half of this lane ----
    const bool isP = w >= 2 * NWR;                         // wave-uniform
    const int wl = w - (isP ? 2 * NWR : 0);                // wave of its matrix: 0 .. 2 NWR - 1
    const int li = 64 * wl + lane;
    const int k = li >> 1, h = li & 1;
    const bool p3 = a.pcols == 3;
    const bool hasL = !isP || p3;                          // wave-uniform: block-Jacobi has no off-diagonal Pinv blocks
    const bool valid = k < N;
    // float indices inside a vector (add the vector's offset; K2 = floats between consecutive row pairs):
    //   register slots 0..2 -> pairs (h ? 4 + s : s): bA + K2 s | slot 3 -> pair 3: b0 + 3 K2 | slots 4..6 -> pairs (h ? s - 4 : s): bB + K2 (s - 4)
    //   own column c = 7h + j:  j = 2i -> bE + K2 i,  j = 2i + 1 -> bO + K2 i       (knot k - 1: subtract 2; knot k + 1: add 2)
    constexpr int KN = L::KN, K2 = 2 * KN;
    const int b0 = 2 * (k + 1);
    const int bA = b0 + (h ? 4 * K2 : 0), bB = b0 + (h ? 0 : 4 * K2);
    const int bE = b0 + (h ? 3 * K2 + 1 : 0), bO = b0 + (h ? 4 * K2 : 1);
    const float w3 = h ? 0.f : 1.f;                        // weight of register slot 3 in sums over a knot's entries (lane 1's is lane 0's duplicate)

    // ---- matrix registers: column half c = 7h + j of D_k and L_k, row pairs in this lane's slot order ----
    f2 Md[7][7], Ml[7][7];                                 // [slot][j]
    {
        const rsrc_t M = make_rsrc((isP ? static_cast<const float*>(a.Pinv) : static_cast<const float*>(a.S)) + (size_t)b * mstride,
                                   (uint32_t)(mstride * sizeof(float)));
        const uint32_t rowb = (uint32_t)k * (ROWF * 4u);
        const bool okD = valid, okL = valid && k > 0 && hasL;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            // slot s holds row pair P_q: lane 0: q = s; lane 1: q = (4, 5, 6, 3, 0, 1, 2)[s].  Byte of (pair q, column 7h + j) inside a
            // block = 56 (7h + j) + 8 q: one lane-variable base per slot, the column as the instruction's immediate offset
            const int q1 = s < 3 ? s + 4 : (s == 3 ? 3 : s - 4);
            const uint32_t bs = rowb + (uint32_t)(NS * 4 * 7) * (uint32_t)h + 8u * (uint32_t)(h ? q1 : s);
            const uint32_t bL = okL ? bs : OOB_OFF, bD = okD ? bs + BLK4 * 16u : OOB_OFF;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                Ml[s][j] = buf_load2(M, bL + (uint32_t)(NS * 4 * j));
                Md[s][j] = buf_load2(M, bD + (uint32_t)(NS * 4 * j));
            }
        }
    }

    // ---- stage vectors: p <- lambda0 (operand of the setup product), lambda <- lambda0, r <- gamma, z and all pads <- 0 ----
    for (int e = tid; e < 4 * L::VS; e += NTHR) lds[e] = 0.f;
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        const float l0 = lam_g[e];
        lds[L::XP + L::at(kk, i)] = l0;
        lds[L::LAM + L::at(kk, i)] = l0;
        lds[L::XR + L::at(kk, i)] = gam[e];
    }
    lds_barrier();

#ifdef MPCG_PROF
    bool prof_on = false;
#endif
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

    // One pass of this wave's matrix over the vector at float offset X.  Returns the lane's OWN four row pairs of
    // D_k x_k + L_k x_{k-1} (complete over both column halves); z = L_k^T x_k goes to Z[k]; the wave's share of x^T M x to red[w].
    struct Own { f2 v[4]; };
    auto pass = [&](int X, float* red, int pb) -> Own {
        MPCG_STAMP(pb + 0);
        const float* x = lds + X;
        // x_k, all rows, in this lane's slot order; x_{k-1}[own columns]
        f2 xk[7];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            xk[s] = *reinterpret_cast<const f2*>(x + bA + K2 * s);
            xk[4 + s] = *reinterpret_cast<const f2*>(x + bB + K2 * s);
        }
        xk[3] = *reinterpret_cast<const f2*>(x + b0 + 3 * K2);
        float xmc[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) xmc[j] = x[((j & 1) ? bO : bE) + K2 * (j >> 1) - 2];
        f2 acc[7];
        float cterm = 0.f;
        if (hasL) {
            // transposed: z[c_j] = sum over row pairs of L[pair][c_j] (.) x_k[pair]; two groups of four / three independent chains
            // (register budget: fourteen chain registers at once spill matrix rows)
            float z[7];
            {
                f2 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = Ml[0][j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = __builtin_elementwise_fma(Ml[s][j], xk[s], t[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = t[j].x + t[j].y;
            }
            {
                f2 t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) t[j] = Ml[0][4 + j] * xk[0];
#pragma unroll
                for (int s = 1; s < 7; ++s)
#pragma unroll
                    for (int j = 0; j < 3; ++j) t[j] = __builtin_elementwise_fma(Ml[s][4 + j], xk[s], t[j]);
#pragma unroll
                for (int j = 0; j < 3; ++j) z[4 + j] = t[j].x + t[j].y;
            }
            if (valid) {
#pragma unroll
                for (int j = 0; j < 7; ++j) lds[L::Z + ((j & 1) ? bO : bE) + K2 * (j >> 1)] = z[j];
            }
            // second copy of the coupling term of the inner product: x_{k-1}^T (L_k^T x_k), own columns
#pragma unroll
            for (int j = 0; j < 7; ++j) cterm = fmaf(z[j], xmc[j], cterm);
        }
        __builtin_amdgcn_sched_barrier(0);             // (register budget: x_k[own columns] is not requested before the transposed chains retire)
        MPCG_STAMP(pb + 1);
        float xkc[7];                                            // x_k[own columns]
#pragma unroll
        for (int j = 0; j < 7; ++j) xkc[j] = x[((j & 1) ? bO : bE) + K2 * (j >> 1)];
        if (hasL) {
            // direct, off-diagonal half: acc = L[:, c_j] x_{k-1}[c_j]
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Ml[s][0] * f2{xmc[0], xmc[0]};
#pragma unroll
            for (int j = 1; j < 7; ++j)
#pragma unroll
                for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Ml[s][j], f2{xmc[j], xmc[j]}, acc[s]);
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][0], f2{xkc[0], xkc[0]}, acc[s]);
        } else {
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = Md[s][0] * f2{xkc[0], xkc[0]};
        }
        // direct, diagonal half
#pragma unroll
        for (int j = 1; j < 7; ++j)
#pragma unroll
            for (int s = 0; s < 7; ++s) acc[s] = __builtin_elementwise_fma(Md[s][j], f2{xkc[j], xkc[j]}, acc[s]);
        MPCG_STAMP(pb + 2);
        // merge the two column halves: own slot s + the partner's slot (4, 5, 6, 3)[s]
        Own o;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f2 oth = acc[s < 3 ? s + 4 : 3];
            o.v[s] = f2{acc[s].x + dpp_partner(oth.x), acc[s].y + dpp_partner(oth.y)};
        }
        // inner product share: x_k . (D x_k + L x_{k-1}) over this lane's OWN rows (lane 1's slot 3 duplicates lane 0's: weight 0) + the coupling copy
        f2 d0 = o.v[0] * xk[0], d1 = o.v[1] * xk[1];
        d0 = __builtin_elementwise_fma(o.v[2], xk[2], d0);
        d1 = __builtin_elementwise_fma(o.v[3], xk[3] * f2{w3, w3}, d1);
        const f2 dd = d0 + d1;
        const float part = wave_fold((dd.x + dd.y) + cterm);
        if (lane == 0) red[wl] = part;
        MPCG_STAMP(pb + 3);
        return o;
    };
    // own entries (slots 0..3) of knot k (+ dk) of the vector at float offset X
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
    // sum of the NW/2 wave partials of one inner product (the waves of one matrix), same order in every thread: deterministic
    auto sum_red = [&](const float* red) -> float {
        if constexpr (NW == 8) {
            const f4 v = *reinterpret_cast<const f4*>(red);
            return ((v.x + v.y) + v.z) + v.w;
        } else {
            const f2 v = *reinterpret_cast<const f2*>(red);
            return v.x + v.y;
        }
    };

    // The S waves and the Pinv waves run the same barrier sequence through two SEPARATE code paths (the role is wave-uniform): written as
    // one path with `if (isP)` around each piece, the pass result `own` is a loop-carried value of the "other" role in the compiler's
    // eyes — eight registers live through both passes, which spill matrix rows.
    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    auto run_role = [&](auto role_tag) {
        constexpr bool P = decltype(role_tag)::value;
        // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
        if constexpr (!P) {
            const Own own = pass(L::XP, red_v, 0);
            lds_barrier();
            const Own zin = load_own(L::Z, 1), r0 = load_own(L::XR, 0);
            Own r1;
#pragma unroll
            for (int s = 0; s < 4; ++s) r1.v[s] = r0.v[s] - (own.v[s] + zin.v[s]);
            store_own(L::XR, r1);
            lds_barrier();
            lds_barrier();
        } else {
            lds_barrier();
            lds_barrier();
            const Own own = pass(L::XR, red_e, 8);
            lds_barrier();
            Own p1 = own;
            if (p3) {
                const Own zin = load_own(L::Z, 1);
#pragma unroll
                for (int s = 0; s < 4; ++s) p1.v[s] = own.v[s] + zin.v[s];
            }
            store_own(L::XP, p1);
        }
        float eta = uniform(sum_red(red_e));                  // (read before the barrier below: the next write of red_e is two barriers away)
        lds_barrier();
        // every matrix load has been consumed on the waves that ran a setup pass; say so, or the compiler keeps vmcnt waits inside the loop
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        if (fabsf(eta) < a.exit_tol) { max_iter_exit = 0; return; }
        for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
            prof_on = b == 0 && it == 20;
#endif
            if constexpr (!P) {
                // upsilon = S p ; v = p . upsilon
                const Own own = pass(L::XP, red_v, 0);
                MPCG_STAMP(4);
                lds_barrier();
                MPCG_STAMP(5);
                // alpha = eta / v ; r -= alpha upsilon (own entries)
                const Own zin = load_own(L::Z, 1), cur = load_own(L::XR, 0);
                const float alpha = uniform(eta / sum_red(red_v));
                Own nw;
#pragma unroll
                for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] - alpha * (own.v[s] + zin.v[s]);
                store_own(L::XR, nw);
                MPCG_STAMP(6);
                lds_barrier();
                MPCG_STAMP(7);
                lds_barrier();                              // (the Pinv pass)
                MPCG_STAMP(13);
                // eta' ; exit test
                const float eta_new = uniform(sum_red(red_e));
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
                eta = eta_new;
                MPCG_STAMP(14);
                lds_barrier();
                MPCG_STAMP(15);
            } else {
                MPCG_STAMP(4);
                lds_barrier();                              // (the S pass)
                MPCG_STAMP(5);
                // alpha ; lambda += alpha p (own entries)
                {
                    const Own pk = load_own(L::XP, 0), cur = load_own(L::LAM, 0);
                    const float alpha = uniform(eta / sum_red(red_v));
                    Own nw;
#pragma unroll
                    for (int s = 0; s < 4; ++s) nw.v[s] = cur.v[s] + alpha * pk.v[s];
                    store_own(L::LAM, nw);
                }
                MPCG_STAMP(6);
                lds_barrier();
                MPCG_STAMP(7);
                // r~ = Pinv r ; eta' = r . r~
                const Own own = pass(L::XR, red_e, 8);
                MPCG_STAMP(12);
                lds_barrier();
                MPCG_STAMP(13);
                // eta' ; exit test ; p = r~ + (eta'/eta) p (own entries)
                Own zin;
                if (p3) zin = load_own(L::Z, 1);
                const Own pold = load_own(L::XP, 0);
                const float eta_new = uniform(sum_red(red_e));
                iters = (uint32_t)(it + 1);
                if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
                const float beta = uniform(eta_new / eta);
                Own pn;
                if (p3) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) pn.v[s] = (own.v[s] + zin.v[s]) + beta * pold.v[s];
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) pn.v[s] = own.v[s] + beta * pold.v[s];
                }
                store_own(L::XP, pn);
                eta = eta_new;
                MPCG_STAMP(14);
                lds_barrier();
                MPCG_STAMP(15);
            }
        }
    };
    if (isP) run_role(std::true_type{}); else run_role(std::false_type{});

    // ---- write back ----
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const int kk = e / NS, i = e - kk * NS;
        lam_g[e] = lds[L::LAM + L::at(kk, i)];
        if (a.r_out) a.r_out[(size_t)b * vstride + e] = lds[L::XR + L::at(kk, i)];
        if (a.p_out) a.p_out[(size_t)b * vstride + e] = lds[L::XP + L::at(kk, i)];
    }
    if (tid == 0) {
        a.iters[b] = iters;
        a.max_iter_exit[b] = (uint8_t)max_iter_exit;
    }
}

}  // namespace mpcg
