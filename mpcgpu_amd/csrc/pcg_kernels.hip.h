// pcg_kernels.hip.h — hand-written CDNA4 (gfx950) kernels for the block-tridiagonal PCG.
//
// Design: ONE workgroup solves ONE trajectory, start to finish, inside one
// launch.  There is no inter-workgroup communication, hence no grid barrier (the reference's
// structure — one block per knot + cooperative grid sync, include/pcg/sqp.cuh:230 — costs 26-100 us
// per sync on this chip).  The iterate vectors live in LDS for the whole solve; S and Pinv are
// streamed from HBM/L2 every iteration with fully coalesced 16-byte loads; wavefront shuffles do
// the per-block-row reductions and the PCG inner products.
//
// Lane mapping for a 14x14 column-major block (196 floats = 49 float4, 16-byte aligned because
// 196*4 = 49*16): float4 #f of a block covers flat elements 4f..4f+3; since lcm(4,14) = 28 the
// (row, column-parity) pattern of a float4 depends only on f mod 7.  Lane l < 49 takes float4
// f = l of each block, i.e. residue r = l % 7 and column pair g = l / 7 (columns 2g, 2g+1):
//     r : rows of column a=2g            rows of column b=2g+1
//     0 : 0 1 2 3
//     1 : 4 5 6 7
//     2 : 8 9 10 11
//     3 : 12 13                           0 1
//     4 :                                 2 3 4 5
//     5 :                                 6 7 8 9
//     6 :                                 10 11 12 13
// so every lane has 4 accumulators with FIXED rows, a wave-level load instruction reads 784
// contiguous bytes, and one block row (3 blocks, 2352 contiguous bytes) is 3 such loads.
// Partial sums are combined with shuffles: over g (lane stride 7: offsets 28, 14, 7; lanes >= 49
// hold zeros), then slot e with slot e+14 (same row, other column parity).  Result: lane q < 4
// holds rows 4q..4q+3 of the block row's 14-vector (lane 3: rows 12, 13, and two zero pads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace mpcg {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int NS = 14;          // state size this specialisation is written for
constexpr int BLK4 = 49;        // float4 per 14x14 block
constexpr int ROW4 = 147;       // float4 per block row (3 blocks)
constexpr int ROWF = 588;       // floats per block row

// Workgroup barrier that waits only for this wave's LDS traffic.  __syncthreads() would also wait
// for vmcnt(0), i.e. drain the matrix prefetch that is deliberately in flight across the barrier.
// All cross-wave data in these kernels goes through LDS, so lgkmcnt(0) + s_barrier is sufficient.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Matrix loads go through a buffer resource (SRD) per trajectory matrix: the hardware bounds check
// returns 0.0 WITHOUT memory traffic for any lane whose offset is >= num_records.  That gives a
// branch-free, fixed count of 3 loads per block row — which is what lets s_waitcnt vmcnt(3) keep the
// next row's loads in flight while this row is consumed — and covers every masked case by sending
// the lane to OOB_OFF: lanes 49..63, the never-written blocks (0,left)/(N-1,right), the off-diagonal
// blocks in block-Jacobi mode and the padded steps of a wave with an odd row count.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t OOB_OFF = 0x40000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}

template <bool NT>
__device__ __forceinline__ f4 buf_load4(rsrc_t r, uint32_t voff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, NT ? 2 : 0);   // aux 2 = nt
    return __builtin_bit_cast(f4, v);
}

struct Rows { f4 m0, m1, m2; };   // this lane's float4 of the left / diagonal / right block

// Issue the 3 loads of block row k (bd layout, matrix starts at byte 0 of the SRD).
// cols: 3 or 1 (wave-uniform); lane_off = lane*16 for lanes < 49, OOB_OFF otherwise.
template <bool NT>
__device__ __forceinline__ Rows load_rows(rsrc_t M, int k, int N, int cols, uint32_t lane_off) {
    const bool valid = k < N;
    const uint32_t row = (uint32_t)k * (ROWF * 4u) + lane_off;
    const uint32_t o0 = (valid && cols == 3 && k > 0) ? row : OOB_OFF;
    const uint32_t o1 = valid ? row + BLK4 * 16u : OOB_OFF;
    const uint32_t o2 = (valid && cols == 3 && k < N - 1) ? row + 2u * BLK4 * 16u : OOB_OFF;
    Rows R;
    R.m0 = buf_load4<NT>(M, o0);
    R.m1 = buf_load4<NT>(M, o1);
    R.m2 = buf_load4<NT>(M, o2);
    return R;
}

// Per-lane constants of the mapping above.
struct LaneMap {
    int g2;        // 2*g : first column of this lane's column pair (clamped for idle lanes)
    bool a01;      // accumulators 0,1 multiply column a (else b)
    bool a23;      // accumulators 2,3 multiply column a (else b)
    __device__ __forceinline__ explicit LaneMap(int lane) {
        const int l = lane < BLK4 ? lane : 0;
        const int r = l % 7;
        g2 = 2 * (l / 7);
        a01 = r <= 3;
        a23 = r <= 2;
    }
};

// acc += block * x  for this lane's float4; xk points at the 14(+2)-vector the block multiplies.
__device__ __forceinline__ void fma_block(f4& acc, const f4 m, const float* xk, const LaneMap& L) {
    const f2 x = *reinterpret_cast<const f2*>(xk + L.g2);   // 8-byte aligned: KS*4 and 2g*4 are
    const float x01 = L.a01 ? x.x : x.y;
    const float x23 = L.a23 ? x.x : x.y;
    acc.x = fmaf(m.x, x01, acc.x);
    acc.y = fmaf(m.y, x01, acc.y);
    acc.z = fmaf(m.z, x23, acc.z);
    acc.w = fmaf(m.w, x23, acc.w);
}

__device__ __forceinline__ f4 shfl_down4(f4 v, int d) {
    f4 o;
    o.x = __shfl_down(v.x, d); o.y = __shfl_down(v.y, d);
    o.z = __shfl_down(v.z, d); o.w = __shfl_down(v.w, d);
    return o;
}

// Combine the 49 lanes' partial sums; valid result in lanes 0..3 (see header comment).
__device__ __forceinline__ f4 reduce_rows(f4 a, int lane) {
    a += shfl_down4(a, 28);
    a += shfl_down4(a, 14);
    a += shfl_down4(a, 7);
    f4 o;
    o.x = a.x + __shfl_down(a.z, 3);
    o.y = a.y + __shfl_down(a.w, 3);
    o.z = a.z + __shfl_down(a.x, 4);
    o.w = a.w + __shfl_down(a.y, 4);
    if (lane == 3) { o.z = 0.f; o.w = 0.f; }   // rows 14,15 do not exist: keep the LDS pads zero
    return o;
}

__device__ __forceinline__ float dot4(f4 a, f4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// ------------------------------------------------------------------------------------------------
// Persistent per-trajectory PCG.  grid = batch, block = NW*64.
//
// Lane mapping of THIS kernel (differs from the SpMV kernel's in where r and g sit in the lane id):
// lane = 8*r + g, r = float4 residue (0..6), g = column pair (0..6); lanes with r == 7 or g == 7
// idle (their loads go to the SRD's out-of-bounds path and return 0).  Lane (r,g) still loads
// float4 #(7g + r) of every block, so a wave instruction still covers one contiguous 784-byte
// block.  With g in the low 3 bits the sum over g is a reduction inside groups of 8 consecutive
// lanes = three v_add_f32 with DPP row_shl:4/2/1 — pure VALU, no LDS crossbar in that part.
// Lane 8r then holds the 4 "slots" e = 4r..4r+3 (e < 28); row i of the result is slot i + slot i+14
// (the two column-parity halves), fetched from lanes 8(r+3) / 8(r+4) with one round of 4
// ds_bpermute.  The summation tree is the same as the SpMV kernel's:
// ((g0+g4)+(g2+g6)) + ((g1+g5)+g3), then half a + half b.
// Forming the rows BEFORE the inner products matters numerically: the halves cancel, and dotting
// the un-combined slots (tried) made fp32 CG drift ~10x faster.
//
// LDS (floats, each region rounded to 4): xp[(N+2)*14] p with a zero knot either side |
//   xr[(N+2)*14] r likewise | lam[N*14] | tmp[N*14] upsilon, then r~ | red[2*NW] |
//   matrix cache: per wave, per matrix, RL rows x 3 blocks x 49 float4.
// All vector accesses are 8-byte (float2): 14 floats = 56 B keeps every knot 8-byte aligned.
// ------------------------------------------------------------------------------------------------

__host__ __device__ constexpr size_t r4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t pcg_lds_floats(int N, int NW) {
    return 2 * r4((size_t)(N + 2) * NS) + 2 * r4((size_t)N * NS) + r4(2 * (size_t)NW);
}
// LDS matrix cache: per wave, per matrix, RL rows of 3 blocks of 49 lane-private float4
__host__ __device__ constexpr size_t pcg_lds_cache_floats(int NW, int RL) {
    return (size_t)NW * 2 * RL * 3 * BLK4 * 4;
}

struct PcgArgs {
    const float* S; const float* Pinv; const float* gamma; float* lambda;
    float* r_out; float* p_out;            // optional [batch][N][n] (may be null)
    uint32_t* iters; uint8_t* max_iter_exit;
    int N; int max_iter; float exit_tol; int pcols;   // pcols: 3 = SS, 1 = block-Jacobi
    int lds_rows;                          // RL: block rows per matrix per wave cached in LDS
};

// sum over the 8 consecutive lanes of each group; valid in the group's lane 0.
// Order: ((g0+g4)+(g2+g6)) + ((g1+g5)+(g3+g7)), g7 == 0.
// Twelve v_add_f32 with a DPP row_shl operand (lane i adds lane i+n of its 16-lane row; lanes whose
// source falls outside the row add 0).  Written as ONE asm statement because hipcc otherwise emits
// v_mov_b32_dpp + v_pk_add_f32 pairs (VOP3P cannot carry DPP) — 18 instructions + hazard nops.
// Hazards (a VALU write followed by a DPP read of the same VGPR needs 2 wait states): the leading
// s_nop covers values written just before the statement; inside, every DPP read is 4 instructions
// behind its producer.
__device__ __forceinline__ f4 reduce_g(f4 a) {
    float x = a.x, y = a.y, z = a.z, w = a.w;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    return f4{x, y, z, w};
}

// RR = block rows per matrix per wave held in REGISTERS for the whole solve (loaded once), then
// a.lds_rows rows per matrix per wave held in LDS, the remaining rows streamed every iteration.
// Wave w owns rows k = w + NW*t; t < RR: registers, RR <= t < RR+RL: LDS, t >= RR+RL: stream.
template <int NW, int RR, bool NT>
__global__ __launch_bounds__(NW * 64, (RR == 0 ? 4 : NW / 4)) void pcg_traj_kernel(PcgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    constexpr int NTHR = NW * 64;

    float* xp = lds;                                   // knot j at xp + (j+1)*NS
    float* xr = xp + r4((size_t)(N + 2) * NS);
    float* lam = xr + r4((size_t)(N + 2) * NS);        // knot j at lam + j*NS
    float* tmp = lam + r4((size_t)N * NS);             // knot j at tmp + j*NS
    float* red_v = tmp + r4((size_t)N * NS);
    float* red_e = red_v + NW;
    f4* mc_base = reinterpret_cast<f4*>(red_v + r4(2 * (size_t)NW));

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const rsrc_t rS = make_rsrc(a.S + (size_t)b * mstride, (uint32_t)(mstride * sizeof(float)));
    const rsrc_t rP = make_rsrc(a.Pinv + (size_t)b * mstride, (uint32_t)(mstride * sizeof(float)));
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- lane roles ----
    const int lr = lane >> 3, lg = lane & 7;
    const bool active = lr < 7 && lg < 7;
    const int f_idx = active ? 7 * lg + lr : 0;        // which float4 of a block this lane owns
    const uint32_t lane_off = active ? (uint32_t)f_idx * 16u : OOB_OFF;
    // accumulators 0,1 multiply column 2g (residues r <= 3) or 2g+1; accumulators 2,3 column 2g (r <= 2) or 2g+1
    const int c01 = active ? 2 * lg + (lr <= 3 ? 0 : 1) : 0;
    const int c23 = active ? 2 * lg + (lr <= 2 ? 0 : 1) : 0;
    const bool head = lg == 0 && lr < 4;               // lanes 0, 8, 16, 24 end up with rows 4r..4r+3

    // Block rows owned by this wave: k = w + NW*t, t < T.
    const int T = max(0, (N - w + NW - 1) / NW);
    const int RL = a.lds_rows;
    const int t0s = min(T, RR + RL);                 // first streamed row index
    // streamed step count, padded to a multiple of 4 so the four register buffers (rowA..rowD) keep
    // fixed roles across passes; a padded step has k >= N, loads nothing (OOB) and computes nothing.
    const int TS = ((T - t0s) + 3) & ~3;

    // ---- resident rows: registers ----
    Rows regS[RR > 0 ? RR : 1], regP[RR > 0 ? RR : 1];
#pragma unroll
    for (int t = 0; t < RR; ++t) {
        regS[t] = load_rows<NT>(rS, w + NW * t, N, 3, lane_off);
        regP[t] = load_rows<NT>(rP, w + NW * t, N, a.pcols, lane_off);
    }
    // ---- resident rows: LDS cache (lane-private float4 slots, filled once) ----
    f4* mc = mc_base + (size_t)w * 2 * RL * 3 * BLK4;
    for (int j = 0; j < RL; ++j) {
        const int k = w + NW * (RR + j);
        const Rows a0 = load_rows<NT>(rS, k, N, 3, lane_off);
        const Rows a1 = load_rows<NT>(rP, k, N, a.pcols, lane_off);
        if (active) {
            f4* d0 = mc + (size_t)(j * 3) * BLK4 + f_idx;
            f4* d1 = mc + (size_t)((RL + j) * 3) * BLK4 + f_idx;
            d0[0] = a0.m0; d0[BLK4] = a0.m1; d0[2 * BLK4] = a0.m2;
            d1[0] = a1.m0; d1[BLK4] = a1.m1; d1[2 * BLK4] = a1.m2;
        }
    }

    // ---- matrix stream: S rows, Pinv rows, S rows, ... always one block row ahead of use, and
    //      NOT drained at workgroup barriers (lds_barrier) ----
    int st_t = 0, st_pass = 0;             // position of the NEXT row to load
    auto load_next = [&]() -> Rows {
        const int k = w + NW * (t0s + st_t);
        Rows R = st_pass ? load_rows<NT>(rP, k, N, a.pcols, lane_off) : load_rows<NT>(rS, k, N, 3, lane_off);
        if (++st_t >= TS) { st_t = 0; st_pass ^= 1; }
        return R;
    };
    Rows rowA, rowB, rowC, rowD;           // the stream runs two block rows (one pair) ahead of use
    rowA = load_next();                    // (a wave with no streamed rows gets zeros from the OOB path)
    rowB = load_next();

    // ---- stage vectors: xp <- lambda0 (operand of the setup SpMV), lam <- lambda0, xr <- gamma ----
    for (int e = tid; e < (N + 2) * NS; e += NTHR) { xp[e] = 0.f; xr[e] = 0.f; }
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const float l0 = lam_g[e];
        xp[NS + e] = l0;
        lam[e] = l0;
        xr[NS + e] = gam[e];
    }
    lds_barrier();

    // acc += block * x for this lane's float4; xk = the 14-vector the block multiplies
    auto fma_blk = [&](f4& acc, const f4 m, const float* xk) {
        const float x01 = xk[c01];
        const float x23 = xk[c23];
        acc.x = fmaf(m.x, x01, acc.x);
        acc.y = fmaf(m.y, x01, acc.y);
        acc.z = fmaf(m.z, x23, acc.z);
        acc.w = fmaf(m.w, x23, acc.w);
    };
    // One block row in two halves so that two rows can be in flight per wave (their dependency chains
    // LDS read -> FMA -> DPP -> bpermute -> LDS write are ~600 cycles each and otherwise serialise):
    //   begin : x reads, 12 FMAs, DPP reduction over g, issue of the 4 bpermutes, d reads  (branch-free)
    //   finish: halves a+b, tmp[k] = M[k,:] x, part += d[k] . (M[k,:] x)
    struct Pend { f4 h; float b0, b1, b2, b3; f2 da, db; int k; bool valid; };
    auto begin = [&](const Rows& use, int t, const float* xv, const float* dv) -> Pend {
        Pend q;
        const int k = w + NW * t;
        q.valid = k < N;
        q.k = q.valid ? k : 0;                          // rows beyond N are all-zero (OOB loads): any knot will do
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        fma_blk(acc, use.m0, xv + (q.k + 0) * NS);
        fma_blk(acc, use.m1, xv + (q.k + 1) * NS);
        fma_blk(acc, use.m2, xv + (q.k + 2) * NS);
        const f2* d2 = reinterpret_cast<const f2*>(dv + (q.k + 1) * NS + 4 * (lr & 3));
        q.da = d2[0];
        q.db = d2[1];        // lane 24 reads rows 14,15 = next knot's 0,1 (padded array): multiplied by 0 below
        q.h = reduce_g(acc);
        q.b0 = __shfl_down(q.h.z, 24);                  // slot e+14 of slots e = 4r, 4r+1
        q.b1 = __shfl_down(q.h.w, 24);
        q.b2 = __shfl_down(q.h.x, 32);                  // ... of slots 4r+2, 4r+3
        q.b3 = __shfl_down(q.h.y, 32);
        return q;
    };
    auto finish = [&](const Pend& q, float& part) {
        if (head && q.valid) {
            f4 y = {q.h.x + q.b0, q.h.y + q.b1, q.h.z + q.b2, q.h.w + q.b3};
            if (lr == 3) { y.z = 0.f; y.w = 0.f; }                           // rows 14, 15 do not exist
            f2* out = reinterpret_cast<f2*>(tmp + q.k * NS + 4 * lr);         // 56k + 16r bytes
            out[0] = f2{y.x, y.y};
            if (lr < 3) out[1] = f2{y.z, y.w};
            part += fmaf(y.w, q.db.y, fmaf(y.z, q.db.x, fmaf(y.y, q.da.y, y.x * q.da.x)));
        }
    };
    auto lds_row = [&](int mat, int j) -> Rows {
        Rows R;
        const f4* src = mc + (size_t)((mat * RL + j) * 3) * BLK4 + f_idx;
        R.m0 = src[0]; R.m1 = src[BLK4]; R.m2 = src[2 * BLK4];
        if (!active) { const f4 z = {0.f, 0.f, 0.f, 0.f}; R.m0 = z; R.m1 = z; R.m2 = z; }
        return R;
    };
    auto pass = [&](auto which, const float* xv, const float* dv) -> float {
        constexpr int MAT = decltype(which)::value;       // 0: S, 1: Pinv (must alternate, S first)
        float part = 0.f;
        // registers
#pragma unroll
        for (int t = 0; t + 1 < RR; t += 2) {
            const Pend p0 = begin(MAT ? regP[t] : regS[t], t, xv, dv);
            const Pend p1 = begin(MAT ? regP[t + 1] : regS[t + 1], t + 1, xv, dv);
            finish(p0, part);
            finish(p1, part);
        }
        if constexpr (RR & 1) {
            const Pend p0 = begin(MAT ? regP[RR - 1] : regS[RR - 1], RR - 1, xv, dv);
            finish(p0, part);
        }
        // LDS cache
        int j = 0;
        for (; j + 1 < RL; j += 2) {
            const Rows r0 = lds_row(MAT, j), r1 = lds_row(MAT, j + 1);
            const Pend p0 = begin(r0, RR + j, xv, dv);
            const Pend p1 = begin(r1, RR + j + 1, xv, dv);
            finish(p0, part);
            finish(p1, part);
        }
        if (j < RL) {
            const Rows r0 = lds_row(MAT, j);
            const Pend p0 = begin(r0, RR + j, xv, dv);
            finish(p0, part);
        }
        // stream: pair (A,B) is consumed while (C,D) is in flight, and vice versa
        for (int t = 0; t < TS; t += 4) {
            rowC = load_next();
            rowD = load_next();
            {
                const Pend p0 = begin(rowA, t0s + t, xv, dv);
                const Pend p1 = begin(rowB, t0s + t + 1, xv, dv);
                finish(p0, part);
                finish(p1, part);
            }
            rowA = load_next();
            rowB = load_next();
            {
                const Pend p0 = begin(rowC, t0s + t + 2, xv, dv);
                const Pend p1 = begin(rowD, t0s + t + 3, xv, dv);
                finish(p0, part);
                finish(p1, part);
            }
        }
        // heads sit in lanes 0, 8, 16, 24: fold them into lane 0 (once per pass): (r0+r2)+(r1+r3)
        part += __shfl_down(part, 16);
        part += __shfl_down(part, 8);
        return part;                        // lane 0
    };
    using MatS = std::integral_constant<int, 0>;
    using MatP = std::integral_constant<int, 1>;
    auto block_sum = [&](const float* red) -> float {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[i];
        return s;
    };
    // vector items: float2 #e of an [N][14] vector
    const int NV2 = N * (NS / 2);
    f2* xp2 = reinterpret_cast<f2*>(xp + NS);
    f2* xr2 = reinterpret_cast<f2*>(xr + NS);
    f2* lam2 = reinterpret_cast<f2*>(lam);
    const f2* tmp2 = reinterpret_cast<const f2*>(tmp);
    auto tmp_row = [&](int e) -> f2 { return tmp2[e]; };

    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
    (void)pass(MatS{}, xp, xp);
    lds_barrier();
    for (int e = tid; e < NV2; e += NTHR) xr2[e] = xr2[e] - tmp_row(e);
    lds_barrier();
    {
        const float part = pass(MatP{}, xr, xr);
        if (lane == 0) red_e[w] = part;
    }
    lds_barrier();
    float eta = block_sum(red_e);
    for (int e = tid; e < NV2; e += NTHR) xp2[e] = tmp_row(e);
    lds_barrier();

    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    if (fabsf(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
            // upsilon = S p ; v = p . upsilon
            {
                const float part = pass(MatS{}, xp, xp);
                if (lane == 0) red_v[w] = part;
            }
            lds_barrier();
            const float alpha = eta / block_sum(red_v);
            // lambda += alpha p ; r -= alpha upsilon
            for (int e = tid; e < NV2; e += NTHR) {
                lam2[e] = lam2[e] + alpha * xp2[e];
                xr2[e] = xr2[e] - alpha * tmp_row(e);
            }
            lds_barrier();
            // r~ = Pinv r ; eta' = r . r~
            {
                const float part = pass(MatP{}, xr, xr);
                if (lane == 0) red_e[w] = part;
            }
            lds_barrier();
            const float eta_new = block_sum(red_e);
            iters = (uint32_t)(it + 1);
            if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
            const float beta = eta_new / eta;
            // p = r~ + beta p
            for (int e = tid; e < NV2; e += NTHR) xp2[e] = tmp_row(e) + beta * xp2[e];
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

// ------------------------------------------------------------------------------------------------
// Stand-alone batched block-tridiagonal SpMV  y = M x  (roofline kernel).
// One wave per block row, waves stride over the (trajectory, knot) list so that the chip reads
// one contiguous region at a time; x is read from global (L1/L2-resident, 56 B per knot).
// ------------------------------------------------------------------------------------------------
struct SpmvArgs { const float* M; const float* x; float* y; int N; int batch; int cols; };

template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void bt_spmv_kernel(SpmvArgs a) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    const long total = (long)a.batch * N;
    const long gw = (long)blockIdx.x * NW + w, GW = (long)gridDim.x * NW;
    const LaneMap L(lane);
    // One SRD per launch would need > 4 GiB of range at large batch, so the descriptor is rebuilt
    // per task from the (wave-uniform) trajectory index.
    const uint32_t lane_off = lane < BLK4 ? (uint32_t)lane * 16u : OOB_OFF;
    const size_t mstride = (size_t)N * ROWF;
    auto load_task = [&](long q) -> Rows {
        const bool valid = q < total;
        const long qq = valid ? q : 0;
        const int bt = (int)(qq / N);
        const int k = valid ? (int)(qq - (long)bt * N) : N;        // k = N -> all three loads OOB
        const rsrc_t r = make_rsrc(a.M + (size_t)bt * mstride, (uint32_t)(mstride * sizeof(float)));
        return load_rows<NT>(r, k, N, a.cols, lane_off);
    };

    auto compute = [&](const Rows& use, long q) {
        if (q >= total) return;
        const int k = (int)(q % N);
        const float* xk = a.x + (size_t)q * NS;            // x of knot k of this trajectory
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        // a neighbour that does not exist has an all-zero block; point at xk to stay in range
        const float* xl = (k > 0) ? xk - NS : xk;
        const float* xrr = (k < N - 1) ? xk + NS : xk;
        fma_block(acc, use.m0, xl, L);
        fma_block(acc, use.m1, xk, L);
        fma_block(acc, use.m2, xrr, L);
        const f4 y = reduce_rows(acc, lane);
        float* yk = a.y + (size_t)q * NS + 4 * lane;       // 56q + 16*lane bytes: 8-byte aligned
        if (lane < 3) {
            *reinterpret_cast<f2*>(yk) = f2{y.x, y.y};
            *reinterpret_cast<f2*>(yk + 2) = f2{y.z, y.w};
        } else if (lane == 3) {
            *reinterpret_cast<f2*>(yk) = f2{y.x, y.y};
        }
    };
    Rows rowA = load_task(gw), rowB;
    for (long q = gw; q < total; q += 2 * GW) {
        rowB = load_task(q + GW);
        compute(rowA, q);
        rowA = load_task(q + 2 * GW);
        compute(rowB, q + GW);
    }
}

}  // namespace mpcg
