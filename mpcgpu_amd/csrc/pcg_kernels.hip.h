// pcg_kernels.hip.h — hand-written CDNA4 (gfx950) kernels for the block-tridiagonal PCG.
//
// Two kernels, two lane mappings:
//   pcg_traj_kernel  — the solver.  ONE workgroup solves ONE trajectory, start to finish, inside one
//       launch: no inter-workgroup communication, hence no grid barrier (the reference's structure —
//       one block per knot + cooperative grid sync, include/pcg/sqp.cuh:230 — costs 26-100 us per sync
//       on this chip).  Iterate vectors live in LDS; as much of S and Pinv as fits lives in registers
//       and LDS for the whole solve, the rest is re-read every iteration.  Row-pair x block mapping,
//       three block rows per wave step (described at the kernel).
//   bt_spmv_kernel   — stand-alone batched SpMV, a pure HBM stream.  float4 mapping described next
//       (it was also the PCG kernel's first mapping; the solver moved away from it because its
//       7-lane reduction + parity merge cost ~40 VALU instructions per block row).
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
// Lane mapping of THIS kernel (the SpMV kernel above keeps the float4 mapping, which is the better
// one for a pure HBM stream).  A wave step handles a TRIPLE of consecutive block rows k0, k0+1, k0+2
// (k0 = 3*tr).  lane = 21*s + 7*rho + q:
//     s   = block column (0 left, 1 diagonal, 2 right),  rho = row of the triple,  q = row pair;
// lane (s,rho,q) owns rows 2q, 2q+1 of block (k0+rho, s): fourteen float2, one per column
// (element (2q, u) is at 56u + 8q bytes of the block: 8-byte aligned).  63 of 64 lanes work, a block
// row costs 9.33 VGPRs instead of 12, and per triple the arithmetic is 14 packed FMAs (sequential
// over the block's columns, like the textbook loop), two bpermute rounds to add the three blocks of
// a row (s = 0,1,2 sit 21 lanes apart), and 2 FMAs for the inner product: ~9 VALU instructions per
// block row instead of ~40 with the float4 mapping, whose 4-row-slices need a 7-lane reduction and
// a second pass to merge column parities.  The price is a strided wave-level load (nine 56-byte
// segments per instruction), paid only by the block rows that are streamed.
//
// Residency: wave w owns triples tr = w + NW*j.  j < RT: held in REGISTERS for the whole solve;
// RT <= j < RT+LT: held in LDS (lane-private, 112 B per lane and triple); the rest is re-read every
// iteration through two register buffers that run one triple ahead, also across the barriers.
//
// LDS (floats, each region rounded to 4): xp[(N+2)*14] p with a zero knot either side |
//   xr[(N+2)*14] r likewise | lam[N*14] | tmp[N*14] upsilon, then r~ | red[2*NW] |
//   matrix cache: per wave, per matrix, LT triples x 64 lanes x 14 float2.
// All vector accesses are 8-byte (float2): 14 floats = 56 B keeps every knot 8-byte aligned.
// ------------------------------------------------------------------------------------------------

__host__ __device__ constexpr size_t r4(size_t x) { return (x + 3) & ~(size_t)3; }
__host__ __device__ constexpr size_t pcg_lds_floats(int N, int NW) {
    return 2 * r4((size_t)(N + 2) * NS) + 2 * r4((size_t)N * NS) + r4(2 * (size_t)NW);
}
// LDS matrix cache: per wave, per matrix, LT triples of SLOT_LANES lane records x 14 element pairs (esz = 4 or 2).
// Only 63 lanes carry data (lane 63 is idle and aliases lane 62's record): 7,056 B per fp32 triple instead of
// 7,168 — at N=128 the 2 KB this saves over the uniform cache is what lets a third extra slot fit.
constexpr int SLOT_LANES = 63;
__host__ __device__ constexpr size_t pcg_lds_cache_floats(int NW, int LT, int esz = 4) {
    return (size_t)NW * 2 * LT * SLOT_LANES * (7 * esz);
}

struct PcgArgs {
    const void* S; const void* Pinv;       // bd layout, element type = the kernel's MT (float or _Float16)
    const float* gamma; float* lambda;
    float* r_out; float* p_out;            // optional [batch][N][n] (may be null)
    uint32_t* iters; uint8_t* max_iter_exit;
    int N; int max_iter; float exit_tol; int pcols;   // pcols: 3 = SS, 1 = block-Jacobi
    int lds_rows;                          // LT: triples per matrix per wave cached in LDS
    int lds_extra_s, lds_extra_p;          // <.,.,1> kernels: waves 0..x-1 cache one more triple of S / of Pinv in LDS
    // fix-up launches behind a cluster kernel: trajectory b runs only if redo_flags[b * redo_stride] != redo_skip
    // (flag = members of the cluster that finished the trajectory, skip = G)
    const unsigned long long* redo_flags = nullptr;
    int redo_stride = 0;
    unsigned redo_skip = 0;
    unsigned long long* redo_count = nullptr;   // fix-up launches: += 1 per trajectory they re-solve (the handle's "cluster_fixups" counter)
    // fix-up launches: the warm start is read from here ([batch][N][n], the handle's copy of lambda made in front of the cluster launch) — the
    // members of a cluster that DID finish a trajectory have written their knots of `lambda` already.  nullptr: from `lambda` itself.
    const float* lam0 = nullptr;
    // dispatch order (pcg_lpk_kernel, pcg_lpkc_kernel, ...): workgroup / draw q solves trajectory sched_pick(order, q, order_tag) (nullptr: q
    // itself).  See sched_order_kernel.
    const uint32_t* order = nullptr;
    uint32_t order_tag = 0;       // batch of THIS call: the stored permutation is used only if it was made for the same batch
    int esz = 4;                  // bytes per element of S / Pinv: 4 float, 2 _Float16 storage (the register-resident kernels convert at load)
};

// order[0] = the batch the permutation order[1 ..] was computed for (0: none yet).  The tag is checked on the DEVICE, in stream order with
// the kernel that wrote it: whatever the host believed when the call was enqueued (or captured into a graph that has not run yet), a
// workgroup never follows a permutation of another batch size, or an uninitialised one.
__device__ __forceinline__ int sched_pick(const uint32_t* order, int q, uint32_t tag) {
    if (!order) return q;
    return order[0] == tag ? (int)order[1 + q] : q;
}

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Matrix storage type MT: float (the reference's layout, bit for bit) or _Float16 (same bd layout with
// 2-byte elements, produced by f32_to_f16_kernel; arithmetic stays fp32 — v_fma_mix_f32).
template <typename MT> struct MatT;
template <> struct MatT<float> {
    typedef f2 pair;                                  // two consecutive rows of one column
    typedef f4 chunk;                                 // LDS cache granule = two columns
    static __device__ __forceinline__ pair load(rsrc_t r, uint32_t voff) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(f2, (u2)__builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0));
    }
    // acc.{x,y} += m.{x,y} * x
    static __device__ __forceinline__ void fma(f2& acc, pair m, float x) {
        acc = __builtin_elementwise_fma(m, f2{x, x}, acc);      // one v_pk_fma_f32 (x broadcast by op_sel)
    }
};
template <> struct MatT<_Float16> {
    typedef unsigned pair;                            // two halves kept PACKED in one VGPR for the whole solve
    typedef f2 chunk;
    static __device__ __forceinline__ pair load(rsrc_t r, uint32_t voff) {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0);
    }
    // v_fma_mix_f32: f16 source half selected by op_sel, f32 multiplicand and accumulator.  Written as asm
    // because hipcc otherwise hoists the f16->f32 conversions of the loop-invariant resident triples out of
    // the PCG loop and keeps them as floats: twice the registers, the whole point lost.
    static __device__ __forceinline__ void fma(f2& acc, pair m, float x) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.x) : "v"(m), "v"(x));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc.y) : "v"(m), "v"(x));
    }
};

// DPP wave shift: lane i receives the value of lane i+1 (0 beyond lane 63).  VALU-only neighbour exchange —
// no LDS crossbar trip (ds_bpermute) on the critical path of every triple.
__device__ __forceinline__ f2 wave_shl1(f2 v) {
    constexpr int DPP_WAVE_SHL1 = 0x130;
    const float vx = v.x, vy = v.y;     // (scalar temporaries: __builtin_bit_cast on a vector element lvalue reads element 0)
    f2 o;
    o.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vx), DPP_WAVE_SHL1, 0xf, 0xf, true));
    o.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vy), DPP_WAVE_SHL1, 0xf, 0xf, true));
    return o;
}

// Phase timing for tools/prof_phases.py (diagnostic build only, -DMPCG_PROF): workgroup 0 stamps s_memtime at
// the phase boundaries of one iteration, one row of 32 stamps per wave.
#ifdef MPCG_PROF
__device__ long long g_pcg_prof[16 * 32];
// per workgroup: s_memrealtime at entry and exit, HW_ID, XCC_ID (which CU it ran on: the gaps between consecutive workgroups of a CU)
__device__ long long g_wg_prof[4096 * 4];
#define MPCG_WG_STAMP(slot)                                                                                             \
    do {                                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                                    \
            long long t_;                                                                                               \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            g_wg_prof[blockIdx.x * 4 + (slot)] = t_;                                                                    \
            if ((slot) == 0) {                                                                                          \
                unsigned hw_, xcc_;                                                                                     \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                       \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                     \
                g_wg_prof[blockIdx.x * 4 + 2] = hw_;                                                                    \
                g_wg_prof[blockIdx.x * 4 + 3] = xcc_;                                                                   \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
#define MPCG_STAMP(i)                                                                                   \
    do {                                                                                                \
        if (prof_on) {                                                                                  \
            long long t_;                                                                               \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            if (lane == 0) g_pcg_prof[w * 32 + (i)] = t_;                                               \
        }                                                                                               \
    } while (0)
#else
#define MPCG_STAMP(i) do {} while (0)
#define MPCG_WG_STAMP(slot) do {} while (0)
#endif

// RT = triples per matrix per wave held in registers.  SB = register buffers of the stream:
// 2 = ping-pong (one triple ahead), 1 = single buffer refilled as soon as it has been consumed
// (28 VGPRs cheaper: one more resident triple), 0 = no stream at all (launcher guarantees that every
// triple is resident).
template <int NW, int RT, int SB, typename MT>
__global__ __launch_bounds__(NW * 64, (RT == 0 ? 4 : (NW == 4 && RT <= 3 ? 2 : NW / 4))) void pcg_traj_kernel(PcgArgs a) {
    typedef typename MatT<MT>::pair mpair;
    typedef typename MatT<MT>::chunk mchunk;
    struct Trip { mpair m[NS]; };                      // this lane's two rows of its block, one pair per column
    constexpr uint32_t ESZ = sizeof(MT);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int N = a.N;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    constexpr int NTHR = NW * 64;
    if (a.redo_flags && __hip_atomic_load(a.redo_flags + (size_t)b * a.redo_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.redo_skip)
        return;                                        // (uniform) nothing to redo for this trajectory
    if (a.redo_flags && a.redo_count && tid == 0) __hip_atomic_fetch_add(a.redo_count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    float* xp = lds;                                   // knot j at xp + (j+1)*NS
    float* xr = xp + r4((size_t)(N + 2) * NS);
    float* lam = xr + r4((size_t)(N + 2) * NS);        // knot j at lam + j*NS
    float* tmp = lam + r4((size_t)N * NS);             // knot j at tmp + j*NS
    float* red_v = tmp + r4((size_t)N * NS);
    float* red_e = red_v + NW;
    mchunk* mc_base = reinterpret_cast<mchunk*>(red_v + r4(2 * (size_t)NW));

    const size_t mstride = (size_t)N * ROWF, vstride = (size_t)N * NS;
    const rsrc_t rS = make_rsrc(static_cast<const MT*>(a.S) + (size_t)b * mstride, (uint32_t)(mstride * ESZ));
    const rsrc_t rP = make_rsrc(static_cast<const MT*>(a.Pinv) + (size_t)b * mstride, (uint32_t)(mstride * ESZ));
    const float* gam = a.gamma + (size_t)b * vstride;
    float* lam_g = a.lambda + (size_t)b * vstride;

    // ---- lane roles ----
    const bool active = lane < 63;
    // Two lane orders:
    //   streaming kernels (SB > 0): lane = 21 s + 7 rho + q — the seven row pairs of a block column are adjacent
    //       lanes, so each streamed load instruction reads 56 contiguous bytes per block; the three blocks of a row
    //       sit 21 lanes apart and are merged with two ds_bpermute rounds;
    //   all-resident kernels (SB == 0, nothing is loaded inside the PCG loop): lane = 3 (7 rho + q) + s — the
    //       three blocks of a row are ADJACENT lanes and are merged with two DPP wave shifts (VALU only): no LDS
    //       crossbar round trip on the critical path of every triple (N=64: 108 -> 127 M it/s; the same order
    //       makes a streamed triple 1.7x slower, hence the split).
    constexpr bool ADJ = SB == 0;
    const int lg = active ? (ADJ ? lane / 3 : lane % 21) : 0;          // 7*rho + q
    const int ls = active ? (ADJ ? lane - 3 * lg : lane / 21) : 0;     // block column
    const int lrho = lg / 7;                           // row within the triple
    const int lq = lg - 7 * lrho;                      // row pair
    const bool head = active && ls == 0;               // s == 0 lanes receive the finished rows
    const uint32_t lane_byte = (uint32_t)(ls * 196 + lq * 2) * ESZ;   // inside the block row

    // triples owned by this wave: tr = w + NW*j, j < TT
    const int NTR = (N + 2) / 3;
    const int TT = max(0, (NTR - w + NW - 1) / NW);
    const int LT = a.lds_rows;
    // <.,.,1> kernels: the LDS left over after the uniform cache (fewer than 2 NW slots) is handed out one triple
    // at a time, to (wave 0, S), (wave 1, S), ..., then (wave 0, Pinv), ... — the waves that own the most triples
    // first, and S before Pinv so that at least one of the two passes loses its longest stream
    const int LTs = LT + (SB == 1 && w < a.lds_extra_s ? 1 : 0), LTp = LT + (SB == 1 && w < a.lds_extra_p ? 1 : 0);
    const int j0sS = min(TT, RT + LTs), j0sP = min(TT, RT + LTp);   // first streamed triple of S / of Pinv
    // streamed steps (even for the A/B roles of SB == 2, where LTs == LTp)
    const int TSs = SB == 2 ? (((TT - j0sS) + 1) & ~1) : (TT - j0sS);
    const int TSp = SB == 2 ? TSs : (TT - j0sP);

    // byte offset of this lane's (row, block) in the trajectory's matrix, or OOB_OFF when that block
    // must not be read: rows >= N, the never-written blocks (0,left) and (N-1,right), the
    // off-diagonal blocks in block-Jacobi mode, lane 63.
    auto trip_off = [&](int j, int cols) -> uint32_t {
        const int k = 3 * (w + NW * j) + lrho;
        const bool ok = active && j < TT && k < N && !(ls == 0 && k == 0) && !(ls == 2 && k == N - 1) && (cols == 3 || ls == 1);
        return ok ? (uint32_t)k * (ROWF * ESZ) + lane_byte : OOB_OFF;
    };
    auto load_trip = [&](rsrc_t M, int j, int cols) -> Trip {
        const uint32_t off = trip_off(j, cols);
        Trip t;
#pragma unroll
        for (int u = 0; u < NS; ++u) t.m[u] = MatT<MT>::load(M, off + (NS * ESZ) * u);
        return t;
    };

    // ---- resident triples: registers ----
    Trip regS[RT > 0 ? RT : 1], regP[RT > 0 ? RT : 1];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        regS[j] = load_trip(rS, j, 3);
        regP[j] = load_trip(rP, j, a.pcols);
    }
    // ---- resident triples: LDS cache, lane-private records of 7 chunks (chunk = two columns) ----
    const int slane = active ? lane : SLOT_LANES - 1;  // (idle lane 63 reads lane 62's record and writes nothing)
    mchunk* mc = mc_base + (size_t)w * 2 * LT * SLOT_LANES * 7;
    mchunk* mc_x = mc_base + (size_t)NW * 2 * LT * SLOT_LANES * 7;             // extra slots: index = wave (S), NW + wave (Pinv)
    auto slot = [&](int mat, int j) -> mchunk* {
        return j < LT ? mc + ((size_t)(mat * LT + j) * SLOT_LANES + slane) * 7
                      : mc_x + ((size_t)(mat * NW + w) * SLOT_LANES + slane) * 7;
    };
    auto pack2 = [](mpair lo, mpair hi) -> mchunk {
        if constexpr (sizeof(MT) == 4) return mchunk{lo.x, lo.y, hi.x, hi.y};
        else return mchunk{__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};   // (bit pattern only)
    };
    for (int j = 0; j < LTs + LTp; ++j) {              // one triple at a time
        const bool isP = j >= LTs;
        const int jj = isP ? j - LTs : j;
        const Trip t0 = isP ? load_trip(rP, RT + jj, a.pcols) : load_trip(rS, RT + jj, 3);
        mchunk* d0 = slot(isP, jj);
        if (active) {
#pragma unroll
            for (int u = 0; u < 7; ++u) d0[u] = pack2(t0.m[2 * u], t0.m[2 * u + 1]);
        }
    }
    auto lds_trip = [&](int mat, int j) -> Trip {
        const mchunk* src = slot(mat, j);
        Trip t;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const mchunk v = src[u];
            if constexpr (sizeof(MT) == 4) {
                t.m[2 * u] = mpair{v.x, v.y};
                t.m[2 * u + 1] = mpair{v.z, v.w};
            } else {
                // (scalar temporaries on purpose: __builtin_bit_cast applied directly to the vector
                //  element lvalue v.y read element 0 with this compiler)
                const float c0 = v.x, c1 = v.y;
                t.m[2 * u] = __builtin_bit_cast(mpair, c0);
                t.m[2 * u + 1] = __builtin_bit_cast(mpair, c1);
            }
        }
        return t;
    };

    // ---- matrix stream: S triples, Pinv triples, S triples, ... one triple ahead of use, and NOT
    //      drained at workgroup barriers (lds_barrier) ----
    int st_j = 0, st_pass = 0;             // position of the NEXT triple to load
    auto load_next = [&]() -> Trip {                   // (SB == 1: only called when TSs + TSp > 0)
        if constexpr (SB == 1) {
            if ((st_pass ? TSp : TSs) == 0) st_pass ^= 1;   // this wave streams nothing of that matrix
        }
        // ONE load sequence through a selected descriptor — `st_pass ? load_trip(rP…) : load_trip(rS…)` compiled to a
        // speculative first load, a branch and an `s_waitcnt vmcnt(0)` before the other thirteen: every refill was
        // waited for on the spot, a full memory latency per streamed triple
        const bool isP = st_pass != 0;
        const rsrc_t M = isP ? rP : rS;
        const Trip t = load_trip(M, (isP ? j0sP : j0sS) + st_j, isP ? a.pcols : 3);
        if (++st_j >= (isP ? TSp : TSs)) { st_j = 0; st_pass ^= 1; }
        return t;
    };
    Trip bufA, bufB;
    if constexpr (SB > 0) {
        if (SB == 2 || TSs + TSp > 0) bufA = load_next();     // (nothing to stream: zeros from the OOB path)
    }

    // ---- stage vectors: xp <- lambda0 (operand of the setup SpMV), lam <- lambda0, xr <- gamma ----
    for (int e = tid; e < (N + 2) * NS; e += NTHR) { xp[e] = 0.f; xr[e] = 0.f; }
    lds_barrier();
    for (int e = tid; e < N * NS; e += NTHR) {
        const float l0 = a.lam0 ? a.lam0[(size_t)b * vstride + e] : lam_g[e];
        xp[NS + e] = l0;
        lam[e] = l0;
        xr[NS + e] = gam[e];
    }
    lds_barrier();

    // One triple in two halves so that two can be in flight per wave:
    //   begin : 14 x values, 14 packed FMAs, issue of the bpermutes that bring blocks 1 and 2 to the
    //           block-0 lanes, d read                                                   (branch-free)
    //   finish: left + diagonal + right, tmp[k] = M[k,:] x, part += d[k] . (M[k,:] x)
    struct Pend { f2 a0, a1, a2, d; int k; bool valid; };
#ifdef MPCG_PROF
    bool prof_on = false;
#endif
    auto begin = [&](const Trip& t, int j, const float* xv, const float* dv) -> Pend {
        Pend q;
        const int k = 3 * (w + NW * j) + lrho;
        q.valid = k < N;
        q.k = q.valid ? k : 0;                          // rows beyond N are all-zero (OOB loads): any knot will do
        const f2* x2 = reinterpret_cast<const f2*>(xv + (q.k + ls) * NS);   // knot k-1+s of the padded vector
        // two accumulator pairs (even / odd columns) halve the dependent FMA chain: 7 deep instead of 14
        f2 acc = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const f2 x = x2[u];
            MatT<MT>::fma(acc, t.m[2 * u], x.x);
            MatT<MT>::fma(acc1, t.m[2 * u + 1], x.y);
        }
        acc += acc1;
        q.d = *reinterpret_cast<const f2*>(dv + (q.k + 1) * NS + 2 * lq);
        q.a0 = acc;
        if constexpr (ADJ) {
            q.a1 = wave_shl1(acc);                       // block 1's partial, from lane + 1
            q.a2 = wave_shl1(q.a1);                      // block 2's partial, from lane + 2
        } else {
            q.a1.x = __shfl_down(acc.x, 21);
            q.a1.y = __shfl_down(acc.y, 21);
            q.a2.x = __shfl_down(acc.x, 42);
            q.a2.y = __shfl_down(acc.y, 42);
        }
        return q;
    };
    // LEAN: 256-register waves whose resident triples + stream buffer leave fewer than ~60 working registers
    // (<8,3,1,float>: 168 + 28).  One triple in flight instead of two (the second wave of the SIMD fills the latencies),
    // and an LDS-cached triple is consumed chunk by chunk instead of being copied to 28 registers first.
    // Otherwise the compiler spills resident matrix rows and reloads them from scratch in every S pass, in
    // order behind the in-flight stream load: measured 2700 of 17200 cycles per iteration (profiles/
    // r01e_phases_N128.txt).  Same FMA order as the other path: results are bit-identical.
    constexpr bool LEAN = NW >= 8 && (2 * RT + (SB > 0 ? SB : 0)) * 7 * (int)sizeof(MT) > 190;
    auto begin_lds = [&](int mat, int jl, const float* xv, const float* dv) -> Pend {
        Pend q;
        const int k = 3 * (w + NW * (RT + jl)) + lrho;
        q.valid = k < N;
        q.k = q.valid ? k : 0;
        const f2* x2 = reinterpret_cast<const f2*>(xv + (q.k + ls) * NS);
        const mchunk* src = slot(mat, jl);
        f2 acc = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const mchunk v = src[u];
            const f2 x = x2[u];
            mpair m0, m1;
            if constexpr (sizeof(MT) == 4) {
                m0 = mpair{v.x, v.y};
                m1 = mpair{v.z, v.w};
            } else {
                const float c0 = v.x, c1 = v.y;
                m0 = __builtin_bit_cast(mpair, c0);
                m1 = __builtin_bit_cast(mpair, c1);
            }
            MatT<MT>::fma(acc, m0, x.x);
            MatT<MT>::fma(acc1, m1, x.y);
        }
        acc += acc1;
        q.d = *reinterpret_cast<const f2*>(dv + (q.k + 1) * NS + 2 * lq);
        q.a0 = acc;
        if constexpr (ADJ) {
            q.a1 = wave_shl1(acc);                       // block 1's partial, from lane + 1
            q.a2 = wave_shl1(q.a1);                      // block 2's partial, from lane + 2
        } else {
            q.a1.x = __shfl_down(acc.x, 21);
            q.a1.y = __shfl_down(acc.y, 21);
            q.a2.x = __shfl_down(acc.x, 42);
            q.a2.y = __shfl_down(acc.y, 42);
        }
        return q;
    };
    auto finish = [&](const Pend& q, float& part) {
        if (head && q.valid) {
            const f2 y = (q.a0 + q.a1) + q.a2;
            *reinterpret_cast<f2*>(tmp + q.k * NS + 2 * lq) = y;             // 56k + 8q bytes
            part += fmaf(y.y, q.d.y, y.x * q.d.x);
        }
    };
    auto pass = [&](auto which, const float* xv, const float* dv) -> float {
        constexpr int MAT = decltype(which)::value;       // 0: S, 1: Pinv (must alternate, S first)
        float part = 0.f;
        const int LTm = MAT ? LTp : LTs, j0s = MAT ? j0sP : j0sS, TS = MAT ? TSp : TSs;
        MPCG_STAMP(MAT * 8 + 0);
        // registers: two triples in flight per wave, unless the register budget is tight (LEAN)
        if constexpr (LEAN) {
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const Pend p0 = begin(MAT ? regP[j] : regS[j], j, xv, dv);
                finish(p0, part);
            }
        } else {
#pragma unroll
            for (int j = 0; j + 1 < RT; j += 2) {
                const Pend p0 = begin(MAT ? regP[j] : regS[j], j, xv, dv);
                const Pend p1 = begin(MAT ? regP[j + 1] : regS[j + 1], j + 1, xv, dv);
                finish(p0, part);
                finish(p1, part);
            }
            if constexpr (RT & 1) {
                const Pend p0 = begin(MAT ? regP[RT - 1] : regS[RT - 1], RT - 1, xv, dv);
                finish(p0, part);
            }
        }
        MPCG_STAMP(MAT * 8 + 1);
        // LDS cache
        for (int j = 0; j < LTm; ++j) {
            if constexpr (LEAN) {
                const Pend p0 = begin_lds(MAT, j, xv, dv);
                finish(p0, part);
            } else {
                const Trip t0 = lds_trip(MAT, j);
                const Pend p0 = begin(t0, RT + j, xv, dv);
                finish(p0, part);
            }
        }
        MPCG_STAMP(MAT * 8 + 2);
        if constexpr (SB == 2) {
            // stream: A is consumed while B is in flight, and vice versa
            for (int j = 0; j < TS; j += 2) {
                bufB = load_next();
                const Pend p0 = begin(bufA, j0s + j, xv, dv);
                bufA = load_next();
                const Pend p1 = begin(bufB, j0s + j + 1, xv, dv);
                finish(p0, part);
                finish(p1, part);
            }
        } else if constexpr (SB == 1) {
            // stream through one buffer: the refill is issued the moment the FMAs have read it and
            // flies during finish + the next resident work (and the other waves' turns)
            for (int j = 0; j < TS; ++j) {
                const Pend p0 = begin(bufA, j0s + j, xv, dv);
                if (TSs + TSp > 1 || j + 1 < TS) bufA = load_next();   // (a lone streamed triple stays in its buffer)
                finish(p0, part);
            }
        }
        MPCG_STAMP(MAT * 8 + 3);
        // heads are lanes 0..20 (others hold 0): fold into lane 0, once per pass.  Four DPP adds inside
        // each 16-lane row (VALU latency) + one readlane, instead of five dependent ds_bpermute round
        // trips (~100 cycles each, on the critical path of every barrier phase).
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
        if constexpr (ADJ) {                // heads are lanes 0, 3, ..., 60: all four 16-lane rows, in order
            const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 32));
            const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(pb, 48));
            MPCG_STAMP(MAT * 8 + 4);
            return ((part + r1) + r2) + r3;
        }
        MPCG_STAMP(MAT * 8 + 4);
        return part + r1;                   // heads are lanes 0..20: (lanes 0..15) + (lanes 16..20); valid in lane 0
    };
    using MatS = std::integral_constant<int, 0>;
    using MatP = std::integral_constant<int, 1>;
    auto block_sum = [&](const float* red) -> float {      // same order in every thread: deterministic
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i += 4) {
            const f4 v = *reinterpret_cast<const f4*>(red + i);
            s += v.x; s += v.y; s += v.z; s += v.w;
        }
        return s;
    };
    // vector items: float2 #e of an [N][14] vector
    const int NV2 = N * (NS / 2);
    f2* xp2 = reinterpret_cast<f2*>(xp + NS);
    f2* xr2 = reinterpret_cast<f2*>(xr + NS);
    f2* lam2 = reinterpret_cast<f2*>(lam);
    const f2* tmp2 = reinterpret_cast<const f2*>(tmp);

    // ---- setup: r = gamma - S lambda0 ; r~ = Pinv r ; p = r~ ; eta = r . r~ ----
    (void)pass(MatS{}, xp, xp);
    lds_barrier();
    for (int e = tid; e < NV2; e += NTHR) xr2[e] = xr2[e] - tmp2[e];
    lds_barrier();
    {
        const float part = pass(MatP{}, xr, xr);
        if (lane == 0) red_e[w] = part;
    }
    lds_barrier();
    float eta = block_sum(red_e);
    for (int e = tid; e < NV2; e += NTHR) xp2[e] = tmp2[e];
    lds_barrier();

    uint32_t iters = 0;
    uint32_t max_iter_exit = 1;
    if (fabsf(eta) < a.exit_tol) {
        max_iter_exit = 0;
    } else {
        for (int it = 0; it < a.max_iter; ++it) {
#ifdef MPCG_PROF
            prof_on = b == 0 && it == 20;
#endif
            // upsilon = S p ; v = p . upsilon
            {
                const float part = pass(MatS{}, xp, xp);
                if (lane == 0) red_v[w] = part;
            }
            lds_barrier();
            MPCG_STAMP(5);
            const float alpha = eta / block_sum(red_v);
            // lambda += alpha p ; r -= alpha upsilon
            for (int e = tid; e < NV2; e += NTHR) {
                lam2[e] = lam2[e] + alpha * xp2[e];
                xr2[e] = xr2[e] - alpha * tmp2[e];
            }
            MPCG_STAMP(6);
            lds_barrier();
            MPCG_STAMP(7);
            // r~ = Pinv r ; eta' = r . r~
            {
                const float part = pass(MatP{}, xr, xr);
                if (lane == 0) red_e[w] = part;
            }
            lds_barrier();
            MPCG_STAMP(13);
            const float eta_new = block_sum(red_e);
            iters = (uint32_t)(it + 1);
            if (fabsf(eta_new) < a.exit_tol) { max_iter_exit = 0; break; }
            const float beta = eta_new / eta;
            // p = r~ + beta p
            for (int e = tid; e < NV2; e += NTHR) xp2[e] = tmp2[e] + beta * xp2[e];
            eta = eta_new;
            MPCG_STAMP(14);
            lds_barrier();
            MPCG_STAMP(15);
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
// Shared by the clustered kernel (pcg_lpk_cluster.hip.h: G workgroups on G CUs solve ONE trajectory) and its host side: the scratch
// of epoch-tagged hand-off cells and completion flags.  Hand-off = the R2 recipe of cdna_hip_programming.md §6 G16: {epoch, value}
// granules written with one relaxed agent-scope store and polled with relaxed agent-scope loads — the tag is the flag, no fences;
// every spin is bounded.  (Rounds 1-2 had two more clustered kernels on this machinery — a row-triple one here and a lane-per-block one;
// both retired in round 4, HISTORY.md.)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;
// "This cluster gave up" flags: one per trajectory of the launch, each in a 128-byte line of its own at the FRONT of the
// scratch buffer, read by the fix-up launch with agent-scope loads.  They must not share a line with the polled cells: a
// plainly cached copy of such a line (left in some XCD's L2 by the fix-up kernel) made the next graph replay's pollers
// read last run's epochs and time out.
constexpr int CL_FLAG_STRIDE = 16;       // u64 words between flags
// A poll is one sc1 load + s_sleep (25-70 ns): 2^16 polls = 1.5-4.5 ms.  Members of a launch are dispatched within a
// microsecond of each other on a free GPU, so a wait this long means a peer is not resident (another stream holds its
// CU): the member gives up and the host-side fix-up launch re-solves the trajectory with the single-workgroup kernel.
constexpr unsigned CL_SPIN_LIMIT = 1u << 16;

struct ClusterArgs {
    int kl_max;                              // knots of the largest member
    PcgArgs p;
    unsigned long long* scratch;         // [clusters * G][LPBC_WG_WORDS] hand-off cells, zeroed before the launch
    unsigned long long* fail_flags;      // [batch][CL_FLAG_STRIDE], zeroed before the launch: count of members that finished the trajectory
    int G;
    unsigned long long* queue = nullptr; // next trajectory to hand out (zeroed before the launch)
    int batch = 0;                       // trajectories of the call
    int clusters = 0;                    // clusters of the launch (the grid holds 8 ceil(clusters / 8) of them)
    int l2_handoff = 1;                  // 1 = hand-offs through the XCD's L2 when all members of a cluster share an XCD (verified in the kernel)
    int test_fail = 0;                   // tests only ("cluster_test_fail"): the last member of cluster 0 gives up at the write-back of its first trajectory
};

// Zero-fill of the cluster scratch (flags + hand-off cells) in front of every cluster launch.  A kernel of our own
// rather than hipMemsetAsync: captured into a hipGraph next to other fills, the memset NODE replayed with another
// node's fill value (ROCm 7.2; observed 7168 = the element count of a neighbouring tensor fill) — harmless for the
// epoch-tagged cells, fatal for the flags.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned long long* p, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) p[i] = 0ull;
}
// Behind a cluster launch that NO fix-up launch follows ("cluster_fixup" = 0, or a horizon the fix-up kernel cannot hold): every trajectory
// whose completion count is short of G is REPORTED here — d_iters = 0xFFFFFFFF, d_max_iter_exit = 2 — whatever the members themselves stored.
// A member that gives up writes that pair itself, but member 0 may still pass its last hand-off (the failed peer published before it timed
// out) and store a valid-looking count over it, and a cluster that gives up stops drawing from the queue: trajectories it would have drawn
// get no store at all.  The completion counts know both.
__global__ __launch_bounds__(256) void cluster_report_kernel(const unsigned long long* flags, int stride, unsigned long long G, int batch,
                                                             uint32_t* iters, uint8_t* max_iter_exit) {
    const int b = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (b >= batch) return;
    if (__hip_atomic_load(flags + (size_t)b * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != G) {
        iters[b] = 0xFFFFFFFFu;
        max_iter_exit[b] = 2;
    }
}
// The same fill + the handle's copy of the caller's lambda ([batch][N][n], n32 dwords): what a fix-up launch warm-starts from (PcgArgs::lam0).
// One launch in front of every cluster launch.
__global__ __launch_bounds__(256) void cluster_prologue_kernel(unsigned long long* p, size_t count, const uint32_t* src, uint32_t* dst, size_t n32) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (size_t j = i; j < count; j += stride) p[j] = 0ull;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const size_t n4 = n32 / 4;
        for (size_t j = i; j < n4; j += stride) reinterpret_cast<u4v*>(dst)[j] = reinterpret_cast<const u4v*>(src)[j];
        for (size_t j = 4 * n4 + i; j < n32; j += stride) dst[j] = src[j];
    } else {
        for (size_t j = i; j < n32; j += stride) dst[j] = src[j];
    }
}

// The symmetry latch / the "check_symmetry" debug option: the lane-pair kernels read only the left and diagonal block
// columns of S and Pinv and use L_{k+1}^T where the reference's kernel reads block (k, right).  One wavefront per (trajectory,
// k < N-1): counts the pairs whose blocks differ by more than rel_tol x the largest entry of the pair,
//   max_ij | M[k, right](i, j) - M[k+1, left](j, i) |  >  rel_tol * max | M[k, right], M[k+1, left] |        (NaN counts as a violation).
template <typename MT>
__global__ __launch_bounds__(256) void bd_symmetry_check_kernel(const MT* __restrict__ M, int N, int batch, float rel_tol,
                                                                unsigned long long* __restrict__ violations, unsigned long long* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)batch * (N - 1)) return;
    const long b = item / (N - 1);
    const int k = (int)(item - b * (N - 1));
    const MT* R = M + ((size_t)b * N + k) * ROWF + 2 * NS * NS;             // block (k, right), column-major
    const MT* Lt = M + ((size_t)b * N + k + 1) * ROWF;                      // block (k+1, left)
    float dmax = 0.f, amax = 0.f;
    bool bad = false;
    for (int e = lane; e < NS * NS; e += 64) {
        const int i = e % NS, j = e / NS;
        const float x = (float)R[j * NS + i], y = (float)Lt[i * NS + j];
        float d;
        if constexpr (sizeof(MT) == 8) d = (float)fabs((double)R[j * NS + i] - (double)Lt[i * NS + j]);      // (the difference in the storage type: the double tolerance is 1e-6)
        else d = fabsf(x - y);
        bad |= !(d == d);
        dmax = fmaxf(dmax, d);
        amax = fmaxf(amax, fmaxf(fabsf(x), fabsf(y)));
    }
    for (int o = 32; o; o >>= 1) {
        dmax = fmaxf(dmax, __shfl_xor(dmax, o));
        amax = fmaxf(amax, __shfl_xor(amax, o));
        bad |= __shfl_xor((int)bad, o) != 0;
    }
    if (lane == 0 && (bad || dmax > rel_tol * amax)) {
        if (violations) atomicAdd(violations, 1ull);        // the debug option's count
        if (flag) atomicOr(flag, 1ull);                     // the handle's latch (0 / 1: what the gated launches compare with)
    }
}

// mpcg_probe_hbm_read: a pure read of n4 float4 (tools/_prof/read_rate.hip: 7.07 TB/s with two workgroups per CU, profiles/r04_spmv.txt)
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const f4* __restrict__ in, float* out, size_t n4) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += __builtin_nontemporal_load(in + i);
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// fp32 -> fp16 copy of a bd-layout matrix (round to nearest even), 8 elements per thread.
__global__ __launch_bounds__(256) void f32_to_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t count) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= count) {
        const f4 a = *reinterpret_cast<const f4*>(src + i), b = *reinterpret_cast<const f4*>(src + i + 4);
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 o = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        *reinterpret_cast<h8*>(dst + i) = o;
    } else {
        for (size_t e = i; e < count; ++e) dst[e] = (_Float16)src[e];
    }
}

// Dispatch order for the NEXT call from the iteration counts of this one: trajectories sorted by descending count (counting sort over
// min(iters, 1023) / 16; one workgroup).  Why: warm-started solves leave the loop at different iterations, and the hardware hands the
// workgroups of a launch to the shader engines in index order — a free CU waits behind a busy engine's turn.  1024 N=128 systems with
// 0..167 iterations (tools/_prof/order_experiment.py): 1.09 ms in random order, 0.73 ms sorted by count (0.79 ms sorted the other way:
// neighbours of similar length keep the engines in step).  Consecutive SQP iterations of an MPC loop need similar counts per trajectory,
// so the previous call's counts are the predictor.  Only the order of dispatch changes: every trajectory is solved exactly as before.
// (64 buckets of 16 iterations: the prefix over the buckets is one wavefront's shuffle scan — the kernel sits between the solve and the
//  caller's D2H of the results, so it is kept to three barriers: ~3 us.)
__global__ __launch_bounds__(1024) void sched_order_kernel(const uint32_t* __restrict__ iters, int batch, uint32_t* __restrict__ order) {
    constexpr int NB = 64, NT = 1024;
    __shared__ unsigned hist[NB];
    const int t = threadIdx.x;
    auto key = [](uint32_t it) -> int { const uint32_t c = it < 1023u ? it : 1023u; return (NB - 1) - (int)(c >> 4); };   // bucket 0 = the longest solves
    if (t < NB) hist[t] = 0;
    __syncthreads();
    for (int i = t; i < batch; i += NT) atomicAdd(&hist[key(iters[i])], 1u);
    __syncthreads();
    if (t < NB) {                                          // exclusive prefix over the 64 buckets inside wavefront 0
        const unsigned mine = hist[t];
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < NB; d <<= 1) {
            const unsigned up = __shfl_up(incl, d);
            if (t >= d) incl += up;
        }
        hist[t] = incl - mine;
    }
    __syncthreads();
    for (int i = t; i < batch; i += NT) {
        const unsigned pos = atomicAdd(&hist[key(iters[i])], 1u);
        order[1 + pos] = (uint32_t)i;
    }
    if (t == 0) order[0] = (uint32_t)batch;                // (visible, together with the permutation, to the next kernel of the stream)
}

// ------------------------------------------------------------------------------------------------
// Stand-alone batched block-tridiagonal SpMV  y = M x  (roofline kernel).
// One wavefront per block row at a time; a wavefront walks SPANS of SPMV_SPAN = 16 consecutive block rows (37,632 contiguous bytes of
// the matrix), spans strided over the grid.  x is read from global (L2-resident, 56 B per knot).  Round 4, three findings
// (tools/_prof/read_pattern.hip, tools/_prof/spmv_sweep.py, profiles/r04_spmv.txt):
//   * the load shape (49 lanes x 16 B, three loads per row) reads 1.2 GB at 6.8-7.1 TB/s when nothing else is in the kernel: not the limit;
//   * the x pairs used to be loaded inside compute(): vector-memory results return in issue order, so waiting for them also waited for the
//     NEXT task's matrix loads issued just before — the double buffer overlapped nothing (one wavefront per SIMD: 2.4 -> 3.9 TB/s);
//   * the y stores — 2 % of the bytes — cost 25 % of the time: 56 bytes per row from four lanes, 2.3 rows of DIFFERENT wavefronts (on different
//     XCDs: different L2s) per 128-byte line, i.e. every line went to memory as several partial writes.  Now the 16 rows of a span are staged
//     in LDS (896 B = exactly seven lines, the span starts on a line boundary) and leave as ONE store of 56 lanes x 16 B.
//   * x: one 16-byte load per lane and SPAN (18 knots = 1008 B), a span ahead, parked in LDS — instead of three 8-byte global loads per lane and
//     ROW in the same in-order queue as the matrix stream (+11 % on top of the y stage).
// ------------------------------------------------------------------------------------------------
struct SpmvArgs { const float* M; const float* x; float* y; int N; int batch; int cols; };
#ifndef SPMV_SPAN_ROWS
#define SPMV_SPAN_ROWS 16
#endif
#ifndef SPMV_DEPTH
#define SPMV_DEPTH 2
#endif
constexpr int SPMV_SPAN = SPMV_SPAN_ROWS;                  // rows per span: a multiple of 16 (16 x 56 B of y = seven whole 128-byte lines)
constexpr int SPMV_D = SPMV_DEPTH;                         // block rows in flight per wavefront
static_assert(SPMV_SPAN % 16 == 0 && SPMV_SPAN % SPMV_D == 0, "whole lines of y, whole rounds of the buffers");

template <int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void bt_spmv_kernel(SpmvArgs a) {
    constexpr int XF = (SPMV_SPAN + 2) * NS;               // floats of x a span multiplies: its rows' knots and one knot either side
    constexpr int XP = (XF / 4 + 63) / 64;                 // ... as 64-lane pieces of float4 (18 x 14 = 252 floats = 63 float4: one piece)
    __shared__ __attribute__((aligned(16))) float ystage[NW][SPMV_SPAN * NS];
    __shared__ __attribute__((aligned(16))) float xstage[NW][XP * 256];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    const long total = (long)a.batch * N, totalF = total * NS;
    const long spans = (total + SPMV_SPAN - 1) / SPMV_SPAN;
    const long gw = (long)blockIdx.x * NW + w, GW = (long)gridDim.x * NW;
    const LaneMap L(lane);
    // One SRD per launch would need > 4 GiB of range at large batch, so the descriptor is rebuilt
    // per task from the (wave-uniform) trajectory index.
    const uint32_t lane_off = lane < BLK4 ? (uint32_t)lane * 16u : OOB_OFF;
    const size_t mstride = (size_t)N * ROWF;
    struct Task { Rows m; int kq; };
    auto load_task = [&](long q) -> Task {
        const bool valid = q < total;
        const long qq = valid ? q : 0;
        const int bt = (int)(qq / N);
        const int kq = (int)(qq - (long)bt * N);
        const int k = valid ? kq : N;                              // k = N -> all three loads OOB
        const rsrc_t r = make_rsrc(a.M + (size_t)bt * mstride, (uint32_t)(mstride * sizeof(float)));
        Task t;
        t.m = load_rows<NT>(r, k, N, a.cols, lane_off);
        t.kq = kq;
        return t;
    };
    // x of the span that starts at row q0: knots q0 - 1 .. q0 + SPAN of the flat [batch N][14] array, ONE 16-byte load per lane, requested a
    // whole span ahead and parked in LDS for the span's rows (three 8-byte LDS reads per row and lane instead of three global loads: those were
    // 48 vector-memory instructions per span in the same in-order queue as the matrix stream).  Entries outside the array read as zero.
    auto load_x = [&](long q0, f4 (&xp)[XP]) {
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const long gi = (q0 - 1) * NS + 4 * (lane + 64 * p);
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (gi >= 0 && gi + 4 <= totalF) v = *reinterpret_cast<const f4*>(a.x + gi);          // (8-byte aligned: the hardware takes dword-aligned 16-byte loads)
            else {
                if (gi >= 0 && gi < totalF) v.x = a.x[gi];
                if (gi + 1 >= 0 && gi + 1 < totalF) v.y = a.x[gi + 1];
                if (gi + 2 >= 0 && gi + 2 < totalF) v.z = a.x[gi + 2];
                if (gi + 3 >= 0 && gi + 3 < totalF) v.w = a.x[gi + 3];
            }
            xp[p] = v;
        }
    };
    auto fma4 = [&](f4& acc, const f4 m, const f2 x) {
        const float x01 = L.a01 ? x.x : x.y;
        const float x23 = L.a23 ? x.x : x.y;
        acc.x = fmaf(m.x, x01, acc.x);
        acc.y = fmaf(m.y, x01, acc.y);
        acc.z = fmaf(m.z, x23, acc.z);
        acc.w = fmaf(m.w, x23, acc.w);
    };
    float* ys = ystage[w];
    const float* xs = xstage[w] + L.g2;                    // slot s = knot q0 - 1 + s; this lane's column pair
    // row r of the current span: its 14 results (lanes 0..3 of reduce_rows: 4 + 4 + 4 + 2) into the stage.  A neighbour that does not exist
    // has an all-zero block: its operand is the row's own knot, to stay finite.
    auto compute = [&](const Task& use, int r) {
        const f2 xk = *reinterpret_cast<const f2*>(xs + (r + 1) * NS);
        const f2 xl = *reinterpret_cast<const f2*>(xs + (use.kq > 0 ? r : r + 1) * NS);
        const f2 xr = *reinterpret_cast<const f2*>(xs + (use.kq < N - 1 ? r + 2 : r + 1) * NS);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        fma4(acc, use.m.m0, xl);
        fma4(acc, use.m.m1, xk);
        fma4(acc, use.m.m2, xr);
        const f4 y = reduce_rows(acc, lane);
        float* yr = ys + r * NS + 4 * lane;                // (8-byte aligned)
        if (lane < 3) {
            *reinterpret_cast<f2*>(yr) = f2{y.x, y.y};
            *reinterpret_cast<f2*>(yr + 2) = f2{y.z, y.w};
        } else if (lane == 3) {
            *reinterpret_cast<f2*>(yr) = f2{y.x, y.y};
        }
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // (rows past the end of the batch compute on out-of-bounds = zero blocks and are not stored)
    constexpr int D = SPMV_D;
    Task buf[D];
    f4 xp[XP];
    load_x(gw * SPMV_SPAN, xp);
#pragma unroll
    for (int d = 0; d < D - 1; ++d) buf[d] = load_task(gw * SPMV_SPAN + d);
    for (long sp = gw; sp < spans; sp += GW) {
        const long q0 = sp * SPMV_SPAN;
        const long qn = (sp + GW) * SPMV_SPAN;             // first row of this wavefront's next span
#pragma unroll
        for (int p = 0; p < XP; ++p) *reinterpret_cast<f4*>(xstage[w] + 4 * (lane + 64 * p)) = xp[p];
        wave_sync();
        load_x(qn, xp);                                    // the next span's x: a span ahead
#pragma unroll 1
        for (int r = 0; r < SPMV_SPAN; r += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int ahead = r + d + D - 1;           // the row requested now: D - 1 rows ahead, into the buffer freed last
                buf[(d + D - 1) % D] = load_task(ahead < SPMV_SPAN ? q0 + ahead : qn + (ahead - SPMV_SPAN));
                compute(buf[d], r + d);
            }
        }
        // the span's y: 16 x 14 floats = 56 float4, one coalesced store (fewer at the ragged end of the batch)
        wave_sync();
        const long rows_here = total - q0 < SPMV_SPAN ? total - q0 : SPMV_SPAN;
#pragma unroll
        for (int e = 4 * lane; e < SPMV_SPAN * NS; e += 256) {
            if (e < rows_here * NS) {
                const f4 v = *reinterpret_cast<const f4*>(ys + e);
                float* yo = a.y + (size_t)q0 * NS + e;
                if (e + 4 <= rows_here * NS) *reinterpret_cast<f4*>(yo) = v;
                else { yo[0] = v.x; yo[1] = v.y; }        // (rows x 14 floats is even: a ragged tail is two floats)
            }
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA variant of the batched SpMV (BASELINE config 5: "MFMA per-knot block GEMV on") — an EXPERIMENT
// kept for measurement, not the default.  One wave per block row; every 14x14 block is padded to 16x16
// in registers and multiplied as four v_mfma_f32_16x16x4_f32 (A = 16 rows x 4 columns of the block,
// B = the 4 matching x entries in column 0 of a 4x16 operand, zeros elsewhere), the 12 MFMAs of a block
// row accumulating into one 16x16 tile of which only column 0 is the result.  15/16 of every MFMA is
// wasted by construction (one right-hand side per matrix), and fp32-input MFMA runs at the vector FMA
// rate on gfx950, so this cannot beat the VALU kernel; the numbers are in DESIGN.md §3.4.
// A-operand loads: lane l reads element (row l&15, column 4c + (l>>4)): four 56-byte column segments
// per instruction; rows 14, 15 / columns 14, 15 / missing blocks go to the SRD's out-of-bounds path.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void bt_spmv_mfma_kernel(SpmvArgs a) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N;
    const long total = (long)a.batch * N;
    const long gw = (long)blockIdx.x * NW + w, GW = (long)gridDim.x * NW;
    const int li = lane & 15, lk = lane >> 4;
    const size_t mstride = (size_t)N * ROWF;
    for (long q = gw; q < total; q += GW) {
        const int bt = (int)(q / N), k = (int)(q - (long)bt * N);
        const rsrc_t r = make_rsrc(a.M + (size_t)bt * mstride, (uint32_t)(mstride * sizeof(float)));
        const float* xk = a.x + (size_t)q * NS;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const bool blk = (a.cols == 3 || s == 1) && !(s == 0 && k == 0) && !(s == 2 && k == N - 1);
            const float* xs = xk + (s - 1) * NS;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 4 * c + lk;
                const bool ok = blk && li < NS && j < NS;
                const uint32_t off = ok ? (uint32_t)(k * ROWF + s * 196 + j * NS + li) * 4u : OOB_OFF;
                typedef unsigned u1;
                const float av = __builtin_bit_cast(float, (u1)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
                const float bv = (blk && li == 0 && j < NS) ? xs[j] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
        }
        // column 0 of the tile: lanes 0, 16, 32, 48 hold rows 4*(lane>>4) .. +3
        if (li == 0) {
            float* yk = a.y + (size_t)q * NS + 4 * lk;
            *reinterpret_cast<f2*>(yk) = f2{acc.x, acc.y};
            if (lk < 3) *reinterpret_cast<f2*>(yk + 2) = f2{acc.z, acc.w};
        }
    }
}

}  // namespace mpcg
