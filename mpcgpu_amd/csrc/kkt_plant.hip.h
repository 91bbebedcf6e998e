// kkt_plant.hip.h — batched KKT block assembly for a fixed-base serial chain of revolute joints (IIWA-14): the HIP twin of
// generate_kkt_submatrices (reference include/common/kkt.cuh:22-163) together with the plant functions it calls
// (include/dynamics/iiwa/iiwa_eepos_plant.cuh: forwardDynamicsAndGradient :127-155, trackingCostGradientAndHessian :307-390,
// _lastblock :392-411) and the Euler integrator (include/common/integrator.cuh:56-104, 143-162).  SURVEY.md §8f row 4.
//
// The reference runs GRiD-generated, robot-specific code (10 k lines of unrolled recursions) with one thread block per
// knot.  Here the robot is DATA (struct PlantDev: the constant "tree" part of every joint's spatial transform — the joint rotation
// about its z axis is applied in the kernel — and the spatial inertias) and the algorithms are the generic ones, mapped for a
// 64-wide wavefront:
//   one wavefront per FOUR (trajectory, knot) pairs, 16 lanes each; a lane runs a whole recursive Newton-Euler pass, and every
//   quantity the knot needs falls out of TWO such rounds:
//     round 0  lanes 0..6: columns of the joint-space inertia matrix M = ID(q, 0, e_j) — and, for free, column j of the body
//              Jacobian of the last link (its spatial acceleration at the end of the forward sweep);  lane 7: bias c = ID(q, qd, 0);
//              lanes 8..10: the same sweep started from a unit angular base acceleration e_x / e_y / e_z — the last link's
//              acceleration is then [R e_i ; R (e_i x p)]: the POSE of the end effector (R world -> link, p its origin), i.e. the
//              forward kinematics and the geometric Jacobian cost no sweep of their own (round 2 ran them in a separate lane:
//              a divergent branch the wavefront paid for in full);
//              then lanes 0..6: column j of Minv by a Cholesky solve (7x7, redundantly factorised per lane), qdd_j = Minv_j (u - c),
//              the end-effector Jacobian column and the cost gradient entry.
//     round 1  lanes 0..6: ID(q + h e_j, qd, qdd), lanes 7..13: ID(q, qd + h e_j, qdd): ONE-SIDED differences of the inverse
//              dynamics — the nominal value is known without evaluating it, ID(q, qd, qdd) = u because qdd = Minv (u - c)
//              (with h = 3e-8 in float64 the entries of A differ from central-difference values by 1.3e-7, the rounding of a float
//              near 1); each lane turns its difference into ITS column of dqdd/d(q, qd) = -Minv dID in registers and writes its column
//              of A (and of Q, B, R) as float in the reference's dense layouts (column-major blocks, C = -A, -B).
// Arithmetic is float64 inside (the difference quotients need it; fp64 FMA is full rate on the MI355X), results are rounded to float
// on the way out.  Round 6: everything below the sine / cosine is templated on the arithmetic type R, and `"kkt_f32"` = 1 (opt-in) runs the analytic
// kernel in float — linsys_t's own arithmetic, what the reference's GRiD code computes in (forwardDynamicsAndGradient<T>, T = float): outputs within
// 1.5e-6 of the float64 restatement (relative to max(1, |block|); float64 inside: 2e-7).  With R = float that is only 8 % faster (0.303 against 0.328 ms per
// 1024 x 127 knots): in double the fp64 pipe is 75 % busy at two wavefronts per SIMD, and a float instruction takes the same issue slot as a double one.
// What pays is R = kkt_f2: TWO knots per lane, every value a float pair, every multiply-add a v_pk_fma_f32 (model constants straight from scalar
// registers through op_sel) — the same instruction stream as the double build (+15 %: pair moves, one-cycle packed hazards) for twice the knots:
// **0.204 ms** (1.6x); the R = float build's results to float rounding (the compiler contracts the two builds' expressions differently: 2e-6; either within 5e-6 of the float64 restatement over 1024 windows), and
// independent of the batch (which knots share a lane is; a half's arithmetic is not).  "kkt_f32" = 1 is this build, = 2 the one-knot float build.
// LDS is what bounds the resident wavefronts: 16.4 KB (double) / 19.9 KB (packed) per wavefront = EIGHT per CU (two per SIMD, round 3; 30 KB = five
// before), which is what hides the dependent-issue latency of the recursion.
#pragma once
// Float64 work checked by tolerance, not by bits: multiply-adds are FUSED here (the bit-exact headers switch contraction off and back on).
#pragma clang fp contract(fast)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

constexpr int PJ = 7;                    // joints of the compiled specialisation (state 2 PJ, control PJ)
constexpr int KKT_LANES = 64;            // one wavefront per KKT_ITEMS (trajectory, knot) pairs
constexpr int KKT_THREADS = KKT_LANES;
constexpr int KKT_ITEMS = 4;             // (trajectory, knot) pairs per wavefront: 16 lanes each
constexpr int KKT_GL = KKT_LANES / KKT_ITEMS;
constexpr int KKT_RL = 2 * PJ;           // lanes of a group that own a record (the other two never run the recursion)
#ifndef KKT_ABLATE
#define KKT_ABLATE 0      // timing experiments only (tools/kkt_ablate.sh): 1 no round-1 recursion, 2 no Cholesky, 4 no sincos, 8 no output stores, 16 no round-0 recursion
#endif
constexpr double KKT_FD_H = 3e-8;        // one-sided difference step (truncation h/2 |f''| ~ roundoff eps |f| / h in float64)
// Per-lane record of the recursion in LDS: the forces of links 0..5 wait there for the backward sweep (the last link's stays in
// registers), 6 rows each; tau_k overwrites row 2 of link k's force once that is consumed, tau_6 takes the 37th row (an odd row
// count = conflict-free 8-byte accesses at lane stride).
constexpr int RN_ROWS = 6 * (PJ - 1) + 1;
__host__ __device__ constexpr int RN_TAU(int k) { return k < PJ - 1 ? 6 * k + 2 : 6 * (PJ - 1); }
constexpr int RN_AW = 3, RN_AU = 9;      // rows (3 each) where a lane leaves the last link's acceleration after round 0 (consumed force rows)

// X_k(q_k) = blkdiag(Rz, Rz) [[ET, 0], [BT, ET]],  Rz(q) = [[c, s, 0], [-s, c, 0], [0, 0, 1]]  (row-major 3x3 blocks)
// Spatial inertia of a rigid body: [[Ibar, skew(h)], [skew(h)^T, m 1]] (Ibar symmetric about the link frame's origin, h = m c) — ten
// numbers, I [w; u] = [Ibar w + h x u ; m u - h x w]: 24 multiply-adds instead of 36.
template <typename R> struct PlantDevT {
    R ET[PJ][9], BT[PJ][9];
    R Ib[PJ][10];                                // Ixx Ixy Ixz Iyy Iyz Izz  hx hy hz  m
};
typedef PlantDevT<double> PlantDev;              // R = double: the round-2..5 kernel (float64 inside, the checker of the float build); R = float (round 6): linsys_t's own
                                                 // arithmetic, what the reference's GRiD code runs in (gato_plant::forwardDynamicsAndGradient<T>, T = linsys_t = float)

template <typename R> struct KktArgsT {
    const PlantDevT<R>* plant;
    const float* eePos_traj;             // [batch][N][6]
    const float* xs;                     // [batch][n]
    const float* xu;                     // [batch][(n+m)N - m]
    float* G; float* C; float* g; float* c;
    int N; int batch;
    R dt, qd_cost, r_cost;
    int analytic;                        // 1: round 1 = the analytic gradient recursion of the inverse dynamics (default); 0: one-sided differences
};
typedef KktArgsT<double> KktArgs;

// The model tables are read through the CONSTANT address space (same 64-bit address as the global pointer): loads from it are
// invariant by definition, so a uniform address makes them scalar loads (s_load, scalar cache).  Through the plain global
// pointer the compiler must assume the kernel's own stores may clobber the table and emits ~1,300 vector loads per knot
// (rocprofv3: SQ_INSTS_VMEM_RD; waves waited on memory half of their cycles).
template <typename R> struct PlantC {
    typedef const __attribute__((address_space(4))) R creal;
    creal* base;
    __device__ __forceinline__ creal* at(size_t byte_off, int k, int per) const { return base + byte_off / sizeof(R) + (size_t)k * per; }
    __device__ __forceinline__ creal* ET(int k) const { return at(offsetof(PlantDevT<R>, ET), k, 9); }
    __device__ __forceinline__ creal* BT(int k) const { return at(offsetof(PlantDevT<R>, BT), k, 9); }
    __device__ __forceinline__ creal* Ib(int k) const { return at(offsetof(PlantDevT<R>, Ib), k, 10); }
};

// ---- arithmetic types of the kernel: double (default), float ("kkt_f32" = 1 below a full batch), and kkt_f2 = TWO knots per lane in packed float
//      (round 6: v_pk_fma_f32 does two knots' multiply-add in the issue slot of one — the recursions are bound by VALU issue, not by latency) ----
typedef float kkt_f2 __attribute__((ext_vector_type(2)));
template <typename R> struct KktR {                          // one knot per lane
    typedef R scalar;                                        // type of the model tables and of the kernel's scalar arguments
    typedef float rec;                                       // what the link forces of the analytic round wait as in LDS
    static constexpr int KP = 1;
    static constexpr bool is_double = sizeof(R) == 8;
    __device__ static __forceinline__ R mk(scalar a, scalar) { return a; }
    __device__ static __forceinline__ scalar get(R x, int) { return x; }
    __device__ static __forceinline__ R splat(double v) { return (R)v; }
    __device__ static __forceinline__ rec to_rec(R x) { return (float)x; }
    __device__ static __forceinline__ R from_rec(rec x) { return (R)x; }
    __device__ static __forceinline__ R rsq(R x) {
        if constexpr (is_double) return __builtin_amdgcn_rsq(x);
        else return __builtin_amdgcn_rsqf(x);
    }
};
template <> struct KktR<kkt_f2> {                            // two knots per lane: .x = the even item of the lane group's pair, .y = the odd one
    typedef float scalar;
    typedef kkt_f2 rec;
    static constexpr int KP = 2;
    static constexpr bool is_double = false;
    __device__ static __forceinline__ kkt_f2 mk(float a, float b) { return kkt_f2{a, b}; }
    __device__ static __forceinline__ float get(kkt_f2 x, int h) { return h ? x.y : x.x; }
    __device__ static __forceinline__ kkt_f2 splat(double v) { return kkt_f2{(float)v, (float)v}; }
    __device__ static __forceinline__ kkt_f2 to_rec(kkt_f2 x) { return x; }
    __device__ static __forceinline__ kkt_f2 from_rec(kkt_f2 x) { return x; }
    __device__ static __forceinline__ kkt_f2 rsq(kkt_f2 x) { return kkt_f2{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)}; }
};
#define KR(v) (KktR<R>::splat(v))

template <typename R> struct KktItemLds {   // per-knot scratch in LDS (840 B in double)
    R Minv[PJ][PJ];
    R Qdd[PJ];
    R Xq[2 * PJ], U[PJ];                 // [q; qd], u of this knot
    R Sc[2][PJ];                         // sin / cos of q
    R Gq[PJ], Gq1[PJ];                   // J^T (ee - goal_k), J^T (ee - goal_{k+1})
};

// LDS is addressed through explicit address-space pointers: through generic pointers the accesses become flat loads whose 64-bit
// addresses (one per record row touched) the compiler hoists out of the knot loop and spills.
template <typename R> struct KktLds {
    typedef __attribute__((address_space(3))) volatile R vr;
    typedef __attribute__((address_space(3))) KktItemLds<R> item;
};

// sin and cos of a joint angle: Cody-Waite reduction by pi/2 in two fused steps + the classic minimax kernels on [-pi/4, pi/4] (the
// coefficients every libm uses since fdlibm); absolute error 2.2e-16 for |x| <= 1e4 (checked against numpy on 2e6 points) — a fifth of the
// instructions of the library sincos, whose Payne-Hanek path for huge arguments a joint angle never needs; beyond 1e4 that one is called.
__device__ __forceinline__ void kkt_sincos(double x, double& sn, double& cs) {
    if (!(fabs(x) <= 1e4)) { sincos(x, &sn, &cs); return; }
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s = r + r * z * ps, c = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    sn = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    cs = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

template <typename R> struct RneaTask {                        // what this lane's recursion evaluates
    int sj;                              // joint whose angle is q + h (-1: none): (sin, cos) -> (s + h c, c - h s), exact to h^2 / 2 = 4.5e-16
    int pj;                              // joint whose velocity is qd + h (-1: none)
    R qdscale;                      // 0: qd = 0, 1: qd of the knot
    bool knot_qdd;                       // qdd of the knot (round 1) / unit vector e_unit (round 0)
    int unit;
    int base;                            // unit angular base acceleration e_base (-1: none)
};

// tau = ID(q, qd, qdd) without gravity (gato_plant::GRAVITY = 0, iiwa_eepos_plant.cuh:53).  The link forces that wait for the backward
// sweep live in this lane's record `fl` (rows RN_*), and both sweeps are RUNTIME loops over the joints.  Measured alternatives on
// gfx950 (hipcc 7.2): everything in registers with unrolled sweeps = 3.4 KB of scratch per lane (the 84 force registers, plus the
// transforms the compiler keeps from the forward sweep for the backward one, plus 840 hoisted table loads); plain (non-volatile)
// LDS accesses get store-forwarded back into registers.  The joint transform is applied as the constant tree part (operands straight
// from scalar registers: one SGPR pair per FMA is what the ISA allows, so forming E(q) = E0 + Es sin + Ec cos first, as rounds 1-2 did,
// cost three instructions per matrix entry and sweep) followed by the rotation about z: 4 instructions per 3-vector.
// Returns the last link's spatial acceleration (aw, au) in its own frame.
template <typename R>
__device__ __forceinline__ void rnea(const PlantC<typename KktR<R>::scalar>& P, typename KktLds<R>::vr* fl, typename KktLds<R>::item* I, const RneaTask<R> t, R (&aw_out)[3], R (&au_out)[3]) {
    typedef typename PlantC<typename KktR<R>::scalar>::creal creal;
    R vw[3], vu[3], aw[3], au[3], f[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) { vw[r] = KR(0.0); vu[r] = KR(0.0); aw[r] = KR(0.0); au[r] = KR(0.0); f[r] = KR(0.0); f[3 + r] = KR(0.0); }
    if (t.base >= 0) aw[t.base] = KR(1.0);
#pragma nounroll
    for (int kv = 0; kv < PJ; ++kv) {
        const int k = __builtin_amdgcn_readfirstlane(kv);    // uniform by construction; said so, the model tables come through s_load
        const R qdk = t.qdscale * I->Xq[PJ + k] + (k == t.pj ? KR(KKT_FD_H) : KR(0.0));
        const R qddk = t.knot_qdd ? I->Qdd[k] : (k == t.unit ? KR(1.0) : KR(0.0));
        R sn = I->Sc[0][k], cs = I->Sc[1][k];
        if (k == t.sj) { const R s0 = sn; sn = s0 + KR(KKT_FD_H) * cs; cs = cs - KR(KKT_FD_H) * s0; }
        creal* E = P.ET(k);
        creal* B = P.BT(k);
        R tw[3], tu[3], sw[3], su[3];                   // tree part of v = X v_parent, a = X a_parent
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            tw[r] = E[3 * r] * vw[0] + E[3 * r + 1] * vw[1] + E[3 * r + 2] * vw[2];
            tu[r] = B[3 * r] * vw[0] + B[3 * r + 1] * vw[1] + B[3 * r + 2] * vw[2] + E[3 * r] * vu[0] + E[3 * r + 1] * vu[1] + E[3 * r + 2] * vu[2];
            sw[r] = E[3 * r] * aw[0] + E[3 * r + 1] * aw[1] + E[3 * r + 2] * aw[2];
            su[r] = B[3 * r] * aw[0] + B[3 * r + 1] * aw[1] + B[3 * r + 2] * aw[2] + E[3 * r] * au[0] + E[3 * r + 1] * au[1] + E[3 * r + 2] * au[2];
        }
        R w[3], u[3], bw[3], bu[3];                     // joint rotation about z
        w[0] = cs * tw[0] + sn * tw[1]; w[1] = cs * tw[1] - sn * tw[0]; w[2] = tw[2] + qdk;       // + S qd, S = e_z (angular)
        u[0] = cs * tu[0] + sn * tu[1]; u[1] = cs * tu[1] - sn * tu[0]; u[2] = tu[2];
        bw[0] = cs * sw[0] + sn * sw[1]; bw[1] = cs * sw[1] - sn * sw[0]; bw[2] = sw[2] + qddk;
        bu[0] = cs * su[0] + sn * su[1]; bu[1] = cs * su[1] - sn * su[0]; bu[2] = su[2];
        // + v x (S qd): column 2 of crm(v) times qd
        bw[0] += w[1] * qdk; bw[1] -= w[0] * qdk;
        bu[0] += u[1] * qdk; bu[1] -= u[0] * qdk;
        // f = I a + v x* (I v)
        R Ia[6], Iv[6];
        creal* Ik = P.Ib(k);
        auto imul = [&](const R (&W)[3], const R (&U)[3], R (&o)[6]) {
            o[0] = Ik[0] * W[0] + Ik[1] * W[1] + Ik[2] * W[2] + (Ik[7] * U[2] - Ik[8] * U[1]);
            o[1] = Ik[1] * W[0] + Ik[3] * W[1] + Ik[4] * W[2] + (Ik[8] * U[0] - Ik[6] * U[2]);
            o[2] = Ik[2] * W[0] + Ik[4] * W[1] + Ik[5] * W[2] + (Ik[6] * U[1] - Ik[7] * U[0]);
            o[3] = Ik[9] * U[0] - (Ik[7] * W[2] - Ik[8] * W[1]);
            o[4] = Ik[9] * U[1] - (Ik[8] * W[0] - Ik[6] * W[2]);
            o[5] = Ik[9] * U[2] - (Ik[6] * W[1] - Ik[7] * W[0]);
        };
        imul(bw, bu, Ia);
        imul(w, u, Iv);
        // crf(v) h = [w x n + u x l ; w x l],  h = [n; l]
        f[0] = Ia[0] + (w[1] * Iv[2] - w[2] * Iv[1]) + (u[1] * Iv[5] - u[2] * Iv[4]);
        f[1] = Ia[1] + (w[2] * Iv[0] - w[0] * Iv[2]) + (u[2] * Iv[3] - u[0] * Iv[5]);
        f[2] = Ia[2] + (w[0] * Iv[1] - w[1] * Iv[0]) + (u[0] * Iv[4] - u[1] * Iv[3]);
        f[3] = Ia[3] + (w[1] * Iv[5] - w[2] * Iv[4]);
        f[4] = Ia[4] + (w[2] * Iv[3] - w[0] * Iv[5]);
        f[5] = Ia[5] + (w[0] * Iv[4] - w[1] * Iv[3]);
        if (k < PJ - 1) {
#pragma unroll
            for (int r = 0; r < 6; ++r) fl[6 * k + r] = f[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { vw[r] = w[r]; vu[r] = u[r]; aw[r] = bw[r]; au[r] = bu[r]; }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { aw_out[r] = aw[r]; au_out[r] = au[r]; }
    fl[RN_TAU(PJ - 1)] = f[2];                               // tau_6; f = force of the link being folded into its parent
#pragma nounroll
    for (int kv = PJ - 1; kv >= 1; --kv) {                   // f_parent += X^T f = Xtree^T blkdiag(Rz^T, Rz^T) [n; l] = [ET^T n' + BT^T l' ; ET^T l']
        const int k = __builtin_amdgcn_readfirstlane(kv);
        R sn = I->Sc[0][k], cs = I->Sc[1][k];
        if (k == t.sj) { const R s0 = sn; sn = s0 + KR(KKT_FD_H) * cs; cs = cs - KR(KKT_FD_H) * s0; }
        creal* E = P.ET(k);
        creal* B = P.BT(k);
        const R n0 = cs * f[0] - sn * f[1], n1 = sn * f[0] + cs * f[1], n2 = f[2];
        const R l0 = cs * f[3] - sn * f[4], l1 = sn * f[3] + cs * f[4], l2 = f[5];
        R fp[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            fp[r] = fl[6 * (k - 1) + r] + E[r] * n0 + E[3 + r] * n1 + E[6 + r] * n2 + B[r] * l0 + B[3 + r] * l1 + B[6 + r] * l2;
            fp[3 + r] = fl[6 * (k - 1) + 3 + r] + E[r] * l0 + E[3 + r] * l1 + E[6 + r] * l2;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) f[r] = fp[r];
        fl[RN_TAU(k - 1)] = f[2];                            // (link k-1's own force row 2: consumed just above)
    }
}


// ---- round 1, analytic (round 4): d ID / d(q, qd) at (q, qd, qdd) by the derivative recursion of the inverse dynamics — what the reference
// computes with GRiD's generated forwardDynamicsAndGradient (iiwa_eepos_plant.cuh:127-155); written here from the recursion itself:
//     v_i = X_i v_{i-1} + S qd_i,   a_i = X_i a_{i-1} + S qdd_i + v_i x S qd_i,   f_i = I_i a_i + v_i x* I_i v_i,   F_{i-1} += X_i^T F_i,   tau_i = S^T F_i
// with d(X_i u)/dq_i = -S x (X_i u) and d(X_i^T F)/dq_i = X_i^T (S x* F)  (S = unit rotation about the joint's z axis).  For column j:
//     links i < j: dv = da = 0.   i = j, d/dq_j:  dv = -S x v_j,  da = -S x (X_j a_{j-1}) + dv x S qd_j;    d/dqd_j:  dv = S,  da = v_j x S
//     links i > j: dv_i = X_i dv_{i-1},  da_i = X_i da_{i-1} + dv_i x S qd_i          (the nominal recursion without its S qd / S qdd sources)
//     every i >= j: df_i = I_i da_i + dv_i x* (I_i v_i) + v_i x* (I_i dv_i);   backward as above, plus X_j^T (S x* F_j) into link j-1 for d/dq_j.
// One lane per column (0..6: q, 7..13: qd) exactly like the difference round it replaces — and a 15th lane runs the NOMINAL recursion
// (q, qd, qdd) in lock-step in the same instruction stream: what the column lanes need from it at link i (v_i, I_i v_i, X_i a_{i-1} going up,
// F_i coming down) exists in lane 14's registers at that very point and travels by DPP row broadcast — no nominal sweep of its own, nothing
// of it in LDS.  The link forces wait for the backward sweep in the same LDS region as before, as FLOATS (15 records fit where 14 double
// records were; a derivative needs no cancellation headroom: rounding its forces to 6e-8 relative moves dtau by that much).
template <typename R> struct KktRecLds { typedef __attribute__((address_space(3))) volatile typename KktR<R>::rec vf; };
template <int L>
__device__ __forceinline__ double bc64(double x) {            // x of lane L of this lane's 16-lane group
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x150 + L, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x150 + L, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int L>
__device__ __forceinline__ float bc64(float x) {              // (the float build: one row_newbcast move)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + L, 0xf, 0xf, true));
}
template <int L>
__device__ __forceinline__ kkt_f2 bc64(kkt_f2 x) {            // (the packed build: two moves, as for a double)
    return kkt_f2{bc64<L>(x.x), bc64<L>(x.y)};
}
constexpr int KKT_NOM = 2 * PJ;          // the lane of the nominal recursion
// Records of the analytic round: one per lane (15) — in the packed build 14: lanes 0..12 their own, lane 13 inside lane 6's, lane 14 (nominal) the 14th (see rnea_grad)
template <typename R> struct KktGrad {
    static constexpr bool SHARE = KktR<R>::KP == 2;
    static constexpr int RECS = SHARE ? 2 * PJ : 2 * PJ + 1;
    __device__ static __forceinline__ int rec(int l) { return !SHARE || l < 2 * PJ - 1 ? l : (l == 2 * PJ - 1 ? PJ - 1 : 2 * PJ - 1); }
    // row of dtau_i in the record of lane l: RN_TAU(i); for lane 13 of the packed build the free row behind it (tau_6: row 0)
    __device__ static __forceinline__ int tau(int l, int i) { return SHARE && l == 2 * PJ - 1 ? (i < PJ - 1 ? 6 * i + 3 : 0) : RN_TAU(i); }
};
// l: lane in the group (0..14 run).  On return this lane's record holds dtau_i (rows RN_TAU(i), float) for its column.
template <typename R>
__device__ __forceinline__ void rnea_grad(const PlantC<typename KktR<R>::scalar>& P, typename KktRecLds<R>::vf* fl, typename KktLds<R>::item* I, const int l) {
    typedef typename PlantC<typename KktR<R>::scalar>::creal creal;
    const bool nom = l == KKT_NOM;
    const bool isq = l < PJ;
    const int col = l < PJ ? l : l - PJ;                      // (lane 14: 7 — never equal to a link index)
    const R nmask = nom ? KR(0.0) : KR(1.0);
    // A column's link forces are ZERO below its own joint on the way up (dv = da = 0 there).  The packed build neither stores those rows nor uses what it reads
    // from them — which leaves the whole force part of the records of lanes 6 and 13 (column joint 6) unused, so the two SHARE one record (KktGrad: lane 13's
    // tau rows sit one row behind lane 6's): 14 pair records instead of 15 = 19.9 instead of 21.1 KB per wavefront = eight wavefronts per CU instead of seven.
    // (The one-knot builds keep a record per lane and the plain accesses: the selects cost the double build 7 %.)
    constexpr bool SHARE = KktGrad<R>::SHARE;
    const int first = nom ? 0 : col;
    R vw[3], vu[3], aw[3], au[3], f[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) { vw[r] = KR(0.0); vu[r] = KR(0.0); aw[r] = KR(0.0); au[r] = KR(0.0); f[r] = KR(0.0); f[3 + r] = KR(0.0); }
#pragma nounroll
    for (int kv = 0; kv < PJ; ++kv) {
        const int k = __builtin_amdgcn_readfirstlane(kv);
        const R qdk = I->Xq[PJ + k], qddk = I->Qdd[k];
        const R sn = I->Sc[0][k], cs = I->Sc[1][k];
        creal* E = P.ET(k);
        creal* B = P.BT(k);
        R tw[3], tu[3], sw[3], su[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            tw[r] = E[3 * r] * vw[0] + E[3 * r + 1] * vw[1] + E[3 * r + 2] * vw[2];
            tu[r] = B[3 * r] * vw[0] + B[3 * r + 1] * vw[1] + B[3 * r + 2] * vw[2] + E[3 * r] * vu[0] + E[3 * r + 1] * vu[1] + E[3 * r + 2] * vu[2];
            sw[r] = E[3 * r] * aw[0] + E[3 * r + 1] * aw[1] + E[3 * r + 2] * aw[2];
            su[r] = B[3 * r] * aw[0] + B[3 * r + 1] * aw[1] + B[3 * r + 2] * aw[2] + E[3 * r] * au[0] + E[3 * r + 1] * au[1] + E[3 * r + 2] * au[2];
        }
        R w[3], u[3], bw[3], bu[3];
        w[0] = cs * tw[0] + sn * tw[1]; w[1] = cs * tw[1] - sn * tw[0]; w[2] = tw[2] + (nom ? qdk : KR(0.0));
        u[0] = cs * tu[0] + sn * tu[1]; u[1] = cs * tu[1] - sn * tu[0]; u[2] = tu[2];
        bw[0] = cs * sw[0] + sn * sw[1]; bw[1] = cs * sw[1] - sn * sw[0]; bw[2] = sw[2];
        bu[0] = cs * su[0] + sn * su[1]; bu[1] = cs * su[1] - sn * su[0]; bu[2] = su[2];
        // what the column lanes need of the nominal recursion at this link: v_k (after its S qd), X_k a_{k-1} (before S qdd and the cross term)
        R nvw[3], nvu[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { nvw[r] = bc64<KKT_NOM>(w[r]); nvu[r] = bc64<KKT_NOM>(u[r]); }
        const R naw0 = bc64<KKT_NOM>(bw[0]), naw1 = bc64<KKT_NOM>(bw[1]), nau0 = bc64<KKT_NOM>(bu[0]), nau1 = bc64<KKT_NOM>(bu[1]);
        if (k == col) {                                       // this lane's own joint: the sources of its column (everything above was zero)
            if (isq) {
                w[0] = nvw[1]; w[1] = -nvw[0]; w[2] = KR(0.0); u[0] = nvu[1]; u[1] = -nvu[0]; u[2] = KR(0.0);            // dv = -S x v
                bw[0] = naw1; bw[1] = -naw0; bw[2] = KR(0.0); bu[0] = nau1; bu[1] = -nau0; bu[2] = KR(0.0);               // da = -S x (X a_parent) [+ dv x S qd below]
            } else {
                w[0] = KR(0.0); w[1] = KR(0.0); w[2] = KR(1.0); u[0] = KR(0.0); u[1] = KR(0.0); u[2] = KR(0.0);                            // dv = S
                bw[0] = nvw[1]; bw[1] = -nvw[0]; bw[2] = KR(0.0); bu[0] = nvu[1]; bu[1] = -nvu[0]; bu[2] = KR(0.0);       // da = v x S
            }
        }
        if (nom) bw[2] += qddk;                               // + S qdd (nominal lane only)
        // + (v or dv) x (S qd_k): column 2 of crm(.) times qd_k — the same expression in the nominal and in the column lanes
        bw[0] += w[1] * qdk; bw[1] -= w[0] * qdk;
        bu[0] += u[1] * qdk; bu[1] -= u[0] * qdk;
        R Ia[6], Iv[6];
        creal* Ik = P.Ib(k);
        auto imul = [&](const R (&W)[3], const R (&U)[3], R (&o)[6]) {
            o[0] = Ik[0] * W[0] + Ik[1] * W[1] + Ik[2] * W[2] + (Ik[7] * U[2] - Ik[8] * U[1]);
            o[1] = Ik[1] * W[0] + Ik[3] * W[1] + Ik[4] * W[2] + (Ik[8] * U[0] - Ik[6] * U[2]);
            o[2] = Ik[2] * W[0] + Ik[4] * W[1] + Ik[5] * W[2] + (Ik[6] * U[1] - Ik[7] * U[0]);
            o[3] = Ik[9] * U[0] - (Ik[7] * W[2] - Ik[8] * W[1]);
            o[4] = Ik[9] * U[1] - (Ik[8] * W[0] - Ik[6] * W[2]);
            o[5] = Ik[9] * U[2] - (Ik[6] * W[1] - Ik[7] * W[0]);
        };
        imul(bw, bu, Ia);
        imul(w, u, Iv);
        // f = I a + x x* (I v_nom) + v_nom x* (I x)        x = this lane's v-like vector; the nominal lane: x = v_nom, second cross term off
        R nIv[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) nIv[r] = bc64<KKT_NOM>(Iv[r]);
        const R zw0 = nmask * nvw[0], zw1 = nmask * nvw[1], zw2 = nmask * nvw[2], zu0 = nmask * nvu[0], zu1 = nmask * nvu[1], zu2 = nmask * nvu[2];
        f[0] = Ia[0] + (w[1] * nIv[2] - w[2] * nIv[1]) + (u[1] * nIv[5] - u[2] * nIv[4]) + (zw1 * Iv[2] - zw2 * Iv[1]) + (zu1 * Iv[5] - zu2 * Iv[4]);
        f[1] = Ia[1] + (w[2] * nIv[0] - w[0] * nIv[2]) + (u[2] * nIv[3] - u[0] * nIv[5]) + (zw2 * Iv[0] - zw0 * Iv[2]) + (zu2 * Iv[3] - zu0 * Iv[5]);
        f[2] = Ia[2] + (w[0] * nIv[1] - w[1] * nIv[0]) + (u[0] * nIv[4] - u[1] * nIv[3]) + (zw0 * Iv[1] - zw1 * Iv[0]) + (zu0 * Iv[4] - zu1 * Iv[3]);
        f[3] = Ia[3] + (w[1] * nIv[5] - w[2] * nIv[4]) + (zw1 * Iv[5] - zw2 * Iv[4]);
        f[4] = Ia[4] + (w[2] * nIv[3] - w[0] * nIv[5]) + (zw2 * Iv[3] - zw0 * Iv[5]);
        f[5] = Ia[5] + (w[0] * nIv[4] - w[1] * nIv[3]) + (zw0 * Iv[4] - zw1 * Iv[3]);
        if (k < PJ - 1 && (!SHARE || k >= first)) {
#pragma unroll
            for (int r = 0; r < 6; ++r) fl[6 * k + r] = KktR<R>::to_rec(f[r]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { vw[r] = w[r]; vu[r] = u[r]; aw[r] = bw[r]; au[r] = bu[r]; }
    }
    fl[KktGrad<R>::tau(l, PJ - 1)] = KktR<R>::to_rec(f[2]);
#pragma nounroll
    for (int kv = PJ - 1; kv >= 1; --kv) {                   // F_parent += X_k^T (F_k [+ S x* F_k(nominal) in the d/dq_k lane])
        const int k = __builtin_amdgcn_readfirstlane(kv);
        const R sn = I->Sc[0][k], cs = I->Sc[1][k];
        creal* E = P.ET(k);
        creal* B = P.BT(k);
        // S x* [n; l] = [e_z x n ; e_z x l] = (-n1, n0, 0 ; -l1, l0, 0) of the nominal lane's accumulated force of link k
        const R nf0 = bc64<KKT_NOM>(f[0]), nf1 = bc64<KKT_NOM>(f[1]), nf3 = bc64<KKT_NOM>(f[3]), nf4 = bc64<KKT_NOM>(f[4]);
        const bool mine = isq && k == col;
        const R g0 = f[0] - (mine ? nf1 : KR(0.0)), g1 = f[1] + (mine ? nf0 : KR(0.0)), g3 = f[3] - (mine ? nf4 : KR(0.0)), g4 = f[4] + (mine ? nf3 : KR(0.0));
        const R n0 = cs * g0 - sn * g1, n1 = sn * g0 + cs * g1, n2 = f[2];
        const R l0 = cs * g3 - sn * g4, l1 = sn * g3 + cs * g4, l2 = f[5];
        R fp[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R own_n = KktR<R>::from_rec(fl[6 * (k - 1) + r]), own_l = KktR<R>::from_rec(fl[6 * (k - 1) + 3 + r]);      // (read by every lane; SHARE: discarded below the column's joint)
            if (SHARE && k - 1 < first) { own_n = KR(0.0); own_l = KR(0.0); }
            fp[r] = own_n + E[r] * n0 + E[3 + r] * n1 + E[6 + r] * n2 + B[r] * l0 + B[3 + r] * l1 + B[6 + r] * l2;
            fp[3 + r] = own_l + E[r] * l0 + E[3 + r] * l1 + E[6 + r] * l2;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) f[r] = fp[r];
        fl[KktGrad<R>::tau(l, k - 1)] = KktR<R>::to_rec(f[2]);
    }
}

// Output staging (floats, in the group's records once the last sweep is over): [Q R | Q_last] [-A -B] [q r | q_last] [c_0] [c_{k+1}]
typedef __attribute__((address_space(3))) float kkt_lds_f;
constexpr int ST_G = 0, ST_Q1 = ST_G + 14 * 14 + 7 * 7, ST_C = ST_Q1 + 14 * 14, ST_g = ST_C + 14 * 14 + 14 * 7, ST_g1 = ST_g + 21, ST_c0 = ST_g1 + 14,
              ST_c1 = ST_c0 + 14, ST_END = ST_c1 + 14;
// records of a group: round 0 runs PJ + 4 records of R; round 1 either 2 PJ records of R (differences) or KktGrad<R>::RECS records of KktR<R>::rec (analytic: float; packed: float pairs)
constexpr int KKT_R0 = PJ + 4;
__host__ __device__ constexpr int kkt_rec_lanes(bool analytic) { return analytic ? KKT_R0 : KKT_RL; }
// elements of type R a group's record region holds: the round-0 records, the analytic round's 15 (packed: 14) records and the float staging area (one knot at a time) all fit
// (double: the 11 (analytic) / 14 round-0 records are the largest of the three, as before; float: the staging area is — 3,192 B per group; packed: the 14 pair records, 4,144 B)
template <typename R> __host__ __device__ constexpr int kkt_rec_elems(bool analytic) {
    const int r0 = kkt_rec_lanes(analytic) * RN_ROWS;
    const int r1 = (int)((KktGrad<R>::RECS * RN_ROWS * sizeof(typename KktR<R>::rec) + sizeof(R) - 1) / sizeof(R)), st = (int)((ST_END * sizeof(float) + sizeof(R) - 1) / sizeof(R));
    return r0 > r1 ? (r0 > st ? r0 : st) : (r1 > st ? r1 : st);
}
static_assert(kkt_rec_elems<double>(true) == KKT_R0 * RN_ROWS && kkt_rec_elems<double>(false) == KKT_RL * RN_ROWS, "the double build's LDS footprint is round 5's");
template <int LEN>
__device__ __forceinline__ void kkt_copy_out(float* dst, kkt_lds_f* src, int l) {
#pragma unroll
    for (int e = 0; e + KKT_GL <= LEN; e += KKT_GL) dst[e + l] = src[e + l];
    if (LEN % KKT_GL != 0 && l < LEN % KKT_GL) dst[LEN - LEN % KKT_GL + l] = src[LEN - LEN % KKT_GL + l];
}

#ifndef KKT_WAVES_ANALYTIC
#define KKT_WAVES_ANALYTIC 2
#endif
#ifndef KKT_WAVES_F32
#define KKT_WAVES_F32 2      // (the float build at three wavefronts per SIMD: 168 VGPRs, 17 spilled, 0.322 ms per 1024 x 127 knots; at two: 191, none, 0.303; double: 0.328)
#endif
// R = double | float: a lane group of 16 = one (trajectory, knot) pair per trip; R = kkt_f2: TWO — items 2 i and 2 i + 1 of the wavefront's eight — in the halves of every value.
template <bool ANALYTIC, typename R = double>
__global__ __launch_bounds__(KKT_THREADS, sizeof(R) == 4 ? KKT_WAVES_F32 : ANALYTIC ? KKT_WAVES_ANALYTIC : 2) void generate_kkt_kernel(KktArgsT<typename KktR<R>::scalar> a) {
    typedef KktR<R> T;
    typedef typename T::scalar S;
    constexpr int KP = T::KP;
    static_assert(ANALYTIC || T::is_double, "the difference quotients need float64");
    typedef typename KktLds<R>::vr kkt_lds_vd;
    typedef typename KktLds<R>::item kkt_lds_item;
    typedef typename PlantC<S>::creal creal;
    constexpr int n = 2 * PJ, m = PJ, nn = n * n, mm = m * m, nm = n * m;
    __shared__ KktItemLds<R> sI[KKT_ITEMS];
    constexpr int RL = kkt_rec_lanes(ANALYTIC);             // round-0 records per group: 11 (analytic: 13.0 KB per wavefront in double) or 14 (16.6 KB)
    constexpr int RE = kkt_rec_elems<R>(ANALYTIC);          // elements of a group's record region (float: the staging area decides, 14.4 KB per wavefront; packed: 20.6 KB)
    __shared__ R sF[KKT_ITEMS][RE];                         // the recursion records
    static_assert(sizeof(KktItemLds<R>) * KKT_ITEMS + sizeof(R) * KKT_ITEMS * RE <= (ANALYTIC && KP == 1 ? 16384 : 20480), "ten / eight wavefronts per CU");
    // The model tables are read with RUNTIME joint indices.  With compile-time indices (unrolled sweeps) all table entries are
    // loop-invariant loads that the compiler hoists into registers: 512 VGPR + AGPR and scratch.
    const int lane = threadIdx.x, gi = lane / KKT_GL, l = lane - gi * KKT_GL;
    kkt_lds_item* I = (kkt_lds_item*)&sI[gi];
    kkt_lds_vd* recs = (kkt_lds_vd*)&sF[gi][0];
    auto rec = [&](int j) -> kkt_lds_vd* { return recs + j * RN_ROWS; };                // record of lane j of this group
    kkt_lds_vd* fl = rec(l < RL ? l : 0);                    // (lanes beyond the records never touch theirs)
    kkt_lds_f* st = (kkt_lds_f*)&sF[gi][0];
    const PlantC<S> P{reinterpret_cast<creal*>(reinterpret_cast<unsigned long long>(a.plant))};
    const int N = a.N;
    const long total = (long)a.batch * (N - 1);
    // A wavefront's trips cover CONSECUTIVE groups of four (packed: eight) knots (not a grid stride): the knots' pieces of g (84 B), c (56 B), G (980 B) and
    // C (1176 B) are then neighbours in memory and most 128-byte lines are completed inside one L2 instead of leaving two XCDs as partial writes.
    constexpr int PER_TRIP = KKT_ITEMS * KP;
    const long groups = (total + PER_TRIP - 1) / PER_TRIP, per = (groups + gridDim.x - 1) / gridDim.x;
    const long g_begin = (long)blockIdx.x * per, g_end = g_begin + per < groups ? g_begin + per : groups;
    for (long grp = g_begin; grp < g_end; ++grp) {
        const long base = grp * PER_TRIP + (long)gi * KP;
        bool live[KP];                                      // (a half without a knot recomputes the last one and writes nothing)
        int bb[KP], kk[KP];
        const float* xu[KP];
#pragma unroll
        for (int hf = 0; hf < KP; ++hf) {
            live[hf] = base + hf < total;
            const long item = live[hf] ? base + hf : total - 1;
            if (hf == 0 || !live[hf]) {
                bb[hf] = (int)(item / (N - 1));              // (a 32-bit division where the knot count allows, behind a wave-uniform test, measured SLOWER: 0.336 against 0.330 ms in double)
                kk[hf] = (int)(item - (long)bb[hf] * (N - 1));
            } else {                                        // the knot behind the first half's
                const bool wrap = kk[0] + 1 == N - 1;
                bb[hf] = bb[0] + (wrap ? 1 : 0);
                kk[hf] = wrap ? 0 : kk[0] + 1;
            }
            xu[hf] = a.xu + (size_t)bb[hf] * ((size_t)(n + m) * N - m) + (size_t)kk[hf] * (n + m);      // x_k, u_k, x_{k+1}
        }
        if (l < n) I->Xq[l] = T::mk((S)xu[0][l], (S)xu[KP - 1][l]);
        if (l < m) {
            I->U[l] = T::mk((S)xu[0][n + l], (S)xu[KP - 1][n + l]);
            double sn_[KP], cs_[KP];                          // (seven sine / cosine pairs per knot: in double in every build, rounded to R)
#pragma unroll
            for (int hf = 0; hf < KP; ++hf) {
                if (KKT_ABLATE & 4) { sn_[hf] = (double)xu[hf][l]; cs_[hf] = 1.0 - sn_[hf]; } else
                kkt_sincos((double)xu[hf][l], sn_[hf], cs_[hf]);
            }
            I->Sc[0][l] = T::mk((S)sn_[0], (S)sn_[KP - 1]);
            I->Sc[1][l] = T::mk((S)cs_[0], (S)cs_[KP - 1]);
        }
        __syncthreads();
        // ---- round 0: lanes 0..6 inertia-matrix columns ID(q, 0, e_l), lane 7 bias ID(q, qd, 0), lanes 8..10 the pose sweeps ----
        R a6w[3], a6u[3];
        if (l < PJ + 4) {
            RneaTask<R> t;
            t.sj = -1; t.pj = -1; t.qdscale = (l == PJ) ? KR(1.0) : KR(0.0); t.knot_qdd = false; t.unit = l < PJ ? l : -1; t.base = l > PJ ? l - PJ - 1 : -1;
            if (!(KKT_ABLATE & 16)) rnea(P, fl, I, t, a6w, a6u);
#pragma unroll
            for (int r = 0; r < 3; ++r) { fl[RN_AW + r] = a6w[r]; fl[RN_AU + r] = a6u[r]; }
        }
        __syncthreads();
        // ---- Minv (column l through a Cholesky solve of the symmetrised M), qdd_l = Minv_l . (u - bias)  (Minv is symmetric: row l = column l),
        //      end-effector position, Jacobian column l, cost gradient entries ----
        if (l < PJ && !(KKT_ABLATE & 2)) {
            R Lm[PJ][PJ], rd[PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i)
#pragma unroll
                for (int jj = 0; jj <= i; ++jj) {
                    R sv = KR(0.5) * (rec(jj)[RN_TAU(i)] + rec(i)[RN_TAU(jj)]);      // M[i][jj] = tau_i of lane jj
#pragma unroll
                    for (int t = 0; t < jj; ++t) sv -= Lm[i][t] * Lm[jj][t];
                    if (i == jj) {
                        // 1 / sqrt(pivot) from the hardware estimate + two Newton steps (full double precision for these O(1) pivots): the
                        // correctly rounded sqrt and division of the textbook form are ~30 instructions per pivot
                        R y = T::rsq(sv);
                        y = __builtin_elementwise_fma(y * KR(0.5), __builtin_elementwise_fma(-sv * y, y, KR(1.0)), y);
                        if constexpr (T::is_double) y = __builtin_elementwise_fma(y * KR(0.5), __builtin_elementwise_fma(-sv * y, y, KR(1.0)), y);      // (float: 1 ulp estimate + one Newton step)
                        rd[i] = y;
                        Lm[i][i] = sv * y;
                    }
                    else Lm[i][jj] = sv * rd[jj];
                }
            R y[PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i) {
                R sv = (i == l) ? KR(1.0) : KR(0.0);
#pragma unroll
                for (int t = 0; t < i; ++t) sv -= Lm[i][t] * y[t];
                y[i] = sv * rd[i];
            }
#pragma unroll
            for (int i = PJ - 1; i >= 0; --i) {
                R sv = y[i];
#pragma unroll
                for (int t = i + 1; t < PJ; ++t) sv -= Lm[t][i] * y[t];
                y[i] = sv * rd[i];
            }
            R qdd = KR(0.0);
#pragma unroll
            for (int i = 0; i < PJ; ++i) {
                I->Minv[i][l] = y[i];
                qdd += y[i] * (I->U[i] - rec(PJ)[RN_TAU(i)]);          // bias_i = tau_i of lane 7
            }
            I->Qdd[l] = qdd;
            // pose of the last link from the three base-acceleration sweeps (lanes 8..10): their final acceleration is [W_i ; V_i] =
            // [R e_i ; R (e_i x p)], R = rotation world -> link.  Row i of R^T is W_i, so R^T x = (W_0.x, W_1.x, W_2.x);
            // e_x x p = (0, -pz, py), e_y x p = (pz, 0, -px).
            R W[3][3], V0[3], V1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) W[i][r] = rec(PJ + 1 + i)[RN_AW + r];
#pragma unroll
            for (int r = 0; r < 3; ++r) { V0[r] = rec(PJ + 1)[RN_AU + r]; V1[r] = rec(PJ + 2)[RN_AU + r]; }
            R ee[3], J[3];
            ee[0] = -(W[2][0] * V1[0] + W[2][1] * V1[1] + W[2][2] * V1[2]);
            ee[1] = W[2][0] * V0[0] + W[2][1] * V0[1] + W[2][2] * V0[2];
            ee[2] = -(W[1][0] * V0[0] + W[1][1] * V0[1] + W[1][2] * V0[2]);
            // Jacobian column l = R^T (linear velocity of the last link's origin for qd = e_l) = R^T au of this lane's own sweep
#pragma unroll
            for (int r = 0; r < 3; ++r) J[r] = W[r][0] * a6u[0] + W[r][1] * a6u[1] + W[r][2] * a6u[2];
            const float* goal[KP];
#pragma unroll
            for (int hf = 0; hf < KP; ++hf) goal[hf] = a.eePos_traj + ((size_t)bb[hf] * N + kk[hf]) * 6;
            R s0 = KR(0.0), s1 = KR(0.0);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                s0 += J[r] * (ee[r] - T::mk((S)goal[0][r], (S)goal[KP - 1][r]));
                s1 += J[r] * (ee[r] - T::mk((S)goal[0][6 + r], (S)goal[KP - 1][6 + r]));       // goal of knot k+1: used by the last block only
            }
            I->Gq[l] = s0;
            I->Gq1[l] = s1;
        }
        __syncthreads();
        // ---- round 1: lanes 0..6 ID(q + h e_l, qd, qdd), 7..13 ID(q, qd + h e_(l-7), qdd); each lane then owns column l of
        //      [dqdd/dq, dqdd/dqd] = -Minv (ID(. + h e) - u) / h  and writes column l of A and Q (lanes 0..6: of B and R too) ----
        // analytic gradient (default): 15 lanes — 14 columns + the nominal recursion in lane 14; records re-used as 15 x RN_ROWS floats (packed: float pairs)
        typename KktRecLds<R>::vf* flf = (typename KktRecLds<R>::vf*)recs + (l <= KKT_NOM ? KktGrad<R>::rec(l) : 0) * RN_ROWS;
        if (ANALYTIC && l <= KKT_NOM && !(KKT_ABLATE & 1)) rnea_grad<R>(P, flf, I, l);
        R colv[PJ];
        if (l < n) {
            R d[PJ];
            if constexpr (ANALYTIC) {
#pragma unroll
                for (int i = 0; i < PJ; ++i) d[i] = -T::from_rec(flf[KktGrad<R>::tau(l, i)]);
            } else {
                RneaTask<R> t;
                t.sj = l < PJ ? l : -1; t.pj = l < PJ ? -1 : l - PJ; t.qdscale = KR(1.0); t.knot_qdd = true; t.unit = -1; t.base = -1;
                if (!(KKT_ABLATE & 1)) rnea(P, fl, I, t, a6w, a6u);
#pragma unroll
                for (int i = 0; i < PJ; ++i) d[i] = (fl[RN_TAU(i)] - I->U[i]) * (-KR(1.0) / KR(KKT_FD_H));
            }
            asm volatile("" ::: "memory");                // (the float staging stores below reuse the records: keep them behind these loads)
#pragma unroll
            for (int i = 0; i < PJ; ++i) {
                R sv = KR(0.0);
#pragma unroll
                for (int tt = 0; tt < PJ; ++tt) sv += I->Minv[i][tt] * d[tt];
                colv[i] = sv;
            }
        }
        // The knot's outputs are STAGED in the group's (now free) records as float, in the order they have in memory, and
        // copied out by all 16 lanes in 64-byte runs below.  Written straight from here — a lane per column, 14 lanes 56 bytes
        // apart per store — the ~60 stores per lane were a fifth of the kernel's time (one cache line per lane and store).
        // Packed build: one knot of the pair at a time through the same staging area.
#pragma unroll
        for (int hf = 0; hf < KP; ++hf) {
            const int k = kk[hf], b = bb[hf];
            if (l < n && !(KKT_ABLATE & 8)) {
                const S dt = a.dt;
                const S gql = l < PJ ? T::get(I->Gq[l], hf) : S(0.0), gq1l = l < PJ ? T::get(I->Gq1[l], hf) : S(0.0);
                // column l (column-major):  A = I + dt [[0, I], [dqdd/dq, dqdd/dqd]],  Q = blkdiag(g g^T, QD I)
#pragma unroll
                for (int r = 0; r < n; ++r) {
                    S av = (r == l) ? S(1.0) : S(0.0);
                    if (r < PJ) av += (l == r + PJ) ? dt : S(0.0);
                    else av += dt * T::get(colv[r - PJ], hf);
                    st[ST_C + l * n + r] = (float)(-av);
                    S qv, q1;
                    if (r < PJ) { qv = T::get(I->Gq[r], hf) * gql; q1 = T::get(I->Gq1[r], hf) * gq1l; }
                    else qv = q1 = (r == l) ? a.qd_cost : S(0.0);
                    st[ST_G + l * n + r] = (float)qv;
                    st[ST_Q1 + l * n + r] = (float)q1;
                }
                if (l < m) {
#pragma unroll
                    for (int r = 0; r < n; ++r) st[ST_C + nn + l * n + r] = (float)(-(r < PJ ? S(0.0) : dt * T::get(I->Minv[r - PJ][l], hf)));      // B = dt [0; Minv]
#pragma unroll
                    for (int r = 0; r < m; ++r) st[ST_G + nn + l * m + r] = (float)(r == l ? a.r_cost : S(0.0));
                    st[ST_g + n + l] = (float)(a.r_cost * T::get(I->U[l], hf));
                }
                const S qdl = T::get(I->Xq[l < PJ ? l + PJ : l], hf);              // qd_{l mod 7}
                st[ST_g + l] = (float)(l < PJ ? gql : a.qd_cost * qdl);
                st[ST_g1 + l] = (float)(l < PJ ? gq1l : a.qd_cost * qdl);  // last block only (evaluated at x_{N-2}: iiwa_eepos_plant.cuh:407)
                // integrator defect c_{k+1} = x_{k+1} - (x_k + dt [qd; qdd]);  c_0 = x_0 - x_s
                const S pred = l < PJ ? T::get(I->Xq[l], hf) + dt * qdl : qdl + dt * T::get(I->Qdd[l - PJ], hf);
                st[ST_c1 + l] = (float)((S)xu[hf][(n + m) + l] - pred);
                if (k == 0) st[ST_c0 + l] = (float)((S)xu[hf][l] - (S)a.xs[(size_t)b * n + l]);
            }
            __syncthreads();
            if (live[hf] && !(KKT_ABLATE & 8)) {
                float* G = a.G + (size_t)b * ((size_t)(nn + mm) * N - mm) + (size_t)(nn + mm) * k;
                float* Cm = a.C + (size_t)b * (size_t)(nn + nm) * (N - 1) + (size_t)(nn + nm) * k;
                float* g = a.g + (size_t)b * ((size_t)(n + m) * N - m) + (size_t)(n + m) * k;
                float* c = a.c + (size_t)b * (size_t)n * N + (size_t)n * (k + 1);
                kkt_copy_out<nn + mm>(G, st + ST_G, l);
                kkt_copy_out<nn + nm>(Cm, st + ST_C, l);
                kkt_copy_out<n + m>(g, st + ST_g, l);
                kkt_copy_out<n>(c, st + ST_c1, l);
                if (k == N - 2) {                                 // the last block: Q_{N-1}, q_{N-1} follow R_{N-2}, r_{N-2} in memory
                    kkt_copy_out<nn>(G + nn + mm, st + ST_Q1, l);
                    kkt_copy_out<n>(g + n + m, st + ST_g1, l);
                }
                if (k == 0) kkt_copy_out<n>(c - n, st + ST_c0, l);
            }
            __syncthreads();
        }
    }
}
#undef KR

}  // namespace mpcg
