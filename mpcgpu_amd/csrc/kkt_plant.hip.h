// kkt_plant.hip.h — batched KKT block assembly for a fixed-base serial chain of revolute joints (IIWA-14): the HIP twin of
// generate_kkt_submatrices (reference include/common/kkt.cuh:22-163) together with the plant functions it calls
// (include/dynamics/iiwa/iiwa_eepos_plant.cuh: forwardDynamicsAndGradient :127-155, trackingCostGradientAndHessian :307-390,
// _lastblock :392-411) and the Euler integrator (include/common/integrator.cuh:56-104, 143-162).  SURVEY.md §8f row 4.
//
// The reference runs GRiD-generated, robot-specific code (10 k lines of unrolled recursions) with one thread block per
// knot.  Here the robot is DATA (struct PlantDev: spatial transforms as constant + sin + cos parts, spatial inertias,
// homogeneous transforms) and the algorithms are the generic ones, mapped for a 64-wide wavefront:
//   one wavefront per FOUR (trajectory, knot) pairs, 16 lanes each; a lane runs a whole recursive Newton-Euler pass:
//     round 0  lanes 0..6: columns of the joint-space inertia matrix M = ID(q, 0, e_j);  lane 7: bias c = ID(q, qd, 0)
//              then lanes 0..6: column j of Minv by a Cholesky solve (7x7, redundantly factorised per lane);  qdd = Minv (u - c)
//     round 1  lanes 0..6: ID(q + h e_j, qd, qdd), lanes 7..13: ID(q, qd + h e_j, qdd);  lane 14: forward kinematics, end-effector
//              position and geometric Jacobian z_j x (p_ee - p_j).  Every lane stores tau(+h) - u: ONE-SIDED differences of the
//              inverse dynamics — the nominal value is known without evaluating it, ID(q, qd, qdd) = u because qdd = Minv (u - c)
//              (round 3; rounds 1 and 2 used to be the central pair +-h: a third of the kernel's recursion time for accuracy the
//              float outputs cannot hold — with h = 3e-8 in float64 the entries of A differ from the central-difference values by
//              1.3e-7, the rounding of a float near 1).
//     phase 4  dqdd/d(q,qd) = -Minv dID,  A, B, integrator defect, Gauss-Newton cost blocks, written as float in the
//              reference's dense layouts (column-major blocks, C = -A, -B).
// Arithmetic is float64 inside (the difference quotients need it; the MI355X has the fp64 rate to spare: the whole kernel is
// ~40 kflop per knot), results are rounded to float on the way out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

constexpr int PJ = 7;                    // joints of the compiled specialisation (state 2 PJ, control PJ)
constexpr int KKT_LANES = 64;            // one wavefront per KKT_ITEMS (trajectory, knot) pairs
constexpr int RN_ROWS = 6 * PJ + 1;        // record: link forces [PJ][6] (+1: an odd row count = conflict-free 8-byte accesses at lane stride)
__host__ __device__ constexpr int RN_TAU(int k) { return 6 * k + 2; }             // tau_k overwrites row 2 of link k's force once consumed

struct PlantDev {                        // all row-major 3x3 unless noted
    double E0[PJ][9], Es[PJ][9], Ec[PJ][9];      // rotation block of X_k(q_k) = E0 + Es sin q_k + Ec cos q_k
    double B0[PJ][9], Bs[PJ][9], Bc[PJ][9];      // lower-left block of X_k (= -E r x)
    double I[PJ][36];                            // spatial inertia, row-major 6x6
    double R0[PJ][9], Rs[PJ][9], Rc[PJ][9];      // rotation of the homogeneous transform link k -> parent
    double p[PJ][3];                             // its translation
};

struct KktArgs {
    const PlantDev* plant;
    const float* eePos_traj;             // [batch][N][6]
    const float* xs;                     // [batch][n]
    const float* xu;                     // [batch][(n+m)N - m]
    float* G; float* C; float* g; float* c;
    int N; int batch;
    double dt, qd_cost, r_cost;
};

// The model tables are read through the CONSTANT address space (same 64-bit address as the global pointer): loads from it are
// invariant by definition, so a uniform address makes them scalar loads (s_load, scalar cache).  Through the plain global
// pointer the compiler must assume the kernel's own stores may clobber the table and emits ~1,300 vector loads per knot
// (rocprofv3: SQ_INSTS_VMEM_RD; waves waited on memory half of their cycles).
typedef const __attribute__((address_space(4))) double cdouble;
struct PlantC {
    cdouble* base;
    __device__ __forceinline__ cdouble* at(size_t byte_off, int k, int per) const { return base + byte_off / sizeof(double) + (size_t)k * per; }
    __device__ __forceinline__ cdouble* E0(int k) const { return at(offsetof(PlantDev, E0), k, 9); }
    __device__ __forceinline__ cdouble* Es(int k) const { return at(offsetof(PlantDev, Es), k, 9); }
    __device__ __forceinline__ cdouble* Ec(int k) const { return at(offsetof(PlantDev, Ec), k, 9); }
    __device__ __forceinline__ cdouble* B0(int k) const { return at(offsetof(PlantDev, B0), k, 9); }
    __device__ __forceinline__ cdouble* Bs(int k) const { return at(offsetof(PlantDev, Bs), k, 9); }
    __device__ __forceinline__ cdouble* Bc(int k) const { return at(offsetof(PlantDev, Bc), k, 9); }
    __device__ __forceinline__ cdouble* I(int k) const { return at(offsetof(PlantDev, I), k, 36); }
    __device__ __forceinline__ cdouble* R0(int k) const { return at(offsetof(PlantDev, R0), k, 9); }
    __device__ __forceinline__ cdouble* Rs(int k) const { return at(offsetof(PlantDev, Rs), k, 9); }
    __device__ __forceinline__ cdouble* Rc(int k) const { return at(offsetof(PlantDev, Rc), k, 9); }
    __device__ __forceinline__ cdouble* p(int k) const { return at(offsetof(PlantDev, p), k, 3); }
};

__device__ __forceinline__ void mat3(double (&M)[9], cdouble* c0, cdouble* cs, cdouble* cc, double s, double c) {
#pragma unroll
    for (int e = 0; e < 9; ++e) M[e] = c0[e] + cs[e] * s + cc[e] * c;
}

// tau = ID(q, qd, qdd) without gravity (gato_plant::GRAVITY = 0, iiwa_eepos_plant.cuh:53).  Inputs, the link forces that wait
// for the backward sweep, sin / cos and the result all live in this lane's column `fl` of an LDS record (rows RN_*), and both
// sweeps are RUNTIME loops over the joints.  Measured alternatives on gfx950 (hipcc 7.2): everything in registers with unrolled
// sweeps = 3.4 KB of scratch per lane (the 84 force registers, plus the seven E_k / B_k pairs the compiler keeps from the
// forward sweep for the backward one instead of recomputing them: 252 doubles); plain (non-volatile) LDS accesses get
// store-forwarded back into registers.  This form compiles one joint body in ~225 registers without scratch; the record (LDS capacity)
// is then what bounds the resident wavefronts per CU.
//   sin / cos of the joint angles come from a table sc[variant][2][PJ] shared by the lanes (variant 0: q, 1: q + h e_j, 2: q - h e_j;
//   ONE sincos call per knot fills it — every recursion used to recompute all seven, 29 % of the kernel's VALU instructions):
//   joint k uses variant (k == sj ? sv : 0).
//   qd_k = qdscale * xqd[k] + (k == prow ? ph : 0)   (xqd shared by the lanes),   qdd_k = qdd ? qdd[k] : (k == unit ? 1 : 0)
__device__ __forceinline__ void rnea(const PlantC& P, volatile double* fl, const double* sc, int sj, int sv, const double* xqd, double qdscale,
                                     int prow, double ph, const double* qdd, int unit) {
    double vw[3] = {0, 0, 0}, vu[3] = {0, 0, 0}, aw[3] = {0, 0, 0}, au[3] = {0, 0, 0};
#pragma nounroll
    for (int kv = 0; kv < PJ; ++kv) {
        const int k = __builtin_amdgcn_readfirstlane(kv);    // uniform by construction; said so, the model tables come through s_load
        const double qdk = qdscale * xqd[k] + (k == prow ? ph : 0.0);
        const double qddk = qdd ? qdd[k] : (k == unit ? 1.0 : 0.0);
        const double* sck = sc + (k == sj ? sv : 0) * (2 * PJ) + k;
        const double sn = sck[0], cs = sck[PJ];
        double E[9], B[9];
        mat3(E, P.E0(k), P.Es(k), P.Ec(k), sn, cs);
        mat3(B, P.B0(k), P.Bs(k), P.Bc(k), sn, cs);
        double w[3], u[3], bw[3], bu[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {                    // v = X v_parent, a = X a_parent
            w[r] = E[3 * r] * vw[0] + E[3 * r + 1] * vw[1] + E[3 * r + 2] * vw[2];
            u[r] = B[3 * r] * vw[0] + B[3 * r + 1] * vw[1] + B[3 * r + 2] * vw[2] + E[3 * r] * vu[0] + E[3 * r + 1] * vu[1] + E[3 * r + 2] * vu[2];
            bw[r] = E[3 * r] * aw[0] + E[3 * r + 1] * aw[1] + E[3 * r + 2] * aw[2];
            bu[r] = B[3 * r] * aw[0] + B[3 * r + 1] * aw[1] + B[3 * r + 2] * aw[2] + E[3 * r] * au[0] + E[3 * r + 1] * au[1] + E[3 * r + 2] * au[2];
        }
        w[2] += qdk;                                     // + S qd, S = e_z (angular)
        bw[2] += qddk;
        // + v x (S qd): column 2 of crm(v) times qd
        bw[0] += w[1] * qdk; bw[1] -= w[0] * qdk;
        bu[0] += u[1] * qdk; bu[1] -= u[0] * qdk;
        // f = I a + v x* (I v)
        double Ia[6], Iv[6];
        cdouble* Ik = P.I(k);
        const double v6[6] = {w[0], w[1], w[2], u[0], u[1], u[2]}, a6[6] = {bw[0], bw[1], bw[2], bu[0], bu[1], bu[2]};
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double sa = 0, sv = 0;
#pragma unroll
            for (int cc_ = 0; cc_ < 6; ++cc_) { sa += Ik[6 * r + cc_] * a6[cc_]; sv += Ik[6 * r + cc_] * v6[cc_]; }
            Ia[r] = sa; Iv[r] = sv;
        }
        // crf(v) h = [w x n + u x l ; w x l],  h = [n; l]
        volatile double* f = fl + k * 6;
        f[0] = Ia[0] + (w[1] * Iv[2] - w[2] * Iv[1]) + (u[1] * Iv[5] - u[2] * Iv[4]);
        f[1] = Ia[1] + (w[2] * Iv[0] - w[0] * Iv[2]) + (u[2] * Iv[3] - u[0] * Iv[5]);
        f[2] = Ia[2] + (w[0] * Iv[1] - w[1] * Iv[0]) + (u[0] * Iv[4] - u[1] * Iv[3]);
        f[3] = Ia[3] + (w[1] * Iv[5] - w[2] * Iv[4]);
        f[4] = Ia[4] + (w[2] * Iv[3] - w[0] * Iv[5]);
        f[5] = Ia[5] + (w[0] * Iv[4] - w[1] * Iv[3]);
#pragma unroll
        for (int r = 0; r < 3; ++r) { vw[r] = w[r]; vu[r] = u[r]; aw[r] = bw[r]; au[r] = bu[r]; }
    }
    double fc[6];                                        // force of the link being folded into its parent
#pragma unroll
    for (int r = 0; r < 6; ++r) fc[r] = fl[(PJ - 1) * 6 + r];   // tau_6 = row 2 of the last link's force: already in place
#pragma nounroll
    for (int kv = PJ - 1; kv >= 1; --kv) {               // f_parent += X^T f = [E^T n + B^T l ; E^T l]
        const int k = __builtin_amdgcn_readfirstlane(kv);
        const double* sck = sc + (k == sj ? sv : 0) * (2 * PJ) + k;
        const double sk = sck[0], ck = sck[PJ];
        double E[9], B[9];
        mat3(E, P.E0(k), P.Es(k), P.Ec(k), sk, ck);
        mat3(B, P.B0(k), P.Bs(k), P.Bc(k), sk, ck);
        double fp[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            fp[r] = fl[(k - 1) * 6 + r] + E[r] * fc[0] + E[3 + r] * fc[1] + E[6 + r] * fc[2] + B[r] * fc[3] + B[3 + r] * fc[4] + B[6 + r] * fc[5];
            fp[3 + r] = fl[(k - 1) * 6 + 3 + r] + E[r] * fc[3] + E[3 + r] * fc[4] + E[6 + r] * fc[5];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) fc[r] = fp[r];
        fl[RN_TAU(k - 1)] = fc[2];                       // (link k-1's own force row 2: consumed just above)
    }
}

constexpr int KKT_THREADS = KKT_LANES;
constexpr int KKT_ITEMS = 4;             // (trajectory, knot) pairs per wavefront: 16 lanes each
constexpr int KKT_GL = KKT_LANES / KKT_ITEMS;
constexpr int KKT_KIN_LANE = 2 * PJ;     // lane of a group after the 14 finite-difference tasks: forward kinematics + Jacobian
constexpr double KKT_FD_H = 3e-8;             // one-sided difference step (truncation h/2 |f''| ~ roundoff eps |f| / h in float64)

struct KktItemLds {                      // per-knot scratch in LDS (2.1 KB; LDS capacity is what bounds the resident wavefronts per CU)
    double M[PJ][PJ], Minv[PJ][PJ], Bias[PJ], Qdd[PJ];
    // (the central differences ID(. + h e_j) - ID(. - h e_j) wait in rows 0..6 of the finished recursion's record of lane j / 7 + j;
    //  dqdd/dq overwrites M, which is dead after the Cholesky factorisation)
    double Dqd[PJ][PJ];
    double J[3][PJ], Ee[3], Gq[PJ], Gq1[PJ];
    double Xq[2 * PJ], U[PJ];            // [q; qd], u of this knot
    double Sc[2][2][PJ];                 // sin / cos of q, q + h e_j
};

__global__ __launch_bounds__(KKT_THREADS, 2) void generate_kkt_kernel(KktArgs a) {
    constexpr int n = 2 * PJ, m = PJ, nn = n * n, mm = m * m, nm = n * m;
    __shared__ KktItemLds sI[KKT_ITEMS];
    __shared__ double sF[KKT_LANES][RN_ROWS];               // per-lane record of the recursion: link forces (22 KB)
    // The model tables are read with RUNTIME joint indices.  With compile-time indices (unrolled sweeps) all 840 doubles are
    // loop-invariant loads that the compiler hoists into registers: 512 VGPR + AGPR and scratch.
    const int lane = threadIdx.x, gi = lane / KKT_GL, l = lane - gi * KKT_GL;
    KktItemLds& I = sI[gi];
    volatile double* fl = &sF[lane][0];
    const PlantC P{reinterpret_cast<cdouble*>(reinterpret_cast<unsigned long long>(a.plant))};
    const int N = a.N;
    const long total = (long)a.batch * (N - 1);
    for (long base = (long)blockIdx.x * KKT_ITEMS; base < total; base += (long)gridDim.x * KKT_ITEMS) {
        const bool live = base + gi < total;                // (a group without a knot recomputes the last one and writes nothing)
        const long item = live ? base + gi : total - 1;
        const int b = (int)(item / (N - 1)), k = (int)(item - (long)b * (N - 1));
        const float* xu = a.xu + (size_t)b * ((size_t)(n + m) * N - m) + (size_t)k * (n + m);      // x_k, u_k, x_{k+1}
        if (l < n) I.Xq[l] = (double)xu[l];
        if (l < m) I.U[l] = (double)xu[n + l];
        // sin / cos table through one sincos: lanes 0..13 -> q_j and q_j + h
        {
            const int v = l / PJ, j = l % PJ;
            if (l < 2 * PJ) {
                double sn_, cs_;
                sincos((double)xu[j] + (v == 0 ? 0.0 : KKT_FD_H), &sn_, &cs_);
                I.Sc[v][0][j] = sn_;
                I.Sc[v][1][j] = cs_;
            }
        }
        __syncthreads();
        // ---- two rounds through ONE instance of the recursion (a runtime loop: a second inlined copy doubles the register
        //      pressure).  Round 0: lanes 0..6 inertia-matrix columns ID(q, 0, e_l), lane 7 bias ID(q, qd, 0), then Minv and qdd.
        //      Round 1: lanes 0..6 ID(q + h e_j, qd, qdd), 7..13 ID(q, qd + h e_j, qdd), each minus the nominal torque u; lane 14: kinematics. ----
#pragma nounroll
        for (int round = 0; round < 2; ++round) {
            const bool fd = round > 0;
            if (l < (fd ? 2 * PJ : PJ + 1)) {
                const int kind = l / PJ, jj = l - kind * PJ;            // fd: kind 0 perturbs q_jj, kind 1 qd_jj
                rnea(P, fl, &I.Sc[0][0][0], (fd && kind == 0) ? jj : -1, round, I.Xq + PJ, (fd || l == PJ) ? 1.0 : 0.0,
                     (fd && kind == 1) ? jj : -1, KKT_FD_H, fd ? I.Qdd : nullptr, l);
#pragma unroll
                for (int i = 0; i < PJ; ++i) {
                    const double t = fl[RN_TAU(i)];
                    if (round == 0) { if (l < PJ) I.M[i][l] = t; else I.Bias[i] = t; }
                    else fl[i] = t - I.U[i];                         // ID(. + h e_j) - ID(.) with ID(q, qd, qdd) = u  (the sweep is over: the record is free)
                }
            } else if (round == 1 && l == KKT_KIN_LANE) {
                // forward kinematics; joint origins and axes wait in this lane's (otherwise unused) record
                double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
#pragma nounroll
                for (int jv = 0; jv < PJ; ++jv) {
                    const int jn = __builtin_amdgcn_readfirstlane(jv);
                    double H[9];
                    const double s_ = I.Sc[0][0][jn], c_ = I.Sc[0][1][jn];
                    mat3(H, P.R0(jn), P.Rs(jn), P.Rc(jn), s_, c_);
                    double Rn[9];
                    cdouble* pj = P.p(jn);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        pos[r] += R[3 * r] * pj[0] + R[3 * r + 1] * pj[1] + R[3 * r + 2] * pj[2];
#pragma unroll
                        for (int cc_ = 0; cc_ < 3; ++cc_) Rn[3 * r + cc_] = R[3 * r] * H[cc_] + R[3 * r + 1] * H[3 + cc_] + R[3 * r + 2] * H[6 + cc_];
                    }
#pragma unroll
                    for (int e = 0; e < 9; ++e) R[e] = Rn[e];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { fl[jn * 6 + r] = pos[r]; fl[jn * 6 + 3 + r] = R[3 * r + 2]; }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) I.Ee[r] = pos[r];
#pragma nounroll
                for (int jn = 0; jn < PJ; ++jn) {
                    const double d0 = pos[0] - fl[jn * 6 + 0], d1 = pos[1] - fl[jn * 6 + 1], d2 = pos[2] - fl[jn * 6 + 2];
                    const double z0 = fl[jn * 6 + 3], z1 = fl[jn * 6 + 4], z2 = fl[jn * 6 + 5];
                    I.J[0][jn] = z1 * d2 - z2 * d1;
                    I.J[1][jn] = z2 * d0 - z0 * d2;
                    I.J[2][jn] = z0 * d1 - z1 * d0;
                }
            }
            __syncthreads();
            if (round == 0) {
                // Minv (column l through a Cholesky solve of the symmetrised M), qdd = Minv (u - bias)
                if (l < PJ) {
                    // (one reciprocal per pivot: float64 division and sqrt are ~25-instruction sequences, the textbook form has 42 + 14 divisions)
                    double Lm[PJ][PJ], rd[PJ];
#pragma unroll
                    for (int i = 0; i < PJ; ++i)
#pragma unroll
                        for (int jj = 0; jj <= i; ++jj) {
                            double sv = 0.5 * (I.M[i][jj] + I.M[jj][i]);
#pragma unroll
                            for (int t = 0; t < jj; ++t) sv -= Lm[i][t] * Lm[jj][t];
                            if (i == jj) { Lm[i][i] = sqrt(sv); rd[i] = 1.0 / Lm[i][i]; }
                            else Lm[i][jj] = sv * rd[jj];
                        }
                    double y[PJ];
#pragma unroll
                    for (int i = 0; i < PJ; ++i) {
                        double sv = (i == l) ? 1.0 : 0.0;
#pragma unroll
                        for (int t = 0; t < i; ++t) sv -= Lm[i][t] * y[t];
                        y[i] = sv * rd[i];
                    }
#pragma unroll
                    for (int i = PJ - 1; i >= 0; --i) {
                        double sv = y[i];
#pragma unroll
                        for (int t = i + 1; t < PJ; ++t) sv -= Lm[t][i] * y[t];
                        y[i] = sv * rd[i];
                    }
#pragma unroll
                    for (int i = 0; i < PJ; ++i) I.Minv[i][l] = y[i];
                }
                __syncthreads();
                if (l < PJ) {
                    double sv = 0;
#pragma unroll
                    for (int jj = 0; jj < PJ; ++jj) sv += I.Minv[l][jj] * (I.U[jj] - I.Bias[jj]);
                    I.Qdd[l] = sv;
                }
                __syncthreads();
            }
        }
        // ---- phase 4a: dqdd = -Minv dID ; cost gradient pieces ----
        for (int pi = l; pi < PJ * PJ; pi += KKT_GL) {
            const int i = pi / PJ, j = pi - i * PJ;
            double sq = 0, sd = 0;
#pragma unroll
            for (int t = 0; t < PJ; ++t) {
                sq += I.Minv[i][t] * sF[gi * KKT_GL + j][t];
                sd += I.Minv[i][t] * sF[gi * KKT_GL + PJ + j][t];
            }
            I.M[i][j] = -sq / KKT_FD_H;                       // dqdd/dq
            I.Dqd[i][j] = -sd / KKT_FD_H;
        }
        if (l < PJ) {
            const float* goal = a.eePos_traj + ((size_t)b * N + k) * 6;
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                s0 += I.J[r][l] * (I.Ee[r] - (double)goal[r]);
                s1 += I.J[r][l] * (I.Ee[r] - (double)goal[6 + r]);       // goal of knot k+1: used by the last block only
            }
            I.Gq[l] = s0;
            I.Gq1[l] = s1;
        }
        __syncthreads();
        // ---- phase 4b: outputs, float, the reference's dense layouts ----
        if (live) {
            float* G = a.G + (size_t)b * ((size_t)(nn + mm) * N - mm) + (size_t)(nn + mm) * k;
            float* Cm = a.C + (size_t)b * (size_t)(nn + nm) * (N - 1) + (size_t)(nn + nm) * k;
            float* g = a.g + (size_t)b * ((size_t)(n + m) * N - m) + (size_t)(n + m) * k;
            float* c = a.c + (size_t)b * (size_t)n * N;
            const double dt = a.dt;
            for (int e = l; e < nn; e += KKT_GL) {
                const int col = e / n, r = e - col * n;                    // column-major
                // A = I + dt [[0, I], [dqdd/dq, dqdd/dqd]]
                double av = (r == col) ? 1.0 : 0.0;
                if (r < PJ) av += (col == r + PJ) ? dt : 0.0;
                else av += dt * (col < PJ ? I.M[r - PJ][col] : I.Dqd[r - PJ][col - PJ]);
                Cm[e] = (float)(-av);
                // Q = blkdiag(g g^T, QD I)
                double qv = 0.0;
                if (r < PJ && col < PJ) qv = I.Gq[r] * I.Gq[col];
                else if (r == col) qv = a.qd_cost;
                G[e] = (float)qv;
                if (k == N - 2) {
                    double q1 = 0.0;
                    if (r < PJ && col < PJ) q1 = I.Gq1[r] * I.Gq1[col];
                    else if (r == col) q1 = a.qd_cost;
                    G[(nn + mm) + e] = (float)q1;
                }
            }
            for (int e = l; e < nm; e += KKT_GL) {
                const int col = e / n, r = e - col * n;                    // B = dt [0; Minv]
                Cm[nn + e] = (float)(-(r < PJ ? 0.0 : dt * I.Minv[r - PJ][col]));
            }
            for (int e = l; e < mm; e += KKT_GL) G[nn + e] = (float)((e % m == e / m) ? a.r_cost : 0.0);
            if (l < n) {
                const double qdl = I.Xq[l < PJ ? l + PJ : l];              // qd_{l mod 7}
                g[l] = (float)(l < PJ ? I.Gq[l] : a.qd_cost * qdl);
                if (k == N - 2) g[(n + m) + l] = (float)(l < PJ ? I.Gq1[l] : a.qd_cost * qdl);     // (evaluated at x_{N-2}: iiwa_eepos_plant.cuh:407)
                // integrator defect c_{k+1} = x_{k+1} - (x_k + dt [qd; qdd])
                const double pred = l < PJ ? I.Xq[l] + dt * qdl : qdl + dt * I.Qdd[l - PJ];
                c[(size_t)n * (k + 1) + l] = (float)((double)xu[(n + m) + l] - pred);
                if (k == 0) c[l] = (float)((double)xu[l] - (double)a.xs[(size_t)b * n + l]);
            }
            if (l < m) g[n + l] = (float)(a.r_cost * I.U[l]);
        }
        __syncthreads();
    }
}

}  // namespace mpcg
