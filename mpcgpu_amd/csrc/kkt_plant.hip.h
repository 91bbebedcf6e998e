// kkt_plant.hip.h — batched KKT block assembly for a fixed-base serial chain of revolute joints (IIWA-14): the HIP twin of
// generate_kkt_submatrices (reference include/common/kkt.cuh:22-163) together with the plant functions it calls
// (include/dynamics/iiwa/iiwa_eepos_plant.cuh: forwardDynamicsAndGradient :127-155, trackingCostGradientAndHessian :307-390,
// _lastblock :392-411) and the Euler integrator (include/common/integrator.cuh:56-104, 143-162).  SURVEY.md §8f row 4.
//
// The reference runs GRiD-generated, robot-specific code (10 k lines of unrolled recursions) with one thread block per
// knot.  Here the robot is DATA (struct PlantDev: spatial transforms as constant + sin + cos parts, spatial inertias,
// homogeneous transforms) and the algorithms are the generic ones, mapped for a 64-wide wavefront:
//   one wavefront per (trajectory, knot); every lane runs a whole recursive Newton-Euler pass in registers:
//     phase 1  lanes 0..6: columns of the joint-space inertia matrix M = ID(q, 0, e_j);  lane 7: bias c = ID(q, qd, 0)
//     phase 2  lanes 0..6: column j of Minv by a Cholesky solve (7x7, redundantly factorised per lane);  qdd = Minv (u - c)
//     phase 3  lanes 0..27: ID(q +- h e_j, qd, qdd), ID(q, qd +- h e_j, qdd)  ->  central differences of the inverse dynamics
//              lane 32: forward kinematics, end-effector position and geometric Jacobian z_j x (p_ee - p_j)
//     phase 4  dqdd/d(q,qd) = -Minv dID,  A, B, integrator defect, Gauss-Newton cost blocks, written as float in the
//              reference's dense layouts (column-major blocks, C = -A, -B).
// Arithmetic is float64 inside (h = 1e-6 central differences are exact to ~1e-9 there; the MI355X has the fp64 rate
// to spare: the whole kernel is ~60 kflop per knot), results are rounded to float on the way out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mpcg {

constexpr int PJ = 7;                    // joints of the compiled specialisation (state 2 PJ, control PJ)

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

__device__ __forceinline__ void mat3(double (&M)[9], const double* c0, const double* cs, const double* cc, double s, double c) {
#pragma unroll
    for (int e = 0; e < 9; ++e) M[e] = c0[e] + cs[e] * s + cc[e] * c;
}

// tau = ID(q, qd, qdd) without gravity (gato_plant::GRAVITY = 0, iiwa_eepos_plant.cuh:53).  Everything stays in registers.
__device__ void rnea(const PlantDev& P, const double (&q)[PJ], const double (&qd)[PJ], const double (&qdd)[PJ], double (&tau)[PJ]) {
    double f[PJ][6];
    double sn[PJ], cs[PJ];
    double vw[3] = {0, 0, 0}, vu[3] = {0, 0, 0}, aw[3] = {0, 0, 0}, au[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < PJ; ++k) {
        sincos(q[k], &sn[k], &cs[k]);
        double E[9], B[9];
        mat3(E, P.E0[k], P.Es[k], P.Ec[k], sn[k], cs[k]);
        mat3(B, P.B0[k], P.Bs[k], P.Bc[k], sn[k], cs[k]);
        double w[3], u[3], bw[3], bu[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {                    // v = X v_parent, a = X a_parent
            w[r] = E[3 * r] * vw[0] + E[3 * r + 1] * vw[1] + E[3 * r + 2] * vw[2];
            u[r] = B[3 * r] * vw[0] + B[3 * r + 1] * vw[1] + B[3 * r + 2] * vw[2] + E[3 * r] * vu[0] + E[3 * r + 1] * vu[1] + E[3 * r + 2] * vu[2];
            bw[r] = E[3 * r] * aw[0] + E[3 * r + 1] * aw[1] + E[3 * r + 2] * aw[2];
            bu[r] = B[3 * r] * aw[0] + B[3 * r + 1] * aw[1] + B[3 * r + 2] * aw[2] + E[3 * r] * au[0] + E[3 * r + 1] * au[1] + E[3 * r + 2] * au[2];
        }
        w[2] += qd[k];                                   // + S qd, S = e_z (angular)
        bw[2] += qdd[k];
        // + v x (S qd): column 2 of crm(v) times qd
        bw[0] += w[1] * qd[k]; bw[1] -= w[0] * qd[k];
        bu[0] += u[1] * qd[k]; bu[1] -= u[0] * qd[k];
        // f = I a + v x* (I v)
        double Ia[6], Iv[6];
        const double v6[6] = {w[0], w[1], w[2], u[0], u[1], u[2]}, a6[6] = {bw[0], bw[1], bw[2], bu[0], bu[1], bu[2]};
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double sa = 0, sv = 0;
#pragma unroll
            for (int cc_ = 0; cc_ < 6; ++cc_) { sa += P.I[k][6 * r + cc_] * a6[cc_]; sv += P.I[k][6 * r + cc_] * v6[cc_]; }
            Ia[r] = sa; Iv[r] = sv;
        }
        // crf(v) h = [w x n + u x l ; w x l],  h = [n; l]
        f[k][0] = Ia[0] + (w[1] * Iv[2] - w[2] * Iv[1]) + (u[1] * Iv[5] - u[2] * Iv[4]);
        f[k][1] = Ia[1] + (w[2] * Iv[0] - w[0] * Iv[2]) + (u[2] * Iv[3] - u[0] * Iv[5]);
        f[k][2] = Ia[2] + (w[0] * Iv[1] - w[1] * Iv[0]) + (u[0] * Iv[4] - u[1] * Iv[3]);
        f[k][3] = Ia[3] + (w[1] * Iv[5] - w[2] * Iv[4]);
        f[k][4] = Ia[4] + (w[2] * Iv[3] - w[0] * Iv[5]);
        f[k][5] = Ia[5] + (w[0] * Iv[4] - w[1] * Iv[3]);
#pragma unroll
        for (int r = 0; r < 3; ++r) { vw[r] = w[r]; vu[r] = u[r]; aw[r] = bw[r]; au[r] = bu[r]; }
    }
#pragma unroll
    for (int k = PJ - 1; k >= 0; --k) {
        tau[k] = f[k][2];
        if (k > 0) {                                     // f_parent += X^T f = [E^T n + B^T l ; E^T l]
            double E[9], B[9];
            mat3(E, P.E0[k], P.Es[k], P.Ec[k], sn[k], cs[k]);
            mat3(B, P.B0[k], P.Bs[k], P.Bc[k], sn[k], cs[k]);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                f[k - 1][r] += E[r] * f[k][0] + E[3 + r] * f[k][1] + E[6 + r] * f[k][2] + B[r] * f[k][3] + B[3 + r] * f[k][4] + B[6 + r] * f[k][5];
                f[k - 1][3 + r] += E[r] * f[k][3] + E[3 + r] * f[k][4] + E[6 + r] * f[k][5];
            }
        }
    }
}

constexpr int KKT_THREADS = 64;
constexpr double KKT_FD_H = 1e-6;

__global__ __launch_bounds__(KKT_THREADS) void generate_kkt_kernel(KktArgs a) {
    constexpr int n = 2 * PJ, m = PJ, nn = n * n, mm = m * m, nm = n * m;
    __shared__ double sM[PJ][PJ], sMinv[PJ][PJ], sBias[PJ], sQdd[PJ], sId[4 * PJ][PJ], sDq[PJ][PJ], sDqd[PJ][PJ];
    __shared__ double sJ[3][PJ], sEe[3], sGq[PJ], sGq1[PJ];
    const int lane = threadIdx.x;
    const PlantDev& P = *a.plant;
    const int N = a.N;
    const long total = (long)a.batch * (N - 1);
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const int b = (int)(item / (N - 1)), k = (int)(item - (long)b * (N - 1));
        const float* xu = a.xu + (size_t)b * ((size_t)(n + m) * N - m) + (size_t)k * (n + m);      // x_k, u_k, x_{k+1}
        double q[PJ], qd[PJ], u[PJ];
#pragma unroll
        for (int i = 0; i < PJ; ++i) { q[i] = xu[i]; qd[i] = xu[PJ + i]; u[i] = xu[n + i]; }
        // ---- phase 1: inertia matrix columns and bias ----
        if (lane < 8) {
            double z[PJ], e[PJ], t[PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i) { z[i] = 0.0; e[i] = (i == lane) ? 1.0 : 0.0; }
            if (lane < PJ) {
                rnea(P, q, z, e, t);
#pragma unroll
                for (int i = 0; i < PJ; ++i) sM[i][lane] = t[i];
            } else {
                rnea(P, q, qd, z, t);
#pragma unroll
                for (int i = 0; i < PJ; ++i) sBias[i] = t[i];
            }
        }
        __syncthreads();
        // ---- phase 2: Minv (column `lane` through a Cholesky solve of the symmetrised M), qdd ----
        if (lane < PJ) {
            double Lm[PJ][PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double s = 0.5 * (sM[i][j] + sM[j][i]);
#pragma unroll
                    for (int t = 0; t < j; ++t) s -= Lm[i][t] * Lm[j][t];
                    Lm[i][j] = (i == j) ? sqrt(s) : s / Lm[j][j];
                }
            double y[PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i) {
                double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int t = 0; t < i; ++t) s -= Lm[i][t] * y[t];
                y[i] = s / Lm[i][i];
            }
#pragma unroll
            for (int i = PJ - 1; i >= 0; --i) {
                double s = y[i];
#pragma unroll
                for (int t = i + 1; t < PJ; ++t) s -= Lm[t][i] * y[t];
                y[i] = s / Lm[i][i];
            }
#pragma unroll
            for (int i = 0; i < PJ; ++i) sMinv[i][lane] = y[i];
        }
        __syncthreads();
        if (lane < PJ) {
            double s = 0;
#pragma unroll
            for (int j = 0; j < PJ; ++j) s += sMinv[lane][j] * (u[j] - sBias[j]);
            sQdd[lane] = s;
        }
        __syncthreads();
        // ---- phase 3: central differences of ID at (q, qd, qdd); kinematics on a lane of its own ----
        if (lane < 4 * PJ) {
            double qdd[PJ], qq[PJ], qqd[PJ], t[PJ];
#pragma unroll
            for (int i = 0; i < PJ; ++i) { qdd[i] = sQdd[i]; qq[i] = q[i]; qqd[i] = qd[i]; }
            const int j = lane % PJ, kind = lane / PJ;                  // 0: q + h, 1: q - h, 2: qd + h, 3: qd - h
            const double hh = (kind & 1) ? -KKT_FD_H : KKT_FD_H;
#pragma unroll
            for (int i = 0; i < PJ; ++i) {
                if (i == j && kind < 2) qq[i] += hh;
                if (i == j && kind >= 2) qqd[i] += hh;
            }
            rnea(P, qq, qqd, qdd, t);
#pragma unroll
            for (int i = 0; i < PJ; ++i) sId[lane][i] = t[i];
        } else if (lane == 32) {
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
            double pj[PJ][3], zj[PJ][3];
#pragma unroll
            for (int jn = 0; jn < PJ; ++jn) {
                double s, c, H[9];
                sincos(q[jn], &s, &c);
                mat3(H, P.R0[jn], P.Rs[jn], P.Rc[jn], s, c);
                double Rn[9];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    pos[r] += R[3 * r] * P.p[jn][0] + R[3 * r + 1] * P.p[jn][1] + R[3 * r + 2] * P.p[jn][2];
#pragma unroll
                    for (int cc_ = 0; cc_ < 3; ++cc_) Rn[3 * r + cc_] = R[3 * r] * H[cc_] + R[3 * r + 1] * H[3 + cc_] + R[3 * r + 2] * H[6 + cc_];
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) R[e] = Rn[e];
#pragma unroll
                for (int r = 0; r < 3; ++r) { pj[jn][r] = pos[r]; zj[jn][r] = R[3 * r + 2]; }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) sEe[r] = pos[r];
#pragma unroll
            for (int jn = 0; jn < PJ; ++jn) {
                const double d0 = pos[0] - pj[jn][0], d1 = pos[1] - pj[jn][1], d2 = pos[2] - pj[jn][2];
                sJ[0][jn] = zj[jn][1] * d2 - zj[jn][2] * d1;
                sJ[1][jn] = zj[jn][2] * d0 - zj[jn][0] * d2;
                sJ[2][jn] = zj[jn][0] * d1 - zj[jn][1] * d0;
            }
        }
        __syncthreads();
        // ---- phase 4a: dqdd = -Minv dID ; cost gradient pieces ----
        if (lane < PJ * PJ) {
            const int i = lane / PJ, j = lane % PJ;
            double sq = 0, sd = 0;
#pragma unroll
            for (int t = 0; t < PJ; ++t) {
                sq += sMinv[i][t] * (sId[j][t] - sId[PJ + j][t]);
                sd += sMinv[i][t] * (sId[2 * PJ + j][t] - sId[3 * PJ + j][t]);
            }
            sDq[i][j] = -sq / (2 * KKT_FD_H);
            sDqd[i][j] = -sd / (2 * KKT_FD_H);
        } else if (lane >= 56 && lane < 56 + PJ) {
            const int j = lane - 56;
            const float* goal = a.eePos_traj + ((size_t)b * N + k) * 6;
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                s0 += sJ[r][j] * (sEe[r] - (double)goal[r]);
                s1 += sJ[r][j] * (sEe[r] - (double)goal[6 + r]);       // goal of knot k+1: used by the last block only
            }
            sGq[j] = s0;
            sGq1[j] = s1;
        }
        __syncthreads();
        // ---- phase 4b: outputs, float, the reference's dense layouts ----
        float* G = a.G + (size_t)b * ((size_t)(nn + mm) * N - mm) + (size_t)(nn + mm) * k;
        float* Cm = a.C + (size_t)b * (size_t)(nn + nm) * (N - 1) + (size_t)(nn + nm) * k;
        float* g = a.g + (size_t)b * ((size_t)(n + m) * N - m) + (size_t)(n + m) * k;
        float* c = a.c + (size_t)b * (size_t)n * N;
        const double dt = a.dt;
        for (int e = lane; e < nn; e += KKT_THREADS) {
            const int r = e % n, col = e / n;                          // column-major
            // A = I + dt [[0, I], [dqdd/dq, dqdd/dqd]]
            double av = (r == col) ? 1.0 : 0.0;
            if (r < PJ) av += (col == r + PJ) ? dt : 0.0;
            else av += dt * (col < PJ ? sDq[r - PJ][col] : sDqd[r - PJ][col - PJ]);
            Cm[e] = (float)(-av);
            // Q = blkdiag(g g^T, QD I)
            double qv = 0.0;
            if (r < PJ && col < PJ) qv = sGq[r] * sGq[col];
            else if (r == col) qv = a.qd_cost;
            G[e] = (float)qv;
            if (k == N - 2) {
                double q1 = 0.0;
                if (r < PJ && col < PJ) q1 = sGq1[r] * sGq1[col];
                else if (r == col) q1 = a.qd_cost;
                G[(nn + mm) + e] = (float)q1;
            }
        }
        for (int e = lane; e < nm; e += KKT_THREADS) {
            const int r = e % n, col = e / n;                          // B = dt [0; Minv]
            Cm[nn + e] = (float)(-(r < PJ ? 0.0 : dt * sMinv[r - PJ][col]));
        }
        for (int e = lane; e < mm; e += KKT_THREADS) G[nn + e] = (float)((e % m == e / m) ? a.r_cost : 0.0);
        if (lane < n) {
            g[lane] = (float)(lane < PJ ? sGq[lane] : a.qd_cost * qd[lane - PJ]);
            if (k == N - 2) g[(n + m) + lane] = (float)(lane < PJ ? sGq1[lane] : a.qd_cost * qd[lane - PJ]);     // (evaluated at x_{N-2}: iiwa_eepos_plant.cuh:407)
            // integrator defect c_{k+1} = x_{k+1} - (x_k + dt [qd; qdd])
            const double pred = lane < PJ ? q[lane] + dt * qd[lane] : qd[lane - PJ] + dt * sQdd[lane - PJ];
            c[(size_t)n * (k + 1) + lane] = (float)((double)xu[(n + m) + lane] - pred);
            if (k == 0) c[lane] = (float)((double)xu[lane] - (double)a.xs[(size_t)b * n + lane]);
        } else if (lane < n + m) {
            g[lane] = (float)(a.r_cost * u[lane - n]);
        }
        __syncthreads();
    }
}

}  // namespace mpcg
