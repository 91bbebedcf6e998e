// mpcsim_shim_demo.cpp — the reference's entry-point chain  simulateMPC -> sqpSolvePcg | sqpSolveQdldl  (include/mpcsim.cuh:147,
// :267-269; selected at compile time by LINSYS_SOLVE, :21-25) over THIS repo's shim headers include/mpcsim.cuh,
// include/pcg/sqp.cuh, include/qdldl/sqp.cuh, with the stages outside the library's scope registered through
// mpcgpu_compat::stages<T>() for a synthetic linear-quadratic tracking problem:
//     min sum_k 1/2 (x_k - xbar)^T Q (x_k - xbar) + 1/2 u_k^T R u_k   s.t.  x_{k+1} = A x_k + B u_k,  x_0 = x_s
// For an LQ problem one full SQP step (alpha = -1, the reference's sign convention include/pcg/sqp.cuh:317) lands on the
// optimum, so after each control step the program checks the KKT conditions of the CURRENT iterate on the CPU: dynamics
// defect = 0 and the stationarity residual with the multipliers lambda the linear solver returned.
//   hipcc --offload-arch=gfx950 -O2 -DLINSYS_SOLVE=1 -Iinclude examples/mpcsim_shim_demo.cpp -Lmpcgpu_amd -lmpcg_hip   (and -DLINSYS_SOLVE=0)
#include <cmath>
#include <cstdio>
#include <vector>

#define STATE_SIZE 14
#define KNOT_POINTS 32
#define PCG_MAX_ITER 2000
#include "mpcsim.cuh"

typedef float T;
static const int n = 14, m = 7, N = KNOT_POINTS, nn = n * n, mm = m * m, nm = n * m;

struct Problem {
    std::vector<T> Q, R, A, B, xbar, xs;      // column-major blocks
    Problem() : Q(nn, 0.f), R(mm, 0.f), A(nn, 0.f), B(nm, 0.f), xbar(n), xs(n) {
        unsigned s = 4242u;
        auto rnd = [&s]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) { T v = (i == j) ? 1.0f + 0.2f * rnd() : 0.05f * rnd(); Q[i + j * n] = v; Q[j + i * n] = v; }
        for (int i = 0; i < m; ++i) R[i + i * m] = 0.3f + 0.1f * rnd();
        for (int e = 0; e < nn; ++e) A[e] = ((e % n == e / n) ? 0.95f : 0.0f) + 0.05f * rnd();
        for (int e = 0; e < nm; ++e) B[e] = 0.2f * rnd();
        for (int i = 0; i < n; ++i) { xbar[i] = rnd(); xs[i] = rnd(); }
    }
};
static Problem P;
static int g_steps = 0;
static double g_worst_defect = 0, g_worst_station = 0;

int main() {
    const uint32_t state_size = n, control_size = m, knot_points = N;
    const size_t Gsz = (size_t)(nn + mm) * N - mm, Csz = (size_t)(nn + nm) * (N - 1), gsz = (size_t)(n + m) * N - m;
    auto& st = mpcgpu_compat::stages<T>();
    st.sqp_max_iter = 1;
    st.const_update_freq = false;     // CONST_UPDATE_FREQ = 0 for the LQ run (host-assembled KKT blocks: a control step takes far longer than SQP_MAX_TIME_US)
    // generate_kkt_submatrices stand-in (host-assembled, H2D): G = [Q,R,...,Q], C = [-A,-B,...], g = cost gradient, c = constraint residual
    st.generate_kkt = [&](uint32_t, uint32_t, uint32_t, T* d_G, T* d_C, T* d_g, T* d_c, void*, float, T*, T* d_xs, T* d_xu) {
        std::vector<T> xu(gsz), xs(n), G(Gsz), C(Csz), g(gsz), c((size_t)n * N);
        gpuErrchk(hipMemcpy(xu.data(), d_xu, gsz * sizeof(T), hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(xs.data(), d_xs, n * sizeof(T), hipMemcpyDeviceToHost));
        for (int k = 0; k < N; ++k) {
            const T* x = &xu[(size_t)k * (n + m)];
            for (int e = 0; e < nn; ++e) G[(size_t)k * (nn + mm) + e] = P.Q[e];
            for (int i = 0; i < n; ++i) {
                double a = 0;
                for (int j = 0; j < n; ++j) a += (double)P.Q[i + j * n] * (x[j] - P.xbar[j]);
                g[(size_t)k * (n + m) + i] = (T)a;
            }
            if (k < N - 1) {
                for (int e = 0; e < mm; ++e) G[(size_t)k * (nn + mm) + nn + e] = P.R[e];
                for (int i = 0; i < m; ++i) g[(size_t)k * (n + m) + n + i] = P.R[i + i * m] * x[n + i];
                for (int e = 0; e < nn; ++e) C[(size_t)k * (nn + nm) + e] = -P.A[e];          // stored negated (include/common/kkt.cuh:115-116)
                for (int e = 0; e < nm; ++e) C[(size_t)k * (nn + nm) + nn + e] = -P.B[e];
                const T* xn = &xu[(size_t)(k + 1) * (n + m)];
                for (int i = 0; i < n; ++i) {                                                  // c_{k+1} = x_{k+1} - (A x_k + B u_k)
                    double a = xn[i];
                    for (int j = 0; j < n; ++j) a -= (double)P.A[i + j * n] * x[j];
                    for (int j = 0; j < m; ++j) a -= (double)P.B[i + j * n] * x[n + j];
                    c[(size_t)(k + 1) * n + i] = (T)a;
                }
            }
        }
        for (int i = 0; i < n; ++i) c[i] = xu[i] - xs[i];                                       // c_0 = x_0 - x_s (include/common/kkt.cuh:106-108)
        gpuErrchk(hipMemcpy(d_G, G.data(), Gsz * sizeof(T), hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_C, C.data(), Csz * sizeof(T), hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_g, g.data(), gsz * sizeof(T), hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_c, c.data(), (size_t)n * N * sizeof(T), hipMemcpyHostToDevice));
    };
    // full Newton step xu += alpha dz with alpha = -1 (include/pcg/sqp.cuh:317, 332), no line search; one SQP iteration per control step
    st.globalize_and_step = [&](uint32_t, uint32_t, uint32_t, T* d_xu, T* d_dz, T&, T, uint32_t) {
        std::vector<T> xu(gsz), dz(gsz);
        gpuErrchk(hipMemcpy(xu.data(), d_xu, gsz * sizeof(T), hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(dz.data(), d_dz, gsz * sizeof(T), hipMemcpyDeviceToHost));
        for (size_t e = 0; e < gsz; ++e) xu[e] -= dz[e];
        gpuErrchk(hipMemcpy(d_xu, xu.data(), gsz * sizeof(T), hipMemcpyHostToDevice));
        return false;
    };
    // after the step: KKT check of the iterate on the CPU, then "simulate" = move the start state one knot along the plan
    st.simulate_and_shift = [&](uint32_t, uint32_t, uint32_t, T* d_xs, T* d_xu, T* d_lambda, T*, double, bool& done) {
        std::vector<T> xu(gsz), lam((size_t)n * N), xs(n);
        gpuErrchk(hipMemcpy(xu.data(), d_xu, gsz * sizeof(T), hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(lam.data(), d_lambda, (size_t)n * N * sizeof(T), hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(xs.data(), d_xs, n * sizeof(T), hipMemcpyDeviceToHost));
        double defect = 0, station = 0, scale = 1e-30;
        for (int k = 0; k < N; ++k) {
            const T* x = &xu[(size_t)k * (n + m)];
            for (int i = 0; i < n; ++i) {
                double d = (k == 0) ? x[i] - xs[i] : x[i];
                if (k > 0) {
                    const T* xp = &xu[(size_t)(k - 1) * (n + m)];
                    for (int j = 0; j < n; ++j) d -= (double)P.A[i + j * n] * xp[j];
                    for (int j = 0; j < m; ++j) d -= (double)P.B[i + j * n] * xp[n + j];
                }
                defect = fmax(defect, fabs(d));
                // stationarity of the Lagrangian in x_k with the solver's multipliers: the QP of the LAST step had gradient
                // g(z_old) and G; at z_new = z_old - dz the exact gradient is g(z_old) - G dz, and  G dz + C^T lam = g(z_old):
                //   Q (x_k - xbar) = lam_k - A^T lam_{k+1}        (+ rho-regularisation error of order rho |dz|)
                double a = -lam[(size_t)k * n + i];
                for (int j = 0; j < n; ++j) a += (double)P.Q[i + j * n] * (x[j] - P.xbar[j]);
                if (k < N - 1) for (int t = 0; t < n; ++t) a += (double)P.A[t + i * n] * lam[(size_t)(k + 1) * n + t];
                station = fmax(station, fabs(a));
                scale = fmax(scale, fabs(lam[(size_t)k * n + i]));
            }
        }
        g_worst_defect = fmax(g_worst_defect, defect);
        g_worst_station = fmax(g_worst_station, station / scale);
        // "simulate": the plant follows the plan for one knot; shift the horizon by one knot (just_shift, include/mpcsim.cuh:297-341:
        // xu and lambda move up, the last knot is repeated) so that x_0 of the iterate is the new start state
        for (int i = 0; i < n; ++i) xs[i] = xu[(size_t)(n + m) + i];
        for (int k = 0; k + 1 < N; ++k) {
            for (int i = 0; i < n + (k + 2 < N ? m : 0); ++i) xu[(size_t)k * (n + m) + i] = xu[(size_t)(k + 1) * (n + m) + i];
            for (int i = 0; i < n; ++i) lam[(size_t)k * n + i] = lam[(size_t)(k + 1) * n + i];
        }
        gpuErrchk(hipMemcpy(d_xu, xu.data(), gsz * sizeof(T), hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_lambda, lam.data(), (size_t)n * N * sizeof(T), hipMemcpyHostToDevice));
        gpuErrchk(hipMemcpy(d_xs, xs.data(), n * sizeof(T), hipMemcpyHostToDevice));
        done = ++g_steps >= 4;
        double err = 0;
        for (int i = 0; i < n; ++i) err += fabs(xs[i] - P.xbar[i]);
        return (T)err;
    };

    std::vector<T> xu0(gsz, 0.f);
    T *d_xu_traj, *d_eePos_traj, *d_xs;
    gpuErrchk(hipMalloc(&d_xu_traj, gsz * sizeof(T)));
    gpuErrchk(hipMalloc(&d_eePos_traj, 6 * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_xs, n * sizeof(T)));
    for (int i = 0; i < n; ++i) xu0[i] = P.xs[i];
    gpuErrchk(hipMemcpy(d_xu_traj, xu0.data(), gsz * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemset(d_eePos_traj, 0, 6 * N * sizeof(T)));
    gpuErrchk(hipMemcpy(d_xs, P.xs.data(), n * sizeof(T), hipMemcpyHostToDevice));

    auto res = simulateMPC<T, toplevel_return_type>(state_size, control_size, knot_points, /*traj_steps*/ 4, 1.0f / 64, d_eePos_traj, d_xu_traj, d_xs,
                                                    0, 0, 0, (T)1e-10, std::string("demo"));
    const std::vector<toplevel_return_type>& linsys_times = std::get<0>(res);
    const std::vector<linsys_t>& tracking = std::get<1>(res);
    // ---- the SQP time box (reference include/pcg/sqp.cuh:161-169, CONST_UPDATE_FREQ = 1): the same problem through the solver plugin
    // directly, the step stage asking for more iterations every time.  (a) box off: all sqp_max_iter iterations run; (b) a box
    // already used up when the first stage returns: the loop is left at its first check, no linear system is solved, and
    // sqp_time_exit keeps its recording value 1 (:28 — only the rho > rho_max exit reports 0); (c) a generous box: as (a).
    st.globalize_and_step = [&](uint32_t, uint32_t, uint32_t, T*, T*, T&, T, uint32_t) { return true; };
    T* d_lambda2;
    gpuErrchk(hipMalloc(&d_lambda2, (size_t)n * N * sizeof(T)));
    gpuErrchk(hipMemset(d_lambda2, 0, (size_t)n * N * sizeof(T)));
    T rho = 1e-3f;
    unsigned box_iters[3];
    bool box_flag[3];
    size_t box_solves[3];
    const double boxes[3] = {0.0, 1.0, 60e6};
    for (int c = 0; c < 3; ++c) {
        st.sqp_max_iter = 5;
        st.const_update_freq = c != 0;
        st.sqp_max_time_us = boxes[c];
#if LINSYS_SOLVE == 1
        pcg_config<T> cfg;
        cfg.pcg_exit_tol = (T)1e-10; cfg.pcg_max_iter = PCG_MAX_ITER;
        auto s2 = sqpSolvePcg<T>(state_size, control_size, knot_points, 1.0f / 64, d_eePos_traj, d_lambda2, d_xu_traj, nullptr, cfg, rho, (T)1e-3);
#else
        auto s2 = sqpSolveQdldl<T>(state_size, control_size, knot_points, 1.0f / 64, d_eePos_traj, d_lambda2, d_xu_traj, nullptr, rho, (T)1e-3);
#endif
        box_iters[c] = std::get<3>(s2); box_flag[c] = std::get<4>(s2); box_solves[c] = std::get<1>(s2).size();
    }
    const bool box_ok = box_iters[0] == 5 && box_solves[0] == 5 && box_flag[0] && box_iters[1] == 0 && box_solves[1] == 0 && box_flag[1] &&
                        box_iters[2] == 5 && box_solves[2] == 5;
    printf("{\"linsys_solve\": %d, \"control_steps\": %d, \"linsolves\": %zu, \"mean_linsys_us\": %.1f, \"dynamics_defect\": %.3e, "
           "\"stationarity_rel\": %.3e, \"tracking_first\": %.4f, \"tracking_last\": %.4f, \"time_box\": {\"off_iters\": %u, \"expired_iters\": %u, "
           "\"expired_linsolves\": %zu, \"expired_sqp_time_exit\": %d, \"generous_iters\": %u, \"ok\": %s}}\n",
           LINSYS_SOLVE, g_steps, linsys_times.size(), linsys_times.empty() ? 0.0 : (double)linsys_times.back(), g_worst_defect, g_worst_station,
           (double)tracking.front(), (double)tracking.back(), box_iters[0], box_iters[1], box_solves[1], (int)box_flag[1], box_iters[2], box_ok ? "true" : "false");
    return (g_steps == 4 && g_worst_defect < 1e-3 && g_worst_station < 2e-2 && box_ok) ? 0 : 1;
}
