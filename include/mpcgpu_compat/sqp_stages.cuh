// sqp_stages.cuh — plug points for the stages of MPCGPU's SQP / MPC loop that are OUTSIDE this library's scope
// (KKT assembly with the robot dynamics, merit / line search, plant simulation + horizon
// shift).  The shim versions of sqpSolvePcg / sqpSolveQdldl / simulateMPC (include/pcg/sqp.cuh, include/qdldl/sqp.cuh,
// include/mpcsim.cuh of THIS repo) keep the reference's names, argument lists and return tuples and run the
// linear-system section on libmpcg_hip; wherever the reference calls one of its own out-of-scope kernels they call the
// function registered here.  A maintainer porting MPCGPU registers thin wrappers around the reference's kernels
// (generate_kkt_submatrices, ls_gato_compute_merit + the alpha / rho logic of include/pcg/sqp.cuh:265-353,
// simple_simulate + just_shift); examples/mpcsim_shim_demo.cpp registers a synthetic convex problem.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>

#ifndef CONST_UPDATE_FREQ
#define CONST_UPDATE_FREQ 1      // include/common/settings.cuh:56-58
#endif
#ifndef SQP_MAX_TIME_US
#define SQP_MAX_TIME_US 2000     // include/common/settings.cuh:161-163
#endif

namespace mpcgpu_compat {

template <typename T>
struct sqp_stages {
    // generate_kkt_submatrices<<<knot_points, KKT_THREADS>>> (include/pcg/sqp.cuh:190-204, include/common/kkt.cuh:22-163):
    // fill d_G_dense, d_C_dense, d_g, d_c for the current iterate d_xu.
    std::function<void(uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_G_dense, T* d_C_dense, T* d_g, T* d_c,
                       void* d_dynMem_const, float timestep, T* d_eePos_traj, T* d_xs, T* d_xu)> generate_kkt;
    // everything after compute_dz in one SQP iteration (include/pcg/sqp.cuh:265-353): line search over alpha = -1/2^p,
    // rho adaptation, xu += alpha dz.  Returns false to stop the SQP loop.
    std::function<bool(uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_xu, T* d_dz, T& rho, T rho_reset,
                       uint32_t sqp_iter)> globalize_and_step;
    // simple_simulate + horizon shift of one control step (include/mpcsim.cuh:288-341).  Returns the tracking error; sets
    // done when the reference trajectory is exhausted.
    std::function<T(uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_xs, T* d_xu, T* d_lambda,
                    T* d_eePos_goal, double sqp_solve_time_us, bool& done)> simulate_and_shift;
    // d_dynMem_const of the reference (gato_plant::initializeDynamicsConstMem<T>(), include/mpcsim.cuh:194): handed to every SQP call by
    // simulateMPC and on to the generate_kkt stage.  With the library's own stage (use_mpcg_generate_kkt below) it is an mpcg_plant*.
    void* dynmem = nullptr;
    uint32_t sqp_max_iter = 20;          // SQP_MAX_ITER with TIME_LINSYS (include/common/settings.cuh:152-158)
    // The SQP time box (include/common/settings.cuh:56-58 CONST_UPDATE_FREQ = 1, :161-163 SQP_MAX_TIME_US = 2000): with
    // const_update_freq the loop is left as soon as sqpTimecheck() — wall time since the start of the call, allocation included,
    // as the reference measures it (include/pcg/sqp.cuh:35, :161-169) — exceeds sqp_max_time_us; checked after every stage.
    bool const_update_freq = CONST_UPDATE_FREQ != 0;
    double sqp_max_time_us = SQP_MAX_TIME_US;
};

template <typename T>
inline sqp_stages<T>& stages() {
    static sqp_stages<T> s;
    return s;
}

// The library's own generate_kkt_submatrices (mpcg_generate_kkt: IIWA-14 dynamics, tracking cost, Euler integrator on the device) as the
// generate_kkt stage — the default when an mpcg_plant is supplied; merit function / line search and the plant simulation stay plug points.
// qd_cost / r_cost: QD_COST / R_COST of include/common/settings.cuh:84-94.  Needs gbd_pcg_compat/gpu_pcg.cuh (handle cache) before this header.
#ifdef MPCG_H
template <typename T>
inline void use_mpcg_generate_kkt(mpcg_plant* plant, float qd_cost, float r_cost) {
    auto& st = stages<T>();
    st.dynmem = plant;
    st.generate_kkt = [qd_cost, r_cost](uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_G_dense, T* d_C_dense, T* d_g, T* d_c,
                                         void* d_dynMem_const, float timestep, T* d_eePos_traj, T* d_xs, T* d_xu) {
        mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
        if (mpcg_generate_kkt(h, static_cast<const mpcg_plant*>(d_dynMem_const), control_size, timestep, d_eePos_traj, d_xs, d_xu, qd_cost, r_cost,
                              d_G_dense, d_C_dense, d_g, d_c, 1, /*stream*/ nullptr) != MPCG_OK)
            mpcg_compat::die("generate_kkt_submatrices", h);
    };
}
#endif

inline void require_stage(bool present, const char* which) {
    if (!present) {
        fprintf(stderr, "mpcgpu_compat: stage '%s' is not registered (mpcgpu_compat::stages<T>()); it is outside libmpcg_hip's scope\n", which);
        exit(EXIT_FAILURE);             // the reference's error convention: abort
    }
}

}  // namespace mpcgpu_compat
