// include/mpcsim.cuh (shim) — simulateMPC with the reference's name, argument list and return tuple (reference
// include/mpcsim.cuh:146-150, :421-425) and the reference's compile-time solver switch (:21-25): LINSYS_SOLVE == 1 ->
// sqpSolvePcg (this repo's include/pcg/sqp.cuh), otherwise sqpSolveQdldl (include/qdldl/sqp.cuh).  The MPC loop body
// that is outside this library's scope — plant simulation, tracking error, horizon shift (:288-341) — is the
// simulate_and_shift plug point of mpcgpu_compat/sqp_stages.cuh.  lambda is owned here and warm-starts every SQP call
// (:186, :190, :267), as in the reference.
#pragma once
#include <cstdint>
#include <string>
#include <tuple>
#include <vector>

#ifndef LINSYS_SOLVE
#define LINSYS_SOLVE 1            // include/common/settings.cuh:119-121
#endif
#ifndef TIME_LINSYS
#define TIME_LINSYS 1             // include/common/settings.cuh:107-109
#endif
#ifndef PCG_MAX_ITER
#define PCG_MAX_ITER 200
#endif

#if LINSYS_SOLVE == 1
#include "pcg/sqp.cuh"
#else
#include "qdldl/sqp.cuh"
#endif

typedef float linsys_t;           // include/common/settings.cuh:45-49 (USE_DOUBLES == 0)
#if TIME_LINSYS == 1
typedef double toplevel_return_type;
#else
typedef uint32_t toplevel_return_type;
#endif

template <typename T, typename return_type>
std::tuple<std::vector<toplevel_return_type>, std::vector<linsys_t>, linsys_t> simulateMPC(
    const uint32_t state_size, const uint32_t control_size, const uint32_t knot_points, const uint32_t traj_steps, float timestep,
    T* d_eePos_traj, T* d_xu_traj, T* d_xs, uint32_t start_state_ind, uint32_t goal_state_ind, uint32_t test_iter, T linsys_exit_tol,
    std::string test_output_prefix) {
    (void)traj_steps; (void)start_state_ind; (void)goal_state_ind; (void)test_iter; (void)test_output_prefix;
    auto& st = mpcgpu_compat::stages<T>();
    mpcgpu_compat::require_stage((bool)st.simulate_and_shift, "simulate_and_shift");
    const uint32_t traj_len = (state_size + control_size) * knot_points - control_size;
    const int max_control_updates = 100000;

    std::vector<double> linsys_times;
    std::vector<uint32_t> sqp_iters;
    std::vector<T> tracking_errors;
    T cur_tracking_error = 0;

    // mpc iterates (:185-193)
    T *d_lambda, *d_eePos_goal, *d_xu;
    gpuErrchk(hipMalloc(&d_lambda, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_xu, traj_len * sizeof(T)));
    gpuErrchk(hipMalloc(&d_eePos_goal, 6 * knot_points * sizeof(T)));
    gpuErrchk(hipMemset(d_lambda, 0, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMemcpy(d_eePos_goal, d_eePos_traj, 6 * knot_points * sizeof(T), hipMemcpyDeviceToDevice));
    gpuErrchk(hipMemcpy(d_xu, d_xu_traj, traj_len * sizeof(T), hipMemcpyDeviceToDevice));
    void* d_dynmem = st.dynmem;    // gato_plant::initializeDynamicsConstMem<T>() in the reference (:194): the registered stages' model (an mpcg_plant* with the library's own generate_kkt stage)

    T rho = 1e-3, rho_reset = 1e-3;                                                  // (:219)
#if LINSYS_SOLVE == 1
    pcg_config<T> config;                                                            // (:213-216)
    config.pcg_block = PCG_NUM_THREADS;
    config.pcg_exit_tol = linsys_exit_tol;
    config.pcg_max_iter = PCG_MAX_ITER;
#else
    (void)linsys_exit_tol;
#endif
    for (int control_update_step = 0; control_update_step < max_control_updates; ++control_update_step) {       // (:249)
#if LINSYS_SOLVE == 1
        auto sqp_stats = sqpSolvePcg<T>(state_size, control_size, knot_points, timestep, d_eePos_goal, d_lambda, d_xu, d_dynmem, config, rho, rho_reset);   // (:267)
#else
        auto sqp_stats = sqpSolveQdldl<T>(state_size, control_size, knot_points, timestep, d_eePos_goal, d_lambda, d_xu, d_dynmem, rho, rho_reset);         // (:269)
#endif
        const std::vector<double>& cur_linsys_times = std::get<1>(sqp_stats);                                  // (:272-277)
        linsys_times.insert(linsys_times.end(), cur_linsys_times.begin(), cur_linsys_times.end());
        sqp_iters.push_back(std::get<3>(sqp_stats));
        bool done = false;
        cur_tracking_error = st.simulate_and_shift(state_size, control_size, knot_points, d_xs, d_xu, d_lambda, d_eePos_goal,
                                                   std::get<2>(sqp_stats), done);                              // (:288-341)
        tracking_errors.push_back(cur_tracking_error);
        if (done) break;
    }
    gpuErrchk(hipFree(d_lambda));
    gpuErrchk(hipFree(d_xu));
    gpuErrchk(hipFree(d_eePos_goal));
#if TIME_LINSYS == 1
    return std::make_tuple(linsys_times, tracking_errors, cur_tracking_error);                                  // (:421-425)
#else
    return std::make_tuple(sqp_iters, tracking_errors, cur_tracking_error);
#endif
}
