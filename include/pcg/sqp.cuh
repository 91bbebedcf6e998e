// include/pcg/sqp.cuh (shim) — sqpSolvePcg with the reference's name, argument list and return tuple
// (reference include/pcg/sqp.cuh:22, :392), its linear-system section (:207-259: form_schur_system -> pcg launch + the two
// D2H copies, timed -> compute_dz) running on libmpcg_hip.so through the source-level shims of this repo.  The stages of the
// SQP iteration that are outside this library's scope are the plug points of mpcgpu_compat/sqp_stages.cuh.
// Selected by LINSYS_SOLVE == 1 in include/mpcsim.cuh, exactly as in the reference (include/mpcsim.cuh:21-25).
#pragma once
#include <time.h>
#include <cstdint>
#include <tuple>
#include <vector>

#include "../gbd_pcg_compat/gpu_pcg.cuh"
#include "../mpcgpu_compat/linsys_steps.cuh"
#include "../mpcgpu_compat/sqp_stages.cuh"

#ifndef STATE_SIZE
#define STATE_SIZE 14
#endif
#ifndef PCG_NUM_THREADS
#define PCG_NUM_THREADS 128
#endif

namespace mpcgpu_compat {
inline double time_delta_us(const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; }

// Device buffers of one SQP call (reference include/pcg/sqp.cuh:94-135), sized as at :43-47
template <typename T>
struct sqp_buffers {
    T *d_G_dense, *d_C_dense, *d_g, *d_c, *d_S, *d_Pinv, *d_gamma, *d_dz, *d_xs, *d_r, *d_p, *d_v_temp, *d_eta_new_temp;
    uint32_t* d_pcg_iters;
    bool* d_pcg_exit;
    sqp_buffers(uint32_t n, uint32_t m, uint32_t N) {
        const size_t G = (size_t)(n * n + m * m) * N - m * m, C = (size_t)(n * n + n * m) * (N - 1), g = (size_t)(n + m) * N - m;
        gpuErrchk(hipMalloc(&d_G_dense, G * sizeof(T)));   gpuErrchk(hipMalloc(&d_C_dense, C * sizeof(T)));
        gpuErrchk(hipMalloc(&d_g, g * sizeof(T)));         gpuErrchk(hipMalloc(&d_c, (size_t)n * N * sizeof(T)));
        gpuErrchk(hipMalloc(&d_S, 3 * (size_t)n * n * N * sizeof(T)));
        gpuErrchk(hipMalloc(&d_Pinv, 3 * (size_t)n * n * N * sizeof(T)));
        gpuErrchk(hipMalloc(&d_gamma, (size_t)n * N * sizeof(T)));
        gpuErrchk(hipMalloc(&d_dz, g * sizeof(T)));        gpuErrchk(hipMalloc(&d_xs, n * sizeof(T)));
        gpuErrchk(hipMalloc(&d_r, (size_t)n * N * sizeof(T)));   gpuErrchk(hipMalloc(&d_p, (size_t)n * N * sizeof(T)));
        gpuErrchk(hipMalloc(&d_v_temp, (N + 1) * sizeof(T)));    gpuErrchk(hipMalloc(&d_eta_new_temp, (N + 1) * sizeof(T)));
        gpuErrchk(hipMalloc(&d_pcg_iters, sizeof(uint32_t)));    gpuErrchk(hipMalloc(&d_pcg_exit, sizeof(bool)));
    }
    ~sqp_buffers() {
        for (void* p : {(void*)d_G_dense, (void*)d_C_dense, (void*)d_g, (void*)d_c, (void*)d_S, (void*)d_Pinv, (void*)d_gamma, (void*)d_dz,
                        (void*)d_xs, (void*)d_r, (void*)d_p, (void*)d_v_temp, (void*)d_eta_new_temp, (void*)d_pcg_iters, (void*)d_pcg_exit})
            (void)hipFree(p);
    }
};
}  // namespace mpcgpu_compat

template <typename T>
auto sqpSolvePcg(const uint32_t state_size, const uint32_t control_size, const uint32_t knot_points, float timestep, T* d_eePos_traj,
                 T* d_lambda, T* d_xu, void* d_dynMem_const, pcg_config<T>& config, T& rho, T rho_reset) {
    auto& st = mpcgpu_compat::stages<T>();
    mpcgpu_compat::require_stage((bool)st.generate_kkt, "generate_kkt");
    mpcgpu_compat::require_stage((bool)st.globalize_and_step, "globalize_and_step");
    std::vector<int> pcg_iter_vec;
    std::vector<bool> pcg_exit_vec;
    std::vector<double> linsys_time_vec;
    bool sqp_time_exit = 1;                                                        // (data recording, :28)
    timespec sqp_solve_start, sqp_solve_end, linsys_start, linsys_end;
    gpuErrchk(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &sqp_solve_start);

    mpcgpu_compat::sqp_buffers<T> b(state_size, control_size, knot_points);         // (:94-135; the reference also allocates per call)
    gpuErrchk(hipMemcpy(b.d_xs, d_xu, state_size * sizeof(T), hipMemcpyDeviceToDevice));
    T* d_Ginv_dense = b.d_G_dense;                                                  // (:98 alias)

    // the reference's launch set-up, verbatim in structure (:129, :137-151)
    void* pcg_kernel = (void*)pcg<T, STATE_SIZE, KNOT_POINTS>;
    uint32_t pcg_iters;
    bool pcg_exit;
    uint32_t max_iter = config.pcg_max_iter;
    T exit_tol = config.pcg_exit_tol;
    void* pcgKernelArgs[] = {(void*)&b.d_S, (void*)&b.d_Pinv, (void*)&b.d_gamma, (void*)&d_lambda, (void*)&b.d_r, (void*)&b.d_p,
                             (void*)&b.d_v_temp, (void*)&b.d_eta_new_temp, (void*)&b.d_pcg_iters, (void*)&b.d_pcg_exit,
                             (void*)&max_iter, (void*)&exit_tol};
    size_t ppcg_kernel_smem_size = pcgSharedMemSize<T>(state_size, knot_points);

    // the SQP time box (:161-169): true once the call has used more than SQP_MAX_TIME_US of wall time; the loop is left after
    // whichever stage crosses it (:205, :221, :226, :247, :261, :345) and sqp_time_exit keeps its recording value 1 (:28) —
    // only the stage's own "give up" (rho beyond rho_max, :308-313) reports 0
    timespec sqp_cur;
    auto sqpTimecheck = [&]() {
        if (!st.const_update_freq) return false;
        clock_gettime(CLOCK_MONOTONIC, &sqp_cur);
        return mpcgpu_compat::time_delta_us(sqp_solve_start, sqp_cur) > st.sqp_max_time_us;
    };

    uint32_t sqp_iter = 0;
    for (uint32_t sqpiter = 0; sqpiter < st.sqp_max_iter; ++sqpiter) {
        st.generate_kkt(state_size, control_size, knot_points, b.d_G_dense, b.d_C_dense, b.d_g, b.d_c, d_dynMem_const, timestep,
                        d_eePos_traj, b.d_xs, d_xu);                                                         // (:190-204)
        if (sqpTimecheck()) break;                                                                           // (:205)
        form_schur_system<T>(state_size, control_size, knot_points, b.d_G_dense, b.d_C_dense, b.d_g, b.d_c, b.d_S, b.d_Pinv,
                             b.d_gamma, rho);                                                                // (:207-219)
        if (sqpTimecheck()) break;                                                                           // (:221)
        gpuErrchk(hipDeviceSynchronize());
        if (sqpTimecheck()) break;                                                                           // (:226)
        clock_gettime(CLOCK_MONOTONIC, &linsys_start);                                                       // (:224-228)
        gpuErrchk(mpcgLaunchPcg<T>(pcg_kernel, knot_points, PCG_NUM_THREADS, pcgKernelArgs, ppcg_kernel_smem_size));   // (:230)
        gpuErrchk(hipMemcpy(&pcg_iters, b.d_pcg_iters, sizeof(uint32_t), hipMemcpyDeviceToHost));            // (:231)
        gpuErrchk(hipMemcpy(&pcg_exit, b.d_pcg_exit, sizeof(bool), hipMemcpyDeviceToHost));                  // (:232)
        gpuErrchk(hipDeviceSynchronize());
        clock_gettime(CLOCK_MONOTONIC, &linsys_end);
        linsys_time_vec.push_back(mpcgpu_compat::time_delta_us(linsys_start, linsys_end));                   // (:236-241)
        pcg_iter_vec.push_back((int)pcg_iters);
        pcg_exit_vec.push_back(pcg_exit);                                                                    // (:243-244)
        if (sqpTimecheck()) break;                                                                           // (:247)
        compute_dz<T>(state_size, control_size, knot_points, d_Ginv_dense, b.d_C_dense, b.d_g, d_lambda, b.d_dz);   // (:250-259)
        if (sqpTimecheck()) break;                                                                           // (:261)
        const bool go_on = st.globalize_and_step(state_size, control_size, knot_points, d_xu, b.d_dz, rho, rho_reset, sqpiter);
        ++sqp_iter;                                                                                          // (:306, :341: counted once the step is decided)
        if (!go_on) { sqp_time_exit = 0; break; }                                                            // (:308-313)
        if (sqpTimecheck()) break;                                                                           // (:345)
    }
    gpuErrchk(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &sqp_solve_end);
    double sqp_solve_time = mpcgpu_compat::time_delta_us(sqp_solve_start, sqp_solve_end);
    return std::make_tuple(pcg_iter_vec, linsys_time_vec, sqp_solve_time, sqp_iter, sqp_time_exit, pcg_exit_vec);   // (:392)
}
