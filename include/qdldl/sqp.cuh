// include/qdldl/sqp.cuh (shim) — sqpSolveQdldl with the reference's name, argument list and return tuple
// (reference include/qdldl/sqp.cuh:53), its linear-system section (:257-282: form_schur_system_qdldl -> D2H(values, gamma) ->
// qdldl_solve_schur -> H2D(lambda), timed -> compute_dz) on libmpcg_hip.so: the Schur system is formed on the GPU
// (mpcg_form_schur without a preconditioner + mpcg_bd_to_csr_lowertri = what form_schur_system_qdldl leaves in d_val),
// the LDL^T runs on the HOST in the library's own QDLDL-equivalent (mpcg_ldl_*, mpcgpu_amd/csrc/ldl_host.hpp).
// Selected by LINSYS_SOLVE == 0 in include/mpcsim.cuh (reference include/mpcsim.cuh:21-25).
#pragma once
#include <time.h>
#include <cstdint>
#include <tuple>
#include <vector>

#include "../pcg/sqp.cuh"     // sqp_buffers, stages, time_delta_us (no PCG call is made from this file)

template <typename T>
auto sqpSolveQdldl(uint32_t state_size, uint32_t control_size, uint32_t knot_points, float timestep, T* d_eePos_traj, T* d_lambda,
                   T* d_xu, void* d_dynMem_const, T& rho, T rho_reset) {
    static_assert(std::is_same<T, float>::value, "QDLDL_float == linsys_t == float (reference examples/track_iiwa_qdldl.cu:22, Makefile:16)");
    auto& st = mpcgpu_compat::stages<T>();
    mpcgpu_compat::require_stage((bool)st.generate_kkt, "generate_kkt");
    mpcgpu_compat::require_stage((bool)st.globalize_and_step, "globalize_and_step");
    std::vector<int> linsys_iter_vec;
    std::vector<bool> linsys_exit_vec;
    std::vector<double> linsys_time_vec;
    bool sqp_time_exit = 1;
    timespec sqp_solve_start, sqp_solve_end, linsys_start, linsys_end;
    gpuErrchk(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &sqp_solve_start);

    mpcgpu_compat::sqp_buffers<T> b(state_size, control_size, knot_points);
    gpuErrchk(hipMemcpy(b.d_xs, d_xu, state_size * sizeof(T), hipMemcpyDeviceToDevice));
    mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
    // pattern + elimination tree once per SQP call (:148-198: prep_csr :164, QDLDL_etree :193)
    mpcg_ldl* ldl = nullptr;
    if (mpcg_ldl_create(&ldl, state_size, knot_points) != MPCG_OK) mpcg_compat::die("mpcg_ldl_create", h);
    uint32_t nnz = 0;
    mpcg_ldl_pattern(ldl, nullptr, nullptr, &nnz, nullptr);
    T* d_val;
    gpuErrchk(hipMalloc(&d_val, nnz * sizeof(T)));

    // the SQP time box (reference include/qdldl/sqp.cuh:206-214; checks at :254, :259, :263, :284, :298, :382)
    timespec sqp_cur;
    auto sqpTimecheck = [&]() {
        if (!st.const_update_freq) return false;
        clock_gettime(CLOCK_MONOTONIC, &sqp_cur);
        return mpcgpu_compat::time_delta_us(sqp_solve_start, sqp_cur) > st.sqp_max_time_us;
    };

    uint32_t sqp_iter = 0;
    for (uint32_t sqpiter = 0; sqpiter < st.sqp_max_iter; ++sqpiter) {
        st.generate_kkt(state_size, control_size, knot_points, b.d_G_dense, b.d_C_dense, b.d_g, b.d_c, d_dynMem_const, timestep,
                        d_eePos_traj, b.d_xs, d_xu);
        if (sqpTimecheck()) break;                                                                            // (:254)
        // form_schur_system_qdldl (:257): S in the bd layout, then its lower triangle in CSR order
        if (mpcg_form_schur(h, control_size, b.d_G_dense, b.d_C_dense, b.d_g, b.d_c, b.d_S, nullptr, b.d_gamma, rho, 1, MPCG_PRECOND_NONE, nullptr) != MPCG_OK ||
            mpcg_bd_to_csr_lowertri(h, b.d_S, d_val, 1.0f, 1, nullptr) != MPCG_OK)
            mpcg_compat::die("form_schur_system_qdldl", h);
        if (sqpTimecheck()) break;                                                                            // (:259)
        gpuErrchk(hipDeviceSynchronize());
        if (sqpTimecheck()) break;                                                                            // (:263)
        clock_gettime(CLOCK_MONOTONIC, &linsys_start);                                                        // (:261-265)
        if (mpcg_qdldl_solve_schur(h, ldl, d_val, b.d_gamma, d_lambda, nullptr) != MPCG_OK) mpcg_compat::die("qdldl_solve_schur", h);   // (:268-273)
        gpuErrchk(hipDeviceSynchronize());
        clock_gettime(CLOCK_MONOTONIC, &linsys_end);
        linsys_time_vec.push_back(mpcgpu_compat::time_delta_us(linsys_start, linsys_end));                    // (:276-281)
        if (sqpTimecheck()) break;                                                                            // (:284)
        compute_dz<T>(state_size, control_size, knot_points, b.d_G_dense, b.d_C_dense, b.d_g, d_lambda, b.d_dz);
        if (sqpTimecheck()) break;                                                                            // (:298)
        const bool go_on = st.globalize_and_step(state_size, control_size, knot_points, d_xu, b.d_dz, rho, rho_reset, sqpiter);
        ++sqp_iter;
        if (!go_on) { sqp_time_exit = 0; break; }                                                             // (:345-350)
        if (sqpTimecheck()) break;                                                                            // (:382)
    }
    gpuErrchk(hipFree(d_val));
    mpcg_ldl_destroy(ldl);
    gpuErrchk(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &sqp_solve_end);
    double sqp_solve_time = mpcgpu_compat::time_delta_us(sqp_solve_start, sqp_solve_end);
    return std::make_tuple(linsys_iter_vec, linsys_time_vec, sqp_solve_time, sqp_iter, sqp_time_exit, linsys_exit_vec);
}
