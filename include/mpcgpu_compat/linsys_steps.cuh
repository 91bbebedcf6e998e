// linsys_steps.cuh — source-level stand-ins for the two MPCGPU functions either side of the PCG solve,
// with the reference's own names and signatures, routed to libmpcg_hip.so:
//   form_schur_system<T>(state_size, control_size, knot_points, d_G_dense, d_C_dense, d_g, d_c,
//                        d_S, d_Pinv, d_gamma, rho)              include/pcg/linsys_setup.cuh:620-656
//   compute_dz<T>(state_size, control_size, knot_points, d_Ginv_dense, d_C_dense, d_g_val,
//                 d_lambda, d_dz)                                 include/common/dz.cuh:124-136
// A maintainer includes this INSTEAD of the reference's "pcg/linsys_setup.cuh" / "dz.cuh" bodies (the
// call sites include/pcg/sqp.cuh:207-219 and :250-259 compile unchanged).  One trajectory per call, default
// stream, like the reference.  Needs gbd_pcg_compat/gpu_pcg.cuh for the cached handle.
#pragma once
#include <cstdint>
#include <type_traits>

#include "../gbd_pcg_compat/gpu_pcg.cuh"

template <typename T>
void form_schur_system(uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_G_dense, T* d_C_dense,
                       T* d_g, T* d_c, T* d_S, T* d_Pinv, T* d_gamma, T rho) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "linsys_t is float or double (include/common/settings.cuh:41-49)");
    mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
    int rc;
    if constexpr (std::is_same<T, float>::value)
        rc = mpcg_form_schur(h, control_size, d_G_dense, d_C_dense, d_g, d_c, d_S, d_Pinv, d_gamma, rho, 1, MPCG_PRECOND_SS, /*stream*/ nullptr);
    else
        rc = mpcg_form_schur_f64(h, control_size, d_G_dense, d_C_dense, d_g, d_c, d_S, d_Pinv, d_gamma, rho, 1, MPCG_PRECOND_SS, /*stream*/ nullptr);
    if (rc != MPCG_OK) mpcg_compat::die("form_schur_system", h);
}

template <typename T>
void compute_dz(uint32_t state_size, uint32_t control_size, uint32_t knot_points, T* d_G_dense, T* d_C_dense, T* d_g_val,
                T* d_lambda, T* d_dz) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "linsys_t is float or double (include/common/settings.cuh:41-49)");
    mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
    int rc;
    if constexpr (std::is_same<T, float>::value) rc = mpcg_compute_dz(h, control_size, d_G_dense, d_C_dense, d_g_val, d_lambda, d_dz, 1, /*stream*/ nullptr);
    else rc = mpcg_compute_dz_f64(h, control_size, d_G_dense, d_C_dense, d_g_val, d_lambda, d_dz, 1, /*stream*/ nullptr);
    if (rc != MPCG_OK) mpcg_compat::die("compute_dz", h);
}

// The linear solve of the reference's other path (LINSYS_SOLVE == 0) — D2H(values, gamma), qdldl_solve_schur
// (include/qdldl/sqp.cuh:22-49), H2D(lambda), timed as one region at include/qdldl/sqp.cuh:261-282 — as one GPU call on
// the bd-layout S and gamma that form_schur_system left on the device: block-tridiagonal direct sweep, no host round trip.
template <typename T>
void block_solve_schur(uint32_t state_size, uint32_t knot_points, T* d_S, T* d_gamma, T* d_lambda) {
    static_assert(std::is_same<T, float>::value, "mpcg_block_solve is single precision");
    mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
    if (mpcg_block_solve(h, d_S, d_gamma, d_lambda, 1, /*stream*/ nullptr) != MPCG_OK) mpcg_compat::die("block_solve_schur", h);
}
