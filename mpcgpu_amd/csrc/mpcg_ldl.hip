// mpcg_ldl.hip — C ABI (include/mpcg.h) of LINSYS_SOLVE == 0: the reference's CPU LDL^T path (include/qdldl/sqp.cuh) as a solver the caller
// selects — pure host code (ldl_host.hpp) plus the reference's timed region on device buffers.  Never a fallback of the GPU path.
#include "mpcg_handle.hpp"
#include "ldl_host.hpp"

extern "C" {

// ---- LINSYS_SOLVE == 0: the reference's CPU LDL^T path as a selectable solver (ldl_host.hpp) ----
struct mpcg_ldl { mpcg_ldl_host::Ldl w; std::string err; };

int mpcg_ldl_create(mpcg_ldl** out, uint32_t state_size, uint32_t knot_points) {
    if (!out) return MPCG_ERR_INVALID;
    *out = nullptr;
    if (state_size == 0 || knot_points == 0 || (uint64_t)state_size * knot_points > (1u << 24)) return MPCG_ERR_INVALID;
    mpcg_ldl* l = new (std::nothrow) mpcg_ldl();
    if (!l) return MPCG_ERR_NOMEM;
    if (mpcg_ldl_host::setup(l->w, (int)state_size, (int)knot_points) != 0) { delete l; return MPCG_ERR_INVALID; }
    *out = l;
    return MPCG_OK;
}

int mpcg_ldl_destroy(mpcg_ldl* l) { delete l; return MPCG_OK; }

int mpcg_ldl_pattern(const mpcg_ldl* l, const int32_t** h_col_ptr, const int32_t** h_row_ind, uint32_t* nnz, uint32_t* sum_lnz) {
    if (!l) return MPCG_ERR_INVALID;
    if (h_col_ptr) *h_col_ptr = l->w.Ap.data();
    if (h_row_ind) *h_row_ind = l->w.Ai.data();
    if (nnz) *nnz = (uint32_t)l->w.Ai.size();
    if (sum_lnz) *sum_lnz = (uint32_t)l->w.sumLnz;
    return MPCG_OK;
}

int mpcg_ldl_solve(mpcg_ldl* l, const float* h_val, const float* h_gamma, float* h_lambda) {
    if (!l || !h_val || !h_gamma || !h_lambda) return MPCG_ERR_INVALID;
    if (mpcg_ldl_host::factor(l->w, h_val) < 0) { l->err = "mpcg_ldl_solve: zero pivot"; return MPCG_ERR_INVALID; }
    if (h_lambda != h_gamma) memcpy(h_lambda, h_gamma, sizeof(float) * (size_t)l->w.An);
    mpcg_ldl_host::solve(l->w, h_lambda);
    return MPCG_OK;
}

int mpcg_qdldl_solve_schur(mpcg_handle* h, mpcg_ldl* l, const float* d_val, const float* d_gamma, float* d_lambda, void* stream) {
    if (!h || !l) return MPCG_ERR_INVALID;
    if (!d_val || !d_gamma || !d_lambda) return fail(h, MPCG_ERR_INVALID, "mpcg_qdldl_solve_schur: null device pointer");
    if ((uint32_t)l->w.n != h->n || (uint32_t)l->w.N != h->N) return fail(h, MPCG_ERR_INVALID, "mpcg_qdldl_solve_schur: pattern and handle differ in shape");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the reference's timed region (include/qdldl/sqp.cuh:268-273): D2H values + gamma, factor + solve, H2D lambda
    HIP_TRY(h, hipMemcpyAsync(l->w.val.data(), d_val, sizeof(float) * l->w.val.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(l->w.rhs.data(), d_gamma, sizeof(float) * (size_t)l->w.An, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    const int rc = mpcg_ldl_solve(l, l->w.val.data(), l->w.rhs.data(), l->w.sol.data());
    if (rc != MPCG_OK) return fail(h, rc, "mpcg_qdldl_solve_schur: zero pivot in the LDL^T factorisation");
    HIP_TRY(h, hipMemcpyAsync(d_lambda, l->w.sol.data(), sizeof(float) * (size_t)l->w.An, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    return MPCG_OK;
}

// diagnostic: copy the first `count` u64 words of the cluster scratch (fail flags first) to the host; synchronises the device

}  // extern "C"
