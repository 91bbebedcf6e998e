// mpcg_producers.hip — C ABI (include/mpcg.h) of the steps either side of the solve: Schur + preconditioner formation, dz
// recovery (float and linsys_t = double), the CSR emitter of the QDLDL path and the block-tridiagonal direct solve, over the gfx950 kernels
// in schur_kernels.hip.h / schur_walk.hip.h / schur_walk_f64.hip.h / block_solve.hip.h.
#include "mpcg_handle.hpp"
#ifndef MPCG_DZ64_CAPMUL
#define MPCG_DZ64_CAPMUL 4      // (grid cap of compute_dz_dpp_f64_kernel in units of 64 workgroups per CU; tools/_prof/dz64_ab.py)
#endif
#include "schur_kernels.hip.h"
#include "schur_walk.hip.h"
#include "schur_walk_f64.hip.h"
#include "block_solve.hip.h"

using namespace mpcg;

// The walking Schur kernels' seam buffer (one 14 x 14 Q^-1 per chunk, float or double).  Sized ONCE, on first use, for whatever the automatic
// chunk length can ask of this handle: chunks shorter than 16 rows are chosen only while the call has fewer than 2 x `want` rows per chunk
// length, i.e. at most 2 x want + batch chunks; 16-row chunks beyond — so calls of different batch sizes never reallocate (a reallocation
// inside a stream capture would leave a dangling pointer in the captured graph: ADVICE r04).  Only a FORCED short "schur_chunk" can ask for
// more; that grows the buffer outside a capture and is refused inside one.
static int ensure_seam_buffer(mpcg_handle* h, size_t chunks_needed, size_t elem_bytes, hipStream_t st) {
    const size_t want = (size_t)h->num_cus * 6 * 4;
    const size_t auto_chunks = std::max<size_t>(2 * want + h->max_batch + 4, (size_t)h->max_batch * (size_t)(((size_t)h->N - 1 + 15) / 16));
    // (the automatic part in doubles whatever this call's type: a float call followed by a linsys_t = double call must not reallocate either)
    const size_t need = 196 * std::max(chunks_needed * elem_bytes, auto_chunks * sizeof(double));
    if (h->seam_qinv_bytes >= need) return MPCG_OK;
    { const int rc = alloc_allowed(h, st, "mpcg_form_schur (the seam buffer; a forced short \"schur_chunk\" needs a larger one)"); if (rc != MPCG_OK) return rc; }
    if (h->seam_qinv) {
        HIP_TRY(h, hipDeviceSynchronize());          // (an earlier call's kernels may still read the old buffer)
        HIP_TRY(h, hipFree(h->seam_qinv));
    }
    h->seam_qinv = nullptr; h->seam_qinv_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->seam_qinv, need));
    h->seam_qinv_bytes = need;
    return MPCG_OK;
}

extern "C" {

int mpcg_block_solve(mpcg_handle* h, const float* d_S, const float* d_gamma, float* d_lambda, uint32_t batch, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_block_solve: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_S || !d_gamma || !d_lambda) return fail(h, MPCG_ERR_INVALID, "mpcg_block_solve: null device pointer");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_block_solve: batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->block_scratch) {                      // first call only (not stream-ordered: hipMalloc)
        const int rc = alloc_allowed(h, static_cast<hipStream_t>(stream), "mpcg_block_solve"); if (rc != MPCG_OK) return rc;
    }
    if (!h->block_scratch)
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->block_scratch),
                             (size_t)h->max_batch * h->N * (14 * 14 + 14) * sizeof(float)));       // W_k (14 x 14) + z_k per knot
    BlockSolveArgs a;
    a.S = d_S; a.gamma = d_gamma; a.lambda = d_lambda; a.work = h->block_scratch; a.N = (int)h->N; a.batch = (int)batch;
    // few trajectories: one per wavefront (columns dealt over the four DPP rows, ~2.5x shorter critical path);
    // many: four per wavefront.  Same bits either way.  "block_solve_wide": -1 auto, 0 / 1 forced.
    // (N=128: one per wave 0.35 / 0.53 ms at batch 1024 / 2048 against 0.71 / 0.75; at 4096 four per wave wins, 0.88 vs 0.93)
    const bool wide = h->block_solve_wide < 0 ? batch <= 12u * (uint32_t)h->num_cus : h->block_solve_wide != 0;
    if (wide) hipLaunchKernelGGL(bt_block_solve_wide_kernel, dim3(batch), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(bt_block_solve_kernel, dim3((batch + 3) / 4), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_form_schur(mpcg_handle* h, uint32_t control_size, float* d_G_dense, const float* d_C_dense, const float* d_g,
                    const float* d_c, float* d_S, float* d_Pinv, float* d_gamma, float rho, uint32_t batch,
                    mpcg_precond precond, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_G_dense || !d_C_dense || !d_g || !d_c || !d_S || (!d_Pinv && precond != MPCG_PRECOND_NONE) || !d_gamma)
        return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur: null device pointer");
    if (control_size != 7) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur: control_size must be 7 (IIWA-14)");
    if (precond != MPCG_PRECOND_NONE && precond != MPCG_PRECOND_JACOBI && precond != MPCG_PRECOND_SS)
        return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur: bad preconditioner");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur: batch exceeds max_batch");
    if ((uint64_t)batch * h->N >= (1ull << 31)) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur: batch * knot_points must stay below 2^31");
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = (int)h->n, m = (int)control_size, N = (int)h->N;
    const size_t Gsz = (size_t)(n * n + m * m) * N - m * m;
    // Register-resident formation (schur_walk.hip.h): a 16-lane row walks a chunk of L consecutive block rows, a second kernel closes the
    // seams between chunks.  L trades parallelism against seam work: as long as a call has fewer than ~6 wavefronts of four chunks per
    // CU the chunks are made shorter (L = 1: every row a seam — one trajectory of the MPC loop's own call; 16 at 1024 x 128 knots).
    // (Its kernels address every array through a buffer resource with 31-bit byte offsets: 2,352 B of S per knot => below 913 k knots.)
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->schur_dpp && (uint64_t)batch * N * 2352u < (1ull << 31)) {
        int wL = h->schur_chunk;
        if (wL <= 0) {
            const long rows = (long)batch * (N - 1), want = (long)h->num_cus * 6 * 4;
            wL = 1;
            while (wL < 16 && rows / (2 * wL) >= want) wL *= 2;
        }
        const int wchunks = (N - 1 + wL - 1) / wL;
        // seam buffer: one Q^-1 per chunk (ensure_seam_buffer: sized once for every automatic chunk length of this handle)
        { const int rc_ = ensure_seam_buffer(h, (size_t)batch * wchunks, sizeof(float), st); if (rc_ != MPCG_OK) return rc_; }
        sw::WalkArgs w;
        w.s.G = d_G_dense; w.s.C = d_C_dense; w.s.g = d_g; w.s.c = d_c; w.s.S = d_S; w.s.Pinv = d_Pinv; w.s.gamma = d_gamma;
        w.s.Ginv_scratch = nullptr; w.s.Ginv_out = d_G_dense;
        w.s.rho = rho; w.s.n = n; w.s.m = m; w.s.N = N; w.s.batch = (int)batch; w.s.ss = precond == MPCG_PRECOND_SS; w.s.pinv = precond != MPCG_PRECOND_NONE;
        w.s.k0_only = 0;
        w.seam_qinv = static_cast<float*>(h->seam_qinv); w.L = wL; w.chunks = wchunks;
        h->last_schur_chunk = wL;
        const long capw = (long)h->num_cus * 64;
        long bw = ((long)batch * wchunks + 3) / 4;
        if (bw > capw) bw = capw;
        hipLaunchKernelGGL(sw::schur_walk_kernel, dim3((unsigned)bw), dim3(64), 0, st, w);
        HIP_TRY(h, hipGetLastError());
        if (wchunks > 1) {
            long bs = ((long)batch * (wchunks - 1) + 3) / 4;
            if (bs > capw) bs = capw;
            hipLaunchKernelGGL(sw::schur_seam_kernel, dim3((unsigned)bs), dim3(64), 0, st, w);
            HIP_TRY(h, hipGetLastError());
        }
        return MPCG_OK;
    }
    h->last_schur_chunk = 0;
    // the LDS kernels (schur_kernels.hip.h; option "schur_dpp" = 0, and calls beyond the 31-bit offsets above): one wavefront per knot,
    // G^-1 through a handle-owned staging buffer
    const size_t need = Gsz * h->max_batch;
    if (h->ginv_scratch_floats < need) {          // first call only (not stream-ordered: hipMalloc)
        { const int rc = alloc_allowed(h, st, "mpcg_form_schur (\"schur_dpp\" = 0)"); if (rc != MPCG_OK) return rc; }
        if (h->ginv_scratch) HIP_TRY(h, hipFree(h->ginv_scratch));
        h->ginv_scratch = nullptr; h->ginv_scratch_floats = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->ginv_scratch), need * sizeof(float)));
        h->ginv_scratch_floats = need;
    }
    SchurArgs a;
    a.G = d_G_dense; a.C = d_C_dense; a.g = d_g; a.c = d_c; a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma;
    a.Ginv_scratch = h->ginv_scratch; a.Ginv_out = d_G_dense;
    a.rho = rho; a.n = n; a.m = m; a.N = N; a.batch = (int)batch; a.ss = precond == MPCG_PRECOND_SS; a.pinv = precond != MPCG_PRECOND_NONE;
    a.k0_only = 0;
    long blocks = (long)batch * N;
    const long cap = (long)h->num_cus * 64;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((form_schur_kernel<14, 7>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL((complete_ss_kernel<14, 7>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_compute_dz(mpcg_handle* h, uint32_t control_size, const float* d_Ginv_dense, const float* d_C_dense,
                    const float* d_g, const float* d_lambda, float* d_dz, uint32_t batch, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_compute_dz: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_Ginv_dense || !d_C_dense || !d_g || !d_lambda || !d_dz)
        return fail(h, MPCG_ERR_INVALID, "mpcg_compute_dz: null device pointer");
    if (control_size != 7) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_compute_dz: control_size must be 7 (IIWA-14)");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_compute_dz: batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    DzArgs a{d_Ginv_dense, d_C_dense, d_g, d_lambda, d_dz, (int)h->n, (int)control_size, (int)h->N, (int)batch};
    long blocks = (long)batch * h->N;
    const long cap = (long)h->num_cus * 64;
    if (blocks > cap) blocks = cap;
    if (h->dz_dpp && h->N >= 2 && (uint64_t)batch * h->N * 1176u < (1ull << 31)) {     // (31-bit byte offsets into C)
        long b4 = ((long)batch * h->N + 3) / 4;
        if (b4 > cap * 4) b4 = cap * 4;
        hipLaunchKernelGGL(sw::compute_dz_dpp_kernel, dim3((unsigned)b4), dim3(64), 0, static_cast<hipStream_t>(stream), a);
    } else {
        hipLaunchKernelGGL((compute_dz_kernel<14, 7>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, static_cast<hipStream_t>(stream), a);
    }
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}


// ---- linsys_t = double (USE_DOUBLES = 1, include/common/settings.cuh:41-49): the steps either side of the solve in double precision.
// Round 5: the walking formation and the four-knots-per-wavefront dz recovery in double (schur_walk_f64.hip.h); options "schur_dpp" / "dz_dpp"
// = 0 select the one-wavefront-per-knot LDS kernels of schur_kernels.hip.h instantiated for double.  Same arithmetic order as the float path,
// bit-identical to the oracle's double instantiation either way. ----
int mpcg_form_schur_f64(mpcg_handle* h, uint32_t control_size, double* d_G_dense, const double* d_C_dense, const double* d_g,
                        const double* d_c, double* d_S, double* d_Pinv, double* d_gamma, double rho, uint32_t batch,
                        mpcg_precond precond, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur_f64: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_G_dense || !d_C_dense || !d_g || !d_c || !d_S || (!d_Pinv && precond != MPCG_PRECOND_NONE) || !d_gamma)
        return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur_f64: null device pointer");
    if (control_size != 7) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur_f64: control_size must be 7 (IIWA-14)");
    if (precond != MPCG_PRECOND_NONE && precond != MPCG_PRECOND_JACOBI && precond != MPCG_PRECOND_SS)
        return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur_f64: bad preconditioner");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_form_schur_f64: batch exceeds max_batch");
    if ((uint64_t)batch * h->N >= (1ull << 31)) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_form_schur_f64: batch * knot_points must stay below 2^31");     // (as mpcg_form_schur: the LDS kernels index knots with int)
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = (int)h->n, m = (int)control_size, N = (int)h->N;
    const size_t Gsz = (size_t)(n * n + m * m) * N - m * m;
    // Round 5: the register-resident formation in double (schur_walk_f64.hip.h) — the float path's design, chunk policy and seam buffer; its
    // buffer resources carry 31-bit byte offsets: 4,704 B of S per knot => below 456 k knots per call (beyond: the LDS kernels).
    if (h->schur_dpp && (uint64_t)batch * N * 4704u < (1ull << 31)) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        int wL = h->schur_chunk;
        if (wL <= 0) {
            const long rows = (long)batch * (N - 1), want = (long)h->num_cus * 6 * 4;
            wL = 1;
            while (wL < 16 && rows / (2 * wL) >= want) wL *= 2;
        }
        const int wchunks = (N - 1 + wL - 1) / wL;
        { const int rc_ = ensure_seam_buffer(h, (size_t)batch * wchunks, sizeof(double), st); if (rc_ != MPCG_OK) return rc_; }
        sw64::WalkArgs64 w;
        w.s.G = d_G_dense; w.s.C = d_C_dense; w.s.g = d_g; w.s.c = d_c; w.s.S = d_S; w.s.Pinv = d_Pinv; w.s.gamma = d_gamma;
        w.s.Ginv_scratch = nullptr; w.s.Ginv_out = d_G_dense;
        w.s.rho = rho; w.s.n = n; w.s.m = m; w.s.N = N; w.s.batch = (int)batch; w.s.ss = precond == MPCG_PRECOND_SS; w.s.pinv = precond != MPCG_PRECOND_NONE;
        w.s.k0_only = 0;
        w.seam_qinv = static_cast<double*>(h->seam_qinv); w.L = wL; w.chunks = wchunks;
        h->last_schur_chunk = wL;
        const long capw = (long)h->num_cus * 64;
        long bw = ((long)batch * wchunks + 3) / 4;
        if (bw > capw) bw = capw;
        hipLaunchKernelGGL(sw64::schur_walk_f64_kernel, dim3((unsigned)bw), dim3(64), 0, st, w);
        HIP_TRY(h, hipGetLastError());
        if (wchunks > 1) {
            long bs = ((long)batch * (wchunks - 1) + 3) / 4;
            if (bs > capw) bs = capw;
            hipLaunchKernelGGL(sw64::schur_seam_f64_kernel, dim3((unsigned)bs), dim3(64), 0, st, w);
            HIP_TRY(h, hipGetLastError());
        }
        return MPCG_OK;
    }
    h->last_schur_chunk = 0;
    const size_t need = Gsz * h->max_batch;
    if (h->ginv_scratch_f64_elems < need) {       // first call only (not stream-ordered: hipMalloc)
        { const int rc = alloc_allowed(h, static_cast<hipStream_t>(stream), "mpcg_form_schur_f64 (\"schur_dpp\" = 0)"); if (rc != MPCG_OK) return rc; }
        if (h->ginv_scratch_f64) HIP_TRY(h, hipFree(h->ginv_scratch_f64));
        h->ginv_scratch_f64 = nullptr; h->ginv_scratch_f64_elems = 0;
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->ginv_scratch_f64), need * sizeof(double)));
        h->ginv_scratch_f64_elems = need;
    }
    SchurArgsT<double> a;
    a.G = d_G_dense; a.C = d_C_dense; a.g = d_g; a.c = d_c; a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma;
    a.Ginv_scratch = h->ginv_scratch_f64; a.Ginv_out = d_G_dense;
    a.rho = rho; a.n = n; a.m = m; a.N = N; a.batch = (int)batch; a.ss = precond == MPCG_PRECOND_SS; a.pinv = precond != MPCG_PRECOND_NONE;
    a.k0_only = 0;
    long blocks = (long)batch * N;
    const long cap = (long)h->num_cus * 64;
    if (blocks > cap) blocks = cap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((form_schur_kernel<14, 7, double>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    hipLaunchKernelGGL((complete_ss_kernel<14, 7, double>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_compute_dz_f64(mpcg_handle* h, uint32_t control_size, const double* d_Ginv_dense, const double* d_C_dense,
                        const double* d_g, const double* d_lambda, double* d_dz, uint32_t batch, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_compute_dz_f64: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_Ginv_dense || !d_C_dense || !d_g || !d_lambda || !d_dz)
        return fail(h, MPCG_ERR_INVALID, "mpcg_compute_dz_f64: null device pointer");
    if (control_size != 7) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_compute_dz_f64: control_size must be 7 (IIWA-14)");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_compute_dz_f64: batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    DzArgsT<double> a{d_Ginv_dense, d_C_dense, d_g, d_lambda, d_dz, (int)h->n, (int)control_size, (int)h->N, (int)batch};
    long blocks = (long)batch * h->N;
    const long cap = (long)h->num_cus * 64;
    if (h->dz_dpp && h->N >= 2 && (uint64_t)batch * h->N * 2352u < (1ull << 31)) {      // four knots per wavefront (schur_walk_f64.hip.h); 31-bit byte offsets into C
        // (grid-stride over quads of knots, capped like the float kernel at 4 x cap one-wavefront workgroups.  Round 5 capped this one at cap —
        //  126 VGPRs, four wavefronts per SIMD against the float kernel's seven — which left the 1024 x 128 call two quads per workgroup:
        //  0.1165-0.1175 ms against 0.1126-0.1128 with the float kernel's cap, tools/_prof/dz64_ab.py)
        long bq = (blocks + 3) / 4;
        if (bq > cap * MPCG_DZ64_CAPMUL) bq = cap * MPCG_DZ64_CAPMUL;
        hipLaunchKernelGGL(sw64::compute_dz_dpp_f64_kernel, dim3((unsigned)bq), dim3(64), 0, static_cast<hipStream_t>(stream), a);
        HIP_TRY(h, hipGetLastError());
        return MPCG_OK;
    }
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((compute_dz_kernel<14, 7, double>), dim3((unsigned)blocks), dim3(SCH_THREADS), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_prep_csr(mpcg_handle* h, int32_t* d_col_ptr, int32_t* d_row_ind, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_prep_csr: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_col_ptr || !d_row_ind) return fail(h, MPCG_ERR_INVALID, "mpcg_prep_csr: null device pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(prep_csr_kernel, dim3(h->N), dim3(SCH_THREADS), 0, static_cast<hipStream_t>(stream), (int)h->n, (int)h->N,
                       d_col_ptr, d_row_ind);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_bd_to_csr_lowertri(mpcg_handle* h, const float* d_S, float* d_val, float mult, uint32_t batch, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_bd_to_csr_lowertri: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_S || !d_val) return fail(h, MPCG_ERR_INVALID, "mpcg_bd_to_csr_lowertri: null device pointer");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_bd_to_csr_lowertri: batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    CsrArgs a{d_S, d_val, mult, (int)h->n, (int)h->N, (int)batch};
    long blocks = (long)batch * h->N;
    const long cap = (long)h->num_cus * 64;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(bd_to_csr_kernel, dim3((unsigned)blocks), dim3(SCH_THREADS), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

}  // extern "C"
