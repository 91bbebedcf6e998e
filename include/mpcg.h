/* mpcg.h — C ABI of libmpcg_hip.so: MI355X (gfx950) PCG solver for MPCGPU's block-tridiagonal
 * Schur system  S * lambda = gamma.
 *
 * This is the drop-in boundary for ONE path of A2R-Lab/MPCGPU: the linear-system solve inside
 * sqpSolvePcg (reference include/pcg/sqp.cuh:230).  Citations below are file:line in the reference
 * tree.  The reference's implementation of that path (pcg<T,STATE_SIZE,KNOT_POINTS>, pcg_config,
 * pcgSharedMemSize, checkPcgOccupancy) lives in its un-vendored submodule A2R-Lab/GBD-PCG; the
 * entry points here replace its call sites.
 *
 * Conventions (identical to the reference's device buffers):
 *   - S, Pinv: "bd" layout, per trajectory [N][3][n*n] floats; block (k,col) column-major at
 *     k*3n^2 + col*n^2; col 0/1/2 = left / diagonal / right block of block row k
 *     (include/pcg/linsys_setup.cuh:36-57, 490-507).  Stored NEGATED (:15-19).  Blocks (0,col 0) and
 *     (N-1,col 2) are never written by the reference (:97,118) and are never read here.
 *   - BLOCK SYMMETRY.  PCG needs symmetric S and Pinv, and the reference's are: it writes S[k,right] as the transposed copy of
 *     S[k+1,left] (include/pcg/linsys_setup.cuh:536-557, bit for bit) and forms the symmetric-stair Pinv[k,right] / Pinv[k+1,left] as the
 *     same triple product associated two ways (:97-136; in float they differ by ~3e-5 of the largest entry on the bench's systems).  The register-resident kernels that serve fp32 horizons above 32 knots by
 *     default ("last_kernel_family" 6, 7 and 11) READ ONLY THE LEFT AND DIAGONAL block columns and apply L_{k+1}^T where the reference's kernel
 *     reads block (k,right); the right blocks of d_S / d_Pinv may then hold anything (tests poison them with NaN).  On the reference's
 *     matrices the results agree to that round-off of Pinv (iterates after K iterations: 1e-6 .. 3e-5, inside the fp32 band; bit-identical for S).
 *     THE HANDLE CHECKS THIS ONCE, BY ITSELF (round 4): until it knows, every solve that would run a lower-triangle kernel is launched
 *     guarded — a check kernel tests
 *         max | M[k,right] - M[k+1,left]^T |  <=  1e-2 max | M[k,right], M[k+1,left] |     for M = S and (SS) Pinv, every k
 *     (a test for STRUCTURAL asymmetry — a caller-made Pinv — not for the rounding differences above)
 *     into a flag on the device, the lower-triangle kernel runs only if the flag is clear and a kernel that reads all three block columns
 *     only if it is set — so even the FIRST solve of a caller with a non-symmetric Pinv is the solve of its three columns.  The flag reaches
 *     the host by an asynchronous copy that later calls poll (no call synchronises, nothing is copied per solve once the answer is in):
 *     from then on the handle launches plain lower-triangle kernels ("symmetry_state" 1) or, after a violation, three-column kernels for
 *     good ("symmetry_state" 2; mpcg_last_error() carries a warning).  The latch is per handle: a caller that changes the structure of its
 *     matrices later must use a fresh handle — or "check_symmetry" = 1 (debug), which verifies every solve with a blocking 8-byte copy
 *     and reports the number of offending block pairs as "last_symmetry_violations".  A caller that fills ONLY the left and diagonal block
 *     columns (the right one unwritten) says so with "assume_symmetric" = 1 before its first solve: no check, lower-triangle kernels at
 *     once.  Families 0 and 5 read all three columns anyway; so does family 3 for floats.  DOUBLE PRECISION beyond 32 knots uses the same
 *     latch: the lane-quad kernels (families 9, 10: the lower block triangle in registers) and the streaming kernel (family 3: skips the right
 *     block column, a third of its HBM bytes) run once it says symmetric.  A handle that does not know yet checks the matrices of its first
 *     mpcg_pcg_solve_f64 call with ONE blocking 8-byte copy (never during graph capture: a captured call on such a handle runs a kernel that
 *     reads all three columns — family 8 up to 256 knots, family 3 beyond).
 *   - gamma, lambda: [N][n] floats per trajectory; lambda is in/out (warm start,
 *     include/mpcsim.cuh:186,267,337).
 *   - every pointer named d_* is a DEVICE pointer on the handle's device; `stream` is a hipStream_t
 *     passed as void* (NULL = default stream).  Calls are stream-ordered and never synchronise.
 *   - batched entry points take `batch` independent trajectories stored back to back.
 *   - GRAPH CAPTURE.  Every compute entry point is pure stream work and may be captured into a hipGraph.  What the float PCG entry points need
 *     is allocated by mpcg_create; mpcg_form_schur(_f64), mpcg_block_solve and a FORCED "cluster" on a horizon the automatic policy gives to one
 *     CU allocate a handle-owned work buffer at their first call (hipMalloc is not stream work): make that call once outside the capture — a
 *     first call on a capturing stream returns MPCG_ERR_INVALID with a message and leaves the capture intact.  linsys_t = double beyond 32
 *     knots (mpcg_pcg_solve_f64 / _ref_f64: cluster kernels with a queue + flags buffer and a double-sized copy of lambda0): mpcg_create makes
 *     those buffers while they are small (max_batch x knot_points x state_size x 8 B <= 8 MB — every handle of the C++ shim), larger handles at
 *     their first double solve outside a capture or when the caller sets "reserve_f64" = 1 ahead of it (float-only callers never pay for them).
 *
 * All functions return MPCG_OK (0) or a negative mpcg_status; mpcg_last_error() gives the text.
 * Nothing here falls back to a CPU implementation: without a gfx950 device every compute entry
 * point fails with MPCG_ERR_HIP.
 */
#ifndef MPCG_H
#define MPCG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): the BLOCK SYMMETRY contract is checked by the handle itself (a caller that fills ONLY the left + diagonal block columns must
 * set "assume_symmetric" = 1 — round-3 text allowed garbage in the right blocks without it); options pcg_lpb / cluster_lpb / cluster_lpk /
 * cluster_waves / cluster_adj / schur_fma / schur_inplace are gone; mpcg_probe_hbm_read and "kkt_analytic" are new.
 * This reaches the reference's real-time horizon too: float 16 < N <= 32 at >= 2.5 trajectories per CU runs a lower-triangle kernel since
 * round 5 (below that batch, and at N <= 16, the row-per-lane kernel reads all three columns as before) — such calls are launched GUARDED
 * (check kernel + two gated kernels) until the handle's latch resolves, for good on a handle whose matrices violated the contract, and results
 * differ bitwise between batch sizes either side of the threshold.  mpcg_get_option("symmetry_state") = 2 / the warning in mpcg_last_error()
 * say when a handle has been sent to the three-column kernels; the C++ shim prints that warning once per handle to stderr.
 * Round 6, same ABI: "reserve_f64" (below, GRAPH CAPTURE); a latch that resolved on block-Jacobi calls re-opens at the first SS call; the
 * default float kernel for 32 < knot_points <= 128 is the lane-quad kernel ("pcg_lqb", family 11) — another summation order: results differ
 * from round 5's in the last float bits (inside the stated band), and latency-sized calls at 33..64 knots, which the three-column row-per-lane
 * kernel served, now run a lower-triangle kernel too (guarded until the latch resolves; "pcg_lqb" = 0 restores round 5's choices). */
#define MPCG_ABI_VERSION 2

typedef enum mpcg_status {
    MPCG_OK = 0,
    MPCG_ERR_INVALID = -1,      /* bad argument (null pointer, batch > max_batch, ...) */
    MPCG_ERR_UNSUPPORTED = -2,  /* state_size / knot_points outside the compiled specialisations */
    MPCG_ERR_HIP = -3,          /* a HIP runtime call failed (text in mpcg_last_error) */
    MPCG_ERR_NOMEM = -4
} mpcg_status;

/* Preconditioner stored in d_Pinv (include/pcg/linsys_setup.cuh):
 * block-Jacobi = only the diagonal blocks Pinv[k,1] are read (:202-210, 510-524);
 * symmetric stair = all three block columns (:97-136). */
typedef enum mpcg_precond {
    MPCG_PRECOND_NONE = 0,       /* mpcg_form_schur only: S and gamma, no Pinv at all (for mpcg_block_solve) */
    MPCG_PRECOND_JACOBI = 1,
    MPCG_PRECOND_SS = 3
} mpcg_precond;

typedef struct mpcg_handle mpcg_handle;

/* Library / build identification. */
int mpcg_abi_version(void);
const char *mpcg_build_info(void);

/* Handle = per-device solver context for fixed (state_size, knot_points).  Replaces the
 * per-call cudaMalloc of PCG scratch in sqpSolvePcg (include/pcg/sqp.cuh:116-135): the solver
 * needs no global scratch at all (r, p, upsilon live in LDS), the handle only caches launch
 * configuration and owns three device buffers: the hand-off cells of the cluster kernel (1 KiB per CU, allocated
 * here) and, from their first use on, the staging buffer of mpcg_form_schur and the sweep scratch of
 * mpcg_block_solve.  max_batch bounds `batch` of later calls.  device < 0 = current device.  One handle per
 * (device, knot_points) and per concurrently used stream: calls on the same handle must not overlap on the host
 * side (launch knobs are chosen per call) and their device work must be ordered (one stream, or events) because
 * they share those buffers; different handles are independent. */
/* state_size = 14 (IIWA-14) is the tuned specialisation and the only one every entry point supports.  Any other
 * 1 <= state_size <= 64 gets a handle whose PCG entry points (mpcg_pcg_solve, mpcg_pcg_solve_ref, mpcg_pcg_solve_f64,
 * mpcg_pcg_solve_ref_f64, mpcg_pcg_lds_bytes, mpcg_check_pcg_occupancy) run a generic, functional kernel (matrices streamed
 * every iteration; "last_kernel_family" = 3) — same layouts with n x n blocks, same semantics; the other entry points
 * return MPCG_ERR_UNSUPPORTED on such a handle. */
int mpcg_create(mpcg_handle **out, int device, uint32_t state_size, uint32_t knot_points, uint32_t max_batch);
int mpcg_destroy(mpcg_handle *h);
const char *mpcg_last_error(const mpcg_handle *h);   /* h may be NULL: last error of mpcg_create */

/* Replaces pcgSharedMemSize<T>(state_size, knot_points) (include/pcg/sqp.cuh:151): the dynamic LDS bytes of
 * the launch a default-configured single-trajectory solve makes (the reference passes this value as the launch's
 * smem argument; here the library launches, the number is informational but exact: it equals the
 * "last_kernel_lds_bytes" option after such a solve).  0 if the shape is unsupported.  _f64: the same for
 * linsys_t = double (mpcg_pcg_solve_ref_f64). */
size_t mpcg_pcg_lds_bytes(uint32_t state_size, uint32_t knot_points);
size_t mpcg_pcg_lds_bytes_f64(uint32_t state_size, uint32_t knot_points);

/* Replaces checkPcgOccupancy<T>(kernel, threads, n, N) (examples/track_iiwa_pcg.cu:24).  The
 * reference aborts when its N cooperative blocks cannot be co-resident; this solver has no
 * inter-workgroup dependency, so the check cannot fail for a supported shape.  Writes the number
 * of trajectories one GPU solves concurrently (workgroups/CU * CUs) to *resident_trajectories. */
int mpcg_check_pcg_occupancy(mpcg_handle *h, uint32_t *resident_trajectories);

/* THE HOT PATH.  Replaces
 *   cudaLaunchCooperativeKernel(pcg<T,n,N>, N, PCG_NUM_THREADS, args, smem)  (include/pcg/sqp.cuh:230)
 * for `batch` trajectories at once.  For trajectory b:
 *   r = gamma - S lambda; rt = Pinv r; p = rt; eta = r.rt
 *   repeat <= max_iter: ups = S p; alpha = eta / (p.ups); lambda += alpha p; r -= alpha ups;
 *                       rt = Pinv r; eta' = r.rt; if |eta'| < exit_tol stop;
 *                       p = rt + (eta'/eta) p; eta = eta'
 * d_iters[b]         = completed lambda updates            (reference: d_pcg_iters, sqp.cuh:131-132)
 * d_max_iter_exit[b] = 1 if the loop ran out of iterations (reference: d_pcg_exit,  sqp.cuh:133-135,
 *                      meaning include/mpcsim.cuh:382-387), 0 if |eta| fell below exit_tol.
 * If |eta| < exit_tol already after the setup step, 0 iterations are done and the flag is 0.
 * max_iter / exit_tol are pcg_config<T>::pcg_max_iter / pcg_exit_tol (include/mpcsim.cuh:213-216). */
int mpcg_pcg_solve(mpcg_handle *h,
                   const float *d_S, const float *d_Pinv, const float *d_gamma, float *d_lambda,
                   uint32_t batch, uint32_t max_iter, float exit_tol, mpcg_precond precond,
                   uint32_t *d_iters, uint8_t *d_max_iter_exit, void *stream);

/* Same solve for ONE trajectory with exactly the reference kernel's 12 arguments, in order
 * (include/pcg/sqp.cuh:137-150).  d_r and d_p receive the final residual and search direction
 * (the reference uses them as global scratch); d_v_temp and d_eta_new_temp (the reference's
 * per-block partial-sum buffers, :124-125) are accepted and left untouched.  d_pcg_exit points
 * to a 1-byte bool as in the reference.  The symmetric-stair preconditioner is assumed, as in the
 * reference (its Pinv always carries the off-diagonal blocks). */
int mpcg_pcg_solve_ref(mpcg_handle *h,
                       float *d_S, float *d_Pinv, float *d_gamma, float *d_lambda,
                       float *d_r, float *d_p, float *d_v_temp, float *d_eta_new_temp,
                       uint32_t *d_pcg_iters, uint8_t *d_pcg_exit,
                       uint32_t pcg_max_iter, float pcg_exit_tol, void *stream);

/* Building block P2/P3 of the path, exposed for measurement and reuse: batched block-tridiagonal
 * matrix-vector product  y = M x  with M in bd layout (cols = 3) or its diagonal blocks only
 * (cols = 1).  x, y: [batch][N][n]. */
int mpcg_bt_spmv(mpcg_handle *h, const float *d_M, const float *d_x, float *d_y,
                 uint32_t batch, int cols, void *stream);

/* Measurement aid, not part of the path: read `bytes` bytes at d_src once (16-byte nontemporal loads,
 * two workgroups per CU, nothing else in the kernel) — the HBM read ceiling of THIS device for
 * mpcg_bt_spmv's roofline fraction to be read against (bench.py times it next to the SpMV).
 * d_src 16-byte aligned, bytes a multiple of 16; d_sink: 4 bytes of device memory (written only
 * if the data sum to one particular value). */
int mpcg_probe_hbm_read(mpcg_handle *h, const void *d_src, size_t bytes, float *d_sink, void *stream);

/* ---- reduced-precision matrix storage (BASELINE config 5's fp16 sweep) ----
 * S and Pinv may be kept in IEEE half precision (same bd layout, 2-byte elements): half the HBM
 * footprint and half the bytes of the one load per solve.  Beyond N = 36 the register-resident
 * kernels convert the blocks to fp32 once, while loading them; all arithmetic, gamma and lambda
 * stay fp32.  The solve is then exactly the fp32 PCG of the ROUNDED matrices: its
 * answer differs from the fp32-storage answer by the storage rounding (relative 2^-11 per entry), not
 * by anything iteration-dependent; entries must be within the half range (|x| < 65504).
 * mpcg_convert_f32_to_f16: round-to-nearest-even copy of `count` elements (any bd-layout buffer).
 * mpcg_pcg_solve_f16: mpcg_pcg_solve with d_S16 / d_Pinv16 produced by the conversion. */
int mpcg_convert_f32_to_f16(mpcg_handle *h, const float *d_src, uint16_t *d_dst, size_t count, void *stream);
int mpcg_pcg_solve_f16(mpcg_handle *h,
                       const uint16_t *d_S16, const uint16_t *d_Pinv16, const float *d_gamma, float *d_lambda,
                       uint32_t batch, uint32_t max_iter, float exit_tol, mpcg_precond precond,
                       uint32_t *d_iters, uint8_t *d_max_iter_exit, void *stream);

/* ---- the steps either side of the solve, same layouts as the reference ----
 *
 * mpcg_form_schur replaces form_schur_system<T>(state_size, control_size, knot_points, d_G_dense,
 * d_C_dense, d_g, d_c, d_S, d_Pinv, d_gamma, rho) (include/pcg/linsys_setup.cuh:620-656), batched:
 *   d_G_dense [batch][(n^2+m^2)N - m^2]  in: Q_0,R_0,...,Q_{N-1} (column-major blocks);  out: their
 *             inverses with rho added (the reference's in-place side effect, :371-380, consumed by dz)
 *   d_C_dense [batch][(n^2+nm)(N-1)]     -A_k, -B_k (already negated, include/common/kkt.cuh:115-116)
 *   d_g [batch][(n+m)N - m], d_c [batch][nN]
 *   d_S, d_Pinv, d_gamma                 outputs in the layouts mpcg_pcg_solve consumes
 * precond = MPCG_PRECOND_NONE writes S and gamma only (d_Pinv is not touched and may be NULL): what
 * mpcg_block_solve needs; it saves the inversion of every diagonal block of S and the completion kernel's products.
 * precond = MPCG_PRECOND_JACOBI skips the symmetric-stair completion (:9-137): the off-diagonal
 * blocks of d_Pinv are then left untouched.  The first call allocates a handle-owned staging buffer
 * of max_batch * sizeof(G) (hipMalloc — not stream-ordered); later calls are purely stream-ordered.
 *
 * mpcg_compute_dz replaces compute_dz(state_size, control_size, knot_points, d_Ginv_dense, d_C_dense,
 * d_g, d_lambda, d_dz) (include/common/dz.cuh:124-136), batched; d_dz [batch][(n+m)N - m]. */
int mpcg_form_schur(mpcg_handle *h, uint32_t control_size, float *d_G_dense, const float *d_C_dense,
                    const float *d_g, const float *d_c, float *d_S, float *d_Pinv, float *d_gamma,
                    float rho, uint32_t batch, mpcg_precond precond, void *stream);
int mpcg_compute_dz(mpcg_handle *h, uint32_t control_size, const float *d_Ginv_dense,
                    const float *d_C_dense, const float *d_g, const float *d_lambda, float *d_dz,
                    uint32_t batch, void *stream);

/* ---- CSR side of the reference's QDLDL twin (LINSYS_SOLVE == 0) ----
 * mpcg_prep_csr replaces prep_csr<<<N,64>>>(state_size, knot_points, d_col_ptr, d_row_ind)
 * (include/utils/csr.cuh:40-73; called once per SQP solve at include/qdldl/sqp.cuh:164): pattern of the lower
 * triangle, d_col_ptr [nN+1], d_row_ind [nnz], nnz = (N-1)n^2 + N n(n+1)/2 (include/qdldl/sqp.cuh:148).
 * mpcg_bd_to_csr_lowertri gathers the values form_schur_system_qdldl leaves in d_val
 * (include/qdldl/linsys_setup.cuh:339-351) from a bd-layout S: per trajectory [nnz] floats = mult * (left
 * block, then lower triangle of the diagonal block, row by row).  With the S of mpcg_form_schur (already
 * negated) mult = +1 reproduces the reference's numbers; gamma is shared by both paths.  The CPU LDL^T that consumes
 * them: the reference's own qdldl, or mpcg_ldl_* / mpcg_qdldl_solve_schur below. */
int mpcg_prep_csr(mpcg_handle *h, int32_t *d_col_ptr, int32_t *d_row_ind, void *stream);
int mpcg_bd_to_csr_lowertri(mpcg_handle *h, const float *d_S, float *d_val, float mult, uint32_t batch, void *stream);

/* linsys_t = double (USE_DOUBLES=1, include/common/settings.cuh:41-49): the same solve, same semantics, in double
 * precision.  A functional path (S and Pinv are streamed every iteration), limited to knot_points <= 350 (iterate
 * vectors in LDS).  mpcg_pcg_solve_ref_f64 carries the reference kernel's 12 arguments for pcg<double, n, N>. */
int mpcg_pcg_solve_f64(mpcg_handle *h, const double *d_S, const double *d_Pinv, const double *d_gamma, double *d_lambda,
                       uint32_t batch, uint32_t max_iter, double exit_tol, mpcg_precond precond,
                       uint32_t *d_iters, uint8_t *d_max_iter_exit, void *stream);
int mpcg_pcg_solve_ref_f64(mpcg_handle *h, double *d_S, double *d_Pinv, double *d_gamma, double *d_lambda,
                           double *d_r, double *d_p, double *d_v_temp, double *d_eta_new_temp,
                           uint32_t *d_pcg_iters, uint8_t *d_pcg_exit, uint32_t pcg_max_iter, double pcg_exit_tol,
                           void *stream);

/* The steps either side of the solve for linsys_t = double: mpcg_form_schur / mpcg_compute_dz with every float replaced by double (same
 * layouts, same side effects, same preconditioner choices).  Functional twins — one wavefront per knot, operands in LDS — in the float
 * path's operation order: bit-identical to the oracle's double instantiation (tests/test_gpu_f64.py). */
int mpcg_form_schur_f64(mpcg_handle *h, uint32_t control_size, double *d_G_dense, const double *d_C_dense,
                        const double *d_g, const double *d_c, double *d_S, double *d_Pinv, double *d_gamma,
                        double rho, uint32_t batch, mpcg_precond precond, void *stream);
int mpcg_compute_dz_f64(mpcg_handle *h, uint32_t control_size, const double *d_Ginv_dense,
                        const double *d_C_dense, const double *d_g, const double *d_lambda, double *d_dz,
                        uint32_t batch, void *stream);


/* Batched block-tridiagonal DIRECT solve of S lambda = gamma — the GPU-native counterpart of the reference's second
 * linear-system path (LINSYS_SOLVE == 0: qdldl_solve_schur, include/qdldl/sqp.cuh:22-49, called at :261-282 with
 * D2H(values, gamma) + CPU LDL^T + H2D(lambda) inside the timed region).  Reads d_S / d_gamma exactly as
 * mpcg_form_schur (or form_schur_system) left them, writes d_lambda (no warm start: lambda is output only).
 * Block LU without pivoting, pivot blocks eliminated by the reference's Gauss-Jordan scheme; four trajectories per
 * wavefront, serial in the knot index — the throughput solver for batches (1/50 of the flops of 167 PCG iterations),
 * while mpcg_pcg_solve keeps warm starts and the tolerance knob.  fp32 at cond ~1e5: relative error ~3e-4.
 * Scratch (max_batch x N x 210 floats) is owned by the handle; the first call allocates it, later calls are pure
 * stream work (capturable into a graph). */
int mpcg_block_solve(mpcg_handle* h, const float* d_S, const float* d_gamma, float* d_lambda, uint32_t batch,
                     void* stream);

/* ---- the producer of the path's inputs: KKT block assembly ----
 * mpcg_generate_kkt replaces generate_kkt_submatrices<T><<<knot_points, KKT_THREADS, smem>>>(state_size, control_size, knot_points,
 * d_G_dense, d_C_dense, d_g, d_c, d_dynMem_const, timestep, d_eePos_traj, d_xs, d_xu) (include/common/kkt.cuh:22-163, launched at
 * include/pcg/sqp.cuh:190-204), batched, with the plant functions it calls (include/dynamics/iiwa/iiwa_eepos_plant.cuh:
 * forwardDynamicsAndGradient, trackingCostGradientAndHessian(_lastblock)) and the Euler integrator of
 * include/common/integrator.cuh.  Outputs in the layouts mpcg_form_schur consumes (C holds -A, -B; c_0 = x_0 - x_s).
 * The reference's d_dynMem_const (GRiD's robotModel) becomes an mpcg_plant: the robot as DATA — a fixed-base serial chain
 * of revolute joints given by the tables GRiD tabulates (all column-major; *_trig: entry idx = coef * sin(q_j) for j < nj,
 * coef * cos(q_{j-nj}) otherwise, replacing the constant at idx):
 *   X_const [nj*36] spatial transforms parent -> link, I_spatial [nj*36] spatial inertias, Xhom_const [nj*16] homogeneous
 *   transforms link -> parent.  mpcgpu_amd/data/iiwa14_model.json carries the KUKA iiwa 14 in exactly this form.
 * The cost is the reference's end-effector tracking cost: 1/2 |ee(q_k) - goal_k|^2 (xyz of d_eePos_traj [batch][N][6]) +
 * 1/2 qd_cost |qd|^2 + 1/2 r_cost |u|^2 with its Gauss-Newton Hessian.  Derivatives of the forward dynamics (the A, B blocks) come from the
 * ANALYTIC gradient recursion of the inverse dynamics — what the reference computes with GRiD's forwardDynamicsAndGradient
 * (iiwa_eepos_plant.cuh:127-155): a lane per column (7 x d/dq, 7 x d/dqd) propagates (dv, da) up the chain and dF down, next to the nominal
 * recursion in a 15th lane (csrc/kkt_plant.hip.h) — then -Minv dID; float64 on the device, float outputs.  Option "kkt_analytic" = 0 selects
 * the CHECKER instead: one-sided float64 differences (ID(. + h e_j) - u) / h, h = 3e-8 (the default of rounds 2-3).  Both agree with the float64
 * central-difference host restatement (oracle/iiwa_ref.py) to ~2e-7, the rounding of the float outputs; the analytic recursion stays there for
 * torques of any size (no (ID - u) / h term that amplifies what the explicitly inverted mass matrix leaves of ID(FD(u)) - u).
 * num_joints = 7 is the compiled specialisation.
 * mpcg_plant_create checks what the device kernel relies on and returns MPCG_ERR_UNSUPPORTED / MPCG_ERR_INVALID otherwise: every joint
 * rotates about its own z axis (X_k(q) = blkdiag(Rz(q), Rz(q)) X_k(0), the form of GRiD's tables), the spatial inertias are symmetric and of
 * the rigid-body form [[Ibar, skew(m c)], [skew(m c)^T, m 1]], and Xhom describes the same chain as X (the end-effector position and
 * Jacobian are taken from the spatial transforms on the device; Xhom is used for that consistency check only). */
typedef struct mpcg_plant mpcg_plant;
int mpcg_plant_create(mpcg_plant **out, int device, uint32_t num_joints, const double *X_const, const double *I_spatial,
                      const double *Xhom_const, const int32_t *X_trig_idx, const double *X_trig_coef, const int32_t *X_trig_j,
                      uint32_t n_X_trig, const int32_t *Xhom_trig_idx, const double *Xhom_trig_coef, const int32_t *Xhom_trig_j,
                      uint32_t n_Xhom_trig);
/* The KUKA LBR iiwa 14 of the reference from the tables built into the library (the same numbers as mpcgpu_amd/data/iiwa14_model.json):
 * replaces gato_plant::initializeDynamicsConstMem<T>() (include/dynamics/iiwa/iiwa_eepos_plant.cuh:63-66, called at include/mpcsim.cuh:194). */
int mpcg_plant_create_iiwa14(mpcg_plant **out, int device);
int mpcg_plant_destroy(mpcg_plant *p);
int mpcg_generate_kkt(mpcg_handle *h, const mpcg_plant *plant, uint32_t control_size, float timestep, const float *d_eePos_traj,
                      const float *d_xs, const float *d_xu, float qd_cost, float r_cost, float *d_G_dense, float *d_C_dense,
                      float *d_g, float *d_c, uint32_t batch, void *stream);

/* ---- LINSYS_SOLVE == 0 as a selectable solver: the reference's CPU LDL^T path ----
 * The reference's second linear-system path factors the (negated) Schur matrix on the HOST with QDLDL
 * (include/qdldl/sqp.cuh: pattern prep_csr :164 + QDLDL_etree :193 once per SQP call; per SQP iteration D2H(values,
 * gamma), qdldl_solve_schur = QDLDL_factor + QDLDL_solve :22-49, H2D(lambda) :261-282).  QDLDL (osqp/qdldl, float /
 * int32 build, Makefile:16) is an un-vendored submodule of the reference; libmpcg_hip carries its own implementation of
 * the same published algorithm (elimination tree, up-looking LDL^T without pivoting, triangular solves) so that the
 * A/B switch of include/mpcsim.cuh:21-25 exists in-tree.  These entry points are a SOLVER THE CALLER SELECTS, never a
 * fallback: no GPU entry point calls them.  mpcg_ldl_* are pure host code (no device needed).
 *   mpcg_ldl_create        pattern of the lower triangle in CSR (= include/utils/csr.cuh:40-73, identical to what
 *                          mpcg_prep_csr writes on the device) + elimination tree + workspace, for (state_size, knot_points)
 *   mpcg_ldl_pattern       host pointers to col_ptr [nN+1] / row_ind [nnz]; nnz = (N-1)n^2 + N n(n+1)/2 (:148)
 *   mpcg_ldl_solve         qdldl_solve_schur(:22-49): numeric factorisation of h_val [nnz] + solve; h_lambda may alias h_gamma
 *   mpcg_qdldl_solve_schur the reference's timed region (:268-273) on device buffers: D2H(d_val as mpcg_bd_to_csr_lowertri
 *                          / form_schur_system_qdldl left it, d_gamma), factor + solve, H2D(d_lambda); synchronises `stream`. */
typedef struct mpcg_ldl mpcg_ldl;
int mpcg_ldl_create(mpcg_ldl **out, uint32_t state_size, uint32_t knot_points);
int mpcg_ldl_destroy(mpcg_ldl *l);
int mpcg_ldl_pattern(const mpcg_ldl *l, const int32_t **h_col_ptr, const int32_t **h_row_ind, uint32_t *nnz, uint32_t *sum_lnz);
int mpcg_ldl_solve(mpcg_ldl *l, const float *h_val, const float *h_gamma, float *h_lambda);
int mpcg_qdldl_solve_schur(mpcg_handle *h, mpcg_ldl *l, const float *d_val, const float *d_gamma, float *d_lambda, void *stream);

/* Options (tuning / experiments; defaults are chosen by mpcg_create from knot_points and per call from the batch).
 * Kernel selection of a float solve (state_size 14):
 *   "pcg_rpl" (-1 auto / 0 / 1): the row-per-lane kernel for knot_points <= 64 — a DPP row per knot, vectors in registers; automatic for
 *       knot_points <= 32 (below 2.5 trajectories per CU at 16 < knot_points <= 32; with "pcg_lqb" = 0 also for calls of at most one trajectory per CU up to 64); "rpl_waves" (its wavefronts per trajectory: 0 auto, 4, 8, 16);
 *   "pcg_lpk" (-1 auto / 0 / 1): the lane-pair-per-knot kernel, knot_points <= 128 — everything in registers, a matrix per wavefront; the automatic
 *       choice for fp16 storage only since round 6; = 1 forces it (and keeps the lane-quad kernel out);
 *   "pcg_lqb" (-1 auto / 0 / 1; round 6): the lane-quad-per-knot kernel, knot_points <= 128 — everything in registers, a QUARTER of both matrices
 *       in every wavefront, so that all wavefronts of a workgroup run every pass (two working wavefronts per SIMD).  It takes the lane-pair
 *       kernel's launches (same contract: block lower triangle, gated and fix-up launches, dispatch order); automatic, for both preconditioners, for 32 < knot_points <= 64
 *       at every batch, at 64 < knot_points <= 128, and — with workgroups of 32 knots, four per CU — for
 *       16 < knot_points <= 32 when the call brings at least 2.5 trajectories per CU (block-Jacobi calls run a build of their own); = 0 gives those calls back to the lane-pair kernel (and
 *       the row-per-lane / row-pair kernels their round-5 ranges), = 1 runs it wherever the lane-pair kernel would run;
 *   "cluster" (-1 auto / 0 off / G = 2..8 forced): workgroups (= CUs of one XCD) per trajectory of the clustered lane-pair kernel, automatic
 *       for knot_points > 128 (G = ceil(N / 128)).  Members exchange inner-product partials and boundary knots through the XCD's L2
 *       ("cluster_l2" = 0: always write-through), so all members of a cluster must be resident: the launch holds as many clusters as fit the
 *       chip and each draws trajectories from a queue (any batch = one launch).  A cluster that cannot make progress (a peer not resident:
 *       another stream holds its CU) gives up after a bounded spin and the follow-up launch of a single-workgroup kernel re-solves its
 *       trajectory ("cluster_fixup" = 0: no follow-up launch, d_iters = 0xFFFFFFFF and d_max_iter_exit = 2 for such a trajectory).  The
 *       follow-up launch warm-starts from the handle's own copy of d_lambda, made in front of the cluster launch: members that did finish such a
 *       trajectory have written their knots by then.  ("cluster_test_fail" = 1, tests only: one member gives up at its first write-back.)
 *       linsys_t = double (mpcg_pcg_solve_f64 / _ref_f64): knot_points <= 32 the row-per-lane kernel in double; beyond, with block-symmetric
 *       matrices (the latch), "pcg_lqk" (-1 auto / 0 / 1): the lane-quad-per-knot kernel — the lower block triangle of 64 knots in the registers of
 *       one CU (32 < knot_points <= 64, and 16 < knot_points <= 32 for calls of at least four trajectories per CU; = 1 forces it at any
 *       knot_points <= 64) — and for 64 < knot_points <= 512 its clustered form, G = ceil(N / 64) CUs per
 *       trajectory under the same "cluster" option and hand-off machinery; "pcg_lqk" = 0, a latch that says not symmetric or a capturing first
 *       call: the clustered row-per-lane kernel (32 < knot_points <= 256, G = ceil(N / 32) members, full block rows, all three block columns);
 *       "cluster" = 0 selects the streaming kernel, which is also the clusters' fix-up (up to 350 knots: beyond, its iterate vectors do not fit
 *       LDS and an abandoned trajectory is reported as with "cluster_fixup" = 0);
 *   otherwise (explicit pcg_* knobs, the fix-up launches, fp16 storage at N <= 36) the single-workgroup row-pair kernel: "pcg_waves" (4, 8 or 16
 *       wavefronts per trajectory workgroup), "pcg_reg_rows" (TRIPLES of block rows per matrix and wave kept in registers for the whole
 *       solve; only compiled (waves, rows) pairs are accepted at launch), "pcg_lds_rows" (triples per matrix and wave cached in LDS, -1 =
 *       what fits), "pcg_stream_bufs", "pcg_max_wg_per_cu", "lds_extra" (single-triple LDS slots beyond the uniform cache of the <.,.,1>
 *       kernels: -1 what fits, 0 none); "pcg16_*" = the same for fp16 storage.  Setting any pcg_* knob switches the automatic selection off.
 * "sched_hint" (0 / 1, default 1): an mpcg_pcg_solve call with more trajectories than CUs that runs a register-resident kernel dispatches its
 *       trajectories longest-expected-first, the expectation being the iteration counts the handle's previous call with the same batch size
 *       wrote to d_iters — warm-started solves leave the loop at very different iterations and the dispatch order decides how well the chip stays
 *       filled: -25..35 % on such batches; a scheduling hint only, no result depends on it; one extra ~3 us kernel behind each such solve.
 * "check_symmetry" (debug, 0/1: see BLOCK SYMMETRY above).
 * Around the solve: "schur_dpp" (1: mpcg_form_schur's register-resident formation — a 16-lane DPP row walks a chunk of consecutive block rows, a second
 *       kernel closes the seams between chunks; 0: the LDS kernels; same bits), "schur_chunk" (block rows per chunk: 0 = by call size, from one
 *       row per chunk for a single trajectory to 16 at 1024 x 128 knots; 1..2048 forced; same bits), "dz_dpp" (1: mpcg_compute_dz with four knots
 *       per wavefront, 0: one workgroup per knot; same bits), "block_solve_wide" (mpcg_block_solve: 1 one trajectory per wavefront, 0 four, -1 by
 *       batch size; same bits), "kkt_analytic" (mpcg_generate_kkt: 1 = the analytic gradient recursion of the inverse dynamics, the default;
 *       0 = one-sided float64 differences, the checker), "kkt_f32" (round 6; 1 = the analytic kernel with every recursion in float — linsys_t's own
 *       arithmetic, as the reference's GRiD<float> — and TWO knots per lane in packed float: outputs within 5e-6 of the float64 restatement (relative to max(1, |block|); worst of 1024 windows: 4e-6) instead of
 *       2e-7, 1.6x faster than the default on throughput-sized calls (0.204 against 0.330 ms per 1024 x 127 knots; a single trajectory: 20 against 16 us —
 *       the default is the one for latency-sized calls), a trajectory's results independent of the rest of the batch; 2 = the same arithmetic with one
 *       knot per lane (8 % faster than the default; differs from 1 by float rounding); 0, the default: float64 inside, results rounded to float on
 *       the way out), "nt_loads" / "spmv_blocks_per_cu" (mpcg_bt_spmv), "spmv_mfma" (the MFMA experiment kernel).
 * "assume_symmetric" (0 / 1), "symmetry_state" (read-only; 0 unknown, 1 block-symmetric, 2 violated): see BLOCK SYMMETRY above.
 * "reserve_f64" (= 1: allocate the double cluster kernels' buffers now; see GRAPH CAPTURE above).
 * Read-only: "cluster_fixups" (trajectories re-solved by fix-up launches since mpcg_create — each costs 1.5-4.5 ms of spinning; blocking 8-byte
 *       D2H read), "last_symmetry_violations", "num_cus", "pcg_resident" (1 if the single-workgroup configuration streams nothing inside the PCG
 *       loop), "last_schur_chunk" (block rows per chunk of the last mpcg_form_schur, 0 = the LDS kernels), "last_kernel_family" (kernel of the
 *       last solve: 0 single-workgroup row-pair, 3 generic, 5 row-per-lane, 6 lane-pair-per-knot, 7 clustered lane-pair, 8 clustered row-per-lane
 *       (double), 9 lane-quad-per-knot (double), 10 clustered lane-quad (double), 11 lane-quad-per-knot with both matrices per wavefront (float, round 6); 1, 2, 4 were kernels retired in round 4), "last_kernel_{waves,reg_rows,lds_rows,lds_extra,stream_bufs,cluster,lds_bytes}".
 * WHICH kernel family serves a call depends on knot_points AND, up to 32 knots, on the call's batch (N <= 32: row-per-lane kernel, 8 waves x 1 slot or 4 x 2 by batch; 16 < N <= 32:
 * the lane-quad kernel from 2.5 trajectories per CU; 32 < N <= 128: one kernel at every batch since round 6).  Families sum the inner products in different
 * orders, so the SAME trajectory solved alone and inside a large batch may differ in the last fp32 bits (and, near the tolerance, by an
 * iteration); within one family results are bitwise reproducible run to run and independent of batch composition.  Pin a family with
 * "pcg_rpl" / "pcg_lpk" / "pcg_lqb" / "rpl_waves" when bit-stability across batch sizes matters.  None of the residency knobs changes results within a
 * lane-order family (bitwise identical, tested). */
int mpcg_set_option(mpcg_handle *h, const char *key, int value);
int mpcg_get_option(const mpcg_handle *h, const char *key, int *value);

#ifdef __cplusplus
}
#endif
#endif /* MPCG_H */
