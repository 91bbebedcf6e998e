// mpcg_handle.hpp — what the translation units of libmpcg_hip.so share: the handle behind include/mpcg.h, the error convention, the HIP_TRY
// macro.  The library is four translation units over this header (Makefile: every csrc/*.hip is compiled on its own and linked once):
//   mpcg_pcg.hip        handle / options / the PCG launch policy and entry points (pcg_*.hip.h kernels)
//   mpcg_producers.hip  Schur + preconditioner formation, dz recovery, CSR emitter, block-tridiagonal direct solve (schur_*.hip.h, block_solve.hip.h)
//   mpcg_plant.hip      the robot as data + KKT block assembly (kkt_plant.hip.h)
//   mpcg_ldl.hip        the host LDL^T twin of the reference's QDLDL path (ldl_host.hpp)
// Every kernel header is included by exactly one of them (their non-template kernels have external linkage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <string>

#include "../../include/mpcg.h"

// Launch knobs of the single-workgroup PCG kernels.  The handle holds the user's (or mpcg_create's) values; every
// call works on a COPY that the automatic policy may adjust for that call's batch — the handle is never rewritten
// by a solve (two calls with different batches do not see each other's choices).
struct PcgKnobs {
    int waves = 16;           // wavefronts per trajectory workgroup (4, 8 or 16)
    int reg_rows = 0;         // RT: TRIPLES of block rows per matrix per wave kept in registers (compiled variants only)
    int lds_rows = -1;        // LT: triples per matrix per wave cached in LDS; -1 = as many as fit when reg_rows > 0, else 0
    int waves16 = 8, reg_rows16 = 6, lds_rows16 = -1;   // the same knobs for fp16 matrix storage
    int lds_extra = -1;       // <.,.,1> kernels: single-triple LDS slots beyond the uniform cache (-1 = as many as fit, 0 = none)
    int stream_bufs = -1;     // SB: -1 auto, else 0/1/2 register buffers for the streamed triples
    int max_wg_per_cu = 0;    // 0 = whatever fits; k > 0 pads the LDS request so at most k workgroups share a CU
};

// What the last solve on this handle actually launched (read-only "last_kernel_*" options; tests assert on it).
enum { FAM_NONE = -1, FAM_TRAJ = 0, /* 1, 2, 4: kernels retired in round 4 (HISTORY.md) */ FAM_GENERIC = 3, FAM_RPL = 5, FAM_LPK = 6, FAM_LPKC = 7, FAM_RPLC64 = 8, FAM_LQK64 = 9, FAM_LQKC64 = 10, FAM_LQB = 11 };
struct LastKernel { int family = FAM_NONE, waves = 0, reg_rows = 0, lds_rows = 0, stream_bufs = 0, cluster = 0, lds_bytes = 0, lds_extra = 0; };

struct mpcg_handle {
    int device = 0;
    uint32_t n = 0, N = 0, max_batch = 0;
    int num_cus = 0;
    PcgKnobs k;
    LastKernel last;
    int nt_loads = 1;         // non-temporal hint on the matrix stream
    int rpl = -1;             // row-per-lane kernel (pcg_rpl.hip.h, N <= 64): -1 auto, 0 off, 1 forced
    int rpl_waves = 0;        //   its wavefronts per trajectory: 0 auto, 4 / 8 / 16
    int lqk = -1;             // lane-quad-per-knot kernel in double (pcg_lqk_f64.hip.h, N <= 64): -1 auto (32 < N <= 64 once the latch says block-symmetric), 0 off, 1 forced
    int lpk = -1;             // lane-pair-per-knot kernel (pcg_lpk.hip.h, N <= 128): -1 auto (36 < N <= 128 beyond the row-per-lane kernel's calls), 0 off, 1 forced
    int block_solve_wide = -1; // mpcg_block_solve: one trajectory per wavefront (1), four (0), by batch size (-1)
    int schur_dpp = 1;        // 1: register-resident Schur formation (schur_walk.hip.h: the chunk-walking kernel + its seam kernel), 0: the LDS versions
    int sched_hint = 1;       // dispatch the trajectories of a large call longest-expected-first, predicted by the previous call's iteration counts (sched_order_kernel)
    uint32_t* sched_order = nullptr;   // [1 + max_batch] {batch it was made for, dispatch order}: written after every hinted solve, checked on the device
    int schur_chunk = 0;      //   block rows per chunk of the walking kernel: 0 auto (by call size), 1..2048 forced
    int kkt_analytic = 1;     // mpcg_generate_kkt: 1 = analytic gradient recursion of the inverse dynamics (as the reference's GRiD code), 0 = one-sided float64 differences (the checker)
    int kkt_f32 = 0;          // mpcg_generate_kkt: 1 = the analytic kernel in float arithmetic (linsys_t's own, as the reference's GRiD<float>); 0 = float64 inside
    int dz_dpp = 1;           // 1: four-knots-per-wavefront dz recovery (schur_walk.hip.h), 0: the one-workgroup-per-knot LDS kernel
    int last_schur_chunk = 0; //   what the last mpcg_form_schur used (0: the LDS kernels)
    void* seam_qinv = nullptr;       // schur_walk: one Q^-1 per chunk seam (float or double; ensure_seam_buffer)
    size_t seam_qinv_bytes = 0;
    int cluster = -1;         // workgroups per trajectory of the clustered lane-pair kernel (pcg_lpk_cluster.hip.h): 0 off, -1 auto (N > 128), G > 0 forced
    int cluster_l2 = 1;       // clustered lane-pair kernel: 1 = L2-resident hand-offs when a cluster's members share an XCD, 0 = always write-through
    int lqb = -1;             // the float lane-quad kernel (pcg_lqb.hip.h) in place of the lane-pair kernel: -1 auto, 0 off, 1 wherever the lane-pair kernel would run
    int cluster_fixup = 1;    // 1: a trajectory whose cluster gave up (bounded spin) is re-solved by the single-workgroup kernel
    int cluster_test_fail = 0; // tests only: the last member of cluster 0 gives up at the write-back of its first trajectory (ClusterArgs::test_fail)
    int check_symmetry = 0;   // debug: 1 = every solve that would run a lower-triangle kernel first verifies block symmetry of S and Pinv (synchronises)
    int last_sym_violations = 0;   //   block pairs that failed the check in the last solve (then solved by a three-column kernel)
    // The symmetry latch (default): until the handle knows, every lower-triangle solve is launched GUARDED (check kernel -> device flag ->
    // gated lower-triangle kernel -> gated three-column kernel), and the flag travels to the host by an asynchronous copy that a later
    // call polls: no solve ever synchronises for it.  0 unknown, 1 block-symmetric (plain launches from now on), 2 violated (three-column kernels).
    int sym_state = 0;
    bool sym_pending = false;
    unsigned long long sym_guard_seq = 0, sym_armed_seq = 0;   // guarded launches issued / the one whose flag copy is in flight (sym_poll)
    // Pinv has off-diagonal blocks only in SS calls: a latch that resolved on block-Jacobi calls alone has never seen a Pinv.  sym_pinv_guarded:
    // some guarded check so far included Pinv (the device flag is sticky: an armed copy covers every check issued before it);
    // sym_armed_pinv: its value when the copy in flight was armed; sym_pinv_ok: state 1 covers Pinv too (else the first SS call re-opens the latch).
    bool sym_pinv_guarded = false, sym_armed_pinv = false, sym_pinv_ok = false;
    bool sym_flag_reset = false;   // "assume_symmetric" = 0 after a violation: the sticky device flag is cleared by the next solve, on its stream
    hipEvent_t sym_event = nullptr;
    unsigned long long* sym_host = nullptr;      // pinned
    unsigned long long* cluster_scratch = nullptr;
    // the handle's copy of the caller's lambda ([batch][N][n]) made in front of every cluster launch: what the fix-up launch warm-starts from
    void* lam_backup = nullptr;
    size_t lam_backup_bytes = 0;
    unsigned long long* cluster64_scratch = nullptr;   // the clustered row-per-lane kernel in double (pcg_rpl_cluster_f64.hip.h): queue | flags | cells, first use
    bool auto_cfg = true;     // launch knobs still at mpcg_create's choice (any valid pcg_* set_option clears this)
    bool generic = false;     // state_size != 14: only the PCG entry points work, through pcg_generic_kernel
    int spmv_blocks_per_cu = 3;    // (sweep at 4096 trajectories = 1.2 GB of S, a true HBM stream: profiles/r04_spmv.txt; until round 4: 4)
    int spmv_mfma = 0;        // 1 = the MFMA experiment kernel for mpcg_bt_spmv
    float* block_scratch = nullptr;  // W_k, z_k of mpcg_block_solve: max_batch x N x 210 floats (first call)
    float* ginv_scratch = nullptr;   // staging for the in-place G <- G^-1 of mpcg_form_schur
    size_t ginv_scratch_floats = 0;
    double* ginv_scratch_f64 = nullptr;   // the same for mpcg_form_schur_f64
    size_t ginv_scratch_f64_elems = 0;
    std::string err;
};

// error text of a failed mpcg_create (no handle yet); defined in mpcg_pcg.hip
extern thread_local std::string mpcg_create_err;

static inline int fail(mpcg_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else mpcg_create_err = msg;
    return code;
}
#define HIP_TRY(h, expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((h), MPCG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// hipFree / hipEventDestroy / hipHostFree are "unsafe" calls while ANOTHER stream of the calling thread is being captured in the default (global)
// capture mode: they invalidate that capture.  A destroy can come at any time — a garbage collector finalising an old handle while the caller
// captures a solve of a new one (tests/test_gpu_graph.py, seen as a one-in-two failure of a capture inside the full suite) — so the destroy
// functions run in relaxed mode, which permits them.
struct RelaxedCaptureScope {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    RelaxedCaptureScope() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    ~RelaxedCaptureScope() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

// Buffers the handle allocates at the first call that needs them (hipMalloc is not stream work): a call that would have to allocate while its
// stream is being captured is refused — message, MPCG_ERR_INVALID, capture intact — instead of failing inside the capture.  (mpcg.h, GRAPH CAPTURE)
static inline int alloc_allowed(mpcg_handle* h, hipStream_t st, const char* what) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) return MPCG_OK;
    return fail(h, MPCG_ERR_INVALID, std::string(what) + ": the handle allocates a work buffer at the first such call — make one outside the stream capture first");
}
