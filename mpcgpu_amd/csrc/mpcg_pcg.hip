// mpcg_pcg.hip — C ABI (include/mpcg.h) of the hot path: handle and options, the PCG launch policy (which kernel family serves a call), the
// PCG / SpMV entry points, over the gfx950 kernels in pcg_*.hip.h.  Host side is plain C++/HIP: no torch types, no CPU fallback.
#include "mpcg_handle.hpp"
#include "pcg_kernels.hip.h"
#include "pcg_lpk.hip.h"
#include "pcg_lqb.hip.h"
#include "pcg_lpk_cluster.hip.h"
#include "pcg_rpl.hip.h"
#include "pcg_rpl_cluster_f64.hip.h"
#include "pcg_lqk_f64.hip.h"
#include "pcg_lqk_cluster_f64.hip.h"
#include "pcg_f64.hip.h"

using namespace mpcg;

thread_local std::string mpcg_create_err;

// the symmetry latch: has the asynchronous copy of the device flag landed?
// (never while `st` is being captured into a graph: an event query is not a capturable operation and would invalidate the capture —
//  a capturing caller stays on the guarded launches, which are plain kernel launches)
static void sym_poll(mpcg_handle* h, hipStream_t st = nullptr, bool have_stream = false) {
    if (!h->sym_pending) return;
    if (have_stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    }
    // (the query itself in relaxed capture mode: with ANOTHER stream of this thread being captured in global mode an event query is an
    //  "unsafe" call that would invalidate that unrelated capture — mpcg_get_option("symmetry_state") passes no stream at all)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    const hipError_t q = hipEventQuery(h->sym_event);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (q != hipSuccess) return;
    h->sym_pending = false;
    if (*h->sym_host == 0 && h->sym_guard_seq != h->sym_armed_seq) {
        // guarded solves were issued AFTER this copy was armed: a clean flag says nothing about their matrices.  Stay unknown — the next
        // guarded call arms a new copy, which (the device flag is sticky) covers them all.
        return;
    }
    if (*h->sym_host) {
        h->sym_state = 2;
        h->err = "warning: S / Pinv of a solve on this handle were not block-symmetric (block (k, right) != block (k+1, left)^T): the handle now "
                 "runs kernels that read all three block columns (include/mpcg.h, BLOCK SYMMETRY)";
    } else {
        h->sym_state = 1;
        h->sym_pinv_ok = h->sym_armed_pinv;
    }
}

// n = 14 is the tuned specialisation (every entry point); any other 1 <= n <= 64 is served by the generic PCG kernel only
// (mpcg_pcg_solve / _ref / _f64: pcg_generic_kernel, pcg_f64.hip.h), as long as its iterate vectors fit the LDS.
static bool shape_supported(uint32_t n, uint32_t N) { return n == (uint32_t)NS && N >= 2 && N <= 2048; }
static bool generic_shape_supported(uint32_t n, uint32_t N) {
    return n >= 1 && n <= 64 && n != (uint32_t)NS && N >= 2 && N <= 2048 && pcg_generic_lds_elems((int)N, (int)n) * sizeof(float) <= 160 * 1024;
}

static size_t lds_bytes_for(uint32_t N, int nw) { return pcg_lds_floats((int)N, nw) * sizeof(float); }
static constexpr size_t kLdsMax = 160 * 1024;
static constexpr uint32_t kRplMaxN = 64;          // row-per-lane kernel: full block rows of S and Pinv in registers
static constexpr uint32_t kLpbMaxN = 128;         // one block per lane: 8 waves x 64 lanes hold 2 (2N - 1) blocks

static int stream_bufs_for(const mpcg_handle* h, const PcgKnobs& k, int nw, int esz);
static void choose_auto(const mpcg_handle* h, PcgKnobs& k, uint32_t batch, int esz);
static constexpr uint32_t kLqkMaxN = 64;      // linsys_t = double: knots one CU holds as lane quads (pcg_lqk_f64.hip.h)
static constexpr uint32_t kCluster64MaxN = 64 * LQKC_MAX_G;      // linsys_t = double: the longest horizon a cluster kernel holds (eight members of 64 knots)
static size_t default_launch_lds_bytes(uint32_t N, int num_cus);

extern "C" {

int mpcg_abi_version(void) { return MPCG_ABI_VERSION; }

const char* mpcg_build_info(void) {
    return "libmpcg_hip gfx950 fp32 n=14 (row-per-lane DPP PCG for short horizons, lane-pair register-resident PCG to N = 128, clustered lane-pair PCG beyond, wave64 row-triple streaming PCG; linsys_t = double: row-per-lane to N = 32, lane-quad register-resident PCG to N = 64, clustered to 512; chunk-walking Schur formation in float and double; IIWA-14 KKT blocks with the analytic inverse-dynamics gradient)";
}

// device scratch of the clustered kernel: [queue line][flags: one 128-byte line per trajectory of the CALL, up to max_batch][cells: 1 KB
// per member of the launch, up to two members per CU]
static size_t cluster_alloc_words(const mpcg_handle* h) {
    return (size_t)2 * h->num_cus * LPBC_WG_WORDS + CL_FLAG_STRIDE + (size_t)h->max_batch * CL_FLAG_STRIDE
           + 16;                         // + one line for the "cluster_fixups" counter (never re-zeroed by a launch)
}
static unsigned long long* fixup_counter(const mpcg_handle* h) { return h->cluster_scratch + cluster_alloc_words(h) - 16; }
// A handle whose latch resolved on block-Jacobi calls knows S only: the first SS call (Pinv has off-diagonal blocks) re-opens it, so that its
// Pinv goes through a check before a lower-triangle kernel reads it.  Also where a pending reset of the device flag is issued.
static int sym_prepare(mpcg_handle* h, int pcols, hipStream_t st) {
    if (h->sym_state == 1 && pcols == 3 && !h->sym_pinv_ok) h->sym_state = 0;
    if (h->sym_flag_reset) {
        hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, st, fixup_counter(h) + 9, (size_t)1);
        HIP_TRY(h, hipGetLastError());
        h->sym_flag_reset = false;
    }
    return MPCG_OK;
}
// The handle's copy of lambda0 (mpcg_handle::lam_backup).  mpcg_create sizes it for every horizon the automatic policy gives to a cluster kernel;
// a forced "cluster" = G on a shorter horizon allocates at its first launch (hipMalloc: not inside a stream capture).
static int ensure_lam_backup(mpcg_handle* h, size_t bytes, hipStream_t st) {
    if (h->lam_backup_bytes >= bytes) return MPCG_OK;
    { const int rc = alloc_allowed(h, st, "a cluster launch whose copy of lambda mpcg_create did not size (a forced \"cluster\" on a horizon the automatic policy gives to one CU; a "
                                          "first double solve on a handle created with more than 8 MB of double iterates: \"reserve_f64\")"); if (rc != MPCG_OK) return rc; }
    if (h->lam_backup) {
        HIP_TRY(h, hipDeviceSynchronize());              // (an earlier call's fix-up launch may still read the old copy)
        HIP_TRY(h, hipFree(h->lam_backup));
        h->lam_backup = nullptr; h->lam_backup_bytes = 0;
    }
    HIP_TRY(h, hipMalloc(&h->lam_backup, bytes));
    h->lam_backup_bytes = bytes;
    return MPCG_OK;
}
// queue + flags + cells zeroed and lambda copied, one launch (cluster_prologue_kernel)
static int launch_cluster_prologue(mpcg_handle* h, unsigned long long* words, size_t zw, const void* lambda, size_t lam_bytes, hipStream_t st) {
    const int rc = ensure_lam_backup(h, lam_bytes, st);
    if (rc != MPCG_OK) return rc;
    const size_t work = zw > lam_bytes / 16 ? zw : lam_bytes / 16;
    size_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks * 256 < zw) blocks = (zw + 255) / 256;
    hipLaunchKernelGGL(cluster_prologue_kernel, dim3((unsigned)blocks), dim3(256), 0, st, words, zw, static_cast<const uint32_t*>(lambda),
                       static_cast<uint32_t*>(h->lam_backup), lam_bytes / 4);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}
// no fix-up launch behind a cluster launch: report the trajectories whose completion count is short of G (cluster_report_kernel)
static int launch_cluster_report(mpcg_handle* h, const unsigned long long* flags, int G, uint32_t batch, uint32_t* iters, uint8_t* exits, hipStream_t st) {
    hipLaunchKernelGGL(cluster_report_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, flags, (int)CL_FLAG_STRIDE, (unsigned long long)G, (int)batch, iters, exits);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}
static size_t cluster64_alloc_words(const mpcg_handle* h);
static constexpr size_t kEagerF64Bytes = (size_t)8 << 20;
// the double cluster kernels' buffers: queue + flags + cells, and the double-sized copy of lambda0 their fix-up launch starts from.  Blocking
// (hipMalloc): mpcg_create for small handles, "reserve_f64" = 1, or the first double cluster solve outside a capture.
static int reserve_f64(mpcg_handle* h) {
    if (h->generic || h->N <= 32 || h->N > kCluster64MaxN) return MPCG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->cluster64_scratch) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->cluster64_scratch), cluster64_alloc_words(h) * sizeof(unsigned long long)));
        HIP_TRY(h, hipMemset(h->cluster64_scratch, 0, cluster64_alloc_words(h) * sizeof(unsigned long long)));
    }
    const size_t bytes = (size_t)h->max_batch * h->N * h->n * sizeof(double);
    if (h->lam_backup_bytes < bytes) {
        if (h->lam_backup) {
            HIP_TRY(h, hipDeviceSynchronize());              // (an earlier call's fix-up launch may still read the old copy)
            HIP_TRY(h, hipFree(h->lam_backup));
            h->lam_backup = nullptr; h->lam_backup_bytes = 0;
        }
        HIP_TRY(h, hipMalloc(&h->lam_backup, bytes));
        h->lam_backup_bytes = bytes;
    }
    return MPCG_OK;
}

size_t mpcg_pcg_lds_bytes(uint32_t state_size, uint32_t knot_points) {
    if (generic_shape_supported(state_size, knot_points)) return pcg_generic_lds_elems((int)knot_points, (int)state_size) * sizeof(float);
    if (!shape_supported(state_size, knot_points)) return 0;
    if (knot_points > kLpbMaxN && lds_bytes_for(knot_points, 16) > kLdsMax) return 0;
    return default_launch_lds_bytes(knot_points, 256);
}

size_t mpcg_pcg_lds_bytes_f64(uint32_t state_size, uint32_t knot_points) {
    if (!shape_supported(state_size, knot_points) && !generic_shape_supported(state_size, knot_points)) return 0;
    if (state_size == NS && knot_points <= 32) return pcg_rpl_lds_floats((int)knot_points, knot_points <= 16 ? 4 : 8) * sizeof(double);   // row-per-lane kernel
    if (state_size == NS && knot_points <= kLqkMaxN) return pcg_lqk_lds_doubles(8) * sizeof(double);                                   // lane-quad kernel (block-symmetric matrices: the reference's)
    if (state_size == NS && knot_points <= kCluster64MaxN) return pcg_lqkc_lds_doubles() * sizeof(double);                             // a member of the clustered lane-quad kernel
    const size_t b = pcg_generic_lds_elems((int)knot_points, (int)state_size) * sizeof(double);
    return b <= kLdsMax ? b : 0;
}

int mpcg_create(mpcg_handle** out, int device, uint32_t state_size, uint32_t knot_points, uint32_t max_batch) {
    if (!out) return fail(nullptr, MPCG_ERR_INVALID, "mpcg_create: out is null");
    *out = nullptr;
    const bool generic = generic_shape_supported(state_size, knot_points);
    if (!generic && (!shape_supported(state_size, knot_points) || lds_bytes_for(knot_points, 16) > kLdsMax))
        return fail(nullptr, MPCG_ERR_UNSUPPORTED,
                    "mpcg_create: state_size = 14 (tuned; every entry point) or 1..64 (generic PCG kernel only), 2 <= knot_points, "
                    "and the iterate vectors must fit 160 KiB of LDS");
    if (max_batch == 0) return fail(nullptr, MPCG_ERR_INVALID, "mpcg_create: max_batch is 0");
    int ndev = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(nullptr, MPCG_ERR_HIP, "mpcg_create: no HIP device");
    if (device < 0) HIP_TRY(nullptr, hipGetDevice(&device));
    if (device >= ndev) return fail(nullptr, MPCG_ERR_INVALID, "mpcg_create: device out of range");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MPCG_ERR_UNSUPPORTED, std::string("mpcg_create: built for gfx950, device is ") + prop.gcnArchName);
    mpcg_handle* h = new (std::nothrow) mpcg_handle();
    if (!h) return fail(nullptr, MPCG_ERR_NOMEM, "mpcg_create: out of host memory");
    h->device = device; h->n = state_size; h->N = knot_points; h->max_batch = max_batch;
    h->generic = generic;
    h->num_cus = prop.multiProcessorCount;
    h->nt_loads = 1;                    // SpMV kernel only: the matrix is read once — non-temporal loads, +4..9 % (profiles/r02_tune_spmv.txt, r04_spmv.txt)
    choose_auto(h, h->k, 1, 4);         // knobs of the single-workgroup kernels as a batch-1 call would pick them
    choose_auto(h, h->k, 1, 2);
    // hand-off cells of the cluster kernel (512 B per member, up to two members per CU), allocated here so that every solve is pure stream work
    // and can be captured into a hipGraph
    if (hipSetDevice(device) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&h->cluster_scratch), cluster_alloc_words(h) * sizeof(unsigned long long)) != hipSuccess) {
        delete h;
        return fail(nullptr, MPCG_ERR_NOMEM, "mpcg_create: cannot allocate the cluster scratch");
    }
    (void)hipMemset(h->cluster_scratch, 0, cluster_alloc_words(h) * sizeof(unsigned long long));
    if (hipMalloc(reinterpret_cast<void**>(&h->sched_order), ((size_t)max_batch + 1) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(h->sched_order, 0, ((size_t)max_batch + 1) * sizeof(uint32_t)) != hipSuccess) {      // tag 0 = no permutation yet (sched_pick)
        (void)hipFree(h->cluster_scratch);
        delete h;
        return fail(nullptr, MPCG_ERR_NOMEM, "mpcg_create: cannot allocate the dispatch-order buffer");
    }
    // linsys_t = double beyond N = 32 runs cluster kernels with a queue + flags + cells buffer of their own and a DOUBLE-sized copy of lambda0.
    // A float caller never needs either (N = 128, max_batch 4096 = 59 MB per handle for nothing), so mpcg_create makes them only
    // while they are small (<= 8 MB of double iterates: every handle of the C++ shim, max_batch = 1) — then a fresh handle's first
    // mpcg_pcg_solve_f64 may still be a captured one; larger handles allocate at the first double solve, outside a capture, or when the caller
    // sets "reserve_f64" = 1 (mpcg.h, GRAPH CAPTURE).  Float clusters (N > 128) get their float-sized copy here as before.
    if (!generic && knot_points > 32) {
        const size_t bytes64 = (size_t)max_batch * knot_points * state_size * sizeof(double);
        if (knot_points <= kCluster64MaxN && bytes64 <= kEagerF64Bytes) {
            const int rc = reserve_f64(h);
            if (rc != MPCG_OK) { const std::string e = h->err; (void)mpcg_destroy(h); return fail(nullptr, rc, "mpcg_create: " + e); }
        } else if (knot_points > kLpbMaxN) {
            const size_t bytes = (size_t)max_batch * knot_points * state_size * sizeof(float);
            if (hipMalloc(&h->lam_backup, bytes) != hipSuccess) {
                (void)mpcg_destroy(h);
                return fail(nullptr, MPCG_ERR_NOMEM, "mpcg_create: cannot allocate the copy of lambda the cluster fix-up starts from");
            }
            h->lam_backup_bytes = bytes;
        }
    }
    if (hipEventCreateWithFlags(&h->sym_event, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->sym_host), sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        (void)mpcg_destroy(h);
        return fail(nullptr, MPCG_ERR_NOMEM, "mpcg_create: cannot allocate the symmetry latch");
    }
    *h->sym_host = 0;
    *out = h;
    return MPCG_OK;
}

int mpcg_destroy(mpcg_handle* h) {
    if (h) {
        RelaxedCaptureScope relaxed;
        (void)hipSetDevice(h->device);
        if (h->lam_backup) (void)hipFree(h->lam_backup);
        if (h->block_scratch) (void)hipFree(h->block_scratch);
        if (h->ginv_scratch) (void)hipFree(h->ginv_scratch);
        if (h->seam_qinv) (void)hipFree(h->seam_qinv);
        if (h->sym_event) (void)hipEventDestroy(h->sym_event);
        if (h->sym_host) (void)hipHostFree(h->sym_host);
        if (h->ginv_scratch_f64) (void)hipFree(h->ginv_scratch_f64);
        if (h->cluster_scratch) (void)hipFree(h->cluster_scratch);
        if (h->cluster64_scratch) (void)hipFree(h->cluster64_scratch);
        if (h->sched_order) (void)hipFree(h->sched_order);
    }
    delete h;
    return MPCG_OK;
}

const char* mpcg_last_error(const mpcg_handle* h) { return h ? h->err.c_str() : mpcg_create_err.c_str(); }

// key -> int member of the handle (plain knobs without range checks)
static int* knob_ptr(mpcg_handle* h, const char* key) {
    if (!strcmp(key, "pcg_reg_rows")) return &h->k.reg_rows;          // validated at launch
    if (!strcmp(key, "pcg_lds_rows")) return &h->k.lds_rows;
    if (!strcmp(key, "pcg_stream_bufs")) return &h->k.stream_bufs;
    if (!strcmp(key, "lds_extra")) return &h->k.lds_extra;
    if (!strcmp(key, "block_solve_wide")) return &h->block_solve_wide;
    if (!strcmp(key, "pcg16_waves")) return &h->k.waves16;
    if (!strcmp(key, "pcg16_reg_rows")) return &h->k.reg_rows16;
    if (!strcmp(key, "pcg16_lds_rows")) return &h->k.lds_rows16;
    return nullptr;
}

int mpcg_set_option(mpcg_handle* h, const char* key, int value) {
    if (!h || !key) return MPCG_ERR_INVALID;
    // an explicit pcg_* knob switches the automatic per-call configuration off — once the key has validated
    const bool is_pcg = !strncmp(key, "pcg_", 4);
    if (!strcmp(key, "pcg_waves")) {
        if (value != 4 && value != 8 && value != 16) return fail(h, MPCG_ERR_INVALID, "pcg_waves must be 4, 8 or 16");
        h->k.waves = value; h->auto_cfg = false; return MPCG_OK;
    }
    if (!strcmp(key, "pcg_max_wg_per_cu")) {
        if (value < 0 || value > 8) return fail(h, MPCG_ERR_INVALID, "pcg_max_wg_per_cu out of range");
        h->k.max_wg_per_cu = value; h->auto_cfg = false; return MPCG_OK;
    }
    if (!strcmp(key, "pcg_rpl")) {
        if (value < -1 || value > 1) return fail(h, MPCG_ERR_INVALID, "pcg_rpl must be -1 (auto), 0 or 1");
        if (value == 1 && h->N > kRplMaxN) return fail(h, MPCG_ERR_UNSUPPORTED, "pcg_rpl: the row-per-lane kernel holds knot_points <= 64");
        h->rpl = value; return MPCG_OK;
    }
    if (!strcmp(key, "rpl_waves")) {
        if (value != 0 && value != 4 && value != 8 && value != 16) return fail(h, MPCG_ERR_INVALID, "rpl_waves must be 0 (auto), 4, 8 or 16");
        h->rpl_waves = value; return MPCG_OK;
    }
    if (!strcmp(key, "pcg_lqk")) {
        if (value < -1 || value > 1) return fail(h, MPCG_ERR_INVALID, "pcg_lqk must be -1 (auto), 0 (off) or 1 (forced)");
        if (value == 1 && h->N > kLqkMaxN) return fail(h, MPCG_ERR_UNSUPPORTED, "pcg_lqk: the lane-quad kernel holds knot_points <= 64");
        h->lqk = value; return MPCG_OK;
    }
    if (!strcmp(key, "pcg_lqb")) {
        if (value < -1 || value > 1) return fail(h, MPCG_ERR_INVALID, "pcg_lqb must be -1 (auto), 0 (off) or 1 (in place of the lane-pair kernel)");
        h->lqb = value; return MPCG_OK;
    }
    if (!strcmp(key, "pcg_lpk")) {
        if (value < -1 || value > 1) return fail(h, MPCG_ERR_INVALID, "pcg_lpk must be -1 (auto), 0 (off) or 1 (forced)");
        if (value == 1 && h->N > kLpbMaxN) return fail(h, MPCG_ERR_UNSUPPORTED, "pcg_lpk: the lane-pair kernel holds knot_points <= 128");
        h->lpk = value; return MPCG_OK;
    }
    if (int* p = knob_ptr(h, key)) { *p = value; if (is_pcg) h->auto_cfg = false; return MPCG_OK; }
    if (!strcmp(key, "nt_loads")) { h->nt_loads = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "cluster_fixup")) { h->cluster_fixup = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "cluster_l2")) { h->cluster_l2 = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "reserve_f64")) { return value ? reserve_f64(h) : MPCG_OK; }
    if (!strcmp(key, "cluster_test_fail")) { h->cluster_test_fail = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "check_symmetry")) { h->check_symmetry = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "assume_symmetric")) {      // 1: the caller vouches (or fills only the lower block triangle): no check; 0: back to "unknown"
        // (1 vouches for Pinv too.  0 also forgets a violation: the sticky device flag is cleared by the next solve, on its stream)
        h->sym_state = value ? 1 : 0; h->sym_pending = false; h->sym_pinv_ok = value != 0;
        h->sym_pinv_guarded = h->sym_armed_pinv = false;
        if (!value) h->sym_flag_reset = true;
        return MPCG_OK;
    }
    if (!strcmp(key, "schur_dpp")) { h->schur_dpp = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "schur_chunk")) { if (value < 0 || value > 2048) return fail(h, MPCG_ERR_INVALID, "schur_chunk must be 0 (auto) or 1..2048 block rows"); h->schur_chunk = value; return MPCG_OK; }
    if (!strcmp(key, "dz_dpp")) { h->dz_dpp = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "kkt_analytic")) { h->kkt_analytic = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "kkt_f32")) { h->kkt_f32 = value == 2 ? 2 : value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "sched_hint")) { h->sched_hint = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "cluster")) {
        if (value < -1 || value > 32) return fail(h, MPCG_ERR_INVALID, "cluster must be -1 (auto), 0 (off) or 1..32 workgroups per trajectory");
        h->cluster = value; return MPCG_OK;
    }
    if (!strcmp(key, "spmv_mfma")) { h->spmv_mfma = value ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "spmv_blocks_per_cu")) {
        if (value < 1 || value > 64) return fail(h, MPCG_ERR_INVALID, "spmv_blocks_per_cu out of range");
        h->spmv_blocks_per_cu = value; return MPCG_OK;
    }
    return fail(h, MPCG_ERR_INVALID, std::string("unknown option ") + key);
}

int mpcg_get_option(const mpcg_handle* h, const char* key, int* value) {
    if (!h || !key || !value) return MPCG_ERR_INVALID;
    if (!strcmp(key, "pcg_waves")) { *value = h->k.waves; return MPCG_OK; }
    if (!strcmp(key, "pcg_max_wg_per_cu")) { *value = h->k.max_wg_per_cu; return MPCG_OK; }
    if (!strcmp(key, "pcg_lpk")) { *value = h->lpk; return MPCG_OK; }
    if (!strcmp(key, "pcg_lqb")) { *value = h->lqb; return MPCG_OK; }
    if (!strcmp(key, "pcg_lqk")) { *value = h->lqk; return MPCG_OK; }
    if (!strcmp(key, "pcg_rpl")) { *value = h->rpl; return MPCG_OK; }
    if (!strcmp(key, "rpl_waves")) { *value = h->rpl_waves; return MPCG_OK; }
    if (const int* p = knob_ptr(const_cast<mpcg_handle*>(h), key)) { *value = *p; return MPCG_OK; }
    if (!strcmp(key, "nt_loads")) { *value = h->nt_loads; return MPCG_OK; }
    if (!strcmp(key, "pcg_resident")) { *value = stream_bufs_for(h, h->k, h->k.waves, 4) == 0; return MPCG_OK; }   // 1: the single-workgroup configuration streams nothing
    if (!strcmp(key, "cluster")) { *value = h->cluster; return MPCG_OK; }
    if (!strcmp(key, "cluster_fixup")) { *value = h->cluster_fixup; return MPCG_OK; }
    if (!strcmp(key, "cluster_l2")) { *value = h->cluster_l2; return MPCG_OK; }
    if (!strcmp(key, "check_symmetry")) { *value = h->check_symmetry; return MPCG_OK; }
    if (!strcmp(key, "last_symmetry_violations")) { *value = h->last_sym_violations; return MPCG_OK; }
    if (!strcmp(key, "symmetry_state")) { sym_poll(const_cast<mpcg_handle*>(h)); *value = h->sym_state; return MPCG_OK; }
    if (!strcmp(key, "schur_dpp")) { *value = h->schur_dpp; return MPCG_OK; }
    if (!strcmp(key, "schur_chunk")) { *value = h->schur_chunk; return MPCG_OK; }
    if (!strcmp(key, "last_schur_chunk")) { *value = h->last_schur_chunk; return MPCG_OK; }
    if (!strcmp(key, "dz_dpp")) { *value = h->dz_dpp; return MPCG_OK; }
    if (!strcmp(key, "kkt_analytic")) { *value = h->kkt_analytic; return MPCG_OK; }
    if (!strcmp(key, "kkt_f32")) { *value = h->kkt_f32; return MPCG_OK; }
    if (!strcmp(key, "sched_hint")) { *value = h->sched_hint; return MPCG_OK; }
    if (!strcmp(key, "spmv_blocks_per_cu")) { *value = h->spmv_blocks_per_cu; return MPCG_OK; }
    if (!strcmp(key, "spmv_mfma")) { *value = h->spmv_mfma; return MPCG_OK; }
    if (!strcmp(key, "num_cus")) { *value = h->num_cus; return MPCG_OK; }
    if (!strcmp(key, "auto_cfg")) { *value = h->auto_cfg ? 1 : 0; return MPCG_OK; }
    if (!strcmp(key, "cluster_fixups")) {
        // trajectories re-solved by fix-up launches since mpcg_create (their cluster gave up after the bounded spin): a blocking
        // 8-byte D2H copy — it waits for the device work queued before it
        unsigned long long v = 0;
        if (hipSetDevice(h->device) != hipSuccess || hipMemcpy(&v, fixup_counter(h), sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return MPCG_ERR_HIP;
        *value = v > 0x7fffffffull ? 0x7fffffff : (int)v;
        return MPCG_OK;
    }
    // what the last solve on this handle launched
    if (!strcmp(key, "last_kernel_family")) { *value = h->last.family; return MPCG_OK; }      // 0 single-workgroup row-pair, 3 generic, 5 row-per-lane, 6 lane-pair, 7 clustered lane-pair, 8 clustered row-per-lane (double)
    if (!strcmp(key, "last_kernel_waves")) { *value = h->last.waves; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_reg_rows")) { *value = h->last.reg_rows; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_lds_rows")) { *value = h->last.lds_rows; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_lds_extra")) { *value = h->last.lds_extra; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_stream_bufs")) { *value = h->last.stream_bufs; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_cluster")) { *value = h->last.cluster; return MPCG_OK; }
    if (!strcmp(key, "last_kernel_lds_bytes")) { *value = h->last.lds_bytes; return MPCG_OK; }
    return MPCG_ERR_INVALID;
}

}  // extern "C"

// ---- launch helpers -----------------------------------------------------------------------------
// Triples (3 block rows) per matrix per wave cached in LDS for this launch configuration.
static int lds_rows_for(const mpcg_handle* h, const PcgKnobs& k, int nw, int esz) {
    const size_t base = lds_bytes_for(h->N, nw);
    const int ntr = ((int)h->N + 2) / 3;
    const int TT = (ntr + nw - 1) / nw;                           // triples per matrix of wave 0
    const int rt = esz == 2 ? k.reg_rows16 : k.reg_rows;
    const int want_max = TT > rt ? TT - rt : 0;
    int lt = esz == 2 ? k.lds_rows16 : k.lds_rows;
    if (lt < 0) {
        if (rt <= 0) return 0;
        const size_t per_pair = pcg_lds_cache_floats(nw, 1, esz) * sizeof(float);
        lt = base < kLdsMax ? (int)((kLdsMax - base) / per_pair) : 0;
    }
    if (lt > want_max) lt = want_max;
    return lt;
}

// LDS bytes requested at launch: vectors + matrix cache, raised to floor(160 KiB / k) when the handle
// limits residency to k workgroups per CU.
static size_t lds_request(const mpcg_handle* h, const PcgKnobs& k, int nw, int esz) {
    size_t need = lds_bytes_for(h->N, nw) + pcg_lds_cache_floats(nw, lds_rows_for(h, k, nw, esz), esz) * sizeof(float);
    if (k.max_wg_per_cu > 0) {
        size_t pad = (kLdsMax / (size_t)k.max_wg_per_cu) & ~(size_t)15;
        if (pad > need) need = pad;
    }
    return need;
}

// <.,.,1> kernels: single-triple LDS slots that still fit after the uniform cache, at most two per wave that streams
static int lds_extra_for(const mpcg_handle* h, const PcgKnobs& k, int nw, int sb, int esz, int* streaming_waves_out) {
    *streaming_waves_out = 0;
    if (sb != 1 || k.lds_extra == 0) return 0;
    const int rt = esz == 2 ? k.reg_rows16 : k.reg_rows;
    const int lt = lds_rows_for(h, k, nw, esz);
    const int ntr = ((int)h->N + 2) / 3;
    int streaming_waves = 0;                                   // waves whose triples exceed registers + uniform cache
    for (int w = 0; w < nw; ++w) streaming_waves += (ntr - w + nw - 1) / nw > rt + lt;
    const size_t used = lds_bytes_for(h->N, nw) + pcg_lds_cache_floats(nw, lt, esz) * sizeof(float);
    const size_t slot = pcg_lds_cache_floats(1, 1, esz) * sizeof(float) / 2;      // one wave, one matrix, one triple
    int e = used < kLdsMax ? (int)((kLdsMax - used) / slot) : 0;
    // slot layout [S: nw][Pinv: nw]: a Pinv slot of wave w sits nw + w slots in, so Pinv slots only fit if ...
    if (e > streaming_waves) {
        const int room = e - nw;                    // slots available beyond the S bank
        e = streaming_waves + (room > 0 ? (room < streaming_waves ? room : streaming_waves) : 0);
    }
    if (k.lds_extra > 0 && e > k.lds_extra) e = k.lds_extra;
    *streaming_waves_out = streaming_waves;
    return e;
}

// LDS layout of one <NW,RT,SB> launch: (bytes, LT, extra S slots, extra Pinv slots)
struct TrajLds { size_t bytes; int lt, extra_s, extra_p; };
static TrajLds traj_lds(const mpcg_handle* h, const PcgKnobs& k, int nw, int sb, int esz) {
    TrajLds t;
    t.lt = lds_rows_for(h, k, nw, esz);
    int streaming_waves = 0;
    const int extra = lds_extra_for(h, k, nw, sb, esz, &streaming_waves);
    t.extra_s = extra < streaming_waves ? extra : streaming_waves;             // S first: one pass loses its longest stream
    t.extra_p = extra - t.extra_s;
    // (the kernel lays the extra slots out as [S: NW][Pinv: NW]; only the first extra_s / NW + extra_p are touched)
    const int extra_span = t.extra_p > 0 ? nw + t.extra_p : t.extra_s;
    t.bytes = lds_bytes_for(h->N, nw) + pcg_lds_cache_floats(nw, t.lt, esz) * sizeof(float)
            + (size_t)extra_span * (pcg_lds_cache_floats(1, 1, esz) * sizeof(float) / 2);
    const size_t padded = lds_request(h, k, nw, esz);
    if (t.bytes <= kLdsMax && padded > t.bytes && padded <= kLdsMax) t.bytes = padded;
    return t;
}

template <int NW, int RT, int SB, typename MT>
static int launch_pcg_t(mpcg_handle* h, const PcgKnobs& k, PcgArgs a, uint32_t batch, hipStream_t st, bool record) {
    const TrajLds t = traj_lds(h, k, NW, SB, (int)sizeof(MT));
    a.lds_rows = t.lt; a.lds_extra_s = t.extra_s; a.lds_extra_p = t.extra_p;
    if (t.bytes > kLdsMax) return fail(h, MPCG_ERR_INVALID, "pcg_lds_rows does not fit 160 KiB of LDS");
    auto kern = pcg_traj_kernel<NW, RT, SB, MT>;
    if (t.bytes > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.bytes));
    hipLaunchKernelGGL(kern, dim3(batch), dim3(NW * 64), t.bytes, st, a);
    HIP_TRY(h, hipGetLastError());
    if (record) h->last = LastKernel{FAM_TRAJ, NW, RT, t.lt, SB, 0, (int)t.bytes, t.extra_s + t.extra_p};
    return MPCG_OK;
}

template <int NW, int RT, int SB, typename MT>
static int occupancy_t(mpcg_handle* h, const PcgKnobs& k, int* blocks_per_cu) {
    const TrajLds t = traj_lds(h, k, NW, SB, (int)sizeof(MT));
    if (t.bytes > kLdsMax) return fail(h, MPCG_ERR_INVALID, "pcg_lds_rows does not fit 160 KiB of LDS");
    auto kern = pcg_traj_kernel<NW, RT, SB, MT>;
    if (t.bytes > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.bytes));
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kern, NW * 64, t.bytes));
    return MPCG_OK;
}

// Compiled (waves, register triples, stream buffers) variants.  X(NW, RT, SB)
#define MPCG_PCG_VARIANTS(X)                                                              \
    X(16, 0, 2) X(8, 0, 2) X(4, 0, 2)                                                     \
    X(16, 1, 0) X(16, 1, 1) X(16, 2, 1)                                                   \
    X(8, 2, 2) X(8, 2, 1) X(8, 2, 0) X(8, 3, 1) X(8, 3, 0) X(8, 4, 0)                                \
    X(4, 3, 0) X(4, 4, 2) X(4, 6, 1) X(4, 7, 1) X(4, 7, 0)
// fp16 matrix storage: a triple costs 14 registers instead of 28
#define MPCG_PCG_VARIANTS16(X)                                                            \
    X(16, 0, 2) X(8, 0, 2)                                                                \
    X(8, 4, 1) X(8, 5, 1) X(8, 6, 0)                                                      \
    X(4, 10, 1) X(4, 12, 1) X(4, 12, 0)

// stream buffers actually needed: 0 when every triple of every wave is resident
static int stream_bufs_for(const mpcg_handle* h, const PcgKnobs& k, int nw, int esz) {
    const int ntr = ((int)h->N + 2) / 3;
    const int TT = (ntr + nw - 1) / nw;
    const int rt = esz == 2 ? k.reg_rows16 : k.reg_rows;
    const bool all_resident = TT <= rt + lds_rows_for(h, k, nw, esz);
    if (k.stream_bufs >= 0) return (k.stream_bufs == 0 && !all_resident) ? 1 : k.stream_bufs;
    return all_resident ? 0 : -1;       // -1: any compiled SB > 0 (1 preferred)
}

// Automatic per-call configuration of the single-workgroup kernels (round-1 sweeps on MI355X: profiles/r01_tune11.txt,
// r01_tune12.txt, r01e_tune_nsweep.txt).  Writes into the caller's COPY of the knobs.
static void choose_auto(const mpcg_handle* h, PcgKnobs& k, uint32_t batch, int esz) {
    const uint32_t N = h->N;
    if (esz == 2) {     // fp16 storage: <= 48 triples all in registers
        if (N <= 144) { k.waves16 = 8; k.reg_rows16 = 6; } else { k.waves16 = 4; k.reg_rows16 = 12; }
        k.lds_rows16 = -1;
        return;
    }
    if (N <= 48) {
        // short horizons (<= 16 triples): with more trajectories than CUs, 4 waves x 3 register triples (+ 1 in LDS
        // beyond N=36) need < 256 registers, so TWO trajectories share a CU and fill each other's barrier and
        // reduction latencies (batch 2048: N=32 269 vs 171 M it/s, N=48 187 vs 166); up to one trajectory per CU
        // the 8-wave kernel with everything in registers is the faster solve
        if (batch > (uint32_t)h->num_cus) { k.waves = 4; k.reg_rows = 3; k.lds_rows = -1; }
        else { k.waves = 8; k.reg_rows = 2; k.lds_rows = 0; }
    } else if (N <= 96) {   // <= 32 triples: three in registers + one in LDS per wave and matrix
        k.waves = 8; k.reg_rows = 3; k.lds_rows = -1;
    } else {
        // long horizons: 8 waves (two per SIMD) with 3 register triples + 1 LDS triple per wave and matrix, the rest
        // streamed, beat 4 fat waves with 7 + 2 since <8,3,1> runs spill-free (N=128, batch 1024: 4.06 ms vs 5.23 ms,
        // profiles/r01e_phases_N128.txt); beyond N=256 (reached only with the cluster kernel switched off) the fat
        // waves' larger resident share wins again (profiles/r01_tune12.txt)
        if (N < 256 || batch < (uint32_t)h->num_cus) { k.waves = 8; k.reg_rows = 3; }
        else { k.waves = 4; k.reg_rows = 7; }
        k.lds_rows = -1;
    }
}

// ---- lane-pair-per-knot kernel (pcg_lpk.hip.h): everything register-resident, N <= 128 ----
template <int NWR>
static int launch_lpk_t(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    const size_t lds = pcg_lpk_lds_floats(LpkLds<NWR>::NW) * sizeof(float);
    auto kern = pcg_lpk_kernel<NWR>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(batch), dim3(LpkLds<NWR>::NW * 64), lds, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{FAM_LPK, LpkLds<NWR>::NW, 0, 0, 0, 0, (int)lds, 0};
    return MPCG_OK;
}
// ---- lane-quad-per-knot kernel, both matrices in every wavefront (pcg_lqb.hip.h): the lane-pair kernel's contract, two working wavefronts per SIMD ----
template <int NMAXQ>
static int launch_lqb_t(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    const size_t lds = pcg_lqb_lds_floats(NMAXQ) * sizeof(float);
    void (*kern)(PcgArgs) = a.pcols == 3 ? pcg_lqb_kernel<NMAXQ, true> : pcg_lqb_kernel<NMAXQ, false>;      // (SS / block-Jacobi builds)
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(batch), dim3(NMAXQ * 4), lds, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{FAM_LQB, NMAXQ / 16, 0, 0, 0, 0, (int)lds, 0};
    return MPCG_OK;
}
// The float lane-quad kernel takes the lane-pair kernel's launches (same contract: lower block triangle, gated / fix-up launches, dispatch order).
// Automatic use ("pcg_lqb" = 1 / 0 forces / forbids, an explicit "pcg_lpk" = 1 keeps the lane-pair kernel; steady clocks, tools/_prof/lqb_ab.py,
// lqb_n32.py, lqb_n32_forced.py; the first sweep: profiles/r06_lqb_policy.txt): WHEREVER the lane-pair kernel ran in fp32, both preconditioners
// (block-Jacobi is a build of its own, pcg_lqb_kernel<., false>: no off-diagonal Pinv blocks, no z) —
//   * 32 < knot_points <= 64, every batch: two independent workgroups per CU, all of their wavefronts working: one trajectory of 64 knots 1.03 against
//     1.44 us per iteration (block-Jacobi 0.94 / 1.22), batch 1024 3.02 / 3.71 (2.45 / 3.09); also ahead of the row-per-lane kernel's latency-sized calls;
//   * 64 < knot_points <= 128: 1.58 / 1.62 us per iteration of one trajectory, 6.31 / 6.54 per 1024 (block-Jacobi 1.32 / 1.34, 5.29 / 5.51);
//   * 16 < knot_points <= 32, throughput-sized calls (lpk_half): equal to the lane-pair kernel's half build at 2.5-4 trajectories per CU (+-1 %),
//     1.10x (SS) / 1.13-1.19x (block-Jacobi) beyond.
static bool lqb_auto(const mpcg_handle* h, int esz) { return esz == 4 && h->lqb == -1 && h->lpk == -1 && h->rpl != 1 && h->auto_cfg && h->cluster <= 0; }
static bool use_lqb(const mpcg_handle* h, int esz) {
    if (esz != 4 || h->lqb == 0) return false;
    return h->lqb == 1 || lqb_auto(h, esz);
}
static int launch_lpk(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    if (use_lqb(h, a.esz)) {
        if (h->N <= 32) return launch_lqb_t<32>(h, a, batch, st);
        return h->N <= 64 ? launch_lqb_t<64>(h, a, batch, st) : launch_lqb_t<128>(h, a, batch, st);
    }
    if (h->N <= 32) return launch_lpk_t<0>(h, a, batch, st);          // the half build: one wavefront per matrix, four workgroups per CU
    return h->N <= 64 ? launch_lpk_t<1>(h, a, batch, st) : launch_lpk_t<2>(h, a, batch, st);
}
// Automatic use: 36 < N <= 128 (where the row-per-lane kernel has not taken the call).  Its per-lane work does not shrink with the horizon
// (a lane pair per knot whatever N), so up to N = 36 — where the row-pair kernel <4,3,0> fits two trajectories per CU — that one stays
// ahead in throughput (N=36: 292 vs 222 M it/s at batch 2048); beyond it the order flips (HISTORY.md §3.1c).
// ... and, round 5, 16 < N <= 32 for calls of at least 2.5 trajectories per CU: the HALF build (one wavefront per matrix, 32 knots, four workgroups
// per CU) takes 1.45-1.55 us per iteration for up to four trajectories per CU, the row-per-lane kernel 1.26 us for two, 2.1 for three, 2.5-2.7 for
// four: N = 32, batch 1024 / 4096: 378 / 434 -> 661 / 629 M it/s (tools/_prof/lpk_half.py); at N <= 16 the row-per-lane kernel stays ahead (780-920 M).
static bool lpk_half(const mpcg_handle* h, int esz, uint32_t batch) {
    return esz == 4 && h->lpk != 0 && h->rpl != 1 && h->auto_cfg && h->cluster <= 0 && h->N > 16 && h->N <= 32 && 2ull * batch >= 5ull * (uint32_t)h->num_cus;
}
static bool use_lpk(const mpcg_handle* h, int esz, uint32_t batch) {
    if ((esz != 4 && esz != 2) || h->N > kLpbMaxN || h->lpk == 0) return false;     // (fp16 storage: converted once at the load)
    // (with the lane-quad kernel behind launch_lpk: from 33 knots, at every batch — it beats the row-pair kernel at 33..36 and the row-per-lane kernel's
    //  latency-sized calls up to 64, use_lqb)
    return h->lpk == 1 || (h->auto_cfg && h->cluster <= 0 && h->N > (lqb_auto(h, esz) ? 32u : 36u)) || lpk_half(h, esz, batch);
}

// ---- row-per-lane kernel (pcg_rpl.hip.h): short horizons, NW wavefronts x RHO slots of four knots ----
#define MPCG_RPL_VARIANTS(X) X(4, 1) X(4, 2) X(8, 1) X(8, 2) X(16, 1)
template <int NW, int RHO, bool PC3>
static int launch_rpl_t(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    const size_t lds = pcg_rpl_lds_floats((int)h->N, NW) * sizeof(float);
    hipLaunchKernelGGL((pcg_rpl_kernel<NW, RHO, PC3>), dim3(batch), dim3(NW * 64), lds, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{FAM_RPL, NW, RHO, 0, 0, 0, (int)lds, 0};
    return MPCG_OK;
}
// (NW, RHO) of a call: NW RHO slots of four knots must cover the horizon.  Automatic (profiles/r02_rpl_quick.txt): N <= 16 four
// wavefronts; N <= 32 eight wavefronts x one slot for a latency-sized call (at most one trajectory per CU), four x two slots for
// throughput (N=32 SS 395 vs 376 M it/s, block-Jacobi 623 vs 484 M); N <= 64 eight x two.
static void rpl_shape(const mpcg_handle* h, uint32_t batch, int* nw, int* rho) {
    const int slots = ((int)h->N + 3) / 4;
    int w = h->rpl_waves;
    if (w == 0) w = slots <= 4 ? 4 : slots <= 8 ? (batch > (uint32_t)h->num_cus ? 4 : 8) : 8;
    *nw = w;
    *rho = (slots + w - 1) / w;
}
static int launch_rpl(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    int nw, rho;
    rpl_shape(h, batch, &nw, &rho);
#define X(NW_, RHO_)                                                                                   \
    if (nw == NW_ && rho == RHO_)                                                                      \
        return a.pcols == 3 ? launch_rpl_t<NW_, RHO_, true>(h, a, batch, st) : launch_rpl_t<NW_, RHO_, false>(h, a, batch, st);
    MPCG_RPL_VARIANTS(X)
#undef X
    return fail(h, MPCG_ERR_UNSUPPORTED, "row-per-lane kernel: no compiled (waves, slots) variant covers this horizon with rpl_waves");
}
// Automatic use (no explicit pcg_* knob): N <= 32 always (N=32: 0.150 vs 0.247 ms for one trajectory, 395 vs 277 M it/s at batch 2048;
// N=16: 875 vs 317 M); 32 < N <= 64 for latency-sized calls only (N=64 one trajectory 0.248 vs 0.341 ms, but 170 vs 211 M it/s at
// batch 2048, where the lane-pair / row-pair kernels stay ahead).
static bool use_rpl(const mpcg_handle* h, int esz, uint32_t batch) {
    if (esz != 4 || h->N > kRplMaxN || h->rpl == 0) return false;
    if (h->rpl == 1) return true;
    if (h->lpk == 1) return false;
    if (lpk_half(h, esz, batch)) return false;
    return h->auto_cfg && h->cluster <= 0 && (h->N <= 32 || (batch <= (uint32_t)h->num_cus && !lqb_auto(h, esz)));
}

static int launch_traj(mpcg_handle* h, const PcgKnobs& k, const PcgArgs& a, uint32_t batch, hipStream_t st, int esz, bool record);

// ---- clustered lane-pair kernel (pcg_lpk_cluster.hip.h): G members x up to 128 knots, all blocks in registers ----
// Members must be co-resident, which a plain launch cannot guarantee when another stream holds CUs: a member that waits longer than the
// bounded spin gives up; every member that FINISHES a trajectory counts itself in that trajectory's flag, and the launch is followed by a
// launch of a single-workgroup kernel in which only the trajectories whose count is not G run — the caller always gets a solved system.
static int lpkc_members(const mpcg_handle* h) {
    const int nmax = 128;
    const int G = h->cluster > 0 ? h->cluster : ((int)h->N + nmax - 1) / nmax;
    if (G < 2 || G > LPBC_MAX_G || G > h->num_cus || G > (int)h->N) return 0;
    if (((int)h->N + G - 1) / G > nmax) return 0;        // the largest member: ceil(N / G) knots
    return G;
}
// clusters the chip holds: the kernel pins cluster cl to XCD cl % 8 (all its members on one XCD, 32 CUs each), so residency is a
// per-XCD count — 8 x floor(CUs of one XCD / G).  (Sized chip-wide, G = 3, 5, 6, 7 over-subscribed some XCDs by a member that could
// not start while the persistent clusters held the CUs: its peers spun to the limit and the trajectories fell to the fix-up launch.)
static uint32_t lpkc_resident_clusters(const mpcg_handle* h, int G) {
    const int xcd_slots = h->num_cus / 8;
    return h->num_cus >= 8 && xcd_slots >= G ? (uint32_t)(8 * (xcd_slots / G)) : (uint32_t)(h->num_cus / G);
}
// returns 1 when it does not apply.  One persistent launch for any batch: clusters draw trajectories from a queue.
// scratch: [queue: one 128-byte line][flags: one line per trajectory of the call][cells: 1 KB per member of the launch] — what a call uses
// is contiguous, so one small fill precedes every launch
static int try_launch_cluster(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st, int esz, bool guarded = false) {
    if (h->cluster == 0 || (esz != 4 && esz != 2)) return 1;
    if (h->cluster < 0 && (!h->auto_cfg || h->N <= kLpbMaxN)) return 1;     // explicit pcg_* knobs, or a horizon one CU holds
    constexpr int NWR = 2;
    const int G = lpkc_members(h);
    if (G == 0) return 1;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "batch exceeds max_batch");
    const uint32_t resident = lpkc_resident_clusters(h, G);
    const uint32_t clusters = batch < resident ? batch : resident;
    const size_t lds = pcg_lpkc_lds_floats(4 * NWR) * sizeof(float);
    void (*kern)(ClusterArgs) = pcg_lpkc_kernel<NWR>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PcgKnobs kf = h->k;
    choose_auto(h, kf, 1, esz);
    const bool fixup = h->cluster_fixup && (h->N <= kLpbMaxN || lds_bytes_for(h->N, kf.waves) <= kLdsMax);
    ClusterArgs ca;
    ca.kl_max = 64 * NWR;
    ca.p = a;
    ca.queue = h->cluster_scratch;
    ca.fail_flags = ca.queue + CL_FLAG_STRIDE;
    ca.scratch = ca.fail_flags + (size_t)batch * CL_FLAG_STRIDE;
    ca.G = G;
    ca.batch = (int)batch;
    ca.clusters = (int)clusters;
    ca.l2_handoff = h->cluster_l2;
    ca.test_fail = h->cluster_test_fail;
    // one launch in front: the queue counter, this call's flags and the cells of this launch zeroed (their tags restart at 1 every launch) and,
    // when a fix-up launch follows, lambda0 copied for it
    const size_t zw = CL_FLAG_STRIDE + (size_t)batch * CL_FLAG_STRIDE + (size_t)clusters * G * LPBC_WG_WORDS;
    if (fixup) {
        const int rc = launch_cluster_prologue(h, h->cluster_scratch, zw, a.lambda, (size_t)batch * h->N * NS * sizeof(float), st);
        if (rc != MPCG_OK) return rc;
    } else {
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, st, h->cluster_scratch, zw);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(kern, dim3(((clusters + 7) / 8) * 8 * (unsigned)G), dim3(NWR * 256), lds, st, ca);
    HIP_TRY(h, hipGetLastError());
    if (fixup) {
        PcgArgs c = a;
        c.redo_flags = ca.fail_flags;
        c.redo_stride = CL_FLAG_STRIDE;
        c.redo_skip = (unsigned)G;
        // a GUARDED call (launch_guarded: the handle does not know yet whether the caller's matrices are block-symmetric) relies on this fix-up
        // being a three-column kernel — every completion count is 0 after a gated exit — so it is never the lower-triangle lane-pair kernel
        // then (a forced "cluster" = G at N <= 128), and gated exits are not counted as abandoned trajectories
        c.redo_count = guarded ? nullptr : fixup_counter(h);
        c.lam0 = static_cast<const float*>(h->lam_backup);     // (members that finished a trajectory their cluster did not have written their knots of lambda)
        const int rc = h->N <= kLpbMaxN && !guarded ? launch_lpk(h, c, batch, st) : launch_traj(h, kf, c, batch, st, esz, /*record=*/false);
        if (rc != MPCG_OK) return rc;
    } else {
        const int rc = launch_cluster_report(h, ca.fail_flags, G, batch, a.iters, a.max_iter_exit, st);
        if (rc != MPCG_OK) return rc;
    }
    h->last = LastKernel{FAM_LPKC, 4 * NWR, 0, 0, 0, G, (int)lds, 0};
    return MPCG_OK;
}

static int launch_traj(mpcg_handle* h, const PcgKnobs& k, const PcgArgs& a, uint32_t batch, hipStream_t st, int esz, bool record) {
    const int waves = esz == 2 ? k.waves16 : k.waves;
    const int rt = esz == 2 ? k.reg_rows16 : k.reg_rows;
    const int sb = stream_bufs_for(h, k, waves, esz);
    for (int want : {sb, sb == 0 ? 1 : -2, sb <= 0 ? 2 : -2}) {     // exact, then fall back to a streaming build
        if (want == -2) continue;
        if (esz == 4) {
#define X(NW_, RT_, SB_)                                                                       \
            if (waves == NW_ && rt == RT_ && (want == SB_ || (want == -1 && SB_ > 0)))             \
                return launch_pcg_t<NW_, RT_, SB_, float>(h, k, a, batch, st, record);
            MPCG_PCG_VARIANTS(X)
#undef X
        } else {
#define X(NW_, RT_, SB_)                                                                       \
            if (waves == NW_ && rt == RT_ && (want == SB_ || (want == -1 && SB_ > 0)))             \
                return launch_pcg_t<NW_, RT_, SB_, _Float16>(h, k, a, batch, st, record);
            MPCG_PCG_VARIANTS16(X)
#undef X
        }
    }
    return fail(h, MPCG_ERR_UNSUPPORTED, "no compiled kernel variant for this (pcg_waves, pcg_reg_rows, pcg_stream_bufs)");
}

// Kernel selection of one solve:
//   1. row-per-lane kernel: fp32, N <= 32 (N <= 64 for latency-sized calls), automatic configuration (or "pcg_rpl" = 1);
//   2. lane-pair kernel: fp32, 36 < N <= 128, automatic configuration (or "pcg_lpk" = 1);
//   3. clustered lane-pair kernel: forced ("cluster" = G), or automatic configuration and a horizon one CU cannot hold;
//   4. single-workgroup kernel <waves, reg_rows, stream_bufs> — the handle's knobs, adjusted per call by the
//      automatic policy unless the caller set any pcg_* knob.
static int launch_generic_f32(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st);

// Tolerance of the block-symmetry checks: block pairs (k, right) / (k+1, left) that differ by more than 1 % of their largest entry.  The
// reference's construction gives 0 for S; its symmetric-stair Pinv[k,right] / Pinv[k+1,left] are the same triple product associated two ways
// in float — measured on the bench's systems: 2.5e-5 of the largest entry in the median, 6e-5 at worst, more on worse-conditioned blocks —
// so the check looks for STRUCTURAL asymmetry (a caller-made Pinv), not for rounding.
static constexpr float kSymRelTol = 1e-2f;
// (double: the two associations agree to ~1e-13; anything at the 1e-6 level is a different matrix, and a lower-triangle kernel would solve a
//  different system than the caller's without a word — ADVICE r05)
static constexpr double kSymRelTol64 = 1e-6;

// block pairs of S and (SS only) Pinv that fail the check.  Blocking: waits for `st`.
// S (and Pinv when it has off-diagonal blocks) of one call through bd_symmetry_check_kernel, in the call's storage type
static void launch_symmetry_check(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st, unsigned blocks, unsigned long long* cnt, unsigned long long* flag) {
    for (const void* m : {a.S, a.pcols == 3 ? a.Pinv : nullptr}) {
        if (!m) continue;
        if (a.esz == 2)
            hipLaunchKernelGGL(bd_symmetry_check_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, static_cast<const _Float16*>(m), (int)h->N, (int)batch, kSymRelTol, cnt, flag);
        else
            hipLaunchKernelGGL(bd_symmetry_check_kernel<float>, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(m), (int)h->N, (int)batch, kSymRelTol, cnt, flag);
    }
}
static int symmetry_violations(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st, int* out) {
    unsigned long long* cnt = fixup_counter(h) + 8;
    HIP_TRY(h, hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
    const long items = (long)batch * ((long)h->N - 1);
    const unsigned blocks = (unsigned)((items + 3) / 4);
    launch_symmetry_check(h, a, batch, st, blocks, cnt, nullptr);
    HIP_TRY(h, hipGetLastError());
    unsigned long long v = 0;
    HIP_TRY(h, hipMemcpyAsync(&v, cnt, sizeof v, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    *out = v > 0x7fffffffull ? 0x7fffffff : (int)v;
    return MPCG_OK;
}

// A lower-triangle launch of a handle that does not know yet whether its caller's matrices are block-symmetric (mpcg.h, BLOCK SYMMETRY):
//   check kernel(s)  -> OR into the handle's device flag (never reset: the latch)
//   the lower-triangle kernel, gated: its workgroups leave at once if the flag is set
//   a three-column kernel, gated the other way: runs only if the flag is set
//   (clustered kernel: its own fix-up launch is that kernel — every completion count is 0 after a gated exit)
// and, outside graph capture, an asynchronous copy of the flag to pinned host memory + an event that sym_poll() queries on later calls.
static int launch_guarded(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    unsigned long long* flag = fixup_counter(h) + 9;
    const long items = (long)batch * ((long)h->N - 1);
    const unsigned blocks = (unsigned)((items + 3) / 4);
    ++h->sym_guard_seq;
    if (a.pcols == 3) h->sym_pinv_guarded = true;
    launch_symmetry_check(h, a, batch, st, blocks, nullptr, flag);
    HIP_TRY(h, hipGetLastError());
    PcgArgs p = a;
    p.redo_flags = flag; p.redo_stride = 0; p.redo_skip = 1; p.redo_count = nullptr;         // skip if the flag is 1
    bool need_fallback = true;
    int rc;
    const int esz = a.esz;
    if (use_lpk(h, esz, batch)) rc = launch_lpk(h, p, batch, st);
    else {
        rc = try_launch_cluster(h, p, batch, st, esz, /*guarded=*/true);
        if (rc == 1) return 1;                                                                // (does not apply: the caller falls through)
        need_fallback = !h->cluster_fixup;
    }
    if (rc != MPCG_OK) return rc;
    const LastKernel primary = h->last;
    if (need_fallback) {
        PcgArgs f = a;
        f.redo_flags = flag; f.redo_stride = 0; f.redo_skip = 0; f.redo_count = nullptr;     // skip if the flag is 0
        if (esz == 4 && h->N <= kRplMaxN && h->rpl != 0) rc = launch_rpl(h, f, batch, st);   // full block rows in registers
        else {
            PcgKnobs k = h->k;
            choose_auto(h, k, batch, esz);
            rc = launch_traj(h, k, f, batch, st, esz, /*record=*/false);
        }
        if (rc != MPCG_OK) return rc;
        h->last = primary;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!h->sym_pending && hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
        HIP_TRY(h, hipMemcpyAsync(h->sym_host, flag, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipEventRecord(h->sym_event, st));
        h->sym_pending = true;
        h->sym_armed_seq = h->sym_guard_seq;
        h->sym_armed_pinv = h->sym_pinv_guarded;
    }
    return MPCG_OK;
}

static int launch_pcg(mpcg_handle* h, const PcgArgs& a_in, uint32_t batch, hipStream_t st, int esz) {
    HIP_TRY(h, hipSetDevice(h->device));
    PcgArgs a = a_in;
    a.esz = esz;
    if (h->generic) {
        if (esz != 4) return fail(h, MPCG_ERR_UNSUPPORTED, "fp16 matrix storage exists for state_size = 14 only");
        return launch_generic_f32(h, a, batch, st);
    }
    if (use_rpl(h, esz, batch)) return launch_rpl(h, a, batch, st);      // (an explicit "pcg_lpk" = 1 wins over the automatic choice of this one)
    // "check_symmetry" (debug, off by default): the kernels below read only the left + diagonal block columns (mpcg.h "Block symmetry").
    // Verify the precondition on this call's matrices; a call that violates it is solved by a kernel that reads all three columns.
    bool lower_ok = true;
    h->last_sym_violations = 0;
    if (h->check_symmetry && (use_lpk(h, esz, batch) || (h->cluster != 0 && (h->cluster > 0 || h->N > kLpbMaxN)))) {
        int v = 0;
        const int rc = symmetry_violations(h, a, batch, st, &v);
        if (rc != MPCG_OK) return rc;
        h->last_sym_violations = v;
        lower_ok = v == 0;
    } else if (a.redo_flags == nullptr && (use_lpk(h, esz, batch) || (h->cluster != 0 && (h->cluster > 0 || (h->auto_cfg && h->N > kLpbMaxN))))) {
        // the symmetry latch (see launch_guarded): no synchronisation, no per-solve D2H
        sym_poll(h, st, true);
        { const int rc = sym_prepare(h, a.pcols, st); if (rc != MPCG_OK) return rc; }
        if (h->sym_state == 2) lower_ok = false;
        else if (h->sym_state == 0) {
            const int rc = launch_guarded(h, a, batch, st);
            if (rc != 1) return rc;
        }
    }
    if (lower_ok) {
        if (use_lpk(h, esz, batch)) return launch_lpk(h, a, batch, st);
        const int rc = try_launch_cluster(h, a, batch, st, esz);
        if (rc != 1) return rc;
    } else if (esz == 4 && h->N <= kRplMaxN && h->rpl != 0) {
        return launch_rpl(h, a, batch, st);              // full block rows in registers
    }
    PcgKnobs k = h->k;
    if (h->auto_cfg) choose_auto(h, k, batch, esz);
    return launch_traj(h, k, a, batch, st, esz, /*record=*/true);
}

static int occupancy(mpcg_handle* h, const PcgKnobs& k, int* per_cu) {
    const int sb = stream_bufs_for(h, k, k.waves, 4);
    for (int want : {sb, sb == 0 ? 1 : -2, sb <= 0 ? 2 : -2}) {
        if (want == -2) continue;
#define X(NW_, RT_, SB_) \
        if (k.waves == NW_ && k.reg_rows == RT_ && (want == SB_ || (want == -1 && SB_ > 0))) \
            return occupancy_t<NW_, RT_, SB_, float>(h, k, per_cu);
        MPCG_PCG_VARIANTS(X)
#undef X
    }
    return fail(h, MPCG_ERR_UNSUPPORTED, "no compiled kernel variant for this (pcg_waves, pcg_reg_rows, pcg_stream_bufs)");
}

// LDS bytes of the launch a default-configured batch-1 solve makes (what pcgSharedMemSize stands for)
static size_t default_launch_lds_bytes(uint32_t N, int num_cus) {
    if (N <= 32) return pcg_rpl_lds_floats((int)N, N <= 16 ? 4 : 8) * sizeof(float);           // row-per-lane kernel (a batch-1 call)
    if (N <= kLpbMaxN) return pcg_lqb_lds_floats(N <= 64 ? 64 : 128) * sizeof(float);  // lane-quad kernel
    mpcg_handle tmp;
    tmp.N = N; tmp.n = NS; tmp.num_cus = num_cus;
    if (lpkc_members(&tmp) > 0) return pcg_lpkc_lds_floats(8) * sizeof(float);          // clustered lane-pair kernel
    choose_auto(&tmp, tmp.k, 1, 4);                                                     // (beyond 8 x 128 knots: the single-workgroup streaming kernel)
    const int sb = stream_bufs_for(&tmp, tmp.k, tmp.k.waves, 4);
    return traj_lds(&tmp, tmp.k, tmp.k.waves, sb == 0 ? 0 : 1, 4).bytes;
}

template <typename T, int NFIX>
static int launch_generic(mpcg_handle* h, PcgArgsG<T> a, uint32_t batch, void* stream) {
    HIP_TRY(h, hipSetDevice(h->device));
    a.n = (int)h->n;
    const size_t lds = pcg_generic_lds_elems((int)h->N, (int)h->n) * sizeof(T);
    if (lds > kLdsMax) return fail(h, MPCG_ERR_UNSUPPORTED, "generic / double-precision kernel: the iterate vectors do not fit 160 KiB of LDS");
    const int nthr = pcg_generic_threads((int)h->N, (int)h->n);
    void (*kern)(PcgArgsG<T>) = nthr == F64_THREADS_WIDE ? pcg_generic_kernel<T, NFIX, F64_THREADS_WIDE> : pcg_generic_kernel<T, NFIX, F64_THREADS>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(batch), dim3(nthr), lds, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{3, nthr / 64, 0, 0, 2, 0, (int)lds, 0};      // family 3 = generic streaming kernel
    return MPCG_OK;
}
// double precision, N <= 32 (the real-time horizon): the row-per-lane kernel in double, one slot per wavefront
static constexpr uint32_t kRplMaxN64 = 32;
template <int NW, bool PC3>
static int launch_rpl_f64_t(mpcg_handle* h, const PcgArgs64& a, uint32_t batch, hipStream_t st) {
    const size_t lds = pcg_rpl_lds_floats((int)h->N, NW) * sizeof(double);
    hipLaunchKernelGGL((pcg_rpl_kernel_f64<NW, 1, PC3>), dim3(batch), dim3(NW * 64), lds, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{FAM_RPL, NW, 1, 0, 0, 0, (int)lds, 0};
    return MPCG_OK;
}
// ---- linsys_t = double, 32 < N <= 256: the row-per-lane kernel across G = ceil(N / 32) CUs of one XCD (pcg_rpl_cluster_f64.hip.h) ----
// Same launch shape as the clustered lane-pair kernel (persistent clusters pinned to XCDs, one fill of queue + flags + cells in front, a fix-up
// launch behind: here the streaming kernel, gated on the completion counts).  Returns 1 when it does not apply.
static int rplc64_members(const mpcg_handle* h) {
    const int G = h->cluster > 0 ? h->cluster : ((int)h->N + RPLC_KMAX - 1) / RPLC_KMAX;
    if (G < 2 || G > RPLC_MAX_G || G > h->num_cus || G > (int)h->N) return 0;
    if (((int)h->N + G - 1) / G > RPLC_KMAX) return 0;
    return G;
}
static size_t cluster64_alloc_words(const mpcg_handle* h) {
    return CL_FLAG_STRIDE + (size_t)h->max_batch * CL_FLAG_STRIDE + (size_t)h->num_cus * RPLC_WG_WORDS + 16;
}
template <typename T, int NFIX>
static int launch_generic(mpcg_handle* h, PcgArgsG<T> a, uint32_t batch, void* stream);
static int try_launch_cluster_f64(mpcg_handle* h, const PcgArgs64& a, uint32_t batch, hipStream_t st) {
    if (h->cluster == 0 || h->generic) return 1;
    const int G = rplc64_members(h);
    if (G == 0) return 1;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->cluster64_scratch || (h->cluster_fixup && h->lam_backup_bytes < (size_t)h->max_batch * h->N * NS * sizeof(double))) {
        // (mpcg_create made them for handles with <= 8 MB of double iterates; "reserve_f64" = 1 makes them ahead of a capture)
        { const int rc = alloc_allowed(h, st, "mpcg_pcg_solve_f64 (or set \"reserve_f64\" = 1 first)"); if (rc != MPCG_OK) return rc; }
        { const int rc = reserve_f64(h); if (rc != MPCG_OK) return rc; }
    }
    const uint32_t resident = lpkc_resident_clusters(h, G);
    const uint32_t clusters = batch < resident ? batch : resident;
    const size_t lds = pcg_rplc_lds_doubles() * sizeof(double);
    void (*kern)(ClusterArgs64) = a.pcols == 3 ? pcg_rplc_f64_kernel<true> : pcg_rplc_f64_kernel<false>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ClusterArgs64 ca;
    ca.p = a;
    ca.queue = h->cluster64_scratch;
    ca.fail_flags = ca.queue + CL_FLAG_STRIDE;
    ca.scratch = ca.fail_flags + (size_t)batch * CL_FLAG_STRIDE;
    ca.G = G; ca.batch = (int)batch; ca.clusters = (int)clusters; ca.l2_handoff = h->cluster_l2;
    ca.test_fail = h->cluster_test_fail;
    const size_t zw = CL_FLAG_STRIDE + (size_t)batch * CL_FLAG_STRIDE + (size_t)clusters * G * RPLC_WG_WORDS;
    if (h->cluster_fixup) {
        const int rc = launch_cluster_prologue(h, h->cluster64_scratch, zw, a.lambda, (size_t)batch * h->N * NS * sizeof(double), st);
        if (rc != MPCG_OK) return rc;
    } else {
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, st, h->cluster64_scratch, zw);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(kern, dim3(((clusters + 7) / 8) * 8 * (unsigned)G), dim3(RPLC_NW * 64), lds, st, ca);
    HIP_TRY(h, hipGetLastError());
    if (h->cluster_fixup) {                             // trajectories whose cluster gave up: the streaming kernel, all three block columns
        PcgArgs64 c = a;
        c.lower = 0;
        c.redo_flags = ca.fail_flags; c.redo_stride = CL_FLAG_STRIDE; c.redo_skip = (unsigned long long)G;
        c.redo_count = fixup_counter(h);
        c.lam0 = static_cast<const double*>(h->lam_backup);
        const int rc = launch_generic<double, 14>(h, c, batch, st);
        if (rc != MPCG_OK) return rc;
    } else {
        const int rc = launch_cluster_report(h, ca.fail_flags, G, batch, a.iters, a.max_iter_exit, st);
        if (rc != MPCG_OK) return rc;
    }
    h->last = LastKernel{FAM_RPLC64, RPLC_NW, 0, 0, 0, G, (int)lds, 0};
    return MPCG_OK;
}

// ---- linsys_t = double, 64 < N <= 512, block-symmetric matrices: the lane-quad kernel across G = ceil(N / 64) CUs of one XCD (pcg_lqk_cluster_f64.hip.h).
// Launch shape, scratch and fix-up of try_launch_cluster_f64.  Returns 1 when it does not apply.
static int lqkc_members(const mpcg_handle* h) {
    const int nmax = 64;
    const int G = h->cluster > 0 ? h->cluster : ((int)h->N + nmax - 1) / nmax;
    if (G < 2 || G > LQKC_MAX_G || G > h->num_cus || G > (int)h->N) return 0;
    if (((int)h->N + G - 1) / G > nmax) return 0;
    return G;
}
static int try_launch_cluster_lqk_f64(mpcg_handle* h, const PcgArgs64& a, uint32_t batch, hipStream_t st) {
    if (h->cluster == 0 || h->generic || h->lqk == 0) return 1;
    const int G = lqkc_members(h);
    if (G == 0) return 1;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->cluster64_scratch || (h->cluster_fixup && h->lam_backup_bytes < (size_t)h->max_batch * h->N * NS * sizeof(double))) {
        // (mpcg_create made them for handles with <= 8 MB of double iterates; "reserve_f64" = 1 makes them ahead of a capture)
        { const int rc = alloc_allowed(h, st, "mpcg_pcg_solve_f64 (or set \"reserve_f64\" = 1 first)"); if (rc != MPCG_OK) return rc; }
        { const int rc = reserve_f64(h); if (rc != MPCG_OK) return rc; }
    }
    const uint32_t resident = lpkc_resident_clusters(h, G);
    const uint32_t clusters = batch < resident ? batch : resident;
    const size_t lds = pcg_lqkc_lds_doubles() * sizeof(double);
    void (*kern)(ClusterArgs64) = pcg_lqkc_f64_kernel<2>;
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ClusterArgs64 ca;
    ca.p = a;
    ca.queue = h->cluster64_scratch;
    ca.fail_flags = ca.queue + CL_FLAG_STRIDE;
    ca.scratch = ca.fail_flags + (size_t)batch * CL_FLAG_STRIDE;
    ca.G = G; ca.batch = (int)batch; ca.clusters = (int)clusters; ca.l2_handoff = h->cluster_l2;
    ca.test_fail = h->cluster_test_fail;
    const size_t zw = CL_FLAG_STRIDE + (size_t)batch * CL_FLAG_STRIDE + (size_t)clusters * G * LQKC_WG_WORDS;
    // (the fix-up is the streaming kernel: beyond the horizon whose iterate vectors fit its LDS — N = 350 in double — an abandoned trajectory is
    //  reported, d_iters = 0xFFFFFFFF / d_max_iter_exit = 2, as with "cluster_fixup" = 0)
    const bool fixup = h->cluster_fixup && pcg_generic_lds_elems((int)h->N, (int)h->n) * sizeof(double) <= kLdsMax;
    if (fixup) {
        const int rc = launch_cluster_prologue(h, h->cluster64_scratch, zw, a.lambda, (size_t)batch * h->N * NS * sizeof(double), st);
        if (rc != MPCG_OK) return rc;
    } else {
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((zw + 255) / 256)), dim3(256), 0, st, h->cluster64_scratch, zw);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(kern, dim3(((clusters + 7) / 8) * 8 * (unsigned)G), dim3(512), lds, st, ca);
    HIP_TRY(h, hipGetLastError());
    if (fixup) {                                        // trajectories whose cluster gave up: the streaming kernel (two block columns: the latch says symmetric)
        PcgArgs64 c = a;
        c.lower = 1;
        c.redo_flags = ca.fail_flags; c.redo_stride = CL_FLAG_STRIDE; c.redo_skip = (unsigned long long)G;
        c.redo_count = fixup_counter(h);
        c.lam0 = static_cast<const double*>(h->lam_backup);
        const int rc = launch_generic<double, 14>(h, c, batch, st);
        if (rc != MPCG_OK) return rc;
    } else {
        const int rc = launch_cluster_report(h, ca.fail_flags, G, batch, a.iters, a.max_iter_exit, st);
        if (rc != MPCG_OK) return rc;
    }
    h->last = LastKernel{FAM_LQKC64, 8, 0, 0, 0, G, (int)lds, 0};
    return MPCG_OK;
}

// The double kernels that read only the left + diagonal block columns use the float path's latch (mpcg.h, BLOCK SYMMETRY): a handle that does not
// know yet checks THIS call's matrices once, with one blocking 8-byte copy (never during capture: a capturing call on such a handle runs a
// kernel that reads all three columns).
static int f64_symmetry_latch(mpcg_handle* h, const PcgArgs64& a, uint32_t batch, hipStream_t st) {
    sym_poll(h, st, true);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (h->sym_state == 1 && a.pcols == 3 && !h->sym_pinv_ok) {
        // latched on block-Jacobi calls: this call's Pinv has never been looked at.  (A capturing call cannot check: it runs a three-column kernel.)
        h->sym_state = 0;
    }
    if (h->sym_state != 0 || h->sym_pending) return MPCG_OK;
    if (capturing) return MPCG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    { const int rc = sym_prepare(h, a.pcols, st); if (rc != MPCG_OK) return rc; }
    unsigned long long* flag = fixup_counter(h) + 9;
    const long items = (long)batch * ((long)h->N - 1);
    const unsigned blocks = (unsigned)((items + 3) / 4);
    for (const double* m : {a.S, a.pcols == 3 ? a.Pinv : (const double*)nullptr})
        if (m) hipLaunchKernelGGL(bd_symmetry_check_kernel<double>, dim3(blocks), dim3(256), 0, st, m, (int)h->N, (int)batch, (float)kSymRelTol64, (unsigned long long*)nullptr, flag);
    HIP_TRY(h, hipGetLastError());
    unsigned long long v = 0;
    HIP_TRY(h, hipMemcpyAsync(&v, flag, sizeof v, hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipStreamSynchronize(st));
    h->sym_state = v ? 2 : 1;
    h->sym_pinv_ok = !v && a.pcols == 3;
    if (v) h->err = "warning: S / Pinv of a solve on this handle were not block-symmetric (block (k, right) != block (k+1, left)^T): the handle now "
                    "runs kernels that read all three block columns (include/mpcg.h, BLOCK SYMMETRY)";
    return MPCG_OK;
}
// linsys_t = double, N <= 64: a lane quad per knot, the lower block triangle in the registers of ONE CU (pcg_lqk_f64.hip.h)
template <int NWR>
static int launch_lqk_f64_t(mpcg_handle* h, const PcgArgs64& a, uint32_t batch, hipStream_t st) {
    const size_t lds = pcg_lqk_lds_doubles(4 * NWR) * sizeof(double);
    void (*kern)(PcgArgs64) = pcg_lqk_f64_kernel<NWR>;
    if (lds > 48 * 1024)
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(batch), dim3(NWR * 256), lds, st, a);
    HIP_TRY(h, hipGetLastError());
    h->last = LastKernel{FAM_LQK64, 4 * NWR, 0, 0, 0, 0, (int)lds, 0};
    return MPCG_OK;
}

static int launch_f64(mpcg_handle* h, PcgArgs64 a, uint32_t batch, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    // 32 < N <= 64 (or "pcg_lqk" = 1 at any N <= 64): the lane-quad kernel, once the latch says the matrices are block-symmetric.  Also 16 < N <= 32
    // when the call brings at least four trajectories per CU: two 32-knot workgroups per CU take 4.2 us per iteration of 1,024 trajectories at
    // every such N and 30-39 us to start, the row-per-lane kernel 4.4 us and 38-64 us — 40-iteration solves of 1,024 trajectories: 200-205 against
    // 218-246 us (tools/_prof/f64_fixed_cost.py); below that batch the row-per-lane kernel's 1.0-1.1 us per iteration of ONE trajectory wins (1.7).
    const bool lqk_short = h->N > 16 && batch >= 4u * (uint32_t)h->num_cus;
    if (!h->generic && h->N <= kLqkMaxN && h->lqk != 0 && (h->lqk == 1 || (h->auto_cfg && (h->N > kRplMaxN64 || lqk_short) && h->cluster < 0))) {
        const int rc = f64_symmetry_latch(h, a, batch, st);
        if (rc != MPCG_OK) return rc;
        if (h->sym_state == 1) {
            HIP_TRY(h, hipSetDevice(h->device));
            return h->N <= 32 ? launch_lqk_f64_t<1>(h, a, batch, st) : launch_lqk_f64_t<2>(h, a, batch, st);
        }
    }
    if (!h->generic && h->N <= kRplMaxN64 && h->rpl != 0 && (h->rpl == 1 || h->auto_cfg)) {
        HIP_TRY(h, hipSetDevice(h->device));
        if (h->N <= 16) return a.pcols == 3 ? launch_rpl_f64_t<4, true>(h, a, batch, st) : launch_rpl_f64_t<4, false>(h, a, batch, st);
        return a.pcols == 3 ? launch_rpl_f64_t<8, true>(h, a, batch, st) : launch_rpl_f64_t<8, false>(h, a, batch, st);
    }
    if (h->generic) return launch_generic<double, 0>(h, a, batch, stream);
    // 64 < N <= 512, block-symmetric matrices: lane-quad clusters of ceil(N / 64) CUs
    if (h->N > kRplMaxN64 && h->cluster != 0 && h->lqk != 0 && lqkc_members(h) > 0) {
        const int rcl = f64_symmetry_latch(h, a, batch, st);
        if (rcl != MPCG_OK) return rcl;
        if (h->sym_state == 1) {
            const int rc = try_launch_cluster_lqk_f64(h, a, batch, st);
            if (rc != 1) return rc;
        }
    }
    {   // 32 < N <= 256: clusters of ceil(N / 32) CUs keep full block rows of S and Pinv in registers ("cluster" = 0: the streaming kernel below)
        const int rc = try_launch_cluster_f64(h, a, batch, st);
        if (rc != 1) return rc;
    }
    // The streaming kernel reads a third less when it may take block (k, right) from block (k+1, left) (mpcg.h, BLOCK SYMMETRY)
    {
        const int rc = f64_symmetry_latch(h, a, batch, st);
        if (rc != MPCG_OK) return rc;
    }
    a.lower = h->sym_state == 1 ? 1 : 0;
    return launch_generic<double, 14>(h, a, batch, stream);
}
// state_size != 14, float: the PcgArgs of the tuned path re-packed for the generic kernel
static int launch_generic_f32(mpcg_handle* h, const PcgArgs& a, uint32_t batch, hipStream_t st) {
    PcgArgsG<float> g;
    g.S = static_cast<const float*>(a.S); g.Pinv = static_cast<const float*>(a.Pinv); g.gamma = a.gamma; g.lambda = a.lambda;
    g.r_out = a.r_out; g.p_out = a.p_out; g.iters = a.iters; g.max_iter_exit = a.max_iter_exit;
    g.N = a.N; g.max_iter = a.max_iter; g.exit_tol = a.exit_tol; g.pcols = a.pcols;
    return launch_generic<float, 0>(h, g, batch, st);
}


extern "C" {

// Resident trajectories of the configuration a throughput-sized call (batch = max_batch) would launch.
int mpcg_check_pcg_occupancy(mpcg_handle* h, uint32_t* resident_trajectories) {
    if (!h || !resident_trajectories) return MPCG_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    int per_cu = 0;
    if (h->generic) {
        const size_t lds = pcg_generic_lds_elems((int)h->N, (int)h->n) * sizeof(float);
        const int nthr = pcg_generic_threads((int)h->N, (int)h->n);
        void (*kern)(PcgArgsG<float>) = nthr == F64_THREADS_WIDE ? pcg_generic_kernel<float, 0, F64_THREADS_WIDE> : pcg_generic_kernel<float, 0, F64_THREADS>;
        if (lds > 48 * 1024) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nthr, lds));
    } else if (use_rpl(h, 4, h->max_batch)) {
        // mirrors launch_pcg for a throughput-sized call (batch = max_batch): the row-per-lane kernel in the shape that call would take
        int nw, rho;
        rpl_shape(h, h->max_batch, &nw, &rho);
        const size_t lds = pcg_rpl_lds_floats((int)h->N, nw) * sizeof(float);
#define X(NW_, RHO_) if (nw == NW_ && rho == RHO_) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_rpl_kernel<NW_, RHO_, true>, NW_ * 64, lds));
        MPCG_RPL_VARIANTS(X)
#undef X
    } else if (use_lpk(h, 4, h->max_batch) && use_lqb(h, 4)) {
        const int nmax = h->N <= 32 ? 32 : h->N <= 64 ? 64 : 128;
        const size_t lds = pcg_lqb_lds_floats(nmax) * sizeof(float);
        const void* kern = nmax == 32 ? reinterpret_cast<const void*>(pcg_lqb_kernel<32, true>) : nmax == 64 ? reinterpret_cast<const void*>(pcg_lqb_kernel<64, true>) : reinterpret_cast<const void*>(pcg_lqb_kernel<128, true>);
        HIP_TRY(h, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (nmax == 32) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lqb_kernel<32, true>, 128, lds));
        else if (nmax == 64) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lqb_kernel<64, true>, 256, lds));
        else HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lqb_kernel<128, true>, 512, lds));
    } else if (use_lpk(h, 4, h->max_batch)) {
        const size_t lds = pcg_lpk_lds_floats(h->N <= 32 ? 2 : h->N <= 64 ? 4 : 8) * sizeof(float);
        if (h->N <= 32) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lpk_kernel<0>, 128, lds));
        else if (h->N <= 64) HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lpk_kernel<1>, 256, lds));
        else {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(pcg_lpk_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pcg_lpk_kernel<2>, 512, lds));
        }
    } else if (h->cluster != 0 && (h->cluster > 0 || (h->auto_cfg && h->N > kLpbMaxN)) && lpkc_members(h) > 0) {
        // clustered kernel (the condition of try_launch_cluster): resident CLUSTERS = trajectories in flight, the count the launch itself uses
        *resident_trajectories = lpkc_resident_clusters(h, lpkc_members(h));
        return MPCG_OK;
    } else {
        PcgKnobs k = h->k;
        if (h->auto_cfg) choose_auto(h, k, h->max_batch, 4);
        const int rc = occupancy(h, k, &per_cu);
        if (rc != MPCG_OK) return rc;
    }
    if (per_cu < 1) return fail(h, MPCG_ERR_UNSUPPORTED, "PCG workgroup does not fit on a CU");
    *resident_trajectories = (uint32_t)per_cu * (uint32_t)h->num_cus;
    return MPCG_OK;
}

int mpcg_pcg_solve(mpcg_handle* h, const float* d_S, const float* d_Pinv, const float* d_gamma, float* d_lambda,
                   uint32_t batch, uint32_t max_iter, float exit_tol, mpcg_precond precond,
                   uint32_t* d_iters, uint8_t* d_max_iter_exit, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (!d_S || !d_Pinv || !d_gamma || !d_lambda || !d_iters || !d_max_iter_exit)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve: null device pointer");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve: batch exceeds max_batch");
    if (precond != MPCG_PRECOND_JACOBI && precond != MPCG_PRECOND_SS)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve: bad preconditioner");
    if (max_iter > 0x7fffffffu) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve: max_iter too large");
    if ((reinterpret_cast<uintptr_t>(d_S) | reinterpret_cast<uintptr_t>(d_Pinv)) & 15u)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve: d_S / d_Pinv must be 16-byte aligned");
    PcgArgs a;
    a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda;
    a.r_out = nullptr; a.p_out = nullptr;
    a.iters = d_iters; a.max_iter_exit = d_max_iter_exit;
    a.N = (int)h->N; a.max_iter = (int)max_iter; a.exit_tol = exit_tol; a.pcols = (int)precond; a.lds_rows = 0;
    // Calls with more trajectories than CUs: dispatch longest-expected first, the expectation being the previous call's iteration
    // counts for the same batch (pcg_kernels.hip.h: sched_order_kernel).  A scheduling hint only: no result depends on it.
    const bool hinted = h->sched_hint && !h->generic && batch > (uint32_t)h->num_cus;
    if (hinted) { a.order = h->sched_order; a.order_tag = batch; }       // (used only if the stored permutation was made for this batch: sched_pick)
    const int rc = launch_pcg(h, a, batch, static_cast<hipStream_t>(stream), 4);
    if (rc == MPCG_OK && hinted && (h->last.family == FAM_LPK || h->last.family == FAM_LQB || h->last.family == FAM_LPKC || h->last.family == FAM_RPL)) {
        hipLaunchKernelGGL(sched_order_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), d_iters, (int)batch, h->sched_order);
        HIP_TRY(h, hipGetLastError());
    }
    return rc;
}

int mpcg_pcg_solve_ref(mpcg_handle* h, float* d_S, float* d_Pinv, float* d_gamma, float* d_lambda,
                       float* d_r, float* d_p, float* d_v_temp, float* d_eta_new_temp,
                       uint32_t* d_pcg_iters, uint8_t* d_pcg_exit,
                       uint32_t pcg_max_iter, float pcg_exit_tol, void* stream) {
    (void)d_v_temp; (void)d_eta_new_temp;   // reference's per-block partial-sum scratch: not needed
    if (!h) return MPCG_ERR_INVALID;
    if (!d_S || !d_Pinv || !d_gamma || !d_lambda || !d_pcg_iters || !d_pcg_exit)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_ref: null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_S) | reinterpret_cast<uintptr_t>(d_Pinv)) & 15u)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_ref: d_S / d_Pinv must be 16-byte aligned");
    if (pcg_max_iter > 0x7fffffffu) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_ref: max_iter too large");
    PcgArgs a;
    a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda;
    a.r_out = d_r; a.p_out = d_p;
    a.iters = d_pcg_iters; a.max_iter_exit = d_pcg_exit;
    a.N = (int)h->N; a.max_iter = (int)pcg_max_iter; a.exit_tol = pcg_exit_tol; a.pcols = 3; a.lds_rows = 0;
    return launch_pcg(h, a, 1, static_cast<hipStream_t>(stream), 4);
}

int mpcg_bt_spmv(mpcg_handle* h, const float* d_M, const float* d_x, float* d_y, uint32_t batch, int cols, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_bt_spmv: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_M || !d_x || !d_y) return fail(h, MPCG_ERR_INVALID, "mpcg_bt_spmv: null device pointer");
    if (cols != 1 && cols != 3) return fail(h, MPCG_ERR_INVALID, "mpcg_bt_spmv: cols must be 1 or 3");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_bt_spmv: batch exceeds max_batch");
    if (reinterpret_cast<uintptr_t>(d_M) & 15u) return fail(h, MPCG_ERR_INVALID, "mpcg_bt_spmv: d_M must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y)) & 7u)
        return fail(h, MPCG_ERR_INVALID, "mpcg_bt_spmv: d_x / d_y must be 8-byte aligned");
    HIP_TRY(h, hipSetDevice(h->device));
    SpmvArgs a{d_M, d_x, d_y, (int)h->N, (int)batch, cols};
    constexpr int NW = 4;
    const long total = (long)batch * h->N;
    const long tasks = h->spmv_mfma ? total : (total + SPMV_SPAN - 1) / SPMV_SPAN;     // a wavefront's unit: a block row (MFMA experiment) or a span of rows
    long blocks = (tasks + NW - 1) / NW;
    const long cap = (long)h->num_cus * h->spmv_blocks_per_cu;
    if (blocks > cap) blocks = cap;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->spmv_mfma) hipLaunchKernelGGL((bt_spmv_mfma_kernel<NW>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
    else if (h->nt_loads) hipLaunchKernelGGL((bt_spmv_kernel<NW, true>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
    else hipLaunchKernelGGL((bt_spmv_kernel<NW, false>), dim3((unsigned)blocks), dim3(NW * 64), 0, st, a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}


int mpcg_probe_hbm_read(mpcg_handle* h, const void* d_src, size_t bytes, float* d_sink, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (!d_src || !d_sink) return fail(h, MPCG_ERR_INVALID, "mpcg_probe_hbm_read: null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_src) & 15u) || (bytes & 15u)) return fail(h, MPCG_ERR_INVALID, "mpcg_probe_hbm_read: d_src and bytes must be multiples of 16");
    if (bytes == 0) return MPCG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(hbm_read_probe_kernel, dim3((unsigned)(2 * h->num_cus)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const f4*>(d_src), d_sink, bytes / 16);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_convert_f32_to_f16(mpcg_handle* h, const float* d_src, uint16_t* d_dst, size_t count, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (!d_src || !d_dst) return fail(h, MPCG_ERR_INVALID, "mpcg_convert_f32_to_f16: null device pointer");
    if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) & 15u)
        return fail(h, MPCG_ERR_INVALID, "mpcg_convert_f32_to_f16: pointers must be 16-byte aligned");
    if (count == 0) return MPCG_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t blocks = (count + 2047) / 2048;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_src,
                       reinterpret_cast<_Float16*>(d_dst), count);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

int mpcg_pcg_solve_f16(mpcg_handle* h, const uint16_t* d_S16, const uint16_t* d_Pinv16, const float* d_gamma,
                       float* d_lambda, uint32_t batch, uint32_t max_iter, float exit_tol, mpcg_precond precond,
                       uint32_t* d_iters, uint8_t* d_max_iter_exit, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (h->generic) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_pcg_solve_f16: state_size = 14 only (other state sizes: PCG entry points through the generic kernel)");
    if (!d_S16 || !d_Pinv16 || !d_gamma || !d_lambda || !d_iters || !d_max_iter_exit)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f16: null device pointer");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f16: batch exceeds max_batch");
    if (precond != MPCG_PRECOND_JACOBI && precond != MPCG_PRECOND_SS)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f16: bad preconditioner");
    if (max_iter > 0x7fffffffu) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f16: max_iter too large");
    if ((reinterpret_cast<uintptr_t>(d_S16) | reinterpret_cast<uintptr_t>(d_Pinv16)) & 3u)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f16: d_S16 / d_Pinv16 must be 4-byte aligned");
    PcgArgs a;
    a.S = d_S16; a.Pinv = d_Pinv16; a.gamma = d_gamma; a.lambda = d_lambda;
    a.r_out = nullptr; a.p_out = nullptr;
    a.iters = d_iters; a.max_iter_exit = d_max_iter_exit;
    a.N = (int)h->N; a.max_iter = (int)max_iter; a.exit_tol = exit_tol; a.pcols = (int)precond; a.lds_rows = 0;
    return launch_pcg(h, a, batch, static_cast<hipStream_t>(stream), 2);
}

int mpcg_pcg_solve_f64(mpcg_handle* h, const double* d_S, const double* d_Pinv, const double* d_gamma, double* d_lambda,
                       uint32_t batch, uint32_t max_iter, double exit_tol, mpcg_precond precond,
                       uint32_t* d_iters, uint8_t* d_max_iter_exit, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (!d_S || !d_Pinv || !d_gamma || !d_lambda || !d_iters || !d_max_iter_exit)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f64: null device pointer");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f64: batch exceeds max_batch");
    if (precond != MPCG_PRECOND_JACOBI && precond != MPCG_PRECOND_SS)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f64: bad preconditioner");
    if (max_iter > 0x7fffffffu) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_f64: max_iter too large");
    PcgArgs64 a;
    a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda; a.r_out = nullptr; a.p_out = nullptr;
    a.iters = d_iters; a.max_iter_exit = d_max_iter_exit; a.N = (int)h->N; a.max_iter = (int)max_iter;
    a.exit_tol = exit_tol; a.pcols = precond == MPCG_PRECOND_SS ? 3 : 1;
    return launch_f64(h, a, batch, stream);
}

int mpcg_pcg_solve_ref_f64(mpcg_handle* h, double* d_S, double* d_Pinv, double* d_gamma, double* d_lambda,
                           double* d_r, double* d_p, double* /*d_v_temp*/, double* /*d_eta_new_temp*/,
                           uint32_t* d_pcg_iters, uint8_t* d_pcg_exit, uint32_t pcg_max_iter, double pcg_exit_tol, void* stream) {
    if (!h) return MPCG_ERR_INVALID;
    if (!d_S || !d_Pinv || !d_gamma || !d_lambda || !d_pcg_iters || !d_pcg_exit)
        return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_ref_f64: null device pointer");
    if (pcg_max_iter > 0x7fffffffu) return fail(h, MPCG_ERR_INVALID, "mpcg_pcg_solve_ref_f64: max_iter too large");
    PcgArgs64 a;
    a.S = d_S; a.Pinv = d_Pinv; a.gamma = d_gamma; a.lambda = d_lambda; a.r_out = d_r; a.p_out = d_p;
    a.iters = d_pcg_iters; a.max_iter_exit = d_pcg_exit; a.N = (int)h->N; a.max_iter = (int)pcg_max_iter;
    a.exit_tol = pcg_exit_tol; a.pcols = 3;
    return launch_f64(h, a, 1, stream);
}

}  // extern "C"

extern "C" {

int mpcg_debug_read_cluster_scratch(mpcg_handle* h, unsigned long long* out, int count) {
    if (!h || !out || count < 0 || (size_t)count > cluster_alloc_words(h)) return MPCG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return MPCG_ERR_HIP;
    return hipMemcpy(out, h->cluster_scratch, sizeof(unsigned long long) * (size_t)count, hipMemcpyDeviceToHost) == hipSuccess ? MPCG_OK : MPCG_ERR_HIP;
}

#ifdef MPCG_PROF
// diagnostic build only (tools/prof_phases.py): the s_memtime stamps of workgroup 0
int mpcg_debug_read_prof(long long* out, int count) {
    if (!out || count < 0 || count > 16 * 32) return MPCG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return MPCG_ERR_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mpcg::g_pcg_prof), sizeof(long long) * (size_t)count) == hipSuccess ? MPCG_OK : MPCG_ERR_HIP;
}
// (entry time, exit time, HW_ID, XCC_ID) of the first `count` / 4 workgroups of the last lane-quad launch
int mpcg_debug_read_wg_prof(long long* out, int count) {
    if (!out || count < 0 || count > 4096 * 4) return MPCG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return MPCG_ERR_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mpcg::g_wg_prof), sizeof(long long) * (size_t)count) == hipSuccess ? MPCG_OK : MPCG_ERR_HIP;
}
#endif

}  // extern "C"
