// gpu_pcg.cuh — source-level shim with the names the reference takes from its GBD-PCG submodule,
// implemented over the C ABI of libmpcg_hip.so (include/mpcg.h).  Put this directory on the include
// path in place of -IGBD-PCG/include (reference Makefile:5) and link -lmpcg_hip.
//
//   reference use                                        here
//   pcg_config<T>            include/mpcsim.cuh:213-216  same fields
//   pcgSharedMemSize<T>(n,N) include/pcg/sqp.cuh:151     mpcg_pcg_lds_bytes
//   checkPcgOccupancy<T>(..) examples/track_iiwa_pcg.cu:24  mpcg_check_pcg_occupancy (never aborts for a supported shape)
//   pcg<T,n,N>               include/pcg/sqp.cuh:129     a HOST launcher with the kernel's 12 parameters
//   cudaLaunchCooperativeKernel(pcg_kernel, N, threads, args, smem)   include/pcg/sqp.cuh:230
//                                                        -> mpcgLaunchPcg(pcg_kernel, N, threads, args, smem)
//                                                           (the ONE line of sqp.cuh that changes)
// T = float (linsys_t with USE_DOUBLES=0, include/common/settings.cuh:41-49) is the headline path; T = double (USE_DOUBLES=1) runs the
// row-per-lane kernel in double to 32 knots and the lane-quad kernels beyond (one CU to 64 knots, ceil(N / 64) CUs to 512).
// STATE_SIZE = 14.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <type_traits>
#include <utility>

#include "../mpcg.h"
#include "gpuassert.cuh"
#include "utils.cuh"

template <typename T>
struct pcg_config {
    T pcg_exit_tol = static_cast<T>(1e-6);
    uint32_t pcg_max_iter = 200;
    unsigned pcg_block = 128;          // reference launch shape (PCG_NUM_THREADS); informational here
    unsigned pcg_grid = 0;
    bool empty_pinv = false;
};

namespace mpcg_compat {

inline void die(const char* what, const mpcg_handle* h) {
    fprintf(stderr, "mpcg: %s: %s\n", what, mpcg_last_error(h));
    exit(EXIT_FAILURE);                 // the reference's error convention: gpuErrchk aborts
}

// One cached handle per (host thread, device, knot_points): a handle is re-entrant per handle, not across threads, so two host threads
// (the native multi-device driver pattern: a thread + stream + handle per device, examples/multi_gpu_pcg.cpp) never share one.  The cache
// is THREAD-LOCAL and owns its handles: when a host thread ends, its handles — cluster scratch, dispatch order, the latch's pinned word and
// event — are destroyed with it, so per-solve std::thread churn does not grow device memory and a reused thread::id cannot inherit
// another thread's handle or latch state.  The solver keeps no per-call global scratch.
struct HandleCache {
    std::map<std::pair<int, uint32_t>, mpcg_handle*> handles;
    ~HandleCache() {
        for (auto& kv : handles) (void)mpcg_destroy(kv.second);
    }
};

inline mpcg_handle* handle_for(uint32_t state_size, uint32_t knot_points) {
    static thread_local HandleCache cache;
    int dev = 0;
    gpuErrchk(hipGetDevice(&dev));
    const auto key = std::make_pair(dev, knot_points);
    auto it = cache.handles.find(key);
    if (it != cache.handles.end()) return it->second;
    mpcg_handle* h = nullptr;
    if (mpcg_create(&h, dev, state_size, knot_points, 1) != MPCG_OK) die("mpcg_create", nullptr);
    cache.handles[key] = h;
    return h;
}

// The library's lower-triangle kernels need block-symmetric S / Pinv (include/mpcg.h, BLOCK SYMMETRY — the reference's construction is); a handle
// whose latch found a violation solves with three-column kernels from then on and says so in mpcg_last_error() only.  Surface it once per
// handle: a maintainer who swaps in a hand-made Pinv should see why the solve got slower.  (Host-side state only: no synchronisation.)
inline void warn_once_if_not_block_symmetric(mpcg_handle* h) {
    static thread_local std::set<mpcg_handle*> warned;
    int state = 0;
    if (mpcg_get_option(h, "symmetry_state", &state) == MPCG_OK && state == 2 && warned.insert(h).second)
        fprintf(stderr, "mpcg: %s\n", "S / Pinv of a solve on this handle were not block-symmetric (block (k, right) != block (k+1, left)^T): "
                                       "three-column kernels from now on (include/mpcg.h, BLOCK SYMMETRY)");
}

// the stream the next pcg<> launch of THIS host thread goes to (mpcgLaunchPcg's optional last argument sets it around the call; the
// reference's own call site runs on the default stream, include/pcg/sqp.cuh:225-236)
inline hipStream_t& launch_stream() {
    static thread_local hipStream_t s = nullptr;
    return s;
}

}  // namespace mpcg_compat

template <typename T>
size_t pcgSharedMemSize(uint32_t state_size, uint32_t knot_points) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "linsys_t is float or double");
    if constexpr (std::is_same<T, double>::value) return mpcg_pcg_lds_bytes_f64(state_size, knot_points);
    else return mpcg_pcg_lds_bytes(state_size, knot_points);
}

template <typename T>
bool checkPcgOccupancy(void* /*kernel*/, dim3 /*block*/, uint32_t state_size, uint32_t knot_points) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "linsys_t is float or double");
    uint32_t resident = 0;
    mpcg_handle* h = mpcg_compat::handle_for(state_size, knot_points);
    if (mpcg_check_pcg_occupancy(h, &resident) != MPCG_OK) mpcg_compat::die("checkPcgOccupancy", h);
    return resident > 0;
}

// Host launcher carrying the reference kernel's parameter list (include/pcg/sqp.cuh:137-150).
// (template parameter names differ from the reference's STATE_SIZE / KNOT_POINTS MACROS, include/common/settings.cuh:5-12, which are
//  defined before this header is included in the reference's translation units)
template <typename T, uint32_t StateSize, uint32_t KnotPoints>
void pcg(T* d_S, T* d_Pinv, T* d_gamma, T* d_lambda, T* d_r, T* d_p, T* d_v_temp, T* d_eta_new_temp,
         uint32_t* d_iters, bool* d_max_iter_exit, uint32_t max_iter, T exit_tol) {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "linsys_t is float or double");
    static_assert(sizeof(bool) == 1, "exit flag is one byte");
    mpcg_handle* h = mpcg_compat::handle_for(StateSize, KnotPoints);
    int rc;
    if constexpr (std::is_same<T, float>::value)
        rc = mpcg_pcg_solve_ref(h, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, d_v_temp, d_eta_new_temp, d_iters,
                                reinterpret_cast<uint8_t*>(d_max_iter_exit), max_iter, exit_tol, mpcg_compat::launch_stream());
    else
        rc = mpcg_pcg_solve_ref_f64(h, d_S, d_Pinv, d_gamma, d_lambda, d_r, d_p, d_v_temp, d_eta_new_temp, d_iters,
                                    reinterpret_cast<uint8_t*>(d_max_iter_exit), max_iter, exit_tol, mpcg_compat::launch_stream());
    if (rc != MPCG_OK) mpcg_compat::die("pcg", h);
    mpcg_compat::warn_once_if_not_block_symmetric(h);
}

// Replaces cudaLaunchCooperativeKernel at include/pcg/sqp.cuh:230.  `kernel` is the void* the call
// site made from pcg<T,n,N>; `args` is its pcgKernelArgs array (addresses of the 12 arguments).
// Grid/block/smem are accepted and ignored: the launch shape is the library's business.  T defaults to float; inside
// sqpSolvePcg<T> write mpcgLaunchPcg<T>(...) if the build may use linsys_t = double.  `stream` (optional, as cudaLaunchCooperativeKernel's
// last argument): the stream of this launch; default = the null stream, as at the reference's call site.
template <typename T = float>
inline hipError_t mpcgLaunchPcg(void* kernel, unsigned /*grid*/, unsigned /*block*/, void** args, size_t /*smem*/, hipStream_t stream = nullptr) {
    using F = void (*)(T*, T*, T*, T*, T*, T*, T*, T*, uint32_t*, bool*, uint32_t, T);
    F f = reinterpret_cast<F>(kernel);
    struct StreamScope {
        hipStream_t prev;
        explicit StreamScope(hipStream_t s) : prev(mpcg_compat::launch_stream()) { mpcg_compat::launch_stream() = s; }
        ~StreamScope() { mpcg_compat::launch_stream() = prev; }
    } scope(stream);
    f(*static_cast<T**>(args[0]), *static_cast<T**>(args[1]), *static_cast<T**>(args[2]),
      *static_cast<T**>(args[3]), *static_cast<T**>(args[4]), *static_cast<T**>(args[5]),
      *static_cast<T**>(args[6]), *static_cast<T**>(args[7]), *static_cast<uint32_t**>(args[8]),
      *static_cast<bool**>(args[9]), *static_cast<uint32_t*>(args[10]), *static_cast<T*>(args[11]));
    return hipGetLastError();
}
