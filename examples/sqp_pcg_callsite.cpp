// sqp_pcg_callsite.cpp — the reference's PCG call site (include/pcg/sqp.cuh:116-151, 224-244,
// 380-386) written against the shim include/gbd_pcg_compat/gpu_pcg.cuh + libmpcg_hip.so: same
// buffers, same argument array, same launch line (with mpcgLaunchPcg in place of
// cudaLaunchCooperativeKernel), same two D2H copies.  Solves one synthetic negative-definite
// block-tridiagonal system (built here on the host) and checks the residual on the CPU.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude/gbd_pcg_compat examples/sqp_pcg_callsite.cpp -Lmpcgpu_amd -lmpcg_hip
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "gpu_pcg.cuh"

#define STATE_SIZE 14
#ifndef KNOT_POINTS           // (-DKNOT_POINTS=128 -DUSE_DOUBLES: pcg<double, 14, 128> — the clustered row-per-lane kernel behind the same call site)
#define KNOT_POINTS 32
#endif
#define PCG_NUM_THREADS 128
#ifdef USE_DOUBLES            // linsys_t = double, include/common/settings.cuh:41-49
typedef double T;
#else
typedef float T;
#endif

int main() {
    const uint32_t state_size = STATE_SIZE, knot_points = KNOT_POINTS;
    const uint32_t states_sq = state_size * state_size;
    const int n = state_size, N = knot_points;

    if (!checkPcgOccupancy<T>((void*)pcg<T, STATE_SIZE, KNOT_POINTS>, PCG_NUM_THREADS, state_size, knot_points)) return 2;

    // host problem: -S = tridiag(E^T, D, E) with D = 4I + small symmetric, |E| small; stored negated.
    std::vector<T> h_S(3 * states_sq * knot_points, NAN), h_Pinv(3 * states_sq * knot_points, NAN);
    std::vector<T> h_gamma(n * N), h_lambda(n * N, 0.f);
    auto rnd = [s = 12345u]() mutable { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int k = 0; k < N; ++k) {
        T* row = &h_S[(size_t)k * 3 * states_sq];
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) {
                T v = (i == j) ? 4.0f + 0.5f * rnd() : 0.1f * rnd();
                row[states_sq + i + j * n] = -v;
                row[states_sq + j + i * n] = -v;
            }
        if (k > 0)
            for (int e = 0; e < (int)states_sq; ++e) row[e] = -0.2f * rnd();                 // left block
        for (int i = 0; i < n; ++i) h_gamma[k * n + i] = rnd();
    }
    for (int k = 0; k + 1 < N; ++k)                                                            // right = left(k+1)^T
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
                h_S[(size_t)k * 3 * states_sq + 2 * states_sq + i + j * n] = h_S[(size_t)(k + 1) * 3 * states_sq + j + i * n];
    for (int k = 0; k < N; ++k)                                                                // Pinv = diag(1/S_ii), block-diagonal
        for (int c = 0; c < 3; ++c)
            for (int e = 0; e < (int)states_sq; ++e) {
                bool unused = (k == 0 && c == 0) || (k == N - 1 && c == 2);
                T v = (c == 1 && e % (n + 1) == 0) ? 1.0f / h_S[(size_t)k * 3 * states_sq + states_sq + e] : 0.0f;
                h_Pinv[(size_t)k * 3 * states_sq + c * states_sq + e] = unused ? NAN : v;
            }

    // ---- from here on: include/pcg/sqp.cuh:100-151 ----
    T *d_S, *d_gamma, *d_lambda, *d_Pinv;
    gpuErrchk(hipMalloc(&d_S, 3 * states_sq * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_gamma, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_lambda, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_Pinv, 3 * states_sq * knot_points * sizeof(T)));
    T *d_r, *d_p, *d_v_temp, *d_eta_new_temp;
    gpuErrchk(hipMalloc(&d_r, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_p, state_size * knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_v_temp, knot_points * sizeof(T)));
    gpuErrchk(hipMalloc(&d_eta_new_temp, knot_points * sizeof(T)));
    gpuErrchk(hipMemcpy(d_S, h_S.data(), h_S.size() * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_Pinv, h_Pinv.data(), h_Pinv.size() * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_gamma, h_gamma.data(), h_gamma.size() * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_lambda, h_lambda.data(), h_lambda.size() * sizeof(T), hipMemcpyHostToDevice));

    pcg_config<T> config;
    config.pcg_block = PCG_NUM_THREADS;
    config.pcg_exit_tol = 1e-10f;
    config.pcg_max_iter = 200;

    void* pcg_kernel = (void*)pcg<T, STATE_SIZE, KNOT_POINTS>;
    uint32_t pcg_iters;
    uint32_t* d_pcg_iters;
    gpuErrchk(hipMalloc(&d_pcg_iters, sizeof(uint32_t)));
    bool pcg_exit;
    bool* d_pcg_exit;
    gpuErrchk(hipMalloc(&d_pcg_exit, sizeof(bool)));

    void* pcgKernelArgs[] = {
        (void*)&d_S, (void*)&d_Pinv, (void*)&d_gamma, (void*)&d_lambda, (void*)&d_r, (void*)&d_p,
        (void*)&d_v_temp, (void*)&d_eta_new_temp, (void*)&d_pcg_iters, (void*)&d_pcg_exit,
        (void*)&config.pcg_max_iter, (void*)&config.pcg_exit_tol};
    size_t ppcg_kernel_smem_size = pcgSharedMemSize<T>(state_size, knot_points);

    // ---- include/pcg/sqp.cuh:230-232 ----
    gpuErrchk(mpcgLaunchPcg<T>(pcg_kernel, knot_points, PCG_NUM_THREADS, pcgKernelArgs, ppcg_kernel_smem_size));
    gpuErrchk(hipMemcpy(&pcg_iters, d_pcg_iters, sizeof(uint32_t), hipMemcpyDeviceToHost));
    gpuErrchk(hipMemcpy(&pcg_exit, d_pcg_exit, sizeof(bool), hipMemcpyDeviceToHost));
    gpuErrchk(hipMemcpy(h_lambda.data(), d_lambda, h_lambda.size() * sizeof(T), hipMemcpyDeviceToHost));

    // CPU check: ||gamma - S lambda||_inf / ||gamma||_inf  (double accumulation)
    double rmax = 0, gmax = 0;
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i) {
            double acc = 0;
            for (int c = 0; c < 3; ++c) {
                int kc = k + c - 1;
                if (kc < 0 || kc >= N) continue;
                for (int j = 0; j < n; ++j)
                    acc += (double)h_S[(size_t)k * 3 * states_sq + c * states_sq + i + j * n] * h_lambda[kc * n + j];
            }
            rmax = fmax(rmax, fabs(h_gamma[k * n + i] - acc));
            gmax = fmax(gmax, fabs(h_gamma[k * n + i]));
        }
    // ---- the same launch line from TWO host threads, each on its own stream (mpcgLaunchPcg's optional last argument) and its own iterate
    // buffers: the shim gives every host thread its own library handle, so the launches may overlap; both must reproduce the bits above ----
    int mt_ok = 1;
    {
        auto worker = [&](int id, int* ok) {
            hipStream_t st;
            gpuErrchk(hipStreamCreate(&st));
            T *l, *r, *p_, *v, *e;
            uint32_t* it;
            bool* ex;
            gpuErrchk(hipMalloc(&l, n * N * sizeof(T))); gpuErrchk(hipMalloc(&r, n * N * sizeof(T))); gpuErrchk(hipMalloc(&p_, n * N * sizeof(T)));
            gpuErrchk(hipMalloc(&v, N * sizeof(T))); gpuErrchk(hipMalloc(&e, N * sizeof(T)));
            gpuErrchk(hipMalloc(&it, sizeof(uint32_t))); gpuErrchk(hipMalloc(&ex, sizeof(bool)));
            uint32_t mi = config.pcg_max_iter;
            T tol = config.pcg_exit_tol;
            void* a[] = {(void*)&d_S, (void*)&d_Pinv, (void*)&d_gamma, (void*)&l, (void*)&r, (void*)&p_, (void*)&v, (void*)&e, (void*)&it, (void*)&ex, (void*)&mi, (void*)&tol};
            std::vector<T> out(n * N);
            for (int rep = 0; rep < 10 && *ok; ++rep) {
                gpuErrchk(hipMemsetAsync(l, 0, n * N * sizeof(T), st));
                gpuErrchk(mpcgLaunchPcg<T>(pcg_kernel, knot_points, PCG_NUM_THREADS, a, ppcg_kernel_smem_size, st));
                gpuErrchk(hipMemcpyAsync(out.data(), l, n * N * sizeof(T), hipMemcpyDeviceToHost, st));
                gpuErrchk(hipStreamSynchronize(st));
                if (memcmp(out.data(), h_lambda.data(), n * N * sizeof(T)) != 0) *ok = 0;
            }
            (void)id;
            gpuErrchk(hipFree(l)); gpuErrchk(hipFree(r)); gpuErrchk(hipFree(p_)); gpuErrchk(hipFree(v)); gpuErrchk(hipFree(e)); gpuErrchk(hipFree(it)); gpuErrchk(hipFree(ex));
            gpuErrchk(hipStreamDestroy(st));
        };
        int ok0 = 1, ok1 = 1;
        std::thread t0(worker, 0, &ok0), t1(worker, 1, &ok1);
        t0.join(); t1.join();
        mt_ok = ok0 && ok1;
    }
    printf("{\"pcg_iters\": %u, \"pcg_exit\": %d, \"smem\": %zu, \"rel_residual\": %.3e, \"two_threads_two_streams_same_bits\": %d}\n", pcg_iters, (int)pcg_exit,
           ppcg_kernel_smem_size, rmax / gmax, mt_ok);

    // ---- include/pcg/sqp.cuh:380-386 ----
    gpuErrchk(hipFree(d_pcg_iters));
    gpuErrchk(hipFree(d_pcg_exit));
    gpuErrchk(hipFree(d_Pinv));
    gpuErrchk(hipFree(d_r));
    gpuErrchk(hipFree(d_p));
    gpuErrchk(hipFree(d_v_temp));
    gpuErrchk(hipFree(d_eta_new_temp));
    gpuErrchk(hipFree(d_S));
    gpuErrchk(hipFree(d_gamma));
    gpuErrchk(hipFree(d_lambda));
    return (pcg_exit == false && pcg_iters > 0 && rmax / gmax < 1e-4 && mt_ok) ? 0 : 1;
}
