// sqp_linsys_chain.cpp — one SQP iteration's linear-algebra chain exactly as include/pcg/sqp.cuh writes it
// (:207-219 form_schur_system, :230-232 PCG launch + copies, :250-259 compute_dz), on synthetic KKT blocks,
// compiled against the shim headers and libmpcg_hip.so.  Checks on the CPU that the step satisfies the
// regularised KKT conditions:  (G + rho I) dz + C^T lambda = g   and   C dz = c.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude/gbd_pcg_compat -Iinclude/mpcgpu_compat examples/sqp_linsys_chain.cpp -Lmpcgpu_amd -lmpcg_hip
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "gpu_pcg.cuh"
#include "linsys_steps.cuh"

#define STATE_SIZE 14
#define KNOT_POINTS 16
#define PCG_NUM_THREADS 128
#ifdef USE_DOUBLES            // linsys_t = double (include/common/settings.cuh:41-49): the same chain through form_schur_system<double>,
typedef double T;             // pcg<double, n, N> and compute_dz<double>; the step then satisfies the KKT conditions to 1e-9 instead of 1e-3
#else
typedef float T;
#endif

int main(int argc, char** argv) {
    const bool use_direct = argc > 1 && std::string(argv[1]) == "--direct";   // block_solve_schur instead of pcg<>
    const uint32_t state_size = STATE_SIZE, control_size = 7, knot_points = KNOT_POINTS;
    const int n = 14, m = 7, N = KNOT_POINTS, nn = n * n, mm = m * m, nm = n * m;
    const size_t Gsz = (size_t)(nn + mm) * N - mm, Csz = (size_t)(nn + nm) * (N - 1), gsz = (size_t)(n + m) * N - m;
    const T rho = 0.5f;
    std::vector<T> G(Gsz, 0.f), C(Csz), g(gsz), c(n * N), Graw;
    auto rnd = [s = 777u]() mutable { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int k = 0; k < N; ++k) {
        T* Q = &G[(size_t)k * (nn + mm)];
        for (int j = 0; j < n; ++j)
            for (int i = 0; i <= j; ++i) { T v = (i == j) ? 1.0f + rnd() * 0.2f : 0.05f * rnd(); Q[i + j * n] = v; Q[j + i * n] = v; }
        if (k < N - 1) {
            T* R = Q + nn;
            for (int i = 0; i < m; ++i) R[i + i * m] = 0.3f + 0.1f * rnd();
            T* A = &C[(size_t)k * (nn + nm)];
            for (int e = 0; e < nn; ++e) A[e] = -((e % n == e / n) ? 1.0f : 0.0f) - 0.1f * rnd();   // stored negated
            for (int e = 0; e < nm; ++e) A[nn + e] = -0.2f * rnd();
        }
    }
    for (auto& v : g) v = rnd();
    for (auto& v : c) v = 0.1f * rnd();
    for (int i = 0; i < n; ++i) c[i] = 0.f;
    Graw = G;

    T *d_G_dense, *d_C_dense, *d_g, *d_c, *d_S, *d_Pinv, *d_gamma, *d_lambda, *d_dz, *d_r, *d_p, *d_v_temp, *d_eta_new_temp;
    gpuErrchk(hipMalloc(&d_G_dense, Gsz * sizeof(T)));
    gpuErrchk(hipMalloc(&d_C_dense, Csz * sizeof(T)));
    gpuErrchk(hipMalloc(&d_g, gsz * sizeof(T)));
    gpuErrchk(hipMalloc(&d_c, n * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_S, 3 * nn * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_Pinv, 3 * nn * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_gamma, n * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_lambda, n * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_dz, gsz * sizeof(T)));
    gpuErrchk(hipMalloc(&d_r, n * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_p, n * N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_v_temp, N * sizeof(T)));
    gpuErrchk(hipMalloc(&d_eta_new_temp, N * sizeof(T)));
    gpuErrchk(hipMemcpy(d_G_dense, G.data(), Gsz * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_C_dense, C.data(), Csz * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_g, g.data(), gsz * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemcpy(d_c, c.data(), n * N * sizeof(T), hipMemcpyHostToDevice));
    gpuErrchk(hipMemset(d_lambda, 0, n * N * sizeof(T)));
    T* d_Ginv_dense = d_G_dense;                                           // include/pcg/sqp.cuh:98

    pcg_config<T> config;
    config.pcg_exit_tol = sizeof(T) == 8 ? (T)1e-26 : (T)1e-12;
    config.pcg_max_iter = 1000;
    void* pcg_kernel = (void*)pcg<T, STATE_SIZE, KNOT_POINTS>;
    uint32_t pcg_iters, *d_pcg_iters;
    bool pcg_exit, *d_pcg_exit;
    gpuErrchk(hipMalloc(&d_pcg_iters, sizeof(uint32_t)));
    gpuErrchk(hipMalloc(&d_pcg_exit, sizeof(bool)));
    void* pcgKernelArgs[] = {(void*)&d_S, (void*)&d_Pinv, (void*)&d_gamma, (void*)&d_lambda, (void*)&d_r, (void*)&d_p,
                             (void*)&d_v_temp, (void*)&d_eta_new_temp, (void*)&d_pcg_iters, (void*)&d_pcg_exit,
                             (void*)&config.pcg_max_iter, (void*)&config.pcg_exit_tol};
    size_t ppcg_kernel_smem_size = pcgSharedMemSize<T>(state_size, knot_points);

    // ---- include/pcg/sqp.cuh:207-259 ----
    form_schur_system<T>(state_size, control_size, knot_points, d_G_dense, d_C_dense, d_g, d_c, d_S, d_Pinv, d_gamma, rho);
    gpuErrchk(hipPeekAtLastError());
    if (use_direct) {                                                      // the LINSYS_SOLVE == 0 twin, on the GPU
#ifdef USE_DOUBLES
        fprintf(stderr, "--direct: mpcg_block_solve is single precision\n");
        return 2;
#else
        block_solve_schur<T>(state_size, knot_points, d_S, d_gamma, d_lambda);
        pcg_iters = 0;
        pcg_exit = false;
#endif
    } else {
        gpuErrchk(mpcgLaunchPcg(pcg_kernel, knot_points, PCG_NUM_THREADS, pcgKernelArgs, ppcg_kernel_smem_size));
        gpuErrchk(hipMemcpy(&pcg_iters, d_pcg_iters, sizeof(uint32_t), hipMemcpyDeviceToHost));
        gpuErrchk(hipMemcpy(&pcg_exit, d_pcg_exit, sizeof(bool), hipMemcpyDeviceToHost));
    }
    compute_dz(state_size, control_size, knot_points, d_Ginv_dense, d_C_dense, d_g, d_lambda, d_dz);
    gpuErrchk(hipDeviceSynchronize());

    std::vector<T> dz(gsz), lam(n * N);
    gpuErrchk(hipMemcpy(dz.data(), d_dz, gsz * sizeof(T), hipMemcpyDeviceToHost));
    gpuErrchk(hipMemcpy(lam.data(), d_lambda, n * N * sizeof(T), hipMemcpyDeviceToHost));

    // CPU check (double): constraint rows  C dz = c  with  C = [I at (k,x_k)] + [Abar,Bbar at (k, x_{k-1},u_{k-1})]
    double cerr = 0, serr = 0, cmax = 1e-30, gmax = 1e-30;
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i) {
            double acc = dz[(size_t)k * (n + m) + i];
            if (k > 0) {
                const T* A = &C[(size_t)(k - 1) * (nn + nm)];
                for (int j = 0; j < n; ++j) acc += (double)A[i + j * n] * dz[(size_t)(k - 1) * (n + m) + j];
                for (int j = 0; j < m; ++j) acc += (double)A[nn + i + j * n] * dz[(size_t)(k - 1) * (n + m) + n + j];
            }
            cerr = fmax(cerr, fabs(acc - c[k * n + i]));
            cmax = fmax(cmax, fabs(c[k * n + i]));
        }
    // stationarity rows  (G + rho I) dz + C^T lam = g   (state rows of knot k)
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i) {
            const T* Q = &Graw[(size_t)k * (nn + mm)];
            double acc = 0;
            for (int j = 0; j < n; ++j) acc += ((double)Q[i + j * n] + (i == j ? rho : 0.0)) * dz[(size_t)k * (n + m) + j];
            acc += lam[k * n + i];
            if (k < N - 1) {
                const T* A = &C[(size_t)k * (nn + nm)];
                for (int t = 0; t < n; ++t) acc += (double)A[t + i * n] * lam[(k + 1) * n + t];
            }
            serr = fmax(serr, fabs(acc - g[(size_t)k * (n + m) + i]));
            gmax = fmax(gmax, fabs(g[(size_t)k * (n + m) + i]));
        }
    printf("{\"pcg_iters\": %u, \"pcg_exit\": %d, \"constraint_err\": %.3e, \"stationarity_err\": %.3e}\n", pcg_iters, (int)pcg_exit,
           cerr / cmax, serr / gmax);
    const double tol = sizeof(T) == 8 ? 1e-9 : 1e-3;
    return (pcg_exit == false && cerr / cmax < tol && serr / gmax < tol) ? 0 : 1;
}
