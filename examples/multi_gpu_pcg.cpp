// multi_gpu_pcg.cpp — the NATIVE multi-device driver of the hot path ("one host thread + stream + handle per device",
// batch-partitioned, RCCL only to gather results): plain C++ over the C ABI of libmpcg_hip.so, no Python, no torch.
//
//   ./multi_gpu_pcg [--gpus G] [--batch B] [--knots N] [--steps K] [--warmup W] [--strong]
//
// G defaults to every visible device; it is an error to ask for more than there are.  Trajectories are independent units, so the path
// has NO data-path collective: device d gets a contiguous shard of the batch (weak scaling: B trajectories per device; --strong: B in
// total, shards differ by at most one), builds its Schur systems on the device from host-generated IIWA-shaped KKT blocks
// (mpcg_form_schur = the reference's form_schur_system) and solves them K times; one host thread per device, each with its own stream and
// handle.  Timed region per step, as the reference times a linsolve (include/pcg/sqp.cuh:224-241): lambda <- 0, mpcg_pcg_solve, D2H of
// (iters, exit) — between a start line all threads cross together and the join; value = iterations of all devices / the slowest device's
// time.  Afterwards ONE RCCL collective in a single-process communicator (ncclCommInitAll): all-gather of the per-trajectory iteration
// counts, checked on every device against the sum of the shards.  Prints one JSON line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/mpcg.h"

#define CHECK_HIP(x)                                                                                         \
    do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define CHECK_NCCL(x)                                                                                        \
    do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(e_)); exit(1); } } while (0)
#define CHECK_MPCG(h, x)                                                                                     \
    do { int e_ = (x); if (e_ != MPCG_OK) { fprintf(stderr, "%s: %d %s\n", #x, e_, mpcg_last_error(h)); exit(1); } } while (0)

static const int n = 14, m = 7, nn = n * n, mm = m * m, nm = n * m;

// splitmix64 -> uniform / normal: seeded per trajectory, so a trajectory's system does not depend on which device gets it
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    double normal() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(6.283185307179586 * uni()); }
};

// IIWA-shaped KKT blocks of one trajectory in the reference's dense layouts (SURVEY.md §8d "synthetic inputs": rank-one Gauss-Newton block
// in q + QD_COST on qd, R_COST on u, A = I + dt [[0, I], [dqdd/dq, dqdd/dqd]], B = dt [0; Minv], small defects; C stores -A, -B)
static void make_kkt(int N, uint64_t seed, float* G, float* C, float* g, float* c) {
    Rng r(seed);
    const double dt = 1.0 / 64, QD = 1e-4, RC = 1e-4;
    for (int k = 0; k < N; ++k) {
        float* Q = G + (size_t)k * (nn + mm);
        double gq[7];
        for (int i = 0; i < 7; ++i) gq[i] = 0.3 * r.normal();
        for (int e = 0; e < nn; ++e) Q[e] = 0.f;
        for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) Q[i + j * n] = (float)(gq[i] * gq[j]);
        for (int i = 7; i < n; ++i) Q[i + i * n] = (float)QD;
        float* gk = g + (size_t)k * (n + m);
        for (int i = 0; i < n; ++i) gk[i] = (float)(0.1 * r.normal());
        if (k < N - 1) {
            float* R = Q + nn;
            for (int e = 0; e < mm; ++e) R[e] = 0.f;
            for (int i = 0; i < m; ++i) R[i + i * m] = (float)RC;
            for (int i = 0; i < m; ++i) gk[n + i] = (float)(0.01 * r.normal());
            float* A = C + (size_t)k * (nn + nm);
            for (int col = 0; col < n; ++col)
                for (int row = 0; row < n; ++row) {
                    double a = row == col ? 1.0 : 0.0;
                    if (row < 7) a += (col == row + 7) ? dt : 0.0;
                    else a += dt * 3.0 * r.normal();
                    A[row + col * n] = (float)(-a);
                }
            double W[7][7];
            for (auto& row : W) for (double& v : row) v = r.normal();
            for (int col = 0; col < m; ++col)
                for (int row = 0; row < n; ++row) {
                    double bv = 0.0;
                    if (row >= 7) { for (int t = 0; t < 7; ++t) bv += W[row - 7][t] * W[col][t]; bv = dt * (bv + (row - 7 == col ? 1.0 : 0.0)); }
                    A[nn + row + col * n] = (float)(-bv);
                }
        }
        for (int i = 0; i < n; ++i) c[(size_t)k * n + i] = k == 0 ? 0.f : (float)(1e-2 * r.normal());
    }
}

struct DeviceResult { double ms = 0; unsigned long long iters = 0, gathered = 0; };

int main(int argc, char** argv) {
    int G = 0, B = 1024, N = 128, steps = 10, warmup = 2;
    bool strong = false;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](int& dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (!strcmp(argv[i], "--gpus")) val(G);
        else if (!strcmp(argv[i], "--batch")) val(B);
        else if (!strcmp(argv[i], "--knots")) val(N);
        else if (!strcmp(argv[i], "--steps")) val(steps);
        else if (!strcmp(argv[i], "--warmup")) val(warmup);
        else if (!strcmp(argv[i], "--strong")) strong = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    if (G == 0) G = ndev;
    if (G < 1 || G > ndev) { fprintf(stderr, "multi_gpu_pcg: --gpus %d requested, %d device(s) visible: refusing to run fewer\n", G, ndev); return 2; }
    const int total = strong ? B : B * G;
    if (total < G) { fprintf(stderr, "more devices than trajectories\n"); return 2; }
    auto shard = [&](int d, int& lo, int& hi) { const int base = total / G, rem = total % G; lo = d * base + (d < rem ? d : rem); hi = lo + base + (d < rem ? 1 : 0); };
    int cap = 0;
    for (int d = 0; d < G; ++d) { int lo, hi; shard(d, lo, hi); cap = hi - lo > cap ? hi - lo : cap; }
    const uint32_t max_iter = N == 32 ? 173 : N == 64 ? 167 : N == 128 ? 167 : N == 256 ? 118 : N == 512 ? 67 : 200;   // include/common/settings.cuh:123-139
    const float tol = 1e-4f, rho = 1e-3f;

    std::vector<int> devs(G);
    for (int d = 0; d < G; ++d) devs[d] = d;
    std::vector<ncclComm_t> comms(G);
    CHECK_NCCL(ncclCommInitAll(comms.data(), G, devs.data()));     // single process, one communicator rank per device (xGMI between them)

    std::vector<DeviceResult> res(G);
    std::atomic<int> ready{0}, go{0};
    auto worker = [&](int d) {
        int lo, hi;
        shard(d, lo, hi);
        const int nb = hi - lo;
        CHECK_HIP(hipSetDevice(d));
        hipStream_t st;
        CHECK_HIP(hipStreamCreate(&st));
        mpcg_handle* h = nullptr;
        if (mpcg_create(&h, d, n, (uint32_t)N, (uint32_t)nb) != MPCG_OK) { fprintf(stderr, "mpcg_create: %s\n", mpcg_last_error(nullptr)); exit(1); }
        const size_t Gsz = (size_t)(nn + mm) * N - mm, Csz = (size_t)(nn + nm) * (N - 1), gsz = (size_t)(n + m) * N - m, csz = (size_t)n * N, Ssz = (size_t)3 * nn * N;
        float *dG, *dC, *dg, *dc, *dS, *dP, *dgam, *dlam;
        uint32_t *dit, *dgather;
        uint8_t* dex;
        CHECK_HIP(hipMalloc(&dG, nb * Gsz * 4)); CHECK_HIP(hipMalloc(&dC, nb * Csz * 4)); CHECK_HIP(hipMalloc(&dg, nb * gsz * 4)); CHECK_HIP(hipMalloc(&dc, nb * csz * 4));
        CHECK_HIP(hipMalloc(&dS, nb * Ssz * 4)); CHECK_HIP(hipMalloc(&dP, nb * Ssz * 4)); CHECK_HIP(hipMalloc(&dgam, nb * csz * 4)); CHECK_HIP(hipMalloc(&dlam, nb * csz * 4));
        CHECK_HIP(hipMalloc(&dit, (size_t)cap * 4)); CHECK_HIP(hipMalloc(&dex, nb)); CHECK_HIP(hipMalloc(&dgather, (size_t)cap * G * 4));
        CHECK_HIP(hipMemsetAsync(dit, 0, (size_t)cap * 4, st));
        {
            std::vector<float> hG(Gsz), hC(Csz), hg(gsz), hc(csz);
            for (int b = 0; b < nb; ++b) {       // inputs: generated on the host, outside every timed region
                make_kkt(N, 1000 + (uint64_t)(lo + b), hG.data(), hC.data(), hg.data(), hc.data());
                CHECK_HIP(hipMemcpyAsync(dG + b * Gsz, hG.data(), Gsz * 4, hipMemcpyHostToDevice, st));
                CHECK_HIP(hipMemcpyAsync(dC + b * Csz, hC.data(), Csz * 4, hipMemcpyHostToDevice, st));
                CHECK_HIP(hipMemcpyAsync(dg + b * gsz, hg.data(), gsz * 4, hipMemcpyHostToDevice, st));
                CHECK_HIP(hipMemcpyAsync(dc + b * csz, hc.data(), csz * 4, hipMemcpyHostToDevice, st));
                CHECK_HIP(hipStreamSynchronize(st));
            }
        }
        CHECK_MPCG(h, mpcg_form_schur(h, m, dG, dC, dg, dc, dS, dP, dgam, rho, (uint32_t)nb, MPCG_PRECOND_SS, st));
        std::vector<uint32_t> hit(nb);
        std::vector<uint8_t> hex(nb);
        auto step = [&]() {
            CHECK_HIP(hipMemsetAsync(dlam, 0, nb * csz * 4, st));
            CHECK_MPCG(h, mpcg_pcg_solve(h, dS, dP, dgam, dlam, (uint32_t)nb, max_iter, tol, MPCG_PRECOND_SS, dit, dex, st));
            CHECK_HIP(hipMemcpyAsync(hit.data(), dit, nb * 4, hipMemcpyDeviceToHost, st));
            CHECK_HIP(hipMemcpyAsync(hex.data(), dex, nb, hipMemcpyDeviceToHost, st));
        };
        for (int w = 0; w < warmup; ++w) step();
        CHECK_HIP(hipStreamSynchronize(st));
        ready.fetch_add(1);
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();     // start line: every device begins the timed region together
        const auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) step();
        CHECK_HIP(hipStreamSynchronize(st));
        res[d].ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t v : hit) res[d].iters += v;
        // the path's only collective: per-trajectory iteration counts of every shard to every device (shards padded to the largest)
        CHECK_NCCL(ncclAllGather(dit, dgather, (size_t)cap, ncclUint32, comms[d], st));
        std::vector<uint32_t> all((size_t)cap * G);
        CHECK_HIP(hipMemcpyAsync(all.data(), dgather, all.size() * 4, hipMemcpyDeviceToHost, st));
        CHECK_HIP(hipStreamSynchronize(st));
        unsigned long long sum = 0;
        for (int e = 0; e < G; ++e) { int l2, h2; shard(e, l2, h2); for (int b = 0; b < h2 - l2; ++b) sum += all[(size_t)e * cap + b]; }
        res[d].gathered = sum;              // (checked against the sum of the shards below)
        for (void* p : {(void*)dG, (void*)dC, (void*)dg, (void*)dc, (void*)dS, (void*)dP, (void*)dgam, (void*)dlam, (void*)dit, (void*)dex, (void*)dgather}) (void)hipFree(p);
        mpcg_destroy(h);
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int d = 0; d < G; ++d) th.emplace_back(worker, d);
    while (ready.load() < G) std::this_thread::yield();
    go.store(1, std::memory_order_release);
    for (auto& t : th) t.join();
    for (auto& c_ : comms) ncclCommDestroy(c_);

    double slowest = 0;
    unsigned long long iters = 0;
    for (auto& r : res) { slowest = r.ms > slowest ? r.ms : slowest; iters += r.iters; }
    bool consistent = true;
    for (auto& r : res) consistent = consistent && r.gathered == iters;
    printf("{\"metric\": \"pcg_iterations_per_sec\", \"value\": %.1f, \"unit\": \"iter/s\", \"n_gpus\": %d, \"host\": \"C++ threads, one per device (examples/multi_gpu_pcg.cpp)\", "
           "\"scaling\": \"%s\", \"knot_points\": %d, \"global_batch\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"per_device_ms_per_step\": [",
           (double)iters / (slowest / steps * 1e-3), G, strong ? "strong" : "weak", N, total, steps, warmup, slowest / steps);
    for (int d = 0; d < G; ++d) printf("%s%.4f", d ? ", " : "", res[d].ms / steps);
    printf("], \"mean_pcg_iters\": %.2f, \"results_gather\": {\"collective\": \"ncclAllGather (RCCL, single-process communicator)\", \"trajectories\": %d, "
           "\"consistent_with_shard_sums\": %s}}\n", (double)iters / total, total, consistent ? "true" : "false");
    return consistent ? 0 : 1;
}
