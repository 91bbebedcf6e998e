/* mpcg_oracle.c — CPU oracle for the MPCGPU PCG hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (mpcgpu_amd/, include/mpcg.h) never links or calls it.
 *
 * PARITY UNPINNED — see mpcg_oracle_impl.inc header and DESIGN.md §Oracle.
 * Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "mpcg_oracle.h"

#define REAL float
#define SUF(x) x##_f32
#include "mpcg_oracle_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF(x) x##_f64
#include "mpcg_oracle_impl.inc"
#undef REAL
#undef SUF

/* CSR pattern of the lower triangle of a symmetric block-tridiagonal matrix,
 * include/utils/csr.cuh:40-73 (prep_csr): row (k,i) has (k>0)*n + i+1 entries, first column index
 * (k>0)*(k-1)*n.  nnz = (N-1)n^2 + N n(n+1)/2 (include/qdldl/sqp.cuh:148). */
void orc_prep_csr(int n, int N, int *col_ptr, int *row_ind)
{
    int o = 0;
    col_ptr[0] = 0;
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i) {
            int len = (k > 0) * n + i + 1;
            int first = (k > 0) * (k - 1) * n;
            for (int c = 0; c < len; ++c) row_ind[o++] = first + c;
            col_ptr[k * n + i + 1] = o;
        }
}

/* Ground truth: fp64 direct solve of S x = b for block-tridiagonal S (bd layout, negated storage,
 * i.e. -S is SPD).  Block LDL^T sweep with a dense Cholesky of every pivot block.
 * Returns 0, or -(k+1) if pivot block k of -S is not positive definite. */
static int chol_f64(int n, double *A)          /* in place, lower, column-major */
{
    for (int j = 0; j < n; ++j) {
        double d = A[j + j * n];
        for (int t = 0; t < j; ++t) d -= A[j + t * n] * A[j + t * n];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[j + j * n] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i + j * n];
            for (int t = 0; t < j; ++t) s -= A[i + t * n] * A[j + t * n];
            A[i + j * n] = s / d;
        }
    }
    return 0;
}
static void chol_solve_f64(int n, const double *L, double *x, int nrhs)   /* x: n x nrhs col-major */
{
    for (int c = 0; c < nrhs; ++c) {
        double *v = x + (size_t)c * n;
        for (int i = 0; i < n; ++i) {
            double s = v[i];
            for (int t = 0; t < i; ++t) s -= L[i + t * n] * v[t];
            v[i] = s / L[i + i * n];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = v[i];
            for (int t = i + 1; t < n; ++t) s -= L[t + i * n] * v[t];
            v[i] = s / L[i + i * n];
        }
    }
}

int orc_bt_direct_solve_f64(int n, int N, const double *S, const double *b, double *x)
{
    const int nn = n * n;
    /* work on M = -S (SPD) and rhs = -b */
    double *Dk = (double *)malloc(sizeof(double) * ((size_t)N * nn * 2 + (size_t)N * n + 2 * nn));
    if (!Dk) return -1000000;
    double *W = Dk + (size_t)N * nn;      /* W_k = Dk^-1 * U_k  (U_k = M[k,k+1]) */
    double *y = W + (size_t)N * nn;
    double *T = y + (size_t)N * n, *T2 = T + nn;
    int rc = 0;
    for (int k = 0; k < N && !rc; ++k) {
        double *D = Dk + (size_t)k * nn;
        for (int e = 0; e < nn; ++e) D[e] = -S[(size_t)k * 3 * nn + nn + e];
        for (int i = 0; i < n; ++i) y[(size_t)k * n + i] = -b[(size_t)k * n + i];
        if (k > 0) {
            /* L_k = M[k,k-1];  D_k -= L_k W_{k-1};  y_k -= L_k z_{k-1}, z = D^-1 y (stored in x) */
            const double *Lk = S + (size_t)k * 3 * nn;          /* stored = -M[k,k-1] */
            const double *Wm = W + (size_t)(k - 1) * nn;
            for (int c = 0; c < n; ++c)
                for (int r = 0; r < n; ++r) {
                    double acc = 0;
                    for (int t = 0; t < n; ++t) acc += (-Lk[r + t * n]) * Wm[t + c * n];
                    D[r + c * n] -= acc;
                }
            for (int r = 0; r < n; ++r) {
                double acc = 0;
                for (int t = 0; t < n; ++t) acc += (-Lk[r + t * n]) * x[(size_t)(k - 1) * n + t];
                y[(size_t)k * n + r] -= acc;
            }
        }
        /* symmetrise against storage noise, factor */
        for (int c = 0; c < n; ++c)
            for (int r = c + 1; r < n; ++r) {
                double a = 0.5 * (D[r + c * n] + D[c + r * n]);
                D[r + c * n] = a; D[c + r * n] = a;
            }
        if (chol_f64(n, D)) { rc = -(k + 1); break; }
        for (int i = 0; i < n; ++i) x[(size_t)k * n + i] = y[(size_t)k * n + i];
        chol_solve_f64(n, D, x + (size_t)k * n, 1);             /* z_k = D_k^-1 y_k */
        if (k < N - 1) {
            double *Wk = W + (size_t)k * nn;
            for (int e = 0; e < nn; ++e) Wk[e] = -S[(size_t)k * 3 * nn + 2 * nn + e];
            chol_solve_f64(n, D, Wk, n);                        /* W_k = D_k^-1 U_k */
        }
    }
    if (!rc)
        for (int k = N - 2; k >= 0; --k) {                      /* x_k = z_k - W_k x_{k+1} */
            const double *Wk = W + (size_t)k * nn;
            for (int r = 0; r < n; ++r) {
                double acc = 0;
                for (int t = 0; t < n; ++t) acc += Wk[r + t * n] * x[(size_t)(k + 1) * n + t];
                x[(size_t)k * n + r] -= acc;
            }
        }
    (void)T; (void)T2;
    free(Dk);
    return rc;
}

/* Batch-parallel timing harness for bench.py's cpu_baseline "all_cores" leg: `nthreads`
 * POSIX threads, each with its own QDLDL-style workspace, run numeric factor + solve (the per-linsolve work of
 * include/qdldl/sqp.cuh:22-49) over the `ns` given systems round-robin for `seconds`.  Returns the number of
 * completed solves (negative on a zero pivot / allocation failure).  Same arithmetic as
 * orc_ldl_solve_schur_f32 — this only adds the threads. */
#include <pthread.h>
#include <time.h>

typedef struct {
    int An, nnz, ns, tid, nthreads, sumLnz;
    const int *Ap, *Ai, *Lnz, *etree;
    const float *vals, *bs;
    double seconds;
    long done;
} orc_mt_job;

static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *orc_mt_worker(void *arg)
{
    orc_mt_job *j = (orc_mt_job *)arg;
    const int An = j->An;
    int *Lp = malloc(sizeof(int) * (size_t)(An + 1)), *Li = malloc(sizeof(int) * (size_t)(j->sumLnz + 1));
    int *iwork = malloc(sizeof(int) * 3 * (size_t)An);
    float *Lx = malloc(sizeof(float) * (size_t)(j->sumLnz + 1)), *D = malloc(sizeof(float) * (size_t)An);
    float *Dinv = malloc(sizeof(float) * (size_t)An), *fwork = malloc(sizeof(float) * (size_t)An);
    float *x = malloc(sizeof(float) * (size_t)An);
    unsigned char *bwork = malloc((size_t)An);
    j->done = -1;
    if (Lp && Li && iwork && Lx && D && Dinv && fwork && x && bwork) {
        long cnt = 0;
        int b = j->tid % j->ns;
        const double t_end = orc_now() + j->seconds;
        while (orc_now() < t_end) {
            int rc = orc_ldl_solve_schur_f32(An, j->Ap, j->Ai, j->vals + (size_t)b * j->nnz, j->bs + (size_t)b * An, x,
                                             Lp, Li, Lx, D, Dinv, j->Lnz, j->etree, bwork, iwork, fwork);
            if (rc < 0) { cnt = -1; break; }
            ++cnt;
            b = (b + j->nthreads) % j->ns;
        }
        j->done = cnt;
    }
    free(Lp); free(Li); free(iwork); free(Lx); free(D); free(Dinv); free(fwork); free(x); free(bwork);
    return NULL;
}

long orc_ldl_throughput_f32(int An, const int *Ap, const int *Ai, const int *Lnz, const int *etree, int sumLnz,
                            const float *vals, const float *bs, int ns, int nthreads, double seconds,
                            double *elapsed_out)
{
    if (nthreads < 1 || ns < 1) return -1;
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nthreads);
    orc_mt_job *jobs = malloc(sizeof(orc_mt_job) * (size_t)nthreads);
    if (!th || !jobs) { free(th); free(jobs); return -1; }
    const double t0 = orc_now();
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        orc_mt_job j = { An, Ap[An], ns, t, nthreads, sumLnz, Ap, Ai, Lnz, etree, vals, bs, seconds, 0 };
        jobs[t] = j;
        if (pthread_create(&th[t], NULL, orc_mt_worker, &jobs[t]) != 0) break;
        ++started;
    }
    long total = 0;
    for (int t = 0; t < started; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].done < 0) total = -1;
        else if (total >= 0) total += jobs[t].done;
    }
    if (elapsed_out) *elapsed_out = orc_now() - t0;
    free(th); free(jobs);
    return started == nthreads ? total : -1;
}
