/* mpcg_oracle.h — CPU oracle (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see mpcg_oracle.c). */
#ifndef MPCG_ORACLE_H
#define MPCG_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DECL(REAL, S) \
void orc_store_block_bd_##S(int n, int N, const REAL *src, REAL *dst, int col, int blockrow, REAL mult); \
void orc_load_block_bd_##S(int n, int N, const REAL *src, REAL *dst, int col, int blockrow, int transpose); \
void orc_invert_##S(int n, REAL *A, REAL *Ainv); \
int  orc_form_schur_##S(int n, int m, int N, REAL *G, const REAL *C, const REAL *g, const REAL *c, \
                        REAL *S_, REAL *Pinv, REAL *gamma, REAL rho, int ss); \
void orc_bt_spmv_##S(int n, int N, const REAL *M, const REAL *x, REAL *y, int cols); \
int  orc_pcg_##S(int n, int N, const REAL *S_, const REAL *Pinv, const REAL *gamma, REAL *lambda, \
                 REAL *r, REAL *p, int max_iter, REAL exit_tol, int precond_cols, \
                 uint32_t *iters_out, uint8_t *max_iter_exit_out, REAL *eta_hist); \
void orc_compute_dz_##S(int n, int m, int N, const REAL *Ginv, const REAL *C, const REAL *g, \
                        const REAL *lambda, REAL *dz); \
void orc_bd_to_csr_lowertri_##S(int n, int N, const REAL *S_, REAL *val, REAL mult); \
int  orc_ldl_factor_##S(int n, const int *Ap, const int *Ai, const REAL *Ax, int *Lp, int *Li, REAL *Lx, \
                        REAL *D, REAL *Dinv, const int *Lnz, const int *etree, \
                        unsigned char *bwork, int *iwork, REAL *fwork); \
void orc_ldl_solve_##S(int n, const int *Lp, const int *Li, const REAL *Lx, const REAL *Dinv, REAL *x); \
int  orc_bt_block_solve_##S(int n, int N, const REAL *S_, const REAL *gamma, REAL *lambda, REAL *work); \
int  orc_ldl_solve_schur_##S(int An, const int *Ap, const int *Ai, const REAL *Ax, const REAL *b, REAL *x, \
                             int *Lp, int *Li, REAL *Lx, REAL *D, REAL *Dinv, const int *Lnz, \
                             const int *etree, unsigned char *bwork, int *iwork, REAL *fwork);

ORC_DECL(float, f32)
ORC_DECL(double, f64)
#undef ORC_DECL

int  orc_ldl_etree(int n, const int *Ap, const int *Ai, int *work, int *Lnz, int *etree);
void orc_prep_csr(int n, int N, int *col_ptr, int *row_ind);
int  orc_bt_direct_solve_f64(int n, int N, const double *S, const double *b, double *x);
long orc_ldl_throughput_f32(int An, const int *Ap, const int *Ai, const int *Lnz, const int *etree, int sumLnz,
                            const float *vals, const float *bs, int ns, int nthreads, double seconds,
                            double *elapsed_out);

#ifdef __cplusplus
}
#endif
#endif
