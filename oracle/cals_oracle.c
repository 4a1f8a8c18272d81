/*
 * cals_oracle.c -- CPU restatement (plain C) of the HPAC/CP-CALS concurrent-ALS hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cals_oracle.h for who may use it and for the parity status:
 * pinned by the reference's own restated test-suite, "parity unpinned" against reference-run
 * outputs because the reference cannot be built in this image).
 *
 * All matrices are column-major doubles, exactly as in the reference (include/matrix.h:9-23).
 * Citations are file:line relative to /root/reference.
 */
#include "cals_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *xmalloc(size_t n) {
  void *p = NULL;
  if (n == 0) n = 8;
  if (posix_memalign(&p, 64, n) != 0 || !p) { /* include/tensor.h: 64-byte aligned buffers */
    fprintf(stderr, "cals_oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}

static or_dgemm_fn g_dgemm = NULL;
void or_set_dgemm(or_dgemm_fn fn) { g_dgemm = fn; }

void or_default_params(or_params *p) {
  /* include/cals.h:138-159 defaults */
  p->max_iterations = 200;
  p->tol = 1e-7;
  p->buffer_size = 4200;
  p->mttkrp_method = OR_AUTO;
  p->line_search = 0;
  p->line_search_interval = 5;
  p->line_search_step = 0.0;
  p->line_search_method = OR_LS_NO_ERROR_CHECKING;
  p->force_max_iter = 0;
  p->always_evict_first = 0;
  p->threads = 1;
  p->update_method = OR_UPDATE_UNCONSTRAINED;
}

/* ------------------------------------------------------------------------------------------ */
/* BLAS / LAPACK level restatements (Netlib reference algorithms)                              */
/* ------------------------------------------------------------------------------------------ */

/* cblas_dnrm2 as used by Tensor::norm (include/tensor.h:196) and Ktensor::normalize
 * (src/ktensor.cpp:73,92): plain sqrt of the sum of squares (inputs are O(1), no scaling). */
double or_norm(const double *x, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) s += x[i] * x[i];
  return sqrt(s);
}

/* cblas_idamax (Netlib idamax.f): first index of the largest |x[i]|. src/ktensor.cpp:75 */
static int64_t idamax_ref(const double *x, int64_t n) {
  if (n < 1) return -1;
  int64_t best = 0;
  double dmax = fabs(x[0]);
  for (int64_t i = 1; i < n; i++)
    if (fabs(x[i]) > dmax) {
      best = i;
      dmax = fabs(x[i]);
    }
  return best;
}

/* C(MxN) = op(A)(MxK) * B(KxN) + beta*C, col-major; transA: 0 N, 1 T (Netlib dgemm.f loop order). */
static void dgemm_ref(int transA, int64_t M, int64_t N, int64_t K, const double *A, int64_t lda,
                      const double *B, int64_t ldb, double beta, double *C, int64_t ldc) {
  if (g_dgemm && M > 0 && N > 0 && K > 0 && M < INT32_MAX && N < INT32_MAX && K < INT32_MAX &&
      lda < INT32_MAX && ldb < INT32_MAX) {
    g_dgemm(102, transA ? 112 : 111, 111, (int)M, (int)N, (int)K, 1.0, A, (int)lda, B, (int)ldb,
            beta, C, (int)ldc);
    return;
  }
#pragma omp parallel for schedule(static) if (M * N * K > 200000)
  for (int64_t j = 0; j < N; j++) {
    double *c = C + j * ldc;
    if (beta == 0.0)
      for (int64_t i = 0; i < M; i++) c[i] = 0.0;
    else if (beta != 1.0)
      for (int64_t i = 0; i < M; i++) c[i] *= beta;
    if (!transA) {
      for (int64_t l = 0; l < K; l++) {
        const double t = B[l + j * ldb];
        const double *a = A + l * lda;
        for (int64_t i = 0; i < M; i++) c[i] += t * a[i];
      }
    } else {
      for (int64_t i = 0; i < M; i++) {
        const double *a = A + i * lda;
        double t = 0.0;
        for (int64_t l = 0; l < K; l++) t += a[l] * B[l + j * ldb];
        c[i] += t;
      }
    }
  }
}

/* y = op(A) x, beta = 0 (Netlib dgemv.f). trans: 0 N (y has M), 1 T (y has N). */
static void dgemv_ref(int trans, int64_t M, int64_t N, const double *A, int64_t lda,
                      const double *x, double *y) {
  if (!trans) {
    for (int64_t i = 0; i < M; i++) y[i] = 0.0;
    for (int64_t j = 0; j < N; j++) {
      const double t = x[j];
      const double *a = A + j * lda;
      for (int64_t i = 0; i < M; i++) y[i] += t * a[i];
    }
  } else {
    for (int64_t j = 0; j < N; j++) {
      const double *a = A + j * lda;
      double t = 0.0;
      for (int64_t i = 0; i < M; i++) t += a[i] * x[i];
      y[j] = t;
    }
  }
}

/* dpotrf("L") restated as the unblocked Netlib dpotf2.f (lower): returns info (0 ok, j>0 =>
 * leading minor j not positive definite; factorisation stops there). src/utils/update.cpp:183 */
static int dpotf2_lower_ref(double *A, int64_t n, int64_t lda) {
  for (int64_t j = 0; j < n; j++) {
    double ajj = A[j + j * lda];
    for (int64_t k = 0; k < j; k++) ajj -= A[j + k * lda] * A[j + k * lda];
    if (ajj <= 0.0 || isnan(ajj)) {
      A[j + j * lda] = ajj;
      return (int)(j + 1);
    }
    ajj = sqrt(ajj);
    A[j + j * lda] = ajj;
    for (int64_t i = j + 1; i < n; i++) {
      double s = A[i + j * lda];
      for (int64_t k = 0; k < j; k++) s -= A[i + k * lda] * A[j + k * lda];
      A[i + j * lda] = s / ajj;
    }
  }
  return 0;
}

/* cblas_dtrsm(Right, Lower, Trans, NonUnit): B := B * inv(L^T)  (Netlib dtrsm.f, "Form
 * B := alpha*B*inv( A**T )", lower branch).  src/utils/update.cpp:187 */
static void dtrsm_RLT_ref(int64_t M, int64_t N, const double *L, int64_t ldl, double *B,
                          int64_t ldb) {
  for (int64_t k = 0; k < N; k++) {
    const double temp = 1.0 / L[k + k * ldl];
    double *bk = B + k * ldb;
    for (int64_t i = 0; i < M; i++) bk[i] = temp * bk[i];
    for (int64_t j = k + 1; j < N; j++) {
      const double ljk = L[j + k * ldl];
      if (ljk != 0.0) {
        double *bj = B + j * ldb;
        for (int64_t i = 0; i < M; i++) bj[i] -= ljk * bk[i];
      }
    }
  }
}

/* cblas_dtrsm(Right, Lower, NoTrans, NonUnit): B := B * inv(L)  (Netlib dtrsm.f, "Form
 * B := alpha*B*inv( A )", lower branch: j = n..1).  src/utils/update.cpp:189 */
static void dtrsm_RLN_ref(int64_t M, int64_t N, const double *L, int64_t ldl, double *B,
                          int64_t ldb) {
  for (int64_t j = N - 1; j >= 0; j--) {
    double *bj = B + j * ldb;
    for (int64_t k = j + 1; k < N; k++) {
      const double lkj = L[k + j * ldl];
      if (lkj != 0.0) {
        const double *bk = B + k * ldb;
        for (int64_t i = 0; i < M; i++) bj[i] -= lkj * bk[i];
      }
    }
    const double temp = 1.0 / L[j + j * ldl];
    for (int64_t i = 0; i < M; i++) bj[i] = temp * bj[i];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* L2 math kernels                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* khatri_rao(A, B, K): K[a*IB + b, c] = A[a,c] * B[b,c].  src/utils/mttkrp.cpp:78-103 */
void or_khatri_rao(const double *A, int64_t IA, const double *B, int64_t IB, int64_t cols,
                   double *K) {
#pragma omp parallel for schedule(static) if (IA * IB * cols > 100000)
  for (int64_t c = 0; c < cols; c++) {
    double *kc = K + c * (IA * IB);
    for (int64_t a = 0; a < IA; a++) {
      const double f = A[a + IA * c];
      double *kp = kc + a * IB;
      const double *b = B + IB * c;
      for (int64_t r = 0; r < IB; r++) kp[r] = f * b[r];
    }
  }
}

/* Unfolding, src/tensor.cpp:143-180 */
typedef struct {
  int64_t n_blocks, block_offset, rows, cols, stride;
} unfolding_t;

static unfolding_t implicit_unfold(int n_modes, const int64_t *modes, int mode) {
  unfolding_t u;
  int64_t before = 1, after = 1;
  for (int n = 0; n < mode; n++) before *= modes[n];
  for (int n = mode + 1; n < n_modes; n++) after *= modes[n];
  if (mode == 0) {
    u.n_blocks = 1; u.block_offset = 0; u.rows = modes[mode]; u.cols = after; u.stride = modes[mode];
  } else if (mode == n_modes - 1) {
    u.n_blocks = 1; u.block_offset = 0; u.rows = modes[mode]; u.cols = before; u.stride = before;
  } else {
    u.n_blocks = after; u.block_offset = before * modes[mode]; u.rows = modes[mode];
    u.cols = before; u.stride = before;
  }
  return u;
}

/* mttkrp_impl: explicit KRP chain + block GEMMs.  src/utils/mttkrp.cpp:179-216 (KRP order:
 * push modes != mode in increasing order, pop => K = A_last (.) ... (.) A_first, first fastest)
 * and :218-328 (GEMM over the implicit unfolding; mode 0: N,N beta 0; else T,N beta 1 on zeroed G) */
static void mttkrp_impl(const double *X, int n_modes, const int64_t *modes,
                        double *const *factors, int64_t R, int mode, double *G) {
  /* remaining modes, descending (top of the reference's stack first) */
  int rem[OR_MAX_MODES], nrem = 0;
  for (int i = n_modes - 1; i >= 0; i--)
    if (i != mode) rem[nrem++] = i;
  int64_t rows = modes[rem[0]] * modes[rem[1]];
  double *krp = (double *)xmalloc(sizeof(double) * (size_t)(rows * R));
  or_khatri_rao(factors[rem[0]], modes[rem[0]], factors[rem[1]], modes[rem[1]], R, krp);
  for (int t = 2; t < nrem; t++) { /* khatri_rao_rec, :147-176 */
    int64_t nrows = rows * modes[rem[t]];
    double *next = (double *)xmalloc(sizeof(double) * (size_t)(nrows * R));
    or_khatri_rao(krp, rows, factors[rem[t]], modes[rem[t]], R, next);
    free(krp);
    krp = next;
    rows = nrows;
  }
  unfolding_t u = implicit_unfold(n_modes, modes, mode);
  const int64_t Gr = modes[mode];
  if (mode != 0) memset(G, 0, sizeof(double) * (size_t)(Gr * R));
  for (int64_t b = 0; b < u.n_blocks; b++) {
    const double *Xb = X + b * u.block_offset;
    const double *Kb = krp + b * u.cols;
    if (mode == 0)
      dgemm_ref(0, Gr, R, u.cols, Xb, u.stride, Kb, rows, 0.0, G, Gr);
    else
      dgemm_ref(1, Gr, R, u.cols, Xb, u.stride, Kb, rows, 1.0, G, Gr);
  }
  free(krp);
}

/* mttkrp_twostep + mttkrp_twostep_impl, 3-way only.  Parameter table src/utils/mttkrp.cpp:450-560,
 * TTM + per-column GEMV :330-448 (CPU branches). */
static void mttkrp_twostep(const double *X, const int64_t *modes, double *const *factors,
                           int64_t R, int mode, int method, double *G) {
  const int64_t I = modes[0], J = modes[1], K = modes[2];
  int64_t inter_rows, stride, block_rows, n_blocks = 1, block_offset = 0;
  int gemm_trans, Bidx;
  int gemv_trans, xi, yi;
  int64_t A_rows, A_cols, gv_stride;
  const int ts0 = (method == OR_TWOSTEP0);
  if (mode == 2) {
    if (ts0) { /* (JK x I) * (I x R) */
      inter_rows = J * K; gemm_trans = 1; stride = I; block_rows = inter_rows; Bidx = 0;
      gemv_trans = 1; A_rows = J; A_cols = K; gv_stride = J; xi = 1; yi = 2;
    } else { /* (IK x J) * (J x R) */
      inter_rows = I * K; gemm_trans = 0; block_offset = I * J; stride = I; block_rows = I;
      n_blocks = K; Bidx = 1;
      gemv_trans = 1; A_rows = I; A_cols = K; gv_stride = I; xi = 0; yi = 2;
    }
  } else if (mode == 1) {
    if (ts0) { /* (IJ x K) * (K x R) */
      inter_rows = I * J; gemm_trans = 0; stride = I * J; block_rows = inter_rows; Bidx = 2;
      gemv_trans = 1; A_rows = I; A_cols = J; gv_stride = I; xi = 0; yi = 1;
    } else { /* (JK x I) * (I x R) */
      inter_rows = J * K; gemm_trans = 1; stride = I; block_rows = inter_rows; Bidx = 0;
      gemv_trans = 0; A_rows = J; A_cols = K; gv_stride = J; xi = 2; yi = 1;
    }
  } else {
    if (ts0) { /* (IJ x K) * (K x R) */
      inter_rows = I * J; gemm_trans = 0; stride = I * J; block_rows = inter_rows; Bidx = 2;
      gemv_trans = 0; A_rows = I; A_cols = J; gv_stride = I; xi = 1; yi = 0;
    } else { /* (IK x J) * (J x R) */
      inter_rows = I * K; gemm_trans = 0; block_offset = I * J; stride = I; block_rows = I;
      n_blocks = K; Bidx = 1;
      gemv_trans = 0; A_rows = I; A_cols = K; gv_stride = I; xi = 2; yi = 0;
    }
  }
  (void)yi;
  double *W = (double *)xmalloc(sizeof(double) * (size_t)(inter_rows * R));
  const double *Bf = factors[Bidx];
  const int64_t Brows = modes[Bidx];
  for (int64_t blk = 0; blk < n_blocks; blk++)
    dgemm_ref(gemm_trans, block_rows, R, Brows, X + blk * block_offset, stride, Bf, Brows, 0.0,
              W + blk * block_rows, inter_rows);
  const double *xf = factors[xi];
  const int64_t xrows = modes[xi], yrows = modes[mode];
#pragma omp parallel for schedule(static) if (R * A_rows * A_cols > 100000)
  for (int64_t c = 0; c < R; c++)
    dgemv_ref(gemv_trans, A_rows, A_cols, W + c * inter_rows, gv_stride, xf + c * xrows,
              G + c * yrows);
  free(W);
}

/* mttkrp::mttkrp dispatch, src/utils/mttkrp.cpp:562-614.  The LUT branch (:574-587) is not
 * restated: the LUT files live under the reference's data/ directory and only select among
 * mathematically identical variants; AUTO here is the "no LUT" heuristic (:588-607). */
static int resolve_method(int n_modes, int method, int mode, int threads) {
  if (n_modes != 3) return OR_MTTKRP;
  if (method != OR_AUTO) return method;
  if (threads != 1) return (mode == 1) ? OR_MTTKRP : OR_TWOSTEP1;
  return OR_TWOSTEP0;
}

void or_mttkrp(const double *X, int n_modes, const int64_t *modes, double *const *factors,
               int64_t R, int mode, int method, double *G) {
  if (n_modes != 3 || method == OR_MTTKRP)
    mttkrp_impl(X, n_modes, modes, factors, R, mode, G);
  else
    mttkrp_twostep(X, modes, factors, R, mode, method, G);
}

/* ops::hadamard_but_one, src/utils/utils.cpp:161-172 */
void or_hadamard_but_one(double *const *gramians, int n_modes, int64_t r, int mode) {
  double *H = gramians[mode];
  for (int64_t i = 0; i < r * r; i++) H[i] = 1.0;
  for (int m = 0; m < n_modes; m++) {
    if (m == mode) continue;
    for (int64_t i = 0; i < r * r; i++) H[i] *= gramians[m][i];
  }
}

/* ops::hadamard_all, src/utils/utils.cpp:156-159 (clobbers gramians[0]) */
void or_hadamard_all(double *const *gramians, int n_modes, int64_t r) {
  for (int m = 1; m < n_modes; m++)
    for (int64_t i = 0; i < r * r; i++) gramians[0][i] *= gramians[m][i];
}

/* update::update_factor_unconstrained, src/utils/update.cpp:178-192.  H (r x r, ld r) is
 * destroyed (becomes L); info != 0 only logs, the solves still run (as in the reference). */
int or_update_factor_unconstrained(double *panel, int64_t rows, int64_t r, int64_t ld, double *H) {
  int info = dpotf2_lower_ref(H, r, r);
  if (info) fprintf(stderr, "als_update_factor: DPORTF returned info=%d\n", info);
  dtrsm_RLT_ref(rows, r, H, r, panel, ld);
  dtrsm_RLN_ref(rows, r, H, r, panel, ld);
  return info;
}

/* ------------------------------------------------------------------------------------------ */
/* update::update_factor_non_negative_constrained, src/utils/update.cpp:61-176                 */
/* ------------------------------------------------------------------------------------------ */
/* calculate_sp, update.cpp:18-48: sp = G[P,P]^-1 y[P] over the passive set P = {i: !active[i]}
 * through dposv("L") = dpotrf + dpotrs (restated with the oracle's dpotf2 and two substitutions).
 * Returns the number of passive entries, or -1 when the Cholesky fails (CholFail). */
static int64_t nnls_calculate_sp(const double *y, double *sp, const double *G, double *Gp,
                                 const uint8_t *active, int64_t n) {
  int64_t np = 0;
  for (int64_t i = 0; i < n; i++) np += !active[i];
  int64_t a = 0;
  for (int64_t i = 0; i < n; i++)
    if (!active[i]) {
      sp[a] = y[i];
      int64_t b = 0;
      for (int64_t j = 0; j < n; j++)
        if (!active[j]) Gp[a + np * (b++)] = G[i + n * j];
      a++;
    }
  if (np == 0) return 0; /* dposv with n = 0 returns at once */
  if (dpotf2_lower_ref(Gp, np, np) > 0) return -1;
  for (int64_t j = 0; j < np; j++) { /* L z = b */
    double t = sp[j];
    for (int64_t k = 0; k < j; k++) t -= Gp[j + np * k] * sp[k];
    sp[j] = t / Gp[j + np * j];
  }
  for (int64_t j = np - 1; j >= 0; j--) { /* L^T x = z */
    double t = sp[j];
    for (int64_t k = j + 1; k < np; k++) t -= Gp[k + np * j] * sp[k];
    sp[j] = t / Gp[j + np * j];
  }
  return np;
}

/* Tensor::min() over the np entries sp was resized to (include/tensor.h:260).  With np == 0 the
 * reference's std::min_element(data, data) returns `data` itself and min() reads the stale sp[0]. */
static double nnls_min(const double *sp, int64_t np) {
  double m = sp[0];
  for (int64_t i = 1; i < np; i++)
    if (sp[i] < m) m = sp[i];
  return m;
}

static void nnls_scatter(double *dst, const double *sp, const uint8_t *active, int64_t n) {
  int64_t a = 0;
  for (int64_t i = 0; i < n; i++) dst[i] = active[i] ? 0.0 : sp[a++];
}

/* calculate_lagrangian_multipliers, update.cpp:50-56: w = y - G d */
static void nnls_multipliers(const double *y, const double *G, const double *d, double *w,
                             int64_t n) {
  for (int64_t i = 0; i < n; i++) w[i] = 0.0;
  for (int64_t j = 0; j < n; j++) /* dgemv 'N': column sweeps */
    for (int64_t i = 0; i < n; i++) w[i] += G[i + n * j] * d[j];
  for (int64_t i = 0; i < n; i++) w[i] = y[i] - w[i];
}

/* Tensor::max_id(mask), include/tensor.h:232-246: first strictly largest masked entry, 0 if none */
static int64_t nnls_max_id(const double *w, const uint8_t *mask, int64_t n) {
  int64_t id = 0;
  double mx = -DBL_MAX;
  for (int64_t i = 0; i < n; i++)
    if (mask[i] && w[i] > mx) {
      mx = w[i];
      id = i;
    }
  return id;
}

static int nnls_any(const uint8_t *active, int64_t n, int value) {
  for (int64_t i = 0; i < n; i++)
    if ((active[i] != 0) == (value != 0)) return 1;
  return 0;
}

/* The reference's loops have no iteration bound (an active-set method ends after finitely many
 * exchanges in exact arithmetic -- in floating point the exchange rule does cycle on some inputs, e.g. rank-48 models
 * on an all-positive noise tensor: the reference would not return).  The restatement stops a row after
 * OR_NNLS_MAX_EXCHANGES(n) = max(64, 16 n) passes of either loop and reports it (status 2): far above what a
 * terminating row needs (Lawson-Hanson implementations commonly allow 3 n), so that a cycling row cannot hang a
 * test or, on the device, hold a whole launch for thousands of solves. */
#define OR_NNLS_DEFAULT_BOUND(n) ((n) > 4 ? 16 * (n) : 64)
/* Test knobs (tests/test_oracle_nnls_termination.py pins the termination rule on the oracle alone): a bound
 * override (0 = the default max(64, 16 n)), the cycle rule on / off, and the largest pass count any loop of
 * the last call reached.  With the bound lifted and the cycle rule off the function IS the reference's
 * unbounded loop on every input on which that loop ends before the lifted bound. */
static int64_t or_nnls_bound_override = 0;
static int or_nnls_cycle_rule = 1;
static int64_t or_nnls_max_passes = 0;
void or_nnls_set_termination(int64_t bound, int cycle_rule) {
  or_nnls_bound_override = bound;
  or_nnls_cycle_rule = cycle_rule;
}
int64_t or_nnls_last_max_passes(void) { return or_nnls_max_passes; }
#define OR_NNLS_MAX_EXCHANGES (or_nnls_bound_override > 0 ? or_nnls_bound_override : OR_NNLS_DEFAULT_BOUND(n))
#define OR_NNLS_COUNT(g) do { if ((g) > or_nnls_max_passes) or_nnls_max_passes = (g); } while (0)

/* panel: rows x r (ld) holding the MTTKRP result, overwritten with the constrained solution;
 * H: r x r (ld r) Hadamard product of the other Gramians (NOT destroyed, unlike the unconstrained
 * update); active: rows x r flags, row-major [row][i], carried from sweep to sweep
 * (Ktensor::active_set, include/ktensor.h:37).  Returns 0, 1 when a Cholesky failed in the main
 * loop (the reference would terminate on the uncaught CholFail), 2 when a row hit the bound. */
int or_update_factor_nnls(double *panel, int64_t rows, int64_t r, int64_t ld, const double *H,
                          uint8_t *active_all) {
  const int64_t n = r;
  const double eps = 2.2204e-16; /* update.cpp:65 */
  double one_norm = -DBL_MAX;    /* Matrix::one_norm, include/matrix.h:121-129 */
  for (int64_t c = 0; c < n; c++) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += fabs(H[i + n * c]);
    if (s > one_norm) one_norm = s;
  }
  const double tol = 10 * eps * one_norm * (double)n;
  double *y = (double *)xmalloc(sizeof(double) * (size_t)(6 * n + n * n));
  double *d = y + n, *w = d + n, *s = w + n, *sp = s + n, *Gp = sp + n;
  uint8_t *top = (uint8_t *)xmalloc((size_t)n);
  int status = 0;
  or_nnls_max_passes = 0;
  for (int64_t i = 0; i < n; i++) sp[i] = 0.0;
  for (int64_t row = 0; row < rows; row++) {
    uint8_t *active = active_all + row * n;
    for (int64_t i = 0; i < n; i++) d[i] = 0.0;
    for (int64_t i = 0; i < n; i++) {
      y[i] = panel[row + ld * i];
      if (y[i] > 0) active[i] = 0;
    }
    if (nnls_any(active, n, 0)) { /* warm start from the previous sweep's passive set, :92-121 */
      int failed = 0;
      int64_t np = nnls_calculate_sp(y, sp, H, Gp, active, n);
      if (np < 0) failed = 1;
      if (!failed) {
        nnls_scatter(d, sp, active, n);
        int64_t guard = 0;
        while (nnls_min(sp, np) <= tol) {
          for (int64_t i = 0; i < n; i++)
            if (d[i] <= tol) {
              d[i] = 0.0;
              active[i] = 1;
            }
          if (!nnls_any(active, n, 0)) { failed = 1; break; } /* ZeroPassiveSet */
          np = nnls_calculate_sp(y, sp, H, Gp, active, n);
          if (np < 0) { failed = 1; break; }
          nnls_scatter(d, sp, active, n);
          OR_NNLS_COUNT(guard + 1);
          if (++guard > OR_NNLS_MAX_EXCHANGES) { status |= 2; break; }
        }
      }
      if (failed) { /* catch block, :117-120 */
        for (int64_t i = 0; i < n; i++) active[i] = 1, d[i] = 0.0;
      }
    }
    nnls_multipliers(y, H, d, w, n);
    int64_t guard = 0;
    while (nnls_any(active, n, 1) && w[nnls_max_id(w, active, n)] > tol) { /* main loop, :126-167 */
      const int64_t m = nnls_max_id(w, active, n);
      memcpy(top, active, (size_t)n); /* the set this pass starts from (cycle test below) */
      active[m] = 0;
      int64_t np = nnls_calculate_sp(y, sp, H, Gp, active, n);
      if (np < 0) { status |= 1; break; }
      int64_t guard2 = 0;
      while (nnls_min(sp, np) <= tol) { /* inner loop, :136-157 */
        nnls_scatter(s, sp, active, n);
        double a = DBL_MAX;
        for (int64_t i = 0; i < n; i++)
          if (!active[i] && s[i] <= tol) {
            const double t = d[i] / (d[i] - s[i]);
            if (t < a) a = t;
          }
        for (int64_t i = 0; i < n; i++) {
          d[i] = d[i] + a * (s[i] - d[i]);
          if (fabs(d[i]) < tol && !active[i]) {
            active[i] = 1;
            d[i] = 0;
          }
        }
        np = nnls_calculate_sp(y, sp, H, Gp, active, n);
        if (np < 0) { status |= 1; break; }
        OR_NNLS_COUNT(guard2 + 1);
        if (++guard2 > OR_NNLS_MAX_EXCHANGES) { status |= 2; break; }
      }
      if (np < 0) break;
      nnls_scatter(d, sp, active, n);
      nnls_multipliers(y, H, d, w, n);
      /* A pass that ends on the set it started from (the variable it made passive left again in the inner
       * loop) has reproduced its own starting state -- d and w are functions of the set -- so every further
       * pass repeats it: this is where the reference's exchange rule cycles and its loop never ends.  Stop at the
       * first such pass (same state as after any number of them) and report it like the bound. */
      if (or_nnls_cycle_rule && memcmp(top, active, (size_t)n) == 0) { status |= 2; break; }
      OR_NNLS_COUNT(guard + 1);
      if (++guard > OR_NNLS_MAX_EXCHANGES) { status |= 2; break; }
    }
    for (int64_t i = 0; i < n; i++) panel[row + ld * i] = d[i];
  }
  free(top);
  free(y);
  return status;
}

/* Ktensor::normalize(mode, iteration), src/ktensor.cpp:66-83 */
void or_normalize_mode(double *panel, int64_t rows, int64_t r, int64_t ld, double *lambda,
                       int64_t iteration) {
  for (int64_t c = 0; c < r; c++) {
    double *p = panel + c * ld;
    if (iteration == 1)
      lambda[c] = or_norm(p, rows);
    else
      lambda[c] = p[idamax_ref(p, rows)];
    if (lambda[c] != 0.0) {
      const double s = 1.0 / lambda[c];
      for (int64_t i = 0; i < rows; i++) p[i] *= s;
    }
  }
}

/* Ktensor::normalize(), src/ktensor.cpp:85-99 (no zero check, as in the reference) */
void or_normalize_all(double *const *factors, int n_modes, const int64_t *modes, int64_t r,
                      double *lambda) {
  for (int64_t c = 0; c < r; c++) lambda[c] = 1.0;
  for (int n = 0; n < n_modes; n++)
    for (int64_t c = 0; c < r; c++) {
      double *p = factors[n] + c * modes[n];
      const double coeff = or_norm(p, modes[n]);
      const double s = 1.0 / coeff;
      for (int64_t i = 0; i < modes[n]; i++) p[i] *= s;
      lambda[c] *= coeff;
    }
}

/* Ktensor::denormalize(), src/ktensor.cpp:101-107 */
void or_denormalize(double *factor0, int64_t rows, int64_t r, const double *lambda) {
  for (int64_t c = 0; c < r; c++)
    for (int64_t i = 0; i < rows; i++) factor0[i + c * rows] *= lambda[c];
}

/* ops::update_gramian, src/utils/utils.cpp:174-178: full r x r A^T A */
void or_update_gramian(const double *panel, int64_t rows, int64_t r, int64_t ld, double *gram) {
  for (int64_t b = 0; b < r; b++)
    for (int64_t a = 0; a < r; a++) {
      const double *pa = panel + a * ld, *pb = panel + b * ld;
      double t = 0.0;
      for (int64_t i = 0; i < rows; i++) t += pa[i] * pb[i];
      gram[a + b * r] = t;
    }
}

/* error::compute_fast_error, src/utils/error.cpp:64-89 */
double or_fast_error(double X_norm, const double *lambda, const double *last_factor, int64_t rows,
                     int64_t r, int64_t ld_f, const double *last_G, int64_t ld_g,
                     const double *gram_had) {
  double term2 = 0.0;
  for (int64_t j = 0; j < r; j++)
    for (int64_t i = 0; i < r; i++) term2 += lambda[i] * lambda[j] * gram_had[i + j * r];
  double term3 = 0.0;
  for (int64_t j = 0; j < r; j++)
    for (int64_t i = 0; i < rows; i++)
      term3 += lambda[j] * last_factor[i + j * ld_f] * last_G[i + j * ld_g];
  double e = fmax(X_norm * X_norm + term2 - 2 * term3, 0.0);
  return sqrt(e);
}

/* utils::calculate_jackknifing_norms, src/utils/utils.cpp:103-152: ||X without slice i of mode 0|| */
void or_jk_norms(const double *X, int n_modes, const int64_t *modes, double *norms_out) {
  int64_t cols = 1;
  for (int n = 1; n < n_modes; n++) cols *= modes[n];
  const int64_t I = modes[0];
  for (int64_t i = 0; i < I; i++) norms_out[i] = 0.0;
  for (int64_t j = 0; j < cols; j++)
    for (int64_t i = 0; i < I; i++) norms_out[i] += X[i + I * j] * X[i + I * j];
  double sum0 = 0.0;
  for (int64_t i = 0; i < I; i++) sum0 += norms_out[i];
  for (int64_t i = 0; i < I; i++) norms_out[i] = sqrt(sum0 - norms_out[i]);
}

/* Ktensor::to_tensor / rec_to_tensor, src/ktensor.cpp:32-64: X[idx] = sum_r lambda_r prod_n F_n */
void or_to_tensor(double *const *factors, const double *lambda, int n_modes, const int64_t *modes,
                  int64_t r, double *X_out) {
  int64_t total = 1;
  for (int n = 0; n < n_modes; n++) total *= modes[n];
  int64_t idx[OR_MAX_MODES] = {0};
  for (int64_t e = 0; e < total; e++) {
    double s = 0.0;
    for (int64_t c = 0; c < r; c++) {
      double m = 1.0;
      for (int n = 0; n < n_modes; n++) m *= factors[n][idx[n] + modes[n] * c];
      s += lambda[c] * m;
    }
    X_out[e] = s;
    for (int n = 0; n < n_modes; n++) { /* mode 0 fastest */
      if (++idx[n] < modes[n]) break;
      idx[n] = 0;
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Ktensor state                                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int n_modes;
  int64_t modes[OR_MAX_MODES];
  int64_t rank;
  double *fac[OR_MAX_MODES]; /* where the factor currently lives (own storage or buffer view) */
  double *own[OR_MAX_MODES]; /* the storage the Ktensor owns */
  double *lambda;
  uint8_t *active[OR_MAX_MODES]; /* Ktensor::active_set: [row][i], 1 = constraint active */
  int owns_memory; /* 1 => own[]/lambda were allocated here */
  int64_t iters;
  double fit, old_fit, err;
  int normalized;
  int jk, jk_mode;
  int64_t jk_fiber;
  /* test instrumentation, no counterpart in the reference: how close to a tie the accept / revert tests of the line
   * search were for this Ktensor (or_model::ls_margin); not part of Ktensor::copy */
  double ls_margin;
} kt_t;

static void kt_alloc(kt_t *k, int64_t rank, int n_modes, const int64_t *modes) {
  memset(k, 0, sizeof(*k));
  k->n_modes = n_modes;
  k->rank = rank;
  k->owns_memory = 1;
  for (int n = 0; n < n_modes; n++) {
    k->modes[n] = modes[n];
    k->own[n] = (double *)xmalloc(sizeof(double) * (size_t)(modes[n] * rank));
    k->fac[n] = k->own[n];
    k->active[n] = (uint8_t *)xmalloc((size_t)(modes[n] * rank));
    memset(k->active[n], 1, (size_t)(modes[n] * rank)); /* include/ktensor.h:69 */
  }
  k->lambda = (double *)xmalloc(sizeof(double) * (size_t)rank);
}

static void kt_free(kt_t *k) {
  for (int n = 0; n < k->n_modes; n++) free(k->active[n]);
  if (k->owns_memory) {
    for (int n = 0; n < k->n_modes; n++) free(k->own[n]);
    free(k->lambda);
  }
  memset(k, 0, sizeof(*k));
}

static void kt_from_model(kt_t *k, or_model *m, int n_modes, const int64_t *modes) {
  memset(k, 0, sizeof(*k));
  k->n_modes = n_modes;
  k->rank = m->rank;
  for (int n = 0; n < n_modes; n++) {
    k->modes[n] = modes[n];
    k->own[n] = m->factors[n];
    k->fac[n] = m->factors[n];
    /* a freshly constructed Ktensor: every constraint active (include/ktensor.h:69,108) */
    k->active[n] = (uint8_t *)xmalloc((size_t)(modes[n] * m->rank));
    memset(k->active[n], 1, (size_t)(modes[n] * m->rank));
  }
  k->lambda = m->lambda;
  k->jk = m->jk_enabled;
  k->jk_mode = m->jk_mode;
  k->jk_fiber = m->jk_fiber;
  k->ls_margin = 1e300;
}

static void kt_to_model(const kt_t *k, or_model *m) {
  m->iters = k->iters;
  m->fit = k->fit;
  m->old_fit = k->old_fit;
  m->approx_error = k->err;
  m->ls_margin = k->ls_margin;
}

/* Ktensor::copy, src/ktensor.cpp:163-181 (id and jk are NOT copied) */
static void kt_copy(kt_t *dst, const kt_t *src) {
  dst->err = src->err;
  dst->fit = src->fit;
  dst->old_fit = src->old_fit;
  dst->iters = src->iters;
  dst->normalized = src->normalized;
  memcpy(dst->lambda, src->lambda, sizeof(double) * (size_t)src->rank);
  for (int n = 0; n < src->n_modes; n++) {
    memcpy(dst->fac[n], src->fac[n], sizeof(double) * (size_t)(src->modes[n] * src->rank));
    memcpy(dst->active[n], src->active[n], (size_t)(src->modes[n] * src->rank)); /* :174 */
  }
}

/* Ktensor::set_jk_fiber(0.0), include/ktensor.h:316-325 */
static void kt_zero_jk_fiber(kt_t *k) {
  if (!k->jk) return;
  double *f = k->fac[k->jk_mode];
  for (int64_t c = 0; c < k->rank; c++) f[k->jk_fiber + c * k->modes[k->jk_mode]] *= 0.0;
}

/* Ktensor::calculate_new_fit, include/ktensor.h:178-183 */
static void kt_new_fit(kt_t *k, double X_norm) {
  k->old_fit = k->fit;
  k->fit = 1 - fabs(k->err) / X_norm;
}

static void update_gramians(const kt_t *k, double *const *gram) {
  for (int n = 0; n < k->n_modes; n++)
    or_update_gramian(k->fac[n], k->modes[n], k->rank, k->modes[n], gram[n]);
}

/* ------------------------------------------------------------------------------------------ */
/* line search                                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int iter, interval, updated_last_iter, method;
  double step;
  int extrapolated, reversed;
  kt_t prev, backup;
  const double *T; /* target tensor (ERROR_CHECKING_SERIAL) */
  double T_norm;
} ls_t;

/* error::compute_error, src/utils/error.cpp:7-30 (3-way only): ||X - [[lambda; A,B,C]]|| by
 * explicit reconstruction; denormalize() before, normalize() after (which changes the ktensor). */
static double compute_error_slow(const double *X, kt_t *k) {
  const int64_t I = k->modes[0], J = k->modes[1], K = k->modes[2], R = k->rank;
  or_denormalize(k->fac[0], I, R, k->lambda);
  k->normalized = 0;
  double *krp = (double *)xmalloc(sizeof(double) * (size_t)(J * K * R));
  or_khatri_rao(k->fac[2], K, k->fac[1], J, R, krp);
  double nrm2 = 0.0;
  for (int64_t jk = 0; jk < J * K; jk++)
    for (int64_t i = 0; i < I; i++) {
      double t = 0.0;
      for (int64_t c = 0; c < R; c++) t += k->fac[0][i + I * c] * krp[jk + J * K * c];
      const double d = X[i + I * jk] - t;
      nrm2 += d * d;
    }
  free(krp);
  or_normalize_all(k->fac, k->n_modes, k->modes, R, k->lambda);
  k->normalized = 1;
  return sqrt(nrm2);
}

/* ls::line_search_no_error_checking, src/utils/line_search.cpp:24-71 */
static void ls_no_error_checking(kt_t *k, kt_t *prev, double *const *gram, ls_t *p) {
  or_denormalize(k->fac[0], k->modes[0], k->rank, k->lambda);
  k->normalized = 0;
  or_denormalize(prev->fac[0], prev->modes[0], prev->rank, prev->lambda);
  prev->normalized = 0;
  for (int n = 0; n < k->n_modes; n++) {
    double *a = k->fac[n];
    const double *b = prev->fac[n];
    const int64_t ne = k->modes[n] * k->rank;
    for (int64_t i = 0; i < ne; i++) a[i] += p->step * (a[i] - b[i]);
  }
  or_normalize_all(k->fac, k->n_modes, k->modes, k->rank, k->lambda);
  k->normalized = 1;
  k->err = DBL_MAX; /* std::numeric_limits<double>::max() */
  kt_new_fit(k, 1.0);
  update_gramians(k, gram);
}

/* or_model::ls_margin: smallest |e1 - e2| / max(|e1|, |e2|) over the tests so far (NaN reads as a tie) */
static void ls_note_margin(kt_t *k, double e1, double e2) {
  const double scale = fmax(fmax(fabs(e1), fabs(e2)), 1e-300);
  const double m = fabs(e1 - e2) / scale;
  if (!(m >= k->ls_margin)) k->ls_margin = m;
}

/* ls::line_search_error_checking, src/utils/line_search.cpp:86-153 */
static void ls_error_checking(kt_t *k, kt_t *lsk, double *const *gram, ls_t *p) {
  for (int n = 0; n < k->n_modes; n++) {
    const double *cur = k->fac[n];
    double *old = lsk->fac[n];
    const int64_t ne = k->modes[n] * k->rank;
    for (int64_t i = 0; i < ne; i++) {
      const double diff = cur[i] - old[i];
      old[i] = cur[i];
      old[i] += p->step * diff; /* copy then daxpy */
    }
  }
  for (int64_t c = 0; c < k->rank; c++) lsk->lambda[c] = k->lambda[c];
  const double error = compute_error_slow(p->T, lsk);
  const double old_error = k->err;
  ls_note_margin(k, error, old_error);
  p->reversed = 1;
  if (error < old_error) {
    p->reversed = 0;
    for (int n = 0; n < k->n_modes; n++) {
      memcpy(k->fac[n], lsk->fac[n], sizeof(double) * (size_t)(k->modes[n] * k->rank));
      or_update_gramian(k->fac[n], k->modes[n], k->rank, k->modes[n], gram[n]);
    }
    k->err = error;
    kt_new_fit(k, p->T_norm);
  }
}

/* ls::line_search, src/utils/line_search.cpp:228-283 */
static void line_search(kt_t *k, double *const *gram, ls_t *p) {
  p->reversed = 0;
  p->extrapolated = 0;
  p->iter++;
  if (p->method == OR_LS_NO_ERROR_CHECKING) {
    if (p->updated_last_iter) {
      p->updated_last_iter = 0;
      ls_note_margin(k, p->backup.err, k->err);
      if (p->backup.err < k->err) {
        p->reversed = 1;
        p->iter = 0;
        kt_copy(k, &p->backup);
        update_gramians(k, gram);
      }
    }
    if (p->iter == p->interval) {
      p->extrapolated = 1;
      p->iter = 0;
      p->updated_last_iter = 1;
      kt_copy(&p->backup, k);
      ls_no_error_checking(k, &p->prev, gram, p);
    }
  } else if (p->method == OR_LS_ERROR_CHECKING_SERIAL) {
    if (p->iter == p->interval) {
      p->extrapolated = 1;
      p->iter = 0;
      ls_error_checking(k, &p->prev, gram, p);
    }
  }
}

static void ls_init(ls_t *p, const kt_t *k, const or_params *prm, const double *X, double X_norm) {
  memset(p, 0, sizeof(*p));
  kt_alloc(&p->prev, k->rank, k->n_modes, k->modes);
  kt_alloc(&p->backup, k->rank, k->n_modes, k->modes);
  p->interval = prm->line_search_interval;
  p->step = prm->line_search_step;
  p->method = prm->line_search_method;
  p->T = X;
  p->T_norm = X_norm;
}

static void ls_free(ls_t *p) {
  kt_free(&p->prev);
  kt_free(&p->backup);
}

/* ------------------------------------------------------------------------------------------ */
/* cp_als, src/als.cpp:19-289                                                                  */
/* ------------------------------------------------------------------------------------------ */
int or_cp_als(const double *X, int n_modes, const int64_t *modes, or_model *model,
              const or_params *prm, or_report *rep) {
  if (n_modes < 3 || n_modes > OR_MAX_MODES) return -1;
  /* error::compute_error (error.cpp:7-30) is 3-way code: with N > 3 the reference reads past its workspace */
  if (prm->line_search && prm->line_search_method == OR_LS_ERROR_CHECKING_SERIAL && n_modes > 3) return -1;
  const double t_total = now_s();
  int64_t n_el = 1;
  for (int n = 0; n < n_modes; n++) n_el *= modes[n];
  const double X_norm = or_norm(X, n_el);
  kt_t k;
  kt_from_model(&k, model, n_modes, modes);
  const int64_t r = k.rank;
  const int last = n_modes - 1;
  double *G_last = (double *)xmalloc(sizeof(double) * (size_t)(modes[last] * r));
  double *gram[OR_MAX_MODES];
  for (int n = 0; n < n_modes; n++) gram[n] = (double *)xmalloc(sizeof(double) * (size_t)(r * r));
  update_gramians(&k, gram); /* als.cpp:117-118 */
  ls_t ls;
  if (prm->line_search) ls_init(&ls, &k, prm, X, X_norm);
  int64_t iter = 0, ls_performed = 0, ls_failed = 0;
  int nnls_status = 0;
  double t_mttkrp = 0.0;
  k.iters = 0;
  int converged = 0;
  const double t_loop = now_s();
  do {
    iter++;
    k.iters += 1;
    if (prm->line_search && ls.iter == ls.interval - 1) kt_copy(&ls.prev, &k); /* :157-158 */
    for (int n = 0; n < n_modes; n++) {
      const double t0 = now_s();
      const int method = resolve_method(n_modes, prm->mttkrp_method, n, prm->threads);
      /* MTTKRP result overwrites factor n (mttkrp.cpp:311 returns u.get_factor(mode)); the
       * other factors are inputs, so compute into a scratch and copy. */
      double *Gn = (double *)xmalloc(sizeof(double) * (size_t)(modes[n] * r));
      or_mttkrp(X, n_modes, modes, k.fac, r, n, method, Gn);
      memcpy(k.fac[n], Gn, sizeof(double) * (size_t)(modes[n] * r));
      free(Gn);
      t_mttkrp += now_s() - t0;
      if (n == last) memcpy(G_last, k.fac[n], sizeof(double) * (size_t)(modes[n] * r));
      or_hadamard_but_one(gram, n_modes, r, n);
      if (prm->update_method == OR_UPDATE_UNCONSTRAINED) /* als.cpp:184-187 */
        or_update_factor_unconstrained(k.fac[n], modes[n], r, modes[n], gram[n]);
      else
        nnls_status |= or_update_factor_nnls(k.fac[n], modes[n], r, modes[n], gram[n], k.active[n]);
      if (k.jk && k.jk_mode == n) kt_zero_jk_fiber(&k);
      or_normalize_mode(k.fac[n], modes[n], r, modes[n], k.lambda, k.iters);
      or_update_gramian(k.fac[n], modes[n], r, modes[n], gram[n]);
    }
    or_hadamard_all(gram, n_modes, r);
    const double error =
        or_fast_error(X_norm, k.lambda, k.fac[last], modes[last], r, modes[last], G_last,
                      modes[last], gram[0]);
    k.err = error;
    kt_new_fit(&k, X_norm);
    if (prm->line_search) {
      if (!(ls.method == OR_LS_NO_ERROR_CHECKING && k.iters >= prm->max_iterations)) {
        if (prm->line_search_step == 0) ls.step = cbrt((double)k.iters);
        line_search(&k, gram, &ls);
        if (ls.extrapolated) ls_performed++;
        if (ls.reversed) ls_failed++;
      }
    }
    if (!prm->force_max_iter)
      converged = (fabs(k.old_fit - k.fit) < prm->tol) || (k.iters >= prm->max_iterations);
    else
      converged = k.iters >= prm->max_iterations;
  } while (!converged);
  const double t_end = now_s();
  kt_to_model(&k, model);
  if (rep) {
    memset(rep, 0, sizeof(*rep));
    rep->iter = iter;
    rep->n_ktensors = 1;
    rep->ktensor_comp_sum = r;
    rep->ls_performed = ls_performed;
    rep->ls_failed = ls_failed;
    rep->X_norm = X_norm;
    rep->total_time = t_end - t_total;
    rep->loop_time = t_end - t_loop;
    rep->mttkrp_time = t_mttkrp;
    rep->nnls_status = nnls_status;
  }
  if (prm->line_search) ls_free(&ls);
  for (int n = 0; n < n_modes; n++) free(gram[n]);
  free(G_last);
  kt_free(&k); /* the active sets; the factors belong to the caller */
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* MultiKtensor (src/multi_ktensor.cpp) + cp_cals (src/cals.cpp)                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  kt_t kt;          /* RegistryEntry::ktensor (reference to the caller's Ktensor) */
  or_model *model;  /* where results go */
  double *gram[OR_MAX_MODES];
  int64_t col;
  int64_t id;
  ls_t ls;
  int has_ls;
} entry_t;

typedef struct {
  int n_modes;
  int64_t modes[OR_MAX_MODES];
  int64_t buffer_size;
  double *buf[OR_MAX_MODES]; /* I_n x buffer_size, ld = I_n */
  int64_t *occ;              /* occupancy_vec: id per column, 0 = free */
  int64_t occupancy, start, end, active_cols;
  entry_t **reg;             /* registry in ascending id (std::map order) */
  int64_t n_reg, cap_reg;
  int64_t unique_kt_id;
  int flag_jk;
} mkt_t;

/* MultiKtensor::adjust_edges, src/multi_ktensor.cpp:165-186 (cell 0 is never examined) */
static void mkt_adjust_edges(mkt_t *m) {
  m->start = 0;
  int64_t end = m->buffer_size;
  for (int64_t i = end - 1; i > m->start; i--) {
    if (m->occ[i] == 0)
      end--;
    else
      break;
  }
  m->end = end;
  m->active_cols = end - m->start;
}

/* MultiKtensor::check_availability, src/multi_ktensor.cpp:14-39; -1 == BufferFull */
static int64_t mkt_check_availability(const mkt_t *m, int64_t rank) {
  int64_t comp_counter = 0, pos_index = -1, prev_occ = -1;
  for (int64_t i = 0; i < m->buffer_size; i++) {
    if (comp_counter == rank) break;
    if (m->occ[i] == 0 && prev_occ != 0) {
      pos_index = i;
      comp_counter++;
    } else if (m->occ[i] == 0 && prev_occ == 0)
      comp_counter++;
    else
      comp_counter = 0;
    prev_occ = m->occ[i];
  }
  if (pos_index == -1 || comp_counter != rank) return -1;
  return pos_index;
}

/* MultiKtensor::add, src/multi_ktensor.cpp:41-130; returns 0, or -1 for BufferFull */
static int mkt_add(mkt_t *m, or_model *model, const or_params *prm, const double *X,
                   double X_norm) {
  const int64_t pos = mkt_check_availability(m, model->rank);
  if (pos < 0) return -1;
  entry_t *e = (entry_t *)calloc(1, sizeof(entry_t));
  kt_from_model(&e->kt, model, m->n_modes, m->modes);
  e->model = model;
  const int64_t r = model->rank;
  /* Ktensor::attach, src/ktensor.cpp:109-125: copy into the buffer, then view it */
  for (int n = 0; n < m->n_modes; n++) {
    double *dst = m->buf[n] + pos * m->modes[n];
    memcpy(dst, e->kt.own[n], sizeof(double) * (size_t)(m->modes[n] * r));
    e->kt.fac[n] = dst;
  }
  const int64_t id = m->unique_kt_id++;
  for (int64_t i = 0; i < r; i++) m->occ[pos + i] = id;
  m->occupancy += r;
  for (int n = 0; n < m->n_modes; n++) {
    e->gram[n] = (double *)xmalloc(sizeof(double) * (size_t)(r * r));
    or_update_gramian(e->kt.fac[n], m->modes[n], r, m->modes[n], e->gram[n]);
  }
  e->kt.iters = 1;
  if (e->kt.jk) m->flag_jk = 1;
  e->col = pos;
  e->id = id;
  if (prm->line_search) {
    ls_init(&e->ls, &e->kt, prm, X, X_norm);
    e->has_ls = 1;
  }
  if (m->n_reg == m->cap_reg) {
    m->cap_reg = m->cap_reg ? 2 * m->cap_reg : 64;
    m->reg = (entry_t **)realloc(m->reg, sizeof(entry_t *) * (size_t)m->cap_reg);
  }
  m->reg[m->n_reg++] = e; /* ids are increasing => stays sorted */
  mkt_adjust_edges(m);
  return 0;
}

/* MultiKtensor::remove, src/multi_ktensor.cpp:132-163 + Ktensor::detach, src/ktensor.cpp:127-135 */
static void mkt_remove(mkt_t *m, int64_t id) {
  int64_t at = -1;
  for (int64_t i = 0; i < m->n_reg; i++)
    if (m->reg[i]->id == id) at = i;
  if (at < 0) return;
  entry_t *e = m->reg[at];
  const int64_t r = e->kt.rank;
  for (int n = 0; n < m->n_modes; n++) {
    const size_t bytes = sizeof(double) * (size_t)(m->modes[n] * r);
    memcpy(e->kt.own[n], e->kt.fac[n], bytes);
    memset(e->kt.fac[n], 0, bytes);
    e->kt.fac[n] = e->kt.own[n];
  }
  kt_to_model(&e->kt, e->model);
  for (int64_t i = 0; i < m->buffer_size; i++)
    if (m->occ[i] == id) m->occ[i] = 0;
  m->occupancy -= r;
  for (int n = 0; n < m->n_modes; n++) free(e->gram[n]);
  if (e->has_ls) ls_free(&e->ls);
  kt_free(&e->kt); /* the active sets; the factors belong to the caller */
  free(e);
  memmove(&m->reg[at], &m->reg[at + 1], sizeof(entry_t *) * (size_t)(m->n_reg - at - 1));
  m->n_reg--;
  mkt_adjust_edges(m);
}

/* MultiKtensor::compress, src/multi_ktensor.cpp:188-264: the move list (id, offset) is built on
 * the un-moved occupancy vector (:201-209), then applied left to right (:212-244). */
static void mkt_compress(mkt_t *m) {
  int64_t col_offset = 0, added = -1, n_req = 0;
  int64_t *req_id = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m->n_reg + 1));
  int64_t *req_off = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m->n_reg + 1));
  for (int64_t c = 0; c < m->buffer_size; c++) {
    const int64_t cell = m->occ[c];
    if (cell == added)
      continue;
    else if (cell == 0)
      col_offset++;
    else if (col_offset != 0) {
      req_id[n_req] = cell;
      req_off[n_req++] = col_offset;
      added = cell;
    }
  }
  for (int64_t q = 0; q < n_req; q++) {
    entry_t *e = NULL;
    for (int64_t i = 0; i < m->n_reg; i++)
      if (m->reg[i]->id == req_id[q]) e = m->reg[i];
    const int64_t off = req_off[q], r = e->kt.rank;
    for (int n = 0; n < m->n_modes; n++) { /* Ktensor::attach(new_data): copy + re-point */
      double *nd = e->kt.fac[n] - off * m->modes[n];
      memmove(nd, e->kt.fac[n], sizeof(double) * (size_t)(m->modes[n] * r));
      e->kt.fac[n] = nd;
    }
    for (int64_t i = e->col; i < e->col + r; i++) {
      const int64_t t = m->occ[i - off];
      m->occ[i - off] = m->occ[i];
      m->occ[i] = t;
    }
    e->col -= off;
  }
  free(req_id);
  free(req_off);
  mkt_adjust_edges(m);
}

int or_cp_cals(const double *X, int n_modes, const int64_t *modes, or_model *models,
               int64_t n_models, const or_params *prm, or_report *rep) {
  if (n_modes < 3 || n_modes > OR_MAX_MODES) return -1;
  /* error::compute_error (error.cpp:7-30) is 3-way code: with N > 3 the reference reads past its workspace */
  if (prm->line_search && prm->line_search_method == OR_LS_ERROR_CHECKING_SERIAL && n_modes > 3) return -1;
  for (int64_t i = 0; i < n_models; i++)
    if (models[i].rank > prm->buffer_size || models[i].rank < 1)
      return -2; /* the reference would spin forever; reject instead */
  const double t_total = now_s();
  int64_t n_el = 1;
  for (int n = 0; n < n_modes; n++) n_el *= modes[n];
  const double X_norm = or_norm(X, n_el); /* cals.cpp:36 */
  const int last = n_modes - 1;
  const int64_t bs = prm->buffer_size;

  mkt_t m;
  memset(&m, 0, sizeof(m));
  m.n_modes = n_modes;
  m.buffer_size = bs;
  m.unique_kt_id = 1;
  for (int n = 0; n < n_modes; n++) {
    m.modes[n] = modes[n];
    m.buf[n] = (double *)xmalloc(sizeof(double) * (size_t)(modes[n] * bs));
    memset(m.buf[n], 0, sizeof(double) * (size_t)(modes[n] * bs));
  }
  m.occ = (int64_t *)calloc((size_t)bs, sizeof(int64_t));
  mkt_adjust_edges(&m);

  double *G_last = (double *)xmalloc(sizeof(double) * (size_t)(modes[last] * bs));
  double *Gn = NULL;
  {
    int64_t mx = 0;
    for (int n = 0; n < n_modes; n++)
      if (modes[n] > mx) mx = modes[n];
    Gn = (double *)xmalloc(sizeof(double) * (size_t)(mx * bs));
  }
  double *X_norms_jk = NULL;

  int64_t q_head = 0; /* the KtensorQueue: models[q_head..n_models) */
  int64_t sweep = 0, n_kt = 0, comp_sum = 0, ls_performed = 0, ls_failed = 0;
  int nnls_status = 0;
  double t_mttkrp = 0.0;
  int converged = 0;
  const double t_loop = now_s();
  do {
    sweep++;
    /* admission, cals.cpp:182-192 */
    while (q_head < n_models) {
      if (mkt_add(&m, &models[q_head], prm, X, X_norm) != 0) break;
      n_kt++;
      comp_sum += models[q_head].rank;
      q_head++;
    }
    if (m.flag_jk && !X_norms_jk) { /* cals.cpp:198-200 */
      X_norms_jk = (double *)xmalloc(sizeof(double) * (size_t)modes[0]);
      or_jk_norms(X, n_modes, modes, X_norms_jk);
    }
    if (prm->line_search) { /* cals.cpp:203-211 */
#pragma omp parallel for schedule(dynamic)
      for (int64_t i = 0; i < m.n_reg; i++) {
        entry_t *e = m.reg[i];
        if (e->ls.iter == e->ls.interval - 1) kt_copy(&e->ls.prev, &e->kt);
      }
    }
    const int64_t R = m.active_cols;
    for (int n = 0; n < n_modes; n++) {
      const double t0 = now_s();
      const int method = resolve_method(n_modes, prm->mttkrp_method, n, prm->threads);
      or_mttkrp(X, n_modes, modes, m.buf, R, n, method, Gn);
      memcpy(m.buf[n], Gn, sizeof(double) * (size_t)(modes[n] * R));
      t_mttkrp += now_s() - t0;
      if (n == last) memcpy(G_last, m.buf[n], sizeof(double) * (size_t)(modes[n] * R));
#pragma omp parallel for schedule(dynamic)
      for (int64_t i = 0; i < m.n_reg; i++) { /* cals.cpp:239-256 */
        entry_t *e = m.reg[i];
        const int64_t r = e->kt.rank;
        or_hadamard_but_one(e->gram, n_modes, r, n);
        if (prm->update_method == OR_UPDATE_UNCONSTRAINED) /* cals.cpp:244-248 */
          or_update_factor_unconstrained(e->kt.fac[n], modes[n], r, modes[n], e->gram[n]);
        else {
          const int st = or_update_factor_nnls(e->kt.fac[n], modes[n], r, modes[n], e->gram[n],
                                               e->kt.active[n]);
          if (st) {
#pragma omp atomic
            nnls_status |= st;
          }
        }
        if (e->kt.jk && e->kt.jk_mode == n) kt_zero_jk_fiber(&e->kt);
        or_normalize_mode(e->kt.fac[n], modes[n], r, modes[n], e->kt.lambda, e->kt.iters);
        or_update_gramian(e->kt.fac[n], modes[n], r, modes[n], e->gram[n]);
      }
    }
#pragma omp parallel for schedule(dynamic)
    for (int64_t i = 0; i < m.n_reg; i++) { /* cals.cpp:281-303 */
      entry_t *e = m.reg[i];
      const int64_t r = e->kt.rank;
      or_hadamard_all(e->gram, n_modes, r);
      const double *kt_G_last = G_last + (e->col - m.start) * modes[last];
      double X_norm_kt = X_norm;
      if (e->kt.jk) X_norm_kt = X_norms_jk[e->kt.jk_fiber];
      const double error = or_fast_error(X_norm_kt, e->kt.lambda, e->kt.fac[last], modes[last], r,
                                         modes[last], kt_G_last, modes[last], e->gram[0]);
      e->kt.err = error;
      kt_new_fit(&e->kt, X_norm);
    }
    if (prm->line_search) { /* cals.cpp:310-331 */
      for (int64_t i = 0; i < m.n_reg; i++) {
        entry_t *e = m.reg[i];
        if (!(e->ls.method == OR_LS_NO_ERROR_CHECKING && e->kt.iters >= prm->max_iterations)) {
          if (prm->line_search_step == 0) e->ls.step = cbrt((double)e->kt.iters);
          line_search(&e->kt, e->gram, &e->ls);
          if (e->ls.extrapolated) ls_performed++;
          if (e->ls.reversed) ls_failed++;
        }
      }
    }
    /* eviction list, cals.cpp:336-354 */
    int64_t n_rm = 0;
    int64_t *rm = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(m.n_reg + 1));
    for (int64_t i = 0; i < m.n_reg; i++) {
      entry_t *e = m.reg[i];
      if (!prm->always_evict_first) {
        if (!prm->force_max_iter) {
          if (fabs(e->kt.old_fit - e->kt.fit) < prm->tol || e->kt.iters >= prm->max_iterations)
            rm[n_rm++] = e->id;
          else
            e->kt.iters += 1;
        } else if (e->kt.iters >= prm->max_iterations)
          rm[n_rm++] = e->id;
        else
          e->kt.iters += 1;
      } else {
        const int64_t id = m.occ[0]; /* get_leftmost_id, include/multi_ktensor.h:95-100 */
        if (id > 0) rm[n_rm++] = id;
        break;
      }
    }
    for (int64_t i = 0; i < n_rm; i++) mkt_remove(&m, rm[i]);
    free(rm);
    mkt_compress(&m);
    if (q_head >= n_models && m.n_reg == 0) converged = 1;
  } while (!converged);
  const double t_end = now_s();

  if (rep) {
    memset(rep, 0, sizeof(*rep));
    rep->iter = sweep;
    rep->n_ktensors = n_kt;
    rep->ktensor_comp_sum = comp_sum;
    rep->ls_performed = ls_performed;
    rep->ls_failed = ls_failed;
    rep->X_norm = X_norm;
    rep->total_time = t_end - t_total;
    rep->loop_time = t_end - t_loop;
    rep->mttkrp_time = t_mttkrp;
    rep->nnls_status = nnls_status;
  }
  free(X_norms_jk);
  free(Gn);
  free(G_last);
  free(m.occ);
  free(m.reg);
  for (int n = 0; n < n_modes; n++) free(m.buf[n]);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* jackknife driver                                                                            */
/* ------------------------------------------------------------------------------------------ */
int or_lsap_bruteforce(int n, const double *cost, int maximize, int64_t *col_of_row) {
  if (n < 1 || n > 9) return -1;
  int perm[9], best[9];
  for (int i = 0; i < n; i++) perm[i] = i;
  double best_v = 0.0;
  int have = 0;
  for (;;) {
    double v = 0.0;
    for (int i = 0; i < n; i++) v += cost[i + n * perm[i]];
    if (!have || (maximize ? v > best_v : v < best_v)) {
      best_v = v;
      have = 1;
      memcpy(best, perm, sizeof(int) * (size_t)n);
    }
    /* next permutation (lexicographic) */
    int k = n - 2;
    while (k >= 0 && perm[k] > perm[k + 1]) k--;
    if (k < 0) break;
    int l = n - 1;
    while (perm[l] < perm[k]) l--;
    int t = perm[k]; perm[k] = perm[l]; perm[l] = t;
    for (int a = k + 1, b = n - 1; a < b; a++, b--) { t = perm[a]; perm[a] = perm[b]; perm[b] = t; }
  }
  for (int i = 0; i < n; i++) col_of_row[i] = best[i];
  return 0;
}

/* jk_permutation_adjustment for ONE replica, utils.cpp:54-101 -- restated with the reference's
 * orientation: it builds M(a, b) = <Bov_a, Bm_b> + <Cov_a, Cm_b> (a = overall column, b = replica
 * column) COLUMN-major (utils.cpp:69-74) and passes the buffer to
 * solve_rectangular_linear_sum_assignment (utils.cpp:79), which reads ROW-major
 * (rectangular_lsap.cpp:93).  The problem actually solved has row b = replica column, column a =
 * overall column, i.e. the cost matrix Mt(b, a) = M(a, b); solved[b] = the overall column matched to
 * replica column b.  or_lsap_bruteforce takes a column-major cost with rows first, so Mt is stored as
 * Mt[b + r * a].  Then new(:, cur) = old(:, solved[cur]) in every mode (utils.cpp:81-97). */
int or_jk_permutation_adjust(int n_modes, const int64_t *modes, int64_t r, const double *const *overall,
                             double *const *replica) {
  if (r < 1 || r > 9 || n_modes < 3) return -1;
  double *M = (double *)xmalloc(sizeof(double) * (size_t)(r * r));
  for (int64_t b = 0; b < r; b++)
    for (int64_t a = 0; a < r; a++) {
      double s = 0.0, t = 0.0;
      for (int64_t q = 0; q < modes[1]; q++) s += overall[1][q + modes[1] * a] * replica[1][q + modes[1] * b];
      for (int64_t q = 0; q < modes[2]; q++) t += overall[2][q + modes[2] * a] * replica[2][q + modes[2] * b];
      M[b + r * a] = s + t;
    }
  int64_t solved[9];
  if (or_lsap_bruteforce((int)r, M, 1, solved)) { free(M); return -3; }
  free(M);
  for (int n = 0; n < n_modes; n++) {
    const size_t bytes = sizeof(double) * (size_t)(modes[n] * r);
    double *copy = (double *)xmalloc(bytes);
    memcpy(copy, replica[n], bytes);
    for (int64_t cur = 0; cur < r; cur++)
      if (solved[cur] != cur)
        memcpy(replica[n] + modes[n] * cur, copy + modes[n] * solved[cur], sizeof(double) * (size_t)modes[n]);
    free(copy);
  }
  return 0;
}

int or_jk_cp_cals(const double *X, int n_modes, const int64_t *modes, const or_model *kt_vector,
                  int64_t n_models, const or_params *params, or_model *results, or_report *rep) {
  const int64_t I0 = modes[0];
  if (n_modes < 3 || I0 <= 1) return -1;
  /* ktensors(kt_vector): denormalize(); normalize()  (cals.cpp:399-405) */
  kt_t *over = (kt_t *)calloc((size_t)n_models, sizeof(kt_t));
  for (int64_t k = 0; k < n_models; k++) {
    kt_alloc(&over[k], kt_vector[k].rank, n_modes, modes);
    for (int n = 0; n < n_modes; n++)
      memcpy(over[k].fac[n], kt_vector[k].factors[n], sizeof(double) * (size_t)(modes[n] * kt_vector[k].rank));
    memcpy(over[k].lambda, kt_vector[k].lambda, sizeof(double) * (size_t)kt_vector[k].rank);
    or_denormalize(over[k].fac[0], modes[0], over[k].rank, over[k].lambda);
    or_normalize_all(over[k].fac, n_modes, modes, over[k].rank, over[k].lambda);
    /* generate_jk_ktensors (utils.cpp:40-52): plain copies flagged jk(0, i) */
    for (int64_t i = 0; i < I0; i++) {
      or_model *m = &results[k * I0 + i];
      m->rank = over[k].rank;
      for (int n = 0; n < n_modes; n++)
        memcpy(m->factors[n], over[k].fac[n], sizeof(double) * (size_t)(modes[n] * m->rank));
      memcpy(m->lambda, over[k].lambda, sizeof(double) * (size_t)m->rank);
      m->jk_enabled = 1;
      m->jk_mode = 0;
      m->jk_fiber = i;
    }
  }
  int rc = or_cp_cals(X, n_modes, modes, results, n_models * I0, params, rep);
  if (rc) return rc;
  for (int64_t k = 0; k < n_models; k++) {
    const int64_t r = over[k].rank;
    for (int64_t i = 0; i < I0; i++) { /* cals.cpp:431-437 */
      or_model *m = &results[k * I0 + i];
      for (int64_t c = 0; c < r; c++) m->factors[0][i + I0 * c] *= 0.0;
      or_denormalize(m->factors[0], I0, r, m->lambda);
      or_normalize_all(m->factors, n_modes, modes, r, m->lambda);
      for (int64_t c = 0; c < r; c++) m->factors[0][i + I0 * c] = NAN;
    }
    /* jk_permutation_adjustment, utils.cpp:54-101 */
    for (int64_t i = 0; i < I0; i++) {
      or_model *m = &results[k * I0 + i];
      if (or_jk_permutation_adjust(n_modes, modes, r, (const double *const *)over[k].fac, m->factors)) return -3;
    }
    kt_free(&over[k]);
  }
  free(over);
  return 0;
}

void or_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
