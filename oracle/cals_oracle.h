/*
 * cals_oracle.h -- CPU restatement (plain C) of the HPAC/CP-CALS concurrent-ALS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cp-cals_amd/ (the product) may include, link or
 * load this.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: the reference cannot be built in this image (every BLAS backend of
 * include/cals_blas.h needs a vendor header -- mkl.h / cblas.h / blis.h / MATLAB blas.h -- that
 * the image lacks, and writing a stand-in header is not allowed), and the reference ships no
 * golden vectors (all its tests are differential).  The oracle is therefore
 * pinned against the reference's OWN TEST SUITE restated on it (tests/test_oracle_reference_suite.py:
 * tests/als/test_als.cpp and tests/cals/test_cals.cpp of the reference), NOT against outputs of
 * the reference run here: "parity unpinned" with respect to reference-produced numbers.
 * One exception: the reference's extern/rectangular_lsap/rectangular_lsap.cpp is self-contained, so
 * oracle/Makefile compiles it from where it lies into oracle/_ref/librectangular_lsap.so, and the
 * assignment step of the jackknife post-processing (or_jk_permutation_adjust, and the product's own
 * solver) IS pinned against that build (tests/test_lsap_and_jk_permutation.py).
 *
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * BLAS/LAPACK are third-party to the reference (MKL 2021.4 / OpenBLAS in its CI); their
 * arithmetic is restated from the published Netlib reference algorithms (LAPACK 3.x dpotf2.f,
 * BLAS dtrsm.f / dgemm.f / dgemv.f / idamax.f); dnrm2 is the plain sqrt(sum of squares).
 */
#ifndef CALS_ORACLE_H
#define CALS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAX_MODES 8

/* mttkrp::MTTKRP_METHOD, include/utils/mttkrp.h:23-31 */
enum { OR_MTTKRP = 0, OR_TWOSTEP0 = 1, OR_TWOSTEP1 = 2, OR_AUTO = 3 };
/* ls::LS_METHOD, include/utils/line_search.h:8 */
enum { OR_LS_NO_ERROR_CHECKING = 0, OR_LS_ERROR_CHECKING_SERIAL = 1 };
/* update::UPDATE_METHOD, include/utils/update.h:7 */
enum { OR_UPDATE_UNCONSTRAINED = 0, OR_UPDATE_NNLS = 1 };

/* CalsParams (include/cals.h:138-159) / AlsParams (include/als.h:142-166), fields on the path. */
typedef struct {
  int64_t max_iterations;     /* default 200 */
  double tol;                 /* default 1e-7 */
  int64_t buffer_size;        /* default 4200 (cp_cals only) */
  int mttkrp_method;          /* default OR_AUTO */
  int line_search;            /* default 0 */
  int line_search_interval;   /* default 5 */
  double line_search_step;    /* default 0 => cbrt(iter) */
  int line_search_method;     /* default OR_LS_NO_ERROR_CHECKING */
  int force_max_iter;         /* default 0 */
  int always_evict_first;     /* default 0 */
  int threads;                /* what get_threads() would return: drives the AUTO heuristic */
  int update_method;          /* default OR_UPDATE_UNCONSTRAINED */
} or_params;

/* CalsReport fields that are results (include/cals.h:27-52). */
typedef struct {
  int64_t iter;             /* outer sweeps executed */
  int64_t n_ktensors;
  int64_t ktensor_comp_sum;
  int64_t ls_performed;
  int64_t ls_failed;
  double X_norm;
  double total_time;        /* seconds, whole call */
  double loop_time;         /* seconds, do{}while loop only */
  double mttkrp_time;       /* seconds inside mttkrp() */
  int nnls_status;          /* OR of or_update_factor_nnls' return values (0 = clean) */
} or_report;

/* One model (Ktensor, include/ktensor.h:24-44).  Factors are col-major I_n x rank, ld = I_n. */
typedef struct {
  int64_t rank;
  double *factors[OR_MAX_MODES]; /* caller-owned storage, in/out */
  double *lambda;                /* rank, in/out */
  int jk_enabled;                /* JackKniffing, include/ktensor.h:18-22 */
  int jk_mode;
  int64_t jk_fiber;
  /* outputs */
  int64_t iters;
  double fit, old_fit, approx_error;
  /* test instrumentation (no counterpart in the reference): the smallest relative distance |e1 - e2| / max(|e1|, |e2|)
   * between the two errors of any accept / revert test of the line search (line_search.cpp:239, :116) this model went
   * through; 1e300 = none.  Rounding-sized = that decision was a tie. */
  double ls_margin;
} or_model;

void or_default_params(or_params *p);

/* GEMM backend hook: when set, the big MTTKRP GEMMs go through it (e.g. MKL's cblas_dgemm,
 * resolved with dlsym by the caller); NULL => the oracle's own triple loop.
 * Signature = cblas_dgemm with Order fixed to CblasColMajor(102), trans: 111 N / 112 T. */
typedef void (*or_dgemm_fn)(int order, int transa, int transb, int m, int n, int k, double alpha,
                            const double *a, int lda, const double *b, int ldb, double beta,
                            double *c, int ldc);
void or_set_dgemm(or_dgemm_fn fn);
/* OpenMP threads for the per-model loops / own GEMM (the reference's set_threads, cals_blas.h:184). */
void or_set_threads(int n);

/* ---- building blocks (each usable alone from the tests) ---- */
double or_norm(const double *x, int64_t n);
void or_khatri_rao(const double *A, int64_t IA, const double *B, int64_t IB, int64_t cols,
                   double *K);
/* MTTKRP of the multi-factor block: factors[n] is I_n x R (ld = I_n); result -> G (I_mode x R).
 * workspace must hold (product of all modes / smallest mode) * R doubles at least twice over
 * (or pass NULL to let the oracle allocate). method: OR_MTTKRP | OR_TWOSTEP0 | OR_TWOSTEP1. */
void or_mttkrp(const double *X, int n_modes, const int64_t *modes, double *const *factors,
               int64_t R, int mode, int method, double *G);
void or_hadamard_but_one(double *const *gramians, int n_modes, int64_t r, int mode);
void or_hadamard_all(double *const *gramians, int n_modes, int64_t r);
int or_update_factor_unconstrained(double *panel, int64_t rows, int64_t r, int64_t ld, double *H);
/* active: rows x r bytes, [row][i], 1 = constraint active; in/out across sweeps */
int or_update_factor_nnls(double *panel, int64_t rows, int64_t r, int64_t ld, const double *H,
                          uint8_t *active);
/* test knobs of the NNLS termination rule: bound (0 = default max(64, 16 r)), cycle rule on / off; the largest
 * number of passes any loop of the last or_update_factor_nnls call made */
void or_nnls_set_termination(int64_t bound, int cycle_rule);
int64_t or_nnls_last_max_passes(void);
void or_normalize_mode(double *panel, int64_t rows, int64_t r, int64_t ld, double *lambda,
                       int64_t iteration);
void or_normalize_all(double *const *factors, int n_modes, const int64_t *modes, int64_t r,
                      double *lambda);
void or_denormalize(double *factor0, int64_t rows, int64_t r, const double *lambda);
void or_update_gramian(const double *panel, int64_t rows, int64_t r, int64_t ld, double *gram);
double or_fast_error(double X_norm, const double *lambda, const double *last_factor, int64_t rows,
                     int64_t r, int64_t ld_f, const double *last_G, int64_t ld_g,
                     const double *gram_had);
void or_jk_norms(const double *X, int n_modes, const int64_t *modes, double *norms_out);
void or_to_tensor(double *const *factors, const double *lambda, int n_modes, const int64_t *modes,
                  int64_t r, double *X_out);

/* ---- the two drivers ---- */
/* cp_als, src/als.cpp:19-289 */
int or_cp_als(const double *X, int n_modes, const int64_t *modes, or_model *model,
              const or_params *params, or_report *rep);
/* cp_cals, src/cals.cpp:19-395: models are consumed in order (the KtensorQueue). */
int or_cp_cals(const double *X, int n_modes, const int64_t *modes, or_model *models,
               int64_t n_models, const or_params *params, or_report *rep);

/* jk_cp_cals, src/cals.cpp:397-446: results must hold n_models * modes[0] models with
 * caller-allocated factors (I_n x rank) and lambda; they receive the jackknife replicas (replica i of
 * model k at results[k * modes[0] + i]) after re-normalisation and column matching
 * (utils::jk_permutation_adjustment, src/utils/utils.cpp:54-101).  The assignment problem, which
 * the reference hands to SciPy's rectangular_lsap (extern/), is solved here by exhaustive search
 * (rank <= 9): an independent check of the product's solver.  The reference's own rectangular_lsap.cpp
 * is self-contained, so oracle/Makefile also builds it from where it lies (oracle/_ref/) and the tests
 * pin both against it. */
int or_lsap_bruteforce(int n, const double *cost_colmajor, int maximize, int64_t *col_of_row);
/* utils::jk_permutation_adjustment for one replica (factors in place), with the reference's
 * orientation of the assignment problem (see the definition). */
int or_jk_permutation_adjust(int n_modes, const int64_t *modes, int64_t r, const double *const *overall,
                             double *const *replica);
int or_jk_cp_cals(const double *X, int n_modes, const int64_t *modes, const or_model *kt_vector,
                  int64_t n_models, const or_params *params, or_model *results, or_report *rep);

#ifdef __cplusplus
}
#endif
#endif
