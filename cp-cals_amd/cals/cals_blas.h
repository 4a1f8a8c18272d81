// include/cals_blas.h of HPAC/CP-CALS selects a host BLAS vendor (MKL / BLIS / OpenBLAS / MATLAB) and
// declares the thread control.  This library's hot path calls NO BLAS -- it runs as hand-written HIP
// kernels in libcals_hip.so -- so what remains of this header is what the FRONT-ENDS use from it:
//   * global set_threads / get_threads (src/examples/driver.cpp:128, matlab_cp_cals_jk.cpp:147),
//   * the CBLAS enums and cblas_dgemm, which the MEX jackknife glue calls directly
//     (matlab/matlab_cp_cals_jk.cpp:176-181), plus the level-1 routines the value classes' host
//     conveniences are written with in the reference.
// They are small plain-loop host functions (r x r and I x r operands of post-processing code), not a
// compute path.
#ifndef CALS_AMD_CALS_BLAS_H
#define CALS_AMD_CALS_BLAS_H

#include <cmath>
#include <cstddef>

#define CALS_BACKEND "MI355X-HIP"

enum CBLAS_ORDER { CblasRowMajor = 101, CblasColMajor = 102 };
enum CBLAS_TRANSPOSE { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 };
enum CBLAS_UPLO { CblasUpper = 121, CblasLower = 122 };
enum CBLAS_DIAG { CblasNonUnit = 131, CblasUnit = 132 };
enum CBLAS_SIDE { CblasLeft = 141, CblasRight = 142 };

inline double cblas_dnrm2(ptrdiff_t n, const double *x, ptrdiff_t incx) {
  double scale = 0.0, ssq = 1.0;  // scaled sum of squares: no overflow for huge entries
  for (ptrdiff_t i = 0; i < n; i++) {
    const double a = std::fabs(x[i * incx]);
    if (a == 0.0) continue;
    if (scale < a) {
      ssq = 1.0 + ssq * (scale / a) * (scale / a);
      scale = a;
    } else {
      ssq += (a / scale) * (a / scale);
    }
  }
  return scale * std::sqrt(ssq);
}
inline double cblas_dasum(ptrdiff_t n, const double *x, ptrdiff_t incx) {
  double s = 0.0;
  for (ptrdiff_t i = 0; i < n; i++) s += std::fabs(x[i * incx]);
  return s;
}
inline ptrdiff_t cblas_idamax(ptrdiff_t n, const double *x, ptrdiff_t incx) {  // first index of max |x_i|
  ptrdiff_t best = 0;
  for (ptrdiff_t i = 1; i < n; i++)
    if (std::fabs(x[i * incx]) > std::fabs(x[best * incx])) best = i;
  return best;
}
inline void cblas_dcopy(ptrdiff_t n, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  for (ptrdiff_t i = 0; i < n; i++) y[i * incy] = x[i * incx];
}
inline void cblas_dscal(ptrdiff_t n, double a, double *x, ptrdiff_t incx) {
  for (ptrdiff_t i = 0; i < n; i++) x[i * incx] *= a;
}
inline void cblas_daxpy(ptrdiff_t n, double a, const double *x, ptrdiff_t incx, double *y, ptrdiff_t incy) {
  for (ptrdiff_t i = 0; i < n; i++) y[i * incy] += a * x[i * incx];
}
// C = alpha * op(A) * op(B) + beta * C, column-major only (all the reference's call sites)
inline void cblas_dgemm(CBLAS_ORDER, CBLAS_TRANSPOSE ta, CBLAS_TRANSPOSE tb, ptrdiff_t M, ptrdiff_t N, ptrdiff_t K,
                        double alpha, const double *A, ptrdiff_t lda, const double *B, ptrdiff_t ldb, double beta,
                        double *C, ptrdiff_t ldc) {
  for (ptrdiff_t j = 0; j < N; j++)
    for (ptrdiff_t i = 0; i < M; i++) {
      double s = 0.0;
      for (ptrdiff_t k = 0; k < K; k++) {
        const double a = (ta == CblasNoTrans) ? A[i + k * lda] : A[k + i * lda];
        const double b = (tb == CblasNoTrans) ? B[k + j * ldb] : B[j + k * ldb];
        s += a * b;
      }
      C[i + j * ldc] = alpha * s + (beta == 0.0 ? 0.0 : beta * C[i + j * ldc]);
    }
}
inline void cblas_dgemv(CBLAS_ORDER, CBLAS_TRANSPOSE ta, ptrdiff_t M, ptrdiff_t N, double alpha, const double *A,
                        ptrdiff_t lda, const double *x, ptrdiff_t incx, double beta, double *y, ptrdiff_t incy) {
  const ptrdiff_t rows = (ta == CblasNoTrans) ? M : N, inner = (ta == CblasNoTrans) ? N : M;
  for (ptrdiff_t i = 0; i < rows; i++) {
    double s = 0.0;
    for (ptrdiff_t k = 0; k < inner; k++) s += ((ta == CblasNoTrans) ? A[i + k * lda] : A[k + i * lda]) * x[k * incx];
    y[i * incy] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[i * incy]);
  }
}

void set_threads(int threads);  // include/cals_blas.h:184: host BLAS threads; recorded, no effect on the device path
int get_threads();              // include/cals_blas.h:186

#endif
