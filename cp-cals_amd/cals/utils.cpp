// Post-processing helpers of the jackknife front-ends (reference: src/utils/utils.cpp), the small host
// operations MultiKtensor's registry needs (ops::), the reconstruction error (error::), the lookup-table
// stub, and the linear-sum-assignment solver the reference takes from extern/rectangular_lsap.
#include <algorithm>
#include <cstring>
#include <limits>
#include <numeric>
#include <stdexcept>

#include "cals.h"

namespace cals {

// ---------------------------------------------------------------------------------------------
// Linear sum assignment.  The reference hands this to SciPy's rectangular_lsap
// (extern/rectangular_lsap/rectangular_lsap.cpp: Crouse's shortest-augmenting-path variant of
// Jonker-Volgenant).  Own implementation of the same published algorithm (D. F. Crouse, "On
// implementing 2D rectangular assignment algorithms", IEEE T-AES 52(4), 2016), including its two
// tie rules -- the unvisited columns are scanned from the last to the first, and among equally short
// paths one that ends in an unassigned column wins -- so that degenerate costs (ties) give the
// assignment the reference gives, not merely one of equal value.  tests: against oracle/_ref's build
// of the reference's own file, scipy.optimize and exhaustive search.
// ---------------------------------------------------------------------------------------------
namespace {
struct LsapSolver {
  int64_t nr, nc;                 // nr <= nc
  std::vector<double> c;          // row-major nr x nc, non-negative, minimisation form
  std::vector<double> u, v, dist;
  std::vector<int64_t> pred, col_of_row, row_of_col, todo;
  std::vector<char> row_seen, col_seen;

  LsapSolver(int64_t nr_, int64_t nc_)
      : nr(nr_), nc(nc_), c((size_t)(nr_ * nc_)), u((size_t)nr_, 0.0), v((size_t)nc_, 0.0), dist((size_t)nc_),
        pred((size_t)nc_, -1), col_of_row((size_t)nr_, -1), row_of_col((size_t)nc_, -1), todo((size_t)nc_),
        row_seen((size_t)nr_), col_seen((size_t)nc_) {}

  // shortest augmenting path from row `start`; returns the free column it ends in (-1: infeasible)
  int64_t grow(int64_t start, double &reach) {
    std::fill(row_seen.begin(), row_seen.end(), 0);
    std::fill(col_seen.begin(), col_seen.end(), 0);
    std::fill(dist.begin(), dist.end(), std::numeric_limits<double>::infinity());
    int64_t n_todo = nc;
    for (int64_t k = 0; k < nc; k++) todo[(size_t)k] = nc - 1 - k;  // last column first
    double level = 0.0;
    int64_t i = start;
    for (;;) {
      row_seen[(size_t)i] = 1;
      int64_t pick = -1;
      double best = std::numeric_limits<double>::infinity();
      for (int64_t k = 0; k < n_todo; k++) {
        const int64_t j = todo[(size_t)k];
        const double via = level + c[(size_t)(i * nc + j)] - u[(size_t)i] - v[(size_t)j];
        if (via < dist[(size_t)j]) {
          dist[(size_t)j] = via;
          pred[(size_t)j] = i;
        }
        if (dist[(size_t)j] < best || (dist[(size_t)j] == best && row_of_col[(size_t)j] < 0)) {
          best = dist[(size_t)j];
          pick = k;
        }
      }
      level = best;
      if (pick < 0 || level == std::numeric_limits<double>::infinity()) return -1;
      const int64_t j = todo[(size_t)pick];
      col_seen[(size_t)j] = 1;
      todo[(size_t)pick] = todo[(size_t)(--n_todo)];
      if (row_of_col[(size_t)j] < 0) {
        reach = level;
        return j;
      }
      i = row_of_col[(size_t)j];
    }
  }

  int run() {
    for (int64_t row = 0; row < nr; row++) {
      double reach = 0.0;
      const int64_t sink = grow(row, reach);
      if (sink < 0) return -1;
      u[(size_t)row] += reach;  // dual update
      for (int64_t i = 0; i < nr; i++)
        if (row_seen[(size_t)i] && i != row) u[(size_t)i] += reach - dist[(size_t)col_of_row[(size_t)i]];
      for (int64_t j = 0; j < nc; j++)
        if (col_seen[(size_t)j]) v[(size_t)j] -= reach - dist[(size_t)j];
      for (int64_t j = sink;;) {  // flip the path
        const int64_t i = pred[(size_t)j];
        row_of_col[(size_t)j] = i;
        std::swap(col_of_row[(size_t)i], j);
        if (i == row) break;
      }
    }
    return 0;
  }
};
}  // namespace

// n x n, column-major cost: col_of_row[i] = column assigned to row i
int solve_linear_sum_assignment(int n, const double *cost, bool maximize, int64_t *col_of_row) {
  if (n < 1 || !cost || !col_of_row) return -2;
  std::vector<double> rm((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) rm[(size_t)i * n + j] = cost[(size_t)i + (size_t)n * j];
  std::vector<int64_t> rows((size_t)n);
  return ::solve_rectangular_linear_sum_assignment(n, n, rm.data(), maximize, rows.data(), col_of_row);
}

namespace mttkrp {
MttkrpLut read_lookup_table(std::vector<dim_t> const &, int, bool, bool) { return {}; }
}  // namespace mttkrp

namespace error {
double compute_error(const Tensor &X, const Ktensor &ktensor) {
  const Tensor R = ktensor.to_tensor();
  double s = 0.0;
  for (dim_t i = 0; i < X.get_n_elements(); i++) s += (X[i] - R[i]) * (X[i] - R[i]);
  return std::sqrt(s);
}
double compute_fast_error(double X_norm, const std::vector<double> &lambda, const Matrix &last_factor,
                          const Matrix &last_mttkrp, const Matrix &gramian_hadamard) {
  double term2 = 0.0;  // sum of lambda_i lambda_j H_ij, column by column (error.cpp:71-74)
  for (dim_t j = 0; j < gramian_hadamard.get_cols(); j++)
    for (dim_t i = 0; i < gramian_hadamard.get_rows(); i++) term2 += lambda[i] * lambda[j] * gramian_hadamard(i, j);
  double term3 = 0.0;  // <A diag(lambda), G> (error.cpp:77-80)
  for (dim_t j = 0; j < last_factor.get_cols(); j++)
    for (dim_t i = 0; i < last_factor.get_rows(); i++) term3 += lambda[j] * last_factor(i, j) * last_mttkrp(i, j);
  return std::sqrt(std::fmax(X_norm * X_norm + term2 - 2 * term3, 0.0));
}
}  // namespace error

namespace ops {
void update_gramian(const Matrix &factor, Matrix &gramian) {  // A^T A, full r x r (utils.cpp:174-178)
  cblas_dgemm(CblasColMajor, CblasTrans, CblasNoTrans, (ptrdiff_t)gramian.get_rows(), (ptrdiff_t)gramian.get_cols(),
              (ptrdiff_t)factor.get_rows(), 1.0, factor.get_data(), (ptrdiff_t)factor.get_col_stride(), factor.get_data(),
              (ptrdiff_t)factor.get_col_stride(), 0.0, gramian.get_data(), (ptrdiff_t)gramian.get_col_stride());
}
void update_gramians(const Ktensor &ktensor, vector<Matrix> &gramians) {
  dim_t n = 0;
  for (const auto &f : ktensor.get_factors()) update_gramian(f, gramians[n++]);
}
Matrix &hadamard_but_one(std::vector<Matrix> &matrices, dim_t mode) {  // utils.cpp:161-172
  Matrix &out = matrices[mode];
  std::fill(out.get_data(), out.get_data() + out.get_n_elements(), 1.0);
  for (dim_t n = 0; n < matrices.size(); n++)
    if (n != mode) out.hadamard(matrices[n]);
  return out;
}
void hadamard_all(std::vector<Matrix> &matrices) {  // utils.cpp:156-159: result in matrices[0]
  for (dim_t n = 1; n < matrices.size(); n++) matrices[0].hadamard(matrices[n]);
}
}  // namespace ops

namespace utils {

std::string mode_string(vector<dim_t> const &modes) {  // src/utils/utils.cpp:9-16
  std::string m;
  for (auto const &v : modes) m += std::to_string(v) + '-';
  if (!m.empty()) m.pop_back();
  return m;
}

Ktensor concatenate_ktensors(vector<Ktensor> const &ktensors) {  // src/utils/utils.cpp:18-38
  const dim_t comp = ktensors[0].get_components();
  Ktensor out(ktensors.size() * comp, ktensors[0].get_modes());
  dim_t index = 0;
  for (auto const &kt : ktensors) {
    for (dim_t i = 0; i < comp; i++) out.get_lambda()[index * comp + i] = kt.get_lambda()[i];
    for (dim_t m = 0; m < kt.get_n_modes(); m++)
      for (dim_t c = 0; c < comp; c++)
        for (dim_t r = 0; r < kt.get_factor(m).get_rows(); r++)
          out.get_factor(m)(r, index * comp + c) = kt.get_factor(m)(r, c);
    index++;
  }
  return out;
}

void generate_jk_ktensors(Ktensor const &reference_ktensor, vector<Ktensor> &jk_ktensor_v) {
  // src/utils/utils.cpp:40-52: one copy per mode-0 slice, flagged jk(mode 0, fiber i)
  const dim_t I0 = reference_ktensor.get_modes()[0];
  if (I0 <= 1) throw std::string("Can't do Jack-knife with just one sample.");
  for (dim_t i = 0; i < I0; i++) {
    Ktensor copy(reference_ktensor);
    copy.to_jk(0, i);
    jk_ktensor_v.push_back(std::move(copy));
  }
}

void jk_permutation_adjustment(Ktensor &ktensor, vector<Ktensor> &jk_ktensor_v) {
  // src/utils/utils.cpp:54-101, restated LITERALLY including its orientation: M(i, j) =
  // <Bov_i, Bm_j> + <Cov_i, Cm_j> (i = column of the overall model, j = column of the replica) is
  // built COLUMN-major and handed to the assignment solver, which reads ROW-major
  // (rectangular_lsap.cpp:93: cost[i * nc + j]).  The solver therefore works on M^T: its row r is
  // replica column r and solved[r] is the overall column matched to it.  The reference then sets
  // new(:, cur) = old(:, solved[cur]) -- the INVERSE of the permutation that would line the replica up
  // with the overall model (that one is new(:, solved[r]) = old(:, r)).  The two coincide when the
  // matching is an involution (identity, swaps: the common case, which is why the reference's own
  // FunctionCorrectness test cannot tell); for a 3-cycle they differ.  Parity with the reference is
  // the contract here, so the call and the copy are kept exactly as the reference has them
  // (DESIGN.md section 5 "Reference quirks kept"; tests/test_lsap_and_jk_permutation.py pins a 3-cycle).
  const auto &modes = ktensor.get_modes();
  const dim_t comp = ktensor.get_components();
  const Matrix &Bov = ktensor.get_factor(1), &Cov = ktensor.get_factor(2);
  for (dim_t m = 0; m < modes[0]; m++) {
    Ktensor &kt = jk_ktensor_v[m];
    const Matrix &Bm = kt.get_factor(1), &Cm = kt.get_factor(2);
    Matrix M(comp, comp);
    for (dim_t j = 0; j < comp; j++)
      for (dim_t i = 0; i < comp; i++) {
        double s = 0.0, t = 0.0;
        for (dim_t r = 0; r < modes[1]; r++) s += Bov(r, i) * Bm(r, j);
        for (dim_t r = 0; r < modes[2]; r++) t += Cov(r, i) * Cm(r, j);
        M(i, j) = s + t;
      }
    std::vector<int64_t> init_v(comp), solved_v(comp);
    ::solve_rectangular_linear_sum_assignment((intptr_t)comp, (intptr_t)comp, M.get_data(), true, init_v.data(),
                                              solved_v.data());
    for (dim_t mode = 0; mode < ktensor.get_n_modes(); mode++) {
      Matrix &f = kt.get_factor(mode);
      Matrix copy(f.get_rows(), f.get_cols());
      copy.copy(f);
      for (dim_t cur = 0; cur < comp; cur++) {
        const dim_t swap = (dim_t)solved_v[cur];
        if (swap != cur)
          for (dim_t r = 0; r < f.get_rows(); r++) f(r, cur) = copy(r, swap);
      }
    }
  }
}

}  // namespace utils

namespace utils {
vector<double> calculate_jackknifing_norms(Tensor const &tensor) {  // utils.cpp:103-152
  const auto &modes = tensor.get_modes();
  const dim_t I = modes[0], cols = tensor.get_n_elements() / std::max<dim_t>(I, 1);
  vector<double> ss(I, 0.0);
  for (dim_t j = 0; j < cols; j++)
    for (dim_t i = 0; i < I; i++) ss[i] += tensor[i + I * j] * tensor[i + I * j];
  const double sum0 = std::accumulate(ss.cbegin(), ss.cend(), 0.0);
  for (auto &v : ss) v = std::sqrt(sum0 - v);
  return ss;
}
}  // namespace utils

}  // namespace cals

// extern/rectangular_lsap/rectangular_lsap.h:44, same contract: row-major nr x nc cost; on return
// (a[k], b[k]), k < min(nr, nc), are the assigned (row, column) pairs sorted by row.  Returns 0,
// -1 (infeasible) or -2 (NaN / -inf entry), the reference's RECTANGULAR_LSAP_* codes.
extern "C" int solve_rectangular_linear_sum_assignment(intptr_t nr, intptr_t nc, double *input_cost,
                                                        bool maximize, int64_t *a, int64_t *b) {
  if (nr == 0 || nc == 0) return 0;
  if (nr < 0 || nc < 0 || !input_cost || !a || !b) return -2;
  const bool tall = nc < nr;  // the solver wants rows <= columns: work on the transpose
  const int64_t R = tall ? nc : nr, Cn = tall ? nr : nc;
  cals::LsapSolver sv(R, Cn);
  for (int64_t i = 0; i < nr; i++)
    for (int64_t j = 0; j < nc; j++) {
      const double x = input_cost[(size_t)(i * nc + j)];
      sv.c[(size_t)(tall ? j * nr + i : i * nc + j)] = maximize ? -x : x;
    }
  double lo = sv.c[0];
  for (double x : sv.c) lo = (x < lo) ? x : lo;
  for (double &x : sv.c) {
    x -= lo;  // non-negative costs
    if (x != x || x == -std::numeric_limits<double>::infinity()) return -2;
  }
  if (sv.run()) return -1;
  if (!tall) {
    for (int64_t i = 0; i < nr; i++) {
      a[i] = i;
      b[i] = sv.col_of_row[(size_t)i];
    }
  } else {  // solver rows are the caller's columns: list the pairs by the caller's row
    std::vector<int64_t> order((size_t)R);
    for (int64_t k = 0; k < R; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(),
              [&](int64_t x, int64_t y) { return sv.col_of_row[(size_t)x] < sv.col_of_row[(size_t)y]; });
    for (int64_t k = 0; k < R; k++) {
      a[k] = sv.col_of_row[(size_t)order[(size_t)k]];
      b[k] = order[(size_t)k];
    }
  }
  return 0;
}

// C entry point of the assignment solver (tests bind it with ctypes)
extern "C" int cals_lsap_solve(int n, const double *cost_colmajor, int maximize, int64_t *col_of_row) {
  return cals::solve_linear_sum_assignment(n, cost_colmajor, maximize != 0, col_of_row);
}

// C entry point of utils::jk_permutation_adjustment (tests bind it with ctypes): `overall` = the n_modes
// factors of the overall model, `replicas` = modes[0] * n_modes factor pointers, replica-major; the
// replicas' columns are reordered in place.
extern "C" int cals_jk_permutation_adjustment(int n_modes, const int64_t *modes, int64_t rank,
                                               const double *const *overall, double *const *replicas) {
  if (n_modes < 3 || !modes || rank < 1 || !overall || !replicas) return -1;
  std::vector<dim_t> md(modes, modes + n_modes);
  cals::Ktensor ov((dim_t)rank, md);
  for (int n = 0; n < n_modes; n++) ov.set_factor(n, overall[n]);
  std::vector<cals::Ktensor> reps;
  for (dim_t m = 0; m < md[0]; m++) {
    reps.emplace_back((dim_t)rank, md);
    for (int n = 0; n < n_modes; n++) reps.back().set_factor(n, replicas[m * n_modes + n]);
  }
  cals::utils::jk_permutation_adjustment(ov, reps);
  for (dim_t m = 0; m < md[0]; m++)
    for (int n = 0; n < n_modes; n++) {
      const cals::Matrix &f = reps[m].get_factor((dim_t)n);
      std::copy(f.get_data(), f.get_data() + f.get_n_elements(), replicas[m * n_modes + n]);
    }
  return 0;
}
