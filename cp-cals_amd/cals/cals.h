// C++ host layer over the C ABI (include/cals_hip.h): the value classes and the cp_cals() entry
// point of HPAC/CP-CALS with the same names, argument meaning and ownership rules, so that a caller
// written against the reference's headers (include/cals.h, tensor.h, matrix.h, ktensor.h) compiles
// against this one.  Only what the hot path's callers touch is provided.
// Everything numeric on the path runs in libcals_hip.so; the small host loops below (fill,
// normalize, to_tensor ...) are the Ktensor conveniences the callers use around cp_cals.
#ifndef CALS_AMD_CALS_H
#define CALS_AMD_CALS_H

#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iostream>
#include <memory>
#include <queue>
#include <random>
#include <stdexcept>
#include <string>
#include <fstream>
#include <sstream>
#include <vector>

typedef size_t dim_t;  // include/definitions.h:18

namespace cals {
using std::vector;

namespace update {  // include/utils/update.h:7
enum UPDATE_METHOD { UNCONSTRAINED = 0, NNLS, LENGTH };
}
namespace mttkrp {  // include/utils/mttkrp.h:23-31 (kept for source compatibility; one fused kernel here)
enum MTTKRP_METHOD { MTTKRP = 0, TWOSTEP0, TWOSTEP1, AUTO, LENGTH };
// include/utils/mttkrp.h:15-19, 100-101.  The lookup tables choose among the reference's CPU/CUDA
// MTTKRP variants per (mode, rank, threads); the device engine picks its own plan (cals_hip_tree), so
// the table is accepted and ignored and read_lookup_table returns an empty one.
struct MttkrpLut {
  std::vector<std::vector<int>> lut_v{};
  std::vector<int> keys_v{};
};
inline MttkrpLut read_lookup_table(std::vector<dim_t> const &, int, bool = false, bool = false) { return {}; }
}
namespace ls {  // include/utils/line_search.h:8
enum LS_METHOD { NO_ERROR_CHECKING = 0, ERROR_CHECKING_SERIAL, ERROR_CHECKING_PARALLEL, LENGTH };
}

// include/tensor.h: column-major dense tensor, owning buffer or view
class Tensor {
 protected:
  dim_t n_elements{0};
  vector<dim_t> modes{};
  std::unique_ptr<double[]> owned{};
  double *data{nullptr};
  int rank{0};

 public:
  Tensor() = default;
  explicit Tensor(const vector<dim_t> &modes_) : modes(modes_) {
    n_elements = 1;
    for (auto m : modes) n_elements *= m;
    owned.reset(new double[n_elements]);
    data = owned.get();
  }
  Tensor(const vector<dim_t> &modes_, double *view) : modes(modes_), data(view) {  // tensor.cpp:28-33
    n_elements = 1;
    for (auto m : modes) n_elements *= m;
  }
  Tensor(dim_t rows, dim_t cols, double *view = nullptr) : modes{rows, cols} {
    n_elements = rows * cols;
    if (view)
      data = view;
    else {
      owned.reset(new double[n_elements]);
      data = owned.get();
    }
  }
  // text file: first line = mode sizes separated by blanks, then one value per line, mode 0
  // fastest (src/tensor.cpp:35-65)
  explicit Tensor(const std::string &file_name);
  Tensor(Tensor &&) = default;
  Tensor &operator=(Tensor &&) = default;
  Tensor(const Tensor &rhs) : n_elements(rhs.n_elements), modes(rhs.modes), rank(rhs.rank) {
    if (rhs.is_view())
      data = rhs.data;
    else {
      owned.reset(new double[n_elements]);
      data = owned.get();
      std::copy(rhs.data, rhs.data + n_elements, data);
    }
  }
  Tensor &operator=(const Tensor &rhs) {
    if (this != &rhs) {
      Tensor t(rhs);
      *this = std::move(t);
    }
    return *this;
  }
  virtual ~Tensor() = default;

  dim_t get_n_elements() const noexcept { return n_elements; }
  const vector<dim_t> &get_modes() const noexcept { return modes; }
  dim_t get_n_modes() const noexcept { return modes.size(); }
  double *get_data() const noexcept { return data; }
  int get_rank() const noexcept { return rank; }
  void set_rank(int r) noexcept { rank = r; }
  bool is_view() const noexcept { return owned == nullptr && data != nullptr; }
  double &operator[](dim_t i) noexcept { return data[i]; }
  double operator[](dim_t i) const noexcept { return data[i]; }
  double norm() const {  // include/tensor.h:196
    double s = 0.0;
    for (dim_t i = 0; i < n_elements; i++) s += data[i] * data[i];
    return std::sqrt(s);
  }
  Tensor &fill(const std::function<double()> &&gen) {  // tensor.cpp:137-141
    for (dim_t i = 0; i < n_elements; i++) data[i] = gen();
    return *this;
  }
  Tensor &zero() {
    for (dim_t i = 0; i < n_elements; i++) data[i] = 0.0;
    return *this;
  }
  Tensor &randomize() {  // tensor.cpp:122-130
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::random_device device;
    std::mt19937 generator(device());
    return fill([&]() { return dist(generator); });
  }
  Tensor &copy(const Tensor &rhs) {
    std::copy(rhs.data, rhs.data + n_elements, data);
    return *this;
  }
};

// include/matrix.h
class Matrix : public Tensor {
  dim_t rows{0}, cols{0}, col_stride{0};

 public:
  Matrix() = default;
  Matrix(dim_t r, dim_t c) : Tensor(r, c), rows(r), cols(c), col_stride(r) {}
  Matrix(dim_t r, dim_t c, double *view) : Tensor(r, c, view), rows(r), cols(c), col_stride(r) {}
  dim_t get_rows() const noexcept { return rows; }
  dim_t get_cols() const noexcept { return cols; }
  dim_t get_col_stride() const noexcept { return col_stride; }
  double &operator()(dim_t r, dim_t c) noexcept { return data[r + c * col_stride]; }
  double operator()(dim_t r, dim_t c) const noexcept { return data[r + c * col_stride]; }
};

// include/ktensor.h
class Ktensor {
  int id{-1};
  dim_t components{0}, iters{0};
  double fit{0.0}, old_fit{0.0}, approx_error{0.0};
  bool normalized{false};
  struct { bool enabled{false}; dim_t fiber{0}; dim_t mode{0}; } jk;
  vector<dim_t> modes{};
  vector<double> lambda{};
  vector<Matrix> factors{};
  static int &next_id() { static int v = 1; return v; }

 public:
  Ktensor() = default;
  Ktensor(dim_t components_, const vector<dim_t> &modes_)
      : id(next_id()++), components(components_), modes(modes_), lambda(components_, 0.0) {
    for (auto m : modes) factors.emplace_back(m, components);
  }
  Ktensor(dim_t components_, const vector<dim_t> &modes_, dim_t jk_fiber, dim_t jk_mode = 0)
      : Ktensor(components_, modes_) {
    jk.enabled = true;
    jk.fiber = jk_fiber;
    jk.mode = jk_mode;
  }
  Ktensor(Ktensor &&) = default;
  Ktensor &operator=(Ktensor &&) = default;
  Ktensor(const Ktensor &rhs)  // a copy gets a fresh id (include/ktensor.h:95-110)
      : id(next_id()++), components(rhs.components), jk(rhs.jk), modes(rhs.modes), lambda(rhs.lambda),
        factors(rhs.factors) {}
  Ktensor &operator=(const Ktensor &rhs) {
    if (this != &rhs) {
      id = next_id()++;
      components = rhs.components;
      lambda = rhs.lambda;
      jk = rhs.jk;
      modes = rhs.modes;
      factors = rhs.factors;
    }
    return *this;
  }

  dim_t get_components() const noexcept { return components; }
  dim_t get_iters() const noexcept { return iters; }
  int get_id() const noexcept { return id; }
  bool is_jk() const noexcept { return jk.enabled; }
  dim_t get_jk_mode() const noexcept { return jk.mode; }
  dim_t get_jk_fiber() const noexcept { return jk.fiber; }
  double get_approximation_error() const noexcept { return approx_error; }
  double get_fit() const noexcept { return fit; }
  const vector<dim_t> &get_modes() const noexcept { return modes; }
  dim_t get_n_modes() const noexcept { return factors.size(); }
  vector<Matrix> &get_factors() noexcept { return factors; }
  const vector<Matrix> &get_factors() const noexcept { return factors; }
  Matrix &get_factor(dim_t m) noexcept { return factors.at(m); }
  const Matrix &get_factor(dim_t m) const noexcept { return factors.at(m); }
  const Matrix &get_last_factor() const noexcept { return factors.back(); }
  vector<double> &get_lambda() noexcept { return lambda; }
  const vector<double> &get_lambda() const noexcept { return lambda; }
  void set_iters(dim_t v) noexcept { iters = v; }
  // include/ktensor.h:161-170
  void set_factor(int index, const double *src) noexcept {
    Matrix &t = get_factor((dim_t)index);
    std::copy(src, src + t.get_n_elements(), t.get_data());
  }
  void set_lambda(double const *src) noexcept {
    for (size_t i = 0; i < lambda.size(); i++) lambda[i] = src[i];
  }
  void set_approximation_error(double v) noexcept { approx_error = v; }
  void set_fit(double f, double of) noexcept { fit = f; old_fit = of; }
  double calculate_new_fit(double X_norm) noexcept {  // include/ktensor.h:178-183
    old_fit = fit;
    fit = 1 - std::fabs(approx_error) / X_norm;
    return fit;
  }
  double get_fit_diff() const noexcept { return std::fabs(old_fit - fit); }

  Ktensor &normalize() {  // ktensor.cpp:85-99
    for (auto &l : lambda) l = 1.0;
    for (auto &f : factors)
      for (dim_t c = 0; c < components; c++) {
        double s = 0.0;
        for (dim_t i = 0; i < f.get_rows(); i++) s += f(i, c) * f(i, c);
        const double coeff = std::sqrt(s), inv = 1 / coeff;
        for (dim_t i = 0; i < f.get_rows(); i++) f(i, c) *= inv;
        lambda[c] *= coeff;
      }
    normalized = true;
    return *this;
  }
  Ktensor &denormalize() {  // ktensor.cpp:101-107
    auto &f = factors[0];
    for (dim_t c = 0; c < components; c++)
      for (dim_t i = 0; i < f.get_rows(); i++) f(i, c) *= lambda[c];
    normalized = false;
    return *this;
  }
  void set_jk_fiber(double value) noexcept {  // include/ktensor.h:316-325
    if (!jk.enabled) return;
    auto &f = factors[jk.mode];
    for (dim_t c = 0; c < components; c++)
      f(jk.fiber, c) = std::isnan(value) ? NAN : f(jk.fiber, c) * value;
  }
  Ktensor &fill(std::function<double()> &&func) {  // ktensor.cpp:21-30
    for (auto &f : factors) f.fill(std::forward<decltype(func)>(func));
    if (jk.enabled) set_jk_fiber(0.0);
    return normalize();
  }
  Ktensor &randomize() {  // ktensor.cpp:11-19
    for (auto &f : factors) f.randomize();
    if (jk.enabled) set_jk_fiber(0.0);
    return normalize();
  }
  Ktensor &to_jk(dim_t mode, dim_t fiber) {
    jk.enabled = true;
    jk.mode = mode;
    jk.fiber = fiber;
    return *this;
  }
  // include/ktensor.h:270-303: the jk model without its fiber row (one row less in the jk mode)
  Ktensor to_regular() const {
    if (!jk.enabled) return *this;
    vector<dim_t> reg_modes(modes);
    reg_modes[jk.mode] -= 1;
    Ktensor out(components, reg_modes);
    for (dim_t f = 0; f < modes.size(); f++)
      for (dim_t c = 0; c < components; c++)
        for (dim_t i = 0, o = 0; i < modes[f]; i++) {
          if (f == jk.mode && i == jk.fiber) continue;
          out.factors[f](o++, c) = factors[f](i, c);
        }
    out.lambda = lambda;
    return out;
  }
  // Ktensor::copy (src/ktensor.cpp:163-181): state, lambda and factors; not id, not jk
  Ktensor &copy(const Ktensor &rhs) {
    approx_error = rhs.approx_error;
    fit = rhs.fit;
    old_fit = rhs.old_fit;
    iters = rhs.iters;
    normalized = rhs.normalized;
    lambda = rhs.lambda;
    for (dim_t f = 0; f < factors.size(); f++) factors[f].copy(rhs.factors[f]);
    return *this;
  }
  Tensor to_tensor() const {  // ktensor.cpp:32-64
    Tensor X(modes);
    vector<dim_t> idx(modes.size(), 0);
    for (dim_t e = 0; e < X.get_n_elements(); e++) {
      double s = 0.0;
      for (dim_t r = 0; r < components; r++) {
        double m = 1.0;
        for (dim_t f = 0; f < factors.size(); f++) m *= factors[f](idx[f], r);
        s += lambda[r] * m;
      }
      X[e] = s;
      for (dim_t n = 0; n < modes.size(); n++) {
        if (++idx[n] < modes[n]) break;
        idx[n] = 0;
      }
    }
    return X;
  }
};

typedef std::queue<std::reference_wrapper<Ktensor>> KtensorQueue;  // include/cals.h:22

// include/cals.h:27-63 (timer matrices are not produced: hipEvent statistics are available
// through cals_hip_get_kernel_stats instead)
struct CalsReport {
  int tensor_rank{0};
  dim_t n_modes{0};
  vector<dim_t> modes{};
  double X_norm{0.0};
  dim_t iter{0};
  dim_t max_iter{0};
  int n_threads{1};
  dim_t buffer_size{0};
  int n_ktensors{0};
  int ktensor_comp_sum{0};
  double tol{0.0};
  bool cuda{true};
  update::UPDATE_METHOD update_method{update::UNCONSTRAINED};
  std::string output_file_name{};
  bool line_search{false};
  int line_search_interval{0};
  double line_search_step{0.0};
  dim_t ls_performed{0};
  dim_t ls_failed{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  double total_time{0.0};
};

// include/cals.h:138-159.  `cuda` selects the device path in the reference; this library HAS only
// the device path (MI355X), so it defaults to true and cp_cals throws if it is false.
struct CalsParams {
  update::UPDATE_METHOD update_method{update::UNCONSTRAINED};
  mttkrp::MTTKRP_METHOD mttkrp_method{mttkrp::AUTO};
  mttkrp::MttkrpLut mttkrp_lut{};  // accepted and ignored (see read_lookup_table)
  dim_t max_iterations{200};
  double tol{1e-7};
  bool cuda{true};
  dim_t buffer_size{4200};
  bool line_search{false};
  int line_search_interval{5};
  double line_search_step{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  bool force_max_iter{false};
  bool always_evict_first{false};
  int device{0};  // added: HIP device ordinal (default preserves single-GPU behaviour)
  // added: more than one ordinal = one engine per listed device inside this process, the queue shared
  // through an atomic counter (each model is fitted by exactly one device); claim_models = how many
  // models a device takes at a time when none of its own is waiting for buffer columns
  std::vector<int> devices{};
  int claim_models{8};
  // added: storage/arithmetic type on the device, FP64 (the reference's) or FP32 (BASELINE config 4:
  // fp32 tensor copies, factors and MFMA; Gramians, solves, lambda, error stay fp64)
  enum PRECISION { FP64 = 0, FP32 = 1 };
  PRECISION precision{FP64};
  void print() const;
};

// include/timer.h
class Timer {
  double t0{0.0}, elapsed{0.0};
  static double now();

 public:
  void start() { t0 = now(); }
  void stop() { elapsed = now() - t0; }
  void reset() { elapsed = 0.0; }
  double get_time() const { return elapsed; }
};

// include/als.h:142-166 / :27-63 (fields on the path).  cp_als here is the same device engine with
// a single model in flight (the reference's tests demand CALS == ALS per model anyway).
struct AlsParams {
  update::UPDATE_METHOD update_method{update::UNCONSTRAINED};
  mttkrp::MTTKRP_METHOD mttkrp_method{mttkrp::AUTO};
  mttkrp::MttkrpLut mttkrp_lut{};
  dim_t max_iterations{200};
  double tol{1e-7};
  bool cuda{true};
  bool line_search{false};
  int line_search_interval{5};
  double line_search_step{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  bool force_max_iter{false};
  bool suppress_lut_warning{false};
  int device{0};
  void print() const;
};
struct AlsReport {
  dim_t iter{0};
  dim_t ls_performed{0}, ls_failed{0};
  double X_norm{0.0};
  double total_time{0.0};
};
AlsReport cp_als(const Tensor &X, Ktensor &ktensor, AlsParams &als_params);
// include/als.h:218: the reference's OpenMP-over-models ALS; here all models run concurrently on the device
vector<AlsReport> cp_omp_als(const Tensor &X, vector<Ktensor> &ktensor, AlsParams &params);

// include/als.h:22-25, 168-170
struct JKTime {
  double pre_als_time{0.0};
  double als_time{0.0};
};
struct JKReport {
  JKTime jk_time;
  vector<vector<Ktensor>> results;
};

namespace utils {  // include/utils/utils.h:17-23
std::string mode_string(vector<dim_t> const &modes);
Ktensor concatenate_ktensors(vector<Ktensor> const &ktensors);
void generate_jk_ktensors(Ktensor const &reference_ktensor, vector<Ktensor> &jk_ktensor_v);
void jk_permutation_adjustment(Ktensor &ktensor, vector<Ktensor> &jk_ktensor_v);
}  // namespace utils

// Linear sum assignment on an n x n COLUMN-major cost matrix: col_of_row[i] = column assigned to row i
// (convenience over solve_rectangular_linear_sum_assignment below, which reads row-major).
int solve_linear_sum_assignment(int n, const double *cost_colmajor, bool maximize, int64_t *col_of_row);
}  // namespace cals
// extern/rectangular_lsap/rectangular_lsap.h:44 (utils.cpp:79 and the MEX front-ends call it directly):
// ROW-major nr x nc cost, any shape; (a[k], b[k]) = assigned (row, column) pairs sorted by row.  Own
// implementation of the same algorithm (Crouse's shortest augmenting paths) with the same tie rules.
extern "C" int solve_rectangular_linear_sum_assignment(intptr_t nr, intptr_t nc, double *input_cost,
                                                        bool maximize, int64_t *a, int64_t *b);
namespace cals {

inline void set_threads(int) {}   // include/cals_blas.h:184-186: host BLAS threads; no meaning here
inline int get_threads() { return 1; }

// Jackknife driver (src/cals.cpp:397-446): for every model of kt_vector, modes[0] jackknife
// replicas fitted in ONE cp_cals call, then re-normalised and column-matched to the original.
JKReport jk_cp_cals(const Tensor &X, vector<Ktensor> &kt_vector, CalsParams &cals_params);
// include/als.h:203,220: jackknife by ALS on the sub-sampled tensors (the comparator of jk_cp_cals)
JKReport jk_cp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &als_params);
JKReport jk_cp_omp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &als_params);

// Fits every Ktensor of the queue to X with concurrent ALS on the GPU and overwrites it with the
// result (factors, lambda, error, fit, iters); the queue is empty on return (include/cals.h:183-196).
// Throws std::runtime_error on any engine error (the reference exit()s on device errors).
CalsReport cp_cals(const Tensor &X, KtensorQueue &kt_queue, CalsParams &cals_params);

}  // namespace cals
#endif
