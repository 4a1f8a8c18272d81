// include/ktensor.h of HPAC/CP-CALS: a CP model -- N factor matrices (I_n x components, column-major),
// the weights lambda, and the state an ALS run leaves in it (iterations, error, fit).  Same members and
// semantics as the reference's class; the numerics of a fit run in libcals_hip.so, the small loops here
// are the conveniences callers use around cp_cals (fill / normalize / to_tensor / jackknife helpers).
#ifndef CALS_AMD_KTENSOR_H
#define CALS_AMD_KTENSOR_H

#include <cfloat>
#include <cmath>
#include <vector>

#include "cals_blas.h"
#include "matrix.h"

using std::multiplies;
using std::vector;

static int universal_ktensor_id = 1;  // include/ktensor.h:14 (one counter per translation unit there too)

namespace cals {

struct JackKniffing {  // sic (include/ktensor.h:18-22)
  bool enabled{false};
  dim_t fiber{0};
  dim_t mode{0};
};

class Ktensor {
  int id{-1};
  dim_t components{0};
  dim_t iters{0};
  double fit{0.0};
  double old_fit{0.0};
  double approx_error{0.0};
  bool normalized{false};
  JackKniffing jk{false, 0, 0};
  // NNLS active sets per mode (rows x components flags).  Built on first use: the device engine keeps its
  // own 64-bit masks in HBM, and 2048 models x 900 rows of vector<bool> would cost more than they serve.
  vector<vector<vector<bool>>> active_set{};
  vector<dim_t> modes{};
  vector<double> lambda{};
  vector<Matrix> factors{};

 public:
  Ktensor() = default;
  ~Ktensor() = default;

  Ktensor(dim_t components_, const vector<dim_t> &modes_)
      : id(universal_ktensor_id++), components(components_), modes(modes_), lambda(components_, 0.0) {
    assert(components > 0);
    factors.reserve(modes.size());
    for (auto m : modes) factors.emplace_back(m, components);
  }
  // jackknife replica: slice `jk_fiber` of mode `jk_mode` is left out (include/ktensor.h:83-88)
  Ktensor(dim_t components_, const vector<dim_t> &modes_, dim_t jk_fiber, dim_t jk_mode = 0)
      : Ktensor(components_, modes_) {
    jk.enabled = true;
    jk.fiber = jk_fiber;
    jk.mode = jk_mode;
  }

  Ktensor(Ktensor &&rhs) = default;
  Ktensor &operator=(Ktensor &&rhs) = default;
  // a copy is a NEW model: fresh id, no run state, active sets reset (include/ktensor.h:95-110)
  Ktensor(const Ktensor &rhs)
      : id(universal_ktensor_id++), components(rhs.components), jk(rhs.jk), modes(rhs.modes), lambda(rhs.lambda),
        factors(rhs.factors) {}
  Ktensor &operator=(const Ktensor &rhs) {
    if (this == &rhs) return *this;
    id = universal_ktensor_id++;
    components = rhs.components;
    lambda = rhs.lambda;
    jk = rhs.jk;
    modes = rhs.modes;
    factors = rhs.factors;
    return *this;
  }

  // ---- getters / setters (include/ktensor.h:134-209) ----
  [[nodiscard]] dim_t get_components() const noexcept { return factors.empty() ? components : factors[0].get_cols(); }
  [[nodiscard]] dim_t get_iters() const noexcept { return iters; }
  [[nodiscard]] int get_id() const noexcept { return id; }
  [[nodiscard]] bool is_jk() const noexcept { return jk.enabled; }
  [[nodiscard]] dim_t get_jk_mode() const noexcept { return jk.mode; }
  [[nodiscard]] dim_t get_jk_fiber() const noexcept { return jk.fiber; }
  [[nodiscard]] double get_approximation_error() const noexcept { return approx_error; }
  [[nodiscard]] double get_fit() const noexcept { return fit; }          // added (the reference keeps fit private)
  [[nodiscard]] double get_old_fit() const noexcept { return old_fit; }  // added
  [[nodiscard]] vector<dim_t> const &get_modes() const noexcept { return modes; }
  vector<Matrix> &get_factors() noexcept { return factors; }
  [[nodiscard]] vector<Matrix> const &get_factors() const noexcept { return factors; }
  [[nodiscard]] vector<double> const &get_lambda() const noexcept { return lambda; }
  vector<double> &get_lambda() noexcept { return lambda; }
  dim_t get_n_modes() const noexcept { return static_cast<dim_t>(factors.size()); }
  Matrix const &get_last_factor() const noexcept { return factors.back(); }
  Matrix const &get_factor(dim_t mode) const noexcept { return factors.at(mode); }
  Matrix &get_factor(dim_t mode) noexcept { return factors.at(mode); }
  vector<vector<bool>> &get_active_set(const dim_t mode) noexcept {
    if (active_set.size() != factors.size()) {
      active_set.clear();
      for (auto &f : factors) active_set.emplace_back(f.get_rows(), vector<bool>(get_components(), true));
    }
    return active_set.at(mode);
  }

  void set_iters(dim_t new_iters) noexcept { iters = new_iters; }
  void set_approximation_error(double new_error) noexcept { approx_error = new_error; }
  void set_fit(double new_fit, double new_old_fit) noexcept {  // added: cp_cals writes the device's values back
    fit = new_fit;
    old_fit = new_old_fit;
  }
  void set_factor(int index, const double *data) noexcept {
    Matrix &target = get_factor(static_cast<dim_t>(index));
    std::copy(data, data + target.get_n_elements(), target.get_data());
  }
  void set_lambda(double const *data) noexcept {
    for (size_t i = 0; i < lambda.size(); i++) lambda[i] = data[i];
  }

  double calculate_new_fit(double X_norm) noexcept {  // include/ktensor.h:178-183
    assert(X_norm != 0);
    old_fit = fit;
    fit = 1 - std::fabs(approx_error) / X_norm;
    return fit;
  }
  [[nodiscard]] double get_fit_diff() const noexcept { return std::fabs(old_fit - fit); }

  void print(const std::string &&text = "Ktensor") const;

  // Point every factor at data_ptrs[n], taking the contents along; detach copies them back into the
  // model's own storage and zeroes the place they leave (src/ktensor.cpp:109-135: how MultiKtensor packs
  // a model into / out of the multi-factor columns).  multi_thread is accepted for compatibility.
  Ktensor &attach(vector<double *> &data_ptrs, bool multi_thread = true);
  Ktensor &detach();

  // all modes: unit 2-norm columns, lambda = product of the norms (src/ktensor.cpp:85-99)
  Ktensor &normalize();
  // one mode as the ALS sweep does it (src/ktensor.cpp:66-83): iteration 1 -> lambda = column 2-norms,
  // later -> lambda = the entry of largest magnitude WITH its sign; zero lambda leaves the column alone
  Ktensor &normalize(dim_t mode, dim_t iteration = 1);
  // fold lambda into factor 0 (src/ktensor.cpp:101-107)
  Ktensor &denormalize();
  Ktensor &randomize();
  Ktensor &fill(function<double()> &&func);
  Tensor to_tensor() const;

  // state, lambda, active sets and factor CONTENTS of rhs; not its id, not its jk flag (src/ktensor.cpp:163-181)
  Ktensor &copy(const Ktensor &rhs);

  Ktensor &to_jk(dim_t mode, dim_t fiber) {
    jk.enabled = true;
    jk.mode = mode;
    jk.fiber = fiber;
    return *this;
  }
  // a jackknife model without its (zeroed) fiber row: one row less in the jk mode (include/ktensor.h:270-303)
  Ktensor to_regular() const;

  // scale the jk fiber row by `value`; NaN marks it (include/ktensor.h:316-325)
  void set_jk_fiber(double value) noexcept {
    if (!jk.enabled) return;
    Matrix &f = get_factor(jk.mode);
    for (dim_t c = 0; c < f.get_cols(); c++) f(jk.fiber, c) = std::isnan(value) ? NAN : f(jk.fiber, c) * value;
  }
};

}  // namespace cals
#endif
