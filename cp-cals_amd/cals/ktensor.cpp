// Ktensor members (reference: src/ktensor.cpp).  Host conveniences only -- what runs inside a fit is in
// libcals_hip.so; normalize(mode, iteration) restates the per-mode normalisation of the ALS sweep
// (src/ktensor.cpp:66-83) for callers that drive a model by hand, with cblas_idamax's first-index tie rule.
#include <cstring>
#include <iostream>

#include "ktensor.h"

namespace cals {

Ktensor &Ktensor::randomize() {
  for (auto &f : factors) f.randomize();
  if (jk.enabled) set_jk_fiber(0.0);
  return normalize();
}

Ktensor &Ktensor::fill(function<double()> &&func) {
  for (auto &f : factors) f.fill(std::forward<decltype(func)>(func));  // factor 0..N-1, each column-major
  if (jk.enabled) set_jk_fiber(0.0);
  return normalize();
}

Tensor Ktensor::to_tensor() const {
  vector<dim_t> dims(get_n_modes());
  for (dim_t n = 0; n < get_n_modes(); n++) dims[n] = factors[n].get_rows();
  Tensor X(dims);
  vector<dim_t> idx(dims.size(), 0);  // odometer over the elements, mode 0 fastest
  const dim_t r = get_components();
  for (dim_t e = 0; e < X.get_n_elements(); e++) {
    double s = 0.0;
    for (dim_t c = 0; c < r; c++) {
      double m = 1.0;
      for (dim_t n = 0; n < dims.size(); n++) m *= factors[n](idx[n], c);
      s += lambda[c] * m;
    }
    X[e] = s;
    for (dim_t n = 0; n < dims.size(); n++) {
      if (++idx[n] < dims[n]) break;
      idx[n] = 0;
    }
  }
  return X;
}

Ktensor &Ktensor::normalize(dim_t mode, dim_t iteration) {
  Matrix &f = factors.at(mode);
  for (dim_t c = 0; c < f.get_cols(); c++) {
    double *col = f.get_data() + c * f.get_col_stride();
    if (iteration == 1)
      lambda[c] = cblas_dnrm2((ptrdiff_t)f.get_rows(), col, 1);
    else
      lambda[c] = col[cblas_idamax((ptrdiff_t)f.get_rows(), col, 1)];  // signed
    if (lambda[c] != 0) cblas_dscal((ptrdiff_t)f.get_rows(), 1 / lambda[c], col, 1);
  }
  return *this;
}

Ktensor &Ktensor::normalize() {
  for (auto &l : lambda) l = 1.0;
  for (auto &f : factors)
    for (dim_t c = 0; c < get_components(); c++) {
      double *col = f.get_data() + c * f.get_col_stride();
      const double coeff = cblas_dnrm2((ptrdiff_t)f.get_rows(), col, 1);
      cblas_dscal((ptrdiff_t)f.get_rows(), 1 / coeff, col, 1);
      lambda[c] *= coeff;
    }
  normalized = true;
  return *this;
}

Ktensor &Ktensor::denormalize() {
  Matrix &f = factors[0];
  for (dim_t c = 0; c < get_components(); c++)
    cblas_dscal((ptrdiff_t)f.get_rows(), lambda[c], f.get_data() + c * f.get_col_stride(), 1);
  normalized = false;
  return *this;
}

Ktensor &Ktensor::attach(vector<double *> &data_ptrs, bool) {
  assert(data_ptrs.size() == get_n_modes());
  dim_t n = 0;
  for (auto &f : factors) {
    double *dst = data_ptrs[n++];
    // source and destination overlap when compress shifts a model by less than its own width
    // (src/multi_ktensor.cpp:226-229): memmove semantics
    std::memmove(dst, f.get_data(), sizeof(double) * f.get_n_elements());
    f.attach(dst);
  }
  return *this;
}

Ktensor &Ktensor::detach() {
  for (auto &f : factors) {
    double *packed = f.get_data();
    f.detach();
    if (packed == f.get_data()) continue;  // was not attached
    std::copy(packed, packed + f.get_n_elements(), f.get_data());
    std::fill(packed, packed + f.get_n_elements(), 0.0);
  }
  return *this;
}

void Ktensor::print(const std::string &&text) const {
  using std::cout;
  using std::endl;
  cout << "----------------------------------------" << endl << text << endl;
  cout << "----------------------------------------" << endl;
  cout << "Rank: " << get_components() << endl << "Num Modes: " << get_n_modes() << endl << "Modes: [ ";
  for (const auto &f : factors) cout << f.get_rows() << " ";
  cout << "]" << endl << "Weights: [";
  for (const auto &l : lambda) cout << l << " ";
  cout << " ] " << endl;
  for (const auto &f : factors) f.print("factor");
  cout << "----------------------------------------" << endl;
}

Ktensor &Ktensor::copy(const Ktensor &rhs) {
  approx_error = rhs.approx_error;
  fit = rhs.fit;
  old_fit = rhs.old_fit;
  iters = rhs.iters;
  normalized = rhs.normalized;
  lambda = rhs.lambda;
  active_set = rhs.active_set;
  for (dim_t n = 0; n < factors.size(); n++) factors[n].copy(rhs.get_factor(n));
  return *this;
}

Ktensor Ktensor::to_regular() const {
  if (!jk.enabled) return *this;
  vector<dim_t> reg_modes(modes);
  reg_modes[jk.mode] -= 1;
  Ktensor out(get_components(), reg_modes);
  for (dim_t n = 0; n < modes.size(); n++) {
    const Matrix &src = factors[n];
    Matrix &dst = out.get_factor(n);
    for (dim_t c = 0; c < src.get_cols(); c++)
      for (dim_t i = 0, o = 0; i < src.get_rows(); i++) {
        if (n == jk.mode && i == jk.fiber) continue;
        dst(o++, c) = src(i, c);
      }
  }
  out.get_lambda() = lambda;
  return out;
}

}  // namespace cals
