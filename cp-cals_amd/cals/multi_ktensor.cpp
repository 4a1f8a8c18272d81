// MultiKtensor on the host (reference: src/multi_ktensor.cpp), over the packing rules the device engine
// exports through the C ABI: cals_hip_host_first_fit (check_availability, :14-39),
// cals_hip_host_compress_plan (the move list of compress, :196-209) and cals_hip_host_active_cols
// (adjust_edges, :165-186).  The engine applies the same three functions to its HBM-resident buffers.
#include "multi_ktensor.h"

#include <cstdint>
#include <cstring>

#include "../../include/cals_hip.h"
#include "utils/utils.h"

namespace cals {

namespace {
std::vector<int64_t> as_i64(const vector<dim_t> &v) { return std::vector<int64_t>(v.begin(), v.end()); }
}  // namespace

MultiKtensor::MultiKtensor(vector<dim_t> &modes_, dim_t buffer_size)
    : Ktensor(buffer_size, modes_), col_owner_(buffer_size, 0), mode_sizes_(modes_) {
  for (auto &f : get_factors()) {
    f.zero();  // free columns read as zero (the engine's buffers start zeroed too)
    f.resize(f.get_rows(), 0);
  }
}

int MultiKtensor::check_availability(Ktensor &ktensor) {
  const auto occ = as_i64(col_owner_);
  const int64_t pos = cals_hip_host_first_fit(occ.data(), (int64_t)occ.size(), (int64_t)ktensor.get_components());
  if (pos < 0) throw BufferFull();
  return static_cast<int>(pos);
}

MultiKtensor &MultiKtensor::add(Ktensor &ktensor) {
  const int pos = check_availability(ktensor);  // BufferFull propagates
  vector<double *> where(ktensor.get_n_modes());
  dim_t n = 0;
  for (auto &f : get_factors()) where[n++] = f.reset_data().get_data() + (dim_t)pos * f.get_col_stride();
  ktensor.attach(where);

  const dim_t id = next_id_++;
  const dim_t r = ktensor.get_components();
  for (dim_t c = 0; c < r; c++) col_owner_[(dim_t)pos + c] = id;
  cols_in_use_ += static_cast<int>(r);

  vector<Matrix> gramians(ktensor.get_n_modes());
  for (auto &g : gramians) g = Matrix(r, r);
  ops::update_gramians(ktensor, gramians);
  ktensor.set_iters(1);
  if (ktensor.is_jk()) any_jk_ = true;

  RegistryEntry entry{ktensor, std::move(gramians), pos, id};
  if (with_line_search_) {
    entry.ls_params.prev_ktensor = Ktensor(r, mode_sizes_);
    entry.ls_params.backup_ktensor = Ktensor(r, mode_sizes_);
    entry.ls_params.cuda = device_;
    entry.ls_params.interval = ls_defaults_.interval;
    entry.ls_params.step = ls_defaults_.step;
    entry.ls_params.method = ls_defaults_.method;
    entry.ls_params.T = ls_defaults_.T;
  }
  packed_.insert(std::pair<int, RegistryEntry>(static_cast<int>(id), std::move(entry)));
  return adjust_edges();
}

MultiKtensor &MultiKtensor::remove(dim_t ktensor_id) {
  RegistryEntry &entry = packed_.at(static_cast<int>(ktensor_id));
  Ktensor &kt = entry.ktensor;
  kt.detach();  // contents back into the model's own storage, the columns zeroed
  for (auto &cell : col_owner_)
    if (cell == ktensor_id) cell = 0;
  cols_in_use_ -= static_cast<int>(kt.get_components());
  packed_.erase(static_cast<int>(ktensor_id));
  return adjust_edges();
}

MultiKtensor &MultiKtensor::adjust_edges() {
  first_col_ = 0;
  const auto occ = as_i64(col_owner_);
  width_ = static_cast<int>(cals_hip_host_active_cols(occ.data(), (int64_t)occ.size()));
  for (auto &f : get_factors()) {
    f.set_data(f.reset_data().get_data() + (dim_t)first_col_ * f.get_col_stride());
    f.resize(f.get_rows(), static_cast<dim_t>(width_ - first_col_));
  }
  return *this;
}

MultiKtensor &MultiKtensor::compress() {
  const auto occ = as_i64(col_owner_);
  std::vector<int64_t> ids(occ.size()), offs(occ.size());
  const int64_t n_moves =
      cals_hip_host_compress_plan(occ.data(), (int64_t)occ.size(), ids.data(), offs.data(), (int64_t)occ.size());
  vector<double *> where(get_n_modes());
  for (int64_t k = 0; k < n_moves; k++) {  // left to right, as the reference applies them
    RegistryEntry &entry = packed_.at(static_cast<int>(ids[(size_t)k]));
    Ktensor &kt = entry.ktensor;
    const dim_t off = (dim_t)offs[(size_t)k];
    dim_t n = 0;
    for (auto &f : kt.get_factors()) where[n++] = f.get_data() - off * f.get_col_stride();
    kt.attach(where, kt.get_components() < off);  // overlapping moves are handled inside attach (memmove)
    for (dim_t c = (dim_t)entry.col; c < (dim_t)entry.col + kt.get_components(); c++)
      std::swap(col_owner_[c - off], col_owner_[c]);
    entry.col -= static_cast<int>(off);
  }
  return adjust_edges();
}

}  // namespace cals
