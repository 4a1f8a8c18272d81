// include/multi_ktensor.h of HPAC/CP-CALS: the packing that makes CALS concurrent -- the factor matrices
// of all in-flight models side by side as column blocks of one wide "multi-factor" per mode.
//
// In the device engine this structure lives in HBM for a whole run (cals_hip_engine: multi-factor
// buffers, per-model Gramians, lambda, line-search copies; cp-cals_amd/csrc/cals_hip_engine.cpp) and its
// bookkeeping -- first-fit column allocator, compress move list, active width -- is exported through the
// C ABI as cals_hip_host_first_fit / _compress_plan / _active_cols.  THIS class is the reference's
// host-side MultiKtensor with the same public interface, built on exactly those three functions, for
// callers that pack models themselves (and as the CPU-checkable statement of the engine's packing rules).
#ifndef CALS_AMD_MULTI_KTENSOR_H
#define CALS_AMD_MULTI_KTENSOR_H

#include <exception>
#include <map>

#include "ktensor.h"
#include "utils/line_search.h"

namespace cals {
// everything the buffer knows about one packed model (include/multi_ktensor.h:12-20)
struct RegistryEntry {
  Ktensor &ktensor;
  vector<Matrix> gramians;
  int col;
  dim_t id;
  ls::LineSearchParams ls_params{};
};
typedef std::map<int, RegistryEntry> Registry;

struct BufferFull : public std::exception {  // include/multi_ktensor.h:123-127
  [[nodiscard]] const char *what() const noexcept override {
    return "Buffer is full, wait until some ktensors converge.";
  }
};

class MultiKtensor : public Ktensor {
  // packing state (own layout; the public interface below is the reference's)
  Registry packed_;                  // in-flight models by id
  vector<dim_t> col_owner_;          // id of the model in every column, 0 = free
  vector<dim_t> mode_sizes_;
  dim_t next_id_{1};
  int cols_in_use_{0};
  int first_col_{0}, width_{0};      // active window of the buffer: columns [first_col_, width_)
  bool any_jk_ = false;
  // what add() copies into a model's RegistryEntry
  bool device_{false};
  bool with_line_search_{false};
  ls::LineSearchParams ls_defaults_{};

  int check_availability(Ktensor &ktensor);  // first fit; throws BufferFull
  MultiKtensor &adjust_edges();

 public:
  MultiKtensor() = default;
  ~MultiKtensor() = default;
  MultiKtensor &operator=(MultiKtensor &&mk) = default;

  // empty buffer of buffer_size columns per mode (src/multi_ktensor.cpp:8-12)
  explicit MultiKtensor(vector<dim_t> &modes, dim_t buffer_size);

  // pack a model: its factors move into the first run of free columns that fits and point there; N
  // Gramians are formed; iters = 1 (src/multi_ktensor.cpp:41-130).  Throws BufferFull.
  MultiKtensor &add(Ktensor &ktensor);
  // unpack: factors back into the model's own storage, columns zeroed and freed (src/multi_ktensor.cpp:132-163)
  MultiKtensor &remove(dim_t ktensor_id);
  // shift the surviving models left over the gaps (src/multi_ktensor.cpp:188-264)
  MultiKtensor &compress();

  Registry &get_registry() { return packed_; }
  [[nodiscard]] int get_start() const noexcept { return first_col_; }
  [[nodiscard]] bool get_flag_jk() const noexcept { return any_jk_; }
  void set_cuda(bool value) { device_ = value; }
  void set_line_search(bool value) { with_line_search_ = value; }
  void set_line_search_params(ls::LineSearchParams &params) { ls_defaults_ = params; }
  [[maybe_unused]] vector<dim_t> &get_modes() { return mode_sizes_; }
  int get_leftmost_id() { return col_owner_.empty() ? -1 : static_cast<int>(col_owner_[0]); }
};
}  // namespace cals
#endif
