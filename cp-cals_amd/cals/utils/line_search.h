// include/utils/line_search.h of HPAC/CP-CALS: per-model extrapolation every `interval` iterations.
// Both methods run on the device (ls_snapshot_kernel / ls_kernel; ls_ec_* + one extra MTTKRP); the
// struct is what MultiKtensor's registry carries per model.
#ifndef CALS_AMD_UTILS_LINE_SEARCH_H
#define CALS_AMD_UTILS_LINE_SEARCH_H

#include <string>

#include "ktensor.h"

namespace cals {
namespace ls {

enum LS_METHOD { NO_ERROR_CHECKING = 0, ERROR_CHECKING_SERIAL, ERROR_CHECKING_PARALLEL, LENGTH };

static const std::string ls_method_names[LS_METHOD::LENGTH] = {
    "no-error-checking",
    "error-checking-serial",
    "error-checking-parallel",
};

struct LineSearchParams {
  // what the caller configures (CalsParams::line_search_*)
  LS_METHOD method{NO_ERROR_CHECKING};
  int interval{};
  double step{0.0};  // 0 => cbrt(model iteration), src/cals.cpp:317-318
  bool cuda{false};
  // the model's state between sweeps
  int iter{};
  bool updated_last_iter{};
  bool extrapolated{false};
  bool reversed{false};
  Ktensor prev_ktensor{};    // snapshot taken when iter == interval - 1
  Ktensor backup_ktensor{};  // NO_ERROR_CHECKING: what a failed extrapolation reverts to
  Tensor const *T{nullptr};  // ERROR_CHECKING_*: the target tensor
};

}  // namespace ls
}  // namespace cals
#endif
