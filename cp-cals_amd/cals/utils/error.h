// include/utils/error.h of HPAC/CP-CALS.  compute_fast_error (src/utils/error.cpp:64-89) is fused into the
// last mode's update_kernel on the device; the reconstruction-based compute_error is offered here as a
// plain host function because callers and tests use it as the ground truth for a model's error.
#ifndef CALS_AMD_UTILS_ERROR_H
#define CALS_AMD_UTILS_ERROR_H

#include "ktensor.h"

namespace cals::error {
// || X - to_tensor(ktensor) ||_F by explicit reconstruction, any number of modes (host loops)
double compute_error(const cals::Tensor &X, const cals::Ktensor &ktensor);
}  // namespace cals::error
#endif
