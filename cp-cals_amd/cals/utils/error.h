// include/utils/error.h of HPAC/CP-CALS.  compute_fast_error (src/utils/error.cpp:64-89) is fused into the
// last mode's update_kernel on the device; the reconstruction-based compute_error is offered here as a
// plain host function because callers and tests use it as the ground truth for a model's error.
#ifndef CALS_AMD_UTILS_ERROR_H
#define CALS_AMD_UTILS_ERROR_H

#include "ktensor.h"

#include <vector>

namespace cals::error {
// || X - to_tensor(ktensor) ||_F by explicit reconstruction, any number of modes (host loops)
double compute_error(const cals::Tensor &X, const cals::Ktensor &ktensor);
// The FastALS error identity on matrices the CALLER holds (include/utils/error.h:22-26, src/utils/error.cpp:64-89):
// sqrt(max(||X||^2 + sum_ij l_i l_j H_ij - 2 sum_ij l_j A_ij G_ij, 0)).  The engine never calls this: its own
// error is evaluated inside the last mode's update_kernel while G is still in registers.
double compute_fast_error(double X_norm, const std::vector<double> &lambda, const cals::Matrix &last_factor,
                          const cals::Matrix &last_mttkrp, const cals::Matrix &gramian_hadamard);
}  // namespace cals::error
#endif
