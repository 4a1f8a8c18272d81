// include/utils/update.h of HPAC/CP-CALS: how a factor is updated from its MTTKRP.  The update itself --
// Hadamard of the other Gramians, Cholesky + two triangular solves (UNCONSTRAINED) or a row-wise
// active-set NNLS -- runs per model inside libcals_hip.so (update_kernel / nnls_kernel); the enum selects it.
#ifndef CALS_AMD_UTILS_UPDATE_H
#define CALS_AMD_UTILS_UPDATE_H

#include <string>

#include "matrix.h"

namespace cals::update {
enum UPDATE_METHOD { UNCONSTRAINED = 0, NNLS, LENGTH };
static const std::string update_method_names[UPDATE_METHOD::LENGTH] = {"unconstrained", "nnls"};
}  // namespace cals::update
#endif
