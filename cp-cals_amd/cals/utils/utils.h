// include/utils/utils.h of HPAC/CP-CALS: the post-processing helpers of the jackknife front-ends.
#ifndef CALS_AMD_UTILS_UTILS_H
#define CALS_AMD_UTILS_UTILS_H

#include <string>

#include "ktensor.h"

namespace cals::utils {
std::string mode_string(std::vector<dim_t> const &modes);  // "I-J-K"

// one wide Ktensor holding the columns of all inputs side by side (src/utils/utils.cpp:18-38)
cals::Ktensor concatenate_ktensors(std::vector<cals::Ktensor> const &ktensors);

// modes[0] copies of the model, copy i flagged jk(mode 0, fiber i) (src/utils/utils.cpp:40-52)
void generate_jk_ktensors(cals::Ktensor const &reference_ktensor, std::vector<cals::Ktensor> &jk_ktensor_v);

// reorder every replica's columns after the overall model's (src/utils/utils.cpp:54-101) -- with the
// reference's orientation of the assignment problem, see the definition
void jk_permutation_adjustment(cals::Ktensor &ktensor, std::vector<cals::Ktensor> &jk_ktensor_v);

// || X without mode-0 slice i || for every i (src/utils/utils.cpp:103-152); host loops -- the engine
// computes the same on the device at cals_hip_set_tensor
vector<double> calculate_jackknifing_norms(cals::Tensor const &tensor);
}  // namespace cals::utils

namespace cals::ops {
// A^T A of a factor (src/utils/utils.cpp:174-178): host loops, used by MultiKtensor::add for the registry's
// Gramians; the engine's own Gramians are formed on the matrix cores
void update_gramian(const cals::Matrix &factor, cals::Matrix &gramian);
void update_gramians(const cals::Ktensor &ktensor, vector<Matrix> &gramians);
Matrix &hadamard_but_one(std::vector<cals::Matrix> &matrices, dim_t mode);
void hadamard_all(std::vector<cals::Matrix> &matrices);
}  // namespace cals::ops
#endif
