// include/utils/utils.h of HPAC/CP-CALS: the post-processing helpers of the jackknife front-ends
// (namespace cals::utils) and the small Gramian / Hadamard operations of the model registry (cals::ops).
#ifndef CALS_AMD_UTILS_UTILS_H
#define CALS_AMD_UTILS_UTILS_H

#include <string>

#include "ktensor.h"

namespace cals {

namespace ops {
// Host loops on r x r / I x r operands, used by MultiKtensor::add for the registry's Gramians; the engine's
// own Gramians are formed on the matrix cores inside update_kernel.
Matrix &hadamard_but_one(std::vector<Matrix> &matrices, dim_t mode);  // src/utils/utils.cpp:161-172
void hadamard_all(std::vector<Matrix> &matrices);                     // :156-159, result in matrices[0]
void update_gramians(const Ktensor &ktensor, vector<Matrix> &gramians);
void update_gramian(const Matrix &factor, Matrix &gramian);           // A^T A, :174-178
}  // namespace ops

namespace utils {
// || X without mode-0 slice i || for every i (src/utils/utils.cpp:103-152); host loops -- the engine computes
// the same on the device at cals_hip_set_tensor
vector<double> calculate_jackknifing_norms(Tensor const &tensor);

// reorder every replica's columns after the overall model's (src/utils/utils.cpp:54-101) -- with the
// reference's orientation of the assignment problem, see the definition
void jk_permutation_adjustment(Ktensor &ktensor, std::vector<Ktensor> &jk_ktensor_v);

// modes[0] copies of the model, copy i flagged jk(mode 0, fiber i) (src/utils/utils.cpp:40-52)
void generate_jk_ktensors(Ktensor const &reference_ktensor, std::vector<Ktensor> &jk_ktensor_v);

// one wide Ktensor holding the columns of all inputs side by side (src/utils/utils.cpp:18-38)
Ktensor concatenate_ktensors(std::vector<Ktensor> const &ktensors);

std::string mode_string(std::vector<dim_t> const &modes);  // "I-J-K"
}  // namespace utils

}  // namespace cals
#endif
