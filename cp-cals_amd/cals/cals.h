// include/cals.h of HPAC/CP-CALS: the boundary of the hot path,
//     cals::CalsReport cals::cp_cals(const Tensor &X, KtensorQueue &kt_queue, CalsParams &cals_params)
// with the same names, argument meaning, ownership rules and report fields -- over the C ABI of the
// MI355X engine (include/cals_hip.h, libcals_hip.so).  A caller written against the reference's headers
// (`#include "cals.h"`, "als.h", "timer.h", "ktensor.h", ...) compiles against this directory unchanged:
//     g++ -std=c++17 -I cp-cals_amd/cals -I cp-cals_amd/cals/utils caller.cpp -L cp-cals_amd -lcals -lcals_hip
// (tests/test_cpp_headers_and_sanitizers.py does exactly that with the reference's own src/examples/driver.cpp).
#ifndef CALS_AMD_CALS_H
#define CALS_AMD_CALS_H

#include <cfloat>
#include <fstream>
#include <functional>
#include <iostream>
#include <numeric>
#include <queue>

#include "als.h"
#include "ktensor.h"
#include "multi_ktensor.h"
#include "tensor.h"
#include "timer.h"
#include "utils/error.h"
#include "utils/line_search.h"
#include "utils/mttkrp.h"
#include "utils/update.h"
#include "utils/utils.h"
#include "rectangular_lsap/rectangular_lsap.h"

namespace cals {
typedef std::queue<std::reference_wrapper<Ktensor>> KtensorQueue;  // include/cals.h:22

// include/cals.h:27-132.  The reference compiles the per-iteration matrices in only with -DWITH_TIME=1;
// here they always exist and a run fills them when CalsParams::with_time is set (default: WITH_TIME).
struct CalsReport {
  int tensor_rank{0};
  dim_t n_modes{0};
  vector<dim_t> modes{};
  double X_norm{0.0};

  dim_t iter{0};
  dim_t max_iter{0};
  int n_threads{1};
  dim_t buffer_size{0};
  int n_ktensors{0};
  int ktensor_comp_sum{0};
  double tol{0.0};
  bool cuda{true};
  update::UPDATE_METHOD update_method{update::UNCONSTRAINED};
  std::string output_file_name{};

  bool line_search{false};
  int line_search_interval{0};
  double line_search_step{0.0};
  dim_t ls_performed{0};
  dim_t ls_failed{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};

  double total_time{0.0};
  // Column it = outer iteration it + 1, seconds.  Rows: als_times AlsTimers::{ITERATION, DEFRAGMENTATION,
  // ERROR, LINE_SEARCH, G_COPY}; mode_times mode n * ModeTimers::LENGTH + {MTTKRP, UPDATE}; mttkrp_times
  // mode n * MttkrpTimers::LENGTH + {MT_KRP, MT_GEMM, TS_GEMM, TS_GEMV}.  Source: cals_hip_sweep_record
  // (host clock of the loop iteration; hipEvent pairs around the kernels).  ERROR and G_COPY are 0 by
  // construction: the fast error is fused into the last mode's update kernel and no G copy exists.
  Matrix als_times{};
  Matrix mode_times{};
  Matrix mttkrp_times{};
  vector<uint64_t> flops_per_iteration{};  // MFMA-kernel flops of the iteration: 2 * prod(modes) * cols per launch
  vector<dim_t> cols{};                    // active columns of the multi-factors per iteration
  int mttkrp_plan{0};                      // added: cals_hip_tree of the engine that ran (0 / A / B / M; 4 = N > 3 group tree)

  void print_header(const std::string &file_name, const std::string &sep = ";") const;
  void print_to_file(const std::string &file_name, const std::string &sep = ";") const;
};

// include/cals.h:138-181.  `cuda` selects the device path in the reference; this library HAS only the
// device path (MI355X), so it defaults to true and cp_cals throws if it is false.
struct CalsParams {
  update::UPDATE_METHOD update_method{update::UPDATE_METHOD::UNCONSTRAINED};
  mttkrp::MTTKRP_METHOD mttkrp_method{mttkrp::MTTKRP_METHOD::AUTO};  // accepted, no effect (utils/mttkrp.h)
  cals::mttkrp::MttkrpLut mttkrp_lut{};                              // accepted, no effect
  dim_t max_iterations{200};
  double tol{1e-7};
  bool cuda{true};
  dim_t buffer_size{4200};
  bool line_search{false};
  int line_search_interval{5};
  double line_search_step{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  bool force_max_iter{false};
  bool always_evict_first{false};

  // ---- added (defaults preserve the reference's behaviour) ----
  int device{0};  // HIP device ordinal
  // more than one ordinal = one engine per listed device inside this process, the queue shared through an
  // atomic counter (each model is fitted by exactly one device); claim_models = how many models a device
  // takes at a time when none of its own is waiting for buffer columns
  std::vector<int> devices{};
  int claim_models{8};
  // storage/arithmetic type on the device: FP64 (the reference's) or FP32 (BASELINE config 4: fp32 tensor
  // copies, factors and MFMA; Gramians, solves, lambda, error stay fp64)
  enum PRECISION { FP64 = 0, FP32 = 1 };
  PRECISION precision{FP64};
  bool with_time{CALS_AMD_WITH_TIME_DEFAULT};  // fill CalsReport's per-iteration matrices
  // keep X's device copies with the Tensor between calls (include/tensor.h:56-59 does the same for the
  // reference's CUDA path); false = build and free a private engine inside every call
  bool reuse_device_tensor{true};

  void print() const;
};

// Fits every Ktensor of the queue to X with concurrent ALS on the GPU and overwrites it with the
// result (factors, lambda, error, fit, iters); the queue is empty on return (include/cals.h:183-196).
// Throws std::runtime_error on any engine error (the reference exit()s on device errors).
CalsReport cp_cals(const Tensor &X, KtensorQueue &kt_queue, CalsParams &cals_params);

// Jackknife driver (src/cals.cpp:397-446): for every model of kt_vector, modes[0] jackknife replicas
// fitted in ONE cp_cals call, then re-normalised and column-matched to the original.
JKReport jk_cp_cals(const Tensor &X, vector<Ktensor> &kt_vector, CalsParams &cals_params);

// n x n COLUMN-major cost: col_of_row[i] = column assigned to row i (convenience over
// solve_rectangular_linear_sum_assignment, which reads row-major)
int solve_linear_sum_assignment(int n, const double *cost_colmajor, bool maximize, int64_t *col_of_row);

}  // namespace cals
#endif
