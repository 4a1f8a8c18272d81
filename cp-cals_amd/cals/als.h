// include/als.h of HPAC/CP-CALS: the single-model comparator of every driver and test.  Here cp_als IS
// the device engine with one model in flight (the reference's tests demand CALS == ALS per model,
// tests/cals/test_cals.cpp:60-86), cp_omp_als all models concurrently.
#ifndef CALS_AMD_ALS_H
#define CALS_AMD_ALS_H

#include <fstream>
#include <iostream>

#include "ktensor.h"
#include "timer.h"
#include "utils/line_search.h"
#include "utils/mttkrp.h"
#include "utils/update.h"
#include "utils/utils.h"

namespace cals {

struct JKTime {
  double pre_als_time{0.0};
  double als_time{0.0};
};
struct JKReport {
  JKTime jk_time{};
  vector<vector<Ktensor>> results;
};

struct AlsReport {  // include/als.h:30-139
  int tensor_rank{0};
  dim_t n_modes{0};
  vector<dim_t> modes{};
  double X_norm{0.0};
  dim_t iter = 0;
  dim_t max_iter{0};
  int n_threads{1};
  int ktensor_id{0};
  dim_t ktensor_components{0};
  double tol{0.0};
  bool cuda{true};
  update::UPDATE_METHOD update_method{update::UNCONSTRAINED};
  bool line_search{false};
  int line_search_interval{0};
  double line_search_step{0.0};
  dim_t ls_performed{0};
  dim_t ls_failed{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  uint64_t flops_per_iteration{0};
  double total_time{0.0};
  Matrix als_times{};   // AlsTimers::LENGTH x iter, seconds (filled when AlsParams::with_time)
  Matrix mode_times{};  // n_modes * ModeTimers::LENGTH x iter
  Matrix mttkrp_times{};

  void print_header(const std::string &file_name, const std::string &sep = ";") const;
  // one row: the run's parameters, then per timer the MINIMUM over the iterations (include/als.h:101-139)
  void print_to_file(const std::string &file_name, const std::string &sep = ";") const;
};

struct AlsParams {  // include/als.h:142-188
  update::UPDATE_METHOD update_method{update::UPDATE_METHOD::UNCONSTRAINED};
  mttkrp::MTTKRP_METHOD mttkrp_method{mttkrp::MTTKRP_METHOD::AUTO};
  cals::mttkrp::MttkrpLut mttkrp_lut{};
  dim_t max_iterations{200};
  double tol{1e-7};
  bool cuda{true};  // the reference's switch to its device path; this library HAS only the device path
  bool line_search{false};
  int line_search_interval{5};
  double line_search_step{0};
  ls::LS_METHOD line_search_method{ls::NO_ERROR_CHECKING};
  bool cuda_no_tensor_alloc{false};
  bool force_max_iter{false};
  bool suppress_lut_warning{false};
  int device{0};                                // added: HIP device ordinal
  bool with_time{CALS_AMD_WITH_TIME_DEFAULT};   // added: fill the report's timer matrices

  void print() const;
};

AlsReport cp_als(const Tensor &X, Ktensor &ktensor, AlsParams &als_params);
JKReport jk_cp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &als_params);
vector<AlsReport> cp_omp_als(const Tensor &X, vector<Ktensor> &ktensor, AlsParams &params);
JKReport jk_cp_omp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &params);

}  // namespace cals
#endif
