// cp_als / cp_omp_als / jk_cp_als / jk_cp_omp_als (reference: src/als.cpp) on the device engine: a single
// model is the concurrent engine with one model in flight -- the reference's tests demand CALS == ALS per
// model anyway (tests/cals/test_cals.cpp:60-86) -- and "OpenMP over models" is all models at once.
#include <stdexcept>

#include "cals.h"

namespace cals {

namespace {
CalsParams to_cals_params(const AlsParams &ap, dim_t buffer_size) {
  CalsParams p;
  p.update_method = ap.update_method;
  p.mttkrp_method = ap.mttkrp_method;
  p.max_iterations = ap.max_iterations;
  p.tol = ap.tol;
  p.cuda = ap.cuda;
  p.buffer_size = buffer_size;
  p.line_search = ap.line_search;
  p.line_search_interval = ap.line_search_interval;
  p.line_search_step = ap.line_search_step;
  p.line_search_method = ap.line_search_method;
  p.force_max_iter = ap.force_max_iter;
  p.device = ap.device;
  p.with_time = ap.with_time;
  return p;
}

AlsReport to_als_report(const CalsReport &r, const Ktensor &kt) {
  AlsReport out;
  out.tensor_rank = r.tensor_rank;
  out.n_modes = r.n_modes;
  out.modes = r.modes;
  out.X_norm = r.X_norm;
  out.iter = kt.get_iters();
  out.max_iter = r.max_iter;
  out.n_threads = r.n_threads;
  out.ktensor_id = kt.get_id();
  out.ktensor_components = kt.get_components();
  out.tol = r.tol;
  out.cuda = true;
  out.update_method = r.update_method;
  out.line_search = r.line_search;
  out.line_search_interval = r.line_search_interval;
  out.line_search_step = r.line_search_step;
  out.ls_performed = r.ls_performed;
  out.ls_failed = r.ls_failed;
  out.line_search_method = r.line_search_method;
  out.flops_per_iteration = r.flops_per_iteration.empty() ? 0 : r.flops_per_iteration[0];
  out.total_time = r.total_time;
  out.als_times = r.als_times;
  out.mode_times = r.mode_times;
  out.mttkrp_times = r.mttkrp_times;
  return out;
}
}  // namespace

void AlsParams::print() const {
  using std::cout;
  using std::endl;
  cout << "---------------------------------------" << endl;
  cout << "ALS parameters" << endl;
  cout << "---------------------------------------" << endl;
  cout << "Tolerance:        " << tol << endl;
  cout << "Max Iterations:   " << max_iterations << endl;
  cout << "Mttkrp Method:    " << mttkrp::mttkrp_method_names[mttkrp_method] << " (the device engine picks its own plan)" << endl;
  cout << "Update Method:    " << update::update_method_names[update_method] << endl;
  cout << "Line Search:      " << (line_search ? "true" : "false") << endl;
  if (line_search) {
    cout << "-Line Search Interval: " << line_search_interval << " iterations" << endl;
    cout << "-Line Search Method:   " << ls::ls_method_names[line_search_method] << endl;
  }
  cout << "CUDA:             " << (cuda ? "true" : "false") << " (device path: MI355X HIP engine, device " << device << ")" << endl;
  cout << "---------------------------------------" << endl;
}

void AlsReport::print_header(const std::string &file_name, const std::string &sep) const {
  std::ofstream file(file_name, std::ios::out);
  AlsTimers als_timers;
  ModeTimers mode_timers;
  for (const char *col : {"TENSOR_RANK", "TENSOR_MODES", "KTENSOR_ID", "KTENSOR_COMP", "UPDATE_METHOD", "LINE_SEARCH",
                          "MAX_ITERS", "ITER", "NUM_THREADS", "TOTAL"})
    file << col << sep;
  if (als_times.get_n_elements() > 0) {
    file << "FLOPS" << sep;
    for (const auto &name : als_timers.names) file << name << sep;
    for (dim_t m = 0; m < modes.size(); m++)
      for (const auto &name : mode_timers.names) file << "MODE_" << m << "_" << name << sep;
  }
  file << std::endl;
}

void AlsReport::print_to_file(const std::string &file_name, const std::string &sep) const {
  std::ofstream file(file_name, std::ios::app);
  file << tensor_rank << sep << utils::mode_string(modes) << sep << ktensor_id << sep << ktensor_components << sep
       << update::update_method_names[update_method] << sep << line_search << sep << max_iter << sep << iter << sep
       << n_threads << sep << total_time << sep;
  if (als_times.get_n_elements() > 0) {
    file << flops_per_iteration << sep << std::scientific;
    auto row_min = [&](const Matrix &M, dim_t r) {
      double best = std::numeric_limits<double>::max();
      for (dim_t c = 0; c < M.get_cols(); c++) best = std::min(best, M(r, c));
      return best;
    };
    for (dim_t r = 0; r < als_times.get_rows(); r++) file << row_min(als_times, r) << sep;
    for (dim_t r = 0; r < mode_times.get_rows(); r++) file << row_min(mode_times, r) << sep;
  }
  file << std::endl;
}

// cp_als (include/als.h:190, src/als.cpp:19-289): one model in flight, buffer = its rank.  With
// CalsParams::reuse_device_tensor (default) a loop of cp_als calls on one Tensor shares the engine that
// holds X's copies in HBM (cals_hip_rebind), as the reference's `cuda_no_tensor_alloc` path shares cudata.
AlsReport cp_als(const Tensor &X, Ktensor &ktensor, AlsParams &ap) {
  CalsParams p = to_cals_params(ap, ktensor.get_components());
  KtensorQueue q;
  q.emplace(ktensor);
  const CalsReport r = cp_cals(X, q, p);
  return to_als_report(r, ktensor);
}

// cp_omp_als (include/als.h:218, src/als.cpp:340-360): every model fitted independently by ALS.  The
// reference spreads the models over OpenMP threads; on the device "all of them at once" IS the
// concurrent engine, whose per-model results equal cp_als (tests/cals/test_cals.cpp:60-86).
vector<AlsReport> cp_omp_als(const Tensor &X, vector<Ktensor> &ktensor_v, AlsParams &params) {
  dim_t cols = 0;
  for (auto &k : ktensor_v) cols += k.get_components();
  CalsParams p = to_cals_params(params, std::max<dim_t>(cols, 1));
  KtensorQueue q;
  for (auto &k : ktensor_v) q.emplace(k);
  const CalsReport r = cp_cals(X, q, p);
  vector<AlsReport> reports;
  reports.reserve(ktensor_v.size());
  for (auto &k : ktensor_v) reports.push_back(to_als_report(r, k));
  return reports;
}

// jk_cp_als / jk_cp_omp_als (include/als.h:203,220, src/als.cpp:362-500): the jackknife comparator --
// every replica is a plain model of the SUB-SAMPLED tensor (mode-0 slice i removed).  Replicas of
// all input models that share a removed slice share one engine run on that sub-tensor.
JKReport jk_cp_omp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &als_params) {
  const auto modes = X.get_modes();
  if (modes.size() != 3) throw std::runtime_error("jk_cp_als: 3-way tensors only (src/als.cpp:364-365)");
  const dim_t I0 = modes[0], rest = modes[1] * modes[2];
  vector<Ktensor> ktensors(kt_vector);
  for (auto &k : ktensors) {
    k.denormalize();
    k.normalize();
  }
  auto jk_modes(modes);
  jk_modes[0] -= 1;
  vector<vector<Ktensor>> jk_input(ktensors.size());
  for (auto &k : jk_input) k.resize(I0);
  double pre_time = 0.0, als_time = 0.0;
  for (dim_t i_jk = 0; i_jk < I0; i_jk++) {
    Timer pre, run;
    pre.start();
    Tensor X_jk(jk_modes);
    for (dim_t jj = 0; jj < rest; jj++)
      for (dim_t ii = 0; ii < I0; ii++) {
        if (ii == i_jk) continue;
        X_jk[(ii < i_jk ? ii : ii - 1) + (I0 - 1) * jj] = X[ii + I0 * jj];
      }
    dim_t cols = 0;
    for (size_t i_kt = 0; i_kt < ktensors.size(); i_kt++) {
      const Ktensor &src = ktensors[i_kt];
      Ktensor kt_jk(src.get_components(), jk_modes);
      kt_jk.get_lambda() = src.get_lambda();
      for (dim_t f = 0; f < 3; f++) {
        const Matrix &fs = src.get_factor(f);
        Matrix &fd = kt_jk.get_factor(f);
        for (dim_t jj = 0; jj < fs.get_cols(); jj++)
          for (dim_t ii = 0; ii < fs.get_rows(); ii++) {
            if (f == 0 && ii == i_jk) continue;
            fd((f == 0 && ii > i_jk) ? ii - 1 : ii, jj) = fs(ii, jj);
          }
      }
      cols += src.get_components();
      jk_input[i_kt][i_jk] = std::move(kt_jk);
    }
    pre.stop();
    run.start();
    CalsParams p = to_cals_params(als_params, std::max<dim_t>(cols, 1));
    p.reuse_device_tensor = false;  // X_jk lives for this iteration only
    KtensorQueue q;
    for (size_t i_kt = 0; i_kt < ktensors.size(); i_kt++) q.emplace(jk_input[i_kt][i_jk]);
    cp_cals(X_jk, q, p);
    run.stop();
    pre_time += pre.get_time();
    als_time += run.get_time();
  }
  for (auto &k : jk_input)
    for (auto &m : k) {
      m.denormalize();
      m.normalize();
    }
  for (size_t i = 0; i < ktensors.size(); i++) utils::jk_permutation_adjustment(ktensors[i], jk_input[i]);
  JKReport rep;
  rep.jk_time.pre_als_time = pre_time;
  rep.jk_time.als_time = als_time;
  rep.results = std::move(jk_input);
  return rep;
}

JKReport jk_cp_als(const Tensor &X, vector<Ktensor> &kt_vector, AlsParams &als_params) {
  return jk_cp_omp_als(X, kt_vector, als_params);
}

}  // namespace cals
