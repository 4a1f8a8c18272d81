// cals::cp_cals over the C ABI (include/cals_hip.h).  Reference boundary: include/cals.h:196,
// body src/cals.cpp:19-395.
#include "cals.h"

#include <chrono>

#include "../../include/cals_hip.h"

namespace cals {

void CalsParams::print() const {
  using std::cout;
  using std::endl;
  cout << "---------------------------------------" << endl;
  cout << "CALS parameters" << endl;
  cout << "---------------------------------------" << endl;
  cout << "Tol:             " << tol << endl;
  cout << "Max Iterations:  " << max_iterations << endl;
  cout << "Buffer Size:     " << buffer_size << endl;
  cout << "Line Search:     " << (line_search ? "true" : "false") << endl;
  if (line_search) cout << "-Line Search Interval: " << line_search_interval << " iterations" << endl;
  cout << "Device path:     MI355X HIP engine (device " << device << ")" << endl;
  cout << "---------------------------------------" << endl;
}

namespace {
struct EngineGuard {
  cals_hip_engine *e{nullptr};
  ~EngineGuard() {
    if (e) cals_hip_destroy(e);
  }
};
[[noreturn]] void fail(cals_hip_engine *e, const char *what, int rc) {
  throw std::runtime_error(std::string("cp_cals: ") + what + " failed (" + std::to_string(rc) +
                           "): " + (e ? cals_hip_last_error(e) : "no engine"));
}
}  // namespace

CalsReport cp_cals(const Tensor &X, KtensorQueue &kt_queue, CalsParams &p) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!p.cuda)
    throw std::runtime_error("cp_cals: this library has only the MI355X device path (no CPU "
                             "fallback); CalsParams::cuda must stay true");
  if (p.update_method != update::UNCONSTRAINED)
    throw std::runtime_error("cp_cals: only update::UNCONSTRAINED runs on the device path");
  CalsReport rep;
  rep.tensor_rank = X.get_rank();
  rep.n_modes = X.get_n_modes();
  rep.modes = X.get_modes();
  rep.max_iter = p.max_iterations;
  rep.buffer_size = p.buffer_size;
  rep.tol = p.tol;
  rep.cuda = true;
  rep.update_method = p.update_method;
  rep.line_search = p.line_search;
  rep.line_search_interval = p.line_search_interval;
  rep.line_search_step = p.line_search_step;
  rep.line_search_method = p.line_search_method;

  std::vector<int64_t> modes(X.get_modes().begin(), X.get_modes().end());
  EngineGuard g;
  int rc = cals_hip_create(&g.e, (int)modes.size(), modes.data(), (int64_t)p.buffer_size, p.device);
  if (rc) fail(g.e, "cals_hip_create", rc);
  if ((rc = cals_hip_set_tensor(g.e, X.get_data()))) fail(g.e, "cals_hip_set_tensor", rc);
  cals_hip_params hp;
  cals_hip_default_params(&hp);
  hp.max_iterations = (int64_t)p.max_iterations;
  hp.tol = p.tol;
  hp.line_search = p.line_search ? 1 : 0;
  hp.line_search_interval = p.line_search_interval;
  hp.line_search_step = p.line_search_step;
  hp.line_search_method = (int)p.line_search_method;
  hp.force_max_iter = p.force_max_iter ? 1 : 0;
  hp.always_evict_first = p.always_evict_first ? 1 : 0;
  if ((rc = cals_hip_set_params(g.e, &hp))) fail(g.e, "cals_hip_set_params", rc);

  std::vector<std::reference_wrapper<Ktensor>> kts;
  std::vector<int64_t> tickets;
  while (!kt_queue.empty()) {
    Ktensor &kt = kt_queue.front();
    std::vector<double *> fptr;
    for (auto &f : kt.get_factors()) fptr.push_back(f.get_data());
    int64_t ticket = -1;
    rc = cals_hip_enqueue(g.e, (int64_t)kt.get_components(), fptr.data(), kt.get_lambda().data(),
                          kt.is_jk() ? (int)kt.get_jk_mode() : -1, (int64_t)kt.get_jk_fiber(), &ticket);
    if (rc) fail(g.e, "cals_hip_enqueue", rc);
    kts.push_back(kt);
    tickets.push_back(ticket);
    kt_queue.pop();
  }
  cals_hip_report hr;
  if ((rc = cals_hip_run(g.e, &hr))) fail(g.e, "cals_hip_run", rc);
  for (size_t i = 0; i < kts.size(); i++) {
    cals_hip_model_status st;
    if ((rc = cals_hip_model_result(g.e, tickets[i], &st))) fail(g.e, "cals_hip_model_result", rc);
    Ktensor &kt = kts[i];
    kt.set_iters((dim_t)st.iters);
    kt.set_approximation_error(st.approx_error);
    kt.set_fit(st.fit, st.old_fit);
  }
  rep.X_norm = hr.X_norm;
  rep.iter = (dim_t)hr.iter;
  rep.n_ktensors = (int)hr.n_ktensors;
  rep.ktensor_comp_sum = (int)hr.ktensor_comp_sum;
  rep.ls_performed = (dim_t)hr.ls_performed;
  rep.ls_failed = (dim_t)hr.ls_failed;
  rep.total_time =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rep;
}

}  // namespace cals
