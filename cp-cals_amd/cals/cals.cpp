// cals::cp_cals over the C ABI (include/cals_hip.h).  Reference boundary: include/cals.h:196,
// body src/cals.cpp:19-395.
#include "cals.h"

#include <atomic>
#include <chrono>
#include <exception>
#include <limits>
#include <thread>

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
  cout << "Device path:     MI355X HIP engine (device";
  if (devices.empty())
    cout << " " << device;
  else
    for (int d : devices) cout << " " << d;
  cout << ", " << (precision == FP32 ? "fp32" : "fp64") << " storage)" << endl;
  cout << "---------------------------------------" << endl;
}

double Timer::now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Tensor::Tensor(const std::string &file_name) {
  std::ifstream file(file_name);
  if (!file.is_open()) throw std::runtime_error("Tensor: cannot open " + file_name);
  std::string line;
  std::getline(file, line);
  std::stringstream ss(line);
  dim_t m;
  while (ss >> m) modes.push_back(m);
  n_elements = 1;
  for (auto v : modes) n_elements *= v;
  owned.reset(new double[n_elements]);
  data = owned.get();
  dim_t index = 0;
  double val;
  while (index < n_elements && file >> val) data[index++] = val;
  if (index != n_elements) throw std::runtime_error("Tensor: " + file_name + " holds too few values");
}

void AlsParams::print() const {
  using std::cout;
  using std::endl;
  cout << "---------------------------------------" << endl;
  cout << "ALS parameters" << endl;
  cout << "---------------------------------------" << endl;
  cout << "Tolerance:        " << tol << endl;
  cout << "Max Iterations:   " << max_iterations << endl;
  cout << "Line Search:      " << (line_search ? "true" : "false") << endl;
  cout << "Device path:      MI355X HIP engine (device " << device << ")" << endl;
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

namespace {
cals_hip_params to_hip_params(const CalsParams &p) {
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
  hp.update_method = (p.update_method == update::NNLS) ? 1 : 0;
  return hp;
}

// CalsParams::devices with more than one entry: one engine (and one host thread) per device, each
// with its own replica of X; the models sit behind one shared counter and a device claims a few
// more whenever none of its claimed models is waiting for buffer columns (the pull-based hand-off
// of cp-cals_amd/multi_gpu.py, here inside one process: no collective, nothing but indices shared).
// Every model is fitted by exactly one device with the arithmetic of the single-device path.
void cp_cals_devices(const Tensor &X, std::vector<std::reference_wrapper<Ktensor>> &all, const CalsParams &p,
                     CalsReport &rep) {
  const size_t n_dev = p.devices.size();
  std::vector<int64_t> modes(X.get_modes().begin(), X.get_modes().end());
  std::atomic<size_t> next{0};
  std::vector<cals_hip_report> reports(n_dev);
  std::vector<std::exception_ptr> errors(n_dev);
  auto worker = [&](size_t d) {
    try {
      EngineGuard g;
      int rc = cals_hip_create_ex(&g.e, (int)modes.size(), modes.data(), (int64_t)p.buffer_size, p.devices[d],
                                  p.precision == CalsParams::FP32 ? CALS_HIP_F32 : CALS_HIP_F64);
      if (rc) fail(g.e, "cals_hip_create", rc);
      if ((rc = cals_hip_set_tensor(g.e, X.get_data()))) fail(g.e, "cals_hip_set_tensor", rc);
      cals_hip_params hp = to_hip_params(p);
      if ((rc = cals_hip_set_params(g.e, &hp))) fail(g.e, "cals_hip_set_params", rc);
      std::vector<std::pair<size_t, int64_t>> mine;  // (index into all, ticket)
      bool drained = false;
      const size_t claim = (size_t)std::max(1, p.claim_models);
      for (;;) {
        if (!drained && cals_hip_queue_size(g.e) == 0) {
          const size_t lo = next.fetch_add(claim);
          if (lo >= all.size()) drained = true;
          for (size_t i = lo; i < std::min(lo + claim, all.size()); i++) {
            Ktensor &kt = all[i];
            std::vector<double *> fptr;
            for (auto &f : kt.get_factors()) fptr.push_back(f.get_data());
            int64_t ticket = -1;
            rc = cals_hip_enqueue(g.e, (int64_t)kt.get_components(), fptr.data(), kt.get_lambda().data(),
                                  kt.is_jk() ? (int)kt.get_jk_mode() : -1, (int64_t)kt.get_jk_fiber(), &ticket);
            if (rc) fail(g.e, "cals_hip_enqueue", rc);
            mine.emplace_back(i, ticket);
          }
        }
        if (cals_hip_queue_size(g.e) == 0 && cals_hip_models_in_flight(g.e) == 0) {
          if (drained) break;
          continue;
        }
        if ((rc = cals_hip_step(g.e, nullptr, nullptr))) fail(g.e, "cals_hip_step", rc);
      }
      for (auto &m : mine) {
        cals_hip_model_status st;
        if ((rc = cals_hip_model_result(g.e, m.second, &st))) fail(g.e, "cals_hip_model_result", rc);
        Ktensor &kt = all[m.first];
        kt.set_iters((dim_t)st.iters);
        kt.set_approximation_error(st.approx_error);
        kt.set_fit(st.fit, st.old_fit);
      }
      if ((rc = cals_hip_get_report(g.e, &reports[d]))) fail(g.e, "cals_hip_get_report", rc);
    } catch (...) {
      errors[d] = std::current_exception();
      next.store(all.size());  // the other devices finish what they hold and stop claiming
    }
  };
  std::vector<std::thread> threads;
  for (size_t d = 0; d < n_dev; d++) threads.emplace_back(worker, d);
  for (auto &t : threads) t.join();
  for (auto &e : errors)
    if (e) std::rethrow_exception(e);
  rep.iter = 0;
  rep.n_ktensors = 0;
  rep.ktensor_comp_sum = 0;
  rep.ls_performed = rep.ls_failed = 0;
  for (auto &r : reports) {
    rep.X_norm = r.X_norm;
    rep.iter = std::max(rep.iter, (dim_t)r.iter);  // sweeps of the device that swept most
    rep.n_ktensors += (int)r.n_ktensors;
    rep.ktensor_comp_sum += (int)r.ktensor_comp_sum;
    rep.ls_performed += (dim_t)r.ls_performed;
    rep.ls_failed += (dim_t)r.ls_failed;
  }
}
}  // namespace

CalsReport cp_cals(const Tensor &X, KtensorQueue &kt_queue, CalsParams &p) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!p.cuda)
    throw std::runtime_error("cp_cals: this library has only the MI355X device path (no CPU "
                             "fallback); CalsParams::cuda must stay true");
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

  if (p.devices.size() > 1) {
    std::vector<std::reference_wrapper<Ktensor>> all;
    while (!kt_queue.empty()) {
      all.push_back(kt_queue.front());
      kt_queue.pop();
    }
    cp_cals_devices(X, all, p, rep);
    rep.total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rep;
  }
  std::vector<int64_t> modes(X.get_modes().begin(), X.get_modes().end());
  EngineGuard g;
  int rc = cals_hip_create_ex(&g.e, (int)modes.size(), modes.data(), (int64_t)p.buffer_size,
                              p.devices.size() == 1 ? p.devices[0] : p.device,
                              p.precision == CalsParams::FP32 ? CALS_HIP_F32 : CALS_HIP_F64);
  if (rc) fail(g.e, "cals_hip_create", rc);
  if ((rc = cals_hip_set_tensor(g.e, X.get_data()))) fail(g.e, "cals_hip_set_tensor", rc);
  cals_hip_params hp = to_hip_params(p);
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


// cp_als (include/als.h:190, src/als.cpp:19-289) on the device engine: one model in flight.
AlsReport cp_als(const Tensor &X, Ktensor &ktensor, AlsParams &ap) {
  CalsParams p;
  p.update_method = ap.update_method;
  p.max_iterations = ap.max_iterations;
  p.tol = ap.tol;
  p.cuda = ap.cuda;
  p.buffer_size = ktensor.get_components();
  p.line_search = ap.line_search;
  p.line_search_interval = ap.line_search_interval;
  p.line_search_step = ap.line_search_step;
  p.line_search_method = ap.line_search_method;
  p.force_max_iter = ap.force_max_iter;
  p.device = ap.device;
  KtensorQueue q;
  q.emplace(ktensor);
  CalsReport r = cp_cals(X, q, p);
  AlsReport out;
  out.iter = r.iter;
  out.ls_performed = r.ls_performed;
  out.ls_failed = r.ls_failed;
  out.X_norm = r.X_norm;
  out.total_time = r.total_time;
  return out;
}

// ---------------------------------------------------------------------------------------------
// linear sum assignment: Hungarian algorithm with row/column potentials, O(n^3)
// ---------------------------------------------------------------------------------------------
int solve_linear_sum_assignment(int n, const double *cost, bool maximize, int64_t *col_of_row) {
  if (n < 1 || !cost || !col_of_row) return -2;
  const double INF = std::numeric_limits<double>::infinity();
  auto c = [&](int i, int j) {  // 1-based row i, column j; minimisation form
    const double v = cost[(i - 1) + (size_t)n * (j - 1)];
    return maximize ? -v : v;
  };
  for (int k = 0; k < n * n; k++)
    if (!(cost[k] == cost[k]) || cost[k] == INF || cost[k] == -INF) return -2;
  std::vector<double> u(n + 1, 0.0), v(n + 1, 0.0), minv(n + 1);
  std::vector<int> p(n + 1, 0), way(n + 1, 0);
  std::vector<char> used(n + 1);
  for (int i = 1; i <= n; i++) {
    p[0] = i;
    int j0 = 0;
    std::fill(minv.begin(), minv.end(), INF);
    std::fill(used.begin(), used.end(), 0);
    do {
      used[j0] = 1;
      const int i0 = p[j0];
      double delta = INF;
      int j1 = 0;
      for (int j = 1; j <= n; j++) {
        if (used[j]) continue;
        const double cur = c(i0, j) - u[i0] - v[j];
        if (cur < minv[j]) {
          minv[j] = cur;
          way[j] = j0;
        }
        if (minv[j] < delta) {
          delta = minv[j];
          j1 = j;
        }
      }
      for (int j = 0; j <= n; j++) {
        if (used[j]) {
          u[p[j]] += delta;
          v[j] -= delta;
        } else
          minv[j] -= delta;
      }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  for (int j = 1; j <= n; j++) col_of_row[p[j] - 1] = j - 1;
  return 0;
}

namespace utils {

std::string mode_string(vector<dim_t> const &modes) {  // src/utils/utils.cpp:9-16
  std::string m;
  for (auto const &v : modes) m += std::to_string(v) + '-';
  if (!m.empty()) m.pop_back();
  return m;
}

Ktensor concatenate_ktensors(vector<Ktensor> const &ktensors) {  // src/utils/utils.cpp:18-38
  const dim_t comp = ktensors[0].get_components();
  Ktensor out(ktensors.size() * comp, ktensors[0].get_modes());
  dim_t index = 0;
  for (auto const &kt : ktensors) {
    for (dim_t i = 0; i < comp; i++) out.get_lambda()[index * comp + i] = kt.get_lambda()[i];
    for (dim_t m = 0; m < kt.get_n_modes(); m++)
      for (dim_t c = 0; c < comp; c++)
        for (dim_t r = 0; r < kt.get_factor(m).get_rows(); r++)
          out.get_factor(m)(r, index * comp + c) = kt.get_factor(m)(r, c);
    index++;
  }
  return out;
}

void generate_jk_ktensors(Ktensor const &reference_ktensor, vector<Ktensor> &jk_ktensor_v) {
  // src/utils/utils.cpp:40-52: one copy per mode-0 slice, flagged jk(mode 0, fiber i)
  const dim_t I0 = reference_ktensor.get_modes()[0];
  if (I0 <= 1) throw std::string("Can't do Jack-knife with just one sample.");
  for (dim_t i = 0; i < I0; i++) {
    Ktensor copy(reference_ktensor);
    copy.to_jk(0, i);
    jk_ktensor_v.push_back(std::move(copy));
  }
}

void jk_permutation_adjustment(Ktensor &ktensor, vector<Ktensor> &jk_ktensor_v) {
  // src/utils/utils.cpp:54-101: match the columns of every replica to the overall model by
  // maximising trace(P^T (Bov^T Bm + Cov^T Cm)), then reorder the replica's columns
  const auto &modes = ktensor.get_modes();
  const dim_t comp = ktensor.get_components();
  const Matrix &Bov = ktensor.get_factor(1), &Cov = ktensor.get_factor(2);
  for (dim_t m = 0; m < modes[0]; m++) {
    Ktensor &kt = jk_ktensor_v[m];
    const Matrix &Bm = kt.get_factor(1), &Cm = kt.get_factor(2);
    std::vector<double> M(comp * comp, 0.0);
    for (dim_t j = 0; j < comp; j++)
      for (dim_t i = 0; i < comp; i++) {
        double s = 0.0, t = 0.0;
        for (dim_t r = 0; r < modes[1]; r++) s += Bov(r, i) * Bm(r, j);
        for (dim_t r = 0; r < modes[2]; r++) t += Cov(r, i) * Cm(r, j);
        M[i + comp * j] = s + t;
      }
    std::vector<int64_t> solved(comp);
    solve_linear_sum_assignment((int)comp, M.data(), true, solved.data());
    for (dim_t mode = 0; mode < ktensor.get_n_modes(); mode++) {
      Matrix &f = kt.get_factor(mode);
      Matrix copy(f.get_rows(), f.get_cols());
      copy.copy(f);
      for (dim_t cur = 0; cur < comp; cur++) {
        const dim_t swap = (dim_t)solved[cur];
        if (swap != cur)
          for (dim_t r = 0; r < f.get_rows(); r++) f(r, cur) = copy(r, swap);
      }
    }
  }
}

}  // namespace utils

JKReport jk_cp_cals(const Tensor &X, vector<Ktensor> &kt_vector, CalsParams &cals_params) {
  vector<Ktensor> ktensors(kt_vector);
  for (auto &k : ktensors) {
    k.denormalize();
    k.normalize();
  }
  Timer pre, run;
  pre.start();
  vector<vector<Ktensor>> jk_input(ktensors.size());
  for (size_t i = 0; i < ktensors.size(); i++) utils::generate_jk_ktensors(ktensors[i], jk_input[i]);
  KtensorQueue queue;
  for (auto &k : jk_input)
    for (auto &m : k) queue.emplace(m);
  pre.stop();
  run.start();
  cp_cals(X, queue, cals_params);
  run.stop();
  for (auto &k : jk_input)
    for (auto &m : k) {
      m.set_jk_fiber(0.0);
      m.denormalize();
      m.normalize();
      m.set_jk_fiber(NAN);
    }
  for (size_t i = 0; i < ktensors.size(); i++) utils::jk_permutation_adjustment(ktensors[i], jk_input[i]);
  JKReport rep;
  rep.jk_time.pre_als_time = pre.get_time();
  rep.jk_time.als_time = run.get_time();
  rep.results = std::move(jk_input);
  return rep;
}

namespace {
CalsParams to_cals_params(const AlsParams &ap, dim_t buffer_size) {
  CalsParams p;
  p.update_method = ap.update_method;
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
  return p;
}
}  // namespace

// cp_omp_als (include/als.h:218, src/als.cpp:340-360): every model fitted independently by ALS.  The
// reference spreads the models over OpenMP threads; on the device "all of them at once" IS the
// concurrent engine, whose per-model results equal cp_als (tests/cals/test_cals.cpp:60-86).
vector<AlsReport> cp_omp_als(const Tensor &X, vector<Ktensor> &ktensor_v, AlsParams &params) {
  Timer total;
  total.start();
  dim_t cols = 0;
  for (auto &k : ktensor_v) cols += k.get_components();
  CalsParams p = to_cals_params(params, std::max<dim_t>(cols, 1));
  KtensorQueue q;
  for (auto &k : ktensor_v) q.emplace(k);
  CalsReport r = cp_cals(X, q, p);
  total.stop();
  vector<AlsReport> reports(ktensor_v.size());
  for (size_t i = 0; i < ktensor_v.size(); i++) {
    reports[i].iter = ktensor_v[i].get_iters();
    reports[i].X_norm = r.X_norm;
    reports[i].total_time = total.get_time();
  }
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

extern "C" int solve_rectangular_linear_sum_assignment(intptr_t nr, intptr_t nc, double *input_cost,
                                                        bool maximize, int64_t *a, int64_t *b) {
  if (nr != nc || nr < 1 || !input_cost || !a || !b) return -1;
  const int n = (int)nr;
  std::vector<double> cm((size_t)n * n);  // row-major in, column-major for the solver
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) cm[(size_t)i + (size_t)n * j] = input_cost[(size_t)i * n + j];
  const int rc = cals::solve_linear_sum_assignment(n, cm.data(), maximize, b);
  for (int i = 0; i < n; i++) a[i] = i;
  return rc;
}

// C entry point of the assignment solver (tests bind it with ctypes)
extern "C" int cals_lsap_solve(int n, const double *cost_colmajor, int maximize, int64_t *col_of_row) {
  return cals::solve_linear_sum_assignment(n, cost_colmajor, maximize != 0, col_of_row);
}
