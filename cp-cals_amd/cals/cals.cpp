// cals::cp_cals over the C ABI (include/cals_hip.h).  Reference boundary: include/cals.h:196,
// body src/cals.cpp:19-395.
#include "cals.h"

#include <algorithm>
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
// Linear sum assignment.  The reference hands this to SciPy's rectangular_lsap
// (extern/rectangular_lsap/rectangular_lsap.cpp: Crouse's shortest-augmenting-path variant of
// Jonker-Volgenant).  Own implementation of the same published algorithm (D. F. Crouse, "On
// implementing 2D rectangular assignment algorithms", IEEE T-AES 52(4), 2016), including its two
// tie rules -- the unvisited columns are scanned from the last to the first, and among equally short
// paths one that ends in an unassigned column wins -- so that degenerate costs (ties) give the
// assignment the reference gives, not merely one of equal value.  tests: against oracle/_ref's build
// of the reference's own file, scipy.optimize and exhaustive search.
// ---------------------------------------------------------------------------------------------
namespace {
struct LsapSolver {
  int64_t nr, nc;                 // nr <= nc
  std::vector<double> c;          // row-major nr x nc, non-negative, minimisation form
  std::vector<double> u, v, dist;
  std::vector<int64_t> pred, col_of_row, row_of_col, todo;
  std::vector<char> row_seen, col_seen;

  LsapSolver(int64_t nr_, int64_t nc_)
      : nr(nr_), nc(nc_), c((size_t)(nr_ * nc_)), u((size_t)nr_, 0.0), v((size_t)nc_, 0.0), dist((size_t)nc_),
        pred((size_t)nc_, -1), col_of_row((size_t)nr_, -1), row_of_col((size_t)nc_, -1), todo((size_t)nc_),
        row_seen((size_t)nr_), col_seen((size_t)nc_) {}

  // shortest augmenting path from row `start`; returns the free column it ends in (-1: infeasible)
  int64_t grow(int64_t start, double &reach) {
    std::fill(row_seen.begin(), row_seen.end(), 0);
    std::fill(col_seen.begin(), col_seen.end(), 0);
    std::fill(dist.begin(), dist.end(), std::numeric_limits<double>::infinity());
    int64_t n_todo = nc;
    for (int64_t k = 0; k < nc; k++) todo[(size_t)k] = nc - 1 - k;  // last column first
    double level = 0.0;
    int64_t i = start;
    for (;;) {
      row_seen[(size_t)i] = 1;
      int64_t pick = -1;
      double best = std::numeric_limits<double>::infinity();
      for (int64_t k = 0; k < n_todo; k++) {
        const int64_t j = todo[(size_t)k];
        const double via = level + c[(size_t)(i * nc + j)] - u[(size_t)i] - v[(size_t)j];
        if (via < dist[(size_t)j]) {
          dist[(size_t)j] = via;
          pred[(size_t)j] = i;
        }
        if (dist[(size_t)j] < best || (dist[(size_t)j] == best && row_of_col[(size_t)j] < 0)) {
          best = dist[(size_t)j];
          pick = k;
        }
      }
      level = best;
      if (pick < 0 || level == std::numeric_limits<double>::infinity()) return -1;
      const int64_t j = todo[(size_t)pick];
      col_seen[(size_t)j] = 1;
      todo[(size_t)pick] = todo[(size_t)(--n_todo)];
      if (row_of_col[(size_t)j] < 0) {
        reach = level;
        return j;
      }
      i = row_of_col[(size_t)j];
    }
  }

  int run() {
    for (int64_t row = 0; row < nr; row++) {
      double reach = 0.0;
      const int64_t sink = grow(row, reach);
      if (sink < 0) return -1;
      u[(size_t)row] += reach;  // dual update
      for (int64_t i = 0; i < nr; i++)
        if (row_seen[(size_t)i] && i != row) u[(size_t)i] += reach - dist[(size_t)col_of_row[(size_t)i]];
      for (int64_t j = 0; j < nc; j++)
        if (col_seen[(size_t)j]) v[(size_t)j] -= reach - dist[(size_t)j];
      for (int64_t j = sink;;) {  // flip the path
        const int64_t i = pred[(size_t)j];
        row_of_col[(size_t)j] = i;
        std::swap(col_of_row[(size_t)i], j);
        if (i == row) break;
      }
    }
    return 0;
  }
};
}  // namespace

// n x n, column-major cost: col_of_row[i] = column assigned to row i
int solve_linear_sum_assignment(int n, const double *cost, bool maximize, int64_t *col_of_row) {
  if (n < 1 || !cost || !col_of_row) return -2;
  std::vector<double> rm((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) rm[(size_t)i * n + j] = cost[(size_t)i + (size_t)n * j];
  std::vector<int64_t> rows((size_t)n);
  return ::solve_rectangular_linear_sum_assignment(n, n, rm.data(), maximize, rows.data(), col_of_row);
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
  // src/utils/utils.cpp:54-101, restated LITERALLY including its orientation: M(i, j) =
  // <Bov_i, Bm_j> + <Cov_i, Cm_j> (i = column of the overall model, j = column of the replica) is
  // built COLUMN-major and handed to the assignment solver, which reads ROW-major
  // (rectangular_lsap.cpp:93: cost[i * nc + j]).  The solver therefore works on M^T: its row r is
  // replica column r and solved[r] is the overall column matched to it.  The reference then sets
  // new(:, cur) = old(:, solved[cur]) -- the INVERSE of the permutation that would line the replica up
  // with the overall model (that one is new(:, solved[r]) = old(:, r)).  The two coincide when the
  // matching is an involution (identity, swaps: the common case, which is why the reference's own
  // FunctionCorrectness test cannot tell); for a 3-cycle they differ.  Parity with the reference is
  // the contract here, so the call and the copy are kept exactly as the reference has them
  // (DESIGN.md section 5 "Reference quirks kept"; tests/test_lsap_and_jk_permutation.py pins a 3-cycle).
  const auto &modes = ktensor.get_modes();
  const dim_t comp = ktensor.get_components();
  const Matrix &Bov = ktensor.get_factor(1), &Cov = ktensor.get_factor(2);
  for (dim_t m = 0; m < modes[0]; m++) {
    Ktensor &kt = jk_ktensor_v[m];
    const Matrix &Bm = kt.get_factor(1), &Cm = kt.get_factor(2);
    Matrix M(comp, comp);
    for (dim_t j = 0; j < comp; j++)
      for (dim_t i = 0; i < comp; i++) {
        double s = 0.0, t = 0.0;
        for (dim_t r = 0; r < modes[1]; r++) s += Bov(r, i) * Bm(r, j);
        for (dim_t r = 0; r < modes[2]; r++) t += Cov(r, i) * Cm(r, j);
        M(i, j) = s + t;
      }
    std::vector<int64_t> init_v(comp), solved_v(comp);
    ::solve_rectangular_linear_sum_assignment((intptr_t)comp, (intptr_t)comp, M.get_data(), true, init_v.data(),
                                              solved_v.data());
    for (dim_t mode = 0; mode < ktensor.get_n_modes(); mode++) {
      Matrix &f = kt.get_factor(mode);
      Matrix copy(f.get_rows(), f.get_cols());
      copy.copy(f);
      for (dim_t cur = 0; cur < comp; cur++) {
        const dim_t swap = (dim_t)solved_v[cur];
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

// extern/rectangular_lsap/rectangular_lsap.h:44, same contract: row-major nr x nc cost; on return
// (a[k], b[k]), k < min(nr, nc), are the assigned (row, column) pairs sorted by row.  Returns 0,
// -1 (infeasible) or -2 (NaN / -inf entry), the reference's RECTANGULAR_LSAP_* codes.
extern "C" int solve_rectangular_linear_sum_assignment(intptr_t nr, intptr_t nc, double *input_cost,
                                                        bool maximize, int64_t *a, int64_t *b) {
  if (nr == 0 || nc == 0) return 0;
  if (nr < 0 || nc < 0 || !input_cost || !a || !b) return -2;
  const bool tall = nc < nr;  // the solver wants rows <= columns: work on the transpose
  const int64_t R = tall ? nc : nr, Cn = tall ? nr : nc;
  cals::LsapSolver sv(R, Cn);
  for (int64_t i = 0; i < nr; i++)
    for (int64_t j = 0; j < nc; j++) {
      const double x = input_cost[(size_t)(i * nc + j)];
      sv.c[(size_t)(tall ? j * nr + i : i * nc + j)] = maximize ? -x : x;
    }
  double lo = sv.c[0];
  for (double x : sv.c) lo = (x < lo) ? x : lo;
  for (double &x : sv.c) {
    x -= lo;  // non-negative costs
    if (x != x || x == -std::numeric_limits<double>::infinity()) return -2;
  }
  if (sv.run()) return -1;
  if (!tall) {
    for (int64_t i = 0; i < nr; i++) {
      a[i] = i;
      b[i] = sv.col_of_row[(size_t)i];
    }
  } else {  // solver rows are the caller's columns: list the pairs by the caller's row
    std::vector<int64_t> order((size_t)R);
    for (int64_t k = 0; k < R; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(),
              [&](int64_t x, int64_t y) { return sv.col_of_row[(size_t)x] < sv.col_of_row[(size_t)y]; });
    for (int64_t k = 0; k < R; k++) {
      a[k] = sv.col_of_row[(size_t)order[(size_t)k]];
      b[k] = order[(size_t)k];
    }
  }
  return 0;
}

// C entry point of the assignment solver (tests bind it with ctypes)
extern "C" int cals_lsap_solve(int n, const double *cost_colmajor, int maximize, int64_t *col_of_row) {
  return cals::solve_linear_sum_assignment(n, cost_colmajor, maximize != 0, col_of_row);
}

// C entry point of utils::jk_permutation_adjustment (tests bind it with ctypes): `overall` = the n_modes
// factors of the overall model, `replicas` = modes[0] * n_modes factor pointers, replica-major; the
// replicas' columns are reordered in place.
extern "C" int cals_jk_permutation_adjustment(int n_modes, const int64_t *modes, int64_t rank,
                                               const double *const *overall, double *const *replicas) {
  if (n_modes < 3 || !modes || rank < 1 || !overall || !replicas) return -1;
  std::vector<dim_t> md(modes, modes + n_modes);
  cals::Ktensor ov((dim_t)rank, md);
  for (int n = 0; n < n_modes; n++) ov.set_factor(n, overall[n]);
  std::vector<cals::Ktensor> reps;
  for (dim_t m = 0; m < md[0]; m++) {
    reps.emplace_back((dim_t)rank, md);
    for (int n = 0; n < n_modes; n++) reps.back().set_factor(n, replicas[m * n_modes + n]);
  }
  cals::utils::jk_permutation_adjustment(ov, reps);
  for (dim_t m = 0; m < md[0]; m++)
    for (int n = 0; n < n_modes; n++) {
      const cals::Matrix &f = reps[m].get_factor((dim_t)n);
      std::copy(f.get_data(), f.get_data() + f.get_n_elements(), replicas[m * n_modes + n]);
    }
  return 0;
}
