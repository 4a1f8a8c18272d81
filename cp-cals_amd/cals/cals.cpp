// cals::cp_cals / jk_cp_cals over the C ABI (include/cals_hip.h).  Reference boundary: include/cals.h:196,
// body src/cals.cpp:19-446.  All numerics run in libcals_hip.so; this file moves pointers, parameters and
// results across the boundary and keeps X's device copies alive between calls (DeviceMirror).
#include "cals.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "../../include/cals_hip.h"

namespace cals {

// ---------------------------------------------------------------------------------------------
// DeviceMirror: what Tensor keeps between cp_cals calls (the reference's CUDA path keeps `cudata`,
// include/tensor.h:56-59, and uploads only when it is null, src/cals.cpp:144-147).  One idle engine per
// (device, storage type) that has fitted models to this tensor: its padded permuted copies of X, its plan
// and its device buffers are reused through cals_hip_rebind; only the packing state starts over.
// ---------------------------------------------------------------------------------------------
struct DeviceMirror {
  struct Slot {
    cals_hip_engine *engine{nullptr};
    int device{0};
    int dtype{0};
    const double *data{nullptr};
    std::array<double, 1026> print{};  // sampled fingerprint of X at upload time
    bool busy{false};
  };
  std::mutex mu;
  std::vector<Slot> slots;
  ~DeviceMirror() {
    for (auto &s : slots)
      if (s.engine) cals_hip_destroy(s.engine);
  }
};

namespace {

[[noreturn]] void fail(cals_hip_engine *e, const char *what, int rc) {
  throw std::runtime_error(std::string("cp_cals: ") + what + " failed (" + std::to_string(rc) +
                           "): " + (e ? cals_hip_last_error(e) : "no engine"));
}

// 1024 evenly spaced elements + the size + the middle one.  Not a checksum (that would cost what the upload
// costs): it catches a Tensor that was refilled / rescaled / swapped behind its mirror, not a single edited
// element -- for that the caller says invalidate_device_mirror() (the reference's own device copy,
// include/tensor.h:56-59, is never refreshed at all).
std::array<double, 1026> fingerprint(const Tensor &X) {
  std::array<double, 1026> f{};
  const dim_t n = X.get_n_elements();
  if (n == 0) return f;
  for (dim_t k = 0; k < 1024; k++) f[k] = X[(dim_t)((long double)(n - 1) * k / 1023)];
  f[1024] = (double)n;
  f[1025] = X[n / 2];
  return f;
}

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

// An engine bound to X with `buffer_size` columns for the duration of one call: borrowed from X's
// DeviceMirror (created or re-targeted as needed) or, with reuse off, private to the call.
class EngineLease {
  std::shared_ptr<DeviceMirror> mirror;  // keeps the mirror alive even if X drops it meanwhile
  cals_hip_engine *e{nullptr};
  bool borrowed{false};
  bool ok{false};

  static cals_hip_engine *build(const Tensor &X, int64_t buffer_size, int device, int dtype) {
    std::vector<int64_t> modes(X.get_modes().begin(), X.get_modes().end());
    cals_hip_engine *eng = nullptr;
    // capacity = whole 128-column blocks (the device buffers are sized in those anyway): a loop of calls with
    // growing widths -- cp_als over ascending ranks, as the driver runs it -- then re-targets ONE engine
    // (cals_hip_rebind) instead of rebuilding it and uploading X again on every call
    const int64_t capacity = (buffer_size + 127) / 128 * 128;
    int rc = cals_hip_create_ex(&eng, (int)modes.size(), modes.data(), capacity, device, dtype);
    if (rc == 0) rc = cals_hip_set_tensor(eng, X.get_data());
    if (rc == 0 && capacity != buffer_size) rc = cals_hip_rebind(eng, buffer_size);
    if (rc) {
      const std::string msg = eng ? cals_hip_last_error(eng) : "no engine";
      if (eng) cals_hip_destroy(eng);
      throw std::runtime_error("cp_cals: creating the device engine failed (" + std::to_string(rc) + "): " + msg);
    }
    return eng;
  }

 public:
  EngineLease(const Tensor &X, int64_t buffer_size, int device, int dtype, bool reuse) {
    if (!reuse) {
      e = build(X, buffer_size, device, dtype);
      return;
    }
    static std::mutex create_mu;
    {
      std::lock_guard<std::mutex> g(create_mu);
      if (!X.device_mirror()) X.device_mirror() = std::make_shared<DeviceMirror>();
      mirror = X.device_mirror();
    }
    borrowed = true;
    const auto print = fingerprint(X);
    std::lock_guard<std::mutex> g(mirror->mu);
    mirror->slots.erase(std::remove_if(mirror->slots.begin(), mirror->slots.end(),
                                       [](const DeviceMirror::Slot &s) { return s.engine == nullptr; }),
                        mirror->slots.end());
    for (auto &s : mirror->slots) {
      if (s.busy || s.device != device || s.dtype != dtype) continue;
      if (cals_hip_capacity(s.engine) < buffer_size) {  // this call needs wider buffers: replace the engine
        cals_hip_destroy(s.engine);                       // (build uploads the current X: no re-upload first)
        s.engine = nullptr;
        s.engine = build(X, buffer_size, device, dtype);
        s.data = X.get_data();
        s.print = print;
      } else {
        if (s.data != X.get_data() || s.print != print) {  // X was rewritten behind the mirror: upload again
          const int rc = cals_hip_set_tensor(s.engine, X.get_data());
          if (rc) fail(s.engine, "cals_hip_set_tensor", rc);
          s.data = X.get_data();
          s.print = print;
        }
        const int rc = cals_hip_rebind(s.engine, buffer_size);
        if (rc) fail(s.engine, "cals_hip_rebind", rc);
      }
      s.busy = true;
      e = s.engine;
      return;
    }
    DeviceMirror::Slot s;
    s.engine = build(X, buffer_size, device, dtype);
    s.device = device;
    s.dtype = dtype;
    s.data = X.get_data();
    s.print = print;
    s.busy = true;
    mirror->slots.push_back(s);
    e = s.engine;
  }
  EngineLease(const EngineLease &) = delete;
  EngineLease &operator=(const EngineLease &) = delete;
  void done() { ok = true; }  // the run finished: the engine is idle and reusable
  cals_hip_engine *get() const { return e; }
  ~EngineLease() {
    if (!e) return;
    if (!borrowed) {
      cals_hip_destroy(e);
      return;
    }
    std::lock_guard<std::mutex> g(mirror->mu);
    for (auto &s : mirror->slots)
      if (s.engine == e) {
        if (ok) {
          s.busy = false;
        } else {  // an error left the engine in an unknown state: do not hand it out again
          cals_hip_destroy(e);
          s.engine = nullptr;
        }
      }
  }
};

void enqueue_model(cals_hip_engine *e, Ktensor &kt, int64_t *ticket) {
  std::vector<double *> fptr;
  for (auto &f : kt.get_factors()) fptr.push_back(f.get_data());
  const int rc = cals_hip_enqueue(e, (int64_t)kt.get_components(), fptr.data(), kt.get_lambda().data(),
                                  kt.is_jk() ? (int)kt.get_jk_mode() : -1, (int64_t)kt.get_jk_fiber(), ticket);
  if (rc) fail(e, "cals_hip_enqueue", rc);
}

void read_back(cals_hip_engine *e, Ktensor &kt, int64_t ticket) {
  cals_hip_model_status st;
  const int rc = cals_hip_model_result(e, ticket, &st);
  if (rc) fail(e, "cals_hip_model_result", rc);
  kt.set_iters((dim_t)st.iters);
  kt.set_approximation_error(st.approx_error);
  kt.set_fit(st.fit, st.old_fit);
}

// CalsReport's per-iteration matrices from the engine's sweep log (src/cals.cpp:54-64, 213-217, 269-275,
// 367-370 fill them from chrono timers around the CPU phases)
void fill_timers(cals_hip_engine *e, CalsReport &rep) {
  const int64_t n = cals_hip_get_sweep_log(e, nullptr, 0);
  std::vector<cals_hip_sweep_record> log((size_t)std::max<int64_t>(n, 1));
  cals_hip_get_sweep_log(e, log.data(), n);
  const dim_t its = (dim_t)std::max<int64_t>(n, 1), N = rep.n_modes;
  rep.als_times = Matrix(AlsTimers::LENGTH, its);
  rep.mode_times = Matrix(ModeTimers::LENGTH * N, its);
  rep.mttkrp_times = Matrix(MttkrpTimers::LENGTH * N, its);
  rep.als_times.zero();
  rep.mode_times.zero();
  rep.mttkrp_times.zero();
  rep.flops_per_iteration.assign(its, 0);
  rep.cols.assign(its, 0);
  for (int64_t k = 0; k < n; k++) {
    const cals_hip_sweep_record &L = log[(size_t)k];
    const dim_t it = (dim_t)k;
    rep.cols[it] = (dim_t)L.cols;
    rep.flops_per_iteration[it] = (uint64_t)L.flops;
    rep.als_times(AlsTimers::ITERATION, it) = L.iteration_ms * 1e-3;
    rep.als_times(AlsTimers::DEFRAGMENTATION, it) = L.defrag_ms * 1e-3;
    rep.als_times(AlsTimers::LINE_SEARCH, it) = L.ls_ms * 1e-3;
    for (dim_t m = 0; m < N && m < CALS_HIP_MAX_MODES; m++) {
      rep.mode_times(m * ModeTimers::LENGTH + ModeTimers::MTTKRP, it) = L.mttkrp_ms[m] * 1e-3;
      rep.mode_times(m * ModeTimers::LENGTH + ModeTimers::UPDATE, it) = L.update_ms[m] * 1e-3;
      rep.mttkrp_times(m * MttkrpTimers::LENGTH + MttkrpTimers::MT_KRP, it) = L.krp_ms[m] * 1e-3;
      rep.mttkrp_times(m * MttkrpTimers::LENGTH + MttkrpTimers::MT_GEMM, it) = L.fused_ms[m] * 1e-3;
      rep.mttkrp_times(m * MttkrpTimers::LENGTH + MttkrpTimers::TS_GEMM, it) = L.ttm_ms[m] * 1e-3;
      rep.mttkrp_times(m * MttkrpTimers::LENGTH + MttkrpTimers::TS_GEMV, it) = L.contract_ms[m] * 1e-3;
    }
  }
}

// CalsParams::devices with more than one entry: one engine (and one host thread) per device, each
// with its own replica of X; the models sit behind one shared counter and a device claims a few
// more whenever none of its claimed models is waiting for buffer columns (the pull-based hand-off
// of cp-cals_amd/multi_gpu.py, here inside one process: no collective, nothing but indices shared).
// Every model is fitted by exactly one device with the arithmetic of the single-device path.
void cp_cals_devices(const Tensor &X, std::vector<std::reference_wrapper<Ktensor>> &all, const CalsParams &p,
                     CalsReport &rep) {
  const size_t n_dev = p.devices.size();
  std::atomic<size_t> next{0};
  std::vector<cals_hip_report> reports(n_dev);
  std::vector<std::exception_ptr> errors(n_dev);
  const int dtype = p.precision == CalsParams::FP32 ? CALS_HIP_F32 : CALS_HIP_F64;
  auto worker = [&](size_t d) {
    try {
      EngineLease lease(X, (int64_t)p.buffer_size, p.devices[d], dtype, p.reuse_device_tensor);
      cals_hip_engine *e = lease.get();
      cals_hip_params hp = to_hip_params(p);
      int rc = cals_hip_set_params(e, &hp);
      if (rc) fail(e, "cals_hip_set_params", rc);
      if (p.with_time && d == 0) cals_hip_set_sweep_log(e, 1);
      std::vector<std::pair<size_t, int64_t>> mine;  // (index into all, ticket)
      bool drained = false;
      const size_t claim = (size_t)std::max(1, p.claim_models);
      for (;;) {
        if (!drained && cals_hip_queue_size(e) == 0) {
          const size_t lo = next.fetch_add(claim);
          if (lo >= all.size()) drained = true;
          for (size_t i = lo; i < std::min(lo + claim, all.size()); i++) {
            int64_t ticket = -1;
            enqueue_model(e, all[i], &ticket);
            mine.emplace_back(i, ticket);
          }
        }
        if (cals_hip_queue_size(e) == 0 && cals_hip_models_in_flight(e) == 0) {
          if (drained) break;
          continue;
        }
        if ((rc = cals_hip_step(e, nullptr, nullptr))) fail(e, "cals_hip_step", rc);
      }
      for (auto &m : mine) read_back(e, all[m.first], m.second);
      if ((rc = cals_hip_get_report(e, &reports[d]))) fail(e, "cals_hip_get_report", rc);
      if (d == 0) {
        rep.mttkrp_plan = cals_hip_tree(e);
        if (p.with_time) {  // the first device's log stands for the run
          fill_timers(e, rep);
          cals_hip_set_sweep_log(e, 0);
        }
      }
      lease.done();
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

namespace mttkrp {
// see utils/mttkrp.h; include/utils/mttkrp.h:77-81, src/utils/mttkrp.cpp:562-614
Matrix &mttkrp(const Tensor &X, Ktensor &u, std::vector<Matrix> &workspace, dim_t mode, MttkrpParams &params) {
  (void)workspace;
  const dim_t n_modes = X.get_n_modes();
  if (mode >= n_modes) throw std::runtime_error("mttkrp: mode out of range");
  if (u.get_factors().size() != n_modes) throw std::runtime_error("mttkrp: Ktensor and Tensor differ in modes");
  const dim_t r = u.get_components();
  EngineLease lease(X, (int64_t)r, 0, CALS_HIP_F64, true);
  std::vector<const double *> f(n_modes);
  for (dim_t n = 0; n < n_modes; n++) f[n] = u.get_factor(n).get_data();
  Matrix &G = u.get_factor(mode);
  double ms = 0.0;
  const int rc = cals_hip_mttkrp(lease.get(), (int64_t)r, f.data(), (int)mode, G.get_data(), &ms);
  if (rc) fail(lease.get(), "cals_hip_mttkrp", rc);
  lease.done();
  for (int i = 0; i < MttkrpTimers::LENGTH; i++) params.mttkrp_timers.timers[i].reset();
  params.mttkrp_timers.timers[MttkrpTimers::MT_GEMM].set_time(ms * 1e-3);
  uint64_t in_out = 0;
  for (dim_t n = 0; n < n_modes; n++) in_out += (uint64_t)X.get_modes()[n] * r;
  params.flops = 2ull * (uint64_t)X.get_n_elements() * r;
  params.memops = (uint64_t)X.get_n_elements() + in_out;
  return G;
}
}  // namespace mttkrp

void CalsParams::print() const {
  using std::cout;
  using std::endl;
  cout << "---------------------------------------" << endl;
  cout << "CALS parameters" << endl;
  cout << "---------------------------------------" << endl;
  cout << "Tol:             " << tol << endl;
  cout << "Max Iterations:  " << max_iterations << endl;
  cout << "Buffer Size:     " << buffer_size << endl;
  cout << "Mttkrp Method:   " << mttkrp::mttkrp_method_names[mttkrp_method] << " (the device engine picks its own plan)" << endl;
  cout << "Update Method:   " << update::update_method_names[update_method] << endl;
  cout << "Line Search:     " << (line_search ? "true" : "false") << endl;
  if (line_search) {
    cout << "-Line Search Interval: " << line_search_interval << " iterations" << endl;
    cout << "-Line Search Method:   " << ls::ls_method_names[line_search_method] << endl;
  }
  cout << "CUDA:            " << (cuda ? "true" : "false") << " (device path: MI355X HIP engine, device";
  if (devices.empty())
    cout << " " << device;
  else
    for (int d : devices) cout << " " << d;
  cout << ", " << (precision == FP32 ? "fp32" : "fp64") << " storage)" << endl;
  cout << "---------------------------------------" << endl;
}

void CalsReport::print_header(const std::string &file_name, const std::string &sep) const {
  std::ofstream file(file_name, std::ios::out);
  AlsTimers als_timers;
  ModeTimers mode_timers;
  for (const char *col : {"TENSOR_RANK", "TENSOR_MODES", "BUFFER_SIZE", "N_KTENSORS", "KTENSOR_COMP_SUM", "UPDATE_METHOD",
                          "LINE_SEARCH", "MAX_ITERS", "ITER", "NUM_THREADS", "TOTAL"})
    file << col << sep;
  if (!flops_per_iteration.empty()) {
    file << "FLOPS" << sep << "COLS" << sep;
    for (const auto &name : als_timers.names) file << name << sep;
    for (dim_t m = 0; m < modes.size(); m++)
      for (const auto &name : mode_timers.names) file << "MODE_" << m << "_" << name << sep;
  }
  file << std::endl;
}

void CalsReport::print_to_file(const std::string &file_name, const std::string &sep) const {
  std::ofstream file(file_name, std::ios::app);
  const bool timed = !flops_per_iteration.empty();
  for (dim_t it = 0; it < iter; it++) {  // one row per outer iteration
    file << tensor_rank << sep << utils::mode_string(modes) << sep << buffer_size << sep << n_ktensors << sep
         << ktensor_comp_sum << sep << update::update_method_names[update_method] << sep << line_search << sep
         << max_iter << sep << it + 1 << sep << n_threads << sep << total_time << sep;
    if (timed && it < flops_per_iteration.size()) {
      file << flops_per_iteration[it] << sep << cols[it] << sep << std::scientific;
      for (dim_t r = 0; r < als_times.get_rows(); r++) file << als_times(r, it) << sep;
      for (dim_t r = 0; r < mode_times.get_rows(); r++) file << mode_times(r, it) << sep;
      file << std::defaultfloat;
    }
    file << std::endl;
  }
}

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
  rep.n_threads = get_threads();
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
  EngineLease lease(X, (int64_t)p.buffer_size, p.devices.size() == 1 ? p.devices[0] : p.device,
                    p.precision == CalsParams::FP32 ? CALS_HIP_F32 : CALS_HIP_F64, p.reuse_device_tensor);
  cals_hip_engine *e = lease.get();
  cals_hip_params hp = to_hip_params(p);
  int rc = cals_hip_set_params(e, &hp);
  if (rc) fail(e, "cals_hip_set_params", rc);
  if (p.with_time) cals_hip_set_sweep_log(e, 1);

  std::vector<std::reference_wrapper<Ktensor>> kts;
  std::vector<int64_t> tickets;
  while (!kt_queue.empty()) {
    Ktensor &kt = kt_queue.front();
    int64_t ticket = -1;
    enqueue_model(e, kt, &ticket);
    kts.push_back(kt);
    tickets.push_back(ticket);
    kt_queue.pop();
  }
  cals_hip_report hr;
  if ((rc = cals_hip_run(e, &hr))) fail(e, "cals_hip_run", rc);
  for (size_t i = 0; i < kts.size(); i++) read_back(e, kts[i], tickets[i]);
  rep.X_norm = hr.X_norm;
  rep.iter = (dim_t)hr.iter;
  rep.n_ktensors = (int)hr.n_ktensors;
  rep.ktensor_comp_sum = (int)hr.ktensor_comp_sum;
  rep.ls_performed = (dim_t)hr.ls_performed;
  rep.ls_failed = (dim_t)hr.ls_failed;
  rep.mttkrp_plan = cals_hip_tree(e);
  if (p.with_time) {
    fill_timers(e, rep);
    cals_hip_set_sweep_log(e, 0);
  }
  lease.done();
  rep.total_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rep;
}

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

}  // namespace cals
