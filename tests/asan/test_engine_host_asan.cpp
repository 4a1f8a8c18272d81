// The engine's HOST side under AddressSanitizer + UBSan, on the fake device of fake_device.cpp (no GPU, no
// libamdhip64): queueing, first-fit, eviction, compress, admission staging, asynchronous copy-out into
// the callers' storage, rebind, sweep log, several engines, the C++ layer above -- including the exact
// call pattern of the CLI driver (one cp_cals, then a loop of cp_als: 13 engine bindings) whose one
// unexplained exit-time SIGSEGV of round 1 motivated this harness.
// On the fake device the numeric kernels do nothing, so every model must come back from the engine
// BIT-IDENTICAL to what went in (fp32 engines: rounded to float once), with the iteration counts of the
// fake convergence rule -- any slip in the column bookkeeping shows as a wrong value, any slip in a size
// or an offset as a sanitizer report.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "als.h"
#include "cals.h"
#include "../../include/cals_hip.h"
#include "fake_device.h"

static int failures = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      failures++;                                                    \
    }                                                                \
  } while (0)

struct HostModel {
  int64_t rank;
  std::vector<std::vector<double>> f, f0;
  std::vector<double> lam, lam0;
  int64_t ticket = -1;
};

static std::vector<HostModel> make_models(const std::vector<int64_t> &modes, const std::vector<int> &ranks, unsigned seed) {
  std::mt19937 gen(seed);
  std::uniform_real_distribution<double> dist(-1.0, 1.0);
  std::vector<HostModel> out;
  for (int r : ranks) {
    HostModel m;
    m.rank = r;
    for (auto I : modes) {
      std::vector<double> v((size_t)(I * r));
      for (auto &x : v) x = dist(gen);
      m.f.push_back(v);
    }
    m.lam.resize((size_t)r);
    for (auto &x : m.lam) x = dist(gen);
    m.f0 = m.f;
    m.lam0 = m.lam;
    out.push_back(std::move(m));
  }
  return out;
}

static void c_abi_life_cycles(int dtype) {
  const std::vector<int64_t> modes = {23, 17, 9};
  std::mt19937 gen(7 + dtype);
  for (int round = 0; round < 7; round++) {  // 5, 6: ranks up to 150 (6: with the NNLS update and line search)
    std::vector<int> ranks;
    const int n_models = 20 + (int)(gen() % 60);
    for (int k = 0; k < n_models; k++) ranks.push_back(1 + (int)(gen() % (round >= 5 ? 150 : 12)));
    const int64_t buffer = round >= 5 ? 400 : 16 + (int64_t)(gen() % 40);
    auto models = make_models(modes, ranks, 100 + round);
    cals_hip_engine *e = nullptr;
    CHECK(cals_hip_create_ex(&e, 3, modes.data(), buffer, 0, dtype) == CALS_HIP_OK);
    std::vector<double> X((size_t)(23 * 17 * 9), 0.5);
    CHECK(cals_hip_set_tensor(e, X.data()) == CALS_HIP_OK);
    cals_hip_params p;
    cals_hip_default_params(&p);
    p.max_iterations = 25;
    p.line_search = (round % 2);
    p.update_method = (round == 3 || round == 6) ? 1 : 0;
    p.always_evict_first = (round == 4);
    CHECK(cals_hip_set_params(e, &p) == CALS_HIP_OK);
    if (round == 2) CHECK(cals_hip_set_sweep_log(e, 1) == CALS_HIP_OK);
    for (auto &m : models) {
      if (m.rank > buffer) continue;
      std::vector<double *> ptr;
      for (auto &v : m.f) ptr.push_back(v.data());
      CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), (round == 1 && m.rank % 2) ? 0 : -1, m.rank % 23,
                             &m.ticket) == CALS_HIP_OK);
    }
    cals_hip_report rep;
    if (round % 3 == 0) {
      CHECK(cals_hip_run(e, &rep) == CALS_HIP_OK);
    } else {  // the step-wise loop of the multi-GPU work queue
      int guard = 0;
      while ((cals_hip_queue_size(e) || cals_hip_models_in_flight(e)) && guard++ < 100000)
        CHECK(cals_hip_step(e, nullptr, nullptr) == CALS_HIP_OK);
      CHECK(cals_hip_synchronize(e) == CALS_HIP_OK);
      CHECK(cals_hip_get_report(e, &rep) == CALS_HIP_OK);
    }
    int64_t fitted = 0;
    for (auto &m : models) {
      if (m.ticket < 0) continue;
      fitted++;
      cals_hip_model_status st;
      CHECK(cals_hip_model_result(e, m.ticket, &st) == CALS_HIP_OK);
      CHECK(st.evicted == 1 && st.iters >= 1 && st.iters <= 25);
      double margin = -1.0;  // the fake line search leaves 0.25 behind for every model it "extrapolated"
      CHECK(cals_hip_debug_ls_margin(e, m.ticket, &margin) == CALS_HIP_OK && (margin == 1e300 || margin == 0.25));
      for (size_t n = 0; n < 3; n++)
        for (size_t i = 0; i < m.f[n].size(); i++) {
          const double want = dtype == CALS_HIP_F32 ? (double)(float)m.f0[n][i] : m.f0[n][i];
          if (m.f[n][i] != want) {
            CHECK(m.f[n][i] == want);
            i = m.f[n].size();
          }
        }
      CHECK(m.lam == m.lam0);
    }
    CHECK(rep.n_ktensors == fitted);
    if (round == 2) {
      std::vector<cals_hip_sweep_record> log(4096);
      const int64_t n = cals_hip_get_sweep_log(e, log.data(), 4096);
      CHECK(n == rep.iter || round % 3 != 0);
      CHECK(n > 0 && log[0].cols >= 1 && log[0].cols <= buffer);
    }
    // the next "cp_cals call" on the same tensor
    CHECK(cals_hip_rebind(e, buffer + 1) == CALS_HIP_ERR_FULL);
    CHECK(cals_hip_rebind(e, std::max<int64_t>(buffer / 2, 12)) == CALS_HIP_OK);
    auto again = make_models(modes, {3, 5, 2, 7, 1, 4}, 999);
    for (auto &m : again) {
      std::vector<double *> ptr;
      for (auto &v : m.f) ptr.push_back(v.data());
      CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), -1, 0, &m.ticket) == CALS_HIP_OK);
    }
    CHECK(cals_hip_run(e, &rep) == CALS_HIP_OK && rep.n_ktensors == 6);
    for (auto &m : again)
      for (size_t n = 0; n < 3; n++)
        for (size_t i = 0; i < m.f[n].size(); i += 7)
          CHECK(m.f[n][i] == (dtype == CALS_HIP_F32 ? (double)(float)m.f0[n][i] : m.f0[n][i]));
    CHECK(cals_hip_destroy(e) == CALS_HIP_OK);
  }
}

static void stepwise_api() {
  const std::vector<int64_t> modes = {12, 10, 8, 4};  // 4-way: Khatri-Rao workspace path
  cals_hip_engine *e = nullptr;
  CHECK(cals_hip_create(&e, 4, modes.data(), 24, 0) == CALS_HIP_OK);
  std::vector<double> X(12 * 10 * 8 * 4, 1.0);
  CHECK(cals_hip_set_tensor(e, X.data()) == CALS_HIP_OK);
  auto models = make_models(modes, {5, 9, 7, 6, 3}, 5);
  for (auto &m : models) {
    std::vector<double *> ptr;
    for (auto &v : m.f) ptr.push_back(v.data());
    CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), -1, 0, &m.ticket) == CALS_HIP_OK);
  }
  int64_t n = 0;
  CHECK(cals_hip_admit(e, &n) == CALS_HIP_OK && n == 3 && cals_hip_active_cols(e) == 21);
  CHECK(cals_hip_sweep(e, 4) == CALS_HIP_OK);
  std::vector<double> F(12 * 21), lam(21), G((size_t)CALS_HIP_MAX_RANK * 21);
  CHECK(cals_hip_debug_get_factor(e, 0, F.data()) == CALS_HIP_OK && F[0] == models[0].f0[0][0]);
  CHECK(cals_hip_debug_get_lambda(e, lam.data()) == CALS_HIP_OK);
  CHECK(cals_hip_debug_get_gramian(e, 3, G.data()) == CALS_HIP_OK);
  cals_hip_model_status st;
  int64_t col = -1;
  CHECK(cals_hip_debug_model_status(e, models[1].ticket, &st, &col) == CALS_HIP_OK && col == 5 && st.iters == 5);
  CHECK(cals_hip_evict(e, &n) == CALS_HIP_OK);
  cals_hip_report rep;
  CHECK(cals_hip_run(e, &rep) == CALS_HIP_OK);
  for (auto &m : models) CHECK(m.f == m.f0);
  CHECK(cals_hip_enqueue(e, 65, nullptr, nullptr, -1, 0, nullptr) != CALS_HIP_OK);
  CHECK(cals_hip_destroy(e) == CALS_HIP_OK);
}

// N > 3: the two-group dimension tree's host side (virtual layouts, Khatri-Rao workspace of a 3- and a 4-mode
// outer group, T and partial-tile sizing) for every N up to the maximum
static void n_way_group_tree() {
  for (const std::vector<int64_t> &modes : {std::vector<int64_t>{5, 4, 3, 6, 2}, std::vector<int64_t>{3, 4, 2, 3, 2, 4},
                                            std::vector<int64_t>{2, 3, 2, 3, 2, 3, 2}, std::vector<int64_t>{3, 2, 3, 2, 2, 3, 2, 2},
                                            std::vector<int64_t>{40, 33, 6, 50}}) {
    cals_hip_engine *e = nullptr;
    CHECK(cals_hip_create(&e, (int)modes.size(), modes.data(), 300, 0) == CALS_HIP_OK);
    CHECK(cals_hip_tree(e) == 4);
    size_t total = 1;
    for (auto d : modes) total *= (size_t)d;
    std::vector<double> X(total, 0.25);
    CHECK(cals_hip_set_tensor(e, X.data()) == CALS_HIP_OK);
    auto models = make_models(modes, {5, 9, 200, 7, 60, 3}, 11);
    for (auto &m : models) {
      std::vector<double *> ptr;
      for (auto &v : m.f) ptr.push_back(v.data());
      CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), -1, 0, &m.ticket) == CALS_HIP_OK);
    }
    cals_hip_report rep;
    CHECK(cals_hip_run(e, &rep) == CALS_HIP_OK);
    for (auto &m : models) CHECK(m.f == m.f0);
    CHECK(cals_hip_destroy(e) == CALS_HIP_OK);
  }
}

static void cpp_layer_driver_pattern() {
  using namespace cals;
  std::vector<dim_t> modes = {30, 25, 20};
  Tensor X(modes);
  X.randomize();
  std::vector<dim_t> components;
  for (int c = 1; c <= 4; c++)
    for (int k = 0; k < 3; k++) components.push_back((dim_t)c);
  std::vector<Ktensor> cals_input;
  for (auto c : components) {
    cals_input.emplace_back(c, modes);
    cals_input.back().randomize();
  }
  auto als_input(cals_input);
  const auto before(cals_input);
  KtensorQueue queue;
  for (auto &k : cals_input) queue.emplace(k);
  for (int precision = 0; precision < 2; precision++) {  // the driver's "-p f32" run mixes both types
    CalsParams cp;
    cp.max_iterations = 1000;
    cp.tol = 1e-5;
    cp.precision = precision ? CalsParams::FP32 : CalsParams::FP64;
    cp.buffer_size = std::accumulate(components.cbegin(), components.cend(), (dim_t)0);
    cp.with_time = true;
    if (queue.empty())
      for (auto &k : cals_input) queue.emplace(k);
    CalsReport rep = cp_cals(X, queue, cp);
    CHECK(rep.n_ktensors == 12 && queue.empty() && rep.cols.size() == rep.iter);
    AlsParams ap;
    ap.max_iterations = 1000;
    ap.tol = 1e-5;
    for (auto &k : als_input) cp_als(X, k, ap);  // 12 more bindings of the Tensor's device mirror
  }
  for (size_t k = 0; k < cals_input.size(); k++)
    for (dim_t n = 0; n < 3; n++)
      for (dim_t i = 0; i < before[k].get_factor(n).get_n_elements(); i++)
        if (cals_input[k].get_factor(n)[i] != (double)(float)before[k].get_factor(n)[i] &&
            cals_input[k].get_factor(n)[i] != before[k].get_factor(n)[i]) {
          CHECK(!"factor changed on the fake device");
          i = before[k].get_factor(n).get_n_elements();
        }
  // two engines on one device from two host threads, shared queue (CalsParams::devices)
  for (auto &k : cals_input) queue.emplace(k);
  CalsParams cp;
  cp.devices = {0, 0};
  cp.buffer_size = 9;
  cp.claim_models = 2;
  cp.line_search = true;
  CalsReport rep = cp_cals(X, queue, cp);
  CHECK(rep.n_ktensors == 12);
  // jackknife driver: generate replicas, one cp_cals, re-normalise, assignment, column reorder
  std::vector<Ktensor> originals(before.begin() + 6, before.begin() + 8);
  cp.devices.clear();
  cp.buffer_size = 3 * 30 * 2;
  JKReport jk = jk_cp_cals(X, originals, cp);
  CHECK(jk.results.size() == 2 && jk.results[0].size() == 30 && std::isnan(jk.results[1][4].get_factor(0)(4, 0)));
  // X rewritten behind the mirror: the fingerprint notices, the tensor is uploaded again
  X[0] += 1.0;
  for (auto &k : cals_input) queue.emplace(k);
  cp.buffer_size = 30;
  cp_cals(X, queue, cp);
  // engine errors surface as exceptions and leave the mirror usable
  Ktensor too_big(40, modes);
  too_big.randomize();
  queue.emplace(too_big);
  bool threw = false;
  try {
    cp_cals(X, queue, cp);
  } catch (const std::runtime_error &) {
    threw = true;
  }
  CHECK(threw);
  while (!queue.empty()) queue.pop();
  for (auto &k : cals_input) queue.emplace(k);
  CHECK(cp_cals(X, queue, cp).n_ktensors == 12);
}

// update::NNLS with models above rank 64 when the device cannot give the scratch nnls_huge_chunks asks for: the engine
// halves the workgroups per model until the allocation fits instead of failing the sweep (sweep_once)
static void nnls_scratch_under_memory_pressure() {
  const std::vector<int64_t> modes = {40, 17, 9};
  auto models = make_models(modes, {100, 7, 150, 90}, 31);
  cals_hip_engine *e = nullptr;
  CHECK(cals_hip_create(&e, 3, modes.data(), 400, 0) == CALS_HIP_OK);
  std::vector<double> X((size_t)(40 * 17 * 9), 0.5);
  CHECK(cals_hip_set_tensor(e, X.data()) == CALS_HIP_OK);
  cals_hip_params p;
  cals_hip_default_params(&p);
  p.max_iterations = 6;
  p.update_method = 1;
  CHECK(cals_hip_set_params(e, &p) == CALS_HIP_OK);
  for (auto &m : models) {
    std::vector<double *> ptr;
    for (auto &v : m.f) ptr.push_back(v.data());
    CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), -1, 0, &m.ticket) == CALS_HIP_OK);
  }
  // three models above rank 64, 10 row groups each at I = 40: 30 blocks of 4.7 MB asked for; 64 MB is all there is
  fake_set_malloc_limit((size_t)64 << 20);
  cals_hip_report rep;
  const int rc = cals_hip_run(e, &rep);
  if (rc != CALS_HIP_OK) std::printf("nnls under memory pressure: %s\n", cals_hip_last_error(e));
  CHECK(rc == CALS_HIP_OK && rep.n_ktensors == 4);
  fake_set_malloc_limit(0);
  for (auto &m : models) CHECK(m.f == m.f0);
  CHECK(cals_hip_destroy(e) == CALS_HIP_OK);
}

// The call patterns of the two round-3 anomalies, replayed on the fake device with the eviction
// schedule of the real run (tests/asan/patterns.txt, written by tools/make_asan_patterns.py from the oracle): the
// model admitted k-th leaves at the iteration at which the real run evicted it, so admission, eviction, compress and
// the pending-T bookkeeping take the very sequence of decisions they took on the GPU.  Every model must come back
// bit-identical with exactly that iteration count; CALS_HIP_VERIFY is on (free columns are checked before every sweep).
struct Pattern {
  std::string name, plan, api;
  std::vector<int64_t> modes;
  int64_t buffer = 0, max_iter = 0;
  int ls = 0, ls_method = 0, ls_interval = 0;
  std::vector<int> ranks;
  std::vector<long long> iters;
};

static std::vector<Pattern> read_patterns(const char *path) {
  std::vector<Pattern> out;
  std::ifstream in(path);
  std::string word;
  while (in >> word) {
    if (word != "pattern") break;
    Pattern p;
    p.modes.resize(3);
    size_t n = 0;
    in >> p.name >> p.modes[0] >> p.modes[1] >> p.modes[2] >> p.buffer >> p.max_iter >> p.ls >> p.ls_method >> p.ls_interval >>
        p.plan >> p.api >> n;
    p.ranks.resize(n);
    p.iters.resize(n);
    for (size_t k = 0; k < n; k++) in >> p.ranks[k] >> p.iters[k];
    out.push_back(p);
  }
  return out;
}

static void replay_patterns(const char *path) {
  const auto patterns = read_patterns(path);
  CHECK(patterns.size() == 2);
  setenv("CALS_HIP_VERIFY", "1", 1);
  for (const auto &p : patterns) {
    if (p.plan != "auto") setenv("CALS_HIP_TREE", p.plan.c_str(), 1);
    auto models = make_models(p.modes, p.ranks, 4242);
    cals_hip_engine *e = nullptr;
    CHECK(cals_hip_create(&e, 3, p.modes.data(), p.buffer, 0) == CALS_HIP_OK);
    CHECK(p.plan != "M" || cals_hip_tree(e) == 3);
    std::vector<double> X((size_t)(p.modes[0] * p.modes[1] * p.modes[2]), 0.5);
    CHECK(cals_hip_set_tensor(e, X.data()) == CALS_HIP_OK);
    cals_hip_params prm;
    cals_hip_default_params(&prm);
    prm.max_iterations = p.max_iter;
    prm.tol = 1e-5;
    prm.line_search = p.ls;
    prm.line_search_method = p.ls_method;
    prm.line_search_interval = p.ls_interval;
    CHECK(cals_hip_set_params(e, &prm) == CALS_HIP_OK);
    fake_set_schedule(p.iters);
    for (auto &m : models) {
      std::vector<double *> ptr;
      for (auto &v : m.f) ptr.push_back(v.data());
      CHECK(cals_hip_enqueue(e, m.rank, ptr.data(), m.lam.data(), -1, 0, &m.ticket) == CALS_HIP_OK);
    }
    auto check_model = [&](size_t k, const cals_hip_model_status &st) {
      CHECK(st.iters == p.iters[k]);
      CHECK(models[k].f == models[k].f0 && models[k].lam == models[k].lam0);
    };
    if (p.api == "step") {  // tests/test_gpu_async_eviction.py: result() polled after every step
      std::vector<char> done(models.size(), 0);
      int guard = 0;
      while ((cals_hip_queue_size(e) || cals_hip_models_in_flight(e)) && guard++ < 10000) {
        const int rc = cals_hip_step(e, nullptr, nullptr);
        if (rc != CALS_HIP_OK) std::printf("step: %s\n", cals_hip_last_error(e));
        CHECK(rc == CALS_HIP_OK);
        for (size_t k = 0; k < models.size(); k++) {
          if (done[k]) continue;
          cals_hip_model_status st;
          CHECK(cals_hip_model_result(e, models[k].ticket, &st) == CALS_HIP_OK);
          if (st.evicted) {  // whatever is reported as evicted must already be final
            done[k] = 1;
            check_model(k, st);
          }
        }
      }
      for (auto d : done) CHECK(d);
    } else {
      cals_hip_report rep;
      const int rc = cals_hip_run(e, &rep);
      if (rc != CALS_HIP_OK) std::printf("run: %s\n", cals_hip_last_error(e));
      CHECK(rc == CALS_HIP_OK && rep.n_ktensors == (int64_t)models.size());
      for (size_t k = 0; k < models.size(); k++) {
        cals_hip_model_status st;
        CHECK(cals_hip_model_result(e, models[k].ticket, &st) == CALS_HIP_OK && st.evicted);
        check_model(k, st);
      }
    }
    fake_set_schedule({});
    CHECK(cals_hip_destroy(e) == CALS_HIP_OK);
    unsetenv("CALS_HIP_TREE");
  }
  unsetenv("CALS_HIP_VERIFY");
}

int main(int argc, char **argv) {
  const char *patterns = argc > 1 ? argv[1] : nullptr;
  // twice: a device that completes everything at once, then one that is as late as the API allows (fake_device.cpp)
  // (CALS_HARNESS_ONLY_LATE=1: the late schedule only -- the ThreadSanitizer run, which is about the two host threads)
  for (int deferred = std::getenv("CALS_HARNESS_ONLY_LATE") ? 1 : 0; deferred < 2; deferred++) {
    fake_set_deferred(deferred != 0);
    c_abi_life_cycles(CALS_HIP_F64);
    c_abi_life_cycles(CALS_HIP_F32);
    stepwise_api();
    n_way_group_tree();
    cpp_layer_driver_pattern();
    nnls_scratch_under_memory_pressure();
    if (patterns) replay_patterns(patterns);
  }
  fake_set_deferred(false);
  std::printf(failures ? "engine host asan: %d FAILED\n" : "engine host asan: all checks passed\n", failures);
  return failures ? 1 : 0;
}
