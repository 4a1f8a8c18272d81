// The reference's tests/cals/test_cals.cpp:13-86 (SimpleCorrectness) restated on the C++ layer
// (cp-cals_amd/cals/cals.h -> C ABI -> HIP engine), with the CPU oracle as the comparator in place
// of cp_als.  Test infrastructure: links oracle/liboracle.so.  Prints the worst differences and
// returns non-zero on failure.
#include <cstdio>
#include <cstring>
#include <vector>

#include "als.h"
#include "cals.h"
#include "../../oracle/cals_oracle.h"

static uint64_t g_state = 0;
static double next_pm1() {  // splitmix64, as cp-cals_amd/inputs.py
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
}

int main(int argc, char **argv) {
  const bool ls = argc > 1 && !strcmp(argv[1], "ls");
  std::vector<dim_t> modes = {13, 12, 11};
  cals::Ktensor P(10, modes);
  P.fill([]() { return next_pm1(); });
  cals::Tensor T = P.to_tensor();

  std::vector<int> ranks;
  for (int r = 1; r <= 12; r++)
    for (int c = 0; c < 5; c++) ranks.push_back(1 + (r * 7 + c * 5) % 12);
  std::vector<cals::Ktensor> kts;
  for (int r : ranks) {
    kts.emplace_back((dim_t)r, modes);
    kts.back().fill([]() { return next_pm1(); });
  }
  // oracle copies
  std::vector<std::vector<std::vector<double>>> ofac(kts.size());
  std::vector<std::vector<double>> olam(kts.size());
  std::vector<or_model> om(kts.size());
  for (size_t i = 0; i < kts.size(); i++) {
    memset(&om[i], 0, sizeof(or_model));
    om[i].rank = (int64_t)kts[i].get_components();
    ofac[i].resize(3);
    for (int n = 0; n < 3; n++) {
      const auto &f = kts[i].get_factor(n);
      ofac[i][n].assign(f.get_data(), f.get_data() + f.get_n_elements());
      om[i].factors[n] = ofac[i][n].data();
    }
    olam[i] = kts[i].get_lambda();
    om[i].lambda = olam[i].data();
  }

  cals::CalsParams prm;
  prm.max_iterations = 200;
  prm.tol = 1e-5;
  prm.buffer_size = 30;
  prm.line_search = ls;
  prm.line_search_interval = 10;
  cals::KtensorQueue q;
  for (auto &k : kts) q.emplace(k);
  cals::CalsReport rep;
  try {
    rep = cals::cp_cals(T, q, prm);
  } catch (const std::exception &e) {
    fprintf(stderr, "cp_cals threw: %s\n", e.what());
    return 2;
  }

  or_params op;
  or_default_params(&op);
  op.max_iterations = 200;
  op.tol = 1e-5;
  op.buffer_size = 30;
  op.line_search = ls;
  op.line_search_interval = 10;
  op.mttkrp_method = OR_MTTKRP;
  or_report orep;
  int64_t omodes[3] = {13, 12, 11};
  or_set_threads(1);
  or_cp_cals(T.get_data(), 3, omodes, om.data(), (int64_t)om.size(), &op, &orep);

  double worst = 0.0;
  int bad_iters = 0;
  for (size_t i = 0; i < kts.size(); i++) {
    cals::Tensor a = kts[i].to_tensor();
    std::vector<double> b(a.get_n_elements());
    or_to_tensor(om[i].factors, om[i].lambda, 3, omodes, om[i].rank, b.data());
    double d = 0.0;
    for (dim_t e = 0; e < a.get_n_elements(); e++) d += (a[e] - b[e]) * (a[e] - b[e]);
    d = std::sqrt(d);
    if (d > worst) worst = d;
    if ((int64_t)kts[i].get_iters() != om[i].iters) bad_iters++;
  }
  printf("cp_cals C++ API: sweeps %zu (oracle %ld), models %d, ls %zu/%zu (oracle %ld/%ld), worst "
         "||T_gpu - T_oracle|| = %.3e, iteration mismatches %d, queue empty %d\n",
         (size_t)rep.iter, (long)orep.iter, rep.n_ktensors, (size_t)rep.ls_performed, (size_t)rep.ls_failed,
         (long)orep.ls_performed, (long)orep.ls_failed, worst, bad_iters, (int)q.empty());
  const bool ok = worst <= 1e-9 && bad_iters == 0 && (int64_t)rep.iter == orep.iter && q.empty() &&
                  (int64_t)rep.ls_performed == orep.ls_performed && (int64_t)rep.ls_failed == orep.ls_failed;
  // error behaviour: cuda = false must fail loudly (no CPU fallback)
  bool threw = false;
  try {
    cals::CalsParams p2;
    p2.cuda = false;
    cals::KtensorQueue q2;
    q2.emplace(kts[0]);
    cals::cp_cals(T, q2, p2);
  } catch (const std::exception &) {
    threw = true;
  }
  printf("cuda=false throws: %d\n", (int)threw);
  // CalsParams::devices: two engines (here on the same GPU, the box has one), one shared queue.  Every
  // model is fitted by exactly one engine with the single-device arithmetic, so the per-model results
  // must equal the oracle's again (same starting points: fresh copies of the initial factors).
  double worst_md = 0.0;
  int bad_md = 0, n_md = 0;
  {
    std::vector<cals::Ktensor> k2;
    g_state = 0;  // the same stream again: P2 (discarded) and the same starting points
    cals::Ktensor P2(10, modes);
    P2.fill([]() { return next_pm1(); });
    for (int r : ranks) {
      k2.emplace_back((dim_t)r, modes);
      k2.back().fill([]() { return next_pm1(); });
    }
    cals::CalsParams pm = prm;
    pm.devices = {0, 0};
    pm.claim_models = 3;
    cals::KtensorQueue q3;
    for (auto &k : k2) q3.emplace(k);
    cals::CalsReport r3;
    try {
      r3 = cals::cp_cals(T, q3, pm);
    } catch (const std::exception &e) {
      fprintf(stderr, "cp_cals(devices) threw: %s\n", e.what());
      return 3;
    }
    n_md = r3.n_ktensors;
    for (size_t i = 0; i < k2.size(); i++) {
      cals::Tensor a = k2[i].to_tensor();
      std::vector<double> b(a.get_n_elements());
      or_to_tensor(om[i].factors, om[i].lambda, 3, omodes, om[i].rank, b.data());
      double d = 0.0;
      for (dim_t e = 0; e < a.get_n_elements(); e++) d += (a[e] - b[e]) * (a[e] - b[e]);
      d = std::sqrt(d);
      if (d > worst_md) worst_md = d;
      if ((int64_t)k2[i].get_iters() != om[i].iters) bad_md++;
    }
  }
  printf("cp_cals with devices = {0, 0}: models %d, worst ||T_gpu - T_oracle|| = %.3e, iteration mismatches %d\n",
         n_md, worst_md, bad_md);
  const bool ok_md = worst_md <= 1e-9 && bad_md == 0 && n_md == (int)ranks.size();
  return (ok && threw && ok_md) ? 0 : 1;
}
