// cals::jk_cp_cals (C++ layer -> C ABI -> HIP engine) against the oracle's jk_cp_cals: the
// reference's CalsJackknifingTests.FunctionCorrectness (tests/cals/test_cals.cpp:299-362) with the
// oracle in place of jk_cp_als.  Test infrastructure: links oracle/liboracle.so.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "als.h"
#include "cals.h"
#include "../../oracle/cals_oracle.h"

static uint64_t g_state = 12345;
static double next_pm1() {
  uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
}

int main() {
  const dim_t n_ktensors = 3, components = 5;
  std::vector<dim_t> modes = {10, 21, 20};
  const int64_t omodes[3] = {10, 21, 20};
  cals::Ktensor P(components, modes);
  P.fill([]() { return next_pm1(); });
  cals::Tensor T = P.to_tensor();

  std::vector<cals::Ktensor> refs;
  for (dim_t k = 0; k < n_ktensors; k++) {
    refs.emplace_back(components, modes);
    refs.back().fill([]() { return next_pm1(); });
  }
  cals::CalsParams prm;
  prm.max_iterations = 40;
  prm.tol = 1e-4;
  prm.buffer_size = 18;
  prm.force_max_iter = true;
  // first fit the overall models (test_cals.cpp:339-340 does this with cp_als)
  cals::AlsParams ap;
  ap.max_iterations = 40;
  ap.tol = 1e-4;
  ap.force_max_iter = true;
  try {
    for (auto &k : refs) cals::cp_als(T, k, ap);
  } catch (const std::exception &e) {
    fprintf(stderr, "cp_als threw: %s\n", e.what());
    return 2;
  }

  // oracle inputs = the fitted overall models
  std::vector<std::vector<std::vector<double>>> of(n_ktensors);
  std::vector<std::vector<double>> ol(n_ktensors);
  std::vector<or_model> om(n_ktensors);
  for (dim_t k = 0; k < n_ktensors; k++) {
    memset(&om[k], 0, sizeof(or_model));
    om[k].rank = components;
    of[k].resize(3);
    for (int n = 0; n < 3; n++) {
      const auto &f = refs[k].get_factor(n);
      of[k][n].assign(f.get_data(), f.get_data() + f.get_n_elements());
      om[k].factors[n] = of[k][n].data();
    }
    ol[k] = refs[k].get_lambda();
    om[k].lambda = ol[k].data();
  }
  const size_t n_res = n_ktensors * modes[0];
  std::vector<or_model> res(n_res);
  std::vector<std::vector<std::vector<double>>> rf(n_res);
  std::vector<std::vector<double>> rl(n_res);
  for (size_t i = 0; i < n_res; i++) {
    memset(&res[i], 0, sizeof(or_model));
    rf[i].resize(3);
    for (int n = 0; n < 3; n++) {
      rf[i][n].assign(modes[n] * components, 0.0);
      res[i].factors[n] = rf[i][n].data();
    }
    rl[i].assign(components, 0.0);
    res[i].lambda = rl[i].data();
  }
  or_params op;
  or_default_params(&op);
  op.max_iterations = 40;
  op.tol = 1e-4;
  op.buffer_size = 18;
  op.force_max_iter = 1;
  op.mttkrp_method = OR_MTTKRP;
  or_set_threads(1);
  or_report orep;
  if (or_jk_cp_cals(T.get_data(), 3, omodes, om.data(), (int64_t)n_ktensors, &op, res.data(), &orep)) {
    fprintf(stderr, "oracle jk failed\n");
    return 3;
  }

  cals::JKReport rep;
  try {
    rep = cals::jk_cp_cals(T, refs, prm);
  } catch (const std::exception &e) {
    fprintf(stderr, "jk_cp_cals threw: %s\n", e.what());
    return 2;
  }
  double worst = 0.0;
  const int64_t jm[3] = {9, 21, 20};
  for (dim_t k = 0; k < n_ktensors; k++)
    for (dim_t i = 0; i < modes[0]; i++) {
      cals::Tensor a = rep.results[k][i].to_regular().to_tensor();
      // oracle replica without its (NaN) fiber row
      const or_model &m = res[k * modes[0] + i];
      std::vector<double> f0(9 * components);
      for (dim_t c = 0; c < components; c++)
        for (dim_t r = 0, o = 0; r < modes[0]; r++)
          if (r != i) f0[(o++) + 9 * c] = m.factors[0][r + modes[0] * c];
      double *facs[3] = {f0.data(), m.factors[1], m.factors[2]};
      std::vector<double> b(a.get_n_elements());
      or_to_tensor(facs, m.lambda, 3, jm, (int64_t)components, b.data());
      double d = 0.0;
      for (dim_t e = 0; e < a.get_n_elements(); e++) d += (a[e] - b[e]) * (a[e] - b[e]);
      d = std::sqrt(d);
      if (!(d <= worst)) worst = d;  // NaN propagates
      // column order must agree too: compare factor 1 directly
      double fd = 0.0;
      for (dim_t e = 0; e < modes[1] * components; e++) {
        const double x = rep.results[k][i].get_factor(1).get_data()[e] - m.factors[1][e];
        fd += x * x;
      }
      if (!(std::sqrt(fd) <= 1e-8)) worst = (worst > std::sqrt(fd)) ? worst : std::sqrt(fd);
    }
  printf("jk_cp_cals C++ API: %zu models x %zu replicas, worst ||T_gpu - T_oracle|| / factor diff = %.3e\n",
         (size_t)n_ktensors, (size_t)modes[0], worst);

  // The reference's own criterion (tests/cals/test_cals.cpp:343-361): jk_cp_cals == jk_cp_als, the
  // jackknife by plain ALS on the sub-sampled tensors (here: the same engine on X without slice i).
  double worst_als = 0.0;
  try {
    cals::JKReport rep2 = cals::jk_cp_als(T, refs, ap);
    for (dim_t k = 0; k < n_ktensors; k++)
      for (dim_t i = 0; i < modes[0]; i++) {
        cals::Tensor a = rep.results[k][i].to_regular().to_tensor();
        cals::Tensor b = rep2.results[k][i].to_tensor();
        double d = 0.0;
        for (dim_t e = 0; e < a.get_n_elements(); e++) d += (a[e] - b[e]) * (a[e] - b[e]);
        d = std::sqrt(d);
        if (!(d <= worst_als)) worst_als = d;
      }
  } catch (const std::exception &e) {
    fprintf(stderr, "jk_cp_als threw: %s\n", e.what());
    return 2;
  }
  printf("jk_cp_cals vs jk_cp_als (sub-sampled tensors): worst ||T_cals - T_als|| = %.3e\n", worst_als);

  // cp_omp_als == cp_als model by model (include/als.h:218)
  double worst_omp = 0.0;
  {
    std::vector<cals::Ktensor> a_in, b_in;
    for (dim_t k = 0; k < 4; k++) {
      a_in.emplace_back((dim_t)(2 + k), modes);
      a_in.back().fill([]() { return next_pm1(); });
    }
    b_in = a_in;
    auto reps = cals::cp_omp_als(T, a_in, ap);
    for (size_t k = 0; k < b_in.size(); k++) {
      cals::AlsReport r1 = cals::cp_als(T, b_in[k], ap);
      if (r1.iter != reps[k].iter) worst_omp = 1.0;
      for (int n = 0; n < 3; n++) {
        const auto &fa = a_in[k].get_factor(n), &fb = b_in[k].get_factor(n);
        for (dim_t e = 0; e < fa.get_n_elements(); e++) {
          const double x = std::fabs(fa.get_data()[e] - fb.get_data()[e]);
          if (!(x <= worst_omp)) worst_omp = x;
        }
      }
    }
  }
  printf("cp_omp_als vs cp_als: worst factor entry difference = %.3e\n", worst_omp);
  // tests/als/test_als.cpp:62-103 (ComputeCorrectResultConstrained3D) through the C++ API:
  // update::NNLS gives non-negative factors and a finite reconstruction error equal to the fast one
  double worst_neg = 0.0, nnls_err_gap = 0.0;
  {
    cals::AlsParams np = ap;
    np.update_method = cals::update::UPDATE_METHOD::NNLS;
    np.max_iterations = 100;
    cals::Ktensor k(5, modes);
    k.fill([]() { return 0.5 * (next_pm1() + 1.0); });
    cals::cp_als(T, k, np);
    for (int n = 0; n < 3; n++) {
      const auto &f = k.get_factor(n);
      for (dim_t e = 0; e < f.get_n_elements(); e++)
        if (!(f.get_data()[e] >= -worst_neg)) worst_neg = -f.get_data()[e];
    }
    cals::Tensor a = k.to_tensor();
    double d = 0.0;
    for (dim_t e = 0; e < a.get_n_elements(); e++) d += (T[e] - a[e]) * (T[e] - a[e]);
    d = std::sqrt(d);
    nnls_err_gap = std::fabs(d - k.get_approximation_error());
    if (!std::isfinite(d) || std::isnan(worst_neg)) worst_neg = 1.0;
  }
  printf("cp_als with update::NNLS: most negative entry = %.3e, |slow - fast error| = %.3e\n", worst_neg,
         nnls_err_gap);
  // source-compatibility members the front-ends use (include/ktensor.h:161-170, utils/mttkrp.h:100,
  // extern/rectangular_lsap/rectangular_lsap.h:44)
  {
    cals::Ktensor k(2, modes);
    std::vector<double> f0(modes[0] * 2, 0.5), l2 = {3.0, 4.0};
    k.set_factor(0, f0.data());
    k.set_lambda(l2.data());
    if (k.get_factor(0)(1, 1) != 0.5 || k.get_lambda()[1] != 4.0) return 4;
    prm.mttkrp_lut = cals::mttkrp::read_lookup_table(modes, 1, true);
    double cost[9] = {4, 1, 3, 2, 0, 5, 3, 2, 2};  // row-major 3 x 3, minimum 1 + 2 + 2 = 5
    int64_t ra[3], cb[3];
    if (solve_rectangular_linear_sum_assignment(3, 3, cost, false, ra, cb) != 0) return 5;
    double tot = 0.0;
    for (int i = 0; i < 3; i++) tot += cost[ra[i] * 3 + cb[i]];
    if (tot != 5.0) return 6;
  }
  return (worst <= 1e-9 && worst_als <= 1e-8 && worst_omp <= 1e-10 && worst_neg == 0.0 && nnls_err_gap <= 1e-8) ? 0 : 1;
}
