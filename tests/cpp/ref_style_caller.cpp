// A caller written the way the reference's front-ends are written -- same includes, same unqualified /
// qualified symbols, same call pattern -- to show that code against HPAC/CP-CALS's headers builds and runs
// against this repo's header set without edits.  Own code (not a copy of any reference file):
//   part 1: the pattern of src/examples/driver.cpp:128-222 (global set_threads, Tensor::randomize,
//           Ktensor assignment + randomize, KtensorQueue, CalsParams / AlsParams fields incl. the nested
//           enum spellings, std::accumulate through the headers' <numeric>, print(), cals::Timer,
//           unqualified cp_cals / cp_als found by ADL);
//   part 2: the pattern of the MEX glue (matlab/matlab.cpp:91-190, matlab_cp_cals_jk.cpp:118-215):
//           Tensor view of foreign memory, set_lambda / set_factor, generate_jk_ktensors, one cp_cals over
//           all replicas, set_jk_fiber / denormalize / normalize, cblas_dgemm + the assignment solver,
//           Matrix views of single columns, concatenate_ktensors;
//   part 3: CalsReport::print_header / print_to_file with the timer matrices (experiments_utils.cpp:163-187);
//   part 4: the MTTKRP micro-benchmark's pattern (include/experiments/bench_mttkrp_cals.h:49-84): mttkrp::mttkrp on
//           one Ktensor, its timers / flops, and error::compute_fast_error on matrices the caller holds.
// Usage: ref_style_caller I-J-K MIN:MAX:COPIES [csv]   (small sizes; exit code 0 = all checks passed)
#include <iostream>
#include <random>
#include <sstream>

#include "als.h"
#include "cals.h"
#include "timer.h"
#include "utils/error.h"
#include "utils/mttkrp.h"
#include "utils/utils.h"
#include <rectangular_lsap/rectangular_lsap.h>

using std::cerr;
using std::cout;
using std::endl;
using std::string;
using std::vector;

using cals::Ktensor;
using cals::Tensor;

static void split(const std::string &str, vector<dim_t> &vect, char delimiter) {
  std::stringstream ss(str);
  std::string token;
  while (std::getline(ss, token, delimiter)) vect.push_back(std::strtoul(token.c_str(), nullptr, 10));
}

static int check(bool ok, const char *what) {
  if (!ok) cerr << "FAILED: " << what << endl;
  return ok ? 0 : 1;
}

int main(int argc, char *argv[]) {
  vector<dim_t> modes, comp;
  split(argc > 1 ? argv[1] : "12-10-8", modes, '-');
  split(argc > 2 ? argv[2] : "1:3:2", comp, ':');
  if (modes.size() < 3 || comp.size() != 3) return 2;
  int bad = 0;

  // ------------------------------------------------------------------ part 1: the CLI driver's pattern
  set_threads(4);
  cals::Tensor X(modes);
  X.randomize();

  vector<int> components;
  for (auto c = static_cast<int>(comp[0]); c <= static_cast<int>(comp[1]); c++)
    for (auto cp = 0; cp < static_cast<int>(comp[2]); cp++) components.push_back(c);

  vector<cals::Ktensor> cals_input(components.size());
  auto i = 0;
  for (auto &ktensor : cals_input) {
    ktensor = cals::Ktensor(components[i++], modes);
    ktensor.randomize();
  }
  auto als_input(cals_input);

  cals::KtensorQueue cals_queue;
  for (auto &p : cals_input) cals_queue.emplace(p);

  cals::CalsParams cals_params;
  cals_params.mttkrp_method = cals::mttkrp::MTTKRP_METHOD::AUTO;
  cals_params.update_method = cals::update::UPDATE_METHOD::UNCONSTRAINED;
  cals_params.max_iterations = 60;
  cals_params.tol = 1e-6;
  cals_params.buffer_size = std::accumulate(components.cbegin(), components.cend(), static_cast<dim_t>(0));
  cals_params.with_time = true;
  cals_params.print();

  cals::Timer cals_timer;
  cals_timer.start();
  auto cals_report = cp_cals(X, cals_queue, cals_params);
  cals_timer.stop();
  bad += check(cals_queue.empty(), "queue drained");
  bad += check(cals_report.n_ktensors == static_cast<int>(components.size()), "all models fitted");
  bad += check(cals_report.n_threads == 4 && get_threads() == 4, "set_threads is recorded");

  cals::AlsParams als_params;
  als_params.mttkrp_method = cals::mttkrp::MTTKRP_METHOD::AUTO;
  als_params.update_method = cals::update::UPDATE_METHOD::UNCONSTRAINED;
  als_params.max_iterations = 60;
  als_params.tol = 1e-6;
  als_params.suppress_lut_warning = true;
  als_params.print();

  cals::Timer als_timer;
  als_timer.start();
  for (auto &kt : als_input) auto als_report = cp_als(X, kt, als_params);
  als_timer.stop();
  // CALS == ALS per model (tests/cals/test_cals.cpp:60-86): same engine arithmetic, one model at a time
  double worst = 0.0;
  for (size_t k = 0; k < cals_input.size(); k++) {
    bad += check(cals_input[k].get_iters() == als_input[k].get_iters(), "iterations equal");
    Tensor a = cals_input[k].to_tensor(), b = als_input[k].to_tensor();
    double d = 0.0;
    for (dim_t e = 0; e < a.get_n_elements(); e++) d += (a[e] - b[e]) * (a[e] - b[e]);
    worst = std::max(worst, std::sqrt(d) / std::max(b.norm(), 1e-300));
  }
  cout << "worst ||T_cals - T_als|| / ||T_als|| = " << worst << endl;
  bad += check(worst < 1e-10, "CALS == ALS");
  cout << "ALS time: " << als_timer.get_time() << "  CALS time: " << cals_timer.get_time() << endl;

  // ------------------------------------------------------------------ part 2: the MEX glue's pattern
  vector<double> foreign(X.get_n_elements());
  for (dim_t e = 0; e < X.get_n_elements(); e++) foreign[e] = X[e];
  auto tensor = cals::Tensor(modes, foreign.data());  // view of memory the caller owns
  bad += check(tensor.is_view() && tensor.get_data() == foreign.data(), "Tensor view");

  const dim_t rank = 3;
  cals::Ktensor given(static_cast<int>(rank), modes);
  {
    std::mt19937 gen(5);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    vector<double> lam(rank, 1.0);
    given.set_lambda(lam.data());
    for (auto n = 0lu; n < modes.size(); ++n) {
      vector<double> buf(modes[n] * rank);
      for (auto &v : buf) v = dist(gen);
      given.set_factor(static_cast<int>(n), buf.data());
    }
  }
  vector<Ktensor> init_ktensors(1);
  init_ktensors[0] = Ktensor(given);
  auto fitted_ktensors(init_ktensors);
  for (auto &ktensor : fitted_ktensors) {
    ktensor.denormalize();
    ktensor.normalize();
  }
  vector<vector<Ktensor>> cals_jk_input(1);
  cals::utils::generate_jk_ktensors(fitted_ktensors[0], cals_jk_input[0]);
  bad += check(cals_jk_input[0].size() == modes[0] && cals_jk_input[0][1].is_jk() &&
                   cals_jk_input[0][1].get_jk_fiber() == 1,
               "generate_jk_ktensors");
  for (auto &k : cals_jk_input)
    for (auto &m : k) cals_queue.emplace(m);
  cals_params.buffer_size = rank * modes[0];
  cals_params.force_max_iter = true;
  cals_params.max_iterations = 15;
  auto report = cals::cp_cals(tensor, cals_queue, cals_params);
  bad += check(report.n_ktensors == static_cast<int>(modes[0]), "jk replicas fitted in one call");
  for (auto &k : cals_jk_input)
    for (auto &m : k) {
      bool zero_row = true;
      for (dim_t c = 0; c < rank; c++) zero_row = zero_row && m.get_factor(0)(m.get_jk_fiber(), c) == 0.0;
      bad += check(zero_row, "jk fiber row is exactly zero after the fit");
      m.set_jk_fiber(0.0);
      m.denormalize();
      m.normalize();
      m.set_jk_fiber(NAN);
    }
  {
    // column matching of every replica against the overall model, spelled with the calls the MEX glue uses:
    // cblas_dgemm on factor data, the assignment solver on the column-major buffer, single-column Matrix views
    Ktensor &overall = fitted_ktensors[0];
    const dim_t nc = overall.get_components();
    auto overlap = [&](cals::Matrix &out, const cals::Matrix &ov, const cals::Matrix &rep) {
      cblas_dgemm(CblasColMajor, CblasTrans, CblasNoTrans, nc, nc, ov.get_rows(), 1.0, ov.get_data(), ov.get_col_stride(),
                  rep.get_data(), rep.get_col_stride(), 0.0, out.get_data(), out.get_col_stride());
    };
    for (dim_t slice = 0; slice < tensor.get_modes()[0]; slice++) {
      Ktensor &replica = cals_jk_input[0][slice];
      cals::Matrix score(nc, nc), part(nc, nc);
      overlap(score, overall.get_factor(1), replica.get_factor(1));
      overlap(part, overall.get_factor(2), replica.get_factor(2));
      for (dim_t e = 0; e < score.get_n_elements(); e++) score[e] += part[e];
      vector<int64_t> rows(nc), match(nc);
      bad += check(solve_rectangular_linear_sum_assignment(nc, nc, score.get_data(), true, rows.data(), match.data()) == 0,
                   "assignment solver");
      for (dim_t n = 0; n < overall.get_n_modes(); n++) {
        cals::Matrix &f = replica.get_factor(n);
        cals::Matrix old(f.get_rows(), f.get_cols());
        old.copy(f);
        for (dim_t c = 0; c < nc; c++)
          if ((dim_t)match[c] != c)
            cals::Matrix(f.get_rows(), 1, f.get_data() + c * f.get_col_stride())
                .copy(cals::Matrix(old.get_rows(), 1, old.get_data() + (dim_t)match[c] * old.get_col_stride()));
      }
    }
    auto wide = cals::utils::concatenate_ktensors(cals_jk_input[0]);
    bad += check(wide.get_components() == rank * modes[0], "concatenate_ktensors");
  }

  // ------------------------------------------------------------------ part 3: the experiments' CSV
  bad += check(cals_report.cols.size() == cals_report.iter && cals_report.flops_per_iteration.size() == cals_report.iter,
               "per-iteration vectors have one entry per outer iteration");
  bad += check(cals_report.als_times.get_rows() == cals::AlsTimers::LENGTH &&
                   cals_report.mode_times.get_rows() == cals::ModeTimers::LENGTH * modes.size() &&
                   cals_report.mttkrp_times.get_rows() == cals::MttkrpTimers::LENGTH * modes.size(),
               "timer matrix shapes (src/cals.cpp:54-58)");
  double it_sum = 0.0, mt_sum = 0.0;
  for (dim_t it = 0; it < cals_report.iter; it++) {
    it_sum += cals_report.als_times(cals::AlsTimers::ITERATION, it);
    for (dim_t n = 0; n < modes.size(); n++)
      mt_sum += cals_report.mode_times(n * cals::ModeTimers::LENGTH + cals::ModeTimers::MTTKRP, it);
    bad += check(cals_report.cols[it] >= 1 && cals_report.flops_per_iteration[it] > 0, "cols / flops filled");
  }
  bad += check(it_sum > 0.0 && mt_sum > 0.0 && mt_sum < it_sum * 1.5 && it_sum <= cals_report.total_time * 1.01,
               "timer matrices are filled and consistent with total_time");
  if (argc > 3) {
    cals_report.output_file_name = argv[3];
    cals_report.print_header(cals_report.output_file_name);
    cals_report.print_to_file(cals_report.output_file_name);
  }
  // ------------------------------------------------------------------ part 4: the MTTKRP micro-benchmark
  if (modes.size() == 3) {
    const dim_t rank = 5;
    auto ktensor = Ktensor(rank, modes);
    ktensor.randomize();
    auto before = ktensor;  // the factors the MTTKRP reads
    vector<cals::Matrix> workspace;
    workspace.emplace_back(cals::Matrix(modes[0] * modes[1], rank));
    auto params = cals::mttkrp::MttkrpParams();
    params.method = cals::mttkrp::MTTKRP_METHOD::MTTKRP;
    for (dim_t mode = 0; mode < 3; mode++) {
      auto &G = cals::mttkrp::mttkrp(X, ktensor, workspace, mode, params);
      bad += check(&G == &ktensor.get_factor(mode), "mttkrp returns the mode's factor");
      // ground truth by plain loops: G[i_mode, c] = sum X[i,j,k] * prod_{n != mode} F_n[i_n, c]
      double worst = 0.0, scale = 0.0;
      const dim_t I = modes[0], J = modes[1], K = modes[2];
      for (dim_t c = 0; c < rank; c++) {
        vector<double> g(modes[mode], 0.0);
        for (dim_t k = 0; k < K; k++)
          for (dim_t j = 0; j < J; j++)
            for (dim_t i = 0; i < I; i++) {
              const dim_t idx[3] = {i, j, k};
              double w = X[i + I * (j + J * k)];
              for (dim_t n = 0; n < 3; n++)
                if (n != mode) w *= before.get_factor(n)(idx[n], c);
              g[idx[mode]] += w;
            }
        for (dim_t m = 0; m < modes[mode]; m++) {
          worst = std::max(worst, std::fabs(g[m] - G(m, c)));
          scale = std::max(scale, std::fabs(g[m]));
        }
      }
      bad += check(worst <= 1e-12 * std::max(1.0, scale), "mttkrp::mttkrp equals the plain-loop MTTKRP");
      bad += check(params.flops == 2ull * X.get_n_elements() * rank, "mttkrp flops");
      bad += check(params.mttkrp_timers.timers[cals::MttkrpTimers::MT_GEMM].get_time() > 0.0 &&
                       params.mttkrp_timers.timers[cals::MttkrpTimers::MT_KRP].get_time() == 0.0,
                   "mttkrp timers");
      ktensor.get_factor(mode).copy(before.get_factor(mode));  // the next mode reads the original factors
    }
    // compute_fast_error on a fitted model equals the engine's own error (fused in the update kernel)
    auto &fitted = cals_input[0];
    vector<cals::Matrix> gramians(3);
    for (dim_t n = 0; n < 3; n++) gramians[n] = cals::Matrix(fitted.get_components(), fitted.get_components());
    cals::ops::update_gramians(fitted, gramians);
    auto probe = fitted;
    auto &G = cals::mttkrp::mttkrp(X, probe, workspace, 2, params);
    cals::Matrix H(fitted.get_components(), fitted.get_components());
    std::fill(H.get_data(), H.get_data() + H.get_n_elements(), 1.0);
    for (dim_t n = 0; n < 3; n++) H.hadamard(gramians[n]);
    const double fe = cals::error::compute_fast_error(X.norm(), fitted.get_lambda(), fitted.get_factor(2), G, H);
    const double slow = cals::error::compute_error(X, fitted);
    bad += check(std::fabs(fe - slow) <= 1e-8 * std::max(1.0, slow), "compute_fast_error equals the reconstruction error");
  }
  cout << (bad ? "ref_style_caller: FAILED" : "ref_style_caller: all checks passed") << endl;
  return bad ? 1 : 0;
}
