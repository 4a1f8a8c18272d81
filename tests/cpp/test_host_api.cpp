// Host side of the C++ layer, no GPU needed: the value classes of the reference's header set
// (Tensor / Matrix / Ktensor / MultiKtensor), the text-file reader, the jackknife helpers, the report
// writers -- checked against hand-worked expectations of the reference's semantics (file:line in the
// comments).  The -m "not gpu" suite runs this binary twice: plain, and built with
// -fsanitize=address,undefined (tests/test_host_api_and_sanitizers.py).
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <numeric>

#include "als.h"
#include "cals.h"
#include "../../include/cals_hip.h"

using namespace cals;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
      failures++;                                                          \
    }                                                                      \
  } while (0)

static double counter_value = 0.0;
static double next_value() {  // deterministic fill: 0.5, -1.0, 1.5, -2.0 ...
  counter_value += 0.5;
  return (static_cast<long>(counter_value * 2) % 2 ? counter_value : -counter_value);
}

static void test_tensor(const std::string &tmp) {
  // Tensor(file): first line = mode sizes, then one value per line, mode 0 fastest (src/tensor.cpp:35-65)
  const std::string path = tmp + "/tensor.txt";
  {
    std::ofstream f(path);
    f << "3 2 2\n";
    for (int i = 0; i < 12; i++) f << 0.25 * i << "\n";
  }
  Tensor T(path);
  CHECK(T.get_n_modes() == 3 && T.get_modes()[0] == 3 && T.get_modes()[2] == 2 && T.get_n_elements() == 12);
  CHECK(T[0] == 0.0 && T[5] == 1.25 && T[11] == 2.75 && !T.is_view());
  bool threw = false;
  try {
    Tensor missing(tmp + "/does_not_exist.txt");
  } catch (const std::runtime_error &e) {
    threw = std::string(e.what()).find("cannot open") != std::string::npos;
  }
  CHECK(threw);
  {
    std::ofstream f(tmp + "/short.txt");
    f << "2 2 2\n1\n2\n3\n";
  }
  threw = false;
  try {
    Tensor too_short(tmp + "/short.txt");
  } catch (const std::runtime_error &) {
    threw = true;
  }
  CHECK(threw);

  // views, copies, soft resize (include/tensor.h:82-190)
  std::vector<double> mem(24, 1.0);
  Tensor V({4, 3, 2}, mem.data());
  CHECK(V.is_view() && V.get_data() == mem.data());
  Tensor Vc(V);
  CHECK(Vc.is_view() && Vc.get_data() == mem.data());  // copies of views stay views
  Tensor O(std::vector<dim_t>{4, 3, 2});
  O.fill([]() { return 2.0; });
  Tensor Oc(O);
  CHECK(!Oc.is_view() && Oc.get_data() != O.get_data() && Oc[23] == 2.0);
  CHECK(std::fabs(O.norm() - std::sqrt(24 * 4.0)) < 1e-14);
  O.zero();
  CHECK(O.norm() == 0.0);
  CHECK(reinterpret_cast<uintptr_t>(O.get_data()) % 64 == 0);  // 64-byte aligned (include/tensor.h:27)

  // implicit_unfold (src/tensor.cpp:143-180) for 4 x 3 x 2
  Unfolding u0 = V.implicit_unfold(0), u1 = V.implicit_unfold(1), u2 = V.implicit_unfold(2);
  CHECK(u0.n_blocks == 1 && u0.rows == 4 && u0.cols == 6 && u0.stride == 4 && u0.block_offset == 0);
  CHECK(u1.n_blocks == 2 && u1.rows == 3 && u1.cols == 4 && u1.stride == 4 && u1.block_offset == 12);
  CHECK(u2.n_blocks == 1 && u2.rows == 2 && u2.cols == 12 && u2.stride == 12 && u2.block_offset == 0);

  // Tensor(rank, modes): a randomised rank-`rank` Ktensor made full (src/tensor.cpp:81-87)
  Tensor R(2, {5, 4, 3});
  CHECK(R.get_rank() == 2 && R.get_n_elements() == 60 && R.norm() > 0.0);
  std::vector<bool> mask(60, false);
  mask[7] = mask[9] = true;
  CHECK(R.max_id(mask) == (R[7] >= R[9] ? 7u : 9u) && R.min() <= R[0]);

  // Matrix (include/matrix.h)
  Matrix A(3, 2), B(3, 2);
  int k = 0;
  A.fill([&]() { return double(++k); });  // column-major 1..6
  B.fill([]() { return 2.0; });
  CHECK(A(2, 0) == 3.0 && A(0, 1) == 4.0 && A.get_col_stride() == 3);
  A.hadamard(B);
  CHECK(A(2, 1) == 12.0);
  CHECK(A.one_norm() == 8.0 + 10.0 + 12.0);
  Matrix At(3, 2);  // declared with the SOURCE's shape; its buffer receives A^T as a 2 x 3 column-major block
  At.transpose_copy(A);  // (src/matrix.cpp:45-52: this[j + i * cols] = rhs[i + j * rows])
  CHECK(At.get_data()[1] == A(0, 1) && At.get_data()[2] == A(1, 0));
  A.resize(3, 1);
  CHECK(A.get_cols() == 1 && A.get_n_elements() == 3 && A.get_max_n_elements() == 6);
  double elsewhere[3] = {7, 8, 9};
  A.attach(elsewhere);
  CHECK(A(1, 0) == 8.0);
  A.detach();
  CHECK(A(1, 0) == 4.0);
}

static void test_ktensor() {
  const std::vector<dim_t> modes = {4, 3, 2};
  Ktensor K(2, modes);
  counter_value = 0.0;
  K.fill(next_value);  // fill order: factor 0..N-1 column-major, then normalize() (src/ktensor.cpp:21-30)
  for (dim_t n = 0; n < 3; n++)
    for (dim_t c = 0; c < 2; c++) {
      double s = 0.0;
      for (dim_t i = 0; i < modes[n]; i++) s += K.get_factor(n)(i, c) * K.get_factor(n)(i, c);
      CHECK(std::fabs(s - 1.0) < 1e-14);
    }
  // lambda = product of the raw column norms: column 0 of factor 0 was (0.5, -1, 1.5, -2)
  const double n00 = std::sqrt(0.25 + 1 + 2.25 + 4);
  CHECK(std::fabs(K.get_factor(0)(0, 0) - 0.5 / n00) < 1e-15);
  Tensor full = K.to_tensor();
  double want = 0.0;  // element (1, 2, 1)
  for (dim_t c = 0; c < 2; c++) want += K.get_lambda()[c] * K.get_factor(0)(1, c) * K.get_factor(1)(2, c) * K.get_factor(2)(1, c);
  CHECK(std::fabs(full[1 + 4 * 2 + 12 * 1] - want) < 1e-15);
  CHECK(std::fabs(error::compute_error(full, K)) < 1e-14);

  // normalize(mode, iteration) (src/ktensor.cpp:66-83): iteration 1 = 2-norm, later = signed max-abs entry
  Ktensor S(K);
  CHECK(S.get_id() != K.get_id() && S.get_iters() == 0);  // a copy is a new model
  S.get_factor(1)(0, 0) = -5.0;
  S.get_factor(1)(1, 0) = 5.0;  // tie in magnitude: the FIRST index wins (cblas_idamax)
  S.get_factor(1)(2, 0) = 1.0;
  S.normalize(1, 2);
  CHECK(S.get_lambda()[0] == -5.0 && S.get_factor(1)(0, 0) == 1.0 && S.get_factor(1)(1, 0) == -1.0);
  S.get_factor(2)(0, 1) = 0.0;
  S.get_factor(2)(1, 1) = 0.0;
  S.normalize(2, 3);
  CHECK(S.get_lambda()[1] == 0.0 && S.get_factor(2)(0, 1) == 0.0);  // zero lambda leaves the column alone
  S.normalize(0, 1);
  double nrm = 0.0;
  for (dim_t i = 0; i < 4; i++) nrm += S.get_factor(0)(i, 1) * S.get_factor(0)(i, 1);
  CHECK(std::fabs(nrm - 1.0) < 1e-14 && S.get_lambda()[1] > 0.0);

  // denormalize folds lambda into factor 0 (src/ktensor.cpp:101-107)
  Ktensor D(K);
  const double before = D.get_factor(0)(2, 1), lam = D.get_lambda()[1];
  D.denormalize();
  CHECK(std::fabs(D.get_factor(0)(2, 1) - before * lam) < 1e-15);

  // jackknife helpers (include/ktensor.h:258-325)
  Ktensor J(2, modes, /*fiber*/ 2, /*mode*/ 0);
  counter_value = 0.0;
  J.fill(next_value);
  CHECK(J.is_jk() && J.get_factor(0)(2, 0) == 0.0 && J.get_factor(0)(2, 1) == 0.0);
  Ktensor Jr = J.to_regular();
  CHECK(!Jr.is_jk() && Jr.get_modes()[0] == 3 && Jr.get_factor(0)(2, 1) == J.get_factor(0)(3, 1) &&
        Jr.get_factor(1)(1, 0) == J.get_factor(1)(1, 0));
  J.set_jk_fiber(NAN);
  CHECK(std::isnan(J.get_factor(0)(2, 0)));
  std::vector<Ktensor> reps;
  utils::generate_jk_ktensors(K, reps);
  CHECK(reps.size() == 4 && reps[3].is_jk() && reps[3].get_jk_fiber() == 3 && reps[3].get_jk_mode() == 0);
  Ktensor wide = utils::concatenate_ktensors(reps);
  CHECK(wide.get_components() == 8 && wide.get_factor(2)(1, 7) == K.get_factor(2)(1, 1) &&
        wide.get_lambda()[6] == K.get_lambda()[0]);
  CHECK(utils::mode_string(modes) == "4-3-2");

  // Ktensor::copy: state + contents, not id / jk (src/ktensor.cpp:163-181)
  Ktensor C(2, modes);
  K.set_iters(9);
  K.set_approximation_error(0.125);
  C.copy(K);
  CHECK(C.get_iters() == 9 && C.get_approximation_error() == 0.125 && C.get_id() != K.get_id() &&
        C.get_factor(1)(2, 1) == K.get_factor(1)(2, 1));
  CHECK(K.get_active_set(1).size() == 3 && K.get_active_set(1)[0].size() == 2 && K.get_active_set(1)[2][1]);

  // jackknife norms (src/utils/utils.cpp:103-152)
  Tensor X(std::vector<dim_t>{3, 2, 2});
  int q = 0;
  X.fill([&]() { return double(++q); });
  auto jkn = utils::calculate_jackknifing_norms(X);
  double all = 0.0, s1 = 0.0;
  for (dim_t e = 0; e < 12; e++) {
    all += X[e] * X[e];
    if (e % 3 == 1) s1 += X[e] * X[e];
  }
  CHECK(jkn.size() == 3 && std::fabs(jkn[1] - std::sqrt(all - s1)) < 1e-12);
}

static void test_multi_ktensor() {
  std::vector<dim_t> modes = {5, 4, 3};
  MultiKtensor mkt(modes, 8);
  CHECK(mkt.get_factor(0).get_cols() == 0 && mkt.get_leftmost_id() == 0);
  Ktensor a(2, modes), b(3, modes), c(2, modes), d(2, modes), e3(3, modes);
  counter_value = 0.0;
  for (Ktensor *k : {&a, &b, &c, &d, &e3}) k->fill(next_value);
  const Ktensor a0(a), b0(b), c0(c), e0(e3);
  auto same = [](const Ktensor &x, const Ktensor &y) {
    for (dim_t n = 0; n < x.get_n_modes(); n++)
      for (dim_t i = 0; i < x.get_factor(n).get_n_elements(); i++)
        if (x.get_factor(n)[i] != y.get_factor(n)[i]) return false;
    return true;
  };
  mkt.add(a).add(b).add(c);  // first fit: columns 0, 2, 5 (src/multi_ktensor.cpp:14-39)
  auto &reg = mkt.get_registry();
  CHECK(reg.at(1).col == 0 && reg.at(2).col == 2 && reg.at(3).col == 5 && mkt.get_factor(1).get_cols() == 7);
  CHECK(a.get_iters() == 1 && same(a, a0) && same(b, b0));
  // the model's factors now LIVE in the multi-factor columns (Ktensor::attach, src/ktensor.cpp:109-125)
  CHECK(b.get_factor(2).get_data() == mkt.get_factor(2).get_data() + 2 * 3);
  // registry Gramians = A^T A (src/multi_ktensor.cpp:88-94)
  double g01 = 0.0;
  for (dim_t i = 0; i < 4; i++) g01 += b.get_factor(1)(i, 0) * b.get_factor(1)(i, 1);
  CHECK(std::fabs(reg.at(2).gramians[1](0, 1) - g01) < 1e-15 && reg.at(2).gramians[1].get_rows() == 3);
  bool full = false;
  try {
    mkt.add(d);  // one free column left
  } catch (const BufferFull &bf) {
    full = std::string(bf.what()).find("Buffer is full") != std::string::npos;
  }
  CHECK(full);
  mkt.remove(2);  // detach: contents back into b's own storage, columns zeroed (src/ktensor.cpp:127-135)
  CHECK(same(b, b0) && b.get_factor(0).get_data() != mkt.get_factor(0).get_data() + 2 * 5);
  CHECK(mkt.get_factor(0).get_data()[2 * 5 + 1] == 0.0 && mkt.get_factor(0).get_cols() == 7);
  mkt.add(e3);  // the freed run of three columns is reused
  CHECK(mkt.get_registry().at(4).col == 2);
  mkt.remove(1);
  CHECK(mkt.get_leftmost_id() == 0);
  mkt.compress();  // e3 moves 2 -> 0 (overlapping move), c moves 5 -> 3 (src/multi_ktensor.cpp:188-264)
  CHECK(mkt.get_registry().at(4).col == 0 && mkt.get_registry().at(3).col == 3 && mkt.get_factor(0).get_cols() == 5);
  CHECK(same(e3, e0) && same(c, c0) && mkt.get_leftmost_id() == 4);
  CHECK(e3.get_factor(1).get_data() == mkt.get_factor(1).get_data());
  mkt.remove(4).remove(3);
  CHECK(mkt.get_registry().empty() && same(e3, e0) && same(c, c0) && mkt.get_factor(2).get_cols() == 1);
  Ktensor jk(2, modes, 1);
  jk.fill(next_value);
  mkt.set_line_search(true);
  mkt.add(jk);
  CHECK(mkt.get_flag_jk() && mkt.get_registry().at(5).ls_params.prev_ktensor.get_components() == 2);
}

static void test_reports_and_failures(const std::string &tmp) {
  CalsReport rep;
  rep.modes = {4, 3, 2};
  rep.n_modes = 3;
  rep.iter = 2;
  rep.flops_per_iteration = {10, 20};
  rep.cols = {5, 4};
  rep.als_times = Matrix(AlsTimers::LENGTH, 2);
  rep.mode_times = Matrix(ModeTimers::LENGTH * 3, 2);
  rep.als_times.zero();
  rep.mode_times.zero();
  rep.als_times(AlsTimers::ITERATION, 1) = 0.5;
  const std::string csv = tmp + "/report.csv";
  rep.print_header(csv);
  rep.print_to_file(csv);
  std::ifstream f(csv);
  std::string header, row1, row2;
  std::getline(f, header);
  std::getline(f, row1);
  std::getline(f, row2);
  CHECK(header.find("TENSOR_MODES;") != std::string::npos && header.find("MODE_2_UPDATE;") != std::string::npos &&
        header.find("LINESEARCH;") != std::string::npos);
  CHECK(row1.find("4-3-2;") != std::string::npos && row2.find(";20;4;5.0") != std::string::npos);
  AlsReport ar;
  ar.modes = {4, 3, 2};
  ar.print_header(tmp + "/als.csv");
  ar.print_to_file(tmp + "/als.csv");

  // no device path without a GPU, and no CPU path at all: loud failures
  Tensor X(std::vector<dim_t>{4, 3, 2});
  X.randomize();
  Ktensor k(2, {4, 3, 2});
  k.randomize();
  KtensorQueue q;
  q.emplace(k);
  CalsParams p;
  p.cuda = false;
  bool threw = false;
  try {
    cp_cals(X, q, p);
  } catch (const std::runtime_error &e) {
    threw = std::string(e.what()).find("no CPU") != std::string::npos;
  }
  CHECK(threw);
  if (cals_hip_device_count() == 0) {
    p.cuda = true;
    threw = false;
    try {
      cp_cals(X, q, p);
    } catch (const std::runtime_error &e) {
      threw = std::string(e.what()).find("no CPU fallback") != std::string::npos;
    }
    CHECK(threw);
  }
  set_threads(3);
  CHECK(get_threads() == 3);
  CHECK(mttkrp::read_lookup_table({4, 3, 2}, 8).keys_v.empty());
  CHECK(update::update_method_names[update::NNLS] == "nnls" && ls::ls_method_names[ls::ERROR_CHECKING_SERIAL] == "error-checking-serial");
}

int main(int argc, char **argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp";
  test_tensor(tmp);
  test_ktensor();
  test_multi_ktensor();
  test_reports_and_failures(tmp);
  std::printf(failures ? "test_host_api: %d FAILED\n" : "test_host_api: all checks passed\n", failures);
  return failures ? 1 : 0;
}
