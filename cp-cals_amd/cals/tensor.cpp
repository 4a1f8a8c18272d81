// Tensor / Matrix members that do not fit in the headers (reference: src/tensor.cpp, src/matrix.cpp),
// and the host BLAS thread control of include/cals_blas.h.
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include "ktensor.h"
#include "matrix.h"
#include "tensor.h"

namespace {
int g_threads = 1;
}
// include/cals_blas.h:184-186 (global, as the reference's driver calls them unqualified).  The reference
// forwards to the BLAS vendor's thread control; the device path has no host BLAS, so the value is only
// remembered (CalsReport::n_threads reports it).
void set_threads(int threads) { g_threads = threads > 0 ? threads : 1; }
int get_threads() { return g_threads; }

namespace cals {

Tensor::Tensor(const std::string &file_name) {
  std::ifstream file(file_name);
  if (!file.is_open()) throw std::runtime_error("Tensor: cannot open " + file_name);
  std::string header;
  std::getline(file, header);
  std::stringstream dims(header);
  for (dim_t m; dims >> m;) modes.push_back(m);
  if (modes.empty()) throw std::runtime_error("Tensor: " + file_name + " has no mode sizes on its first line");
  n_elements = 1;
  for (auto m : modes) n_elements *= m;
  max_n_elements = n_elements;
  own(n_elements);
  dim_t got = 0;
  for (double v; got < n_elements && file >> v;) data[got++] = v;
  if (got != n_elements)
    throw std::runtime_error("Tensor: " + file_name + " holds " + std::to_string(got) + " values, " +
                             std::to_string(n_elements) + " expected");
}

Tensor::Tensor(dim_t rank_, const vector<dim_t> &modes_) {
  Ktensor P(rank_, modes_);
  P.randomize();
  *this = P.to_tensor();
  rank = static_cast<int>(rank_);
}

Tensor &Tensor::randomize() {
  std::uniform_real_distribution<double> dist(-1.0, 1.0);
  std::random_device seed;
  std::mt19937 gen(seed());
  for (dim_t i = 0; i < n_elements; i++) data[i] = dist(gen);
  mirror.reset();
  return *this;
}

Unfolding Tensor::implicit_unfold(const dim_t mode) const {
  dim_t before = 1, after = 1;
  for (dim_t n = 0; n < mode; n++) before *= modes[n];
  for (dim_t n = mode + 1; n < modes.size(); n++) after *= modes[n];
  if (mode == 0) return Unfolding{1, 0, modes[0], after, modes[0]};                 // I x JK..., ld = I
  if (mode + 1 == modes.size()) return Unfolding{1, 0, modes[mode], before, before};  // (IJ..)^T, ld = IJ..
  return Unfolding{after, before * modes[mode], modes[mode], before, before};       // `after` blocks of (before)^T
}

void Tensor::print(const std::string &&text) const {
  using std::cout;
  using std::endl;
  cout << "----------------------------------------" << endl << text << endl << "Modes: ";
  for (auto m : modes) cout << m << " ";
  cout << endl << "data = [ ";
  for (dim_t i = 0; i < n_elements; i++) cout << std::setw(6) << data[i] << "  ";
  cout << "]" << endl << "----------------------------------------" << endl;
}

void Matrix::print(const std::string &&text) const {
  using std::cout;
  using std::endl;
  cout << "----------------------------------------" << endl << text << endl;
  cout << "----------------------------------------" << endl;
  cout << "Rows: " << get_rows() << ", Cols: " << get_cols() << endl;
  const auto old = cout.precision(4);
  for (dim_t r = 0; r < get_rows(); r++) {
    for (dim_t c = 0; c < get_cols(); c++) cout << "  " << std::setw(8) << (*this)(r, c) << "  ";
    cout << endl;
  }
  cout.precision(old);
  cout << "----------------------------------------" << endl;
}

void Matrix::info() const {
  std::cout << "nRows: " << get_rows() << ", nCols: " << get_cols() << ", nElements: " << get_n_elements()
            << ", maxNElements: " << get_max_n_elements() << std::endl;
}

}  // namespace cals
