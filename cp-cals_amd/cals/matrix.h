// include/matrix.h of HPAC/CP-CALS: 2-D column-major Tensor (leading dimension = rows).
#ifndef CALS_AMD_MATRIX_H
#define CALS_AMD_MATRIX_H

#include "tensor.h"

namespace cals {
class Matrix : public Tensor {
  dim_t rows{0};
  dim_t cols{0};
  dim_t col_stride{0};

 public:
  Matrix() = default;
  ~Matrix() = default;
  Matrix(dim_t dim0, dim_t dim1) : Tensor(dim0, dim1), rows(dim0), cols(dim1), col_stride(dim0) {}
  Matrix(dim_t dim0, dim_t dim1, double *view_data) : Tensor(dim0, dim1, view_data), rows(dim0), cols(dim1), col_stride(dim0) {}
  Matrix(Matrix &&rhs) = default;
  Matrix &operator=(Matrix &&rhs) = default;
  Matrix(const Matrix &rhs) = default;
  Matrix &operator=(const Matrix &rhs) = default;

  [[nodiscard]] dim_t get_rows() const noexcept { return rows; }
  [[nodiscard]] dim_t get_cols() const noexcept { return cols; }
  [[nodiscard]] dim_t get_col_stride() const noexcept { return col_stride; }

  double &operator()(dim_t row, dim_t col) { return get_data()[row + col * col_stride]; }
  double operator()(dim_t row, dim_t col) const noexcept { return get_data()[row + col * col_stride]; }

  // "soft" resize within the memory the matrix was created with (MultiKtensor::adjust_edges)
  Matrix &resize(dim_t new_rows, dim_t new_cols) noexcept {
    vector<dim_t> m = {new_rows, new_cols};
    Tensor::resize(new_rows * new_cols, m);
    rows = new_rows;
    cols = new_cols;
    col_stride = new_rows;
    return *this;
  }

  Matrix &hadamard(const Matrix &mat) {  // element-wise product in place (src/matrix.cpp:12-17)
    for (dim_t i = 0; i < get_n_elements(); i++) get_data()[i] *= mat[i];
    return *this;
  }

  void attach(double *data) { set_data(data); }
  void detach() { reset_data(); }

  void print(const std::string &&text = "Matrix") const;
  void info() const;

  [[nodiscard]] double one_norm() const {  // max over columns of the sum of magnitudes
    double best = -DBL_MAX;
    for (dim_t c = 0; c < cols; c++) best = std::max(best, cblas_dasum((ptrdiff_t)rows, get_data() + c * col_stride, 1));
    return best;
  }

  Matrix &transpose_copy(const Matrix &rhs) {  // this (rows x cols) receives rhs as stored transposed
    for (dim_t i = 0; i < rows; i++)
      for (dim_t j = 0; j < cols; j++) get_data()[j + i * cols] = rhs.get_data()[i + j * rows];
    return *this;
  }
};
}  // namespace cals
#endif
