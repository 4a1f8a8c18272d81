// include/tensor.h of HPAC/CP-CALS: dense column-major tensor (mode 0 fastest), owning a 64-byte
// aligned buffer or viewing the caller's memory.  Same members, argument meaning and ownership rules
// as the reference's class, so that its front-ends (driver, MEX glue, tests) compile against this one.
//
// The reference's CUDA build keeps a lazily allocated device copy inside the Tensor
// (include/tensor.h:54-60, uploaded once at src/cals.cpp:144-147).  Here the Tensor keeps a
// DeviceMirror instead: an idle engine of libcals_hip.so holding X's padded permuted copies in HBM
// (DESIGN.md section 2), created by the first cp_cals / cp_als call on this Tensor and reused by the
// later ones (cals_hip_rebind) -- the 216 MB upload, the three permutes and ~3 GB of allocations of
// config 3 happen once, not per call.
#ifndef CALS_AMD_TENSOR_H
#define CALS_AMD_TENSOR_H

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <functional>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "cals_blas.h"
#include "definitions.h"

// the reference's headers export these names into the global namespace and its front-ends rely on it
// (matlab/matlab_cp_cals_jk.cpp uses a bare `vector<Ktensor>`)
using std::function;
using std::unique_ptr;
using std::vector;

namespace cals {

constexpr std::align_val_t alignment = static_cast<std::align_val_t>(64);

// Tensor::implicit_unfold(mode): how the mode-n unfolding sits in the column-major buffer without moving
// data (src/tensor.cpp:143-180): n_blocks matrices of rows x cols, `block_offset` elements apart.
struct Unfolding {
  dim_t n_blocks;
  dim_t block_offset;
  dim_t rows;
  dim_t cols;
  dim_t stride;
};

struct DeviceMirror;  // opaque (cals.cpp): the engine that holds this tensor's copies in HBM

class Tensor {
  struct AlignedDelete {
    void operator()(double *p) const noexcept { operator delete[](p, alignment); }
  };
  int rank{0};
  dim_t n_elements{0};
  dim_t max_n_elements{0};
  vector<dim_t> modes{};
  unique_ptr<double, AlignedDelete> data_up{};
  double *data{nullptr};
  mutable std::shared_ptr<DeviceMirror> mirror{};  // never copied with the Tensor (cf. cudata_up)

  static double *allocate(dim_t n) {
    return static_cast<double *>(operator new[](std::max<dim_t>(n, 1) * sizeof(double), alignment));
  }
  void own(dim_t n) {
    data_up.reset(allocate(n));
    data = data_up.get();
  }

 public:
  Tensor() = default;
  ~Tensor() = default;

  // un-initialised tensor of the given mode sizes
  explicit Tensor(const vector<dim_t> &modes_) : modes(modes_) {
    n_elements = 1;
    for (auto m : modes) n_elements *= m;
    max_n_elements = n_elements;
    own(n_elements);
  }
  // view of memory the caller owns (MATLAB's tensor data, matlab/matlab.cpp:91-117)
  explicit Tensor(const vector<dim_t> &modes_, double *view_data) : modes(modes_), data(view_data) {
    n_elements = 1;
    for (auto m : modes) n_elements *= m;
    max_n_elements = n_elements;
  }
  // text file: first line = the mode sizes separated by blanks, then one value per line, mode 0 fastest
  // (src/tensor.cpp:35-65); throws std::runtime_error if the file is missing or too short
  explicit Tensor(const std::string &file_name);
  // matrices: mode0 x mode1, owning unless view_data is given
  Tensor(dim_t mode0, dim_t mode1, double *view_data = nullptr)
      : n_elements(mode0 * mode1), max_n_elements(mode0 * mode1), modes{mode0, mode1} {
    if (view_data)
      data = view_data;
    else
      own(n_elements);
  }
  // random tensor of CP rank `rank`: a randomised Ktensor turned into a full tensor (src/tensor.cpp:81-87)
  Tensor(dim_t rank, const vector<dim_t> &modes);

  Tensor(Tensor &&rhs) = default;
  Tensor &operator=(Tensor &&rhs) = default;
  // copies of views stay views of the same memory (src/tensor.cpp:89-101); the device mirror is not copied
  Tensor(const Tensor &rhs)
      : rank(rhs.rank), n_elements(rhs.n_elements), max_n_elements(rhs.n_elements), modes(rhs.modes) {
    if (rhs.is_view())
      data = rhs.data;
    else {
      own(n_elements);
      std::copy(rhs.data, rhs.data + n_elements, data);
    }
  }
  Tensor &operator=(const Tensor &rhs) {
    if (this != &rhs) {
      Tensor tmp(rhs);
      *this = std::move(tmp);
    }
    return *this;
  }

  [[nodiscard]] dim_t get_n_elements() const noexcept { return n_elements; }
  [[nodiscard]] dim_t get_max_n_elements() const noexcept { return max_n_elements; }
  [[nodiscard]] dim_t get_n_modes() const noexcept { return static_cast<dim_t>(modes.size()); }
  [[nodiscard]] const vector<dim_t> &get_modes() const noexcept { return modes; }
  [[nodiscard]] double *get_data() const noexcept { return data; }
  // writable access through a non-const Tensor: whatever the caller does with the pointer, the device copy can
  // no longer be trusted (the reference's default path always reads the current host data)
  [[nodiscard]] double *get_data() noexcept {
    mirror.reset();
    return data;
  }
  [[nodiscard]] int get_rank() const noexcept { return rank; }
  void set_rank(int r) noexcept { rank = r; }

  // point somewhere else / back at the owned buffer ("view" mechanics of Matrix::attach / detach)
  void set_data(double *new_data) noexcept {
    data = new_data;
    mirror.reset();
  }
  Tensor &reset_data() noexcept {
    data = data_up.get();
    mirror.reset();
    return *this;
  }
  [[nodiscard]] bool is_view() const noexcept { return data_up == nullptr; }

  double const &operator[](dim_t index) const noexcept { return data[index]; }
  double &operator[](dim_t index) noexcept {
    mirror.reset();  // see get_data()
    return data[index];
  }

  // "soft" resize inside the memory the tensor was created with
  void resize(dim_t new_n_elements, vector<dim_t> &new_modes) {
    assert(new_n_elements <= max_n_elements);
    assert(new_modes.size() == modes.size());
    n_elements = new_n_elements;
    modes = std::move(new_modes);
    mirror.reset();
  }

  [[nodiscard]] double norm() const { return cblas_dnrm2((ptrdiff_t)n_elements, data, 1); }

  Tensor &fill(const function<double()> &&f) {
    for (dim_t i = 0; i < n_elements; i++) data[i] = f();
    mirror.reset();
    return *this;
  }
  Tensor &zero() {
    std::fill(data, data + n_elements, 0.0);
    mirror.reset();
    return *this;
  }
  Tensor &randomize();  // uniform in [-1, 1), seeded from std::random_device (src/tensor.cpp:122-130)
  void copy(const Tensor &ten) noexcept {
    std::copy(ten.get_data(), ten.get_data() + ten.get_n_elements(), data);
    mirror.reset();
  }

  // index / value of the largest element among those the mask allows (NNLS bookkeeping in the reference)
  dim_t max_id(vector<bool> &mask) noexcept {
    dim_t best = 0;
    double top = -DBL_MAX;
    for (dim_t i = 0; i < mask.size() && i < n_elements; i++)
      if (mask[i] && data[i] > top) {
        top = data[i];
        best = i;
      }
    return best;
  }
  double max(vector<bool> &mask) noexcept { return data[max_id(mask)]; }
  double min() noexcept { return *std::min_element(data, data + n_elements); }

  void print(const std::string &&text = "Tensor") const;

  [[nodiscard]] Unfolding implicit_unfold(dim_t mode) const;

  // ---- device mirror (counterpart of get_cudata / allocate_cudata / send_to_device) ----
  // The mirror is dropped by every member that rewrites the data, re-points it or hands out writable access
  // (non-const get_data() / operator[]); a caller that writes through a pointer obtained EARLIER, or through
  // the const overload's pointer, says so with invalidate_device_mirror() (cp_cals also compares a sampled
  // fingerprint of the data and re-uploads when it changed).
  [[nodiscard]] std::shared_ptr<DeviceMirror> &device_mirror() const noexcept { return mirror; }
  void invalidate_device_mirror() const noexcept { mirror.reset(); }
};

}  // namespace cals
#endif
