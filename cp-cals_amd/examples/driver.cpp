// CLI demo on the C++ layer, with the options of the reference's src/examples/driver.cpp
// (-n threads [ignored: no host BLAS], -c MIN:MAX:COPIES, -t I-J-K) plus -d DEVICE, -p f64|f32 and
// -f FILE (the target tensor from a text file, Tensor(file_name), src/tensor.cpp:35-65):
// fits COPIES models of every rank MIN..MAX to a random tensor with concurrent ALS on the GPU and,
// for comparison, one model at a time (cp_als, same engine), and prints both times.
#include <iostream>
#include <numeric>
#include <string>
#include <vector>

#include "als.h"
#include "cals.h"
#include "crash_trace.h"

using std::cerr;
using std::cout;
using std::endl;

static void split(const std::string &s, std::vector<dim_t> &out, char sep) {
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t nx = s.find(sep, pos);
    if (nx == std::string::npos) nx = s.size();
    out.push_back((dim_t)std::strtol(s.substr(pos, nx - pos).c_str(), nullptr, 10));
    pos = nx + 1;
  }
}

int main(int argc, char **argv) {
  crash_trace::install();  // a crash, also one at process exit, leaves its own backtrace on stderr
  std::string tensor_file;
  std::vector<dim_t> modes = {210, 210, 210};
  int min_c = 1, max_c = 10, copies = 5, device = 0;
  std::vector<int> devices;
  bool f32 = false;
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    if ((arg == "-h") || (arg == "--help")) {
      cout << "Usage: " << argv[0] << " [-n THREADS] [-c MIN:MAX:COPIES] [-t I-J-K] [-f TENSOR_FILE] [-d DEVICE] [-p f64|f32]" << endl;
      return 0;
    } else if ((arg == "-n" || arg == "--nthreads") && i + 1 < argc) {
      ++i;  // host BLAS threads: meaningless on the device path
    } else if ((arg == "-c" || arg == "--components") && i + 1 < argc) {
      std::vector<dim_t> v;
      split(argv[++i], v, ':');
      if (v.size() != 3) {
        cerr << "--components option requires one argument of the form MIN:MAX:COPIES." << endl;
        return 1;
      }
      min_c = (int)v[0];
      max_c = (int)v[1];
      copies = (int)v[2];
    } else if ((arg == "-t" || arg == "--tensor") && i + 1 < argc) {
      std::vector<dim_t> v;
      split(argv[++i], v, '-');
      if (v.size() < 3) {
        cerr << "--tensor option requires one argument of the form DIM0-DIM1-DIM2." << endl;
        return 1;
      }
      modes = v;
    } else if ((arg == "-f" || arg == "--file") && i + 1 < argc) {
      tensor_file = argv[++i];
    } else if ((arg == "-d" || arg == "--device") && i + 1 < argc) {
      device = (int)std::strtol(argv[++i], nullptr, 10);
    } else if (arg == "--devices" && i + 1 < argc) {  // e.g. 0,1,2,3: one engine per listed GPU, shared queue
      std::vector<dim_t> v;
      split(argv[++i], v, ',');
      devices.assign(v.begin(), v.end());
    } else if ((arg == "-p" || arg == "--precision") && i + 1 < argc) {
      const std::string v = argv[++i];
      if (v != "f64" && v != "f32") {
        cerr << "--precision takes f64 or f32." << endl;
        return 1;
      }
      f32 = (v == "f32");
    } else {
      cerr << "Unrecognized argument " << arg << endl;
      return 1;
    }
  }
  cals::Tensor X;
  try {
    if (!tensor_file.empty()) {
      X = cals::Tensor(tensor_file);
      modes = X.get_modes();
      cout << "Tensor read from " << tensor_file << ": " << cals::utils::mode_string(modes) << endl;
    } else {
      X = cals::Tensor(modes);
      X.randomize();
    }
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return 2;
  }
  std::vector<dim_t> components;
  for (int c = min_c; c <= max_c; c++)
    for (int k = 0; k < copies; k++) components.push_back((dim_t)c);
  std::vector<cals::Ktensor> cals_input;
  for (auto c : components) {
    cals_input.emplace_back(c, modes);
    cals_input.back().randomize();
  }
  auto als_input(cals_input);

  cals::KtensorQueue queue;
  for (auto &k : cals_input) queue.emplace(k);
  cals::CalsParams cp;
  cp.max_iterations = 1000;
  cp.tol = 1e-5;
  cp.device = device;
  cp.devices = devices;
  cp.precision = f32 ? cals::CalsParams::FP32 : cals::CalsParams::FP64;
  cp.buffer_size = std::accumulate(components.cbegin(), components.cend(), (dim_t)0);
  cp.print();
  cals::Timer t_cals, t_als;
  try {
    cout << "Running CALS..." << endl;
    t_cals.start();
    auto rep = cals::cp_cals(X, queue, cp);
    t_cals.stop();
    cout << "Finished CALS: " << rep.n_ktensors << " models, " << rep.iter << " sweeps." << endl;

    cals::AlsParams ap;
    ap.max_iterations = 1000;
    ap.tol = 1e-5;
    ap.device = device;
    ap.print();
    cout << "Running ALS..." << endl;
    t_als.start();
    for (auto &k : als_input) cals::cp_als(X, k, ap);
    t_als.stop();
    cout << "Finished ALS." << endl;
  } catch (const std::exception &e) {
    cerr << e.what() << endl;
    return 2;
  }
  cout << "======================================================================" << endl;
  cout << "ALS time: " << t_als.get_time() << endl;
  cout << "CALS time: " << t_cals.get_time() << endl;
  cout << "Speedup: " << t_als.get_time() / t_cals.get_time() << endl;
  return 0;
}
