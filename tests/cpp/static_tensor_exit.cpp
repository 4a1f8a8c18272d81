// A Tensor with STATIC storage keeps its device mirror (an idle engine) until static destruction, i.e.
// until after the HIP runtime's own exit handlers have run.  The process must still exit cleanly.
#include <iostream>

#include "cals.h"
#include "../../cp-cals_amd/examples/crash_trace.h"

static cals::Tensor X(std::vector<dim_t>{16, 12, 10});
static std::vector<cals::Ktensor> models;

int main() {
  crash_trace::install();
  X.randomize();
  for (dim_t r = 1; r <= 4; r++) {
    models.emplace_back(r, X.get_modes());
    models.back().randomize();
  }
  cals::KtensorQueue q;
  for (auto &m : models) q.emplace(m);
  cals::CalsParams p;
  p.buffer_size = 10;
  p.max_iterations = 20;
  auto rep = cals::cp_cals(X, q, p);
  std::cout << "fitted " << rep.n_ktensors << " models; mirror alive: " << (X.device_mirror() ? 1 : 0) << std::endl;
  return rep.n_ktensors == 4 && X.device_mirror() ? 0 : 1;
}
