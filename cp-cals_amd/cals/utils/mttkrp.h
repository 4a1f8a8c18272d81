// include/utils/mttkrp.h of HPAC/CP-CALS.  The reference chooses per (mode, rank, threads) among three
// CPU/CUDA MTTKRP variants, by lookup table; the device engine has its own plan (cals_hip_tree: fused
// MTTKRP or dimension-tree TTM + contraction, chosen by a cost model), so the method and the table are
// accepted for source compatibility and have no effect.  The parameter / counter structs keep their
// fields because CalsParams, AlsParams and the reports name them.
#ifndef CALS_AMD_UTILS_MTTKRP_H
#define CALS_AMD_UTILS_MTTKRP_H

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "ktensor.h"
#include "timer.h"

namespace cals::mttkrp {
typedef std::vector<std::map<int, int>> LUT_v;

struct MttkrpLut {
  LUT_v lut_v{};
  std::vector<int> keys_v{};
};

enum MTTKRP_METHOD { MTTKRP = 0, TWOSTEP0, TWOSTEP1, AUTO, LENGTH };
static const std::string mttkrp_method_names[MTTKRP_METHOD::LENGTH] = {"MTTKRP", "TWOSTEP0", "TWOSTEP1", "AUTO"};

struct KrpParams {
  uint64_t flops{0};
  uint64_t memops{0};
  bool cuda{false};
};

struct MttkrpParams {
  MTTKRP_METHOD method{AUTO};
  KrpParams krp_params{};
  MttkrpLut lut{};
  bool cuda{false};
  MttkrpTimers mttkrp_timers;
  uint64_t flops{0};
  uint64_t memops{0};
};

// mttkrp::mttkrp (include/utils/mttkrp.h:77-81, src/utils/mttkrp.cpp:562-614): the MTTKRP of `mode` from the other
// factors of u, written into u's factor `mode` (returned) -- here ONE launch of the fused MTTKRP kernel + the split
// reduction on an engine leased from X's device mirror (cals_hip_mttkrp).  `workspace` is accepted and unused (no
// Khatri-Rao product is materialised), `params.method` / `lut` have no effect; params.flops / memops are the
// algorithmic counts (2 * prod(I) * R flop; |X| + sum_n I_n R elements) and the timers carry the device time of the
// launch in MT_GEMM (others 0).  This is the callable include/experiments/bench_mttkrp_cals.h:49-84 times.
cals::Matrix &mttkrp(const cals::Tensor &X, cals::Ktensor &u, std::vector<cals::Matrix> &workspace, dim_t mode,
                     cals::mttkrp::MttkrpParams &params);

// include/utils/mttkrp.h:100-101.  The tables under data/<BACKEND>/lookup_tables are tuned for MKL / V100
// variants that do not exist here: always returns an empty table (and never warns).
MttkrpLut read_lookup_table(std::vector<dim_t> const &modes, int threads, bool gpu = false, bool suppress_warning = false);
}  // namespace cals::mttkrp
#endif
