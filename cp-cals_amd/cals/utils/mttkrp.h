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

// include/utils/mttkrp.h:100-101.  The tables under data/<BACKEND>/lookup_tables are tuned for MKL / V100
// variants that do not exist here: always returns an empty table (and never warns).
MttkrpLut read_lookup_table(std::vector<dim_t> const &modes, int threads, bool gpu = false, bool suppress_warning = false);
}  // namespace cals::mttkrp
#endif
