// include/timer.h of HPAC/CP-CALS: the wall-clock Timer the front-ends use around cp_cals and the three
// timer families whose names label the columns of CalsReport / AlsReport CSV files.
#ifndef CALS_AMD_TIMER_H
#define CALS_AMD_TIMER_H

#include <chrono>
#include <limits>
#include <string>

namespace cals {
class Timer {
  using clock = std::chrono::steady_clock;
  clock::time_point t0{};
  double seconds{-1.0};  // < 0: never stopped

 public:
  void start() { t0 = clock::now(); }
  void stop() { seconds = std::chrono::duration<double>(clock::now() - t0).count(); }
  void reset() { seconds = 0.0; }
  [[nodiscard]] double get_time() const { return seconds < 0.0 ? 0.0 : seconds; }
};

// include/timer.h:29-52.  On the device engine the slots are filled from hipEvent pairs / the loop's host
// clock (cals_hip_sweep_record): MT_GEMM = the fused MTTKRP kernel, TS_GEMM = the TTM of a dimension-tree
// pair, TS_GEMV = the contraction of T, MT_KRP = the Khatri-Rao kernel of N > 3 modes.
struct MttkrpTimers {
  enum TIMERS { MT_KRP = 0, MT_GEMM, TS_GEMM, TS_GEMV, LENGTH };
  std::string names[LENGTH] = {"MT_KRP", "MT_GEMM", "TS_GEMM", "TS_GEMV"};
  Timer timers[LENGTH];
  Timer &operator[](int t) { return timers[t]; }
};
struct ModeTimers {
  enum TIMERS { MTTKRP = 0, UPDATE, LENGTH };
  std::string names[LENGTH] = {"TOTAL_MTTKRP", "UPDATE"};
  Timer timers[LENGTH];
  Timer &operator[](int t) { return timers[t]; }
};
struct AlsTimers {
  enum TIMERS { ITERATION = 0, DEFRAGMENTATION, ERROR, LINE_SEARCH, G_COPY, LENGTH };
  std::string names[LENGTH] = {"ITERATION", "DEFRAGMENTATION", "ERROR", "LINESEARCH", "G_COPY"};
  Timer timers[LENGTH];
  Timer &operator[](int t) { return timers[t]; }
};
}  // namespace cals
#endif
