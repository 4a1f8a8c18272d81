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
  void set_time(double s) { seconds = s; }  // extension: slots filled from device event pairs
  [[nodiscard]] double get_time() const { return seconds < 0.0 ? 0.0 : seconds; }
};

// The three timer families of include/timer.h:29-52: a Timer per slot, the slot names label the CSV columns
// of CalsReport / AlsReport.  On the device engine the slots are filled from hipEvent pairs / the loop's host
// clock (cals_hip_sweep_record): MT_GEMM = the fused MTTKRP kernel, TS_GEMM = the TTM of a dimension-tree
// pair, TS_GEMV = the contraction of T, MT_KRP = the Khatri-Rao kernel of N > 3 modes.
template <int N>
struct TimerSet {
  std::string names[N];
  Timer timers[N];
  Timer &operator[](int slot) { return timers[slot]; }
};
struct MttkrpTimers : TimerSet<4> {
  enum TIMERS { MT_KRP = 0, MT_GEMM, TS_GEMM, TS_GEMV, LENGTH };
  MttkrpTimers() : TimerSet<4>{{"MT_KRP", "MT_GEMM", "TS_GEMM", "TS_GEMV"}, {}} {}
};
struct ModeTimers : TimerSet<2> {
  enum TIMERS { MTTKRP = 0, UPDATE, LENGTH };
  ModeTimers() : TimerSet<2>{{"TOTAL_MTTKRP", "UPDATE"}, {}} {}
};
struct AlsTimers : TimerSet<5> {
  enum TIMERS { ITERATION = 0, DEFRAGMENTATION, ERROR, LINE_SEARCH, G_COPY, LENGTH };
  AlsTimers() : TimerSet<5>{{"ITERATION", "DEFRAGMENTATION", "ERROR", "LINESEARCH", "G_COPY"}, {}} {}
};
}  // namespace cals
#endif
