// TEST DOUBLE, not a backend.  The engine's host side (cp-cals_amd/csrc/cals_hip_engine.cpp: queue,
// first-fit allocator, registry, eviction, compress, admission staging, pinned arena, asynchronous
// copy-out into the callers' storage, rebind, sweep log) is ~2000 lines of pointer arithmetic that the
// GPU box cannot run under a sanitizer (no GPU AddressSanitizer on the pool).  This file lets that code run
// on the CPU under -fsanitize=address,undefined WITHOUT a GPU and WITHOUT libamdhip64:
//   * the two dozen HIP runtime calls the engine makes become malloc / memcpy on host memory
//     (synchronous "streams"), so every size and offset the engine computes is bounds-checked;
//   * the kernel launchers become host functions.  The ones that only MOVE data (column gather /
//     scatter, slot set-up, status packing, stale-column list) do what the kernels do; the numeric
//     kernels (MTTKRP, TTM, update, line search ...) do NOTHING except the bookkeeping the loop needs
//     (iteration counters, a deterministic pseudo-convergence rule, a pseudo line-search flag).
// Consequence used by the test: a model's factors are touched by data movement only, so what comes back
// at eviction must be bit-identical to what was admitted -- through any sequence of admissions,
// evictions and compress moves.  Nothing here is linked into libcals_hip.so or libcals.so.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

#include "../../cp-cals_amd/csrc/cals_hip_internal.h"
#include "fake_device.h"

// ------------------------------------------------------------------------------- the "device queue"
// Two schedules of the same program.  IMMEDIATE: every stream operation completes inside the call that issued it
// (a device that is always ahead of the host).  DEFERRED (fake_set_deferred(true)): stream operations -- kernels,
// asynchronous copies, memsets -- are queued and run only when the host WAITS for the device (stream / event /
// device synchronisation, hipFree / hipHostFree, the synchronous hipMemcpy / hipMemset): a device that is as late
// as the API allows.  Between the two every ordering the real runtime can produce for a single in-order stream is
// bracketed; what the engine gets wrong under the late one (a pinned staging buffer rewritten or released before
// its copy ran, a host read of a result that was never waited for) shows as a wrong value in the models that
// come back, or as a sanitizer report.  Pageable host memory follows the runtime's rules: an asynchronous copy
// FROM it stages the bytes at the call (snapshot), one TO it completes before the call returns.
namespace {
std::recursive_mutex g_mu;  // CalsParams::devices drives two engines from two host threads
std::deque<std::function<void()>> g_queue;
bool g_deferred = false;
std::map<const char *, size_t> g_pinned;  // hipHostMalloc blocks
bool is_pinned(const void *p) {
  std::lock_guard<std::recursive_mutex> lock(g_mu);
  const char *c = static_cast<const char *>(p);
  auto it = g_pinned.upper_bound(c);
  if (it == g_pinned.begin()) return false;
  --it;
  return c < it->first + it->second;
}
void drain() {
  std::lock_guard<std::recursive_mutex> lock(g_mu);
  while (!g_queue.empty()) {
    auto f = std::move(g_queue.front());
    g_queue.pop_front();
    f();
  }
}
void enqueue(std::function<void()> f) {
  std::lock_guard<std::recursive_mutex> lock(g_mu);
  if (g_deferred)
    g_queue.push_back(std::move(f));
  else
    f();
}
// eviction schedule of a replayed pattern: the model admitted k-th leaves at iteration g_schedule[k]
size_t g_malloc_limit = 0;  // hipMalloc refuses larger requests (0: no limit)
std::vector<long long> g_schedule;
size_t g_admitted = 0;
std::map<int, long long> g_slot_target;
}  // namespace

void fake_set_deferred(bool on) {
  drain();
  g_deferred = on;
}
void fake_set_schedule(const std::vector<long long> &iters_at_eviction) {
  std::lock_guard<std::recursive_mutex> lock(g_mu);
  g_schedule = iters_at_eviction;
  g_admitted = 0;
  g_slot_target.clear();
}
void fake_set_malloc_limit(size_t bytes) { g_malloc_limit = bytes; }
size_t fake_queue_depth() {
  std::lock_guard<std::recursive_mutex> lock(g_mu);
  return g_queue.size();
}

// ------------------------------------------------------------------------------- HIP runtime subset
extern "C" {
hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  std::memset(p, 0, sizeof(*p));
  std::strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *f, size_t *t) {
  *f = *t = (size_t)288 << 30;
  return hipSuccess;
}
hipError_t hipMalloc(void **p, size_t n) {
  if (g_malloc_limit && n > g_malloc_limit) {  // fake_set_malloc_limit: "the device is short of memory"
    *p = nullptr;
    return hipErrorOutOfMemory;
  }
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  drain();  // the runtime waits for the device before it releases memory
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  const hipError_t e = hipMalloc(p, n);
  if (e == hipSuccess) {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    g_pinned[static_cast<const char *>(*p)] = n ? n : 1;
  }
  return e;
}
hipError_t hipHostFree(void *p) {
  drain();
  {
    std::lock_guard<std::recursive_mutex> lock(g_mu);
    g_pinned.erase(static_cast<const char *>(p));
  }
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind kind, hipStream_t) {
  if (kind == hipMemcpyHostToDevice && !is_pinned(s)) {  // pageable source: staged at the call
    std::vector<unsigned char> staged(static_cast<const unsigned char *>(s), static_cast<const unsigned char *>(s) + n);
    enqueue([d, staged = std::move(staged)] { std::memcpy(d, staged.data(), staged.size()); });
  } else if (kind == hipMemcpyDeviceToHost && !is_pinned(d)) {  // pageable destination: complete at return
    drain();
    std::memmove(d, s, n);
  } else {
    enqueue([d, s, n] { std::memmove(d, s, n); });
  }
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  drain();
  std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) {
  drain();
  std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
  enqueue([d, v, n] { std::memset(d, v, n); });
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) {
  *s = reinterpret_cast<hipStream_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
  drain();
  std::free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) {
  drain();
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
  drain();
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) {
  *e = reinterpret_cast<hipEvent_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  std::free(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) {
  drain();  // (one queue for all streams: waiting for an event = waiting for everything recorded before it)
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 0.001f;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake device"; }
}

// ------------------------------------------------------------------------------- kernel launchers
namespace calship {

int mttkrp_pick_mt(int m_tiles) {
  for (int v : {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 19, 20})
    if (v >= m_tiles) return v;
  return 0;
}
int ttm_max_mt(int dtype) { return dtype == CALS_F32 ? 20 : 10; }
bool ttm_shape_ok(long long S, long long Mp, int dtype) { return dtype == CALS_F32 || (15 * S + 1) * Mp * 8 < (1ll << 32); }

// numeric kernels: nothing to do on the fake device
hipError_t mttkrp3_launch(int, int, const MttkrpArgs &, hipStream_t) { return hipSuccess; }
// the TTM and the pack write their whole output (zeros here): the engine's sizing of T and Pt is bounds-checked
hipError_t ttm_launch(const TtmArgs &a, hipStream_t) {
  enqueue([a] {
    const size_t es = a.dtype == CALS_F32 ? 4 : 8;
    std::memset(a.Tout, 0, (size_t)a.NB * CALS_BN * (size_t)a.S * (size_t)a.Mp * es);
    std::memset(a.partial, 0, (size_t)a.NB * a.T * (size_t)a.ldPart * CALS_BN * es);
  });
  return hipSuccess;
}
hipError_t pack_pt_launch(const void *, long long, int, int Ap, int NB, int, void *Pt, int dtype, hipStream_t) {
  enqueue([=] { std::memset(Pt, 0, (size_t)NB * (size_t)Ap * CALS_BN * (dtype == CALS_F32 ? 4 : 8)); });
  return hipSuccess;
}
hipError_t contract_launch(const void *, long long, int, int, const void *, long long, void *, long long, int, int,
                           hipStream_t) {
  return hipSuccess;
}
hipError_t krp_launch(const KrpArgs &, hipStream_t) { return hipSuccess; }
hipError_t group_contract_launch(const GroupContractArgs &, hipStream_t) { return hipSuccess; }
hipError_t finish_launch(const FinishArgs &a, hipStream_t);
// the numeric part is a no-op; the end-of-sweep rule the real launch applies for the last mode (UpdateArgs::fin)
// is the fake finish rule below
hipError_t update_launch(const UpdateArgs &a, int, hipStream_t st, int) {
  if (!a.fin.on) return hipSuccess;
  FinishArgs f{};
  f.slots = a.slots;
  f.n_slots = a.n_slots;
  f.mt = a.mt;
  f.max_iter = a.fin.max_iter;
  f.tol = a.fin.tol;
  f.force_max_iter = a.fin.force_max_iter;
  f.evict_enabled = a.fin.evict_enabled;
  return finish_launch(f, st);
}
hipError_t update_huge_factor_launch(const UpdateArgs &, int, hipStream_t) { return hipSuccess; }
hipError_t nnls_launch(const NnlsArgs &, hipStream_t) { return hipSuccess; }
int nnls_rank_class(int r) { return r <= 16 ? 0 : r <= 24 ? 1 : r <= 32 ? 2 : r <= 48 ? 3 : r <= CALS_RMAX ? 4 : 5; }
size_t nnls_huge_block_doubles() { return (size_t)CALS_GLD * CALS_GLD * 9; }
int nnls_huge_chunks(int I, int n_huge) {  // as nnls_kernel.hip: one row per wavefront (4 per workgroup) within 16 GiB
  const size_t by_budget = ((size_t)16 << 30) / (nnls_huge_block_doubles() * sizeof(double) * (size_t)(n_huge > 0 ? n_huge : 1));
  const int want = (I + 3) / 4;
  const int cap = (int)(by_budget < 1 ? 1 : by_budget > 1024 ? 1024 : by_budget);
  return want < 1 ? 1 : want < cap ? want : cap;
}
hipError_t nnls_reset_launch(const int *, int, const NnlsResetArgs &, hipStream_t) { return hipSuccess; }
hipError_t reduce_partials_launch(const void *, int, int, int, int, void *, int, hipStream_t) { return hipSuccess; }
hipError_t reduce_partials_scatter_launch(const void *, int, int, int, int, void *, const int *, int, hipStream_t) {
  return hipSuccess;
}
hipError_t gram_init_launch(const GramInitArgs &, hipStream_t) { return hipSuccess; }
hipError_t ls_snapshot_launch(const LsArgs &, hipStream_t) { return hipSuccess; }
// error-checking line search: the interval counter as ls_ec_prepare_kernel keeps it (the engine mirrors it on the
// host to decide whether to launch the MTTKRP + decide pair at all); decide "accepts" a deterministic subset
hipError_t ls_ec_prepare_launch(const LsArgs &a, hipStream_t) {
  enqueue([a] {
    for (int k = 0; k < a.n_slots; k++) {
      const int slot = a.slots[k];
      const int it = a.mt.ls_iter[slot] + 1;
      a.mt.ls_iter[slot] = (it == a.interval) ? 0 : it;
      a.mt.flags[slot] = (it == a.interval) ? 1 : 0;
    }
  });
  return hipSuccess;
}
hipError_t ls_ec_decide_launch(const LsArgs &a, hipStream_t) {
  enqueue([a] {
    for (int k = 0; k < a.n_slots; k++) {
      const int slot = a.slots[k];
      if (!(a.mt.flags[slot] & 1)) continue;
      const bool accept = ((slot * 5 + a.mt.iters[slot]) % 3) == 0;
      if (!accept)
        a.mt.flags[slot] = 3;
      else if (a.changed)
        *a.changed |= 1;
    }
  });
  return hipSuccess;
}
hipError_t permute_pad_launch(const void *, int, int, const int *, int, int, int, int, void *, int, long long,
                              hipStream_t) {
  return hipSuccess;
}
hipError_t move_columns_launch(void *, int, long long, long long, long long, long long, hipStream_t) { return hipSuccess; }

hipError_t slice_sumsq_launch(const void *, int, long long I, long long, double *, int, double *ss_out, hipStream_t) {
  enqueue([I, ss_out] {
    for (long long i = 0; i < I; i++) ss_out[i] = 1.0;  // ||X||^2 = I
  });
  return hipSuccess;
}

hipError_t init_slots_launch(const int *desc, int n, const ModelTable &mt, hipStream_t) {
  enqueue([desc, n, mt] {
    for (int k = 0; k < n; k++) {
      const int slot = desc[5 * k];
      mt.col[slot] = desc[5 * k + 1];
      mt.rank[slot] = desc[5 * k + 2];
      mt.jk_mode[slot] = desc[5 * k + 3];
      mt.jk_fiber[slot] = desc[5 * k + 4];
      mt.iters[slot] = 1;
      mt.err[slot] = mt.fit[slot] = mt.old_fit[slot] = 0.0;
      mt.potrf_info[slot] = mt.ls_iter[slot] = mt.ls_updated_last[slot] = mt.flags[slot] = 0;
      mt.ls_margin[slot] = 1e300;
      if (g_admitted < g_schedule.size()) g_slot_target[slot] = g_schedule[g_admitted];
      g_admitted++;
    }
  });
  return hipSuccess;
}

// pseudo line search: a deterministic subset of the models "changes" every sweep (flags bit 0) so that
// the engine's pending-T / stale-column logic runs
hipError_t ls_launch(const LsArgs &a, hipStream_t) {
  enqueue([a] {
    for (int k = 0; k < a.n_slots; k++) {
      const int slot = a.slots[k];
      const bool hit = ((slot * 3 + a.mt.iters[slot]) % 5) == 0 && a.mt.iters[slot] < a.max_iter;
      a.mt.flags[slot] = hit ? 1 : 0;
      if (hit) a.mt.ls_margin[slot] = 0.25;  // recognisable value for the host read-back
      if (hit && a.changed) *a.changed += a.mt.rank[slot];
    }
  });
  return hipSuccess;
}

// cals.cpp:336-354 with a deterministic stand-in for "fit_diff < tol": model in slot s "converges" at
// iteration 3 + (7 s + col) mod 17 -- or, while a replayed pattern's schedule is set (fake_set_schedule), at the
// iteration at which the real run evicted the model admitted in the same position
hipError_t finish_launch(const FinishArgs &a, hipStream_t) {
  enqueue([a] {
    for (int k = 0; k < a.n_slots; k++) {
      const int slot = a.slots[k];
      const long long it = a.mt.iters[slot];
      a.mt.err[slot] = 1000.0 * slot + (double)it;  // recognisable values for the host read-back
      a.mt.old_fit[slot] = a.mt.fit[slot];
      a.mt.fit[slot] = 1.0 / (1.0 + (double)it);
      bool converged = it >= 3 + (7 * slot + a.mt.col[slot]) % 17;
      if (!g_schedule.empty()) {
        auto t = g_slot_target.find(slot);
        converged = t != g_slot_target.end() && it >= t->second;
      }
      const bool evict = a.force_max_iter ? it >= a.max_iter : (converged || it >= a.max_iter);
      if (a.evict_enabled && evict)
        a.mt.flags[slot] |= 4;
      else
        a.mt.iters[slot] = it + 1;
    }
  });
  return hipSuccess;
}

hipError_t pack_status_launch(const int *slots, int n, const ModelTable &mt, const int *changed,
                              const int *nnls_status, StatusRec *out, hipStream_t) {
  enqueue([slots, n, mt, changed, nnls_status, out] {
    StatusRec h{};
    h.flags = changed ? *changed : 0;
    h.pad = nnls_status ? *nnls_status : 0;
    h.iters = n;
    out[0] = h;
    for (int k = 0; k < n; k++) {
      const int slot = slots[k];
      StatusRec r{};
      r.flags = mt.flags[slot];
      r.pad = slot;
      r.iters = mt.iters[slot];
      r.err = mt.err[slot];
      r.fit = mt.fit[slot];
      r.old_fit = mt.old_fit[slot];
      r.ls_margin = mt.ls_margin[slot];
      out[1 + k] = r;
    }
  });
  return hipSuccess;
}

hipError_t stale_cols_launch(const int *slots, int n, const ModelTable &mt, int *idx, hipStream_t) {
  enqueue([slots, n, mt, idx] {
    int at = 0;
    for (int k = 0; k < n; k++) {
      const int slot = slots[k];
      if (mt.flags[slot] & 3)
        for (int c = 0; c < mt.rank[slot]; c++) idx[at++] = mt.col[slot] + c;
    }
  });
  return hipSuccess;
}

// the two data-moving kernels, as model_kernels.hip defines them: buffer b, column k of the list <->
// scratch + scratch_off[b] + rows * words_per_elem * k
hipError_t gather_columns_launch(const ColMoveArgs &a, hipStream_t) {
  enqueue([a] {
    for (int b = 0; b < a.n_bufs; b++) {
      const size_t words = (size_t)a.buf[b].rows * a.buf[b].words_per_elem;
      for (int k = 0; k < a.n_cols; k++) {
        unsigned *src = static_cast<unsigned *>(a.buf[b].ptr) + words * (size_t)a.src[k];
        std::memcpy(a.scratch + a.scratch_off[b] + words * (size_t)k, src, words * sizeof(unsigned));
        if (b < a.zero_src_bufs) std::memset(src, 0, words * sizeof(unsigned));
      }
    }
  });
  return hipSuccess;
}
hipError_t scatter_columns_launch(const ColMoveArgs &a, hipStream_t) {
  enqueue([a] {
    for (int b = 0; b < a.n_bufs; b++) {
      const size_t words = (size_t)a.buf[b].rows * a.buf[b].words_per_elem;
      for (int k = 0; k < a.n_cols; k++)
        std::memcpy(static_cast<unsigned *>(a.buf[b].ptr) + words * (size_t)a.dst[k],
                    a.scratch + a.scratch_off[b] + words * (size_t)k, words * sizeof(unsigned));
    }
  });
  return hipSuccess;
}
hipError_t set_cols_launch(const int *pairs, int n, int *col, hipStream_t) {
  enqueue([pairs, n, col] {
    for (int k = 0; k < n; k++) col[pairs[2 * k]] = pairs[2 * k + 1];
  });
  return hipSuccess;
}

// CALS_HIP_VERIFY checks as model_kernels.hip defines them (the numeric kernels do nothing here, so kept operand and
// recomputation agree trivially; the free-column check is real: eviction must zero, compress must move)
hipError_t verify_launch(const VerifyArgs &a, hipStream_t) {
  enqueue([a] {
    for (int k = 0; k < a.n_slots; k++) {
      const int slot = a.slots[k];
      if (a.skip_flagged && (a.mt.flags[slot] & 3)) continue;
      const int col = a.mt.col[slot], r = a.mt.rank[slot];
      const size_t es = (a.kind == 3 || a.dtype != CALS_F32) ? 8 : 4;
      const char *A = static_cast<const char *>(a.a), *B = static_cast<const char *>(a.b);
      auto differ = [&](long long at) { return std::memcmp(A + at * es, B + at * es, es) != 0; };
      if (a.kind == 3) {
        for (int e = 0; e < r * r; e++)
          if (differ((e % r) + (long long)CALS_GLD * (col + e / r))) ++*a.count;
      } else if (a.kind == 1) {
        for (long long e = 0; e < a.rows * r; e++) {
          const int gc = col + (int)(e % r);
          if (differ(((long long)(gc >> 7) * a.rows + e / r) * CALS_BN + (gc & (CALS_BN - 1)))) ++*a.count;
        }
      } else {
        for (long long e = 0; e < a.rows * r; e++)
          if (differ(a.rows * col + e)) ++*a.count;
      }
    }
  });
  return hipSuccess;
}
hipError_t verify_zero_launch(const void *buf, long long rows, int dtype, const int *cols, int n, int *count,
                              hipStream_t) {
  enqueue([buf, rows, dtype, cols, n, count] {
    const size_t es = dtype == CALS_F32 ? 4 : 8;
    for (int k = 0; k < n; k++) {
      const unsigned char *p = static_cast<const unsigned char *>(buf) + (size_t)rows * es * (size_t)cols[k];
      for (size_t i = 0; i < (size_t)rows * es; i++)
        if (p[i]) {
          ++*count;
          break;
        }
    }
  });
  return hipSuccess;
}

}  // namespace calship
