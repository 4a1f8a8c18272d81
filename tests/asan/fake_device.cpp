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
#include <cstdlib>
#include <cstring>

#include "../../cp-cals_amd/csrc/cals_hip_internal.h"

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
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) {
  std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) {
  std::memset(d, v, n);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t *s) {
  *s = reinterpret_cast<hipStream_t>(std::malloc(8));
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
  std::free(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
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
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
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
hipError_t ttm_launch(const TtmArgs &, hipStream_t) { return hipSuccess; }
hipError_t pack_pt_launch(const void *, long long, int, int, int, int, void *, int, hipStream_t) { return hipSuccess; }
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
int nnls_huge_chunks(int I, int) { return I < 16 ? 1 : (I + 15) / 16 > 16 ? 16 : (I + 15) / 16; }
hipError_t nnls_reset_launch(const int *, int, const NnlsResetArgs &, hipStream_t) { return hipSuccess; }
hipError_t reduce_partials_launch(const void *, int, int, int, int, void *, int, hipStream_t) { return hipSuccess; }
hipError_t reduce_partials_scatter_launch(const void *, int, int, int, int, void *, const int *, int, hipStream_t) {
  return hipSuccess;
}
hipError_t gram_init_launch(const GramInitArgs &, hipStream_t) { return hipSuccess; }
hipError_t ls_snapshot_launch(const LsArgs &, hipStream_t) { return hipSuccess; }
hipError_t ls_ec_prepare_launch(const LsArgs &, hipStream_t) { return hipSuccess; }
hipError_t ls_ec_decide_launch(const LsArgs &, hipStream_t) { return hipSuccess; }
hipError_t permute_pad_launch(const void *, int, int, const int *, int, int, int, int, void *, int, long long,
                              hipStream_t) {
  return hipSuccess;
}
hipError_t move_columns_launch(void *, int, long long, long long, long long, long long, hipStream_t) { return hipSuccess; }

hipError_t slice_sumsq_launch(const void *, int, long long I, long long, double *, int, double *ss_out, hipStream_t) {
  for (long long i = 0; i < I; i++) ss_out[i] = 1.0;  // ||X||^2 = I
  return hipSuccess;
}

hipError_t init_slots_launch(const int *desc, int n, const ModelTable &mt, hipStream_t) {
  for (int k = 0; k < n; k++) {
    const int slot = desc[5 * k];
    mt.col[slot] = desc[5 * k + 1];
    mt.rank[slot] = desc[5 * k + 2];
    mt.jk_mode[slot] = desc[5 * k + 3];
    mt.jk_fiber[slot] = desc[5 * k + 4];
    mt.iters[slot] = 1;
    mt.err[slot] = mt.fit[slot] = mt.old_fit[slot] = 0.0;
    mt.potrf_info[slot] = mt.ls_iter[slot] = mt.ls_updated_last[slot] = mt.flags[slot] = 0;
  }
  return hipSuccess;
}

// pseudo line search: a deterministic subset of the models "changes" every sweep (flags bit 0) so that
// the engine's pending-T / stale-column logic runs
hipError_t ls_launch(const LsArgs &a, hipStream_t) {
  for (int k = 0; k < a.n_slots; k++) {
    const int slot = a.slots[k];
    const bool hit = ((slot * 3 + a.mt.iters[slot]) % 5) == 0 && a.mt.iters[slot] < a.max_iter;
    a.mt.flags[slot] = hit ? 1 : 0;
    if (hit && a.changed) *a.changed += a.mt.rank[slot];
  }
  return hipSuccess;
}

// cals.cpp:336-354 with a deterministic stand-in for "fit_diff < tol": model in slot s "converges" at
// iteration 3 + (7 s + col) mod 17
hipError_t finish_launch(const FinishArgs &a, hipStream_t) {
  for (int k = 0; k < a.n_slots; k++) {
    const int slot = a.slots[k];
    const long long it = a.mt.iters[slot];
    a.mt.err[slot] = 1000.0 * slot + (double)it;  // recognisable values for the host read-back
    a.mt.old_fit[slot] = a.mt.fit[slot];
    a.mt.fit[slot] = 1.0 / (1.0 + (double)it);
    const bool converged = it >= 3 + (7 * slot + a.mt.col[slot]) % 17;
    const bool evict = a.force_max_iter ? it >= a.max_iter : (converged || it >= a.max_iter);
    if (a.evict_enabled && evict)
      a.mt.flags[slot] |= 4;
    else
      a.mt.iters[slot] = it + 1;
  }
  return hipSuccess;
}

hipError_t pack_status_launch(const int *slots, int n, const ModelTable &mt, const int *changed,
                              const int *nnls_status, StatusRec *out, hipStream_t) {
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
    out[1 + k] = r;
  }
  return hipSuccess;
}

hipError_t stale_cols_launch(const int *slots, int n, const ModelTable &mt, int *idx, hipStream_t) {
  int at = 0;
  for (int k = 0; k < n; k++) {
    const int slot = slots[k];
    if (mt.flags[slot] & 3)
      for (int c = 0; c < mt.rank[slot]; c++) idx[at++] = mt.col[slot] + c;
  }
  return hipSuccess;
}

// the two data-moving kernels, as model_kernels.hip defines them: buffer b, column k of the list <->
// scratch + scratch_off[b] + rows * words_per_elem * k
hipError_t gather_columns_launch(const ColMoveArgs &a, hipStream_t) {
  for (int b = 0; b < a.n_bufs; b++) {
    const size_t words = (size_t)a.buf[b].rows * a.buf[b].words_per_elem;
    for (int k = 0; k < a.n_cols; k++) {
      unsigned *src = static_cast<unsigned *>(a.buf[b].ptr) + words * (size_t)a.src[k];
      std::memcpy(a.scratch + a.scratch_off[b] + words * (size_t)k, src, words * sizeof(unsigned));
      if (b < a.zero_src_bufs) std::memset(src, 0, words * sizeof(unsigned));
    }
  }
  return hipSuccess;
}
hipError_t scatter_columns_launch(const ColMoveArgs &a, hipStream_t) {
  for (int b = 0; b < a.n_bufs; b++) {
    const size_t words = (size_t)a.buf[b].rows * a.buf[b].words_per_elem;
    for (int k = 0; k < a.n_cols; k++)
      std::memcpy(static_cast<unsigned *>(a.buf[b].ptr) + words * (size_t)a.dst[k],
                  a.scratch + a.scratch_off[b] + words * (size_t)k, words * sizeof(unsigned));
  }
  return hipSuccess;
}
hipError_t set_cols_launch(const int *pairs, int n, int *col, hipStream_t) {
  for (int k = 0; k < n; k++) col[pairs[2 * k]] = pairs[2 * k + 1];
  return hipSuccess;
}

}  // namespace calship
