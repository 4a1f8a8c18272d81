/*
 * cals_hip.h -- C ABI of the MI355X-native Concurrent-ALS (CALS) engine (libcals_hip.so).
 *
 * The reference (HPAC/CP-CALS) has no FFI: its boundary for this path is the C++ function
 *     cals::CalsReport cals::cp_cals(const Tensor&, KtensorQueue&, CalsParams&)   include/cals.h:196
 * and the value classes Tensor / Ktensor / MultiKtensor.  This header is the plain-C layer that a
 * binding for that boundary needs (C++ header layer: cp-cals_amd/cals/, MEX, ctypes ...): opaque
 * handle, raw pointers and sizes, int status codes, no exceptions, no torch/HIP types.
 * One engine per GPU, one host thread per engine.  All matrices are column-major doubles exactly
 * as the reference's Matrix (include/matrix.h:9-23); file:line citations are relative to the
 * reference tree.
 */
#ifndef CALS_HIP_H
#define CALS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CALS_HIP_MAX_MODES 8
#define CALS_HIP_MAX_RANK 256 /* per-model rank limit.  Ranks 1..32: register / LDS update bodies; 33..64: H in
                               * LDS, rows in registers; 65..256: H and the row solves through global memory
                               * (both update methods; the NNLS active set of a row is (rank + 63) / 64 words) */

/* status codes */
enum {
  CALS_HIP_OK = 0,
  CALS_HIP_ERR_ARG = 1,      /* bad argument (see cals_hip_last_error) */
  CALS_HIP_ERR_HIP = 2,      /* a HIP runtime call failed; reference: cuda_utils.cpp:28-90 exit()s */
  CALS_HIP_ERR_STATE = 3,    /* call out of order (e.g. no tensor set) */
  CALS_HIP_ERR_FULL = 4,     /* BufferFull, include/multi_ktensor.h:123-127 */
  CALS_HIP_ERR_NO_DEVICE = 5 /* no HIP device / kernel image not loadable: the engine never
                                falls back to a CPU path */
};

typedef struct cals_hip_engine cals_hip_engine;

/* The CalsParams fields that steer the loop (include/cals.h:138-159), same names and defaults.
 * mttkrp_method/mttkrp_lut select among CPU variants in the reference and have no meaning here
 * (one fused kernel family). */
typedef struct {
  int64_t max_iterations;   /* 200 */
  double tol;               /* 1e-7 */
  int line_search;          /* 0 */
  int line_search_interval; /* 5 */
  double line_search_step;  /* 0 => cbrt(model iteration), src/cals.cpp:317-318 */
  int line_search_method;   /* ls::LS_METHOD: 0 NO_ERROR_CHECKING, 1 ERROR_CHECKING_SERIAL (the candidate's error
                             * from one extra MTTKRP), 2 ERROR_CHECKING_PARALLEL (never dispatched by the
                             * reference, line_search.cpp:228-283: no step is taken; same here) */
  int force_max_iter;       /* 0 */
  int always_evict_first;   /* 0 */
  int update_method;        /* update::UPDATE_METHOD (include/utils/update.h:7): 0 UNCONSTRAINED, 1 NNLS */
} cals_hip_params;

/* CalsReport result fields (include/cals.h:27-63) + device timings (ms) from hipEvents. */
typedef struct {
  int64_t iter;             /* outer sweeps of the loop, CalsReport::iter */
  int64_t n_ktensors;       /* models admitted */
  int64_t ktensor_comp_sum; /* sum of their ranks */
  int64_t ls_performed;
  int64_t ls_failed;
  double X_norm;
  double total_ms;          /* whole cals_hip_run call (host clock) */
  double loop_ms;           /* do{}while loop only (host clock, device synchronised) */
  int nnls_status;          /* NNLS update, OR over all rows since create: 0 clean; 1 a Cholesky failed in
                             * the main loop (the reference ends on the uncaught CholFail, update.cpp:131);
                             * 2 a row cycled or hit the bound of max(64, 16 rank) set exchanges (the reference has neither test) */
} cals_hip_report;

/* Per-model results that the reference keeps inside Ktensor (include/ktensor.h:27-33). */
typedef struct {
  int64_t iters;
  double fit, old_fit, approx_error;
  int evicted; /* 1 once the factors/lambda have been written back to the caller's buffers */
} cals_hip_model_status;

/* Live kernel statistics (hipEvent pairs around every launch of the named kernel on the
 * engine's stream) gathered while profiling is enabled. */
typedef struct {
  int64_t mttkrp_launches;
  double mttkrp_ms;         /* sum of launch durations */
  double mttkrp_flops;      /* algorithmic: 2 * prod(modes) * active_cols per launch, summed */
  int64_t update_launches;
  double update_ms;
  int64_t other_launches;
  double other_ms;
  /* dimension-tree pair (3-way tensors): the TTM GEMM that replaces two of the three MTTKRPs of a
   * sweep (flops counted as one MTTKRP: 2*prod(modes)*R) and the HBM-bound contraction of T */
  int64_t ttm_launches;
  double ttm_ms;
  double ttm_flops;
  int64_t contract_launches;
  double contract_ms;
  double contract_bytes;
} cals_hip_kernel_stats;

/* One record per outer sweep of cals_hip_run / cals_hip_step while the sweep log is on: what the
 * reference keeps in CalsReport::{als_times, mode_times, mttkrp_times, flops_per_iteration, cols}
 * (include/cals.h:55-63, filled at src/cals.cpp:213-217, 269-275, 367-370).  Host-clock columns are
 * wall times of the loop iteration (the per-sweep status read-back synchronises the stream); device
 * columns are sums of hipEvent pairs around the launches, per mode being updated. */
typedef struct {
  int64_t cols;          /* CalsReport::cols: active columns of the multi-factors */
  int64_t models;        /* models in flight */
  double flops;          /* 2 * prod(modes) * cols per MFMA-kernel launch (fused MTTKRP or TTM), summed */
  double iteration_ms;   /* ALS_TIMERS::ITERATION (host clock: admission .. compress) */
  double admit_ms;       /* host clock: admission phase (enqueue only; the copies run asynchronously) */
  double defrag_ms;      /* ALS_TIMERS::DEFRAGMENTATION: eviction + compress (host clock) */
  double ls_ms;          /* ALS_TIMERS::LINE_SEARCH: snapshot + line-search kernels (device) */
  double mttkrp_ms[CALS_HIP_MAX_MODES];   /* MODE_TIMERS::MTTKRP per mode = fused + ttm + contract + krp */
  double update_ms[CALS_HIP_MAX_MODES];   /* MODE_TIMERS::UPDATE per mode: reduce + (NNLS) + update kernel
                                           * (the fast error of ALS_TIMERS::ERROR is fused into the last mode's) */
  double fused_ms[CALS_HIP_MAX_MODES];    /* MTTKRP_TIMERS::MT_GEMM: the fused MTTKRP kernel */
  double ttm_ms[CALS_HIP_MAX_MODES];      /* MTTKRP_TIMERS::TS_GEMM: the TTM (+ its operand packing) */
  double contract_ms[CALS_HIP_MAX_MODES]; /* MTTKRP_TIMERS::TS_GEMV: the contraction of T */
  double krp_ms[CALS_HIP_MAX_MODES];      /* MTTKRP_TIMERS::MT_KRP: Khatri-Rao of the streamed modes (N > 3) */
} cals_hip_sweep_record;

void cals_hip_default_params(cals_hip_params *p);

/* ---- lifetime ---- */
/* Replaces: MultiKtensor(modes, buffer_size) src/multi_ktensor.cpp:8-12 + the device set-up of
 * cp_cals src/cals.cpp:142-167 (streams, device buffers).  buffer_size = CalsParams::buffer_size
 * (columns of every multi-factor).  device = HIP ordinal. */
int cals_hip_create(cals_hip_engine **out, int n_modes, const int64_t *modes, int64_t buffer_size,
                    int device);
/* Same, with the STORAGE (= arithmetic) type of the tensor copies, the multi-factors and the MTTKRP
 * on the device: CALS_HIP_F64 (the reference's only type, v_mfma_f64_16x16x4_f64) or CALS_HIP_F32
 * (BASELINE config 4: v_mfma_f32_16x16x4_f32, fp32 accumulate).  In fp32 mode the per-model
 * Gramians, Cholesky, solves' accumulators, lambda, error and fit stay fp64; factors cross this ABI
 * as doubles either way (rounded to fp32 once, on admission).  There is no reference counterpart:
 * the reference is fp64 throughout (include/matrix.h:26, `double *data`). */
#define CALS_HIP_F64 0
#define CALS_HIP_F32 1
int cals_hip_create_ex(cals_hip_engine **out, int n_modes, const int64_t *modes,
                       int64_t buffer_size, int device, int dtype);
int cals_hip_dtype(const cals_hip_engine *e);
/* MTTKRP plan of a 3-way engine, fixed at create.  0 = three fused MTTKRPs per sweep
 * (mttkrp::MTTKRP semantics, src/utils/mttkrp.cpp:218-328).  Dimension-tree plans use the
 * reference's two-step association (mttkrp.cpp:330-560) with the TTM T = X x_a P shared by the two
 * modes updated while factor a stays fixed: 1 ("A") = modes 0,1 share X x_2 C; 2 ("B") = modes 1,2
 * share X x_0 A; 3 ("M", multi-sweep) = every TTM serves two consecutive updates of the sequence
 * A B C A B C ..., also across the sweep boundary (3 TTMs per 2 sweeps); a T that a line-search
 * step, an eviction or an admission made stale is dropped and recomputed.  Chosen by a cost model;
 * CALS_HIP_TREE=0|A|B|M in the environment at create overrides it.
 * N > 3 modes: 4 = two-group dimension tree -- modes [0, h) and [h, N) are merged into one mode each (adjacent
 * modes: a reshape), one fused MTTKRP per group against the other group's factors gives T_group, and every mode
 * of the group comes from a per-column contraction of T_group with the group's other factors: 2 MTTKRP-sized
 * contractions per sweep instead of N (the reference runs one Khatri-Rao + GEMM per mode, mttkrp.cpp:147-176,
 * 218-328).  CALS_HIP_TREE=0 keeps the N fused MTTKRPs. */
int cals_hip_tree(const cals_hip_engine *e);
/* Re-targets an IDLE engine (nothing queued or in flight) at another buffer_size <= the capacity it
 * was created with: same tensor copies in HBM, same plan, same device buffers, fresh packing state
 * (occupancy, ids, tickets restart at 0).  Replaces: the next cp_cals call constructing its
 * MultiKtensor (src/cals.cpp:117) while the Tensor keeps its device mirror (include/tensor.h:56-59,
 * src/cals.cpp:144-147: X is uploaded only when it has no device copy yet).  CALS_HIP_ERR_FULL when
 * buffer_size exceeds the capacity, CALS_HIP_ERR_STATE when models are queued or in flight. */
int cals_hip_rebind(cals_hip_engine *e, int64_t buffer_size);
int64_t cals_hip_capacity(const cals_hip_engine *e); /* buffer_size the engine was created with */
int cals_hip_destroy(cals_hip_engine *e);
const char *cals_hip_last_error(const cals_hip_engine *e);

/* Replaces: X.allocate_cudata + send_to_device_async, src/cals.cpp:144-147; X.norm()
 * include/tensor.h:196; utils::calculate_jackknifing_norms src/utils/utils.cpp:103-152.
 * X_host: prod(modes) doubles, mode 0 fastest (include/tensor.h:173).  The engine keeps one
 * zero-padded permuted copy per mode in HBM (layout: DESIGN.md). */
int cals_hip_set_tensor(cals_hip_engine *e, const double *X_host);
/* Same for a tensor held in fp32 on the host (any engine dtype; an F64 engine widens it). */
int cals_hip_set_tensor_f32(cals_hip_engine *e, const float *X_host);

int cals_hip_set_params(cals_hip_engine *e, const cals_hip_params *p);

/* ---- the KtensorQueue ---- */
/* Replaces: kt_queue.emplace(ktensor) by the callers (src/examples/driver.cpp:177-180).
 * factors[n]: I_n x rank col-major (ld = I_n), lambda: rank doubles; both are read at admission
 * and overwritten at eviction (Ktensor::attach/detach, src/ktensor.cpp:109-135) and must stay
 * valid until cals_hip_run returns.  jk_mode < 0: regular model; else the model is a jackknife
 * replica (Ktensor jk ctor include/ktensor.h:83-88).  ticket: handle for cals_hip_model_result. */
int cals_hip_enqueue(cals_hip_engine *e, int64_t rank, double *const *factors, double *lambda,
                     int jk_mode, int64_t jk_fiber, int64_t *ticket);

/* Replaces: the whole cp_cals do{}while loop, src/cals.cpp:174-382: admission, per-mode MTTKRP +
 * update, error, line search, eviction, compress, until queue and registry are empty. */
int cals_hip_run(cals_hip_engine *e, cals_hip_report *rep);

/* Per-model results.  st->evicted = 1 means the model's factors and lambda are complete in the caller's
 * storage: an evicted model's data leaves the device asynchronously (it is scattered into the caller's
 * buffers while the next sweep runs) and this call, cals_hip_run, cals_hip_evict and
 * cals_hip_synchronize complete any transfer still pending before they return. */
int cals_hip_model_result(const cals_hip_engine *e, int64_t ticket, cals_hip_model_status *st);

/* ---- step-wise control (what cals_hip_run is made of; used by bench.py and the tests) ---- */
/* Admission phase only, src/cals.cpp:182-192: first-fit placement, H2D of the model's columns,
 * Gramians, iters = 1.  n_admitted may be NULL. */
int cals_hip_admit(cals_hip_engine *e, int64_t *n_admitted);
/* n_sweeps ALS sweeps over the in-flight models (src/cals.cpp:203-331: LS snapshot, per-mode
 * MTTKRP + update, error/fit, line search, then iters++ for every model) WITHOUT eviction:
 * the caller guarantees no model reaches max_iterations or converges meanwhile (bench: force_max_iter). */
int cals_hip_sweep(cals_hip_engine *e, int64_t n_sweeps);
/* Eviction + compress phase, src/cals.cpp:336-362, applied to the current status. */
int cals_hip_evict(cals_hip_engine *e, int64_t *n_evicted);
/* One iteration of cals_hip_run's loop -- admit what fits, one sweep with the eviction rule of
 * src/cals.cpp:336-354, evict + compress -- for callers that keep feeding the queue while the engine
 * runs (work-queue hand-off between GPUs, cp-cals_amd/multi_gpu.py).  No-op when nothing is queued
 * or in flight.  cals_hip_get_report: the counters of cals_hip_report accumulated so far (iter =
 * sweeps done by this engine, times 0). */
/* (The factors of the models a step evicts reach the callers' storage during the NEXT step -- or at
 * cals_hip_model_result / cals_hip_synchronize, whichever comes first.) */
int cals_hip_step(cals_hip_engine *e, int64_t *n_admitted, int64_t *n_evicted);
int cals_hip_get_report(const cals_hip_engine *e, cals_hip_report *rep);
int64_t cals_hip_active_cols(const cals_hip_engine *e);  /* mkt.get_factor(0).get_cols() */
int64_t cals_hip_models_in_flight(const cals_hip_engine *e);
int64_t cals_hip_queue_size(const cals_hip_engine *e);
int cals_hip_synchronize(cals_hip_engine *e);

/* ---- kernel-level entry point ---- */
/* mttkrp::mttkrp(X, u, workspace, mode, params) for ONE Ktensor (include/utils/mttkrp.h:77-81,
 * src/utils/mttkrp.cpp:562-614; what include/experiments/bench_mttkrp_cals.h:49-84 times): the MTTKRP of mode
 * `mode` from the host factors `factors[n]` (I_n x rank, ld = I_n; factors[mode] is not read) on the engine's
 * tensor, through the fused MTTKRP kernel + the split reduction, into G_host (I_mode x rank, ld = I_mode).
 * The engine must be idle (nothing queued or in flight) and rank <= buffer_size; its buffers are left zeroed as
 * found.  device_ms (may be NULL): time of the MTTKRP kernels on the engine's stream, from a hipEvent pair. */
int cals_hip_mttkrp(cals_hip_engine *e, int64_t rank, const double *const *factors, int mode, double *G_host,
                    double *device_ms);

/* ---- inspection (tests) ---- */
/* MTTKRP of the current multi-factor block for `mode` (mttkrp::mttkrp, src/utils/mttkrp.cpp:562)
 * into G_host (I_mode x active_cols, ld = I_mode) WITHOUT touching the engine state. */
int cals_hip_debug_mttkrp(cals_hip_engine *e, int mode, double *G_host);
/* Same through an explicit path: PLAIN = fused MTTKRP kernel; FIRST = as the first mode of a
 * dimension-tree pair (TTM kernel, G fused); SECOND = as the second mode (TTM of the previous mode
 * from the current factors, then the contraction kernel).  CALS_HIP_ERR_STATE if the engine's plan
 * has no such path for this mode. */
#define CALS_HIP_PATH_PLAIN 0
#define CALS_HIP_PATH_FIRST 1
#define CALS_HIP_PATH_SECOND 2
int cals_hip_debug_mttkrp_path(cals_hip_engine *e, int mode, int path, double *G_host);
/* Copy multi-factor `mode` (I_mode x active_cols) / lambda (active_cols) / the column-indexed
 * Gramian store (CALS_HIP_MAX_RANK x active_cols per mode) to the host. */
int cals_hip_debug_get_factor(cals_hip_engine *e, int mode, double *host);
int cals_hip_debug_get_lambda(cals_hip_engine *e, double *host);
int cals_hip_debug_get_gramian(cals_hip_engine *e, int mode, double *host);
/* Status of an in-flight model (by ticket) as the device currently holds it. */
int cals_hip_debug_model_status(cals_hip_engine *e, int64_t ticket, cals_hip_model_status *st,
                                int64_t *col);
/* jackknife norms / X norm computed on the device at set_tensor */
int cals_hip_debug_get_norms(cals_hip_engine *e, double *X_norm, double *jk_norms /* modes[0] or NULL */);
/* How close to a tie the model's line-search decisions were: the smallest |e1 - e2| / max(|e1|, |e2|) over every
 * accept / revert test the model has been through (ls::line_search's `backup error < error`,
 * src/utils/line_search.cpp:239; the error-checking methods' `new error < old error`, :116); 1e300 when it has been
 * through none.  A converged model's extrapolation is a null step and its test compares two errors that agree to
 * rounding: either outcome is then a valid trajectory of the reference's algorithm.  The tests that tolerate a
 * flipped decision use this to show that every flipped one WAS such a tie.  No counterpart in the reference. */
int cals_hip_debug_ls_margin(cals_hip_engine *e, int64_t ticket, double *margin);

/* ---- host logic of MultiKtensor, exposed so that it can be tested without a GPU ---- */
/* MultiKtensor::check_availability (src/multi_ktensor.cpp:14-39) on an occupancy vector (id per
 * column, 0 = free): first column of the first free run of `rank` columns, or -1 (BufferFull). */
int64_t cals_hip_host_first_fit(const int64_t *occupancy, int64_t n_cols, int64_t rank);
/* MultiKtensor::compress move list (src/multi_ktensor.cpp:196-209): for every model that has free
 * columns to its left, (id, offset); returns the number of moves (at most max_moves written). */
int64_t cals_hip_host_compress_plan(const int64_t *occupancy, int64_t n_cols, int64_t *ids,
                                    int64_t *offsets, int64_t max_moves);
/* MultiKtensor::adjust_edges (src/multi_ktensor.cpp:165-186): active width = last occupied
 * column + 1, never below 1 (cell 0 is not examined). */
int64_t cals_hip_host_active_cols(const int64_t *occupancy, int64_t n_cols);

/* ---- measurement ---- */
/* level: 0 off; 1 hipEvent pairs around every launch; 2 around the MFMA kernels (MTTKRP, TTM) and the
 * contraction only -- the pairs themselves cost ~5 us per launch of queue time, which matters when a
 * sweep is 13 launches of 4-40 us (BASELINE config 2); 3 around the MFMA kernels only. */
int cals_hip_set_profiling(cals_hip_engine *e, int level);
int cals_hip_get_kernel_stats(cals_hip_engine *e, cals_hip_kernel_stats *out);
int cals_hip_reset_kernel_stats(cals_hip_engine *e);
/* hipStream_t the engine launches on (as void*), so callers can bracket it with their own events.  (While models of
 * rank > 32 are in flight the engine forks two launches per mode -- their Hadamard product and Cholesky factor -- to
 * an internal side stream and joins them back into this one by an event before they are needed: work recorded on
 * this stream still brackets everything.) */
void *cals_hip_stream(cals_hip_engine *e);
/* Per-sweep log for CalsReport's timer matrices: on = an event pair around every launch (profiling
 * level 1) and one record per sweep; get returns the number of records (copies at most max_records). */
int cals_hip_set_sweep_log(cals_hip_engine *e, int enabled);
int64_t cals_hip_get_sweep_log(cals_hip_engine *e, cals_hip_sweep_record *out, int64_t max_records);
/* diagnostics (CALS_MTTKRP_CLOCK=1 in the environment at create): median over workgroups of the
 * shader cycles a MTTKRP workgroup ran and of the clock (GHz) it saw (s_memtime / s_memrealtime) */
int cals_hip_debug_clock(cals_hip_engine *e, int n_workgroups, double *cycles_median, double *ghz_median);
/* diagnostics (library built with CALS_DIAG=1, CALS_TTM_TRACE=1 at create): shader-clock stamps taken
 * before the DMA wait / before / after the stage barrier by waves 0 and 4 of 8 workgroups of the
 * last ttm_kernel launch: out[((wg * 2 + group) * 512 + stage) * 4 + {0, 1, 2}] (tools/ttm_trace.py) */
int cals_hip_debug_ttm_trace(cals_hip_engine *e, uint64_t *out, int n);
/* Fatal-signal evidence for host processes that capture file descriptor 2 (test runners): on SIGSEGV / SIGBUS /
 * SIGABRT / SIGFPE / SIGILL the contents of the current stderr file (if it is a regular file other than
 * evidence_fd: the capture of the running test, with the runtime's "Memory access fault by GPU" line in it) and a
 * backtrace of the faulting thread are written to evidence_fd -- a descriptor the caller saved before the capture
 * began --, then the previously installed handler and the default action run.  Replaces nothing in the reference
 * (its fatal device errors print to cerr and exit(EXIT_FAILURE), src/cuda_utils.cpp:28-90). */
int cals_hip_debug_install_crash_trace(int evidence_fd);
/* number of HIP devices visible (0 when none); never initialises a context on failure */
int cals_hip_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* CALS_HIP_H */
