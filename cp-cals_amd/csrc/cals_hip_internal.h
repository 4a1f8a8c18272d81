// Internal declarations shared by the HIP kernels and the engine (not part of the C ABI).
#ifndef CALS_HIP_INTERNAL_H
#define CALS_HIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#define CALS_MAX_MODES 8
#define CALS_RMAX 64          // ranks up to this: register / LDS update bodies, one-wave Cholesky, NNLS masks,
                              // error-checking line search
#define CALS_GLD 256          // leading dimension of the Gramian stores = rank limit per model; ranks above
                              // CALS_RMAX take update_body_huge (H and the row solves through global memory)
#define CALS_RFAST 32         // ranks up to this run the register-resident update bodies
#define CALS_BN 128           // columns of the multi-factor per MTTKRP workgroup

namespace calship {

// hipFuncSetAttribute applies to the current device only: one flag per device for every kernel
// instantiation that raises its dynamic LDS limit (a process may drive several GPUs, one engine each).
// ensure(set): runs `set` (the hipFuncSetAttribute call) once per device, under a lock, and records the
// device as done only AFTER it succeeded -- a second host thread driving another engine on the same
// device either sees the bit (attribute applied) or waits for the lock; a failure is retried next time.
struct AttrOnce {
  std::atomic<unsigned long long> done{0};
  std::mutex mu;
  template <typename F>
  hipError_t ensure(F &&set) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    std::lock_guard<std::mutex> g(mu);
    if (done.load(std::memory_order_relaxed) & bit) return hipSuccess;
    const hipError_t e = set();
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// ---------------------------------------------------------------------------------------------
// MTTKRP  G[m,c] = sum_{a,s} Xp[m,a,s] * P[a,c] * Q[s,c]     (DESIGN.md "MTTKRP kernel")
// ---------------------------------------------------------------------------------------------
enum { CALS_F64 = 0, CALS_F32 = 1 };  // storage type of X, the multi-factors and the partials

struct MttkrpArgs {
  // element type = dtype (double | float)
  const void *Xp;    // permuted, zero-padded tensor copy for this mode: [Mp][Ap][S], m fastest
  const void *P;     // factor of the inner ("a") mode, A x R, ld = ldP
  const void *Q;     // factor (or Khatri-Rao of the factors) of the streamed modes, S x R, ld = ldQ
  void *partial;     // split partial results: [NB*T] tiles of ldPart x CALS_BN, col-major
  int dtype;
  long long S, ldP, ldQ;
  int Mp, Ap, A;
  int R;             // active columns
  int NB, T;         // column blocks, team size (workgroups per column block)
  int ldPart;        // rows of one partial tile (= m_blocks * 16 * MT)
  int grid;          // NB * T
  int dbg_no_units;  // timing diagnostics only: skip the unit loop (prologue + epilogue cost)
  int dbg_no_stagger;  // v3: 1 = all waves take the barrier mid-slab (A/B test of the stagger)
  int dbg_no_barrier;  // timing diagnostics only (v3): drop the per-stage barrier (results garbage)
  int dbg_prio;        // diagnostics (v3): 1 = s_setprio 1 for waves 4-7, 2 = for waves 0-3
  int dbg_no_dma;    // timing diagnostics only (v3): skip the steady-state LDS-DMA (results garbage)
  unsigned long long *dbg_clock;  // diagnostics: per workgroup {s_memtime, s_memrealtime} deltas
};

// smallest supported tile count >= mt (0 if mt > max): instantiated MT values
int mttkrp_pick_mt(int m_tiles);
// mttkrp_kernel_v3.hip: 8-wave tiling, 3-buffer LDS ring, mid-stage barrier
hipError_t mttkrp3_launch(int MT, int m_blocks, const MttkrpArgs &a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Dimension-tree pair (ttm_kernel.hip): T[m,s,c] = sum_a Xp[m,a,s] P[a,c];
// G_first[m,c] = sum_s T Q[s,c] (fused); G_second[s,c] = sum_m T F[m,c] (contract)
// ---------------------------------------------------------------------------------------------
struct TtmArgs {
  const void *Xp;    // [Mp][Ap][S] permuted padded copy of mode `first` with a = the third mode
  const void *Pt;    // packed factor of mode a: [NB][Ap][CALS_BN], zero padded (pack_pt_launch)
  const void *Q;     // factor of mode s (= `second`), S x R, ld = ldQ
  void *partial;     // G_first partial tiles: [NB*T] of ldPart x CALS_BN
  void *Tout;        // T[c][s][m], m fastest, pitch Mp
  int dtype;
  long long S, ldQ;
  int Mp, Ap, R;
  int A;             // true size of mode a (rows >= A of Xp and Pt are zero padding)
  int NB, T;         // column blocks, team size (workgroups per column block and M block; split s)
  int ldPart;
  int grid;          // NB * T workgroups (1-D); each walks all m_blocks M blocks
  int nbw;           // column blocks per XCD-locality group (ttm_kernel's workgroup mapping)
  int m_blocks, k_big, MT;  // M blocks: the first k_big are MT tiles high, the others MT - 1
  int dbg;           // CALS_DIAG builds, CALS_TTM_DBG bits: 1 no T stores, 2 no stagger, 4 one X slab
                     // (L2 hits), 8 no P DMA, 16 no X DMA -- timing experiments, results garbage
  unsigned long long *dbg_trace;  // CALS_DIAG builds + CALS_TTM_TRACE=1: per-stage clock stamps
};
int ttm_max_mt(int dtype);
bool ttm_shape_ok(long long S, long long Mp, int dtype);  // the TTM's 32-bit T store offsets cover this pair
hipError_t ttm_launch(const TtmArgs &a, hipStream_t st);
hipError_t pack_pt_launch(const void *P, long long ldP, int A, int Ap, int NB, int R, void *Pt,
                          int dtype, hipStream_t st);
hipError_t contract_launch(const void *Tb, long long S, int Mp, int M, const void *F,
                           long long ldF, void *out, long long ldOut, int R, int dtype,
                           hipStream_t st);

// Q[s,c] for N > 3: Khatri-Rao of the streamed modes' factors (first streamed mode fastest)
struct KrpArgs {
  const void *F[CALS_MAX_MODES];  // element type = dtype
  long long ld[CALS_MAX_MODES];
  int dims[CALS_MAX_MODES];
  int n;         // number of streamed modes
  long long S;   // prod(dims)
  int R;
  void *Q;       // S x R, ld = S
  int dtype;
};
hipError_t krp_launch(const KrpArgs &a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// per-model state (device arrays indexed by slot)
// ---------------------------------------------------------------------------------------------
struct ModelTable {
  int *col;            // first column in every multi-factor
  int *rank;
  long long *iters;    // Ktensor::iters
  int *jk_mode;        // -1: regular model
  int *jk_fiber;
  double *err, *fit, *old_fit;
  int *potrf_info;     // last dpotrf info (0 ok)
  // line search (ls::LineSearchParams, include/utils/line_search.h:15-32)
  int *ls_iter;
  int *ls_updated_last;
  double *bk_err, *bk_fit, *bk_old_fit;  // backup_ktensor scalars
  long long *bk_iters;
  int *flags;          // bit0: extrapolated this sweep, bit1: reversed this sweep, bit2: evict
  // smallest relative distance |e1 - e2| / max(|e1|, |e2|) between the two errors of any accept / revert test this
  // model has been through (line_search.cpp:239 / :116): a test whose margin is at rounding level is a tie, and
  // either outcome is a valid trajectory (cals_hip_debug_ls_margin; the tests that allow a flipped decision check it)
  double *ls_margin;
};

// smallest rank whose unconstrained update runs as the pipeline of huge_* launches (33 | 49 | 65; measured)
#define CALS_HUGE_FROM_DEFAULT 33
// jackknife descriptor of a model in one int: -1 = regular model, else (fiber << 3) | mode
inline int upd_jk_pack(int jk_mode, long long jk_fiber) { return jk_mode < 0 ? -1 : (int)((jk_fiber << 3) | jk_mode); }
struct UpdateArgs {
  const int *slots;    // active slots, one wave each
  const int4 *wgdesc;  // per registry position {slot, first column, rank, upd_jk_pack}: what the rank <= CALS_RFAST
                       // kernels need to start, in one load (kept current by the engine: upload_slots)
  int n_slots;
  ModelTable mt;
  void *factor;        // multi-factor of this mode (element type = dtype), I x buffer, ld = I
  int dtype;
  int I;
  double *gram[CALS_MAX_MODES];  // column-indexed Gramian stores: CALS_GLD x buffer, ld CALS_GLD
  // end-of-sweep rule (finish_kernel's: eviction flag or iters++, cals.cpp:336-354) applied by the last mode's
  // launch itself when nothing runs between the update and the rule (no line search): one launch less per sweep
  struct {
    int on;
    long long max_iter;
    double tol;
    int force_max_iter;
    int evict_enabled;
  } fin;
  double *hscratch;    // ranks > CALS_RMAX in flight: one CALS_GLD x CALS_GLD block (H / L) per such model,
  int *hcounter;       // handed out through this counter (zeroed before the launch)
  double *lambda;      // per column
  int n_modes, mode;
  int is_last;
  double X_norm;
  const double *jk_norms;
  unsigned long long *dbg_trace;  // CALS_DIAG builds: phase stamps of the first rank-20 model (mode 0)
  int xld;             // set by update_launch: > 0 = leading dimension of the LDS-resident panel
  // NNLS update: nnls_launch already replaced the MTTKRP result in `factor` by the constrained
  // solution and left <x_row, g_row> here ([n_slots][I]); the kernel then skips Cholesky + solves
  const double *rowdot;
  // Split-K partial tiles of the MTTKRP that produced this mode's G ([NB * pT] tiles of ldPart x CALS_BN, element
  // type = dtype), or nullptr when G already stands in `factor`.  Non-null: the bodies for ranks <= CALS_RFAST sum
  // a model's columns over the pT tiles themselves (fixed order t = 0 .. pT-1, fp64, rounded to the storage type:
  // exactly what reduce_partials_kernel writes) -- no reduce launch, no 78 MB round trip at C3.
  const void *partial;
  int pT, ldPart;
  // Packed B-operand tiles of the NEXT dimension-tree TTM whose inner mode is this one (Pt[column block][a][128],
  // a < ptAp, ttm_kernel.hip) or nullptr: the bodies write their model's columns of the normalised factor there
  // on their way out -- no pack_pt launch in front of that TTM.
  void *pt;
  int ptAp;
  // Models of rank >= huge_from (33 by default: every rank the register-resident bodies do not take): their registry
  // positions (n_huge of them; a suffix of the engine's class list), the H / L block of model h is block h of hscratch.
  // Their update is a pipeline of launches (update_launch).  hrowdot: [n_huge][I], written by the solve launch when
  // rowdot is nullptr (unconstrained update).
  const int *huge_idx;
  int n_huge;
  int huge_from;       // smallest rank the pipeline takes (33 | 49 | CALS_RMAX + 1; 0 = CALS_RMAX + 1)
  double *hrowdot;
  int huge_factored;   // update_huge_factor_launch already ran for this mode (on a side stream, next to the MTTKRP)
};
// classes: bit 0 = models of rank <= CALS_RFAST in flight, bit 1 = ranks 33..huge_from - 1, bit 2 = above (0 = unknown:
// every kernel)
hipError_t update_launch(const UpdateArgs &a, int rmax_needed, hipStream_t st, int classes = 0);
// H = hadamard of the other modes' Gramians + its Cholesky factor for the models above CALS_RMAX: the part of their
// update that does not depend on the mode's MTTKRP (then UpdateArgs::huge_factored = 1 for update_launch)
hipError_t update_huge_factor_launch(const UpdateArgs &a, int rmax_needed, hipStream_t st);

// update::update_factor_non_negative_constrained for one mode (nnls_kernel.hip)
struct NnlsArgs {
  const int *slots;
  int n_slots;
  ModelTable mt;
  void *factor;        // in: MTTKRP result, out: constrained solution (element type = dtype)
  int dtype;
  int I;
  double *gram[CALS_MAX_MODES];
  int n_modes, mode;
  unsigned long long *act;  // Ktensor::active_set of this mode: [I x buffer]; word q of (row, model) -- the
                            // constraints of components 64 q .. 64 q + 63, bit set = active -- at
                            // row + I * (col(model) + q)
  double *rowdot;      // out: [n_slots][I]
  int *status;         // sticky OR: 1 Cholesky failure in the main loop, 2 exchange bound reached
  int rmax;            // largest rank in flight (sizes the LDS tiles)
  unsigned rank_classes;  // bit k: a model of nnls_rank_class k is in flight (0: one launch sized by rmax)
  const int *cls_idx;  // registry positions sorted by nnls_rank_class (or nullptr: every launch walks all models and
  int cls_off[8];      // the other classes' workgroups return at once); class k = positions [cls_off[k], cls_off[k+1])
  int seg_first[3], seg_rmax[3], seg_chunks[3];  // set by nnls_launch: the merged launch of the classes <= 16 / 24 / 32
  const int *idx;      // set by nnls_launch: this launch's models (registry positions), n_cls of them; nullptr = all
  int n_cls;
  int rlo, rhi;        // set by nnls_launch: the ranks this launch serves
  int chunks;          // set by nnls_launch: workgroups per model
  unsigned long long *dbg_counts;  // CALS_DIAG builds: {rows, solves, factorisations, main-loop passes, inner passes}
  int huge_chunk_cap;  // > 0: at most this many workgroups per model of rank > CALS_RMAX (the engine's scratch holds no more)
  double *hscratch;    // models above CALS_RMAX: n_huge * min(nnls_huge_chunks(I, n_huge), huge_chunk_cap) blocks of nnls_huge_block_doubles()
  int *hcounter;       // zero at launch: blocks are handed out in arrival order
};
hipError_t nnls_launch(const NnlsArgs &a, hipStream_t st);
int nnls_rank_class(int r);  // 0: <= 16, 1: <= 24, 2: <= 32, 3: <= 48, 4: <= 64, 5: above (nnls_huge_kernel)
size_t nnls_huge_block_doubles();
int nnls_huge_chunks(int I, int n_huge);
struct NnlsResetArgs {
  unsigned long long *act[CALS_MAX_MODES];
  int I[CALS_MAX_MODES];
  int n_modes;
};
hipError_t nnls_reset_launch(const int *desc, int n, const NnlsResetArgs &a, hipStream_t st);

// deterministic reduction of the MTTKRP split partials into the multi-factor of the mode
hipError_t reduce_partials_launch(const void *partial, int T, int ldPart, int I, int R,
                                  void *factor, int dtype, hipStream_t st);
// the same reduction for a compact column set: column k of the tiles -> column idx[k] of the factor
hipError_t reduce_partials_scatter_launch(const void *partial, int T, int ldPart, int I, int n_cols,
                                          void *factor, const int *idx, int dtype, hipStream_t st);
// idx[0 .. count) = columns of the models the last ls_kernel rewrote (mt.flags & 3), registry order
hipError_t stale_cols_launch(const int *slots, int n, const ModelTable &mt, int *idx, hipStream_t st);
// desc: n x {slot, col, rank, jk_mode, jk_fiber}
hipError_t init_slots_launch(const int *desc, int n, const ModelTable &mt, hipStream_t st);

// Dimension tree for N > 3 modes: the modes are split into two groups of adjacent modes; one fused MTTKRP per
// group over the tensor viewed as (group, other modes...) gives T[(i_0 .. i_{h-1}), c] = the tensor contracted
// with the OTHER group's factors, column by column; the MTTKRP of a mode of the group is then a small
// per-column contraction of T with the remaining factors of the group:
//   out[i_n, c] = sum_{i_k, k != n} T[i_0 + d_0 (i_1 + d_1 (...)), c] * prod_{k != n} F_k[i_k, c]
struct GroupContractArgs {
  const void *T;        // [rows = prod d_k] x R, ld = ldT, element type = dtype
  long long ldT;
  int h;                // modes in the group (2 .. 4)
  int dims[4];
  int n_local;          // the mode (position in the group) the result belongs to
  const void *F[4];     // factors of the group's modes (F[n_local] unused), ld = dims[k]
  void *out;            // dims[n_local] x R, ld = dims[n_local]
  int R;
  int dtype;
};
hipError_t group_contract_launch(const GroupContractArgs &a, hipStream_t st);

// Gramians of all modes for freshly admitted models (MultiKtensor::add, multi_ktensor.cpp:88-94)
struct GramInitArgs {
  const int *slots;
  int n_slots;
  ModelTable mt;
  const void *factor[CALS_MAX_MODES];
  int I[CALS_MAX_MODES];
  double *gram[CALS_MAX_MODES];
  int n_modes;
  int dtype;
};
hipError_t gram_init_launch(const GramInitArgs &a, hipStream_t st);

// line search + end-of-sweep bookkeeping
struct LsArgs {
  const int *slots;
  int n_slots;
  ModelTable mt;
  void *factor[CALS_MAX_MODES];  // element type = dtype
  void *prev[CALS_MAX_MODES];
  void *backup[CALS_MAX_MODES];
  int dtype;
  int I[CALS_MAX_MODES];
  double *gram[CALS_MAX_MODES];
  double *lambda, *prev_lambda, *backup_lambda;
  int n_modes;
  int interval;
  double step;            // 0 => cbrt(iters)
  long long max_iter;
  int *changed;           // ls_kernel: += rank of every model whose factors it rewrote (may be null);
                          // the error-checking kernels set it to 1
  // ERROR_CHECKING line search (ls_ec_*): MTTKRP of mode 0 with the extrapolated factors, I[0] x R
  const void *Gs;
  double X_norm;
  // error-checking line search of models above CALS_RMAX: two CALS_GLD^2 scratch blocks per such model
  double *hscratch;
  int *hcounter;
  // NNLS: Ktensor::copy carries the active sets along (src/ktensor.cpp:174); null otherwise
  unsigned long long *act[CALS_MAX_MODES], *act_backup[CALS_MAX_MODES];
};
hipError_t ls_snapshot_launch(const LsArgs &a, hipStream_t st);  // cals.cpp:203-211
hipError_t ls_launch(const LsArgs &a, hipStream_t st);           // cals.cpp:310-331
// ls::ERROR_CHECKING_SERIAL / _PARALLEL (src/utils/line_search.cpp:86-153, 262-271) in two steps
// around an MTTKRP of the extrapolated factors: prepare writes them over the `prev` copies, decide
// evaluates the error and keeps or drops them
hipError_t ls_ec_prepare_launch(const LsArgs &a, hipStream_t st);
hipError_t ls_ec_decide_launch(const LsArgs &a, hipStream_t st);

struct FinishArgs {
  const int *slots;
  int n_slots;
  ModelTable mt;
  long long max_iter;
  double tol;
  int force_max_iter;
  int evict_enabled;  // 0: cals_hip_sweep (iters++ for everybody)
};
hipError_t finish_launch(const FinishArgs &a, hipStream_t st);   // cals.cpp:336-354

// per-sweep status read-back of cals_hip_run: one packed record per in-flight model (registry order)
struct StatusRec {
  int flags, pad;      // pad = slot (header record: flags = line-search "changed" flag, pad = NNLS status)
  long long iters;     // header record: number of records that follow
  double err, fit, old_fit;
  double ls_margin;    // ModelTable::ls_margin
};
// nnls_status (may be null) lands in the header record's `pad`
hipError_t pack_status_launch(const int *slots, int n, const ModelTable &mt, const int *changed,
                              const int *nnls_status, StatusRec *out, hipStream_t st);

// set-up kernels
// X: src_dtype elements (as uploaded), Xp: dst_dtype elements
hipError_t permute_pad_launch(const void *X, int src_dtype, int n_modes, const int *dims, int m_mode,
                              int a_mode, int Mp, int Ap, void *Xp, int dst_dtype, long long S,
                              hipStream_t st);
hipError_t slice_sumsq_launch(const void *X, int src_dtype, long long I, long long cols,
                              double *partial, int n_part, double *ss_out, hipStream_t st);
// move `ncols` columns of an (rows x *) col-major buffer left by `off` columns (compress)
hipError_t move_columns_launch(void *buf, int dtype, long long rows, long long src_col,
                               long long ncols, long long off, hipStream_t st);

// batched column gather / scatter over several buffers (eviction, compress)
#define CALS_MAX_COLBUFS 56
struct ColBuf {
  void *ptr;
  long long rows;
  int words_per_elem;  // 4-byte words per element: 1 (float) or 2 (double, 64-bit masks)
};
struct ColMoveArgs {
  ColBuf buf[CALS_MAX_COLBUFS];
  int n_bufs;
  const int *src, *dst;   // column indices, n_cols each (gather reads src, scatter writes dst)
  int n_cols;
  unsigned *scratch;      // compact copy: buffer b at scratch_off[b] (words), column k at rows * wpe * k
  long long scratch_off[CALS_MAX_COLBUFS];
  int zero_src_bufs;      // gather: zero the source columns of buffers [0, zero_src_bufs)
};
hipError_t gather_columns_launch(const ColMoveArgs &a, hipStream_t st);
hipError_t scatter_columns_launch(const ColMoveArgs &a, hipStream_t st);
hipError_t set_cols_launch(const int *pairs, int n, int *col, hipStream_t st);

// CALS_HIP_VERIFY=1 (debugging): compare two buffers over the elements the in-flight models own (model_kernels.hip)
struct VerifyArgs {
  const int *slots;
  int n_slots;
  ModelTable mt;
  const void *a, *b;   // the kept operand and its recomputation, same layout
  int kind;            // 0: column major, `rows` elements per column (factor-like; T: rows = S * Mp);
                       // 1: Pt[(column block)][rows][CALS_BN]; 3: Gramian store (CALS_GLD x buffer, doubles, |a - b| <= tol)
  long long rows;
  int dtype;           // element type of kinds 0 / 1
  int skip_flagged;    // skip the models the last line-search launch rewrote (mt.flags & 3): their columns are patched
  double tol;
  int *count;          // += number of differing elements
};
hipError_t verify_launch(const VerifyArgs &a, hipStream_t st);
hipError_t verify_zero_launch(const void *buf, long long rows, int dtype, const int *cols, int n, int *count,
                              hipStream_t st);

}  // namespace calship
#endif
