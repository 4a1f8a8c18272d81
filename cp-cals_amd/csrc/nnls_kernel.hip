// Non-negative factor update on gfx950: update::update_factor_non_negative_constrained
// (src/utils/update.cpp:61-176) for one mode and all in-flight models at once.
//
// The reference runs, for every row of a model's factor, an active-set NNLS (fast NNLS of Bro & de
// Jong, warm-started from the passive set the row had in the previous sweep): a handful of
// dposv solves on sub-matrices G[P,P] of the r x r Hadamard product H (r <= 64 here; nnls_huge_kernel at
// the end of the file takes the ranks above).  Rows are
// independent, the solves are tiny and latency bound, and every row follows its own sequence of
// passive sets.  Mapping here: ONE WAVEFRONT PER ROW, lane i = component i.
//   * the active set is a 64-bit mask in scalar registers (ballots), so the whole control flow of
//     the algorithm is wave-uniform: no divergence, every loop count is a scalar;
//   * y, d, w, s live one element per lane; min/max/argmax are wave reductions;
//   * calculate_sp compacts the passive entries to lanes 0..np-1, factors G[P,P] with the unblocked
//     left-looking Cholesky (lane = row; the strict lower triangle goes to a per-wave LDS tile, each
//     lane reads its own row and a broadcast of row j), carries the forward substitution along
//     with the factorisation (z_j is broadcast with v_readlane as soon as column j exists) and runs
//     the back substitution column by column;
//   * H is formed once per workgroup in LDS (hadamard_but_one, src/utils/utils.cpp:161-172).
// The kernel overwrites the MTTKRP result in the multi-factor with the constrained solution and
// leaves <x_row, g_row> per row for the error formula (compute_fast_error's third term needs G);
// set_jk_fiber, normalize, update_gramian and the error stay in update_kernel, which skips its
// Cholesky and triangular solves when it is handed that buffer (UpdateArgs::rowdot).
//
// Arithmetic: operation order of the Cholesky, the forward substitution and w = y - G d follows the
// Netlib algorithms behind dposv/dgemv; the back substitution is column-oriented (the sums run
// from the last passive entry down) and every division by a diagonal entry of the factor is a multiplication
// by its reciprocal (one division per column of a factorisation, none in the substitutions), i.e. equal to the
// reference up to rounding.  The NNLS
// minimiser of a row is unique for an SPD H, so the active-set path may differ on exact ties without
// changing the result beyond the tolerance the algorithm itself uses.
#include "cals_hip_internal.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace calship {

// set exchanges per row: the reference's loops are unbounded and its exchange rule cycles on some inputs (it
// would not return); max(64, 16 r) is far above what a terminating row needs, and a cycling row no longer holds
// the launch for 4096 solves (an all-positive noise tensor at C3's shape: 3 such rows per launch were 1.0 of
// the kernel's 1.35 ms).  Reported as status bit 2.  Same rule as the oracle's OR_NNLS_MAX_EXCHANGES.
#define NNLS_MAX_EXCHANGES(r) ((r) > 4 ? 16 * (r) : 64)

#define WAVE_SYNC()                                           \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

namespace {

__device__ __forceinline__ double bcast(double v, int l) {  // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double first_lane(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return first_lane(v);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return first_lane(v);
}
__device__ __forceinline__ double wave_add(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Offset of row p of a strict lower triangle PACKED by rows (row p has p entries).  Packed, a rank-16 tile is 120
// doubles instead of 16 x 17: the tiles are what bounds the workgroups per CU (3 -> 6 at C3's shape), and the kernel
// is a chain of dependent steps that only more rows in flight hide (measured with padded LDS: 1 / 2 / 3 workgroups
// per CU = 1.18 / 0.68 / 0.53 ms of NNLS per sweep; packed 0.45).  Row p of lane p starts at a triangular number: 32
// consecutive ones hit 32 different 8-byte banks but for tri(0) = tri(1) (row 0 is empty).
__device__ __forceinline__ int tri(int p) { return (p * (p - 1)) >> 1; }
__host__ __device__ inline size_t tri_tile(int rmax) { return ((size_t)(rmax * (rmax - 1) / 2) + 1) & ~(size_t)1; }

struct WaveScratch {
  unsigned n_solves = 0, n_factor = 0;  // CALS_DIAG statistics
  unsigned long long cached;  // passive set whose factor is in Lw / dg (0: none)
  double dg;                  // lane p: 1 / L[p][p] of that factor (the substitutions multiply: an IEEE f64
                              // division is ~25 instructions, and this kernel is issue bound)
  double *Lw;   // strict lower triangle of the Cholesky factor of G[P,P], packed: row p at Lw + tri(p)
  int *idx;     // idx[p]: component of the p-th passive entry
  double *cv;   // compacted right-hand side
  double *xs;   // solution scattered back to component order
  // the factor of the FULL set (every constraint passive), computed once per workgroup: a row of a model fitted to
  // non-negative data starts all-passive in every sweep (y > 0 for every component), while its second solve -- about
  // every other row drops a component -- used to evict that factor from the wave's one-entry cache, so the next row
  // factored the full set again (tools/nnls_counts.py: 1.0 factorisations per row for 1.5 solves)
  const double *Lf;   // strict lower triangle, row p at Lf + tri(p)
  const double *dgf;  // dgf[p] = 1 / L[p][p]
  unsigned long long fmask;  // the full set (0: not available)
};

// calculate_sp (update.cpp:18-48): x[i] = (G[P,P]^-1 y[P])[i] for i in P, 0 elsewhere.
// pas: passive set (wave-uniform), np = popcount(pas) >= 1.  false: the Cholesky failed (CholFail).
// The factor of the last passive set stays in the wave's tile: consecutive rows of a wave (and the
// first solve of a row after the previous row's last) very often share it -- all-passive rows of a
// model with positive factors all do -- and then only the two substitutions run.
__device__ bool solve_passive(const double *Hs, int r, WaveScratch &ws, unsigned long long pas,
                              int np, double y, int lane, double &x) {
  const bool mine = (pas >> lane) & 1ull;
  if (mine) {
    const int pos = __popcll(pas & ((1ull << lane) - 1ull));
    ws.idx[pos] = lane;
    ws.cv[pos] = y;
  }
  WAVE_SYNC();
  const int p = lane;
  const bool valid = p < np;
  const int myi = valid ? ws.idx[p] : 0;
  double t = valid ? ws.cv[p] : 0.0;
  double dg = 1.0;
  double *Lw = ws.Lw;
  const int prow = tri(p);
  ws.n_solves++;
  const bool full = ws.fmask && pas == ws.fmask;
  const double *Ls = full ? ws.Lf : Lw;  // the factor the substitutions read
  if (full || pas == ws.cached) {
    dg = full ? ws.dgf[p] : ws.dg;
    for (int j = 0; j < np; ++j) {  // L z = b with the cached factor, same operation order
      const double zj = bcast(t * dg, j);  // lane j's product = bcast(t, j) * bcast(dg, j), one broadcast instead of two
      if (p == j)
        t = zj;
      else if (valid && p > j)
        t -= Ls[prow + j] * zj;
    }
  } else {
  ws.cached = 0;
  ws.n_factor++;
  for (int j = 0; j < np; ++j) {
    const int ij = __builtin_amdgcn_readfirstlane(ws.idx[j]);
    const bool below = valid && p > j;
    const int lrow = below ? p : j;  // the other lanes redo row j (unused)
    double ajj = Hs[ij + r * ij];
    double sv = Hs[(below ? myi : ij) + r * ij];
    const double *rj = Lw + tri(j), *ri = Lw + tri(lrow);
    int k = 0;
    for (; k + 4 <= j; k += 4) {  // four load pairs in flight; the subtractions stay in k order
      // (row j could also come from lane j by v_readlane instead of a broadcast LDS load: measured slower,
      // 1.25 vs 1.10 ms of update stage per sweep at C3's shape -- the kernel is issue bound, not LDS bound)
      const double l0 = rj[k], l1 = rj[k + 1], l2 = rj[k + 2], l3 = rj[k + 3];
      const double m0 = ri[k], m1 = ri[k + 1], m2 = ri[k + 2], m3 = ri[k + 3];
      ajj -= l0 * l0;
      ajj -= l1 * l1;
      ajj -= l2 * l2;
      ajj -= l3 * l3;
      sv -= m0 * l0;
      sv -= m1 * l1;
      sv -= m2 * l2;
      sv -= m3 * l3;
    }
    for (; k < j; ++k) {
      const double ljk = rj[k];
      ajj -= ljk * ljk;
      sv -= ri[k] * ljk;
    }
    ajj = first_lane(ajj);
    if (!(ajj > 0.0)) return false;
    // 1 / sqrt(a_jj): v_rsq_f64 + two Newton steps (as update_kernel's register Cholesky; an IEEE sqrt followed by
    // an IEEE divide is ~55 instructions per column).  Entries differ from sqrt / divide by an ulp or two.
    double rl = __builtin_amdgcn_rsq(ajj);
    rl = rl * fma(-0.5 * ajj * rl, rl, 1.5);
    rl = rl * fma(-0.5 * ajj * rl, rl, 1.5);
    const double lij = sv * rl;
    if (below) Lw[prow + j] = lij;
    const double zj = bcast(t, j) * rl;  // forward substitution, column by column
    if (p == j) {
      t = zj;
      dg = rl;
    } else if (below) {
      t -= lij * zj;
    }
    WAVE_SYNC();
  }
  ws.cached = pas;
  ws.dg = dg;
  }
  for (int j = np - 1; j >= 0; --j) {  // L^T x = z
    const double xj = bcast(t * dg, j);
    if (p == j)
      t = xj;
    else if (valid && p < j)
      t -= Ls[tri(j) + p] * xj;
  }
  if (valid) ws.xs[myi] = t;
  WAVE_SYNC();
  x = mine ? ws.xs[lane] : 0.0;
  WAVE_SYNC();  // the next solve rewrites idx/cv/xs
  return true;
}

// calculate_lagrangian_multipliers (update.cpp:50-56): w = y - G d, dgemv 'N' order
__device__ __forceinline__ double multipliers(const double *Hs, int r, double y, double d, int lane) {
  double acc = 0.0;
  const int i = lane < r ? lane : 0;
  for (int j = 0; j < r; ++j) acc += Hs[i + r * j] * bcast(d, j);
  return y - acc;
}

}  // namespace

extern __shared__ __attribute__((aligned(16))) unsigned char nnls_dyn[];

template <typename T>
__global__ void __launch_bounds__(256) nnls_kernel(const NnlsArgs a) {
  const int k_pos = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int k_model = a.idx ? a.idx[k_pos] : k_pos;  // registry position (a class launch lists its own models)
  const int slot = a.slots[k_model];
  const int r = a.mt.rank[slot], col = a.mt.col[slot];
  if (r < a.rlo || r > a.rhi) return;  // another launch's rank class (nnls_launch)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = blockDim.x >> 6;
  const int I = a.I;

  double *Hs = reinterpret_cast<double *>(nnls_dyn);  // r x r, ld = r
  WaveScratch ws;
  ws.cached = 0;
  ws.dg = 1.0;
  {
    const size_t tile = tri_tile(a.rmax);
    const size_t per_wave = tile + 64 + 64 + 32;  // doubles (idx: 64 ints)
    double *shared_f = Hs + (size_t)a.rmax * a.rmax;  // the workgroup's full-set factor + its diagonal
    double *base = shared_f + tile + 64 + per_wave * wave;
    ws.Lf = shared_f;
    ws.dgf = shared_f + tile;
    ws.fmask = 0;
    ws.Lw = base;
    ws.cv = base + tile;
    ws.xs = ws.cv + 64;
    ws.idx = reinterpret_cast<int *>(ws.xs + 64);
  }
  // H = hadamard of the other modes' Gramians (hadamard_but_one)
  for (int e = tid; e < r * r; e += blockDim.x) {
    const int i = e % r, j = e / r;
    double h = 1.0;
    for (int m = 0; m < a.n_modes; ++m)
      if (m != a.mode) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
    Hs[i + r * j] = h;
  }
  __syncthreads();
  // tol = 10 eps ||H||_1 n (update.cpp:65-66; Matrix::one_norm = max column sum of |.|)
  double tol;
  {
    double cs = -DBL_MAX;
    if (lane < r) {
      cs = 0.0;
      for (int i = 0; i < r; ++i) cs += fabs(Hs[i + r * lane]);
    }
    tol = 10 * 2.2204e-16 * wave_max(cs) * (double)r;
  }
  const unsigned long long rmask = (r >= 64) ? ~0ull : ((1ull << r) - 1ull);
  {  // wave 0 factors the full set into the shared tile (same routine, a zero right-hand side)
    __shared__ int s_full_ok;
    if (wave == 0) {
      WaveScratch wf = ws;
      wf.Lw = const_cast<double *>(ws.Lf);
      double unused;
      const bool ok = solve_passive(Hs, r, wf, rmask, r, 0.0, lane, unused);
      if (ok) const_cast<double *>(ws.dgf)[lane] = wf.dg;
      if (lane == 0) s_full_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (s_full_ok) ws.fmask = rmask;
  }
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  unsigned long long *actp = a.act + (long long)I * col;
  double *rowdot = a.rowdot + (long long)I * k_model;
  const int rows_per = (I + a.chunks - 1) / a.chunks;
  const int row0 = chunk * rows_per, row1 = min(I, row0 + rows_per);
  int status = 0;
  unsigned n_rows = 0, n_main = 0, n_inner = 0, n_full = 0;  // CALS_DIAG statistics

  for (int row = row0 + wave; row < row1; row += W) {
    n_rows++;
    const bool in = lane < r;
    const double y = in ? (double)fac[row + (long long)I * lane] : 0.0;
    unsigned long long act = uniform64(actp[row]) & rmask;
    act &= ~__ballot(in && y > 0.0);  // "determine previous active set" (update.cpp:87-91)
    double d = 0.0, sp = 0.0;
    int budget = NNLS_MAX_EXCHANGES(r);
    unsigned long long pas = ~act & rmask;
    if (pas == rmask) n_full++;
    if (pas) {  // warm start (update.cpp:93-121)
      bool failed = !solve_passive(Hs, r, ws, pas, __popcll(pas), y, lane, sp);
      if (!failed) {
        d = sp;
        for (;;) {
          const bool ip = (pas >> lane) & 1ull;
          if (!(wave_min(ip ? sp : DBL_MAX) <= tol)) break;
          const bool z = in && d <= tol;
          if (z) d = 0.0;
          act |= __ballot(z);
          pas = ~act & rmask;
          if (!pas) {  // ZeroPassiveSet
            failed = true;
            break;
          }
          if (!solve_passive(Hs, r, ws, pas, __popcll(pas), y, lane, sp)) {
            failed = true;
            break;
          }
          d = sp;
          if (--budget <= 0) {
            status |= 2;
            break;
          }
        }
      }
      if (failed) {  // the catch block: restart from the all-active set
        act = rmask;
        d = 0.0;
      }
    }
    double w = multipliers(Hs, r, y, d, lane);
    for (;;) {  // main loop (update.cpp:126-167)
      if (!act || budget <= 0) break;
      const bool ia = (act >> lane) & 1ull;
      const double wmax = wave_max(ia ? w : -DBL_MAX);
      if (!(wmax > tol)) break;
      n_main++;
      const unsigned long long hit = __ballot(ia && w == wmax);
      const int m = __ffsll((long long)hit) - 1;  // Tensor::max_id: the first of equal maxima
      const unsigned long long act_top = act;  // the set this pass starts from (cycle test below)
      act &= ~(1ull << m);
      pas = ~act & rmask;
      if (!solve_passive(Hs, r, ws, pas, __popcll(pas), y, lane, sp)) {
        status |= 1;  // an uncaught CholFail ends the reference; reported, row left as it is
        break;
      }
      bool stop = false;
      for (;;) {  // inner loop (update.cpp:136-157)
        const bool ip = (pas >> lane) & 1ull;
        if (!(wave_min(ip ? sp : DBL_MAX) <= tol)) break;
        n_inner++;
        const double alpha = wave_min((ip && sp <= tol) ? d / (d - sp) : DBL_MAX);
        if (in) d = d + alpha * (sp - d);
        const bool na = ip && fabs(d) < tol;
        if (na) d = 0.0;
        act |= __ballot(na);
        pas = ~act & rmask;
        if (!pas) {  // every passive entry left at once: the reference would test a stale value
          status |= 2;
          stop = true;
          break;
        }
        if (!solve_passive(Hs, r, ws, pas, __popcll(pas), y, lane, sp)) {
          status |= 1;
          stop = true;
          break;
        }
        if (--budget <= 0) {
          status |= 2;
          break;
        }
      }
      if (stop) break;
      d = sp;
      w = multipliers(Hs, r, y, d, lane);
      // a pass that ends on the set it started from has reproduced its own starting state (d and w are
      // functions of the set): every further pass repeats it -- the reference's loop never ends here.  Stop at
      // the first such pass, report it like the bound (same rule in the oracle).
      if (act == act_top) {
        status |= 2;
        break;
      }
      if (--budget <= 0) {
        status |= 2;
        break;
      }
    }
    if (in) fac[row + (long long)I * lane] = (T)d;
    const double dot = wave_add(in ? d * y : 0.0);
    if (lane == 0) {
      actp[row] = act;
      rowdot[row] = dot;
    }
  }
  if (status && lane == 0) atomicOr(a.status, status);
  (void)n_rows;
  (void)n_main;
  (void)n_inner;
  (void)n_full;
#ifdef CALS_DIAG
  if (a.dbg_counts && lane == 0) {
    atomicAdd(a.dbg_counts + 0, (unsigned long long)n_rows);
    atomicAdd(a.dbg_counts + 1, (unsigned long long)ws.n_solves);
    atomicAdd(a.dbg_counts + 2, (unsigned long long)ws.n_factor);
    atomicAdd(a.dbg_counts + 3, (unsigned long long)n_main);
    atomicAdd(a.dbg_counts + 4, (unsigned long long)n_inner);
    atomicAdd(a.dbg_counts + 5, (unsigned long long)n_full);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Ranks <= 32: SEVERAL ROWS PER WAVEFRONT (round 3): two for ranks 17..32 (GS = 32 lanes per row), four for ranks <= 16
// (GS = 16).  With one row per wave a rank-20 row keeps 20 of 64 lanes busy and the kernel is issue bound.  Here every
// group of GS lanes runs the algorithm above on a row of its own ("group" g = lane / GS, component l = lane % GS): what
// was wave-uniform -- the active / passive sets, loop counts, every
// decision -- is group-uniform and lives in VGPRs; the two groups execute in lockstep wherever their rows take the
// same path (both all-passive with a cached factor: the common case on non-negative data) and under the hardware's
// execution mask where they do not.  Measured at C3's shape (tools/nnls_bench.py): update stage 0.857 -> 0.704 ms per
// sweep, 208.6 -> 216.8 it/s -- far from 2x: a row is a chain of dependent steps (broadcast, LDS read, FMA), and with
// two tiles per wave fewer workgroups fit a CU, so the rows in flight per CU only grow from 24 to 32.  Same operations in the same order per row as nnls_kernel, hence the same
// results bit for bit; reductions run over GS lanes (xor offsets GS/2..1 never leave a group), a broadcast from
// component j goes through the LDS crossbar (ds_bpermute, source lane GS g + j).
// ---------------------------------------------------------------------------------------------------
namespace {

template <int GS>
__device__ __forceinline__ double gbcast(double v, int j, int g) {  // j uniform over the active lanes
  // (GS = 32 also ran as four v_readlane + two selects: 0.736 ms of update stage per sweep at C3's shape against 0.704)
  const int src = (g * GS + j) << 2;
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(src, __double2loint(v)));
}
template <int GS>
__device__ __forceinline__ double gmin(double v) {
#pragma unroll
  for (int off = GS / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return v;
}
template <int GS>
__device__ __forceinline__ double gmax(double v) {
#pragma unroll
  for (int off = GS / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}
template <int GS>
__device__ __forceinline__ double gadd(double v) {
#pragma unroll
  for (int off = GS / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <int GS>
__device__ __forceinline__ unsigned gballot(bool p, int g) {
  const unsigned long long b = __ballot(p);
  return (unsigned)(b >> (GS * g)) & (GS == 32 ? ~0u : ((1u << GS) - 1u));
}

// the larger of the two groups' n (a group is wholly active or wholly idle): a loop counter that STARTS from a
// group-dependent value must be made wave-uniform, because v_readlane takes its lane from a scalar
template <int GS>
__device__ __forceinline__ int gmax_int(int n) {
  const unsigned long long b = __ballot(true);
  int m = 0;
#pragma unroll
  for (int k = 0; k < 64 / GS; ++k) {
    const int nk = ((b >> (GS * k)) & 1ull) ? __builtin_amdgcn_readlane(n, GS * k) : 0;
    m = nk > m ? nk : m;
  }
  return m;
}

struct GroupScratch {
  unsigned cached;  // passive set whose factor is in Lw / dg (0: none)
  double dg;        // lane p of the group: 1 / L[p][p] of that factor
  double *Lw;       // the group's tile: the strict lower triangle PACKED by rows, row p (p entries) at Lw + tri(p)
  int *idx;
  double *cv, *xs;
  const double *Lf, *dgf;  // the workgroup's factor of the full set
  unsigned fmask;
};

// solve_passive for one group (l = component / compacted position, g = group).  pas, np group-uniform.
template <int GS>
__device__ bool gsolve(const double *Hs, int r, GroupScratch &gs, unsigned pas, int np, double y, int l, int g, double &x) {
  const bool mine = (pas >> l) & 1u;
  if (mine) {
    const int pos = __popc(pas & ((1u << l) - 1u));
    gs.idx[pos] = l;
    gs.cv[pos] = y;
  }
  WAVE_SYNC();
  const int p = l;
  const bool valid = p < np;
  const int myi = valid ? gs.idx[p] : 0;
  double t = valid ? gs.cv[p] : 0.0;
  double dg = 1.0;
  double *Lw = gs.Lw;
  const int prow = tri(p);
  const bool full = gs.fmask && pas == gs.fmask;
  const double *Ls = full ? gs.Lf : Lw;
  bool ok = true;
  if (full || pas == gs.cached) {
    dg = full ? gs.dgf[p] : gs.dg;
    for (int j = 0; j < np; ++j) {  // L z = b with the cached factor
      const double zj = gbcast<GS>(t * dg, j, g);
      if (p == j)
        t = zj;
      else if (valid && p > j)
        t -= Ls[prow + j] * zj;
    }
  } else {
    gs.cached = 0;
    for (int j = 0; j < np; ++j) {
      const int ij = gs.idx[j];  // the same address for every lane of the group
      const bool below = valid && p > j;
      const int lrow = below ? p : j;
      double ajj = Hs[ij + r * ij];
      double sv = Hs[(below ? myi : ij) + r * ij];
      const double *rj = Lw + tri(j), *ri = Lw + tri(lrow);
      int k = 0;
      for (; k + 4 <= j; k += 4) {
        const double l0 = rj[k], l1 = rj[k + 1], l2 = rj[k + 2], l3 = rj[k + 3];
        const double m0 = ri[k], m1 = ri[k + 1], m2 = ri[k + 2], m3 = ri[k + 3];
        ajj -= l0 * l0;
        ajj -= l1 * l1;
        ajj -= l2 * l2;
        ajj -= l3 * l3;
        sv -= m0 * l0;
        sv -= m1 * l1;
        sv -= m2 * l2;
        sv -= m3 * l3;
      }
      for (; k < j; ++k) {
        const double ljk = rj[k];
        ajj -= ljk * ljk;
        sv -= ri[k] * ljk;
      }
      if (!(ajj > 0.0)) {  // group-uniform: every lane of the group computed the same a_jj
        ok = false;
        break;
      }
      double rl = __builtin_amdgcn_rsq(ajj);
      rl = rl * fma(-0.5 * ajj * rl, rl, 1.5);
      rl = rl * fma(-0.5 * ajj * rl, rl, 1.5);
      const double lij = sv * rl;
      if (below) Lw[prow + j] = lij;
      const double zj = gbcast<GS>(t, j, g) * rl;
      if (p == j) {
        t = zj;
        dg = rl;
      } else if (below) {
        t -= lij * zj;
      }
      WAVE_SYNC();
    }
    if (ok) {
      gs.cached = pas;
      gs.dg = dg;
    }
  }
  if (ok) {
    // L^T x = z.  The column counter runs down from the LARGER passive set of the two groups (the ascending loops
    // above start at 0 together; this one would start at two different columns and v_readlane reads one lane)
    for (int j = gmax_int<GS>(np) - 1; j >= 0; --j) {
      if (j < np) {
        const double xj = gbcast<GS>(t * dg, j, g);
        if (p == j)
          t = xj;
        else if (valid && p < j)
          t -= Ls[tri(j) + p] * xj;
      }
    }
    if (valid) gs.xs[myi] = t;
  }
  WAVE_SYNC();
  x = (ok && mine) ? gs.xs[l] : 0.0;
  WAVE_SYNC();
  return ok;
}

template <int GS>
__device__ __forceinline__ double gmultipliers(const double *Hs, int r, double y, double d, int l, int g) {
  double acc = 0.0;
  const int i = l < r ? l : 0;
  for (int j = 0; j < r; ++j) acc += Hs[i + r * j] * gbcast<GS>(d, j, g);
  return y - acc;
}

}  // namespace

// One rank class's share of the merged launch (nnls2m_kernel): block `blk` of the class's n_models * chunks, its
// models listed in idx (registry positions), LDS tiles sized for rmax.
template <typename T, int GS>
__device__ __forceinline__ void nnls2_body(const NnlsArgs &a, int blk, int rmax_c, int chunks_c, const int *idx) {
  constexpr int NG = 64 / GS;  // rows per wavefront
  const int k_pos = blk / chunks_c, chunk = blk % chunks_c;
  const int k_model = idx[k_pos];
  const int slot = a.slots[k_model];
  const int r = a.mt.rank[slot], col = a.mt.col[slot];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / GS, l = lane % GS;
  const int W = blockDim.x >> 6;
  const int I = a.I;

  double *Hs = reinterpret_cast<double *>(nnls_dyn);  // r x r, ld = r
  GroupScratch gs;
  gs.cached = 0;
  gs.dg = 1.0;
  {
    const size_t tile = tri_tile(rmax_c);
    const size_t per_group = tile + GS + GS + GS / 2;  // doubles (idx: GS ints)
    double *shared_f = Hs + (size_t)rmax_c * rmax_c;
    double *base = shared_f + tile + 32 + per_group * (size_t)(NG * wave + g);
    gs.Lf = shared_f;
    gs.dgf = shared_f + tile;
    gs.fmask = 0;
    gs.Lw = base;
    gs.cv = base + tile;
    gs.xs = gs.cv + GS;
    gs.idx = reinterpret_cast<int *>(gs.xs + GS);
  }
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  unsigned long long *actp = a.act + (long long)I * col;
  double *rowdot = a.rowdot + (long long)I * k_model;
  const int rows_per = (I + chunks_c - 1) / chunks_c;
  const int row0 = chunk * rows_per, row1 = min(I, row0 + rows_per);
  // the row's g and its stored active set are loaded ONE ROW AHEAD (the first one here, in front of the workgroup's
  // set-up): the round trip runs under the previous row's solves instead of in front of this row's first one
#ifdef CALS_NNLS2_SINGLE
  const int row_first = row0 + wave, row_step = W;
#else
  const int row_first = row0 + NG * wave + g, row_step = NG * W;
#endif
  double y_next = 0.0;
  unsigned long long act_next = 0;
  if (row_first < row1) {
    y_next = (l < r) ? (double)fac[row_first + (long long)I * l] : 0.0;
    act_next = actp[row_first];
  }
  for (int e = tid; e < r * r; e += blockDim.x) {
    const int i = e % r, j = e / r;
    double h = 1.0;
    for (int m = 0; m < a.n_modes; ++m)
      if (m != a.mode) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
    Hs[i + r * j] = h;
  }
  __syncthreads();
  double tol;
  {
    double cs = -DBL_MAX;
    if (l < r) {
      cs = 0.0;
      for (int i = 0; i < r; ++i) cs += fabs(Hs[i + r * l]);
    }
    tol = 10 * 2.2204e-16 * gmax<GS>(cs) * (double)r;
  }
  const unsigned rmask = (r >= 32) ? ~0u : ((1u << r) - 1u);  // (r <= GS: nnls_launch)
  {  // group 0 of wave 0 factors the full set into the shared tile
    __shared__ int s_full_ok2;
    if (wave == 0 && g == 0) {
      GroupScratch gf = gs;
      gf.Lw = const_cast<double *>(gs.Lf);
      double unused;
      const bool ok = gsolve<GS>(Hs, r, gf, rmask, r, 0.0, l, 0, unused);
      if (ok) const_cast<double *>(gs.dgf)[l] = gf.dg;
      if (l == 0) s_full_ok2 = ok ? 1 : 0;
    }
    __syncthreads();
    if (s_full_ok2) gs.fmask = rmask;
  }
  int status = 0;

#ifdef CALS_NNLS2_SINGLE  // debugging: group 0 alone, one row per wave
  for (int rb = row0 + wave; rb < row1; rb += W) {
    const int row = rb;
    if (g == 0) {
#else
  for (int rb = row0 + NG * wave; rb < row1; rb += NG * W) {
    const int row = rb + g;
    if (row < row1) {  // (a chunk's tail leaves the last groups idle)
#endif
      const bool in = l < r;
      const double y = y_next;
      unsigned act = (unsigned)act_next & rmask;
      if (row + row_step < row1) {
        y_next = in ? (double)fac[row + row_step + (long long)I * l] : 0.0;
        act_next = actp[row + row_step];
      }
      act &= ~gballot<GS>(in && y > 0.0, g);
      double d = 0.0, sp = 0.0;
      int budget = NNLS_MAX_EXCHANGES(r);
      unsigned pas = ~act & rmask;
      if (pas) {  // warm start (update.cpp:93-121)
        bool failed = !gsolve<GS>(Hs, r, gs, pas, __popc(pas), y, l, g, sp);
        if (!failed) {
          d = sp;
          for (;;) {
            const bool ip = (pas >> l) & 1u;
            if (!(gmin<GS>(ip ? sp : DBL_MAX) <= tol)) break;
            const bool z = in && d <= tol;
            if (z) d = 0.0;
            act |= gballot<GS>(z, g);
            pas = ~act & rmask;
            if (!pas) {  // ZeroPassiveSet
              failed = true;
              break;
            }
            if (!gsolve<GS>(Hs, r, gs, pas, __popc(pas), y, l, g, sp)) {
              failed = true;
              break;
            }
            d = sp;
            if (--budget <= 0) {
              status |= 2;
              break;
            }
          }
        }
        if (failed) {
          act = rmask;
          d = 0.0;
        }
      }
      double w = gmultipliers<GS>(Hs, r, y, d, l, g);
      for (;;) {  // main loop (update.cpp:126-167)
        if (!act || budget <= 0) break;
        const bool ia = (act >> l) & 1u;
        const double wmax = gmax<GS>(ia ? w : -DBL_MAX);
        if (!(wmax > tol)) break;
        const unsigned hit = gballot<GS>(ia && w == wmax, g);
        const int m = __ffs((int)hit) - 1;  // Tensor::max_id: the first of equal maxima
        const unsigned act_top = act;
        act &= ~(1u << m);
        pas = ~act & rmask;
        if (!gsolve<GS>(Hs, r, gs, pas, __popc(pas), y, l, g, sp)) {
          status |= 1;
          break;
        }
        bool stop = false;
        for (;;) {  // inner loop (update.cpp:136-157)
          const bool ip = (pas >> l) & 1u;
          if (!(gmin<GS>(ip ? sp : DBL_MAX) <= tol)) break;
          const double alpha = gmin<GS>((ip && sp <= tol) ? d / (d - sp) : DBL_MAX);
          if (in) d = d + alpha * (sp - d);
          const bool na = ip && fabs(d) < tol;
          if (na) d = 0.0;
          act |= gballot<GS>(na, g);
          pas = ~act & rmask;
          if (!pas) {
            status |= 2;
            stop = true;
            break;
          }
          if (!gsolve<GS>(Hs, r, gs, pas, __popc(pas), y, l, g, sp)) {
            status |= 1;
            stop = true;
            break;
          }
          if (--budget <= 0) {
            status |= 2;
            break;
          }
        }
        if (stop) break;
        d = sp;
        w = gmultipliers<GS>(Hs, r, y, d, l, g);
        if (act == act_top) {  // the pass reproduced its own starting state: the reference's loop never ends here
          status |= 2;
          break;
        }
        if (--budget <= 0) {
          status |= 2;
          break;
        }
      }
      if (in) fac[row + (long long)I * l] = (T)d;
      const double dot = gadd<GS>(in ? d * y : 0.0);
      if (l == 0) {
        actp[row] = (unsigned long long)act;
        rowdot[row] = dot;
      }
    }
  }
  if (status && l == 0) atomicOr(a.status, status);
}

// Ranks <= 32 in ONE launch: the classes <= 16 (four rows per wave), <= 24 and <= 32 (two rows, LDS tiles of their own
// size) as consecutive block ranges (NnlsArgs::seg_*).  As separate launches on the one stream the classes ran one
// after the other -- 98 + 94 us per mode at C3's shape for 205 + 51 models -- although each of them alone is bound by
// the latency of its rows' dependent steps, not by the chip.
template <typename T>
__global__ void __launch_bounds__(256) nnls2m_kernel(const NnlsArgs a) {
  const int b = blockIdx.x;
  if (b < a.seg_first[1])
    nnls2_body<T, 16>(a, b, a.seg_rmax[0], a.seg_chunks[0], a.cls_idx + a.cls_off[0]);
  else if (b < a.seg_first[2])
    nnls2_body<T, 32>(a, b - a.seg_first[1], a.seg_rmax[1], a.seg_chunks[1], a.cls_idx + a.cls_off[1]);
  else
    nnls2_body<T, 32>(a, b - a.seg_first[2], a.seg_rmax[2], a.seg_chunks[2], a.cls_idx + a.cls_off[2]);
}

size_t nnls2_lds_bytes(int rmax, int waves, int gs) {
  const size_t tile = tri_tile(rmax);
  const size_t per_group = tile + gs + gs + gs / 2;
  return ((size_t)rmax * rmax + tile + 32 + per_group * (64 / gs) * waves) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------------
// Ranks 65..CALS_GLD.  Same algorithm, same operation order, one wavefront per row -- but a row has up to
// four components per lane (component c = lane + 64 q), the active set is NNLS_HQ 64-bit words (word q of
// (row, model) at act[row + I * (col + q)]: a model owns r >= 64 q columns of that buffer), and neither H
// nor a Cholesky factor of up to 256 x 256 fits LDS: H (one copy per workgroup) and each wave's factor
// live in a global scratch block (L2 resident), the factor twice -- column-contiguous (Lt) for the
// factorisation and the forward substitution, row-contiguous (Lr) for the back substitution -- so that
// every load a wave issues is one coalesced line per 8 lanes.  The compacted right-hand side, the diagonal,
// d and the index map stay in LDS.  Built to be right, not fast (as update_body_huge): the reference's
// typical ranks are <= 20.
#define NNLS_HQ (CALS_GLD / 64)
#define NNLS_HWAVES 4

namespace {

typedef unsigned long long u64;
typedef double v2d_t __attribute__((ext_vector_type(2)));

struct HugeWave {
  double *Lt;  // global: L[p][k] at Lt[k * CALS_GLD + p]
  double *Lr;  // global: L[p][k] at Lr[p * CALS_GLD + k]
  double *cv, *xs, *dgs, *dv;  // LDS, CALS_GLD each
  int *idx;                    // LDS
  u64 cached[NNLS_HQ];
  bool has_cache;
};

__device__ __forceinline__ bool any_bits(const u64 (&m)[NNLS_HQ]) {
  u64 o = 0;
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) o |= m[q];
  return o != 0;
}
__device__ __forceinline__ int count_bits(const u64 (&m)[NNLS_HQ]) {
  int n = 0;
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) n += __popcll(m[q]);
  return n;
}

// calculate_sp for the passive set pas (np = its size >= 1); see solve_passive
__device__ bool solve_passive_huge(const double *Hs, int r, HugeWave &ws, const u64 (&pas)[NNLS_HQ], int np,
                                   const double (&y)[NNLS_HQ], int lane, double (&x)[NNLS_HQ]) {
  constexpr int LD = CALS_GLD;
  const u64 below_me = (1ull << lane) - 1ull;
  {
    int base = 0;
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q) {
      if ((pas[q] >> lane) & 1ull) {
        const int pos = base + __popcll(pas[q] & below_me);
        ws.idx[pos] = lane + 64 * q;
        ws.cv[pos] = y[q];
      }
      base += __popcll(pas[q]);
    }
  }
  WAVE_SYNC();
  const int nq = (np + 63) >> 6;
  bool same = ws.has_cache;
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) same = same && (pas[q] == ws.cached[q]);
  if (same) {
    // L z = b with the cached factor, same operation order; column j + 1 of the factor is fetched while
    // column j is applied (a step then costs two LDS round trips, not an L2 one)
    double lc[NNLS_HQ], ln[NNLS_HQ];
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q) lc[q] = (q < nq) ? ws.Lt[lane + 64 * q] : 0.0;
    for (int j = 0; j < np; ++j) {
      const int jn = (j + 1 < np) ? j + 1 : j;
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) ln[q] = (q < nq) ? ws.Lt[(size_t)jn * LD + lane + 64 * q] : 0.0;
      const double zj = ws.cv[j] / ws.dgs[j];
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) {
        const int p = lane + 64 * q;
        if (q < nq) {
          if (p == j)
            ws.cv[p] = zj;
          else if (p < np && p > j)
            ws.cv[p] -= lc[q] * zj;
        }
      }
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) lc[q] = ln[q];
    }
  } else {
    ws.has_cache = false;
    for (int j = 0; j < np; ++j) {
      const int ij = __builtin_amdgcn_readfirstlane(ws.idx[j]);
      double ajj = Hs[ij + (size_t)r * ij];
      double sv[NNLS_HQ];
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) {
        const int p = lane + 64 * q;
        sv[q] = (q < nq && p < np && p > j) ? Hs[ws.idx[p] + (size_t)r * ij] : 0.0;
      }
      // the subtractions stay in k order; rows outside (j, np) compute unused values.  Row j of the factor
      // comes from the row-contiguous copy (16-byte broadcast loads); the loads of eight k go out before
      // their FMAs: one exposed L2 round trip per eight columns instead of one per column.
      const double *Lrj = ws.Lr + (size_t)j * LD, *Ltp = ws.Lt + lane;
      int k = 0;
      for (; k + 8 <= j; k += 8) {
        v2d_t pv[4];
        double lp[NNLS_HQ][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = *reinterpret_cast<const v2d_t *>(Lrj + k + 2 * u);
#pragma unroll
        for (int q = 0; q < NNLS_HQ; ++q)
          if (q < nq) {
#pragma unroll
            for (int u = 0; u < 8; ++u) lp[q][u] = Ltp[(size_t)(k + u) * LD + 64 * q];
          }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double ljk = pv[u >> 1][u & 1];
          ajj -= ljk * ljk;
#pragma unroll
          for (int q = 0; q < NNLS_HQ; ++q)
            if (q < nq) sv[q] -= lp[q][u] * ljk;
        }
      }
      for (; k < j; ++k) {
        const double ljk = Lrj[k];
        ajj -= ljk * ljk;
#pragma unroll
        for (int q = 0; q < NNLS_HQ; ++q)
          if (q < nq) sv[q] -= Ltp[(size_t)k * LD + 64 * q] * ljk;
      }
      ajj = first_lane(ajj);
      if (!(ajj > 0.0)) return false;
      const double ljj = sqrt(ajj);
      const double zj = ws.cv[j] / ljj;  // forward substitution, column by column
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) {
        const int p = lane + 64 * q;
        if (q < nq) {
          if (p == j) {
            ws.cv[p] = zj;
            ws.dgs[p] = ljj;
          } else if (p < np && p > j) {
            const double lij = sv[q] / ljj;
            ws.Lt[(size_t)j * LD + p] = lij;
            ws.Lr[(size_t)p * LD + j] = lij;
            ws.cv[p] -= lij * zj;
          }
        }
      }
      __threadfence_block();  // column j of the factor is read by the other lanes from the next step on
      WAVE_SYNC();
    }
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q) ws.cached[q] = pas[q];
    ws.has_cache = true;
  }
  {  // L^T x = z, row j - 1 of the factor fetched while row j is applied
    double lc[NNLS_HQ], ln[NNLS_HQ];
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q) lc[q] = (q < nq) ? ws.Lr[(size_t)(np - 1) * LD + lane + 64 * q] : 0.0;
    for (int j = np - 1; j >= 0; --j) {
      const int jn = j > 0 ? j - 1 : 0;
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) ln[q] = (q < nq) ? ws.Lr[(size_t)jn * LD + lane + 64 * q] : 0.0;
      const double xj = ws.cv[j] / ws.dgs[j];
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) {
        const int p = lane + 64 * q;
        if (q < nq) {
          if (p == j)
            ws.cv[p] = xj;
          else if (p < j)
            ws.cv[p] -= lc[q] * xj;
        }
      }
      WAVE_SYNC();
#pragma unroll
      for (int q = 0; q < NNLS_HQ; ++q) lc[q] = ln[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) {
    const int p = lane + 64 * q;
    if (q < nq && p < np) ws.xs[ws.idx[p]] = ws.cv[p];
  }
  WAVE_SYNC();
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) x[q] = ((pas[q] >> lane) & 1ull) ? ws.xs[lane + 64 * q] : 0.0;
  WAVE_SYNC();  // the next solve rewrites idx / cv / xs
  return true;
}

// w = y - G d, dgemv 'N' order (j ascending)
__device__ __forceinline__ void multipliers_huge(const double *Hs, int r, HugeWave &ws, const double (&y)[NNLS_HQ],
                                                 const double (&d)[NNLS_HQ], int lane, double (&w)[NNLS_HQ]) {
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q)
    if (lane + 64 * q < r) ws.dv[lane + 64 * q] = d[q];
  WAVE_SYNC();
  double acc[NNLS_HQ];
  int row[NNLS_HQ];
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) {
    acc[q] = 0.0;
    row[q] = (lane + 64 * q < r) ? lane + 64 * q : 0;
  }
#pragma unroll 4
  for (int j = 0; j < r; ++j) {
    const double dj = ws.dv[j];
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q)
      if (64 * q < r) acc[q] += Hs[row[q] + (size_t)r * j] * dj;
  }
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) w[q] = y[q] - acc[q];
  WAVE_SYNC();
}

}  // namespace

size_t nnls_huge_block_doubles() {
  return (size_t)CALS_GLD * CALS_GLD * (1 + 2 * NNLS_HWAVES);
}
// Workgroups per model of rank > CALS_RMAX: ONE ROW PER WAVEFRONT where the scratch budget allows it (a row of such
// a model refactors passive blocks of up to 256 x 256 per exchange -- 25 ms of one wave at rank 256 -- and the rows
// of a mode are all the parallelism there is: with 16 workgroups a wave worked through five rows in turn).
// n_huge models share at most 16 GiB of factor scratch (4.7 MB per workgroup).
int nnls_huge_chunks(int I, int n_huge) {
  const size_t block_bytes = nnls_huge_block_doubles() * sizeof(double);
  const size_t by_budget = ((size_t)16 << 30) / (block_bytes * (size_t)std::max(n_huge, 1));
  const int want = (I + NNLS_HWAVES - 1) / NNLS_HWAVES;
  return std::max(1, std::min(want, (int)std::min<size_t>(std::max<size_t>(by_budget, 1), 1024)));
}

template <typename T>
__global__ void __launch_bounds__(64 * NNLS_HWAVES) nnls_huge_kernel(const NnlsArgs a) {
  const int k_pos = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int k_model = a.idx ? a.idx[k_pos] : k_pos;  // registry position (a class launch lists its own models)
  const int slot = a.slots[k_model];
  const int r = a.mt.rank[slot], col = a.mt.col[slot];
  if (r <= CALS_RMAX) return;  // nnls_kernel's share
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int I = a.I;
  __shared__ double s_vec[NNLS_HWAVES][4][CALS_GLD];
  __shared__ int s_idx[NNLS_HWAVES][CALS_GLD];
  __shared__ int s_block;
  if (tid == 0) s_block = atomicAdd(a.hcounter, 1);
  __syncthreads();
  double *Hs = a.hscratch + (size_t)s_block * ((size_t)CALS_GLD * CALS_GLD * (1 + 2 * NNLS_HWAVES));
  HugeWave ws;
  ws.Lt = Hs + (size_t)CALS_GLD * CALS_GLD * (1 + 2 * wave);
  ws.Lr = ws.Lt + (size_t)CALS_GLD * CALS_GLD;
  ws.cv = s_vec[wave][0];
  ws.xs = s_vec[wave][1];
  ws.dgs = s_vec[wave][2];
  ws.dv = s_vec[wave][3];
  ws.idx = s_idx[wave];
  ws.has_cache = false;
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) ws.cached[q] = 0;

  for (int e = tid; e < r * r; e += blockDim.x) {  // hadamard_but_one, ld = r
    const int i = e % r, j = e / r;
    double h = 1.0;
    for (int m = 0; m < a.n_modes; ++m)
      if (m != a.mode) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
    Hs[i + (size_t)r * j] = h;
  }
  __threadfence_block();
  __syncthreads();
  double tol;  // 10 eps ||H||_1 n
  {
    double cs = -DBL_MAX;
#pragma unroll
    for (int q = 0; q < NNLS_HQ; ++q) {
      const int c = lane + 64 * q;
      if (c < r) {
        double s_ = 0.0;
        for (int i = 0; i < r; ++i) s_ += fabs(Hs[i + (size_t)r * c]);
        cs = fmax(cs, s_);
      }
    }
    tol = 10 * 2.2204e-16 * wave_max(cs) * (double)r;
  }
  u64 rmask[NNLS_HQ];
  bool in[NNLS_HQ];
#pragma unroll
  for (int q = 0; q < NNLS_HQ; ++q) {
    const int left = r - 64 * q;
    rmask[q] = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
    in[q] = lane + 64 * q < r;
  }
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  u64 *actp = a.act + (long long)I * col;
  double *rowdot = a.rowdot + (long long)I * k_model;
  const int rows_per = (I + a.chunks - 1) / a.chunks;
  const int row0 = chunk * rows_per, row1 = min(I, row0 + rows_per);
  int status = 0;

#define HQ_FOR _Pragma("unroll") for (int q = 0; q < NNLS_HQ; ++q)
  for (int row = row0 + wave; row < row1; row += NNLS_HWAVES) {
    double y[NNLS_HQ], d[NNLS_HQ], sp[NNLS_HQ], w[NNLS_HQ];
    u64 act[NNLS_HQ], pas[NNLS_HQ];
    HQ_FOR {
      y[q] = in[q] ? (double)fac[row + (long long)I * (lane + 64 * q)] : 0.0;
      d[q] = 0.0;
      sp[q] = 0.0;
      act[q] = rmask[q] ? (uniform64(actp[row + (long long)I * q]) & rmask[q]) : 0ull;
      act[q] &= ~__ballot(in[q] && y[q] > 0.0);  // "determine previous active set"
      pas[q] = ~act[q] & rmask[q];
    }
    int budget = NNLS_MAX_EXCHANGES(r);
    // min over the passive entries of sp / over the given per-lane values
    auto min_passive_sp = [&]() {
      double v = DBL_MAX;
      HQ_FOR if ((pas[q] >> lane) & 1ull) v = fmin(v, sp[q]);
      return wave_min(v);
    };
    if (any_bits(pas)) {  // warm start (update.cpp:93-121)
      bool failed = !solve_passive_huge(Hs, r, ws, pas, count_bits(pas), y, lane, sp);
      if (!failed) {
        HQ_FOR d[q] = sp[q];
        for (;;) {
          if (!(min_passive_sp() <= tol)) break;
          HQ_FOR {
            const bool z = in[q] && d[q] <= tol;
            if (z) d[q] = 0.0;
            act[q] |= __ballot(z);
            pas[q] = ~act[q] & rmask[q];
          }
          if (!any_bits(pas)) {  // ZeroPassiveSet
            failed = true;
            break;
          }
          if (!solve_passive_huge(Hs, r, ws, pas, count_bits(pas), y, lane, sp)) {
            failed = true;
            break;
          }
          HQ_FOR d[q] = sp[q];
          if (--budget <= 0) {
            status |= 2;
            break;
          }
        }
      }
      if (failed) {  // the catch block: restart from the all-active set
        HQ_FOR {
          act[q] = rmask[q];
          d[q] = 0.0;
        }
      }
    }
    multipliers_huge(Hs, r, ws, y, d, lane, w);
    for (;;) {  // main loop (update.cpp:126-167)
      if (!any_bits(act) || budget <= 0) break;
      double wl = -DBL_MAX;
      HQ_FOR if ((act[q] >> lane) & 1ull) wl = fmax(wl, w[q]);
      const double wmax = wave_max(wl);
      if (!(wmax > tol)) break;
      u64 act_top[NNLS_HQ];  // the set this pass starts from (cycle test below)
      HQ_FOR act_top[q] = act[q];
      {  // Tensor::max_id: the first of equal maxima = lowest component = lowest word, then lowest lane
        bool taken = false;
        HQ_FOR {
          const u64 hit = __ballot(((act[q] >> lane) & 1ull) && w[q] == wmax);
          if (!taken && hit) {
            act[q] &= ~(1ull << (__ffsll((long long)hit) - 1));
            taken = true;
          }
        }
      }
      HQ_FOR pas[q] = ~act[q] & rmask[q];
      if (!solve_passive_huge(Hs, r, ws, pas, count_bits(pas), y, lane, sp)) {
        status |= 1;
        break;
      }
      bool stop = false;
      for (;;) {  // inner loop (update.cpp:136-157)
        if (!(min_passive_sp() <= tol)) break;
        double al = DBL_MAX;
        HQ_FOR if (((pas[q] >> lane) & 1ull) && sp[q] <= tol) al = fmin(al, d[q] / (d[q] - sp[q]));
        const double alpha = wave_min(al);
        HQ_FOR {
          if (in[q]) d[q] = d[q] + alpha * (sp[q] - d[q]);
          const bool na = ((pas[q] >> lane) & 1ull) && fabs(d[q]) < tol;
          if (na) d[q] = 0.0;
          act[q] |= __ballot(na);
        }
        HQ_FOR pas[q] = ~act[q] & rmask[q];
        if (!any_bits(pas)) {
          status |= 2;
          stop = true;
          break;
        }
        if (!solve_passive_huge(Hs, r, ws, pas, count_bits(pas), y, lane, sp)) {
          status |= 1;
          stop = true;
          break;
        }
        if (--budget <= 0) {
          status |= 2;
          break;
        }
      }
      if (stop) break;
      HQ_FOR d[q] = sp[q];
      multipliers_huge(Hs, r, ws, y, d, lane, w);
      {  // the pass ended on the set it started from: a cycle of the exchange rule (see nnls_kernel)
        bool same = true;
        HQ_FOR same = same && act[q] == act_top[q];
        if (same) {
          status |= 2;
          break;
        }
      }
      if (--budget <= 0) {
        status |= 2;
        break;
      }
    }
    double dl = 0.0;
    HQ_FOR {
      if (in[q]) {
        fac[row + (long long)I * (lane + 64 * q)] = (T)d[q];
        dl += d[q] * y[q];
      }
    }
    const double dot = wave_add(dl);
    if (lane == 0) {
      HQ_FOR if (rmask[q]) actp[row + (long long)I * q] = act[q];
      rowdot[row] = dot;
    }
  }
#undef HQ_FOR
  if (status && lane == 0) atomicOr(a.status, status);
}

int nnls_rank_class(int r) { return r <= 16 ? 0 : r <= 24 ? 1 : r <= 32 ? 2 : r <= 48 ? 3 : r <= CALS_RMAX ? 4 : 5; }

size_t nnls_lds_bytes(int rmax, int waves) {
  const size_t per_wave = tri_tile(rmax) + 64 + 64 + 32;
  return ((size_t)rmax * rmax + tri_tile(rmax) + 64 + per_wave * waves) * sizeof(double);
}

hipError_t nnls_launch(const NnlsArgs &a_in, hipStream_t st) {
  if (a_in.n_slots <= 0) return hipSuccess;
  NnlsArgs a = a_in;
  const bool huge = a.rmax > CALS_RMAX;
  if (huge && (!a.hscratch || !a.hcounter)) return hipErrorInvalidValue;
  a.rmax = std::min(std::max(a.rmax, 1), CALS_RMAX);
  const size_t budget = (size_t)160 * 1024 - 1024;
  static AttrOnce once[2];
  const int di = (a.dtype == CALS_F32) ? 1 : 0;
  const void *fn = di ? reinterpret_cast<const void *>(&nnls_kernel<float>)
                      : reinterpret_cast<const void *>(&nnls_kernel<double>);
  const hipError_t ea = once[di].ensure(
      [&] { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget); });
  if (ea != hipSuccess) return ea;
  // One launch per rank class in flight (a_in.rank_classes, bit k = class k of nnls_rank_class): the LDS tiles
  // are sized by the class's largest rank, so a single rank-48 model no longer leaves every workgroup of the
  // rank-1..20 models with 99 KB of LDS and one workgroup per CU (measured: +17 ms per sweep at C3's shape).
  // The workgroups of the other classes return at once.
  static const int class_hi[5] = {16, 24, 32, 48, CALS_RMAX};
  static const bool one_row = getenv("CALS_NNLS_ONE_ROW") != nullptr;  // A/B switch: one row per wave everywhere
  static const int forced_chunks = getenv("CALS_NNLS_CHUNKS") ? atoi(getenv("CALS_NNLS_CHUNKS")) : 0;  // experiments
  int k_first = 0;
  if (a_in.rank_classes && a_in.cls_idx && !one_row && (a_in.rank_classes & 7u)) {
    // ---- ranks <= 32: several rows per wavefront, the three classes merged into one launch ----
    const int waves = 4;
    size_t dyn = 0;
    int blocks = 0;
    for (int k = 0; k < 3; ++k) {
      const int n = a_in.cls_off[k + 1] - a_in.cls_off[k];
      a.seg_first[k] = blocks;
      a.seg_rmax[k] = std::min(class_hi[k], std::min(std::max(a_in.rmax, 1), CALS_RMAX));
      a.seg_chunks[k] = 1;
      if (n <= 0) continue;
      int chunks = std::max(1, std::min((a.I + 4 * waves - 1) / (4 * waves), (4096 + n - 1) / n));
      if (forced_chunks > 0) chunks = std::min(forced_chunks, a.I);
      a.seg_chunks[k] = chunks;
      blocks += n * chunks;
      dyn = std::max(dyn, nnls2_lds_bytes(a.seg_rmax[k], waves, k == 0 ? 16 : 32));
    }
    const void *fn2 = di ? reinterpret_cast<const void *>(&nnls2m_kernel<float>)
                         : reinterpret_cast<const void *>(&nnls2m_kernel<double>);
    static AttrOnce once2[2];
    const hipError_t e2 = once2[di].ensure(
        [&] { return hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget); });
    if (e2 != hipSuccess) return e2;
    if (dyn > budget) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(64 * waves);
    if (di)
      hipLaunchKernelGGL(nnls2m_kernel<float>, grid, block, dyn, st, a);
    else
      hipLaunchKernelGGL(nnls2m_kernel<double>, grid, block, dyn, st, a);
    k_first = 3;
  }
  // One launch per remaining rank class in flight (a_in.rank_classes, bit k = class k of nnls_rank_class), one row per
  // wavefront: the LDS tiles are sized by the class's largest rank, so a single rank-48 model does not leave every
  // workgroup of the rank-1..20 models with 99 KB of LDS and one workgroup per CU (measured: +17 ms per sweep at C3's
  // shape).  With class lists (cls_idx) a launch's workgroups are exactly its models; without, the other classes'
  // workgroups return at once.
  for (int k = k_first; k < 5; ++k) {
    int n_models = a_in.n_slots;
    a.idx = nullptr;
    if (a_in.rank_classes) {
      if (!(a_in.rank_classes & (1u << k))) continue;
      a.rlo = k ? class_hi[k - 1] + 1 : 1;
      a.rhi = class_hi[k];
      if (a_in.cls_idx) {
        n_models = a_in.cls_off[k + 1] - a_in.cls_off[k];
        a.idx = a_in.cls_idx + a_in.cls_off[k];
        if (n_models <= 0) continue;
      }
    } else {  // no class information: one launch sized by the largest rank
      if (k) break;
      a.rlo = 1;
      a.rhi = CALS_RMAX;
    }
    a.rmax = std::min(a.rhi, std::min(std::max(a_in.rmax, 1), CALS_RMAX));
    int waves = 4;
    while (waves > 1 && nnls_lds_bytes(a.rmax, waves) > budget) --waves;
    const size_t dyn = nnls_lds_bytes(a.rmax, waves);
    // rows per workgroup: enough workgroups to fill 256 CUs several times over; at least 4 rows per wave -- ONE for
    // ranks 33..64, whose rows are long (a 64 x 64 factorisation per exchange) and whose models are few: one model of
    // rank 64 next to C3's small ones cost +2.3 ms per sweep with 19 workgroups, its 300 rows four to a wave in turn
    const int min_rows = (a.rhi > 32) ? 1 : 4;
    a.n_cls = n_models;
    a.chunks = std::max(1, std::min((a.I + min_rows * waves - 1) / (min_rows * waves), (4096 + n_models - 1) / n_models));
    if (forced_chunks > 0) a.chunks = std::min(forced_chunks, a.I);
    const dim3 grid((unsigned)(n_models * a.chunks)), block(64 * waves);
    if (di)
      hipLaunchKernelGGL(nnls_kernel<float>, grid, block, dyn, st, a);
    else
      hipLaunchKernelGGL(nnls_kernel<double>, grid, block, dyn, st, a);
  }
  if (huge) {  // the models above CALS_RMAX
    int n_models = a_in.n_slots;
    a.idx = nullptr;
    if (a_in.rank_classes && a_in.cls_idx) {
      n_models = a_in.cls_off[6] - a_in.cls_off[5];
      a.idx = a_in.cls_idx + a_in.cls_off[5];
    }
    a.n_cls = n_models;
    a.chunks = nnls_huge_chunks(a.I, std::max(n_models, 1));
    if (a.huge_chunk_cap > 0) a.chunks = std::min(a.chunks, a.huge_chunk_cap);
    const dim3 hgrid((unsigned)(std::max(n_models, 1) * a.chunks)), hblock(64 * NNLS_HWAVES);
    if (di)
      hipLaunchKernelGGL(nnls_huge_kernel<float>, hgrid, hblock, 0, st, a);
    else
      hipLaunchKernelGGL(nnls_huge_kernel<double>, hgrid, hblock, 0, st, a);
  }
  return hipGetLastError();
}

// A model's active sets at admission: every constraint active (include/ktensor.h:69,108).
// desc as for init_slots_launch: n x {slot, col, rank, jk_mode, jk_fiber}
__global__ void __launch_bounds__(256) nnls_reset_kernel(const int *desc, NnlsResetArgs a) {
  const int col = desc[5 * blockIdx.x + 1], r = desc[5 * blockIdx.x + 2];
  for (int q = 0; 64 * q < r; ++q) {  // word q of a row: components 64 q .. 64 q + 63
    const int left = r - 64 * q;
    const unsigned long long rmask = (left >= 64) ? ~0ull : ((1ull << left) - 1ull);
    for (int m = 0; m < a.n_modes; ++m) {
      unsigned long long *p = a.act[m] + (long long)a.I[m] * (col + q);
      for (int i = threadIdx.x; i < a.I[m]; i += 256) p[i] = rmask;
    }
  }
}

hipError_t nnls_reset_launch(const int *desc, int n, const NnlsResetArgs &a, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(nnls_reset_kernel, dim3(n), dim3(256), 0, st, desc, a);
  return hipGetLastError();
}

}  // namespace calship
