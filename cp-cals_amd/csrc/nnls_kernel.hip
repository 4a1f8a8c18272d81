// Non-negative factor update on gfx950: update::update_factor_non_negative_constrained
// (src/utils/update.cpp:61-176) for one mode and all in-flight models at once.
//
// The reference runs, for every row of a model's factor, an active-set NNLS (fast NNLS of Bro & de
// Jong, warm-started from the passive set the row had in the previous sweep): a handful of
// dposv solves on sub-matrices G[P,P] of the r x r Hadamard product H, r <= 64.  Rows are
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
// from the last passive entry down), i.e. equal to the reference up to rounding.  The NNLS
// minimiser of a row is unique for an SPD H, so the active-set path may differ on exact ties without
// changing the result beyond the tolerance the algorithm itself uses.
#include "cals_hip_internal.h"

#include <algorithm>
#include <cfloat>

namespace calship {

#define NNLS_MAX_EXCHANGES 4096  // set exchanges per row; the reference's loops are unbounded (status bit 2)

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

struct WaveScratch {
  unsigned long long cached;  // passive set whose factor is in Lw / dg (0: none)
  double dg;                  // lane p: L[p][p] of that factor
  double *Lw;   // strict lower triangle of the Cholesky factor of G[P,P], row p at Lw + p * ldw
  int ldw;
  int *idx;     // idx[p]: component of the p-th passive entry
  double *cv;   // compacted right-hand side
  double *xs;   // solution scattered back to component order
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
  const int ldw = ws.ldw;
  if (pas == ws.cached) {
    dg = ws.dg;
    for (int j = 0; j < np; ++j) {  // L z = b with the cached factor, same operation order
      const double zj = bcast(t, j) / bcast(dg, j);
      if (p == j)
        t = zj;
      else if (valid && p > j)
        t -= Lw[p * ldw + j] * zj;
    }
  } else {
  ws.cached = 0;
  for (int j = 0; j < np; ++j) {
    const int ij = __builtin_amdgcn_readfirstlane(ws.idx[j]);
    const bool below = valid && p > j;
    const int lrow = below ? p : j;  // the other lanes redo row j (unused)
    double ajj = Hs[ij + r * ij];
    double sv = Hs[(below ? myi : ij) + r * ij];
    const double *rj = Lw + j * ldw, *ri = Lw + lrow * ldw;
    int k = 0;
    for (; k + 4 <= j; k += 4) {  // four load pairs in flight; the subtractions stay in k order
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
    const double ljj = sqrt(ajj);
    const double lij = sv / ljj;
    if (below) Lw[p * ldw + j] = lij;
    const double zj = bcast(t, j) / ljj;  // forward substitution, column by column
    if (p == j) {
      t = zj;
      dg = ljj;
    } else if (below) {
      t -= lij * zj;
    }
    WAVE_SYNC();
  }
  ws.cached = pas;
  ws.dg = dg;
  }
  for (int j = np - 1; j >= 0; --j) {  // L^T x = z
    const double xj = bcast(t, j) / bcast(dg, j);
    if (p == j)
      t = xj;
    else if (valid && p < j)
      t -= Lw[j * ldw + p] * xj;
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
  const int k_model = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
  const int slot = a.slots[k_model];
  const int r = a.mt.rank[slot], col = a.mt.col[slot];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = blockDim.x >> 6;
  const int I = a.I;

  double *Hs = reinterpret_cast<double *>(nnls_dyn);  // r x r, ld = r
  WaveScratch ws;
  ws.cached = 0;
  ws.dg = 1.0;
  {
    const int ldw = r | 1;
    const size_t per_wave = (size_t)a.rmax * (a.rmax | 1) + 64 + 64 + 32;  // doubles (idx: 64 ints)
    double *base = Hs + (size_t)a.rmax * a.rmax + per_wave * wave;
    ws.Lw = base;
    ws.ldw = ldw;
    ws.cv = base + (size_t)a.rmax * (a.rmax | 1);
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
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  unsigned long long *actp = a.act + (long long)I * col;
  double *rowdot = a.rowdot + (long long)I * k_model;
  const int rows_per = (I + a.chunks - 1) / a.chunks;
  const int row0 = chunk * rows_per, row1 = min(I, row0 + rows_per);
  int status = 0;

  for (int row = row0 + wave; row < row1; row += W) {
    const bool in = lane < r;
    const double y = in ? (double)fac[row + (long long)I * lane] : 0.0;
    unsigned long long act = uniform64(actp[row]) & rmask;
    act &= ~__ballot(in && y > 0.0);  // "determine previous active set" (update.cpp:87-91)
    double d = 0.0, sp = 0.0;
    int budget = NNLS_MAX_EXCHANGES;
    unsigned long long pas = ~act & rmask;
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
      const unsigned long long hit = __ballot(ia && w == wmax);
      const int m = __ffsll((long long)hit) - 1;  // Tensor::max_id: the first of equal maxima
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
}

size_t nnls_lds_bytes(int rmax, int waves) {
  const size_t per_wave = (size_t)rmax * (rmax | 1) + 64 + 64 + 32;
  return ((size_t)rmax * rmax + per_wave * waves) * sizeof(double);
}

hipError_t nnls_launch(const NnlsArgs &a_in, hipStream_t st) {
  if (a_in.n_slots <= 0) return hipSuccess;
  NnlsArgs a = a_in;
  a.rmax = std::min(std::max(a.rmax, 1), CALS_RMAX);
  const size_t budget = (size_t)160 * 1024 - 1024;
  int waves = 4;
  while (waves > 1 && nnls_lds_bytes(a.rmax, waves) > budget) --waves;
  const size_t dyn = nnls_lds_bytes(a.rmax, waves);
  // rows per workgroup: at least 4 per wave, enough workgroups to fill 256 CUs several times over
  int chunks = std::max(1, std::min((a.I + 4 * waves - 1) / (4 * waves), (4096 + a.n_slots - 1) / a.n_slots));
  a.chunks = chunks;
  static AttrOnce once[2];
  const int di = (a.dtype == CALS_F32) ? 1 : 0;
  const void *fn = di ? reinterpret_cast<const void *>(&nnls_kernel<float>)
                      : reinterpret_cast<const void *>(&nnls_kernel<double>);
  const hipError_t ea = once[di].ensure(
      [&] { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget); });
  if (ea != hipSuccess) return ea;
  const dim3 grid((unsigned)(a.n_slots * chunks)), block(64 * waves);
  if (di)
    hipLaunchKernelGGL(nnls_kernel<float>, grid, block, dyn, st, a);
  else
    hipLaunchKernelGGL(nnls_kernel<double>, grid, block, dyn, st, a);
  return hipGetLastError();
}

// A model's active sets at admission: every constraint active (include/ktensor.h:69,108).
// desc as for init_slots_launch: n x {slot, col, rank, jk_mode, jk_fiber}
__global__ void __launch_bounds__(256) nnls_reset_kernel(const int *desc, NnlsResetArgs a) {
  const int col = desc[5 * blockIdx.x + 1], r = desc[5 * blockIdx.x + 2];
  const unsigned long long rmask = (r >= 64) ? ~0ull : ((1ull << r) - 1ull);
  for (int m = 0; m < a.n_modes; ++m) {
    unsigned long long *p = a.act[m] + (long long)a.I[m] * col;
    for (int i = threadIdx.x; i < a.I[m]; i += 256) p[i] = rmask;
  }
}

hipError_t nnls_reset_launch(const int *desc, int n, const NnlsResetArgs &a, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(nnls_reset_kernel, dim3(n), dim3(256), 0, st, desc, a);
  return hipGetLastError();
}

}  // namespace calship
