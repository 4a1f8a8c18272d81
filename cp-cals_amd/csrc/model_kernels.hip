// Batched per-model kernels of the CALS sweep on gfx950: one workgroup (4 wavefronts) per
// in-flight model for the update, one wavefront for the smaller kernels.
//
//   update_kernel   src/cals.cpp:239-256 for one mode, all models at once:
//                   sum of the MTTKRP split partials (fixed order) -> hadamard_but_one
//                   (src/utils/utils.cpp:161-172) -> dpotrf('L') + 2x dtrsm
//                   (src/utils/update.cpp:178-192) -> set_jk_fiber(0) (include/ktensor.h:316-325)
//                   -> Ktensor::normalize(mode, iter) (src/ktensor.cpp:66-83) -> update_gramian
//                   (src/utils/utils.cpp:174-178, on v_mfma_f64_16x16x4_f64); on the last mode also
//                   hadamard_all + compute_fast_error + calculate_new_fit (src/cals.cpp:281-303,
//                   src/utils/error.cpp:64-89, include/ktensor.h:178-183) while G is still in
//                   registers (G_last of src/cals.cpp:230-234 is never materialised).
//   ls_*            line search, NO_ERROR_CHECKING (src/utils/line_search.cpp:24-71, 228-271)
//   finish_kernel   eviction rule + iters++ (src/cals.cpp:336-354)
//
// Everything a model owns is indexed by its first column `col` in the multi-factor buffers:
// factor columns [col, col+r), lambda[col..], and an r x r Gramian per mode stored in columns
// [col, col+r) of a CALS_GLD x buffer matrix (ld = CALS_GLD = the rank limit per model).
#include "cals_hip_internal.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <type_traits>

namespace calship {

typedef double v4d __attribute__((ext_vector_type(4)));

// 1: the backward row solve of update_body_lds runs without `< r` guards on the identity-padded L (bit-identical results;
// C4 -5 us per sweep, C2 / C3 unchanged; the same for the forward solve spills 4 KB: not done)
#ifndef CALS_UPD_BACK_NOGUARD
#define CALS_UPD_BACK_NOGUARD 0
#endif
// update kernels: one workgroup of 4 waves per model
#define UPD_THREADS 256
#define UPD_WAVES 4

#ifdef CALS_DIAG
#define UPD_STAMP(k)                                                                          \
  do {                                                                                        \
    if (a.dbg_trace && a.mode == 0 && r == 20 && blockIdx.x < 24 && threadIdx.x == 0)         \
      a.dbg_trace[k] = __builtin_amdgcn_s_memtime();                                          \
  } while (0)
#define UPD_STAMP_H(k)                                                                        \
  do {                                                                                        \
    if (a.dbg_trace && a.mode == 0 && threadIdx.x == 0) a.dbg_trace[k] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define UPD_STAMP(k) do { } while (0)
#define UPD_STAMP_H(k) do { } while (0)
#endif

// The update bodies are non-inlined functions (one per rank class).  Handing them the kernel's by-value
// UpdateArgs by reference made the compiler copy all 352 bytes of it into every lane's scratch at kernel start
// and read every field back through scratch; instead every body reads the struct where the dispatcher put it --
// the calling kernel's kernarg segment (constant address space): every field read is a scalar load, and there is
// no argument frame.  The kernel hands the segment's address over as an ordinary pointer argument (a VGPR pair under
// the calling convention; __builtin_amdgcn_kernarg_segment_ptr() itself reads as NULL inside a non-kernel function)
// and the body makes it wave-uniform again with two v_readfirstlane.
typedef const __attribute__((address_space(4))) UpdateArgs *UpdArgsPtr;
typedef const __attribute__((address_space(4))) UpdateArgs &UpdArgsRef;
__device__ __forceinline__ UpdArgsPtr upd_kernargs() {
  // kernel side: the kernels below take ONE explicit argument, the UpdateArgs, at offset 0 of the kernarg segment
  return (UpdArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
}
__device__ __forceinline__ UpdArgsRef upd_uniform(UpdArgsPtr p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return *(UpdArgsPtr)(((unsigned long long)hi << 32) | lo);
}

// One row of a model's G: from the factor buffer (already reduced) or summed here over the split-K partial tiles
// (UpdateArgs::partial).  Tile loop outermost: the RMAX loads of a step are independent, the sum of every entry
// runs t = 0 .. pT-1 in order -- reduce_partials_kernel's order and rounding, bit for bit.
template <int RMAX, typename T>
__device__ __forceinline__ void upd_load_g_row(UpdArgsRef a, const T *fac, int i, int I, int col, int r,
                                               double (&x)[RMAX]) {
  if (a.partial == nullptr) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c) x[c] = (c < r) ? (double)fac[i + (long long)I * c] : 0.0;
    return;
  }
  const long long tile = (long long)a.ldPart * CALS_BN;
  // a model's columns are consecutive; they cross a 128-column block boundary at most once (at local column cs)
  const int cs = CALS_BN - (col & (CALS_BN - 1));
  const long long jump = (long long)a.pT * tile - tile;
  const T *p0 = static_cast<const T *>(a.partial) + (long long)(col >> 7) * a.pT * tile +
                (long long)a.ldPart * (col & (CALS_BN - 1)) + i;
#pragma unroll
  for (int c = 0; c < RMAX; ++c) x[c] = 0.0;
  for (int t = 0; t < a.pT; ++t) {
    const T *pt = p0 + (long long)t * tile;
#pragma unroll
    for (int c = 0; c < RMAX; ++c)
      if (c < r) x[c] += (double)pt[(long long)a.ldPart * c + (c >= cs ? jump : 0)];
  }
  if constexpr (!std::is_same<T, double>::value) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c) x[c] = (double)(T)x[c];
  }
}

// Pt[(column block)][a][128] = the model's columns of the normalised factor (rows 0 .. I-1) and zeros in the pad rows
// I .. ptAp-1 (UpdateArgs::pt): element k of the model's I x r panel is (row k / r, column k % r), so consecutive
// lanes write consecutive columns of one row -- r * sizeof(T) contiguous bytes -- instead of 64 different rows.
template <typename T, typename PANEL>
__device__ __forceinline__ void upd_write_pt(UpdArgsRef a, PANEL panel, long long pld, int I, int col, int r, int tid) {
  T *const pt = static_cast<T *>(a.pt);
  const int n = a.ptAp * r;
  int i = tid / r, c = tid - i * r;
  const int di = UPD_THREADS / r, dc = UPD_THREADS - di * r;
  for (int k = tid; k < n; k += UPD_THREADS) {
    const int gc = col + c;
    const T v = (i < I) ? (T)panel[i + pld * c] : (T)0;
    pt[((long long)(gc >> 7) * a.ptAp + i) * CALS_BN + (gc & (CALS_BN - 1))] = v;
    i += di;
    c += dc;
    if (c >= r) {
      c -= r;
      ++i;
    }
  }
}

// finish_kernel's rule for one model (see UpdateArgs::fin)
__device__ __forceinline__ void apply_finish_rule(UpdArgsRef a, int slot) {
  const long long it = a.mt.iters[slot];
  bool evict = false;
  if (a.fin.evict_enabled) {
    if (!a.fin.force_max_iter)
      evict = (fabs(a.mt.old_fit[slot] - a.mt.fit[slot]) < a.fin.tol) || (it >= a.fin.max_iter);
    else
      evict = it >= a.fin.max_iter;
  }
  if (evict)
    a.mt.flags[slot] |= 4;
  else
    a.mt.iters[slot] = it + 1;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <typename PTR>
__device__ __forceinline__ v4d gramian_tile(PTR panel, int row0, int row1, long long ld, int r,
                                            int lane, int bi, int bj);

// Gamma = P^T P for the I x r panel (ld) on the f64 matrix cores (r <= 32 in one pass): each k-step covers 4
// rows; lane (lcol, krow) supplies P[i0+krow, lcol] as both the A and the B operand.
// Written to g (ld = LDG: CALS_GLD for the Gramian stores).  Must be called by a whole wave with EXEC all ones.
template <typename T, int LDG = CALS_GLD>
__device__ __forceinline__ void gramian_wave(const T *panel, int rows, long long ld, int r,
                                             double *g, int lane) {
  const int krow = lane >> 4, lcol = lane & 15;
  if (r > CALS_RFAST) {  // ranks 33..64: one pass over the panel per tile pair
    const int nt = (r + 15) >> 4;
    for (int bi = 0; bi < nt; ++bi)
      for (int bj = bi; bj < nt; ++bj) {
        const v4d t = gramian_tile(panel, 0, rows, ld, r, lane, bi, bj);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = 16 * bi + krow + 4 * reg, cc = 16 * bj + lcol;
          if (row < r && cc < r) {
            g[row + LDG * cc] = t[reg];
            g[cc + LDG * row] = t[reg];
          }
        }
      }
    return;
  }
  const bool two = r > 16;
  v4d a00 = {0.0, 0.0, 0.0, 0.0}, a01 = {0.0, 0.0, 0.0, 0.0}, a11 = {0.0, 0.0, 0.0, 0.0};
  const bool c0ok = lcol < r, c1ok = (16 + lcol) < r;
  for (int i0 = 0; i0 < rows; i0 += 4) {
    const int i = i0 + krow;
    const bool rok = i < rows;
    const double p0 = (rok && c0ok) ? (double)panel[i + ld * lcol] : 0.0;
    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, p0, a00, 0, 0, 0);
    if (two) {
      const double p1 = (rok && c1ok) ? (double)panel[i + ld * (16 + lcol)] : 0.0;
      a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0, p1, a01, 0, 0, 0);
      a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1, p1, a11, 0, 0, 0);
    }
  }
  // f64 C/D layout: lane holds D[row = krow + 4*reg][col = lcol]
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = krow + 4 * reg;
    if (row < r && lcol < r) g[row + LDG * lcol] = a00[reg];
    if (two) {
      if (row < r && 16 + lcol < r) {
        g[row + LDG * (16 + lcol)] = a01[reg];
        g[(16 + lcol) + LDG * row] = a01[reg];
      }
      if (16 + row < r && 16 + lcol < r) g[(16 + row) + LDG * (16 + lcol)] = a11[reg];
    }
  }
}

// One 16 x 16 tile (bi, bj) of P^T P over rows [row0, row1): the general form used for ranks > 32.
template <typename PTR>
__device__ __forceinline__ v4d gramian_tile(PTR panel, int row0, int row1, long long ld, int r,
                                            int lane, int bi, int bj) {
  const int krow = lane >> 4, lcol = lane & 15;
  const bool ciok = (16 * bi + lcol) < r, cjok = (16 * bj + lcol) < r;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  for (int i0 = row0; i0 < row1; i0 += 4) {
    const int i = i0 + krow;
    const bool rok = i < row1;
    const double pi = (rok && ciok) ? (double)panel[i + ld * (16 * bi + lcol)] : 0.0;
    const double pj = (bi == bj) ? pi : ((rok && cjok) ? (double)panel[i + ld * (16 * bj + lcol)] : 0.0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pi, pj, acc, 0, 0, 0);
  }
  return acc;
}

// Partial Gramian of rows [row0, row1) on the matrix cores (see gramian_wave); accumulators out.
template <typename PTR>
__device__ __forceinline__ void gramian_rows(PTR panel, int row0, int row1, int rows,
                                             long long ld, int r, int lane, v4d &a00, v4d &a01,
                                             v4d &a11) {
  const int krow = lane >> 4, lcol = lane & 15;
  const bool two = r > 16;
  const bool c0ok = lcol < r, c1ok = (16 + lcol) < r;
  for (int i0 = row0; i0 < row1; i0 += 8) {
    const int ia = i0 + krow, ib = i0 + 4 + krow;
    const bool ra = ia < row1 && ia < rows, rb = ib < row1 && ib < rows;
    const double p0a = (ra && c0ok) ? (double)panel[ia + ld * lcol] : 0.0;
    const double p0b = (rb && c0ok) ? (double)panel[ib + ld * lcol] : 0.0;
    double p1a = 0.0, p1b = 0.0;
    if (two) {
      p1a = (ra && c1ok) ? (double)panel[ia + ld * (16 + lcol)] : 0.0;
      p1b = (rb && c1ok) ? (double)panel[ib + ld * (16 + lcol)] : 0.0;
    }
    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0a, p0a, a00, 0, 0, 0);
    if (two) {
      a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0a, p1a, a01, 0, 0, 0);
      a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1a, p1a, a11, 0, 0, 0);
    }
    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0b, p0b, a00, 0, 0, 0);
    if (two) {
      a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(p0b, p1b, a01, 0, 0, 0);
      a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(p1b, p1b, a11, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// update: one workgroup of 4 waves per model
// ---------------------------------------------------------------------------------------------

// 26.7 KB: the partial Gramian tiles (gp, written after the last barrier-separated use of H / L, 1 / diagonal and
// the column statistics) share their storage with those -- 39.5 KB as separate fields kept a second workgroup
// off the CU at C3's shape (48 KB panel).  update_body_huge also borrows gp[0] for its 16 x 17 diagonal block
// while it factors (it keeps H / L elsewhere and reads dinv, which lies behind that block).
struct UpdShared {
  union {
    struct {
      double Hs[CALS_RFAST * CALS_RFAST];
      double dinv[CALS_GLD];
      double red[UPD_WAVES][CALS_RFAST][2];
      int redi[UPD_WAVES][CALS_RFAST];
    };
    double gp[UPD_WAVES][3][256];
  };
  double lams[CALS_GLD];
  double redt[UPD_WAVES];
};
static_assert(sizeof(double) * (16 * 17) <= sizeof(double) * CALS_RFAST * CALS_RFAST, "s_dblk must end before dinv");

// noinline: inlining all rank classes into one kernel made the register allocator spill heavily
// (each body alone fits); as separate functions each gets its own allocation.  noreturn: a body is the last thing
// its kernel does, so it ends the wave itself (s_endpgm) -- a returning function would first reload the
// callee-saved registers it saved on entry (47 scratch loads + a wait at rank 20) for a caller that ends at once.
#define UPD_BODY_ATTR __attribute__((noinline, noreturn))
// How a body ends its wave.  __builtin_amdgcn_endpgm() is a "return" to the frame lowering, which puts the reloads of all
// callee-saved registers in front of it (75 scratch loads at rank 20 that nothing ever uses, and the wave is not retired
// before they have come back); an s_endpgm the compiler does not see, followed by unreachable, leaves no return block.
#ifndef CALS_UPD_ASM_ENDPGM
#define CALS_UPD_ASM_ENDPGM 1
#endif
#if CALS_UPD_ASM_ENDPGM
#define UPD_END_WAVE()               \
  do {                               \
    asm volatile("s_endpgm" ::: "memory"); \
    __builtin_unreachable();         \
  } while (0)
#else
#define UPD_END_WAVE() __builtin_amdgcn_endpgm()
#endif
template <int RMAX, typename T>
static __device__ UPD_BODY_ATTR void update_body(UpdArgsPtr a_ptr, int slot, int r, int col, int jkp,
                                                      UpdShared &sh) {
  UpdArgsRef a = upd_uniform(a_ptr);  // the calling kernel's argument block (constant memory, scalar loads)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long iters = a.mt.iters[slot];
  const int jkf = (jkp >= 0 && (jkp & 7) == a.mode) ? (jkp >> 3) : -1;  // jkp: upd_jk_pack
  const int I = a.I;
  double *Hs = sh.Hs;

  UPD_STAMP(0);
  // H = hadamard of the other modes' Gramians (hadamard_but_one)
  for (int e = tid; e < r * r; e += UPD_THREADS) {
    const int i = e % r, j = e / r;
    double h = 1.0;
    for (int m = 0; m < a.n_modes; ++m)
      if (m != a.mode) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
    Hs[i + RMAX * j] = h;
  }
  __syncthreads();

  UPD_STAMP(1);
  // NNLS update: the panel already holds the constrained solution (nnls_kernel.hip)
  const double *rowdot = a.rowdot ? a.rowdot + (long long)I * blockIdx.x : nullptr;
  const bool solved = rowdot != nullptr;
  // dpotrf('L') restated as unblocked dpotf2 (lane = row; every wave computes, wave 0 writes).
  // info != 0: stop, keep going with whatever is in H, as the reference does
  // (update.cpp:183-185 only logs).
  int info = 0;
  for (int j = 0; j < (solved ? 0 : r); ++j) {
    double ajj = Hs[j + RMAX * j];
    for (int k = 0; k < j; ++k) ajj -= Hs[j + RMAX * k] * Hs[j + RMAX * k];
    if (!(ajj > 0.0)) {
      __syncthreads();
      if (tid == 0) Hs[j + RMAX * j] = ajj;
      info = j + 1;
      break;
    }
    ajj = sqrt(ajj);
    double s = 0.0;
    if (lane > j && lane < r) {
      s = Hs[lane + RMAX * j];
      for (int k = 0; k < j; ++k) s -= Hs[lane + RMAX * k] * Hs[j + RMAX * k];
      s = s / ajj;
    }
    __syncthreads();
    if (wave == 0) {
      if (lane == j)
        Hs[j + RMAX * j] = ajj;
      else if (lane > j && lane < r)
        Hs[lane + RMAX * j] = s;
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid < r) sh.dinv[tid] = 1.0 / Hs[tid + RMAX * tid];
  if (tid == 0 && !solved) a.mt.potrf_info[slot] = info;
  __syncthreads();
  const double *dinv = sh.dinv;
  UPD_STAMP(2);

  // storage type T (double | float); all arithmetic below is fp64
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  const bool first = (iters == 1);

  // per-column statistics of this thread's rows: first sweep: st1 = sum of squares; later:
  // st1 = max|x|, st2 = x there, sti = its row (cblas_idamax: first index of the largest |x|)
  double st1[RMAX], st2[RMAX];
  int sti[RMAX];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) {
    st1[c] = first ? 0.0 : -1.0;
    st2[c] = 0.0;
    sti[c] = 0x7fffffff;
  }
  double t3 = 0.0;

  // pass 1, one row per thread: G row -> two triangular solves -> unnormalised factor row
  for (int i = tid; i < I; i += UPD_THREADS) {
    double x[RMAX], g[RMAX];
    // L stays in LDS (broadcast reads): without this barrier LICM hoists all r^2/2 entries of L
    // out of the row loop into VGPRs and the kernel spills.
    asm volatile("" ::: "memory");
    // G row: the MTTKRP result, already reduced over the split partials into the factor buffer
    // (the reference's MTTKRP overwrites factor n too, src/utils/mttkrp.cpp:311)
    upd_load_g_row<RMAX, T>(a, fac, i, I, col, r, x);
#pragma unroll
    for (int c = 0; c < RMAX; ++c) g[c] = x[c];
    if (!solved) {
      // B := B * inv(L^T)   (dtrsm Right, Lower, Trans)
#pragma unroll
      for (int k = 0; k < RMAX; ++k) {
        if (k < r) {
          x[k] = dinv[k] * x[k];
#pragma unroll
          for (int j = k + 1; j < RMAX; ++j)
            if (j < r) x[j] -= Hs[j + RMAX * k] * x[k];
        }
      }
      // B := B * inv(L)     (dtrsm Right, Lower, NoTrans)
#pragma unroll
      for (int j = RMAX - 1; j >= 0; --j) {
        if (j < r) {
#pragma unroll
          for (int k = j + 1; k < RMAX; ++k)
            if (k < r) x[j] -= Hs[k + RMAX * j] * x[k];
          x[j] = dinv[j] * x[j];
        }
      }
    }
    if (i == jkf) {
#pragma unroll
      for (int c = 0; c < RMAX; ++c) x[c] *= 0.0;
    }
    if (solved) {  // <x, g> of this row as nnls_kernel left it; g itself is gone
#pragma unroll
      for (int c = 0; c < RMAX; ++c) g[c] = 0.0;
      if (i != jkf) t3 += rowdot[i];
    }
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      if (c < r) {
        fac[i + (long long)I * c] = (T)x[c];
        t3 += x[c] * g[c];  // = lambda_c * A[i,c] * G[i,c] of compute_fast_error's term3
        if (first) {
          st1[c] += x[c] * x[c];
        } else {
          const double ax = fabs(x[c]);
          if (ax > st1[c]) {
            st1[c] = ax;
            st2[c] = x[c];
            sti[c] = i;
          }
        }
      }
    }
  }

  UPD_STAMP(3);
  // column scales: wave butterflies (independent per column, so they pipeline), then the four
  // waves are combined in fixed order by thread c
#pragma unroll
  for (int c = 0; c < RMAX; ++c) {
    if (c < r) {
      if (first) {
        const double tot = wave_sum(st1[c]);
        if (lane == 0) sh.red[wave][c][0] = tot;
      } else {
        double m = st1[c], v = st2[c];
        int ix = sti[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double m2 = __shfl_xor(m, off);
          const double v2 = __shfl_xor(v, off);
          const int i2 = __shfl_xor(ix, off);
          const bool take = (m2 > m) || (m2 == m && i2 < ix);
          m = take ? m2 : m;
          v = take ? v2 : v;
          ix = take ? i2 : ix;
        }
        if (lane == 0) {
          sh.red[wave][c][0] = m;
          sh.red[wave][c][1] = v;
          sh.redi[wave][c] = ix;
        }
      }
    }
  }
  t3 = wave_sum(t3);
  if (lane == 0) sh.redt[wave] = t3;
  __syncthreads();
  UPD_STAMP(4);
  if (tid < r) {
    double lam;
    if (first) {
      double tot = 0.0;
      for (int w = 0; w < UPD_WAVES; ++w) tot += sh.red[w][tid][0];
      lam = sqrt(tot);
    } else {
      double m = sh.red[0][tid][0], v = sh.red[0][tid][1];
      int ix = sh.redi[0][tid];
      for (int w = 1; w < UPD_WAVES; ++w) {
        const double m2 = sh.red[w][tid][0];
        const int i2 = sh.redi[w][tid];
        if ((m2 > m) || (m2 == m && i2 < ix)) {
          m = m2;
          v = sh.red[w][tid][1];
          ix = i2;
        }
      }
      lam = v;
    }
    sh.lams[tid] = lam;
    a.lambda[col + tid] = lam;
  }
  __syncthreads();
  t3 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
  const double *lams = sh.lams;

  UPD_STAMP(5);
  // pass 2: cblas_dscal by 1/lambda (skipped for lambda == 0); each thread rescales its own rows
  for (int i = tid; i < I; i += UPD_THREADS) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      if (c < r) {
        const double lam = lams[c];
        if (lam != 0.0) fac[i + (long long)I * c] = (T)((1.0 / lam) * (double)fac[i + (long long)I * c]);
      }
    }
  }
  __syncthreads();
  if (a.pt) upd_write_pt<T>(a, (const T *)fac, (long long)I, I, col, r, tid);

  UPD_STAMP(6);
  // update_gramian: rows split over the four waves, partial tiles combined in fixed order
  {
    v4d a00 = {0.0, 0.0, 0.0, 0.0}, a01 = {0.0, 0.0, 0.0, 0.0}, a11 = {0.0, 0.0, 0.0, 0.0};
    const int chunk = ((I + UPD_WAVES - 1) / UPD_WAVES + 7) / 8 * 8;
    const int row0 = wave * chunk, row1 = min(I, row0 + chunk);
    gramian_rows(fac, row0, row1, I, I, r, lane, a00, a01, a11);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      sh.gp[wave][0][lane * 4 + reg] = a00[reg];
      sh.gp[wave][1][lane * 4 + reg] = a01[reg];
      sh.gp[wave][2][lane * 4 + reg] = a11[reg];
    }
  }
  __syncthreads();
  {
    double *g = a.gram[a.mode] + CALS_GLD * (long long)col;
    const int tile = tid >> 6;  // 0: (0,0)  1: (0,1)+(1,0)  2: (1,1); wave 3 idles
    if (tile < 3) {
      const int krow = lane >> 4, lcol = lane & 15;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int e = lane * 4 + reg;
        const double v = ((sh.gp[0][tile][e] + sh.gp[1][tile][e]) + sh.gp[2][tile][e]) + sh.gp[3][tile][e];
        const int row = krow + 4 * reg;  // f64 C/D layout: D[row = krow + 4*reg][col = lcol]
        if (tile == 0) {
          if (row < r && lcol < r) g[row + CALS_GLD * lcol] = v;
        } else if (tile == 1) {
          if (row < r && 16 + lcol < r) {
            g[row + CALS_GLD * (16 + lcol)] = v;
            g[(16 + lcol) + CALS_GLD * row] = v;
          }
        } else {
          if (16 + row < r && 16 + lcol < r) g[(16 + row) + CALS_GLD * (16 + lcol)] = v;
        }
      }
    }
  }

  UPD_STAMP(7);
  if (a.is_last) {
    __syncthreads();
    double t2 = 0.0;
    for (int e = tid; e < r * r; e += UPD_THREADS) {
      const int i = e % r, j = e / r;
      double h = 1.0;
      for (int m = 0; m < a.n_modes; ++m) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
      t2 += lams[i] * lams[j] * h;
    }
    t2 = wave_sum(t2);
    __syncthreads();
    if (lane == 0) sh.redt[wave] = t2;
    __syncthreads();
    if (tid == 0) {
      t2 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
      const double xn = (jkp >= 0) ? a.jk_norms[jkp >> 3] : a.X_norm;
      const double e2 = fmax(xn * xn + t2 - 2.0 * t3, 0.0);
      const double err = sqrt(e2);
      a.mt.err[slot] = err;
      const double of = a.mt.fit[slot];
      a.mt.old_fit[slot] = of;
      a.mt.fit[slot] = 1.0 - fabs(err) / a.X_norm;
      if (a.fin.on) apply_finish_rule(a, slot);
    }
  }
  UPD_END_WAVE();  // the body ends the wave (see UPD_BODY_ATTR): no epilogue that reloads callee-saved registers
}

// The same update with the model's factor panel kept in LDS between the phases (used when
// I x r_max elements fit next to UpdShared; the body above, which goes through global memory between
// its passes, remains for taller modes).  Measured on a rank-20 model at C2/C3 (tools/update_trace.py)
// the first version spent 36 % of its 45-57 us in the Cholesky (every thread recomputing the pivots,
// two block barriers per column) and 28 % in per-thread column statistics reduced by 20 butterflies
// per wave.  Here: the Cholesky runs on ONE wave without block barriers (LDS operations of a wave
// execute in order); the solved rows go to LDS; each wave reduces r/4 columns reading LDS; the
// scaling pass and the Gramian read LDS instead of reloading the panel from HBM.
// Arithmetic: Cholesky, solves, scaling and Gramian are operation-for-operation those of the body
// above; the column norms of a model's first sweep are summed in a different (fixed) order.
extern __shared__ __attribute__((aligned(16))) unsigned char upd_dyn[];

template <int RMAX, typename T>
static __device__ UPD_BODY_ATTR void update_body_lds(UpdArgsPtr a_ptr, int slot, int r, int col, int jkp,
                                                      UpdShared &sh) {
  UpdArgsRef a = upd_uniform(a_ptr);  // the calling kernel's argument block (constant memory, scalar loads)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long iters = a.mt.iters[slot];
  const int jkf = (jkp >= 0 && (jkp & 7) == a.mode) ? (jkp >> 3) : -1;  // jkp: upd_jk_pack
  const int I = a.I;
  const int xld = a.xld;
  double *Hs = sh.Hs;
  T *xs = reinterpret_cast<T *>(upd_dyn);  // I x r panel, ld = xld (storage type, like the factor)

  UPD_STAMP(0);
  // NNLS update: the panel already holds the constrained solution (nnls_kernel.hip)
  const double *rowdot = a.rowdot ? a.rowdot + (long long)I * blockIdx.x : nullptr;
  const bool solved = rowdot != nullptr;
  const double *dinv = sh.dinv;
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  const bool first = (iters == 1);
  double t3 = 0.0;
  // The factorisation publishes L in panels of PW columns; the rows' forward substitution follows it panel by panel
  constexpr int PW = 4;
  static_assert(RMAX % PW == 0, "rank classes are multiples of the panel width");

  // One row: G row -> two triangular solves -> unnormalised factor row, into the LDS panel.
  // PIPE (waves 1-3 while wave 0 factors): barrier p of the workgroup says "columns PW p .. PW p + PW - 1 of L are in
  // LDS"; step k of the forward substitution B := B inv(L^T) needs column k and nothing else, so the rows take a
  // panel's steps right behind its barrier and then wait for the next panel -- a step is ~80 cycles, a column of
  // the factorisation ~500: the forward solve and the load of the G rows hide behind the factorisation.  1 / l_kk is
  // divided out by the row itself there (same value as dinv, which exists only behind the last barrier).
  auto solve_row = [&]<bool PIPE>(int i, bool valid) {
    double x[RMAX], g[RMAX];
    asm volatile("" ::: "memory");  // keep L in LDS (see update_body)
    if (valid) {
      upd_load_g_row<RMAX, T>(a, fac, i, I, col, r, x);
    } else {
#pragma unroll
      for (int c = 0; c < RMAX; ++c) x[c] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < RMAX; ++c) g[c] = x[c];
    if (!solved) {
#pragma unroll
      for (int k = 0; k < RMAX; ++k) {
        if constexpr (PIPE) {
          if (k % PW == 0) {
            __syncthreads();  // barrier k / PW
            asm volatile("" ::: "memory");
          }
        }
        if (k < r) {
          const double dk = PIPE ? 1.0 / Hs[k + RMAX * k] : dinv[k];
          x[k] = dk * x[k];
#pragma unroll
          for (int j = k + 1; j < RMAX; ++j)
            if (j < r) x[j] -= Hs[j + RMAX * k] * x[k];
        }
      }
      if constexpr (PIPE) {
        __syncthreads();  // barrier NPAN: 1 / diagonal is in LDS, the factorisation is through
        asm volatile("" ::: "memory");
      }
#if CALS_UPD_BACK_NOGUARD
      // L is diag(L_r, I) in LDS and x is zero in the padded columns (upd_load_g_row), 1 / diagonal is 1 there: the padded
      // steps subtract exact zeros -- no `< r` guard, one basic block, and the compiler may take a column's entries of L
      // (one contiguous run, read 16 bytes at a time) while the previous column's chain of FMAs is still running
#pragma unroll
      for (int j = RMAX - 1; j >= 0; --j) {
#pragma unroll
        for (int k = j + 1; k < RMAX; ++k) x[j] -= Hs[k + RMAX * j] * x[k];
        x[j] = dinv[j] * x[j];
      }
#else
#pragma unroll
      for (int j = RMAX - 1; j >= 0; --j) {
        if (j < r) {
#pragma unroll
          for (int k = j + 1; k < RMAX; ++k)
            if (k < r) x[j] -= Hs[k + RMAX * j] * x[k];
          x[j] = dinv[j] * x[j];
        }
      }
#endif
    }
    if (i == jkf) {
#pragma unroll
      for (int c = 0; c < RMAX; ++c) x[c] *= 0.0;
    }
    if (solved) {  // <x, g> of this row as nnls_kernel left it; g itself is gone
#pragma unroll
      for (int c = 0; c < RMAX; ++c) g[c] = 0.0;
      if (valid && i != jkf) t3 += rowdot[i];
    }
    if (valid) {
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        if (c < r) {
          xs[i + xld * c] = (T)x[c];
          t3 += x[c] * g[c];
        }
      }
    }
  };

  if (!solved) {
    if (wave == 0) {
      // H = hadamard of the other modes' Gramians (hadamard_but_one), lane = row, the row in REGISTERS straight
      // from the Gramian stores, PADDED to RMAX x RMAX with the identity: the factorisation runs over RMAX columns
      // without a single `c < r` guard (r reaches this non-inlined function in a VGPR, so every guard is an
      // exec-mask save / restore + branch around one FMA); the padding contributes exact zeros, L comes out as
      // diag(L_r, I).
      // (Every load is issued, from a clamped address, and the padding is selected afterwards: loads behind
      // per-lane guards come out as one exposed round trip per entry -- 31 K cycles for 40 of them.)
      double Lr[RMAX];
      const int hl = lane < r ? lane : 0;
#pragma unroll
      for (int c = 0; c < RMAX; ++c) Lr[c] = 1.0;
      if (a.n_modes == 3) {  // both other modes' entries in flight at once: one round trip
        const int m1 = a.mode == 0 ? 1 : 0, m2 = a.mode == 2 ? 1 : 2;
        const double *g1 = a.gram[m1] + hl + CALS_GLD * (long long)col, *g2 = a.gram[m2] + hl + CALS_GLD * (long long)col;
        double t1[RMAX], t2[RMAX];
#pragma unroll
        for (int c = 0; c < RMAX; ++c) {
          const long long off = CALS_GLD * (long long)(c < r ? c : 0);
          t1[c] = g1[off];
          t2[c] = g2[off];
        }
#pragma unroll
        for (int c = 0; c < RMAX; ++c) Lr[c] = (Lr[c] * t1[c]) * t2[c];
      } else {
        for (int m = 0; m < a.n_modes; ++m) {
          if (m == a.mode) continue;
          const double *gm = a.gram[m] + hl + CALS_GLD * (long long)col;
          double t[RMAX];
#pragma unroll
          for (int c = 0; c < RMAX; ++c) t[c] = gm[CALS_GLD * (long long)(c < r ? c : 0)];
#pragma unroll
          for (int c = 0; c < RMAX; ++c) Lr[c] *= t[c];
        }
      }
#pragma unroll
      for (int c = 0; c < RMAX; ++c)
        if (!(lane < r && c < r)) Lr[c] = (lane == c) ? 1.0 : 0.0;
      UPD_STAMP(1);
      // dpotrf('L') restated as unblocked dpotf2, RIGHT-looking: as soon as column j is final, every later column k
      // of the row takes its term  A[i][k] -= L[i][j] L[k][j]  -- the terms of an entry arrive in the order
      // j = 0, 1, ... with the same operands as in dpotf2's left-looking sum, but the RMAX - j - 1 updates of a
      // column step are independent of each other.  L[k][j] comes from lane k by v_readlane (j and k are
      // compile-time after unrolling): the factorisation touches LDS only to publish L.
      // The pivot costs one 1/sqrt: l_jj = a * y, column = s * y with y = 1/sqrt(a) from v_rsq_f64 + two
      // Newton steps and a final correction of l_jj.  Entries differ from IEEE sqrt / divide by at most an
      // ulp or two -- far inside the 1e-12 the kernel tests hold against the oracle.
      // info != 0: stop, keep going with whatever is in H, as the reference does (update.cpp:183-185 only logs):
      // the untouched columns are published as they stand.
      // Schedule (round 4): the step of column j first takes column j + 1 (the one term the next pivot waits for),
      // starts that pivot's 1/sqrt chain -- ~150 cycles of dependent latency on a lone wave -- and applies column j to the
      // columns behind IN THAT CHAIN'S SHADOW: one basic block, so the compiler interleaves them.  The pivot is
      // computed speculatively and committed only if positive (a failed pivot leaves its column as dpotf2 leaves it:
      // with every earlier column applied).  Every entry still receives its terms j = 0, 1, ... in order.
      auto lane_val = [&](double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
      };
      auto pivot = [&](double ajj, double &ljj, double &y) {  // l_jj and 1 / l_jj-ish from a_jj > 0
#if CALS_CHOL_EXACT
        ljj = sqrt(ajj);
        y = 1.0 / ljj;
#else
        y = __builtin_amdgcn_rsq(ajj);
        y = y * fma(-0.5 * ajj * y, y, 1.5);
        y = y * fma(-0.5 * ajj * y, y, 1.5);
        ljj = ajj * y;
        ljj = fma(0.5 * y, fma(-ljj, ljj, ajj), ljj);
#endif
      };
      int info = 0;
      {
        const double a0 = lane_val(Lr[0], 0);
        if (!(a0 > 0.0)) {
          info = 1;  // lane j keeps a_jj in Lr[j], as dpotf2 leaves it (only a real column can fail)
        } else {
          double l00, y0;
          pivot(a0, l00, y0);
#if CALS_CHOL_EXACT
          const double c0 = Lr[0] / l00;
#else
          const double c0 = Lr[0] * y0;
#endif
          Lr[0] = (lane == 0) ? l00 : ((lane > 0) ? c0 : Lr[0]);
        }
      }
#pragma unroll
      for (int j = 0; j < RMAX; ++j) {
        if (info == 0 && j + 1 < RMAX) {  // column j is final
          Lr[j + 1] -= Lr[j] * lane_val(Lr[j], j + 1);
          const double an = lane_val(Lr[j + 1], j + 1);
          const bool ok = an > 0.0;
          double ln, yn;
          pivot(an, ln, yn);  // (garbage when !ok: not committed)
#if CALS_CHOL_EXACT
          const double cn = Lr[j + 1] / ln;
#else
          const double cn = Lr[j + 1] * yn;
#endif
#pragma unroll
          for (int k = j + 2; k < RMAX; ++k) Lr[k] -= Lr[j] * lane_val(Lr[j], k);
          if (ok)
            Lr[j + 1] = (lane == j + 1) ? ln : ((lane > j + 1) ? cn : Lr[j + 1]);
          else
            info = j + 2;
        }
        if (j % PW == PW - 1) {  // a panel is complete: into LDS with it (whole columns: nobody reads above the
          if (lane < RMAX) {     // diagonal), then barrier j / PW
#pragma unroll
            for (int c = j - PW + 1; c <= j; ++c) Hs[lane + RMAX * c] = Lr[c];
          }
          __syncthreads();
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < RMAX) sh.dinv[lane] = 1.0 / Hs[lane + RMAX * lane];
      if (lane == 0) a.mt.potrf_info[slot] = info;
      __syncthreads();  // barrier NPAN
      UPD_STAMP(2);
    } else {
      solve_row.template operator()<true>(tid - 64, tid - 64 < I);
    }
  }
  // the rows the pipelined pass did not take (all of them after the NNLS update): every thread
  for (int i = solved ? tid : UPD_THREADS - 64 + tid; i < I; i += UPD_THREADS) solve_row.template operator()<false>(i, true);
  t3 = wave_sum(t3);
  if (lane == 0) sh.redt[wave] = t3;
  __syncthreads();
  t3 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
  UPD_STAMP(3);

  // column scales (Ktensor::normalize(mode, iteration), src/ktensor.cpp:66-83): first sweep of a model: 2-norm;
  // later: the entry of largest magnitude, first index on ties (cblas_idamax), with its sign.
  // Thread (c = tid % LC, chunk = tid / LC) walks rows [chunk * CH, (chunk + 1) * CH) of column c of the LDS panel in
  // ascending order (LC = 8 | 16 | 32 columns per chunk by rank class: the fewer columns, the more and shorter chunks;
  // odd xld: the columns of a chunk sit on different banks), thread c then folds the chunk results in chunk order --
  // "strictly larger wins" in ascending row order IS idamax's first-index rule, no index comparison and no
  // cross-lane shuffle anywhere.  (Round 3: wave = column group, lanes over rows, six butterfly steps of three
  // 64-bit shuffles per column: 11.8 K of the body's 48.7 K cycles at C2.)
  {
    constexpr int LC = RMAX <= 8 ? 8 : RMAX <= 16 ? 16 : 32;
    constexpr int NCH = UPD_THREADS / LC;
    double *pm = &sh.gp[0][0][0];                // [NCH][LC] chunk maxima / sums   (H, L, 1/diagonal are dead:
    double *pv = pm + UPD_THREADS;               // [NCH][LC] signed entry there     every thread passed the barrier
    const int c = tid % LC, chunk = tid / LC;    //                                  behind the row solves)
    const int CH = (I + NCH - 1) / NCH;
    const int i0 = chunk * CH, i1 = min(I, i0 + CH);
    double m = first ? 0.0 : -1.0, v = 0.0;
    if (c < r) {
      const T *xc = xs + xld * c;
#pragma unroll 4
      for (int i = i0; i < i1; ++i) {
        const double x = (double)xc[i];
        if (first) {
          m += x * x;
        } else {
          const double ax = fabs(x);
          if (ax > m) {
            m = ax;
            v = x;
          }
        }
      }
      pm[chunk * LC + c] = m;
      pv[chunk * LC + c] = v;
    }
    __syncthreads();
    if (tid < r) {
      double mm = pm[tid], vv = pv[tid];
#pragma unroll
      for (int k = 1; k < NCH; ++k) {
        const double m2 = pm[k * LC + tid];
        if (first) {
          mm += m2;
        } else if (m2 > mm) {
          mm = m2;
          vv = pv[k * LC + tid];
        }
      }
      const double lam = first ? sqrt(mm) : vv;
      sh.lams[tid] = lam;
      sh.lams[CALS_RFAST + tid] = (lam != 0.0) ? 1.0 / lam : 1.0;  // cblas_dscal by 1 / lambda, skipped for lambda == 0
      a.lambda[col + tid] = lam;
    }
  }
  __syncthreads();
  const double *lams = sh.lams;
  UPD_STAMP(5);

  // cblas_dscal by 1/lambda (skipped for lambda == 0): the normalised factor goes to HBM once and
  // stays in LDS (rounded to the storage type) for the Gramian.  Thread (row = tid & 127, column parity = tid >> 7).
  for (int i = tid & 127; i < I; i += 128) {
#pragma unroll
    for (int c0 = 0; c0 < RMAX; c0 += 2) {
      const int c = c0 + (tid >> 7);
      if (c < r) {
        const T v = (T)(lams[CALS_RFAST + c] * (double)xs[i + xld * c]);
        fac[i + (long long)I * c] = v;
        xs[i + xld * c] = v;
      }
    }
  }
  __syncthreads();
  if (a.pt) upd_write_pt<T>(a, (const T *)xs, (long long)xld, I, col, r, tid);
  UPD_STAMP(6);

  // update_gramian: rows split over the four waves, partial tiles combined in fixed order
  {
    v4d a00 = {0.0, 0.0, 0.0, 0.0}, a01 = {0.0, 0.0, 0.0, 0.0}, a11 = {0.0, 0.0, 0.0, 0.0};
    const int chunk = ((I + UPD_WAVES - 1) / UPD_WAVES + 7) / 8 * 8;
    const int row0 = wave * chunk, row1 = min(I, row0 + chunk);
    gramian_rows((const T *)xs, row0, row1, I, (long long)xld, r, lane, a00, a01, a11);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      sh.gp[wave][0][lane * 4 + reg] = a00[reg];
      sh.gp[wave][1][lane * 4 + reg] = a01[reg];
      sh.gp[wave][2][lane * 4 + reg] = a11[reg];
    }
  }
  __syncthreads();
  {
    double *g = a.gram[a.mode] + CALS_GLD * (long long)col;
    const int tile = tid >> 6;
    if (tile < 3) {
      const int krow = lane >> 4, lcol = lane & 15;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int e = lane * 4 + reg;
        const double v = ((sh.gp[0][tile][e] + sh.gp[1][tile][e]) + sh.gp[2][tile][e]) + sh.gp[3][tile][e];
        const int row = krow + 4 * reg;
        if (tile == 0) {
          if (row < r && lcol < r) g[row + CALS_GLD * lcol] = v;
        } else if (tile == 1) {
          if (row < r && 16 + lcol < r) {
            g[row + CALS_GLD * (16 + lcol)] = v;
            g[(16 + lcol) + CALS_GLD * row] = v;
          }
        } else {
          if (16 + row < r && 16 + lcol < r) g[(16 + row) + CALS_GLD * (16 + lcol)] = v;
        }
      }
    }
  }
  UPD_STAMP(7);

  if (a.is_last) {
    __syncthreads();
    double t2 = 0.0;
    // (entries of the RMAX x RMAX frame, compile-time divisions; every Gramian load of an entry issued before its use)
    for (int e = tid; e < RMAX * RMAX; e += UPD_THREADS) {
      const int i = e % RMAX, j = e / RMAX;
      if (i < r && j < r) {
        const long long at = i + CALS_GLD * (long long)(col + j);
        double gv[CALS_MAX_MODES];
#pragma unroll
        for (int m = 0; m < CALS_MAX_MODES; ++m) gv[m] = (m < a.n_modes) ? a.gram[m < a.n_modes ? m : 0][at] : 1.0;
        double h = 1.0;
#pragma unroll
        for (int m = 0; m < CALS_MAX_MODES; ++m)
          if (m < a.n_modes) h *= gv[m];
        t2 += lams[i] * lams[j] * h;
      }
    }
    t2 = wave_sum(t2);
    __syncthreads();
    if (lane == 0) sh.redt[wave] = t2;
    __syncthreads();
    if (tid == 0) {
      t2 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
      const double xn = (jkp >= 0) ? a.jk_norms[jkp >> 3] : a.X_norm;
      const double e2 = fmax(xn * xn + t2 - 2.0 * t3, 0.0);
      const double err = sqrt(e2);
      a.mt.err[slot] = err;
      const double of = a.mt.fit[slot];
      a.mt.old_fit[slot] = of;
      a.mt.fit[slot] = 1.0 - fabs(err) / a.X_norm;
      if (a.fin.on) apply_finish_rule(a, slot);
    }
  }
  UPD_END_WAVE();  // the body ends the wave (see UPD_BODY_ATTR): no epilogue that reloads callee-saved registers
}

__device__ __forceinline__ double lane_bcast(double v, int l) {  // l wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// One 16 x 16 tile (bi, bj) of P^T P over rows [row0, row1) as gramian_tile, with the operand loads of eight
// k-steps issued before their MFMAs (the panel is in global memory here: one exposed round trip per eight
// steps instead of one per step).  Trailing steps past row1 multiply zeros: same sums, same order.
template <typename PTR>
__device__ __forceinline__ v4d gramian_tile_b8(PTR panel, int row0, int row1, long long ld, int r, int lane,
                                               int bi, int bj) {
  const int krow = lane >> 4, lcol = lane & 15;
  const bool ciok = (16 * bi + lcol) < r, cjok = (16 * bj + lcol) < r;
  const long long offi = ld * (ciok ? 16 * bi + lcol : 0), offj = ld * (cjok ? 16 * bj + lcol : 0);
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  for (int i0 = row0; i0 < row1; i0 += 32) {
    double pi[8], pj[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 4 * u + krow;
      const int ic = i < row1 ? i : row0;
      pi[u] = (double)panel[ic + offi];
      pj[u] = (double)panel[ic + offj];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool rok = i0 + 4 * u + krow < row1;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((rok && ciok) ? pi[u] : 0.0, (rok && cjok) ? pj[u] : 0.0, acc, 0, 0, 0);
    }
  }
  return acc;
}

// Ranks 33..CALS_RMAX (64) -- since the end of round 3 an A/B alternative only (CALS_HUGE_FROM=49|65 in the
// environment): the update of every rank above CALS_RFAST is the pipeline of huge_* launches below (CALS_HUGE_FROM_DEFAULT).
// Too wide for the register-resident bodies above; H / L (64 columns, ld 66: the
// transposed-copy stores of a wave spread over 16 banks instead of one) live in dynamic LDS and the rows are solved
// through the factor panel itself, one row per thread, SIXTEEN COLUMNS AT A TIME in registers:
//   * dpotf2 on the whole workgroup (thread = row; column j needs sum_{k<j} L[i][k] L[j][k]: own row over the
//     threads, row j broadcast); every L[i][j] is also written TRANSPOSED into the upper triangle of the block, so
//     that both substitutions below read 16 consecutive doubles (ds_read_b128) per step; the Cholesky panels run
//     on one wave;
//   * B := B inv(L^T), left-looking per 16-column block: x_b -= L[b, k] x_k over all earlier columns k in
//     ascending order, then the 16 x 16 triangle -- the operations and their order are dtrsm's
//     Right/Lower/Trans; B := B inv(L) block by block from the right, later columns first, then the triangle
//     (the sums of a column run over the same terms as dtrsm's in a different order: rounding only);
//   * H is padded with the identity up to a multiple of 16 columns, so no step carries a `c < r` guard.
// Same tail as the other bodies (statistics, scaling, Gramian on the matrix cores, error).  After the NNLS
// update (a.rowdot) the panel already holds the solution and only the tail runs.
// (Round 1's body for these ranks unrolled a 64 x 64 guarded substitution per row: 4096 LDS reads and exec-mask
// guards per thread, 0.7 ms per launch at rank 64.  Until round 3 this body, with H / L in a global scratch block,
// also served ranks 65..CALS_GLD on one workgroup: 2.6 ms per mode at rank 256 -- now the huge_* launches below.)
#define UPD_HLDS_LD 66
template <typename T>
static __device__ UPD_BODY_ATTR void update_body_huge(UpdArgsPtr a_ptr, int slot, int r,
                                                           UpdShared &sh) {
  UpdArgsRef a = upd_uniform(a_ptr);  // the calling kernel's argument block (constant memory, scalar loads)
  constexpr int LD = UPD_HLDS_LD;
  constexpr int XB = 16;
  typedef double v2d __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = a.mt.col[slot];
  const long long iters = a.mt.iters[slot];
  const int jkf = (a.mt.jk_mode[slot] == a.mode) ? a.mt.jk_fiber[slot] : -1;
  const int I = a.I;
  const int rp = (r + XB - 1) / XB * XB;
  // [UPD_WAVES][256] partial Gramian tiles
  double *gpb = reinterpret_cast<double *>(upd_dyn) + UPD_HLDS_LD * CALS_RMAX;
  __shared__ double s_piv;
  // the diagonal block of a Cholesky panel: in the Gramian-partials area of UpdShared, idle at that point
  double (*s_dblk)[XB + 1] = reinterpret_cast<double (*)[XB + 1]>(&sh.gp[0][0][0]);
  __shared__ int s_info, s_fail;
  const double *rowdot = a.rowdot ? a.rowdot + (long long)I * blockIdx.x : nullptr;
  const bool solved = rowdot != nullptr;
  UPD_STAMP_H(0);
  double *__restrict__ H = reinterpret_cast<double *>(upd_dyn);  // rp x rp, ld LD

  for (int e = tid; e < (solved ? 0 : rp * rp); e += UPD_THREADS) {
    const int i = e % rp, j = e / rp;
    double h = (i == j) ? 1.0 : 0.0;  // identity padding: L comes out as diag(L_r, I)
    if (i < r && j < r) {
      h = 1.0;
      for (int m = 0; m < a.n_modes; ++m)
        if (m != a.mode) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
    }
    H[i + LD * j] = h;
  }
  if (tid == 0) s_info = 0;
  __threadfence_block();
  __syncthreads();
  UPD_STAMP_H(1);
  // dpotf2 in panels of 16 columns, thread = row (rp <= 256 rows, the padding rows are identity rows):
  //   S[i][c] = H[i][jb + c] - sum_{k < jb} L[i][k] L[jb + c][k]   in registers, four k in flight;
  //   the 16 x 16 diagonal block goes to LDS, where wave 0 factors it column by column (right-looking inside
  //   the block); every row below it then applies the block's columns to its own 16 values.
  // Every entry sees the subtractions of dpotf2 in dpotf2's order (k ascending), so the factor is the unblocked
  // one bit for bit; a non-positive pivot stops at its column with the earlier columns final and the rest of H
  // untouched, exactly the state the column-by-column form leaves (update.cpp:183-185 only logs info).
  for (int jb = 0; jb < (solved ? 0 : rp); jb += XB) {
    const int i = tid;
    const bool act = i >= jb && i < rp;
    double sacc[XB];
#pragma unroll
    for (int c = 0; c < XB; ++c) sacc[c] = 0.0;
    if (act) {
#pragma unroll
      for (int c = 0; c < XB; ++c) sacc[c] = H[i + (long long)LD * (jb + c)];
      for (int k = 0; k < jb; k += 4) {
        double own[4];
        v2d pan[4][XB / 2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          own[u] = H[i + (long long)LD * (k + u)];
          const v2d *hp = reinterpret_cast<const v2d *>(H + jb + (long long)LD * (k + u));  // L[jb + c][k + u]
#pragma unroll
          for (int c = 0; c < XB / 2; ++c) pan[u][c] = hp[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int c = 0; c < XB; c += 2) {
            sacc[c] -= own[u] * pan[u][c >> 1][0];
            sacc[c + 1] -= own[u] * pan[u][c >> 1][1];
          }
        }
      }
      if (i < jb + XB) {
#pragma unroll
        for (int c = 0; c < XB; ++c) s_dblk[i - jb][c] = sacc[c];
      }
    }
    if (tid == 0) s_fail = XB;
    __syncthreads();
    if (wave == 0) {
      // lane = row of the block, the row in registers; column c: the pivot and the finished column travel by
      // v_readlane.  Right-looking inside the block: entry (row, c2) sees the subtractions k = jb, jb + 1, ...
      // in dpotf2's order.  Lanes >= 16 compute on zeros.
      double rv[XB];
#pragma unroll
      for (int c = 0; c < XB; ++c) rv[c] = (lane < XB) ? s_dblk[lane][c] : 0.0;
      int fail = XB;
      double fpiv = 0.0;
#pragma unroll
      for (int c = 0; c < XB; ++c) {
        if (fail == XB) {
          const double piv = lane_bcast(rv[c], c);
          if (!(piv > 0.0)) {
            fail = c;
            fpiv = piv;
          } else {
            const double lcc = sqrt(piv);
            const double l = (lane == c) ? lcc : rv[c] / lcc;
            rv[c] = l;
#pragma unroll
            for (int c2 = c + 1; c2 < XB; ++c2) {
              const double lc2 = lane_bcast(l, c2);  // L[c2][c]
              if (lane > c) rv[c2] -= l * lc2;
            }
          }
        }
      }
      if (lane < XB) {
#pragma unroll
        for (int c = 0; c < XB; ++c) s_dblk[lane][c] = rv[c];
      }
      if (lane == 0 && fail < XB) {
        s_fail = fail;
        s_piv = fpiv;
      }
    }
    __syncthreads();
    const int ncol = s_fail;  // columns of this panel that are final
    if (act) {
      if (i >= jb + XB) {  // below the block: L[i][jb + c] = (S[i][c] - sum_{c' < c} L[i][jb + c'] L[jb + c][jb + c']) / L_cc
#pragma unroll
        for (int c = 0; c < XB; ++c) {
          if (c < ncol) {
            const double l = sacc[c] / s_dblk[c][c];
            sacc[c] = l;
#pragma unroll
            for (int c2 = c + 1; c2 < XB; ++c2) sacc[c2] -= l * s_dblk[c2][c];
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < XB; ++c) sacc[c] = s_dblk[i - jb][c];
      }
#pragma unroll
      for (int c = 0; c < XB; ++c) {
        if (c < ncol && i >= jb + c) {
          H[i + (long long)LD * (jb + c)] = sacc[c];
          if (i > jb + c) H[jb + c + (long long)LD * i] = sacc[c];  // transposed copy: contiguous in c
        }
      }
    }
    if (ncol < XB) {  // info != 0: stop, go on with whatever is in H
      if (tid == 0) {
        H[(jb + ncol) + (long long)LD * (jb + ncol)] = s_piv;
        s_info = jb + ncol + 1;
      }
      break;
    }
    __threadfence_block();
    __syncthreads();
  }
  __syncthreads();
  if (!solved)
    for (int k = tid; k < rp; k += UPD_THREADS) sh.dinv[k] = 1.0 / H[k + LD * k];
  if (tid == 0) a.mt.potrf_info[slot] = s_info;
  __threadfence_block();
  __syncthreads();
  const double *dinv = sh.dinv;
  UPD_STAMP_H(2);

  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  const bool first = (iters == 1);
  double t3 = 0.0;
  // TWO rows per thread (i and i + 256): every L entry a thread fetches serves both, and a mode of 257..512 rows
  // (C3: 300) takes one pass instead of two.  Row slot 1 of a thread without a second row computes on zeros.
  constexpr int NR = 2;
  for (int i0 = tid; i0 < I; i0 += NR * UPD_THREADS) {
    T *x[NR];
    bool ok[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      ok[n] = i0 + n * UPD_THREADS < I;
      x[n] = fac + (ok[n] ? i0 + n * UPD_THREADS : i0);  // row: x[n][I * c]
    }
    if (solved) {
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int i = i0 + n * UPD_THREADS;
        if (!ok[n]) continue;
        if (i != jkf)
          t3 += rowdot[i];
        else
          for (int c = 0; c < r; ++c) x[n][(long long)I * c] = (T)((double)x[n][(long long)I * c] * 0.0);
      }
      continue;
    }
    // B := B * inv(L^T)
    for (int kb = 0; kb < rp; kb += XB) {
      double xb[NR][XB];
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int c = 0; c < XB; ++c)
          xb[n][c] = (ok[n] && kb + c < r) ? (double)x[n][(long long)I * (kb + c)] : 0.0;
      for (int k = 0; k < kb; k += 4) {  // kb is a multiple of 16: whole batches; four columns' loads in flight
        double xk[NR][4];
        v2d h[4][XB / 2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int n = 0; n < NR; ++n) xk[n][u] = ok[n] ? (double)x[n][(long long)I * (k + u)] : 0.0;
          const v2d *hp = reinterpret_cast<const v2d *>(H + kb + (long long)LD * (k + u));  // L[kb + c][k + u]
#pragma unroll
          for (int c = 0; c < XB / 2; ++c) h[u][c] = hp[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int c = 0; c < XB; c += 2) {
#pragma unroll
            for (int n = 0; n < NR; ++n) {
              xb[n][c] -= h[u][c >> 1][0] * xk[n][u];
              xb[n][c + 1] -= h[u][c >> 1][1] * xk[n][u];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        double xk[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          xk[n] = dinv[kb + k] * xb[n][k];
          xb[n][k] = xk[n];
        }
        const v2d *hp = reinterpret_cast<const v2d *>(H + kb + (long long)LD * (kb + k));  // L[kb + c][kb + k]
#pragma unroll
        for (int c = (k + 1) & ~1; c < XB; c += 2) {
          const v2d hh = hp[c >> 1];
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            if (c > k) xb[n][c] -= hh[0] * xk[n];
            xb[n][c + 1] -= hh[1] * xk[n];
          }
        }
      }
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        if (ok[n] && i0 + n * UPD_THREADS != jkf) {
#pragma unroll
          for (int c = 0; c < XB; ++c) t3 += xb[n][c] * xb[n][c];  // padding columns hold exact zeros
        }
#pragma unroll
        for (int c = 0; c < XB; ++c)
          if (ok[n] && kb + c < r) x[n][(long long)I * (kb + c)] = (T)xb[n][c];
      }
    }
    // B := B * inv(L)
    for (int jb = rp - XB; jb >= 0; jb -= XB) {
      double xb[NR][XB];
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int c = 0; c < XB; ++c)
          xb[n][c] = (ok[n] && jb + c < r) ? (double)x[n][(long long)I * (jb + c)] : 0.0;
      // columns jb + 16 .. rp - 1 in batches of four (the padding columns: x = 0 against identity rows)
      for (int k = jb + XB; k < rp; k += 4) {
        double xk[NR][4];
        v2d h[4][XB / 2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int n = 0; n < NR; ++n)
            xk[n][u] = (ok[n] && k + u < r) ? (double)x[n][(long long)I * (k + u)] : 0.0;
          const v2d *hp = reinterpret_cast<const v2d *>(H + jb + (long long)LD * (k + u));  // L[k + u][jb + c], transposed copy
#pragma unroll
          for (int c = 0; c < XB / 2; ++c) h[u][c] = hp[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int c = 0; c < XB; c += 2) {
#pragma unroll
            for (int n = 0; n < NR; ++n) {
              xb[n][c] -= h[u][c >> 1][0] * xk[n][u];
              xb[n][c + 1] -= h[u][c >> 1][1] * xk[n][u];
            }
          }
        }
      }
#pragma unroll
      for (int j = XB - 1; j >= 0; --j) {
        double s_[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) s_[n] = xb[n][j];
        const v2d *hp = reinterpret_cast<const v2d *>(H + jb + (long long)LD * (jb + j));  // L[jb + k][jb + j]
#pragma unroll
        for (int k = (j + 1) & ~1; k < XB; k += 2) {
          const v2d hh = hp[k >> 1];
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            if (k > j) s_[n] -= hh[0] * xb[n][k];
            s_[n] -= hh[1] * xb[n][k + 1];
          }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) xb[n][j] = dinv[jb + j] * s_[n];
      }
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int c = 0; c < XB; ++c)
          if (ok[n] && jb + c < r)
            x[n][(long long)I * (jb + c)] = (T)((i0 + n * UPD_THREADS == jkf) ? xb[n][c] * 0.0 : xb[n][c]);
    }
  }
  t3 = wave_sum(t3);
  if (lane == 0) sh.redt[wave] = t3;
  __threadfence_block();
  __syncthreads();
  t3 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
  UPD_STAMP_H(3);

  for (int c = wave; c < r; c += UPD_WAVES) {  // column scales (Ktensor::normalize(mode, iteration))
    const T *cp = fac + (long long)I * c;
    double lam;
    if (first) {
      double ss = 0.0;
      for (int i = lane; i < I; i += 64) {
        const double x = (double)cp[i];
        ss += x * x;
      }
      lam = sqrt(wave_sum(ss));
    } else {
      double m = -1.0, v = 0.0;
      int ix = 0x7fffffff;
      for (int i = lane; i < I; i += 64) {
        const double x = (double)cp[i];
        const double ax = fabs(x);
        if (ax > m) {
          m = ax;
          v = x;
          ix = i;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double m2 = __shfl_xor(m, off);
        const double v2 = __shfl_xor(v, off);
        const int i2 = __shfl_xor(ix, off);
        const bool take = (m2 > m) || (m2 == m && i2 < ix);
        m = take ? m2 : m;
        v = take ? v2 : v;
        ix = take ? i2 : ix;
      }
      lam = v;
    }
    if (lane == 0) {
      sh.lams[c] = lam;
      a.lambda[col + c] = lam;
    }
  }
  __syncthreads();
  UPD_STAMP_H(4);
  const double *lams = sh.lams;
  for (int c = 0; c < r; ++c) {
    const double lam = lams[c];
    if (lam != 0.0)
      for (int i = tid; i < I; i += UPD_THREADS)
        fac[i + (long long)I * c] = (T)((1.0 / lam) * (double)fac[i + (long long)I * c]);
  }
  __threadfence_block();
  __syncthreads();

  UPD_STAMP_H(5);
  {  // update_gramian, one 16 x 16 tile pair at a time, rows split over the waves
    double *g = a.gram[a.mode] + CALS_GLD * (long long)col;
    const int nt = (r + 15) >> 4;
    const int chunk = ((I + UPD_WAVES - 1) / UPD_WAVES + 3) / 4 * 4;
    const int row0 = wave * chunk, row1 = min(I, row0 + chunk);
    const int krow = lane >> 4, lcol = lane & 15;
    for (int bi = 0; bi < nt; ++bi)
      for (int bj = bi; bj < nt; ++bj) {
        const v4d t = gramian_tile_b8((const T *)fac, row0, row1, (long long)I, r, lane, bi, bj);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) gpb[wave * 256 + lane * 4 + reg] = t[reg];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int e = lane * 4 + reg;
            const double v = ((gpb[e] + gpb[256 + e]) + gpb[512 + e]) + gpb[768 + e];
            const int row = 16 * bi + krow + 4 * reg, cc = 16 * bj + lcol;
            if (row < r && cc < r) {
              g[row + CALS_GLD * cc] = v;
              g[cc + CALS_GLD * row] = v;
            }
          }
        }
        __syncthreads();
      }
  }

  UPD_STAMP_H(6);
  if (a.is_last) {
    __threadfence_block();
    __syncthreads();
    double t2 = 0.0;
    for (int e = tid; e < r * r; e += UPD_THREADS) {
      const int i = e % r, j = e / r;
      double h = 1.0;
      for (int m = 0; m < a.n_modes; ++m) h *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
      t2 += lams[i] * lams[j] * h;
    }
    t2 = wave_sum(t2);
    __syncthreads();
    if (lane == 0) sh.redt[wave] = t2;
    __syncthreads();
    if (tid == 0) {
      t2 = sh.redt[0] + sh.redt[1] + sh.redt[2] + sh.redt[3];
      const int jm = a.mt.jk_mode[slot];
      const double xn = (jm >= 0) ? a.jk_norms[a.mt.jk_fiber[slot]] : a.X_norm;
      const double e2 = fmax(xn * xn + t2 - 2.0 * t3, 0.0);
      const double err = sqrt(e2);
      a.mt.err[slot] = err;
      const double of = a.mt.fit[slot];
      a.mt.old_fit[slot] = of;
      a.mt.fit[slot] = 1.0 - fabs(err) / a.X_norm;
      if (a.fin.on) apply_finish_rule(a, slot);
    }
  }
  UPD_END_WAVE();  // the body ends the wave (see UPD_BODY_ATTR): no epilogue that reloads callee-saved registers
}

// Three kernels, one per body class, each launched only when its class is in flight (update_launch); the
// workgroups of the other classes' models return at once.  One kernel dispatching to all sixteen bodies gave every
// rank-1..20 model the register budget and call-frame scratch of the rank-33..256 body (458 unified registers,
// 484 B of scratch per lane: one workgroup per CU).
//   update_lds_kernel   ranks <= 32, the panel LDS resident: __launch_bounds__(256, 2) = at most 256 registers, so
//                       two workgroups share a CU wherever the LDS allows it (C4: 512 models on 256 CUs)
//   update_hbm_kernel   ranks <= 32, modes too tall for LDS: the bodies that go through HBM between the passes
//   update_huge_kernel  ranks 33..256
#define UPD_DISPATCH(BODY)                      \
  do {                                          \
    if (r <= 4)                                 \
      BODY<4, T>(a_ptr, slot, r, col, jkp, sh);     \
    else if (r <= 8)                            \
      BODY<8, T>(a_ptr, slot, r, col, jkp, sh);     \
    else if (r <= 12)                           \
      BODY<12, T>(a_ptr, slot, r, col, jkp, sh);    \
    else if (r <= 16)                           \
      BODY<16, T>(a_ptr, slot, r, col, jkp, sh);    \
    else if (r <= 20)                           \
      BODY<20, T>(a_ptr, slot, r, col, jkp, sh);    \
    else if (r <= 24)                           \
      BODY<24, T>(a_ptr, slot, r, col, jkp, sh);    \
    else                                        \
      BODY<32, T>(a_ptr, slot, r, col, jkp, sh);    \
  } while (0)

template <typename T>
__global__ void __launch_bounds__(UPD_THREADS, 2) update_lds_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  UpdArgsPtr a_ptr = upd_kernargs();
  UpdArgsRef a = *a_ptr;
  __shared__ UpdShared sh;
  // {slot, first column, rank, jackknife (mode, fiber) packed} of this workgroup's model in ONE load (UpdateArgs::
  // wgdesc, written by the engine with the slot list): slot -> rank -> column were three dependent round trips in
  // front of the first Gramian load
#ifdef CALS_DIAG
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const int4 d = a.wgdesc[blockIdx.x];
  const int slot = d.x, col = d.y, r = d.z, jkp = d.w;
  if (r > CALS_RFAST) return;
#ifdef CALS_DIAG  // kernel entry and "descriptor arrived" next to the body's phase stamps (tools/update_trace.py)
  if (a.dbg_trace && a.mode == 0 && r == 20 && blockIdx.x < 24 && threadIdx.x == 0) {
    a.dbg_trace[8] = t_entry;
    a.dbg_trace[9] = __builtin_amdgcn_s_memtime();
  }
#endif
  UPD_DISPATCH(update_body_lds);
}

template <typename T>
__global__ void __launch_bounds__(UPD_THREADS, 2) update_hbm_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  UpdArgsPtr a_ptr = upd_kernargs();
  UpdArgsRef a = *a_ptr;
  __shared__ UpdShared sh;
  // {slot, first column, rank, jackknife (mode, fiber) packed} of this workgroup's model in ONE load (UpdateArgs::
  // wgdesc, written by the engine with the slot list): slot -> rank -> column were three dependent round trips in
  // front of the first Gramian load
  const int4 d = a.wgdesc[blockIdx.x];
  const int slot = d.x, col = d.y, r = d.z, jkp = d.w;
  if (r > CALS_RFAST) return;
  UPD_DISPATCH(update_body);
}

template <typename T>
__global__ void __launch_bounds__(UPD_THREADS, 1) update_huge_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  UpdArgsPtr a_ptr = upd_kernargs();
  UpdArgsRef a = *a_ptr;
  __shared__ UpdShared sh;
  const int slot = a.slots[blockIdx.x];
  const int r = a.mt.rank[slot];
  const int huge_from = a.huge_from ? a.huge_from : CALS_RMAX + 1;
  if (r <= CALS_RFAST || r > CALS_RMAX || r >= huge_from) return;  // the huge_* launches' models (update_launch)
  update_body_huge<T>(a_ptr, slot, r, sh);
}
#undef UPD_DISPATCH

// ---------------------------------------------------------------------------------------------
// Ranks 33..CALS_GLD (round 3): the update of such a model as a PIPELINE OF LAUNCHES, each spread over as many
// workgroups as its step has independent pieces.  update_body_huge<T, false> ran the whole update on ONE workgroup:
// 2.6 ms per mode for a rank-256 model at C3's shape (Hadamard 0.22, Cholesky 0.42, row solves 1.23 -- 300 rows on
// 256 threads, 131 k dependent FMAs each --, scales 0.18, Gramian 0.57), all of it on the critical path of a sweep.
//   huge_hadamard_kernel  H = hadamard of the other modes' Gramians, identity padded   wave = column
//   huge_potrf_kernel     dpotf2 in panels of 16 columns (the code of the one-workgroup body)   one workgroup
//   huge_solve_kernel     B := B inv(L^T) inv(L) on the matrix cores                   wave = 16 rows
//   huge_scale_kernel     column scales + normalisation                                wave = column
//   huge_gram_kernel      update_gramian                                               workgroup = one 16 x 16 tile pair
//   huge_error_kernel     fast error + end-of-sweep rule (last mode only)              one workgroup
// Model h of the class (UpdateArgs::huge_idx) owns block h of hscratch -- no counter.  After the NNLS update
// (a.rowdot) the factor and the solve launches are skipped, as the `solved` path of the other bodies.
// ---------------------------------------------------------------------------------------------
#ifdef CALS_DIAG  // phase cycle sums of workgroup (0, 0) at mode 0 into the trace (tools/update_trace_huge.py)
#define HUGE_CLK_DECL(N)                     \
  unsigned long long hclk[N] = {};           \
  unsigned long long hclk_t = __builtin_amdgcn_s_memtime()
#define HUGE_CLK(k)                                                \
  do {                                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    hclk[k] += now_ - hclk_t;                                      \
    hclk_t = now_;                                                 \
  } while (0)
#define HUGE_CLK_DUMP(N, at)                                                                       \
  do {                                                                                             \
    if (a.dbg_trace && a.mode == 0 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)      \
      for (int q_ = 0; q_ < N; ++q_) a.dbg_trace[(at) + q_] = hclk[q_];                            \
  } while (0)
#else
#define HUGE_CLK_DECL(N) do { } while (0)
#define HUGE_CLK(k) do { } while (0)
#define HUGE_CLK_DUMP(N, at) do { } while (0)
#endif
#define HUGE_PROLOGUE()                                                      \
  UpdArgsRef a = *upd_kernargs();                                            \
  const int h = blockIdx.y;                                                  \
  const int k_model = a.huge_idx[h];                                         \
  const int slot = a.slots[k_model];                                         \
  const int r = a.mt.rank[slot], col = a.mt.col[slot];                       \
  const int rp = (r + 15) & ~15;                                             \
  const int I = a.I;                                                         \
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;             \
  (void)k_model; (void)rp; (void)I; (void)lane; (void)wave; (void)col

__global__ void __launch_bounds__(256) huge_hadamard_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  double *__restrict__ H = a.hscratch + (long long)h * CALS_GLD * CALS_GLD;
  for (int j = blockIdx.x * 4 + wave; j < rp; j += gridDim.x * 4)
    for (int i = lane; i < rp; i += 64) {
      double v = (i == j) ? 1.0 : 0.0;  // identity padding: L comes out as diag(L_r, I)
      if (i < r && j < r) {
        v = 1.0;
        for (int m = 0; m < a.n_modes; ++m)
          if (m != a.mode) v *= a.gram[m][i + CALS_GLD * (long long)(col + j)];
      }
      H[i + CALS_GLD * j] = v;
    }
}

// dpotrf, lower, in panels of 16 columns (rp <= 256 rows, the padding rows are identity rows), left-looking:
//   (a) S = H[jb :, jb : jb + 16] - L[jb :, 0 : jb] L[jb : jb + 16, 0 : jb]^T  one 16 x 16 tile per wavefront at a time
//       as jb / 4 chained v_mfma_f64_16x16x4 (both operands straight from the L2-resident block: column-major L,
//       16 consecutive doubles per k), the tiles into LDS;
//   (b) wave 0 factors the diagonal tile column by column (right-looking inside the tile: lane = row, the row in
//       registers, pivot and finished column by v_readlane);
//   (c) thread = row below the tile applies the tile's columns to its 16 values (dtrsm) and writes the finished
//       columns -- also TRANSPOSED into the upper triangle, so that (a) and both substitutions of the solve read 16
//       consecutive doubles per step.
// The sums of an entry run over dpotf2's terms, k ascending, four per matrix instruction (the one-workgroup body took
// them one by one: 0.42 ms at rank 256 against 0.07 here).  A non-positive pivot stops at its column with the earlier
// columns final and the rest of H untouched -- the state the column-by-column form leaves (update.cpp:183-185 only
// logs info).
#define HUGE_POTRF_THREADS 512
__global__ void __launch_bounds__(HUGE_POTRF_THREADS) huge_potrf_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  constexpr int XB = 16, LD = CALS_GLD, W = HUGE_POTRF_THREADS / 64;
  double *__restrict__ H = a.hscratch + (long long)h * CALS_GLD * CALS_GLD;
  __shared__ double s_tile[CALS_GLD / XB][XB][XB + 1];  // the panel's S tiles; tile 0 = the diagonal block
  __shared__ double s_piv, s_rdiag[XB];
  __shared__ int s_info, s_fail;
  const int g = lane >> 4, n = lane & 15;
  if (tid == 0) s_info = 0;
  __syncthreads();
  HUGE_CLK_DECL(3);
  for (int jb = 0; jb < rp; jb += XB) {
    const int ntile = (rp - jb) / XB;
    for (int t = wave; t < ntile; t += W) {  // (a)
      const int row0 = jb + XB * t;
      v4d acc;
      // H is symmetric and columns >= jb of rows >= jb are still H's: the transposed entry, contiguous over the lanes
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = H[(jb + n) + (long long)LD * (row0 + g + 4 * v)];
      const double *ap = H + row0 + n + (long long)LD * g;  // L[row0 + n][k + g]
      const double *bp = H + jb + n + (long long)LD * g;    // L[jb + n][k + g]
      // segments of 28 instructions (112 columns): ALL operand loads of a segment go out before its first
      // instruction -- one L2 round trip per segment (a chunk-by-chunk pipeline exposed part of every chunk's: 130 us
      // of this kernel's 240 at rank 256)
      for (int ks = 0; ks < jb; ks += 112) {
        double av[28], bv[28];
#pragma unroll
        for (int u = 0; u < 28; ++u) {
          const bool in = ks + 4 * (u & ~3) < jb;  // jb is a multiple of 16: whole groups of four
          av[u] = in ? -ap[(long long)LD * (ks + 4 * u)] : 0.0;
          bv[u] = in ? bp[(long long)LD * (ks + 4 * u)] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 28; ++u)
          if (ks + 4 * (u & ~3) < jb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) s_tile[t][g + 4 * v][n] = acc[v];
    }
    if (tid == 0) s_fail = XB;
    __syncthreads();
    HUGE_CLK(0);
    if (wave == 0) {  // (b) lanes >= 16 compute on zeros
      double rv[XB];
#pragma unroll
      for (int c = 0; c < XB; ++c) rv[c] = (lane < XB) ? s_tile[0][lane][c] : 0.0;
      int fail = XB;
      double fpiv = 0.0;
#pragma unroll
      for (int c = 0; c < XB; ++c) {
        if (fail == XB) {
          const double piv = lane_bcast(rv[c], c);
          if (!(piv > 0.0)) {
            fail = c;
            fpiv = piv;
          } else {
            // 1 / sqrt(pivot): v_rsq_f64 + two Newton steps, the column scaled by it (dpotf2 scales by ONE / AJJ too;
            // an IEEE sqrt and sixteen IEEE divides per column were 8.5 k cycles per tile, all of it serial)
            double rl = __builtin_amdgcn_rsq(piv);
            rl = rl * fma(-0.5 * piv * rl, rl, 1.5);
            rl = rl * fma(-0.5 * piv * rl, rl, 1.5);
            const double l = (lane == c) ? piv * rl : rv[c] * rl;
            rv[c] = l;
            if (lane == 0) s_rdiag[c] = rl;
#pragma unroll
            for (int c2 = c + 1; c2 < XB; ++c2) {
              const double lc2 = lane_bcast(l, c2);  // L[c2][c]
              if (lane > c) rv[c2] -= l * lc2;
            }
          }
        }
      }
      if (lane < XB) {
#pragma unroll
        for (int c = 0; c < XB; ++c) s_tile[0][lane][c] = rv[c];
      }
      if (lane == 0 && fail < XB) {
        s_fail = fail;
        s_piv = fpiv;
      }
    }
    __syncthreads();
    HUGE_CLK(1);
    const int ncol = s_fail;  // columns of this panel that are final
    const int i = jb + tid;   // (c)
    if (i < rp) {
      double sacc[XB];
#pragma unroll
      for (int c = 0; c < XB; ++c) sacc[c] = s_tile[tid >> 4][tid & 15][c];
      if (tid >= XB) {  // below the tile: L[i][jb + c] = (S[i][c] - sum_{c' < c} L[i][jb + c'] L[jb + c][jb + c']) / L_cc
#pragma unroll
        for (int c = 0; c < XB; ++c) {
          if (c < ncol) {
            const double l = sacc[c] * s_rdiag[c];  // (dtrsm scales by ONE / A(K,K) as well)
            sacc[c] = l;
#pragma unroll
            for (int c2 = c + 1; c2 < XB; ++c2) sacc[c2] -= l * s_tile[0][c2][c];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < XB; ++c) {
        if (c < ncol && i >= jb + c) {
          H[i + (long long)LD * (jb + c)] = sacc[c];
          if (i > jb + c) H[jb + c + (long long)LD * i] = sacc[c];  // transposed copy: contiguous in c
        }
      }
    }
    if (ncol < XB) {  // info != 0: stop, go on with whatever is in H
      if (tid == 0) {
        H[(jb + ncol) + (long long)LD * (jb + ncol)] = s_piv;
        s_info = jb + ncol + 1;
      }
      break;
    }
    __threadfence_block();
    __syncthreads();
    HUGE_CLK(2);
  }
  __syncthreads();
  if (tid == 0) a.mt.potrf_info[slot] = s_info;
  HUGE_CLK_DUMP(3, 0);  // CALS_DIAG: cycles in (a) tiles, (b) diagonal tile, (c) rows below + stores
}

// B := B inv(L^T) inv(L) for 16 rows of the factor per wavefront, the slab (16 x rp doubles) LDS resident.
// Per block of 16 columns, forward:  S = B_blk - Z[:, 0 : kb] L[kb : kb + 16, 0 : kb]^T  as kb / 4 chained
// v_mfma_f64_16x16x4 (A = -Z from the slab, B = the block's 16 rows of L), then the 16 x 16 triangle by substitution,
// one lane per row (the operations and their order are dtrsm's Right/Lower/Trans; the sums of the block part run four
// k per matrix instruction).  Backward (B inv(L)) the same from the right, with the transposed copy of L.
// The rows of L a block needs -- [k][16] doubles, k over the block's columns incl. its own triangle -- arrive by
// LDS-DMA (global_load_lds_dwordx4: eight k per instruction, each 16 consecutive doubles of the L2-resident factor),
// issued ONE BLOCK AHEAD into the other half of a double buffer: they do not depend on the slab, and a lone wave on
// its CU has nothing else to cover an L2 round trip (with register loads in front of their instructions this kernel
// took 173 us at rank 256, three quarters of it waiting; the compiler's counters cannot look across the blocks).
// Z stays fp64 in the slab between the blocks (the one-workgroup body rounded it through the factor's storage type).
// The rows' <z, z> = <x, g> go to hrowdot for the error; the jackknife row is zeroed by the scale launch.
#ifndef GLOBAL_AS
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
#endif
// (one wavefront per workgroup: LDS traffic of a wave is ordered, a wavefront-scope fence is all the steps need)
#define HUGE_WAVE_SYNC()                                      \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)
#define HUGE_SOLVE_LDS_DOUBLES(rp) (16 * ((rp) + 4) + CALS_GLD + 2 * 16 * (rp))  // slab, dinv, two panels of <= rp rows of L
template <typename T>
__global__ void __launch_bounds__(64) huge_solve_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  constexpr int LD = CALS_GLD;
  const int PANEL = 16 * rp;  // a block's panel has at most rp rows of L (the launch sizes the LDS for the largest rp in flight)
  const double *__restrict__ H = a.hscratch + (long long)h * CALS_GLD * CALS_GLD;
  const int g = lane >> 4, n = lane & 15;
  const int row0 = blockIdx.x * 16;
  if (row0 >= I) return;
  const int LDZ = rp + 4;  // rows four 8-byte banks apart: the 64 lanes of an A-operand read (row n, column k + g) hit every bank twice
  double *slab = reinterpret_cast<double *>(upd_dyn);  // [16][LDZ]
  double *dinv = slab + 16 * LDZ;                       // [rp]
  double *Lp = dinv + CALS_GLD;                         // [2][PANEL]: the block's rows of L, [k][16]
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  const bool rok = row0 + n < I;
  // rows k0 .. k1 - 1 of H[(blk + 0..15) + LD k] -> Lp[buf][(k - k0)][16]; lane: 16 bytes = doubles 2 (lane & 7), +1 of k0 + lane / 8
  const double *hsrc = H + 2 * (lane & 7) + (long long)LD * (lane >> 3);
  auto issue_panel = [&](int blk, int k0, int k1, int buf) {
    for (int k = k0; k < k1; k += 8)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(hsrc + blk + (long long)LD * k),
                                       (LDS_AS void *)(Lp + buf * PANEL + (k - k0) * 16), 16, 0, 0);
  };
  issue_panel(0, 0, 16, 0);
  for (int c0 = 0; c0 < rp; c0 += 64) {  // (rp is a multiple of 16: batches of 16 loads in flight, not one round trip each)
    double tmp[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int c = c0 + 4 * u + g;
      tmp[u] = (rok && c < r) ? (double)fac[row0 + n + (long long)I * c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (c0 + 4 * u < rp) slab[n * LDZ + c0 + 4 * u + g] = tmp[u];
  }
  {
    double dv[CALS_GLD / 64];
#pragma unroll
    for (int u = 0; u < CALS_GLD / 64; ++u) dv[u] = (lane + 64 * u < rp) ? H[(lane + 64 * u) * (long long)(LD + 1)] : 1.0;
#pragma unroll
    for (int u = 0; u < CALS_GLD / 64; ++u)
      if (lane + 64 * u < rp) dinv[lane + 64 * u] = 1.0 / dv[u];
  }
  double rd = 0.0;
  int buf = 0;
  HUGE_CLK_DECL(5);
  const double *zp = slab + n * LDZ + g;  // A operand: Z[row n][k + g]
  // the block part: acc -= Z[:, k0 : k1] Lblk^T, Lblk's rows at lp[(k - k0) * 16]; the operands of the next four
  // instructions are read while the current four run
  auto gemm = [&](v4d acc, const double *lp, int k0, int k1) {
    double av[2][4], bv[2][4];
    auto rd4 = [&](int k, int q) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[q][u] = -zp[k + 4 * u];
        bv[q][u] = lp[(k - k0 + 4 * u) * 16];
      }
    };
    if (k0 < k1) rd4(k0, 0);
    for (int k = k0; k < k1; k += 32) {
      if (k + 16 < k1) rd4(k + 16, 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0][u], bv[0][u], acc, 0, 0, 0);
      if (k + 32 < k1) rd4(k + 32, 0);
      if (k + 16 < k1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1][u], bv[1][u], acc, 0, 0, 0);
      }
    }
    return acc;  // (two alternating accumulators measured no faster: the chain of dependent instructions is not the limit)
  };
  for (int kb = 0; kb < rp; kb += 16, buf ^= 1) {  // B := B inv(L^T); panel = rows 0 .. kb + 15 of L[kb + n][k]
    HUGE_CLK(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HUGE_WAVE_SYNC();
    HUGE_CLK(0);
    if (kb + 16 < rp) issue_panel(kb + 16, 0, kb + 32, buf ^ 1);
    HUGE_CLK(1);
    const double *lp = Lp + buf * PANEL;
    v4d acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = slab[(g + 4 * v) * LDZ + kb + n];
    acc = gemm(acc, lp + g * 16 + n, 0, kb);
#pragma unroll
    for (int v = 0; v < 4; ++v) slab[(g + 4 * v) * LDZ + kb + n] = acc[v];
    HUGE_WAVE_SYNC();
    HUGE_CLK(2);
    if (lane < 16) {
      // lane = row of the slab AND row of the triangle: L[kb + lane][kb + c] in dr[c]; the entry a step needs is the
      // same for every row -- it travels by v_readlane into an SGPR pair (as 136 broadcast LDS reads, each in front of
      // its FMA, the triangle was three quarters of this kernel)
      double s[16], dr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        s[c] = slab[lane * LDZ + kb + c];
        dr[c] = lp[(kb + c) * 16 + lane];
      }
      const double dvl = dinv[kb + lane];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double z = lane_bcast(dvl, c) * s[c];
        s[c] = z;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) s[c2] -= lane_bcast(dr[c], c2) * z;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        rd += s[c] * s[c];  // padding columns hold exact zeros
        slab[lane * LDZ + kb + c] = s[c];
      }
    }
    HUGE_WAVE_SYNC();
    HUGE_CLK(3);
  }
  if (lane < 16 && row0 + lane < I) a.hrowdot[(long long)I * h + row0 + lane] = rd;
  issue_panel(rp - 16, rp - 16, rp, buf);
  for (int jb = rp - 16; jb >= 0; jb -= 16, buf ^= 1) {  // B := B inv(L); panel = rows jb .. rp - 1 of the transposed copy
    HUGE_CLK(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HUGE_WAVE_SYNC();
    HUGE_CLK(0);
    if (jb >= 16) issue_panel(jb - 16, jb - 16, rp, buf ^ 1);
    HUGE_CLK(1);
    const double *lp = Lp + buf * PANEL;
    v4d acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = slab[(g + 4 * v) * LDZ + jb + n];
    acc = gemm(acc, lp + (16 + g) * 16 + n, jb + 16, rp);
#pragma unroll
    for (int v = 0; v < 4; ++v) slab[(g + 4 * v) * LDZ + jb + n] = acc[v];
    HUGE_WAVE_SYNC();
    HUGE_CLK(2);
    if (lane < 16) {
      double s[16], dr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        s[c] = slab[lane * LDZ + jb + c];
        dr[c] = lp[c * 16 + lane];  // L[jb + lane][jb + c]
      }
      const double dvl = dinv[jb + lane];
#pragma unroll
      for (int j = 15; j >= 0; --j) {
        double sj = s[j];
#pragma unroll
        for (int k = j + 1; k < 16; ++k) sj -= lane_bcast(dr[j], k) * s[k];  // L[jb + k][jb + j]
        s[j] = lane_bcast(dvl, j) * sj;
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) slab[lane * LDZ + jb + c] = s[c];
    }
    HUGE_WAVE_SYNC();
    HUGE_CLK(3);
  }
  for (int c0 = 0; c0 < rp; c0 += 4) {
    const int c = c0 + g;
    if (rok && c < r) fac[row0 + n + (long long)I * c] = (T)slab[n * LDZ + c];
  }
  HUGE_CLK_DUMP(5, 8);  // CALS_DIAG: cycles in {wait for the panel, DMA issue, block part, triangle, rest}
}

// Column scales (Ktensor::normalize(mode, iteration)) and the normalisation, one wavefront per column; the jackknife
// row is zeroed first (x * 0.0, as the other bodies: NaN / Inf stay visible).
template <typename T>
__global__ void __launch_bounds__(256) huge_scale_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  const int jkf = (a.mt.jk_mode[slot] == a.mode) ? a.mt.jk_fiber[slot] : -1;
  const bool first = (a.mt.iters[slot] == 1);
  T *fac = static_cast<T *>(a.factor) + (long long)I * col;
  for (int c = blockIdx.x * 4 + wave; c < r; c += gridDim.x * 4) {
    T *cp = fac + (long long)I * c;
    if (jkf >= 0 && jkf < I && lane == (jkf & 63)) cp[jkf] = (T)((double)cp[jkf] * 0.0);  // (this lane re-reads it below)
    double lam;
    if (first) {
      double ss = 0.0;
      for (int i = lane; i < I; i += 64) {
        const double x = (double)cp[i];
        ss += x * x;
      }
      lam = sqrt(wave_sum(ss));
    } else {
      double m = -1.0, v = 0.0;
      int ix = 0x7fffffff;
      for (int i = lane; i < I; i += 64) {
        const double x = (double)cp[i];
        const double ax = fabs(x);
        if (ax > m) {
          m = ax;
          v = x;
          ix = i;
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double m2 = __shfl_xor(m, off);
        const double v2 = __shfl_xor(v, off);
        const int i2 = __shfl_xor(ix, off);
        const bool take = (m2 > m) || (m2 == m && i2 < ix);
        m = take ? m2 : m;
        v = take ? v2 : v;
        ix = take ? i2 : ix;
      }
      lam = v;
    }
    if (lane == 0) a.lambda[col + c] = lam;
    if (lam != 0.0)
      for (int i = lane; i < I; i += 64) cp[i] = (T)((1.0 / lam) * (double)cp[i]);
  }
}

// update_gramian: one 16 x 16 tile pair (bi <= bj) per workgroup, rows split over the waves and summed in wave
// order -- the one-workgroup body's sums, tile by tile.
template <typename T>
__global__ void __launch_bounds__(256) huge_gram_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  const int nt = (r + 15) >> 4;
  int bi = 0, p = blockIdx.x;
  while (bi < nt && p >= nt - bi) {
    p -= nt - bi;
    ++bi;
  }
  if (bi >= nt) return;
  const int bj = bi + p;
  const T *fac = static_cast<const T *>(a.factor) + (long long)I * col;
  __shared__ double gpb[UPD_WAVES * 256];
  const int chunk = ((I + UPD_WAVES - 1) / UPD_WAVES + 3) / 4 * 4;
  const int row0 = wave * chunk, row1 = min(I, row0 + chunk);
  const int krow = lane >> 4, lcol = lane & 15;
  const v4d t = gramian_tile_b8(fac, row0, row1, (long long)I, r, lane, bi, bj);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) gpb[wave * 256 + lane * 4 + reg] = t[reg];
  __syncthreads();
  if (wave == 0) {
    double *gm = a.gram[a.mode] + CALS_GLD * (long long)col;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int e = lane * 4 + reg;
      const double v = ((gpb[e] + gpb[256 + e]) + gpb[512 + e]) + gpb[768 + e];
      const int row = 16 * bi + krow + 4 * reg, cc = 16 * bj + lcol;
      if (row < r && cc < r) {
        gm[row + CALS_GLD * cc] = v;
        gm[cc + CALS_GLD * row] = v;
      }
    }
  }
}

// Fast error of the last mode (error::compute_fast_error) + the end-of-sweep rule.  t2 = sum_ij lam_i lam_j prod_m
// G_m[i][j]: wave = column j, lanes over i (coalesced, no index division; as e = tid, tid + 256, ... with i = e % r,
// j = e / r on 256 threads this launch took 0.24 ms at rank 256).
#define HUGE_ERR_THREADS 1024
__global__ void __launch_bounds__(HUGE_ERR_THREADS) huge_error_kernel(const UpdateArgs a_by_value) {
  (void)a_by_value;
  HUGE_PROLOGUE();
  constexpr int W = HUGE_ERR_THREADS / 64;
  const int jkf = (a.mt.jk_mode[slot] == a.mode) ? a.mt.jk_fiber[slot] : -1;
  const double *rowdot = a.rowdot ? a.rowdot + (long long)I * k_model : a.hrowdot + (long long)I * h;
  __shared__ double redt[W];
  double t3 = 0.0;
  for (int i = tid; i < I; i += HUGE_ERR_THREADS)
    if (i != jkf) t3 += rowdot[i];
  t3 = wave_sum(t3);
  if (lane == 0) redt[wave] = t3;
  __syncthreads();
  t3 = 0.0;
#pragma unroll
  for (int w = 0; w < W; ++w) t3 += redt[w];
  __syncthreads();
  double t2 = 0.0;
  {
    constexpr int NQ = CALS_GLD / 64, JB = 4;  // four columns' loads in flight (a round trip per column and mode otherwise)
    double li[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) li[q] = (lane + 64 * q < r) ? a.lambda[col + lane + 64 * q] : 0.0;
    for (int j0 = wave; j0 < r; j0 += W * JB) {
      double hh[JB][NQ];
#pragma unroll
      for (int b = 0; b < JB; ++b)
#pragma unroll
        for (int q = 0; q < NQ; ++q) hh[b][q] = 1.0;
      for (int m = 0; m < a.n_modes; ++m) {
        const double *gp = a.gram[m] + CALS_GLD * (long long)col + lane;
#pragma unroll
        for (int b = 0; b < JB; ++b) {
          const int j = j0 + W * b;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            hh[b][q] *= (j < r && lane + 64 * q < r) ? gp[CALS_GLD * (long long)j + 64 * q] : 0.0;
        }
      }
#pragma unroll
      for (int b = 0; b < JB; ++b) {
        const int j = j0 + W * b;
        const double lj = (j < r) ? a.lambda[col + j] : 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) t2 += li[q] * lj * hh[b][q];
      }
    }
  }
  t2 = wave_sum(t2);
  if (lane == 0) redt[wave] = t2;
  __syncthreads();
  if (tid == 0) {
    t2 = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) t2 += redt[w];
    const int jm = a.mt.jk_mode[slot];
    const double xn = (jm >= 0) ? a.jk_norms[a.mt.jk_fiber[slot]] : a.X_norm;
    const double e2 = fmax(xn * xn + t2 - 2.0 * t3, 0.0);
    const double err = sqrt(e2);
    a.mt.err[slot] = err;
    const double of = a.mt.fit[slot];
    a.mt.old_fit[slot] = of;
    a.mt.fit[slot] = 1.0 - fabs(err) / a.X_norm;
    if (a.fin.on) apply_finish_rule(a, slot);
  }
}
#undef HUGE_PROLOGUE

// G[i, c] = sum_t partial[(c / 128) * T + t][i, c % 128], t = 0..T-1 in this fixed order
// (deterministic split-K reduction of the MTTKRP, summed in fp64), written into the multi-factor
// of the mode.
template <typename E>
__global__ void __launch_bounds__(256) reduce_partials_kernel(const E *partial, int T, int ldPart,
                                                              int I, E *factor) {
  const int c = blockIdx.y;
  const long long tile = (long long)ldPart * CALS_BN;
  const E *base = partial + (long long)(c >> 7) * T * tile + (long long)ldPart * (c & (CALS_BN - 1));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < I; i += gridDim.x * blockDim.x) {
    const E *p = base + i;
    double s = 0.0;
    int t = 0;
    // sixteen tiles' loads in flight, added in tile order (a thread walks the T tiles serially: with four in flight
    // C2's team of 51 was 13 exposed round trips, 6.7 us per launch)
    for (; t + 16 <= T; t += 16) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = (double)p[(t + k) * tile];
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[k];
    }
    for (; t + 4 <= T; t += 4) {
      const double v0 = p[(t + 0) * tile], v1 = p[(t + 1) * tile], v2 = p[(t + 2) * tile],
                   v3 = p[(t + 3) * tile];
      s += v0;
      s += v1;
      s += v2;
      s += v3;
    }
    for (; t < T; ++t) s += (double)p[t * tile];
    factor[i + (long long)I * c] = (E)s;
  }
}

hipError_t reduce_partials_launch(const void *partial, int T, int ldPart, int I, int R,
                                  void *factor, int dtype, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  const dim3 grid((I + 255) / 256, R), block(256);
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(reduce_partials_kernel<float>, grid, block, 0, st, (const float *)partial, T,
                       ldPart, I, (float *)factor);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<double>, grid, block, 0, st, (const double *)partial,
                       T, ldPart, I, (double *)factor);
  return hipGetLastError();
}

// Same reduction for a COMPACT set of columns: column k of the partial tiles goes to column idx[k] of
// the multi-factor (the stale columns of a dimension-tree T that a line-search step rewrote are
// recomputed by the fused MTTKRP on a packed copy of just those columns).
template <typename E>
__global__ void __launch_bounds__(256) reduce_partials_scatter_kernel(const E *partial, int T, int ldPart,
                                                                      int I, E *factor, const int *idx) {
  const int c = blockIdx.y;
  const long long tile = (long long)ldPart * CALS_BN;
  const E *base = partial + (long long)(c >> 7) * T * tile + (long long)ldPart * (c & (CALS_BN - 1));
  E *dst = factor + (long long)I * idx[c];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < I; i += gridDim.x * blockDim.x) {
    const E *p = base + i;
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += (double)p[t * tile];
    dst[i] = (E)s;
  }
}

hipError_t reduce_partials_scatter_launch(const void *partial, int T, int ldPart, int I, int n_cols,
                                          void *factor, const int *idx, int dtype, hipStream_t st) {
  if (n_cols <= 0) return hipSuccess;
  const dim3 grid((I + 255) / 256, n_cols), block(256);
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(reduce_partials_scatter_kernel<float>, grid, block, 0, st, (const float *)partial, T,
                       ldPart, I, (float *)factor, idx);
  else
    hipLaunchKernelGGL(reduce_partials_scatter_kernel<double>, grid, block, 0, st, (const double *)partial,
                       T, ldPart, I, (double *)factor, idx);
  return hipGetLastError();
}

// Columns of the models whose factors the last line-search kernel rewrote (flags bit 0 = extrapolated,
// bit 1 = reverted), in registry order: idx[0 .. count).  One workgroup; an exclusive scan of the ranks.
__global__ void __launch_bounds__(256) stale_cols_kernel(const int *slots, int n, ModelTable mt, int *idx) {
  __shared__ int sh[256];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < n; k0 += 256) {
    const int k = k0 + threadIdx.x;
    int slot = -1, r = 0;
    if (k < n) {
      slot = slots[k];
      if (mt.flags[slot] & 3) r = mt.rank[slot];
    }
    sh[threadIdx.x] = r;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan
      const int v = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += v;
      __syncthreads();
    }
    const int off = base + sh[threadIdx.x] - r;
    if (r) {
      const int col = mt.col[slot];
      for (int c = 0; c < r; ++c) idx[off + c] = col + c;
    }
    __syncthreads();
    if (threadIdx.x == 255) base += sh[255];
    __syncthreads();
  }
}

hipError_t stale_cols_launch(const int *slots, int n, const ModelTable &mt, int *idx, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(stale_cols_kernel, dim3(1), dim3(256), 0, st, slots, n, mt, idx);
  return hipGetLastError();
}

// per-slot scalars of freshly admitted models (MultiKtensor::add: iters = 1, fresh LS state)
__global__ void init_slots_kernel(const int *desc, int n, ModelTable mt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int slot = desc[5 * k];
  mt.col[slot] = desc[5 * k + 1];
  mt.rank[slot] = desc[5 * k + 2];
  mt.jk_mode[slot] = desc[5 * k + 3];
  mt.jk_fiber[slot] = desc[5 * k + 4];
  mt.iters[slot] = 1;
  mt.err[slot] = 0.0;
  mt.fit[slot] = 0.0;
  mt.old_fit[slot] = 0.0;
  mt.potrf_info[slot] = 0;
  mt.ls_iter[slot] = 0;
  mt.ls_updated_last[slot] = 0;
  mt.flags[slot] = 0;
  mt.ls_margin[slot] = 1e300;
}

hipError_t init_slots_launch(const int *desc, int n, const ModelTable &mt, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(init_slots_kernel, dim3((n + 127) / 128), dim3(128), 0, st, desc, n, mt);
  return hipGetLastError();
}

// rmax_needed: largest rank among the models in flight (sizes the LDS panel)
template <typename K>
static hipError_t upd_raise_lds(K kernel, AttrOnce &once, size_t budget) {
  return once.ensure([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)budget);
  });
}

hipError_t update_huge_factor_launch(const UpdateArgs &a, int rmax_needed, hipStream_t st) {
  if (!a.huge_idx || a.n_huge <= 0 || !a.hscratch) return hipErrorInvalidValue;
  const int rp = std::min((rmax_needed + 15) & ~15, CALS_GLD);
  hipLaunchKernelGGL(huge_hadamard_kernel, dim3(rp / 4, (unsigned)a.n_huge), dim3(256), 0, st, a);
  hipLaunchKernelGGL(huge_potrf_kernel, dim3(1, (unsigned)a.n_huge), dim3(HUGE_POTRF_THREADS), 0, st, a);
  return hipGetLastError();
}

// rank_small / rank_big: are there models of rank <= CALS_RFAST / above it in flight (both unknown -> both launched)
hipError_t update_launch(const UpdateArgs &a_in, int rmax_needed, hipStream_t st, int classes) {
  if (a_in.n_slots <= 0) return hipSuccess;
  UpdateArgs a = a_in;
  const size_t es = (a.dtype == CALS_F32) ? sizeof(float) : sizeof(double);
  const int xld = a.I | 1;  // odd leading dimension: the Gramian's 4 x 16 operand reads spread over banks
  const int rmax = std::min(std::max(rmax_needed, 1), CALS_RFAST);  // panel of the fast bodies
  size_t dyn = (size_t)xld * (size_t)rmax * es;
  dyn = (dyn + 15) / 16 * 16;
  const size_t budget = (size_t)160 * 1024 - sizeof(UpdShared) - 1024;
  static const bool no_lds = getenv("CALS_UPDATE_NO_LDS") != nullptr;  // A/B switch
  const bool f32 = a.dtype == CALS_F32;
  const bool small = (classes & 1) != 0 || classes == 0, big = (classes & 2) != 0 || classes == 0;
  const bool huge = (classes & 4) != 0 || (classes == 0 && rmax_needed > CALS_RMAX);
  hipError_t e = hipSuccess;
  if (small) {
    if (dyn <= budget && !no_lds) {
      a.xld = xld;
      static AttrOnce once_f64, once_f32;  // the limit is raised to the whole budget, once per device
      if (f32) {
        if ((e = upd_raise_lds(&update_lds_kernel<float>, once_f32, budget)) != hipSuccess) return e;
        hipLaunchKernelGGL(update_lds_kernel<float>, dim3(a.n_slots), dim3(UPD_THREADS), dyn, st, a);
      } else {
        if ((e = upd_raise_lds(&update_lds_kernel<double>, once_f64, budget)) != hipSuccess) return e;
        hipLaunchKernelGGL(update_lds_kernel<double>, dim3(a.n_slots), dim3(UPD_THREADS), dyn, st, a);
      }
    } else {
      a.xld = 0;
      if (f32)
        hipLaunchKernelGGL(update_hbm_kernel<float>, dim3(a.n_slots), dim3(UPD_THREADS), 0, st, a);
      else
        hipLaunchKernelGGL(update_hbm_kernel<double>, dim3(a.n_slots), dim3(UPD_THREADS), 0, st, a);
    }
  }
  if (big && (rmax_needed > CALS_RFAST || classes == 0)) {  // ranks 33..64
    // ranks 33..64: update_body_huge<T, true> keeps H (64 columns, ld 66) + 4 partial tiles in dynamic LDS
    const size_t hdyn = (size_t)(UPD_HLDS_LD * CALS_RMAX + UPD_WAVES * 256) * sizeof(double);
    a.xld = 0;
    static AttrOnce once_f64, once_f32;
    if (f32) {
      if ((e = upd_raise_lds(&update_huge_kernel<float>, once_f32, budget)) != hipSuccess) return e;
      hipLaunchKernelGGL(update_huge_kernel<float>, dim3(a.n_slots), dim3(UPD_THREADS), hdyn, st, a);
    } else {
      if ((e = upd_raise_lds(&update_huge_kernel<double>, once_f64, budget)) != hipSuccess) return e;
      hipLaunchKernelGGL(update_huge_kernel<double>, dim3(a.n_slots), dim3(UPD_THREADS), hdyn, st, a);
    }
  }
  if (huge) {  // ranks 65..CALS_GLD: the pipeline of huge_* launches over the class's models
    if (!a.huge_idx || a.n_huge <= 0 || !a.hscratch || (!a.rowdot && !a.hrowdot)) return hipErrorInvalidValue;
    const int rp = std::min((rmax_needed + 15) & ~15, CALS_GLD), nt = rp / 16;
    const unsigned nh = (unsigned)a.n_huge;
    a.xld = 0;
    if (!a.rowdot) {
      if (!a.huge_factored && (e = update_huge_factor_launch(a, rmax_needed, st)) != hipSuccess) return e;
      const size_t sdyn = (size_t)HUGE_SOLVE_LDS_DOUBLES(rp) * sizeof(double);
      const dim3 sgrid((a.I + 15) / 16, nh);
      static AttrOnce s_f64, s_f32;
      if (f32) {
        if ((e = upd_raise_lds(&huge_solve_kernel<float>, s_f32, budget)) != hipSuccess) return e;
        hipLaunchKernelGGL(huge_solve_kernel<float>, sgrid, dim3(64), sdyn, st, a);
      } else {
        if ((e = upd_raise_lds(&huge_solve_kernel<double>, s_f64, budget)) != hipSuccess) return e;
        hipLaunchKernelGGL(huge_solve_kernel<double>, sgrid, dim3(64), sdyn, st, a);
      }
    }
    if (f32) {
      hipLaunchKernelGGL(huge_scale_kernel<float>, dim3(rp / 4, nh), dim3(256), 0, st, a);
      hipLaunchKernelGGL(huge_gram_kernel<float>, dim3(nt * (nt + 1) / 2, nh), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL(huge_scale_kernel<double>, dim3(rp / 4, nh), dim3(256), 0, st, a);
      hipLaunchKernelGGL(huge_gram_kernel<double>, dim3(nt * (nt + 1) / 2, nh), dim3(256), 0, st, a);
    }
    if (a.is_last) hipLaunchKernelGGL(huge_error_kernel, dim3(1, nh), dim3(HUGE_ERR_THREADS), 0, st, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// group_contract_kernel (N > 3 dimension tree, see GroupContractArgs): one workgroup per column.
// n_local == 0: thread = i_0 (coalesced over T's fastest index), the other indices in a serial loop;
// n_local > 0: wave = one i_n at a time, lanes over i_0, the remaining indices serial, a wave butterfly at the
// end.  Fixed summation order; HBM / L2 bound (T is read once per mode of its group).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) group_contract_kernel(const GroupContractArgs a) {
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *Tc = static_cast<const T *>(a.T) + a.ldT * c;
  // the group padded to four modes with size-1 modes whose "factor" is the constant 1: plain nested loops, no
  // index decoding (a 64-bit div / mod per element made the first version of this kernel 20x slower)
  int d[4];
  const T *F[4];
  for (int k = 0; k < 4; ++k) {
    d[k] = k < a.h ? a.dims[k] : 1;
    F[k] = k < a.h ? static_cast<const T *>(a.F[k]) + (long long)a.dims[k] * c : nullptr;
  }
  auto f = [&](int k, int i) { return F[k] ? (double)F[k][i] : 1.0; };
  T *out = static_cast<T *>(a.out) + (long long)a.dims[a.n_local] * c;
  __shared__ double s_part[256];
  if (a.n_local == 0) {
    // thread (tx, ty): tx = i_0 within a power-of-two stripe, ty = a contiguous share of the outermost other index
    int stripe = 1;
    while (stripe < d[0] && stripe < 256) stripe <<= 1;
    const int parts = 256 / stripe, tx = tid % stripe, ty = tid / stripe;
    // the outermost mode with more than one entry carries the split
    const int ko = d[3] > 1 ? 3 : (d[2] > 1 ? 2 : 1);
    const int lo = (int)((long long)d[ko] * ty / parts), hi = (int)((long long)d[ko] * (ty + 1) / parts);
    for (int i0 = tx; i0 < d[0] || i0 - tx < d[0]; i0 += stripe) {  // uniform trip count over the stripe
      double acc = 0.0;
      if (i0 < d[0]) {
        for (int i3 = (ko == 3 ? lo : 0); i3 < (ko == 3 ? hi : d[3]); ++i3) {
          const double w3 = f(3, i3);
          for (int i2 = (ko == 2 ? lo : 0); i2 < (ko == 2 ? hi : d[2]); ++i2) {
            const double w23 = w3 * f(2, i2);
            const T *row = Tc + i0 + (long long)d[0] * d[1] * (i2 + (long long)d[2] * i3);
            for (int i1 = (ko == 1 ? lo : 0); i1 < (ko == 1 ? hi : d[1]); ++i1)
              acc += (double)row[(long long)d[0] * i1] * (w23 * f(1, i1));
          }
        }
      }
      s_part[tid] = acc;
      __syncthreads();
      if (ty == 0 && i0 < d[0]) {
        double tot = 0.0;
        for (int p = 0; p < parts; ++p) tot += s_part[tx + stripe * p];  // fixed order
        out[i0] = (T)tot;
      }
      __syncthreads();
    }
    return;
  }
  // n_local > 0: the modes are renamed so that the result's mode is "n" and the two others (besides mode 0)
  // are u and v; one wave per i_n at a time, lanes over i_0, wave butterfly at the end
  const int n = a.n_local;
  int u = -1, v = -1;
  for (int k = 1; k < 4; ++k) {
    if (k == n) continue;
    if (u < 0)
      u = k;
    else
      v = k;
  }
  long long str[4];
  str[0] = 1;
  for (int k = 1; k < 4; ++k) str[k] = str[k - 1] * d[k - 1];
  for (int in = wave; in < d[n]; in += 4) {
    double acc = 0.0;
    for (int iv = 0; iv < d[v]; ++iv) {
      const double wv = f(v, iv);
      for (int iu = 0; iu < d[u]; ++iu) {
        const double w = wv * f(u, iu);
        const T *row = Tc + str[n] * in + str[u] * iu + str[v] * iv;
        double part = 0.0;
        for (int i0 = lane; i0 < d[0]; i0 += 64) part += (double)row[i0] * (double)F[0][i0];
        acc += part * w;
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[in] = (T)acc;
  }
}

hipError_t group_contract_launch(const GroupContractArgs &a, hipStream_t st) {
  if (a.R <= 0) return hipSuccess;
  if (a.h < 2 || a.h > 4 || a.n_local < 0 || a.n_local >= a.h) return hipErrorInvalidValue;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(group_contract_kernel<float>, dim3(a.R), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(group_contract_kernel<double>, dim3(a.R), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Gramians at admission (MultiKtensor::add, src/multi_ktensor.cpp:88-94)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(64) gram_init_kernel(const GramInitArgs a) {
  const int slot = a.slots[blockIdx.x];
  const int m = blockIdx.y;
  const int col = a.mt.col[slot];
  const int r = a.mt.rank[slot];
  gramian_wave<T>(static_cast<const T *>(a.factor[m]) + (long long)a.I[m] * col, a.I[m], a.I[m], r,
                  a.gram[m] + CALS_GLD * (long long)col, threadIdx.x);
}

hipError_t gram_init_launch(const GramInitArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(gram_init_kernel<float>, dim3(a.n_slots, a.n_modes), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL(gram_init_kernel<double>, dim3(a.n_slots, a.n_modes), dim3(64), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// line search (NO_ERROR_CHECKING)
// ---------------------------------------------------------------------------------------------
// prev_ktensor.copy(ktensor) when ls.iter == interval-1 (src/cals.cpp:203-211)
template <typename T>
__global__ void __launch_bounds__(256) ls_snapshot_kernel(const LsArgs a) {
  const int slot = a.slots[blockIdx.x];
  if (a.mt.ls_iter[slot] != a.interval - 1) return;
  const int col = a.mt.col[slot], r = a.mt.rank[slot];
  for (int m = 0; m < a.n_modes; ++m) {
    const long long n = (long long)a.I[m] * r, off = (long long)a.I[m] * col;
    const T *src = static_cast<const T *>(a.factor[m]);
    T *dst = static_cast<T *>(a.prev[m]);
    for (long long e = threadIdx.x; e < n; e += 256) dst[off + e] = src[off + e];
  }
  if (threadIdx.x < r) a.prev_lambda[col + threadIdx.x] = a.lambda[col + threadIdx.x];
}

// ModelTable::ls_margin: how far from a tie the accept / revert test of errors e1, e2 was
__device__ __forceinline__ void ls_note_margin(const ModelTable &mt, int slot, double e1, double e2) {
  const double scale = fmax(fmax(fabs(e1), fabs(e2)), 1e-300);
  const double m = fabs(e1 - e2) / scale;
  if (!(m >= mt.ls_margin[slot])) mt.ls_margin[slot] = m;  // NaN errors read as a tie
}

// ls::line_search (src/utils/line_search.cpp:228-271) for every model, after the error update.
template <typename T>
__global__ void __launch_bounds__(256) ls_kernel(const LsArgs a) {
  const int slot = a.slots[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = a.mt.col[slot], r = a.mt.rank[slot];
  const long long iters = a.mt.iters[slot];
  __shared__ double s_lam[CALS_GLD];
  if (tid == 0) a.mt.flags[slot] = 0;
  // "Make sure extrapolation doesn't happen right before a Ktensor is evicted" (cals.cpp:314-316)
  if (iters >= a.max_iter) return;
  int ls_iter = a.mt.ls_iter[slot] + 1;
  int flags = 0;
  const double step = (a.step == 0.0) ? cbrt((double)iters) : a.step;
  bool regram = false;
  if (a.mt.ls_updated_last[slot]) {
    if (tid == 0) ls_note_margin(a.mt, slot, a.mt.bk_err[slot], a.mt.err[slot]);
    if (a.mt.bk_err[slot] < a.mt.err[slot]) {
      // revert to the backup (Ktensor::copy, src/ktensor.cpp:163-181: scalars, lambda, factors)
      flags |= 2;
      ls_iter = 0;
      for (int m = 0; m < a.n_modes; ++m) {
        const long long n = (long long)a.I[m] * r, off = (long long)a.I[m] * col;
        T *dst = static_cast<T *>(a.factor[m]);
        const T *src = static_cast<const T *>(a.backup[m]);
        for (long long e = tid; e < n; e += 256) dst[off + e] = src[off + e];
      }
      if (tid < r) a.lambda[col + tid] = a.backup_lambda[col + tid];
      for (int m = 0; m < a.n_modes; ++m)
        if (a.act[m]) {  // (r + 63) / 64 mask words per row, word q in column col + q
          const long long n = (long long)a.I[m] * ((r + 63) >> 6), off = (long long)a.I[m] * col;
          for (long long e = tid; e < n; e += 256) a.act[m][off + e] = a.act_backup[m][off + e];
        }
      regram = true;
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (a.mt.ls_updated_last[slot]) {
      a.mt.ls_updated_last[slot] = 0;
      if (flags & 2) {
        a.mt.err[slot] = a.mt.bk_err[slot];
        a.mt.fit[slot] = a.mt.bk_fit[slot];
        a.mt.old_fit[slot] = a.mt.bk_old_fit[slot];
        a.mt.iters[slot] = a.mt.bk_iters[slot];
      }
    }
  }
  __syncthreads();
  if (ls_iter == a.interval) {
    flags |= 1;
    ls_iter = 0;
    // backup_ktensor.copy(ktensor)
    for (int m = 0; m < a.n_modes; ++m) {
      const long long n = (long long)a.I[m] * r, off = (long long)a.I[m] * col;
      T *dst = static_cast<T *>(a.backup[m]);
      const T *src = static_cast<const T *>(a.factor[m]);
      for (long long e = tid; e < n; e += 256) dst[off + e] = src[off + e];
    }
    if (tid < r) a.backup_lambda[col + tid] = a.lambda[col + tid];
    for (int m = 0; m < a.n_modes; ++m)
      if (a.act[m]) {
        const long long n = (long long)a.I[m] * ((r + 63) >> 6), off = (long long)a.I[m] * col;
        for (long long e = tid; e < n; e += 256) a.act_backup[m][off + e] = a.act[m][off + e];
      }
    if (tid == 0) {
      a.mt.bk_err[slot] = a.mt.err[slot];
      a.mt.bk_fit[slot] = a.mt.fit[slot];
      a.mt.bk_old_fit[slot] = a.mt.old_fit[slot];
      a.mt.bk_iters[slot] = a.mt.iters[slot];
      a.mt.ls_updated_last[slot] = 1;
    }
    __syncthreads();
    // line_search_no_error_checking (line_search.cpp:24-71): denormalize both (factor 0 *= lambda),
    // A += step*(A - A_prev) on every mode, normalize() (2-norm per column, lambda = product).
    for (int c = wave; c < r; c += 4) {
      const double lc = a.lambda[col + c], lp = a.prev_lambda[col + c];
      double lam = 1.0;
      for (int m = 0; m < a.n_modes; ++m) {
        T *f = static_cast<T *>(a.factor[m]) + (long long)a.I[m] * (col + c);
        const T *pf = static_cast<const T *>(a.prev[m]) + (long long)a.I[m] * (col + c);
        double ss = 0.0;
        constexpr int NI = 8;  // a column of up to 512 rows lives in registers between its two passes: every load of
        if (a.I[m] <= 64 * NI) {  // the column is in flight at once (the in-place loop below exposes a round trip per 64 rows:
          double xv[NI], pv[NI];  // 144 us per extrapolating sweep at C3), same operations in the same order
#pragma unroll
          for (int q = 0; q < NI; ++q) {
            const int i = lane + 64 * q;
            const bool ok = i < a.I[m];
            xv[q] = ok ? (double)f[ok ? i : 0] : 0.0;
            pv[q] = ok ? (double)pf[ok ? i : 0] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < NI; ++q) {
            if (lane + 64 * q < a.I[m]) {
              double x = xv[q], p = pv[q];
              if (m == 0) {
                x *= lc;
                p *= lp;
              }
              x += step * (x - p);
              ss += x * x;
              xv[q] = (double)(T)x;  // what the in-place form stores and reads back
            }
          }
          const double coeff = sqrt(wave_sum(ss));
          const double s = 1.0 / coeff;
#pragma unroll
          for (int q = 0; q < NI; ++q)
            if (lane + 64 * q < a.I[m]) f[lane + 64 * q] = (T)(s * xv[q]);
          lam *= coeff;
          continue;
        }
        for (int i = lane; i < a.I[m]; i += 64) {
          double x = f[i], p = pf[i];
          if (m == 0) {
            x *= lc;
            p *= lp;
          }
          x += step * (x - p);
          f[i] = (T)x;
          ss += x * x;
        }
        const double coeff = sqrt(wave_sum(ss));
        const double s = 1.0 / coeff;
        for (int i = lane; i < a.I[m]; i += 64) f[i] = (T)(s * (double)f[i]);
        lam *= coeff;
      }
      if (lane == 0) a.lambda[col + c] = lam;
    }
    if (tid == 0) {
      a.mt.err[slot] = DBL_MAX;
      const double of = a.mt.fit[slot];
      a.mt.old_fit[slot] = of;
      a.mt.fit[slot] = 1.0 - fabs(DBL_MAX) / 1.0;
    }
    regram = true;
  }
  __syncthreads();
  if (regram) {  // update_gramians
    for (int m = wave; m < a.n_modes; m += 4)
      gramian_wave<T>(static_cast<const T *>(a.factor[m]) + (long long)a.I[m] * col, a.I[m], a.I[m], r,
                      a.gram[m] + CALS_GLD * (long long)col, lane);
  }
  if (tid == 0) {
    a.mt.ls_iter[slot] = ls_iter;
    a.mt.flags[slot] = flags;
    if (flags && a.changed) atomicAdd(a.changed, r);  // number of columns this line search rewrote
  }
  (void)s_lam;
}

// ---------------------------------------------------------------------------------------------
// line search with error checking (ls::ERROR_CHECKING_SERIAL/_PARALLEL, line_search.cpp:86-153):
// every `interval` model-iterations the extrapolation cur + step (cur - prev) is EVALUATED first and
// kept only if its error is lower.  The reference reconstructs the tensor for that error
// (error.cpp:7-30); here it comes from the same identity as the sweep's error,
//   ||X - M||^2 = ||X||^2 + sum_ij l_i l_j (G0 o G1 o ... )_ij - 2 sum_c l_c <a_c, MTTKRP_0(M)_c>,
// with one extra MTTKRP of the extrapolated factors (launched by the engine between the two
// kernels, only on sweeps in which some model reaches its interval).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ls_ec_prepare_kernel(const LsArgs a) {
  const int slot = a.slots[blockIdx.x];
  const int tid = threadIdx.x;
  const int col = a.mt.col[slot], r = a.mt.rank[slot];
  const long long iters = a.mt.iters[slot];
  const int ls_iter = a.mt.ls_iter[slot] + 1;
  if (ls_iter != a.interval) {
    if (tid == 0) {
      a.mt.ls_iter[slot] = ls_iter;
      a.mt.flags[slot] = 0;
    }
    return;
  }
  const double step = (a.step == 0.0) ? cbrt((double)iters) : a.step;
  // ls_ktensor (= the prev copy) := cur + step (cur - prev); its lambda := the current lambda
  for (int m = 0; m < a.n_modes; ++m) {
    const long long n = (long long)a.I[m] * r, off = (long long)a.I[m] * col;
    const T *cur = static_cast<const T *>(a.factor[m]);
    T *old = static_cast<T *>(a.prev[m]);
    for (long long e = tid; e < n; e += 256) {
      const double c = (double)cur[off + e], diff = c - (double)old[off + e];
      old[off + e] = (T)(c + step * diff);
    }
  }
  if (tid < r) a.prev_lambda[col + tid] = a.lambda[col + tid];
  if (tid == 0) {
    a.mt.ls_iter[slot] = 0;
    a.mt.flags[slot] = 1;  // extrapolated; decide adds "reversed" if the candidate is dropped
  }
}

template <typename T>
__global__ void __launch_bounds__(256) ls_ec_decide_kernel(const LsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ec_dyn[];
  const int slot = a.slots[blockIdx.x];
  if (!(a.mt.flags[slot] & 1)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = a.mt.col[slot], r = a.mt.rank[slot];
  // r x r Hadamard of the candidate's Gramians (H) and one Gramian at a time (tmp): in LDS, ld CALS_RMAX,
  // up to rank CALS_RMAX; above it two CALS_GLD x CALS_GLD blocks of the engine's global scratch, handed out
  // through a counter like update_body_huge's
  __shared__ double red[4];
  __shared__ int s_accept, s_block;
  const bool huge = r > CALS_RMAX;
  if (huge && tid == 0) s_block = atomicAdd(a.hcounter, 2);
  __syncthreads();
  const int ldh = huge ? CALS_GLD : CALS_RMAX;
  double *H = huge ? a.hscratch + (long long)s_block * CALS_GLD * CALS_GLD : reinterpret_cast<double *>(ec_dyn);
  double *tmp = huge ? H + (long long)CALS_GLD * CALS_GLD : H + CALS_RMAX * CALS_RMAX;
  // t2 = sum_ij l_i l_j prod_m (F_m^T F_m)_ij over the candidate's factors (in a.prev)
  for (int m = 0; m < a.n_modes; ++m) {
    if (wave == 0) {
      const T *panel = static_cast<const T *>(a.prev[m]) + (long long)a.I[m] * col;
      if (huge)
        gramian_wave<T, CALS_GLD>(panel, a.I[m], a.I[m], r, m == 0 ? H : tmp, lane);
      else
        gramian_wave<T, CALS_RMAX>(panel, a.I[m], a.I[m], r, m == 0 ? H : tmp, lane);
    }
    __threadfence_block();
    __syncthreads();
    if (m > 0)
      for (int e = tid; e < r * r; e += 256) {
        const int i = e % r, j = e / r;
        H[i + ldh * j] *= tmp[i + ldh * j];
      }
    __threadfence_block();
    __syncthreads();
  }
  double t2 = 0.0, t3 = 0.0;
  for (int e = tid; e < r * r; e += 256) {
    const int i = e % r, j = e / r;
    t2 += a.prev_lambda[col + i] * a.prev_lambda[col + j] * H[i + ldh * j];
  }
  {
    const int I0 = a.I[0];
    const T *A0 = static_cast<const T *>(a.prev[0]) + (long long)I0 * col;
    const T *G0 = static_cast<const T *>(a.Gs) + (long long)I0 * col;
    for (long long e = tid; e < (long long)I0 * r; e += 256) {
      const int c = (int)(e / I0);
      t3 += a.prev_lambda[col + c] * (double)A0[e] * (double)G0[e];
    }
  }
  t2 = wave_sum(t2);
  t3 = wave_sum(t3);
  if (lane == 0) red[wave] = t2;
  __syncthreads();
  t2 = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (lane == 0) red[wave] = t3;
  __syncthreads();
  t3 = red[0] + red[1] + red[2] + red[3];
  // the reference evaluates the candidate against the FULL tensor, jackknife models included
  const double error = sqrt(fmax(a.X_norm * a.X_norm + t2 - 2.0 * t3, 0.0));
  if (tid == 0) {
    ls_note_margin(a.mt, slot, error, a.mt.err[slot]);
    s_accept = (error < a.mt.err[slot]) ? 1 : 0;
  }
  __syncthreads();
  if (!s_accept) {
    if (tid == 0) a.mt.flags[slot] = 3;  // extrapolated + reversed
    return;
  }
  // keep it: compute_error left the candidate normalised (factor 0 scaled by lambda first, then every
  // column of every mode to unit 2-norm, ktensor.cpp:85-107); the factors are copied, lambda is NOT
  // (line_search.cpp:121-127)
  for (int m = 0; m < a.n_modes; ++m) {
    for (int c = wave; c < r; c += 4) {
      const T *src = static_cast<const T *>(a.prev[m]) + (long long)a.I[m] * (col + c);
      T *dst = static_cast<T *>(a.factor[m]) + (long long)a.I[m] * (col + c);
      const double lc = (m == 0) ? a.prev_lambda[col + c] : 1.0;
      double ss = 0.0;
      for (int i = lane; i < a.I[m]; i += 64) {
        const double x = lc * (double)src[i];
        ss += x * x;
      }
      const double s = 1.0 / sqrt(wave_sum(ss));
      for (int i = lane; i < a.I[m]; i += 64) dst[i] = (T)(s * (lc * (double)src[i]));
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int m = wave; m < a.n_modes; m += 4)
    gramian_wave<T>(static_cast<const T *>(a.factor[m]) + (long long)a.I[m] * col, a.I[m], a.I[m], r,
                    a.gram[m] + CALS_GLD * (long long)col, lane);
  if (tid == 0) {
    a.mt.err[slot] = error;
    const double of = a.mt.fit[slot];
    a.mt.old_fit[slot] = of;
    a.mt.fit[slot] = 1.0 - fabs(error) / a.X_norm;
    if (a.changed) atomicOr(a.changed, 1);
  }
}

hipError_t ls_ec_prepare_launch(const LsArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(ls_ec_prepare_kernel<float>, dim3(a.n_slots), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(ls_ec_prepare_kernel<double>, dim3(a.n_slots), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t ls_ec_decide_launch(const LsArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  const size_t dyn = (size_t)2 * CALS_RMAX * CALS_RMAX * sizeof(double);
  static AttrOnce once;
  const hipError_t ea = once.ensure([&] {
    const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_ec_decide_kernel<float>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (e1 != hipSuccess) return e1;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&ls_ec_decide_kernel<double>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  });
  if (ea != hipSuccess) return ea;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(ls_ec_decide_kernel<float>, dim3(a.n_slots), dim3(256), dyn, st, a);
  else
    hipLaunchKernelGGL(ls_ec_decide_kernel<double>, dim3(a.n_slots), dim3(256), dyn, st, a);
  return hipGetLastError();
}

hipError_t ls_snapshot_launch(const LsArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(ls_snapshot_kernel<float>, dim3(a.n_slots), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(ls_snapshot_kernel<double>, dim3(a.n_slots), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t ls_launch(const LsArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(ls_kernel<float>, dim3(a.n_slots), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(ls_kernel<double>, dim3(a.n_slots), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// end of sweep: eviction rule + iters++ (src/cals.cpp:336-354, always_evict_first is host-side)
// ---------------------------------------------------------------------------------------------
__global__ void finish_kernel(const FinishArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n_slots) return;
  const int slot = a.slots[k];
  const long long it = a.mt.iters[slot];
  bool evict = false;
  if (a.evict_enabled) {
    if (!a.force_max_iter)
      evict = (fabs(a.mt.old_fit[slot] - a.mt.fit[slot]) < a.tol) || (it >= a.max_iter);
    else
      evict = it >= a.max_iter;
  }
  if (evict)
    a.mt.flags[slot] |= 4;
  else
    a.mt.iters[slot] = it + 1;
}

// per-sweep status of the in-flight models, packed in registry order for ONE device-to-host copy:
// out[0] = {changed flag of the line search, -, -, -, -}; out[1 + k] = record of slots[k]
__global__ void pack_status_kernel(const int *slots, int n, ModelTable mt, const int *changed,
                                   const int *nnls_status, StatusRec *out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) {
    StatusRec h;
    h.flags = changed ? *changed : 0;
    h.pad = nnls_status ? *nnls_status : 0;
    h.iters = n;
    h.err = h.fit = h.old_fit = 0.0;
    h.ls_margin = 0.0;
    out[0] = h;
  }
  if (k >= n) return;
  const int slot = slots[k];
  StatusRec r;
  r.flags = mt.flags[slot];
  r.pad = slot;
  r.iters = mt.iters[slot];
  r.err = mt.err[slot];
  r.fit = mt.fit[slot];
  r.old_fit = mt.old_fit[slot];
  r.ls_margin = mt.ls_margin[slot];
  out[1 + k] = r;
}

hipError_t pack_status_launch(const int *slots, int n, const ModelTable &mt, const int *changed,
                              const int *nnls_status, StatusRec *out, hipStream_t st) {
  hipLaunchKernelGGL(pack_status_kernel, dim3((std::max(n, 1) + 127) / 128), dim3(128), 0, st, slots, n,
                     mt, changed, nnls_status, out);
  return hipGetLastError();
}

hipError_t finish_launch(const FinishArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(finish_kernel, dim3((a.n_slots + 127) / 128), dim3(128), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// set-up kernels
// ---------------------------------------------------------------------------------------------
// Xp[m + Mp*(a + Ap*s)] = X[...] with the output mode `m_mode` fastest, then the inner mode
// `a_mode`, then the remaining modes in increasing order (first fastest); pads are zero.
struct PermArgs {
  const void *X;
  void *Xp;
  int n_modes;
  int dims[CALS_MAX_MODES];
  long long stride[CALS_MAX_MODES];  // element stride of every mode in X
  int m_mode, a_mode;
  int Mp, Ap;
  long long S;
};

template <typename S, typename T>
__global__ void permute_pad_kernel(const PermArgs a) {
  const long long total = (long long)a.Mp * a.Ap * a.S;
  const int M = a.dims[a.m_mode], A = a.dims[a.a_mode];
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(e % a.Mp);
    const long long t = e / a.Mp;
    const int ai = (int)(t % a.Ap);
    long long s = t / a.Ap;
    double v = 0.0;
    if (m < M && ai < A) {
      long long off = m * a.stride[a.m_mode] + ai * a.stride[a.a_mode];
      for (int k = 0; k < a.n_modes; ++k) {
        if (k == a.m_mode || k == a.a_mode) continue;
        off += (s % a.dims[k]) * a.stride[k];
        s /= a.dims[k];
      }
      v = (double)static_cast<const S *>(a.X)[off];
    }
    static_cast<T *>(a.Xp)[e] = (T)v;
  }
}

hipError_t permute_pad_launch(const void *X, int src_dtype, int n_modes, const int *dims, int m_mode,
                              int a_mode, int Mp, int Ap, void *Xp, int dst_dtype, long long S,
                              hipStream_t st) {
  PermArgs a;
  a.X = X;
  a.Xp = Xp;
  a.n_modes = n_modes;
  long long str = 1;
  for (int k = 0; k < n_modes; ++k) {
    a.dims[k] = dims[k];
    a.stride[k] = str;
    str *= dims[k];
  }
  a.m_mode = m_mode;
  a.a_mode = a_mode;
  a.Mp = Mp;
  a.Ap = Ap;
  a.S = S;
  if (src_dtype == CALS_F32 && dst_dtype == CALS_F32)
    hipLaunchKernelGGL((permute_pad_kernel<float, float>), dim3(4096), dim3(256), 0, st, a);
  else if (src_dtype == CALS_F32)
    hipLaunchKernelGGL((permute_pad_kernel<float, double>), dim3(4096), dim3(256), 0, st, a);
  else if (dst_dtype == CALS_F32)
    hipLaunchKernelGGL((permute_pad_kernel<double, float>), dim3(4096), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((permute_pad_kernel<double, double>), dim3(4096), dim3(256), 0, st, a);
  return hipGetLastError();
}

// Sum of squares of every mode-0 slice (utils::calculate_jackknifing_norms, utils.cpp:103-152),
// deterministic: stage 1 = n_part blocks each reduce a contiguous range of columns into
// partial[b][i]; stage 2 = ordered sum over b.
template <typename S>
__global__ void slice_sumsq_stage1(const S *X, long long I, long long cols, double *partial,
                                   int n_part) {
  const int b = blockIdx.x;
  const long long c0 = cols * b / n_part, c1 = cols * (b + 1) / n_part;
  for (long long i = threadIdx.x; i < I; i += blockDim.x) {
    double s = 0.0;
    for (long long c = c0; c < c1; ++c) {
      const double v = (double)X[i + I * c];
      s += v * v;
    }
    partial[(long long)b * I + i] = s;
  }
}

__global__ void slice_sumsq_stage2(const double *partial, long long I, int n_part, double *ss) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= I) return;
  double s = 0.0;
  for (int b = 0; b < n_part; ++b) s += partial[(long long)b * I + i];
  ss[i] = s;
}

hipError_t slice_sumsq_launch(const void *X, int src_dtype, long long I, long long cols,
                              double *partial, int n_part, double *ss_out, hipStream_t st) {
  if (src_dtype == CALS_F32)
    hipLaunchKernelGGL(slice_sumsq_stage1<float>, dim3(n_part), dim3(256), 0, st, (const float *)X, I,
                       cols, partial, n_part);
  else
    hipLaunchKernelGGL(slice_sumsq_stage1<double>, dim3(n_part), dim3(256), 0, st, (const double *)X,
                       I, cols, partial, n_part);
  hipLaunchKernelGGL(slice_sumsq_stage2, dim3((unsigned)((I + 255) / 256)), dim3(256), 0, st,
                     partial, I, n_part, ss_out);
  return hipGetLastError();
}

// compress: move ncols columns starting at src_col left by off columns; one block walks the
// columns in ascending order (destination < source, so this is overlap-safe).
template <typename E>
__global__ void move_columns_kernel(E *buf, long long rows, long long src_col, long long ncols,
                                    long long off) {
  for (long long c = 0; c < ncols; ++c) {
    const E *s = buf + rows * (src_col + c);
    E *d = buf + rows * (src_col + c - off);
    for (long long i = threadIdx.x; i < rows; i += blockDim.x) d[i] = s[i];
    __syncthreads();
  }
}

hipError_t move_columns_launch(void *buf, int dtype, long long rows, long long src_col,
                               long long ncols, long long off, hipStream_t st) {
  if (ncols <= 0 || off <= 0) return hipSuccess;
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(move_columns_kernel<float>, dim3(1), dim3(256), 0, st, (float *)buf, rows,
                       src_col, ncols, off);
  else
    hipLaunchKernelGGL(move_columns_kernel<double>, dim3(1), dim3(256), 0, st, (double *)buf, rows,
                       src_col, ncols, off);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Batched column traffic of the MultiKtensor life cycle (eviction, compress): all buffers and all
// columns of one operation in ONE launch.  A move goes through a compact scratch copy (gather the
// source columns, then scatter them to their destinations): compress shifts models left into the
// holes, so a destination may be another moved model's source.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_columns_kernel(const ColMoveArgs a) {
  const int k = blockIdx.x, b = blockIdx.y;
  const ColBuf &cb = a.buf[b];
  const long long words = cb.rows * cb.words_per_elem;
  unsigned *src = static_cast<unsigned *>(cb.ptr) + words * a.src[k];
  unsigned *dst = a.scratch + a.scratch_off[b] + words * k;
  const bool zero = b < a.zero_src_bufs;
  for (long long i = threadIdx.x; i < words; i += 256) {
    dst[i] = src[i];
    if (zero) src[i] = 0u;
  }
}

__global__ void __launch_bounds__(256) scatter_columns_kernel(const ColMoveArgs a) {
  const int k = blockIdx.x, b = blockIdx.y;
  const ColBuf &cb = a.buf[b];
  const long long words = cb.rows * cb.words_per_elem;
  const unsigned *src = a.scratch + a.scratch_off[b] + words * k;
  unsigned *dst = static_cast<unsigned *>(cb.ptr) + words * a.dst[k];
  for (long long i = threadIdx.x; i < words; i += 256) dst[i] = src[i];
}

hipError_t gather_columns_launch(const ColMoveArgs &a, hipStream_t st) {
  if (a.n_cols <= 0 || a.n_bufs <= 0) return hipSuccess;
  hipLaunchKernelGGL(gather_columns_kernel, dim3(a.n_cols, a.n_bufs), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t scatter_columns_launch(const ColMoveArgs &a, hipStream_t st) {
  if (a.n_cols <= 0 || a.n_bufs <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_columns_kernel, dim3(a.n_cols, a.n_bufs), dim3(256), 0, st, a);
  return hipGetLastError();
}

// mt.col[slot] = col for n (slot, col) pairs
__global__ void set_cols_kernel(const int *pairs, int n, int *col) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) col[pairs[2 * k]] = pairs[2 * k + 1];
}

hipError_t set_cols_launch(const int *pairs, int n, int *col, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(set_cols_kernel, dim3((n + 127) / 128), dim3(128), 0, st, pairs, n, col);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// CALS_HIP_VERIFY=1 (debugging switch of the engine): recompute-and-compare checks of everything the engine
// keeps ACROSS launches -- the packed operand Pt an update launch leaves behind, a T that waits for its second
// mode (also over the sweep boundary), the Gramians of the other modes, the all-zero free columns.  A stale
// operand is a wrong result of the size of one sweep's progress, which a converging model hides within the
// test tolerances more often than not; these kernels make it an error at the launch that would consume it.
// One workgroup per in-flight model (x a slice of its elements); `count` gets the number of differing elements.
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) verify_kernel(const VerifyArgs a) {
  const int slot = a.slots[blockIdx.x];
  if (a.skip_flagged && (a.mt.flags[slot] & 3)) return;
  const int col = a.mt.col[slot], r = a.mt.rank[slot];
  const E *A = static_cast<const E *>(a.a), *B = static_cast<const E *>(a.b);
  int bad = 0;
  if (a.kind == 3) {  // r x r Gramian block, ld CALS_GLD, relative tolerance (different summation orders)
    for (int e = threadIdx.x; e < r * r; e += 256) {
      const long long at = (e % r) + (long long)CALS_GLD * (col + e / r);
      const double x = (double)A[at], y = (double)B[at];
      if (x != x && y != y) continue;  // a diverged model: NaN on both sides
      if (!(fabs(x - y) <= a.tol * fmax(1.0, fmax(fabs(x), fabs(y))))) bad++;
    }
  } else if (a.kind == 1) {  // Pt[(column block)][row < rows][128]
    for (long long e = threadIdx.x; e < a.rows * r; e += 256) {
      const int gc = col + (int)(e % r);
      const long long at = ((long long)(gc >> 7) * a.rows + e / r) * CALS_BN + (gc & (CALS_BN - 1));
      if (A[at] != B[at] && !(A[at] != A[at] && B[at] != B[at])) bad++;
    }
  } else {  // column major, `rows` elements per column
    for (long long e = threadIdx.x; e < a.rows * r; e += 256) {
      const long long at = a.rows * col + e;
      if (A[at] != B[at] && !(A[at] != A[at] && B[at] != B[at])) bad++;
    }
  }
  if (bad) atomicAdd(a.count, bad);
}

hipError_t verify_launch(const VerifyArgs &a, hipStream_t st) {
  if (a.n_slots <= 0) return hipSuccess;
  if (a.kind == 3 || a.dtype != CALS_F32)
    hipLaunchKernelGGL(verify_kernel<double>, dim3(a.n_slots), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(verify_kernel<float>, dim3(a.n_slots), dim3(256), 0, st, a);
  return hipGetLastError();
}

// the columns listed in `cols` (free columns of the multi-factor) must hold zeros in all `rows` rows
template <typename E>
__global__ void __launch_bounds__(256) verify_zero_kernel(const E *buf, long long rows, const int *cols, int *count) {
  const E *p = buf + rows * cols[blockIdx.x];
  int bad = 0;
  for (long long i = threadIdx.x; i < rows; i += 256)
    if (!(p[i] == (E)0)) bad++;
  if (bad) atomicAdd(count, bad);
}

hipError_t verify_zero_launch(const void *buf, long long rows, int dtype, const int *cols, int n, int *count,
                              hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(verify_zero_kernel<float>, dim3(n), dim3(256), 0, st, (const float *)buf, rows, cols, count);
  else
    hipLaunchKernelGGL(verify_zero_kernel<double>, dim3(n), dim3(256), 0, st, (const double *)buf, rows, cols, count);
  return hipGetLastError();
}

}  // namespace calship
