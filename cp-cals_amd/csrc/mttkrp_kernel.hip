// Fused MTTKRP for the multi-factor block on gfx950 (CDNA4):
//
//   G[m, c] = sum_{a, s} Xp[m, a, s] * P[a, c] * Q[s, c]          c < R (all in-flight models)
//
// which is mttkrp::mttkrp of the reference (src/utils/mttkrp.cpp:562-614; the explicit
// Khatri-Rao product of :78-103/:179-216 and the block GEMMs of :218-328) with the KRP never
// materialised: the B operand of every v_mfma_f64_16x16x4_f64 is formed in registers as
// P[a,c]*Q[s,c].  Xp is the engine's zero-padded copy of X with the output mode fastest, so
// every mode runs this one kernel ("Data layout in HBM").
//
// Work decomposition
//   column block nb  : CALS_BN = 128 columns; 8 waves, wave w owns 16 of them (one MFMA n-tile);
//                      two waves per SIMD so one wave's MFMAs cover the other's LDS latency
//   M block          : MT m-tiles of 16 rows, accumulators acc[MT] (f64x4 each, 8*MT VGPRs)
//   unit             : (ab, s) = one 16-deep slab of the inner mode at one streamed index:
//                      a contiguous 16*Mp doubles of Xp; 4 MFMA k-steps
//   team member tm   : T workgroups share a column block and split the unit range evenly
//                      (split-K); each writes one partial tile, summed in fixed order by the
//                      update kernel (deterministic, no atomics)
//   stage            : SB consecutive units, staged global->LDS by LDS-DMA
//                      (global_load_lds_dwordx4), double-buffered: the DMA of stage i+1 runs
//                      under the MFMAs of stage i, one s_waitcnt vmcnt(0)+barrier per stage.
// The LDS slab keeps Xp's column-major shape [16][LDL] with LDL = 16 (mod 32) doubles so that
// the A-operand ds_read_b64 of lanes (l&15, l>>4) is bank-conflict free.
#include "cals_hip_internal.h"

#include <cstdlib>

namespace calship {

typedef double v4d __attribute__((ext_vector_type(4)));

template <int MT>
struct MtCfg {
  static constexpr int LDL = (MT % 2 == 1) ? 16 * MT : 16 * MT + 16;  // LDS leading dim (doubles)
  static constexpr int SLAB = 16 * LDL;                               // doubles per unit slab
  static constexpr int PIECES = LDL / 8;                              // 1 KiB DMA pieces per slab
  static constexpr int SB_RAW = (80 * 1024) / ((SLAB + CALS_BN) * 8);  // 2 stages <= 160 KiB
  static constexpr int SB = SB_RAW < 1 ? 1 : (SB_RAW > 8 ? 8 : SB_RAW);  // units per stage
  static constexpr int STAGE = SLAB * SB;
  static constexpr int QSTAGE = SB * CALS_BN;                         // Q values per stage
  static constexpr int LDS_BYTES = 2 * (STAGE + QSTAGE) * 8;
};

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// Hand-pipelined A-operand stream for one slab (VAR 4): ds_read_b64 into a ring of D registers,
// each MFMA preceded by a COUNTED s_waitcnt lgkmcnt(n) (LDS returns in order, so "all but the n
// youngest reads" is exactly "operand I has landed").  hipcc's own waits for this pattern are
// lgkmcnt(0), which also waits for the read issued a few cycles earlier and stalls the wave for a
// full LDS latency every few MFMAs.  The reads are inline asm (invisible to hipcc's counters), so
// every wait is placed here; sched_barrier(0) keeps the MFMA below its wait (guide rule 18).
template <int MT, int LDL, int D, int I0, int I1>
struct SlabPipe {
  static __device__ __forceinline__ void run(v4d (&acc)[MT], double (&ring)[D], const double (&bq)[4],
                                             unsigned base) {
    if constexpr (I0 < I1) {
      constexpr int N = 4 * MT;
      constexpr int outstanding = (D - 1 < N - 1 - I0) ? D - 1 : N - 1 - I0;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
      __builtin_amdgcn_sched_barrier(0);
      constexpr int q = I0 / MT, t = I0 % MT;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[I0 % D], bq[q], acc[t], 0, 0, 0);
      if constexpr (I0 + D < N) {
        constexpr int qn = (I0 + D) / MT, tn = (I0 + D) % MT;
        asm volatile("ds_read_b64 %0, %1 offset:%2"
                     : "=v"(ring[I0 % D])
                     : "v"(base), "i"(((4 * qn) * LDL + 16 * tn) * 8));
      }
      SlabPipe<MT, LDL, D, I0 + 1, I1>::run(acc, ring, bq, base);
    }
  }
  static __device__ __forceinline__ void preload(double (&ring)[D], unsigned base) {
    if constexpr (I0 < I1) {
      constexpr int q = I0 / MT, t = I0 % MT;
      asm volatile("ds_read_b64 %0, %1 offset:%2"
                   : "=v"(ring[I0 % D])
                   : "v"(base), "i"(((4 * q) * LDL + 16 * t) * 8));
      SlabPipe<MT, LDL, D, I0 + 1, I1>::preload(ring, base);
    }
  }
};

// VAR (tuning variants, selected with CALS_MTTKRP_VARIANT):
//   bit 0: A-operand reads are unmerged ds_read_b64 software-pipelined PF MFMAs ahead
//   bit 1: the next stage's LDS-DMA is issued after the first MFMA block of the current stage
//          instead of before it
//   4    : as 3, with the hand-pipelined inline-asm operand stream (SlabPipe) and counted waits
template <int MT, int VAR>
__global__ void __launch_bounds__(512, 2) mttkrp_kernel(const MttkrpArgs a) {
  typedef MtCfg<MT> C;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long dbg_t0 = a.dbg_clock ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long dbg_r0 = a.dbg_clock ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int krow = lane >> 4;   // MFMA k index of this lane (0..3)
  const int lcol = lane & 15;   // MFMA m (A operand) / n (B operand, C/D) index

  // XCD-aware bijective remap: workgroups that share an XCD (same blockIdx % 8) get
  // consecutive p, i.e. the same team member of neighbouring column blocks => they stream the
  // same Xp slabs through one L2 (speed only; any placement is correct).
  const int G = a.grid;
  const int b = blockIdx.x;
  const int xcd = b & 7, q8 = G >> 3, r8 = G & 7;
  const int p = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int tm = p / a.NB;
  const int nb = p - tm * a.NB;
  const int m0 = blockIdx.y * (16 * MT);

  const long long S = a.S;
  const long long U = (long long)(a.Ap >> 4) * S;
  const long long u_begin = U * tm / a.T;
  const long long u_end = a.dbg_no_units ? u_begin : U * (tm + 1) / a.T;

  // this lane's column
  const int col = nb * CALS_BN + wave * 16 + lcol;
  const bool cvalid = col < a.R;

  v4d acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

  const long long slab_stride_s = (long long)a.Mp * a.Ap;  // Xp elements per s

  // ---- LDS-DMA of one stage: n slabs (ab, s0..s0+n-1) -> lds[buf] ----
  auto issue_stage = [&](int buf, long long ab, long long s0, int n) {
#pragma unroll
    for (int j = 0; j < C::SB; ++j) {
      if (j < n) {
        const double *src_slab = static_cast<const double *>(a.Xp) + (long long)a.Mp * (16 * ab) + slab_stride_s * (s0 + j);
        double *dst_slab = lds + buf * C::STAGE + j * C::SLAB;
#pragma unroll
        for (int pc = 0; pc < (C::PIECES + 7) / 8; ++pc) {
          const int piece = pc * 8 + wave;
          if (piece < C::PIECES) {
            const int e = piece * 128 + lane * 2;  // LDS element of this lane (16 B = 2 doubles)
            const int acol = e / C::LDL;
            const int m = e - acol * C::LDL;
            int gm = m0 + m;
            gm = gm < a.Mp ? gm : 0;  // rows past the padded tensor: any valid address, unused
            const double *src = src_slab + gm + (long long)a.Mp * acol;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                             (LDS_AS void *)(dst_slab + piece * 128), 16, 0, 0);
          }
        }
      }
    }
  };

  // Q[s0+j, column block] -> LDS by 4-byte LDS-DMA (two lanes per double).  Everything the
  // main loop consumes arrives by LDS-DMA: an ordinary global load whose result is used while a
  // DMA is in flight makes hipcc drain vmcnt(0) and serialises copy and compute.
  double *const qlds = lds + 2 * C::STAGE;
  auto issue_q = [&](int buf, long long s0, int n) {
#pragma unroll
    for (int pc = 0; pc < (4 * C::SB + 7) / 8; ++pc) {
      const int piece = pc * 8 + wave;  // piece = j*4 + quarter, 32 doubles each
      const int j = piece >> 2;
      if (j < n) {
        const int d = (piece & 3) * 32 + (lane >> 1);
        int c = nb * CALS_BN + d;
        c = c < a.R ? c : 0;  // columns past R: any valid address, result never read
        const char *src = (const char *)(static_cast<const double *>(a.Q) + (s0 + j) + a.ldQ * c) + 4 * (lane & 1);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                         (LDS_AS void *)(qlds + buf * C::QSTAGE + piece * 32), 4, 0,
                                         0);
      }
    }
  };
  auto stage_of = [&](long long u, long long &ab, long long &s0, int &n) {
    ab = u / S;
    s0 = u - ab * S;
    long long nn = S - s0;
    if (nn > u_end - u) nn = u_end - u;
    if (nn > C::SB) nn = C::SB;
    n = (int)nn;
  };

  long long u = u_begin;
  long long ab_c = 0, s_c = 0;
  int n_c = 0, buf = 0;
  if (u < u_end) {
    stage_of(u, ab_c, s_c, n_c);
    issue_stage(0, ab_c, s_c, n_c);
    issue_q(0, s_c, n_c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  double preg[4];
  long long ab_loaded = -1;
  unsigned long long dbg_wait = 0, dbg_bar = 0, dbg_stages = 0;
  const double *abase = lds + krow * C::LDL + lcol;

  while (u < u_end) {
    const long long un = u + n_c;
    long long ab_n = 0, s_n = 0;
    int n_n = 0;
    if (ab_c != ab_loaded) {
      // P[16ab + 4q + krow, col]: one inline-asm load + its own wait per value, so hipcc sees a
      // plain register definition and puts no vmcnt wait into the MFMA loop.  Runs once per S
      // units, before the next stage's DMA is issued.
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int arow = (int)(16 * ab_c) + 4 * q + krow;
        const bool ok = (arow < a.A) && cvalid;
        const double *ptr = static_cast<const double *>(a.P) + (ok ? arow + a.ldP * col : 0);
        double v;
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(v)
                     : "v"(ptr)
                     : "memory");
        preg[q] = ok ? v : 0.0;
      }
      ab_loaded = ab_c;
    }
    const bool have_next = un < u_end;
    if (have_next) stage_of(un, ab_n, s_n, n_n);
    if (VAR != 4 && !(VAR & 2) && have_next) {
      issue_stage(buf ^ 1, ab_n, s_n, n_n);
      issue_q(buf ^ 1, s_n, n_n);
    }

    // runtime loop over the stage's slabs (not unrolled: one slab's 4*MT MFMAs is already a long
    // straight-line body); the per-slab Q value comes from LDS.
#pragma unroll 1
    for (int j = 0; j < n_c; ++j) {
      const double q0 = qlds[buf * C::QSTAGE + j * CALS_BN + wave * 16 + lcol];
      const double *sl = abase + buf * C::STAGE + j * C::SLAB;
      if (VAR == 4) {
        double bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = preg[q] * q0;
        constexpr int D = (4 * MT) < 6 ? (4 * MT) : 6;
        constexpr int SPLIT = (MT < 4 * MT) ? MT : 4 * MT;
        double ring[D];
        const unsigned base = (unsigned)(size_t)((LDS_AS const char *)sl);
        asm volatile("" ::: "memory");  // q0 (and its compiler-inserted wait) stay above the asm reads
        SlabPipe<MT, C::LDL, D, 0, D>::preload(ring, base);
        SlabPipe<MT, C::LDL, D, 0, SPLIT>::run(acc, ring, bq, base);
        if (j == 0 && have_next) {
          issue_stage(buf ^ 1, ab_n, s_n, n_n);
          issue_q(buf ^ 1, s_n, n_n);
        }
        SlabPipe<MT, C::LDL, D, SPLIT, 4 * MT>::run(acc, ring, bq, base);
      } else if (VAR & 1) {
        double bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = preg[q] * q0;
        // volatile LDS pointer: no ds_read2 merging, reads stay in program order
        const volatile LDS_AS double *vsl = (const volatile LDS_AS double *)sl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const double av = vsl[(4 * q) * C::LDL + 16 * t];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[q], acc[t], 0, 0, 0);
          }
          if ((VAR & 2) && q == 0 && j == 0 && have_next) {
            issue_stage(buf ^ 1, ab_n, s_n, n_n);
            issue_q(buf ^ 1, s_n, n_n);
          }
        }
        constexpr int PF = 4;
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
        for (int i = 0; i < 4 * MT - PF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double bq = preg[q] * q0;
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            const double av = sl[(4 * q) * C::LDL + 16 * t];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq, acc[t], 0, 0, 0);
          }
          if ((VAR & 2) && q == 0 && j == 0 && have_next) {
            issue_stage(buf ^ 1, ab_n, s_n, n_n);
            issue_q(buf ^ 1, s_n, n_n);
          }
        }
      }
    }

    if (a.dbg_clock) {  // diagnostics build path: where does the end of a stage go?
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long s2 = __builtin_amdgcn_s_memtime();
      dbg_wait += s1 - s0;
      dbg_bar += s2 - s1;
      dbg_stages++;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    u = un;
    ab_c = ab_n;
    s_c = s_n;
    n_c = n_n;
    buf ^= 1;
  }

  if (a.dbg_clock && lane == 0) {  // diagnostics only: shader clock held during the unit loop
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (wave == 0) {
      a.dbg_clock[2 * wg] = __builtin_amdgcn_s_memtime() - dbg_t0;
      a.dbg_clock[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
    if (wg < 32) {  // per-wave stage-end breakdown of the first 32 workgroups
      a.dbg_clock[4096 + (wg * 8 + wave) * 3 + 0] = dbg_wait;
      a.dbg_clock[4096 + (wg * 8 + wave) * 3 + 1] = dbg_bar;
      a.dbg_clock[4096 + (wg * 8 + wave) * 3 + 2] = dbg_stages;
    }
  }
  // ---- epilogue: partial tile [ldPart x 128] of (nb, tm); f64 MFMA C/D layout:
  // lane holds D[row = (lane>>4) + 4*reg][col = lane&15]
  double *pt = static_cast<double *>(a.partial) + ((long long)(nb * a.T + tm)) * ((long long)a.ldPart * CALS_BN);
  const int cl = wave * 16 + lcol;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * t + krow + 4 * r;
      pt[m + (long long)a.ldPart * cl] = acc[t][r];
    }
  }
}

// ---- host side ----
static const int kMtSet[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 19, 20};

int mttkrp_pick_mt(int m_tiles) {
  for (int v : kMtSet)
    if (v >= m_tiles) return v;
  return 0;
}

template <int MT, int VAR>
static hipError_t launch_mt_var(int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  static AttrOnce attr_once;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mttkrp_kernel<MT, VAR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       MtCfg<MT>::LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  dim3 grid(a.grid, m_blocks, 1), block(512, 1, 1);
  hipLaunchKernelGGL((mttkrp_kernel<MT, VAR>), grid, block, MtCfg<MT>::LDS_BYTES, st, a);
  return hipGetLastError();
}

static int mttkrp_variant() {
  static int v = -1;
  if (v < 0) {
    const char *s = getenv("CALS_MTTKRP_VARIANT");
    v = s ? atoi(s) & 7 : 3;
  }
  return v;
}

template <int MT>
static hipError_t launch_mt(int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  switch (mttkrp_variant()) {
    case 0: return launch_mt_var<MT, 0>(m_blocks, a, st);
    case 4: return launch_mt_var<MT, 4>(m_blocks, a, st);
    default: return launch_mt_var<MT, 3>(m_blocks, a, st);
  }
}

size_t mttkrp_lds_bytes(int MT) {
  switch (MT) {
#define CASE(N) case N: return MtCfg<N>::LDS_BYTES;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14)
    CASE(16) CASE(19) CASE(20)
#undef CASE
  }
  return 0;
}

hipError_t mttkrp_launch(int MT, int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  switch (MT) {
#define CASE(N) case N: return launch_mt<N>(m_blocks, a, st);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14)
    CASE(16) CASE(19) CASE(20)
#undef CASE
  }
  return hipErrorInvalidValue;
}

// Khatri-Rao of the streamed modes (N > 3): Q[s, c] = prod_k F_k[i_k(s), c], first mode fastest.
// Restates khatri_rao_rec's ordering (src/utils/mttkrp.cpp:147-176).  Small next to the MTTKRP.
template <typename E>
__global__ void krp_kernel(const KrpArgs a) {
  const long long total = a.S * a.R;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long c = e / a.S;
    long long s = e - c * a.S;
    double v = 1.0;
    for (int k = 0; k < a.n; ++k) {
      const long long i = s % a.dims[k];
      s /= a.dims[k];
      v *= (double)static_cast<const E *>(a.F[k])[i + a.ld[k] * c];
    }
    static_cast<E *>(a.Q)[e] = (E)v;
  }
}

hipError_t krp_launch(const KrpArgs &a, hipStream_t st) {
  const long long total = a.S * a.R;
  if (total <= 0) return hipSuccess;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (a.dtype == CALS_F32)
    hipLaunchKernelGGL(krp_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(krp_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace calship
