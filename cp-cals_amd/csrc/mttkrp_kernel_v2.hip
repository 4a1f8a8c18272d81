// Fused MTTKRP, second tiling ("v2"): two independent 4-wave workgroups per CU.
//
// Same mathematics and data layout as mttkrp_kernel.hip (see there and DESIGN.md):
//   G[m, c] = sum_{a, s} Xp[m, a, s] * P[a, c] * Q[s, c]
// but the workgroup is 4 waves (one per SIMD) with a 2-n-tile (32 column) x MT m-tile wave tile
// (accumulators 16*MT VGPRs, MT <= 10), at most 80 KiB of LDS, so that TWO workgroups are resident
// per CU: while one sits in its end-of-stage barrier / LDS-DMA wait the other one's MFMAs keep the
// matrix pipe busy (the single 8-wave workgroup of v1 leaves the pipe idle ~12 % of the time
// there).  Each A-operand LDS read now feeds two MFMAs.  M is covered by m_blocks workgroup rows
// of MT or MT-1 tiles (balanced split, no padding tile is ever multiplied).
#include "cals_hip_internal.h"

#include <cstdlib>

namespace calship {

typedef double v4d __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int MT>
struct Mt2Cfg {
  static constexpr int LDL = (MT % 2 == 1) ? 16 * MT : 16 * MT + 16;  // LDS leading dim (doubles)
  static constexpr int SLAB = 16 * LDL;                               // doubles per unit slab
  static constexpr int PIECES = LDL / 8;                              // 1 KiB DMA pieces per slab
  static constexpr int SB_RAW = (40 * 1024) / ((SLAB + CALS_BN) * 8);  // 2 stages <= 80 KiB
  static constexpr int SB = SB_RAW < 1 ? 1 : (SB_RAW > 8 ? 8 : SB_RAW);
  static constexpr int STAGE = SLAB * SB;
  static constexpr int QSTAGE = SB * CALS_BN;
  static constexpr int LDS_BYTES = 2 * (STAGE + QSTAGE) * 8;
};

template <int MT, bool EARLY>
__device__ __forceinline__ void mttkrp2_body(const MttkrpArgs &a, double *lds, int tm, int nb,
                                             int m0) {
  typedef Mt2Cfg<MT> C;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int krow = lane >> 4;
  const int lcol = lane & 15;

  const long long S = a.S;
  const long long U = (long long)(a.Ap >> 4) * S;
  const long long u_begin = U * tm / a.T;
  const long long u_end = a.dbg_no_units ? u_begin : U * (tm + 1) / a.T;

  int col[2];
  bool cvalid[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    col[nt] = nb * CALS_BN + wave * 32 + nt * 16 + lcol;
    cvalid[nt] = col[nt] < a.R;
  }

  v4d acc[MT][2];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    acc[t][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    acc[t][1] = (v4d){0.0, 0.0, 0.0, 0.0};
  }

  const long long slab_stride_s = (long long)a.Mp * a.Ap;
  double *const qlds = lds + 2 * C::STAGE;

  auto issue_stage = [&](int buf, long long ab, long long s0, int n) {
#pragma unroll
    for (int j = 0; j < C::SB; ++j) {
      if (j < n) {
        const double *src_slab = static_cast<const double *>(a.Xp) + (long long)a.Mp * (16 * ab) + slab_stride_s * (s0 + j);
        double *dst_slab = lds + buf * C::STAGE + j * C::SLAB;
#pragma unroll
        for (int pc = 0; pc < (C::PIECES + 3) / 4; ++pc) {
          const int piece = pc * 4 + wave;
          if (piece < C::PIECES) {
            const int e = piece * 128 + lane * 2;
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
#pragma unroll
    for (int pc = 0; pc < (4 * C::SB + 3) / 4; ++pc) {
      const int piece = pc * 4 + wave;  // piece = j*4 + quarter, 32 doubles each
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

  long long u = u_begin;
  long long ab_c = 0, s_c = 0;
  int n_c = 0, buf = 0;
  if (u < u_end) {
    ab_c = u / S;
    s_c = u - ab_c * S;
    long long nn = S - s_c;
    if (nn > u_end - u) nn = u_end - u;
    if (nn > C::SB) nn = C::SB;
    n_c = (int)nn;
    issue_stage(0, ab_c, s_c, n_c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  double preg[4][2];
  long long ab_loaded = -1;
  const double *abase = lds + krow * C::LDL + lcol;

  while (u < u_end) {
    // next stage, incrementally (no 64-bit division in the loop)
    const long long un = u + n_c;
    long long ab_n = ab_c, s_n = s_c + n_c;
    if (s_n >= S) {
      s_n = 0;
      ab_n = ab_c + 1;
    }
    int n_n = 0;
    const bool have_next = un < u_end;
    if (have_next) {
      long long nn = S - s_n;
      if (nn > u_end - un) nn = u_end - un;
      if (nn > C::SB) nn = C::SB;
      n_n = (int)nn;
    }
    if (EARLY && have_next) issue_stage(buf ^ 1, ab_n, s_n, n_n);
    if (ab_c != ab_loaded) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int arow = (int)(16 * ab_c) + 4 * q + krow;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bool ok = (arow < a.A) && cvalid[nt];
          const double *ptr = static_cast<const double *>(a.P) + (ok ? arow + a.ldP * col[nt] : 0);
          double v;
          asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(v)
                       : "v"(ptr)
                       : "memory");
          preg[q][nt] = ok ? v : 0.0;
        }
      }
      ab_loaded = ab_c;
    }

#pragma unroll 1
    for (int j = 0; j < n_c; ++j) {
      const double *qp = qlds + buf * C::QSTAGE + j * CALS_BN + wave * 32 + lcol;
      const double q0 = qp[0], q1 = qp[16];
      double bq[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bq[q][0] = preg[q][0] * q0;
        bq[q][1] = preg[q][1] * q1;
      }
      const volatile LDS_AS double *vsl =
          (const volatile LDS_AS double *)(abase + buf * C::STAGE + j * C::SLAB);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const double av = vsl[(4 * q) * C::LDL + 16 * t];
          acc[t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[q][0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[q][1], acc[t][1], 0, 0, 0);
        }
        if (!EARLY && q == 0 && j == 0 && have_next) issue_stage(buf ^ 1, ab_n, s_n, n_n);
      }
      constexpr int PF = (4 * MT) < 3 ? (4 * MT) : 3;
      __builtin_amdgcn_sched_group_barrier(0x100, 2 + PF, 0);  // q0, q1 + PF A operands
#pragma unroll
      for (int i = 0; i < 4 * MT - PF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * PF, 0);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    u = un;
    ab_c = ab_n;
    s_c = s_n;
    n_c = n_n;
    buf ^= 1;
  }

  double *pt = static_cast<double *>(a.partial) + ((long long)(nb * a.T + tm)) * ((long long)a.ldPart * CALS_BN);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int cl = wave * 32 + nt * 16 + lcol;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * t + krow + 4 * r;
        pt[m + (long long)a.ldPart * cl] = acc[t][nt][r];
      }
    }
  }
}

template <int MT, bool EARLY>
__global__ void __launch_bounds__(256, 2) mttkrp2_kernel(const MttkrpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const unsigned long long dbg_t0 = a.dbg_clock ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long dbg_r0 = a.dbg_clock ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // XCD-aware bijective remap (see mttkrp_kernel.hip): consecutive p share an XCD's L2
  const int G = a.grid;
  const int b = blockIdx.x;
  const int xcd = b & 7, q8 = G >> 3, r8 = G & 7;
  const int p = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  if (a.loop_mblocks) {
    // every workgroup walks all M blocks over the same unit range (two passes at I = 300): all
    // workgroups carry identical work, so no CU ends up with two "big" blocks
    const int tm = p / a.NB;
    const int nb = p - tm * a.NB;
    for (int mb = 0; mb < a.m_blocks; ++mb) {
      if (mb < a.k_big) {
        mttkrp2_body<MT, EARLY>(a, lds, tm, nb, 16 * mb * MT);
      } else {
        if constexpr (MT > 1)
          mttkrp2_body<MT - 1, EARLY>(a, lds, tm, nb, 16 * (a.k_big * MT + (mb - a.k_big) * (MT - 1)));
      }
      __syncthreads();  // LDS is reused by the next pass
    }
    if (a.dbg_clock && threadIdx.x == 0) {
      a.dbg_clock[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - dbg_t0;
      a.dbg_clock[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
      a.dbg_clock[4096 + 2 * blockIdx.x] = dbg_r0;
      a.dbg_clock[4096 + 2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID
    }
    return;
  }
  const int per_tm = a.NB * a.m_blocks;
  const int tm = p / per_tm;
  const int rem = p - tm * per_tm;
  const int mb = rem / a.NB;
  const int nb = rem - mb * a.NB;
  if (mb < a.k_big) {
    mttkrp2_body<MT, EARLY>(a, lds, tm, nb, 16 * mb * MT);
  } else {
    if constexpr (MT > 1)
      mttkrp2_body<MT - 1, EARLY>(a, lds, tm, nb, 16 * (a.k_big * MT + (mb - a.k_big) * (MT - 1)));
  }
}

template <int MT, bool EARLY>
static hipError_t launch2_mt_e(const MttkrpArgs &a, hipStream_t st) {
  static AttrOnce attr_once;
  constexpr int lds_bytes = Mt2Cfg<MT>::LDS_BYTES > Mt2Cfg<(MT > 1 ? MT - 1 : 1)>::LDS_BYTES
                                ? Mt2Cfg<MT>::LDS_BYTES
                                : Mt2Cfg<(MT > 1 ? MT - 1 : 1)>::LDS_BYTES;
  if (attr_once.need()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mttkrp2_kernel<MT, EARLY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((mttkrp2_kernel<MT, EARLY>), dim3(a.grid), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

template <int MT>
static hipError_t launch2_mt(const MttkrpArgs &a, hipStream_t st) {
  static int early = -1;
  if (early < 0) {
    const char *s = getenv("CALS_MTTKRP2_EARLY");
    early = (s && atoi(s)) ? 1 : 0;
  }
  return early ? launch2_mt_e<MT, true>(a, st) : launch2_mt_e<MT, false>(a, st);
}

hipError_t mttkrp2_launch(int MT, const MttkrpArgs &a, hipStream_t st) {
  switch (MT) {
#define CASE(N) case N: return launch2_mt<N>(a, st);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace calship
