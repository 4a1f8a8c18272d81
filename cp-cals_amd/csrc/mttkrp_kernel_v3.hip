// Fused MTTKRP, third schedule ("v3"): same mathematics, layout and wave tile as mttkrp_kernel.hip
//   G[m, c] = sum_{a, s} Xp[m, a, s] * P[a, c] * Q[s, c],   8 waves x (MT m-tiles x 16 columns),
// but the stage boundary bubble is removed.  Measured on v1 (tools/clock_probe.py): the pipe
// idles ~1800 of every 21 300 cycles because all 8 waves leave the end-of-stage barrier together and
// refill their operand pipelines (LDS latency, address set-up) at the same time.  Here:
//   * one slab per stage, a ring of THREE LDS buffers;
//   * the only barrier of a stage sits in the MIDDLE of its MFMA stream, where every wave's
//     operand ring is full: "DMA(i+1) has landed for everybody" and "everybody has left stage i-1,
//     so its buffer may be overwritten" are both true there; DMA(i+2) is issued right behind it,
//     one 1-KiB piece per MFMA step, in the shadow of the 64-cycle MFMAs;
//   * no barrier at the stage boundary: each wave refills its operand ring for the next slab on its
//     own schedule, so the two waves of a SIMD do it at different times and cover each other.
// A operands and Q are read by inline-asm ds_read_b64 with counted s_waitcnt lgkmcnt (see SlabPipe
// in mttkrp_kernel.hip for why); nothing else in the loop touches the LGKM counter.
#include "cals_hip_internal.h"

#include <utility>

namespace calship {

typedef double v4d __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// CALS_DIAG builds keep the in-kernel stamps and the timing-only switches (tools/clock_probe.py,
// tools/time_probe.py); the production build compiles them out of the MFMA loop.
#ifdef CALS_DIAG
#define DIAG(x) (x)
#else
#define DIAG(x) (false)
#endif

#ifndef CALS_V3_RING
#define CALS_V3_RING 6
#endif

template <int MT>
struct Mt3Cfg {
  static constexpr int RING = CALS_V3_RING;
  static constexpr int LDL = (MT % 2 == 1) ? 16 * MT : 16 * MT + 16;
  static constexpr int SLAB = 16 * LDL;        // doubles per slab
  static constexpr int BUF = SLAB + CALS_BN;   // + the slab's 128 Q values
  static constexpr int PIECES = LDL / 8;       // 1 KiB DMA pieces per slab
  static constexpr int NP = (PIECES + 7) / 8;  // pieces per wave (upper bound)
  static constexpr int LDS_BYTES = 3 * BUF * 8;
  static constexpr int N = 4 * MT;             // MFMAs (= A operands) per slab per wave
  static constexpr int D = N < RING ? N : RING;  // operand ring depth
  static constexpr int H = N / 2;              // barrier position
};

// per-wave DMA state: everything issue_piece needs, precomputed per stage
struct DmaCtx {
  const double *src_slab;  // Xp slab base for the stage being fetched
  const double *q_src;     // &Q[s, 0] for that stage
  double *dst;             // LDS buffer base (slab then Q)
};

template <int MT>
struct Pipe3 {
  typedef Mt3Cfg<MT> C;

  // one step: wait for operand I, MFMA, refill the ring slot (from this slab, or - near the end -
  // from the next one), and on the steps behind the barrier issue one DMA piece
  // one MFMA step on slab operand I: counted wait, MFMA, refill of the ring slot with operand I+D
  template <int I>
  static __device__ __forceinline__ void step(v4d (&acc)[MT], double (&ring)[C::D],
                                              const double (&bq)[4], unsigned base) {
    // reads in flight behind operand I: the younger operands of this slab
    constexpr int outstanding = (C::D - 1 < C::N - 1 - I) ? C::D - 1 : C::N - 1 - I;
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
    __builtin_amdgcn_sched_barrier(0);
    constexpr int q = I / MT, t = I % MT;
    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[I % C::D], bq[q], acc[t], 0, 0, 0);
    if constexpr (I + C::D < C::N) {
      constexpr int qn = (I + C::D) / MT, tn = (I + C::D) % MT;
      asm volatile("ds_read_b64 %0, %1 offset:%2"
                   : "=v"(ring[I % C::D])
                   : "v"(base), "i"(((4 * qn) * C::LDL + 16 * tn) * 8));
    }
  }
  // the ring is local to a stage: an in-flight inline-asm load must never be live across the loop
  // back-edge, where hipcc may copy registers
  template <int I>
  static __device__ __forceinline__ void preload(double (&ring)[C::D], unsigned base) {
    if constexpr (I < C::D) {
      constexpr int qn = I / MT, tn = I % MT;
      asm volatile("ds_read_b64 %0, %1 offset:%2"
                   : "=v"(ring[I])
                   : "v"(base), "i"(((4 * qn) * C::LDL + 16 * tn) * 8));
      preload<I + 1>(ring, base);
    }
  }
};

template <int MT>
__global__ void __launch_bounds__(512, 2) mttkrp3_kernel(const MttkrpArgs a) {
  typedef Mt3Cfg<MT> C;
  typedef Pipe3<MT> P3;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int krow = lane >> 4;
  const int lcol = lane & 15;
  const unsigned long long dbg_t0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long dbg_r0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // XCD-aware bijective remap (see mttkrp_kernel.hip)
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
  const long long u_end = DIAG(a.dbg_no_units) ? u_begin : U * (tm + 1) / a.T;

  const int col = nb * CALS_BN + wave * 16 + lcol;
  const bool cvalid = col < a.R;

  v4d acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};

  // ---- per-lane DMA source offsets (stage independent) ----
  // piece k of this wave is global piece k*8 + wave: LDS element e = piece*128 + lane*2
  long long src_off[C::NP];
#pragma unroll
  for (int k = 0; k < C::NP; ++k) {
    const int piece = k * 8 + wave;
    const int e = piece * 128 + lane * 2;
    const int acol = e / C::LDL;
    const int m = e - acol * C::LDL;
    int gm = m0 + m;
    gm = gm < a.Mp ? gm : 0;  // rows past the padded tensor: any valid address, unused
    src_off[k] = gm + (long long)a.Mp * acol;
  }
  // Q piece: waves 0..3 fetch 32 doubles each by 4-byte LDS-DMA (two lanes per double)
  long long q_off;
  {
    const int d = wave * 32 + (lane >> 1);
    int c = nb * CALS_BN + d;
    c = c < a.R ? c : 0;  // columns past R: any valid address, result never read
    q_off = a.ldQ * c;    // in doubles; + 4*(lane&1) bytes added at issue
  }
  const long long slab_stride_s = (long long)a.Mp * a.Ap;

  auto issue_piece = [&]<int K>(const double *src_slab, const double *q_row, double *dst) {
    // K < NP: slab piece; K == NP: the Q piece (waves 0..3)
    if constexpr (K < C::NP) {
      const int piece = K * 8 + wave;
      if (piece < C::PIECES)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(src_slab + src_off[K]),
                                         (LDS_AS void *)(dst + piece * 128), 16, 0, 0);
    } else if (wave < 4) {
      const char *src = (const char *)(q_row + q_off) + 4 * (lane & 1);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (LDS_AS void *)(dst + C::SLAB + wave * 32), 4, 0, 0);
    }
  };
  auto issue_all = [&](long long ab, long long s, int bufi) {
    const double *src_slab = a.Xp + (long long)a.Mp * (16 * ab) + slab_stride_s * s;
    const double *q_row = a.Q + s;
    double *dst = lds + bufi * C::BUF;
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
      (issue_piece.template operator()<Ks>(src_slab, q_row, dst), ...);
    }(std::make_integer_sequence<int, C::NP + 1>{});
  };

  const long long n_units = u_end - u_begin;
  long long ab_c = 0, s_c = 0;
  if (n_units > 0) {
    ab_c = u_begin / S;
    s_c = u_begin - ab_c * S;
    issue_all(ab_c, s_c, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // (ab, s) of unit u+1 and u+2, kept incrementally
  long long ab_1 = ab_c, s_1 = s_c + 1;
  if (s_1 >= S) { s_1 = 0; ab_1++; }
  if (n_units > 1) issue_all(ab_1, s_1, 1);  // (kept under dbg_no_dma: results are then garbage anyway)
  long long ab_2 = ab_1, s_2 = s_1 + 1;
  if (s_2 >= S) { s_2 = 0; ab_2++; }

  const unsigned lane_off = (unsigned)((krow * C::LDL + lcol) * 8);
  const unsigned lds0 = (unsigned)(size_t)((LDS_AS const char *)lds);
  const unsigned q_lane_off = (unsigned)((C::SLAB + wave * 16 + lcol) * 8);

  double preg[4] = {0.0, 0.0, 0.0, 0.0};
  int buf = 0;
  auto load_p = [&](long long ab) {
    // P[16ab + 4q + krow, col]: inline-asm load with its own wait (once per S units)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int arow = (int)(16 * ab) + 4 * q + krow;
      const bool ok = (arow < a.A) && cvalid;
      const double *ptr = a.P + (ok ? arow + a.ldP * col : 0);
      double v;
      asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(v)
                   : "v"(ptr)
                   : "memory");
      preg[q] = ok ? v : 0.0;
    }
  };

  // Stagger (MI355X_MICROARCH "Two waves per SIMD", item 9): the two waves of a SIMD run the same
  // program; with the barrier at the same program point they stay in lockstep and hit their
  // slab-switch bubble (Q read + ring refill, ~250 cycles) together.  Waves 0-3 therefore take
  // the stage's barrier in the MIDDLE of a slab, waves 4-7 at the START of the same slab: the
  // invariants hold for both (see header), and one half is always in full MFMA flow while the
  // other refills.
  unsigned long long dbg_bar = 0, dbg_top = 0, dbg_half1 = 0, dbg_half2 = 0;
  auto unit_loop = [&]<bool LATE>() {
    long long ab_loaded = -1;
    for (long long iu = 0; iu < n_units; ++iu) {
      const int buf_n = (buf == 2) ? 0 : buf + 1;
      const int buf_nn = (buf_n == 2) ? 0 : buf_n + 1;
      const unsigned base = lds0 + (unsigned)(buf * C::BUF * 8) + lane_off;
      const bool fetch = (iu + 2 < n_units) && !DIAG(a.dbg_no_dma);
      const double *src_slab = a.Xp + (long long)a.Mp * (16 * ab_2) + slab_stride_s * s_2;
      const double *q_row = a.Q + s_2;
      double *dst = lds + buf_nn * C::BUF;

      if constexpr (LATE) {
        // barrier #iu at the start of slab iu: DMA(iu+1) landed everywhere; slab iu-1 is finished
        // by everybody (waves 0-3 are in the middle of slab iu), so DMA(iu+2) may overwrite it
        const unsigned long long s0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!DIAG(a.dbg_no_barrier)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG(a.dbg_clock)) dbg_bar += __builtin_amdgcn_s_memtime() - s0;
      }
      const unsigned long long st0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
      if (ab_c != ab_loaded) {
        load_p(ab_c);
        ab_loaded = ab_c;
      }
      // Q of this slab, then the first D operands (ring local to the stage: no in-flight
      // inline-asm load is live across the loop back-edge)
      double ring[C::D];
      double qcur;
      asm volatile("ds_read_b64 %0, %1" : "=v"(qcur) : "v"(lds0 + (unsigned)(buf * C::BUF * 8) + q_lane_off));
      P3::template preload<0>(ring, base);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(C::D));  // Q landed (D younger reads in flight)
      __builtin_amdgcn_sched_barrier(0);
      double bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = preg[q] * qcur;
      const unsigned long long st1 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
      if (DIAG(a.dbg_clock)) dbg_top += st1 - st0;

      // ---- first half (LATE: DMA(iu+2) one piece per MFMA step) ----
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (
            [&] {
              P3::template step<Is>(acc, ring, bq, base);
              if constexpr (LATE && Is <= C::NP) {
                if (fetch) issue_piece.template operator()<Is>(src_slab, q_row, dst);
              }
            }(),
            ...);
      }(std::make_integer_sequence<int, C::H>{});

      const unsigned long long st2 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
      if (DIAG(a.dbg_clock)) dbg_half1 += st2 - st1;
      if constexpr (!LATE) {
        // barrier #iu in the middle of slab iu (same invariants)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!DIAG(a.dbg_no_barrier)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG(a.dbg_clock)) dbg_bar += __builtin_amdgcn_s_memtime() - st2;
      }
      const unsigned long long st3 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;

      // ---- second half (!LATE: DMA(iu+2) one piece per MFMA step) ----
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (
            [&] {
              P3::template step<C::H + Is>(acc, ring, bq, base);
              if constexpr (!LATE && Is <= C::NP) {
                if (fetch) issue_piece.template operator()<Is>(src_slab, q_row, dst);
              }
            }(),
            ...);
      }(std::make_integer_sequence<int, C::N - C::H>{});

      if (DIAG(a.dbg_clock)) dbg_half2 += __builtin_amdgcn_s_memtime() - st3;
      buf = buf_n;
      ab_c = ab_1;
      s_c = s_1;
      ab_1 = ab_2;
      s_1 = s_2;
      s_2 = s_2 + 1;
      if (s_2 >= S) { s_2 = 0; ab_2++; }
    }
  };
  if (DIAG(a.dbg_prio == 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (DIAG(a.dbg_prio == 2) && wave < 4) __builtin_amdgcn_s_setprio(1);
  if (wave < 4 || DIAG(a.dbg_no_stagger))
    unit_loop.template operator()<false>();
  else
    unit_loop.template operator()<true>();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (DIAG(a.dbg_clock) && lane == 0) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    if (wave == 0) {
      a.dbg_clock[2 * wg] = __builtin_amdgcn_s_memtime() - dbg_t0;
      a.dbg_clock[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    }
    if (wg < 32) {
      unsigned long long *o = a.dbg_clock + 8192 + (wg * 8 + wave) * 5;
      o[0] = dbg_bar; o[1] = dbg_top; o[2] = dbg_half1; o[3] = dbg_half2; o[4] = (unsigned long long)n_units;
    }
  }

  // ---- epilogue: partial tile [ldPart x 128] of (nb, tm)
  double *pt = a.partial + ((long long)(nb * a.T + tm)) * ((long long)a.ldPart * CALS_BN);
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

template <int MT>
static hipError_t launch3_mt(int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mttkrp3_kernel<MT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       Mt3Cfg<MT>::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(a.grid, m_blocks, 1), block(512, 1, 1);
  hipLaunchKernelGGL(mttkrp3_kernel<MT>, grid, block, Mt3Cfg<MT>::LDS_BYTES, st, a);
  return hipGetLastError();
}

hipError_t mttkrp3_launch(int MT, int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  switch (MT) {
#define CASE(N) case N: return launch3_mt<N>(m_blocks, a, st);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14)
    CASE(16) CASE(19) CASE(20)
#undef CASE
  }
  return hipErrorInvalidValue;
}

}  // namespace calship
