// Fused MTTKRP of the multi-factor block on gfx950, default schedule ("v3"), fp64 and fp32 storage:
//
//   G[m, c] = sum_{a, s} Xp[m, a, s] * P[a, c] * Q[s, c]          c < R (all in-flight models)
//
// = mttkrp::mttkrp of the reference (src/utils/mttkrp.cpp:562-614) with the Khatri-Rao product
// (:78-216) never materialised: the B operand of every v_mfma_{f64,f32}_16x16x4 is formed in
// registers as P[a,c]*Q[s,c].  Data layout and work decomposition: DESIGN.md section 2/3.1.
// Wave tile: 8 waves x (MT m-tiles x 16 columns).  The stage-boundary bubble of a plain
// double-buffered schedule (round 1's first kernel idled ~1800 of every 21 300 cycles because all 8
// waves left the end-of-stage barrier together and refilled their operand pipelines at the same
// time) is removed:
//   * one slab per stage, a ring of THREE LDS buffers;
//   * the only barrier of a stage sits in the MIDDLE of its MFMA stream (waves 0-3) or at its
//     start (waves 4-7: stagger, MI355X_MICROARCH "two waves per SIMD" item 9): there "slab i+1 has
//     landed for everybody" and "everybody has left slab i-1, so its buffer may be overwritten" are
//     both true; DMA(i+2) is issued right behind it, one 1-KiB piece per MFMA step, in the shadow
//     of the MFMAs;
//   * no barrier at the stage boundary: each wave refills its operand ring for the next slab on its
//     own schedule.
// A operands and Q are read by inline-asm ds_read with COUNTED s_waitcnt lgkmcnt(n) (LDS returns in
// order, so "all but the n youngest reads" is exactly "operand I has landed"; hipcc's own waits for
// this pattern are lgkmcnt(0)).  The reads are invisible to hipcc's counters, so every wait is placed
// here; sched_barrier(0) keeps the MFMA below its wait (guide rule 18); no in-flight inline-asm load
// is live across the loop back-edge (hipcc copies registers there -- found as a real bug).
// Nothing the loop consumes comes from an ordinary global load (hipcc would drain vmcnt(0) inside
// the MFMA loop): P is loaded by an inline-asm load with its own wait once per a-block.
#include "mfma_common.h"

namespace calship {

#ifndef CALS_V3_RING
#define CALS_V3_RING 6
#endif

template <int MT, typename T>
struct Mt3Cfg {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int LDL = (MT % 2 == 1) ? 16 * MT : 16 * MT + 16;  // elements; = 16 mod 32
  static constexpr int SLAB = 16 * LDL;              // elements per slab
  static constexpr int BUF = SLAB + CALS_BN;         // + the slab's 128 Q values
  static constexpr int PE = 1024 / ES;               // elements per 1-KiB DMA piece
  static constexpr int PIECES = SLAB / PE;           // LDL/8 (f64) or LDL/16 (f32): exact
  static constexpr int NP = (PIECES + 7) / 8;        // slab pieces per wave (upper bound)
  static constexpr int QPE = 256 / ES;               // Q elements per 4-byte-per-lane DMA piece
  static constexpr int QPIECES = CALS_BN / QPE;      // 4 (f64) or 2 (f32)
  static constexpr int LDS_BYTES = 3 * BUF * ES;
  static constexpr int N = 4 * MT;                   // MFMAs (= A operands) per slab per wave
  static constexpr int RING = std::is_same<T, float>::value ? CALS_V3_RING + 2 : CALS_V3_RING;
  static constexpr int D = N < RING ? N : RING;      // operand ring depth
  static constexpr int H = N / 2;                    // barrier position
};

template <int MT, typename T>
struct Pipe3 {
  typedef Mt3Cfg<MT, T> C;
  typedef typename Acc<T>::type acc_t;

  // one MFMA step on slab operand I: counted wait, MFMA, refill of the ring slot with operand I+D
  template <int I>
  static __device__ __forceinline__ void step(acc_t (&acc)[MT], T (&ring)[C::D], const T (&bq)[4],
                                              unsigned base) {
    // reads in flight behind operand I: the younger operands of this slab
    constexpr int outstanding = (C::D - 1 < C::N - 1 - I) ? C::D - 1 : C::N - 1 - I;
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
    __builtin_amdgcn_sched_barrier(0);
    constexpr int q = I / MT, t = I % MT;
    acc[t] = Acc<T>::mfma(ring[I % C::D], bq[q], acc[t]);
    if constexpr (I + C::D < C::N) {
      constexpr int qn = (I + C::D) / MT, tn = (I + C::D) % MT;
      lds_read_off<T, ((4 * qn) * C::LDL + 16 * tn) * C::ES>(ring[I % C::D], base);
    }
  }
  template <int I>
  static __device__ __forceinline__ void preload(T (&ring)[C::D], unsigned base) {
    if constexpr (I < C::D) {
      constexpr int qn = I / MT, tn = I % MT;
      lds_read_off<T, ((4 * qn) * C::LDL + 16 * tn) * C::ES>(ring[I], base);
      preload<I + 1>(ring, base);
    }
  }
};

template <int MT, typename T>
__global__ void __launch_bounds__(512, 2) mttkrp3_kernel(const MttkrpArgs a) {
  typedef Mt3Cfg<MT, T> C;
  typedef Pipe3<MT, T> P3;
  typedef typename Acc<T>::type acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  T *const lds = reinterpret_cast<T *>(lds_raw);
  const T *const Xp = static_cast<const T *>(a.Xp);
  const T *const Pm = static_cast<const T *>(a.P);
  const T *const Qm = static_cast<const T *>(a.Q);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int krow = lane >> 4;
  const int lcol = lane & 15;
  const unsigned long long dbg_t0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long dbg_r0 = DIAG(a.dbg_clock) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // XCD-aware bijective remap: workgroups that share an XCD (same blockIdx % 8) get consecutive
  // p, i.e. the same team member of neighbouring column blocks => they stream the same Xp slabs
  // through one L2 (speed only; any placement is correct).
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

  acc_t acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (acc_t){0, 0, 0, 0};

  // ---- per-lane DMA source offsets (stage independent) ----
  // piece k of this wave is global piece k*8 + wave: LDS element e = piece*PE + lane*(16/ES)
  long long src_off[C::NP];
#pragma unroll
  for (int k = 0; k < C::NP; ++k) {
    const int piece = k * 8 + wave;
    const int e = piece * C::PE + lane * (16 / C::ES);
    const int acol = e / C::LDL;
    const int m = e - acol * C::LDL;
    int gm = m0 + m;
    gm = gm < a.Mp ? gm : 0;  // rows past the padded tensor: any valid address, unused
    src_off[k] = gm + (long long)a.Mp * acol;
  }
  // Q piece (4-byte LDS-DMA per lane): f64 = two lanes per value, waves 0..3; f32 = one lane per
  // value, waves 0..1
  long long q_off;
  int q_byte;
  {
    const int d = wave * C::QPE + (C::ES == 8 ? (lane >> 1) : lane);
    int c = nb * CALS_BN + d;
    c = (c < a.R && d < CALS_BN) ? c : 0;  // columns past R: any valid address, result never read
    q_off = a.ldQ * c;
    q_byte = (C::ES == 8) ? 4 * (lane & 1) : 0;
  }
  const long long slab_stride_s = (long long)a.Mp * a.Ap;

  auto issue_piece = [&]<int K>(const T *src_slab, const T *q_row, T *dst) {
    // K < NP: slab piece; K == NP: the Q piece
    if constexpr (K < C::NP) {
      const int piece = K * 8 + wave;
      if (piece < C::PIECES)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(src_slab + src_off[K]),
                                         (LDS_AS void *)(dst + piece * C::PE), 16, 0, 0);
    } else if (wave < C::QPIECES) {
      const char *src = (const char *)(q_row + q_off) + q_byte;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (LDS_AS void *)(dst + C::SLAB + wave * C::QPE), 4, 0, 0);
    }
  };
  auto issue_all = [&](int ab, int s, int bufi) {
    const T *src_slab = Xp + (long long)a.Mp * (16 * ab) + slab_stride_s * s;
    const T *q_row = Qm + s;
    T *dst = lds + bufi * C::BUF;
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
      (issue_piece.template operator()<Ks>(src_slab, q_row, dst), ...);
    }(std::make_integer_sequence<int, C::NP + 1>{});
  };

  // 32-bit loop state (the launcher refuses S or a workgroup's unit count >= 2^31): 64-bit counters
  // went through VALU compares in every stage
  const int Si = (int)S;
  const int n_units = (int)(u_end - u_begin);
  int ab_c = 0, s_c = 0;
  if (n_units > 0) {
    ab_c = (int)(u_begin / S);
    s_c = (int)(u_begin - ab_c * S);
    issue_all(ab_c, s_c, 0);
  }
  const int ab_first = ab_c, s_first = s_c;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // (ab, s) of unit u+1 and u+2, kept incrementally (no 64-bit division in the loop)
  int ab_1 = ab_c, s_1 = s_c + 1;
  if (s_1 >= Si) { s_1 = 0; ab_1++; }
  if (n_units > 1) issue_all(ab_1, s_1, 1);
  int ab_2 = ab_1, s_2 = s_1 + 1;
  if (s_2 >= Si) { s_2 = 0; ab_2++; }

  const unsigned lane_off = (unsigned)((krow * C::LDL + lcol) * C::ES);
  const unsigned lds0 = (unsigned)(size_t)((LDS_AS const char *)lds_raw);
  const unsigned q_lane_off = (unsigned)((C::SLAB + wave * 16 + lcol) * C::ES);

  T preg[4] = {0, 0, 0, 0};
  int buf = 0;
  auto load_p = [&](int ab) {
    // P[16ab + 4q + krow, col]: inline-asm load with its own wait (once per S units)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int arow = 16 * ab + 4 * q + krow;
      const bool ok = (arow < a.A) && cvalid;
      const T *ptr = Pm + (ok ? arow + a.ldP * col : 0);
      T v;
      if constexpr (C::ES == 8)
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ptr) : "memory");
      else
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(ptr) : "memory");
      preg[q] = ok ? v : (T)0;
    }
  };

  auto unit_loop = [&]<bool LATE>() {
    int ab_loaded = -1;
    for (int iu = 0; iu < n_units; ++iu) {
      const int buf_n = (buf == 2) ? 0 : buf + 1;
      const int buf_nn = (buf_n == 2) ? 0 : buf_n + 1;
      const unsigned base = lds0 + (unsigned)(buf * C::BUF * C::ES) + lane_off;
      // fp64: the DMA of unit iu + 2 is issued unconditionally -- in the last two stages it re-reads the
      // range's first slab into the buffer nobody consumes any more (waited for after the loop) --
      // instead of a uniform branch in front of every DMA instruction (ttm_kernel.hip has the numbers;
      // the fp32 kernels measured slower with it)
      constexpr bool ALWAYS = (C::ES == 8);
      const bool in_range = iu + 2 < n_units;
      const bool fetch = (ALWAYS || in_range) && !DIAG(a.dbg_no_dma);
      const int ab_f = (in_range || !ALWAYS) ? ab_2 : ab_first;
      const int s_f = (in_range || !ALWAYS) ? s_2 : s_first;
      const T *src_slab = Xp + (long long)a.Mp * (16 * ab_f) + slab_stride_s * (long long)s_f;
      const T *q_row = Qm + s_f;
      T *dst = lds + buf_nn * C::BUF;

      if constexpr (LATE) {
        // barrier #iu at the start of slab iu: DMA(iu+1) landed everywhere; slab iu-1 is finished
        // by everybody (waves 0-3 are in the middle of slab iu), so DMA(iu+2) may overwrite it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!DIAG(a.dbg_no_barrier)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ab_c != ab_loaded) {
        load_p(ab_c);
        ab_loaded = ab_c;
      }
      // Q of this slab, then the first D operands (ring local to the stage)
      T ring[C::D];
      T qcur;
      lds_read<T>(qcur, lds0 + (unsigned)(buf * C::BUF * C::ES) + q_lane_off);
      P3::template preload<0>(ring, base);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(C::D));  // Q landed (D younger reads in flight)
      __builtin_amdgcn_sched_barrier(0);
      T bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = preg[q] * qcur;

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

      if constexpr (!LATE) {
        // barrier #iu in the middle of slab iu (same invariants)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!DIAG(a.dbg_no_barrier)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }

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

      buf = buf_n;
      ab_c = ab_1;
      s_c = s_1;
      ab_1 = ab_2;
      s_1 = s_2;
      s_2 = s_2 + 1;
      if (s_2 >= Si) { s_2 = 0; ab_2++; }
    }
  };
  // DMA pieces ride on the MFMA steps of one half of a slab: the half must have NP + 1 of them
  static_assert(C::NP + 1 <= C::H && C::NP + 1 <= C::N - C::H, "slab too short for its DMA pieces");
  if (wave < 4 || DIAG(a.dbg_no_stagger))
    unit_loop.template operator()<false>();
  else
    unit_loop.template operator()<true>();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (DIAG(a.dbg_clock) && tid == 0) {
    const int wg = blockIdx.x + gridDim.x * blockIdx.y;
    a.dbg_clock[2 * wg] = __builtin_amdgcn_s_memtime() - dbg_t0;
    a.dbg_clock[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
  }

  // ---- epilogue: partial tile [ldPart x 128] of (nb, tm)
  T *pt = static_cast<T *>(a.partial) + ((long long)(nb * a.T + tm)) * ((long long)a.ldPart * CALS_BN);
  const int cl = wave * 16 + lcol;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * t + Acc<T>::row(krow, r);
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

template <int MT, typename T>
static hipError_t launch3_mt(int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  static AttrOnce attr_once;
  hipError_t e = attr_once.ensure([] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&mttkrp3_kernel<MT, T>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, Mt3Cfg<MT, T>::LDS_BYTES);
  });
  if (e != hipSuccess) return e;
  dim3 grid(a.grid, m_blocks, 1), block(512, 1, 1);
  constexpr int lds_bytes = Mt3Cfg<MT, T>::LDS_BYTES;
  hipLaunchKernelGGL((mttkrp3_kernel<MT, T>), grid, block, lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t mttkrp3_launch(int MT, int m_blocks, const MttkrpArgs &a, hipStream_t st) {
  // the kernel keeps its stage counters in 32 bits
  if (a.S >= (1ll << 31) || ((long long)(a.Ap >> 4) * a.S) / (a.T > 0 ? a.T : 1) >= (1ll << 31) - 2)
    return hipErrorInvalidValue;
  if (a.dtype == CALS_F32) {
    switch (MT) {
#define CASE(N) case N: return launch3_mt<N, float>(m_blocks, a, st);
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(10) CASE(12) CASE(14)
      CASE(16) CASE(19) CASE(20)
#undef CASE
    }
    return hipErrorInvalidValue;
  }
  switch (MT) {
#define CASE(N) case N: return launch3_mt<N, double>(m_blocks, a, st);
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
