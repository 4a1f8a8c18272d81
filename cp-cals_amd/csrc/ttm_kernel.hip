// Dimension-tree pair of MTTKRPs for 3-way tensors on gfx950 (fp64 and fp32 storage).
//
// Two consecutive modes of a sweep share the contraction of X with the factor that neither of them
// updates (the reference's two-step MTTKRP, src/utils/mttkrp.cpp:330-560, applied to the whole
// multi-factor block and shared between two modes):
//
//   T[m, s, c] = sum_a Xp[m, a, s] * P[a, c]                        ("TTM", the MFMA GEMM)
//   G_first [m, c] = sum_s T[m, s, c] * Q[s, c]                     (fused here, in registers)
//   G_second[s, c] = sum_m T[m, s, c] * F[m, c]                     (contract_kernel, HBM-bound)
//
// where `first` is the mode indexed by m, `second` the mode indexed by s (updated right after
// `first` in the sweep), a the third mode, P/Q the CURRENT factors of modes a/s and F the factor of
// mode `first` AFTER its update.  T depends on P only, which is not touched between the two modes,
// so the second MTTKRP costs one pass over T instead of a second 2*I*J*K*R-flop GEMM.
//
// ttm_kernel: the wave tiling, LDS-DMA ring, staggered mid-stage barrier and counted-wait operand
// pipeline of mttkrp_kernel_v3.hip, with three differences:
//   * the B operand of the MFMAs is P[a, c] itself, streamed through LDS next to the X slab from a
//     packed copy Pt[column block][a][128 columns] (pack_pt_kernel) -- a (16 x 128) tile per stage;
//   * units run a-block fastest: a wave accumulates T[:, s, :] over all a in `tacc`, and at the end
//     of every s adds tacc * Q[s, c] into `gacc`, stores tacc to T and clears it;
//   * a workgroup team splits the s range (never the a range), so T needs no cross-workgroup
//     reduction; G_first partials go through the same partial tiles + reduce_partials_kernel as the
//     plain MTTKRP.
// Two accumulator sets bound the tile height: MT <= 10 m-tiles in fp64, <= 20 in fp32; taller modes
// are cut into M blocks, the first k_big of them MT tiles high, the others MT - 1, and every
// workgroup walks all M blocks of its (column block, s range) one after the other, so that blocks
// of different height cost every workgroup the same.
// Operand order: fp32 feeds X as the MFMA's A operand and P as B; fp64 swaps them, which transposes
// the accumulator tile so that the T flush and the partial tiles store whole 128-byte lines straight
// from registers (see TtmBody); fp32 flushes T through a per-wave LDS transpose for the same reason.
#include "mfma_common.h"

namespace calship {

#ifndef CALS_TTM_RING
#define CALS_TTM_RING 6
#endif
#ifndef CALS_TTM_ROLL
#define CALS_TTM_ROLL 1  // fp64: rolling T flush inside the first stage of the next s (see TtmBody)
#endif
// Timing-only stripping, compile time (results garbage; tools/ttm_strip.sh builds the variants): bit 1 no T stores,
// 8 no P DMA, 16 no X DMA, 32 no flush FMAs, 64 no Q reads, 128 no stage barrier, 256 P operands read once,
// 512 no vmcnt wait in front of the barrier, 4096 fp32 flush without the LDS round trip, 8192 only half the tiles stored,
// 16384 T stores carry half (fp64) / a quarter (fp32) of their bytes.  (A probe that ADDED a plain 16-byte-per-lane register
// load per P piece showed that any vector-memory instruction costs what an LDS-DMA piece costs, profiles/r03_ttm_strip_ladder.txt.)  (The run-time CALS_TTM_DBG bits of CALS_DIAG builds cost registers:
// the DIAG kernel spills and runs at half speed -- useless for this.)
#ifndef CALS_TTM_STRIP
#define CALS_TTM_STRIP 0
#endif
// Which waves issue a stage's LDS-DMA: 8 = all of them (a piece every 8th wave), 4 = waves 4-7 only (the group that
// takes its barrier at the stage start; waves 0-3 then carry no DMA instruction and no vmcnt wait at all)
#ifndef CALS_TTM_DMAW
#define CALS_TTM_DMAW 8
#endif
// 1: Q[s, :] fetched and read only in the stage of the last a-block of s (the one flush that consumes it) instead
// of in every stage.  One DMA instruction and four LDS reads less per stage, but the wave-uniform branches that
// select it cost more: measured 2.30 ms against 2.274 ms per launch at C3 -- off.
#ifndef CALS_TTM_EARLYVM
#define CALS_TTM_EARLYVM 0  // experiment: waves 0-3 do not wait for a flush stage's T stores at their mid-stage barrier
#endif
#ifndef CALS_TTM_F32_STORES_BEHIND
#define CALS_TTM_F32_STORES_BEHIND 0  // experiment (no gain, profiles/r04_ttm_f32_ladder.txt): fp32 flush stores stay outstanding across the next barrier
#endif
#ifndef CALS_TTM_QLAST
#define CALS_TTM_QLAST 0
#endif
#define STRIP(bit) ((CALS_TTM_STRIP & (bit)) != 0)

template <int MT, typename T>
struct TtmCfg {
  static constexpr int ES = (int)sizeof(T);
  static constexpr int LDL = (MT % 2 == 1) ? 16 * MT : 16 * MT + 16;  // elements; = 16 mod 32
  static constexpr int SLAB = 16 * LDL;              // X slab: 16 a-rows x LDL
  static constexpr int PE = 1024 / ES;               // elements per 1-KiB DMA piece
  static constexpr int PIECES = SLAB / PE;
  static constexpr int DW = (ES == 8) ? CALS_TTM_DMAW : 8;  // DMA-issuing waves (fp64 experiment; fp32: all)
  static constexpr int NP = (PIECES + DW - 1) / DW;  // X pieces per issuing wave (upper bound)
  static constexpr int QOFF = SLAB;                  // the stage's 128 Q values
  static constexpr int QPE = 256 / ES;               // Q elements per 4-byte-per-lane DMA piece
  static constexpr int QPIECES = CALS_BN / QPE;      // 4 (f64) or 2 (f32)
  // P tile (16 a-rows x 128 columns) behind the Q values.  fp64: one 1-KiB DMA instruction per row (16 B per lane),
  // rows at pitch PP (= 16 mod 32 elements).  fp32: a row is 512 B, so one 16-B-per-lane instruction carries TWO
  // rows, which land contiguously (LDS-DMA writes lane * 16 B behind a wave-uniform base): the odd row of a pair
  // is stored with its 16-column halves swapped (column ^ 16, done on the SOURCE address of the lane), which keeps
  // the operand reads -- 32-lane halves = krow pairs (2j, 2j + 1), ds_read_b32 banks = (a / 4) mod 32 -- conflict
  // free without padding.  8 instructions per stage instead of 32 four-byte ones: one per wave instead of four.
  static constexpr int POFF = SLAB + CALS_BN;
  static constexpr int PP = CALS_BN + 16;
  static constexpr int PLB = 16;                     // bytes per lane of one P DMA instruction
  static constexpr int PPE = 64 * PLB / ES;          // elements per P DMA instruction: 128 (one row) | 256 (two rows)
  static constexpr int P_INSTR = 16 * CALS_BN / PPE; // 16 | 8 per stage
  static constexpr int NPP = P_INSTR / DW;           // P instructions per issuing wave: 2 | 1 (DW = 8)
  static constexpr int PTILE = (ES == 8) ? 16 * PP : 16 * CALS_BN;
  static constexpr int PQ = (ES == 8) ? 4 * PP : 4 * CALS_BN;  // element offset between the rows of k-steps q, q + 1
  static constexpr int BUF = POFF + PTILE;
  static constexpr int SP = 144 / ES;                // staging tile pitch: 18 f64 | 36 f32 per column
  static constexpr int STG = 16 * SP;                // per-wave staging tile (T flush transpose)
  static constexpr int LDS_BYTES = (3 * BUF + (ES == 4 ? 8 * STG : 0)) * ES;  // staging: fp32 only
  static constexpr int NQ = (ES == 8) ? 4 : 1;       // Q values a lane needs per s (see TtmBody)
  static constexpr int N = 4 * MT;                   // MFMAs per slab per wave
  static constexpr int RING = (ES == 4) ? 8 : CALS_TTM_RING;
  static constexpr int D = N < RING ? N : RING;      // operand ring depth
  static constexpr int H = N / 2;                    // barrier position
  static constexpr int NDMA = NP + NPP + 1;          // DMA instructions per wave per stage
  static constexpr int HH = (H < N - H) ? H : N - H; // MFMA steps they are spread over
};

template <int MT, typename T>
struct TtmPipe {
  typedef TtmCfg<MT, T> C;
  typedef typename Acc<T>::type acc_t;
  // EXTRA: LDS reads issued behind the ring read that step I consumes, besides the ring's own (the rolling
  // flush reads the stage's Q values mid-stream)
  template <int I, int EXTRA = 0>
  static __device__ __forceinline__ void step(acc_t (&acc)[MT], T (&ring)[C::D], const T (&bq)[4],
                                              unsigned base, bool extra = false) {
    constexpr int outstanding = (C::D - 1 < C::N - 1 - I) ? C::D - 1 : C::N - 1 - I;
    if constexpr (EXTRA > 0) {  // only the wait differs between the two kinds of stage (wave-uniform `extra`)
      if (extra)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding + EXTRA));
      else
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
    } else {
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int q = I / MT, t = I % MT;
    // fp64: operands swapped, the tile comes out transposed (TtmBody comment)
    if constexpr (C::ES == 8)
      acc[t] = Acc<T>::mfma(bq[q], ring[I % C::D], acc[t]);
    else
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
struct TtmBody {
static __device__ __forceinline__ void run(const TtmArgs &a, const int m0, const int tm, const int nb,
                                           unsigned char *lds_raw) {
  typedef TtmCfg<MT, T> C;
  typedef TtmPipe<MT, T> P3;
  typedef typename Acc<T>::type acc_t;
  T *const lds = reinterpret_cast<T *>(lds_raw);
  const T *const Xp = static_cast<const T *>(a.Xp);
  const T *const Pt = static_cast<const T *>(a.Pt);
  const T *const Qm = static_cast<const T *>(a.Q);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int krow = lane >> 4;
  const int lcol = lane & 15;

  const long long S = a.S;
  const int nAb = a.Ap >> 4;
  // fp64 rolling flush: Q[s, :] is consumed once per s, by the flush that follows the last a-block of s: only that
  // stage's buffer gets the Q row (one DMA instruction less in the other stages) and only that stage reads it
  constexpr bool QLAST = (C::ES == 8) && (CALS_TTM_ROLL != 0) && (CALS_TTM_QLAST != 0);
  // 32-bit loop state (S is one mode's size): a 64-bit trip count made the compiler carry the stage
  // counter through VALU compares (v_cmp_le_i64, v_cndmask + v_readfirstlane) in every stage
  const int s_begin = (int)(S * tm / a.T);
  const int s_end = (int)(S * (tm + 1) / a.T);
  const int n_units = (s_end - s_begin) * nAb;


  acc_t tacc[MT], gacc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    tacc[t] = (acc_t){0, 0, 0, 0};
    gacc[t] = (acc_t){0, 0, 0, 0};
  }

  // ---- per-lane DMA source offsets (stage independent) ----
  // 32-bit BYTE offsets from wave-uniform bases: the DMA instructions take the scalar-base form
  // (global_load_lds_dwordx4 v_off, s[base:base+1]) -- no 64-bit VALU add and no address pair per piece
  const int dw = (C::DW == 8) ? wave : wave - 4;  // index among the DMA-issuing waves (negative: issues none)
  unsigned src_off[C::NP];
#pragma unroll
  for (int k = 0; k < C::NP; ++k) {
    const int piece = k * C::DW + dw;
    const int e = piece * C::PE + lane * (16 / C::ES);
    const int acol = e / C::LDL;
    const int m = e - acol * C::LDL;
    int gm = m0 + m;
    gm = gm < a.Mp ? gm : 0;  // rows past the padded tensor: any valid address, unused
    src_off[k] = (unsigned)(gm + a.Mp * acol) * (unsigned)C::ES;  // < 16 Mp elements
  }
  unsigned p_src[C::NPP];
  int p_dst[C::NPP];
#pragma unroll
  for (int k = 0; k < C::NPP; ++k) {
    const int pp = k * C::DW + dw;     // P instruction pp of the stage
    if constexpr (C::ES == 8) {        // row pp, 2 columns per lane
      p_src[k] = (unsigned)(pp * CALS_BN + lane * 2) * (unsigned)C::ES;
      p_dst[k] = C::POFF + pp * C::PP;
    } else {                           // rows 2 pp (lanes 0-31) and 2 pp + 1 (lanes 32-63, halves swapped), 4 columns per lane
      const int h = lane >> 5;
      const int col = ((lane & 31) * 4) ^ (h << 4);
      p_src[k] = (unsigned)((2 * pp + h) * CALS_BN + col) * (unsigned)C::ES;
      p_dst[k] = C::POFF + pp * 2 * CALS_BN;
    }
  }
  long long q_off;
  int q_byte;
  {
    const int d = (dw < 0 ? 0 : dw) * C::QPE + (C::ES == 8 ? (lane >> 1) : lane);
    int c = nb * CALS_BN + d;
    c = (c < a.R && d < CALS_BN) ? c : 0;
    q_off = a.ldQ * c;
    q_byte = (C::ES == 8) ? 4 * (lane & 1) : 0;
  }
  const long long slab_stride_s = (long long)a.Mp * a.Ap;
  const T *const Pt_nb = Pt + (long long)nb * a.Ap * CALS_BN;

  // qn: the unit this DMA feeds is the LAST a-block of its s -- the only stage whose Q[s, :] is ever read (ROLL)
  auto issue_piece = [&]<int K>(const T *src_slab, const T *p_slab, const T *q_row, T *dst, bool qn) {
    if constexpr (K < C::NP) {
      const int piece = K * C::DW + dw;
      if (piece < C::PIECES && !DIAG(a.dbg & 16) && !STRIP(16))  // CALS_DIAG, dbg 16 (timing only): no X DMA
        __builtin_amdgcn_global_load_lds(
            (const GLOBAL_AS void *)(reinterpret_cast<const char *>(src_slab) + (unsigned long long)src_off[K]),
            (LDS_AS void *)(dst + piece * C::PE), 16, 0, 0);
    } else if constexpr (K < C::NP + C::NPP) {
      constexpr int k = K - C::NP;
      if (!DIAG(a.dbg & 8) && !STRIP(8))  // CALS_DIAG, dbg 8 (timing only): no P DMA
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void *)(reinterpret_cast<const char *>(p_slab) + (unsigned long long)p_src[k]),
          (LDS_AS void *)(dst + p_dst[k]), C::PLB, 0, 0);
    } else if (dw < C::QPIECES && qn) {
      const char *src = (const char *)(q_row + q_off) + q_byte;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src,
                                       (LDS_AS void *)(dst + C::QOFF + dw * C::QPE), 4, 0, 0);
    }
  };
  // DMA instructions of MFMA step IS of the issuing half: K = IS, IS + HH, IS + 2 HH, ...
  auto issue_at = [&]<int IS>(const T *src_slab, const T *p_slab, const T *q_row, T *dst, bool qn) {
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
      (
          [&] {
            if constexpr (Ks % C::HH == IS) issue_piece.template operator()<Ks>(src_slab, p_slab, q_row, dst, qn);
          }(),
          ...);
    }(std::make_integer_sequence<int, C::NDMA>{});
  };
  auto issue_all = [&](int ab, int s, int bufi) {
    const T *src_slab = Xp + (long long)a.Mp * (16 * ab) + slab_stride_s * s;
    const T *p_slab = Pt_nb + (long long)(16 * ab) * CALS_BN;
    const T *q_row = Qm + s;
    T *dst = lds + bufi * C::BUF;
    const bool qn = !QLAST || ab == nAb - 1;
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
      (issue_piece.template operator()<Ks>(src_slab, p_slab, q_row, dst, qn), ...);
    }(std::make_integer_sequence<int, C::NDMA>{});
  };

  int ab_c = 0;
  int s_c = s_begin;
  const bool dma_wave = dw >= 0;
  if (n_units > 0 && dma_wave) issue_all(ab_c, s_c, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int ab_1 = ab_c + 1;
  int s_1 = s_c;
  if (ab_1 >= nAb) { ab_1 = 0; s_1++; }
  if (n_units > 1 && dma_wave) issue_all(ab_1, s_1, 1);
  int ab_2 = ab_1 + 1;
  int s_2 = s_1;
  if (ab_2 >= nAb) { ab_2 = 0; s_2++; }

  const unsigned lane_off = (unsigned)((krow * C::LDL + lcol) * C::ES);
  const unsigned lds0 = (unsigned)(size_t)((LDS_AS const char *)lds_raw);
  // Accumulator layouts.  fp32: acc[t][r] = T[m = 16t + 4 krow + r][c = lcol] (X as the MFMA's A
  // operand, P as B).  fp64: the operands are SWAPPED (P as A, X as B; both operand layouts hold
  // element (k = lane>>4, index = lane&15), so the registers are the same) and the tile comes out
  // transposed: acc[t][r] = T[m = 16t + lcol][c = krow + 4r].  For a fixed register the 16 lanes of
  // a krow group then hold 16 consecutive m of one column = one whole 128-byte line, so the T flush
  // and the partial-tile epilogue store full lines straight from registers.  (Stores from the
  // untransposed fp64 layout cover a quarter line per column and make the L2 fetch every line
  // before merging it: rocprofv3 FETCH_SIZE showed 2.0 GB of fills per launch at C3.)
  const unsigned q_lane_off =
      (unsigned)((C::QOFF + wave * 16 + (C::ES == 8 ? krow : lcol)) * C::ES);
  const unsigned p_lane_off =
      (C::ES == 8) ? (unsigned)((C::POFF + krow * C::PP + wave * 16 + lcol) * C::ES)
                   : (unsigned)((C::POFF + (krow >> 1) * 2 * CALS_BN + (krow & 1) * CALS_BN +
                                 ((wave * 16 + lcol) ^ ((krow & 1) << 4))) * C::ES);
  T *const Tout = static_cast<T *>(a.Tout);
  const bool st = !DIAG(a.dbg & 1) && !STRIP(1);  // CALS_DIAG, dbg 1 (timing only): no T stores

  // Columns R .. NB * 128 - 1 of the last column block are stored too (zeros: P is zero padded there;
  // the engine sizes T for whole column blocks): a per-lane column predicate costs every one of the
  // 4 MT stores of a flush a saveexec / branch / restore.
  // fp64: per register r the column krow + 4r of this wave
  T *tc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gc = nb * CALS_BN + wave * 16 + krow + 4 * r;
    tc[r] = Tout + ((long long)gc * S) * a.Mp + m0 + lcol;
  }
  // fp64 rolling flush: the same addresses as a wave-uniform base (SGPR pair, carries s) + a 32-bit per-lane
  // byte offset per register r (ttm_launch checks 15 * S * Mp * 8 + 8 * Mp < 2^32)
  constexpr bool ROLL = (C::ES == 8) && (CALS_TTM_ROLL != 0);
  unsigned tvo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    tvo[r] = STRIP(2048) ? (unsigned)(((krow + 4 * r) * 16 + lcol) * C::ES)
                         : (unsigned)((((long long)(krow + 4 * r) * S) * a.Mp + lcol) * C::ES);
  const char *const t_wave = reinterpret_cast<const char *>(Tout) +
                             (((long long)(nb * CALS_BN + wave * 16) * S) * a.Mp + m0) * C::ES;
  // fp32: T flush through a per-wave LDS staging tile (two tiles = 32 rows = one 128-byte line per
  // column), so that every global_store_dwordx4 writes 8 whole lines
  T *const stg = lds + 3 * C::BUF + wave * C::STG;
  const int j8 = lane & 7;
  T *tb[2];
  bool tv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c8 = (lane >> 3) + 8 * h;
    const int gc = nb * CALS_BN + wave * 16 + c8;
    tv[h] = st;
    tb[h] = STRIP(2048) ? Tout + ((blockIdx.x & 255) * 8 + wave) * 4096 + c8 * 64 + 4 * j8  // (timing only: an L2-resident window)
                        : Tout + ((long long)gc * S) * a.Mp + m0 + 4 * j8;
  }
  // fp32, odd MT: the last tile alone (16 rows = 64 B per column, 4 lanes per column)
  const int gc4 = nb * CALS_BN + wave * 16 + (lane >> 2);
  const bool tv4 = st;
  T *const tb4 = Tout + ((long long)gc4 * S) * a.Mp + m0 + 4 * (lane & 3);

  // end of an s: G += T * Q[s, c]; T -> HBM (non-temporal: written once, read once by the
  // contraction); T = 0
  auto flush = [&](int s, const T (&qv)[C::NQ]) {
    asm volatile("" ::: "memory");
    const long long so = STRIP(2048) ? 0 : (long long)s * a.Mp;
    if constexpr (C::ES == 8) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gacc[t][r] += tacc[t][r] * qv[r];
          if (st) __builtin_nontemporal_store(tacc[t][r], tc[r] + so + 16 * t);
        }
        tacc[t] = (acc_t){0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int t = 0; t + 1 < MT; t += 2) {
        if constexpr (STRIP(4096)) {  // timing only: the same 16-byte stores straight from the registers (wrong places), no LDS round trip
#pragma unroll
          for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gacc[t + k][r] += tacc[t + k][r] * qv[0];
            if (tv[k]) __builtin_nontemporal_store(tacc[t + k], reinterpret_cast<acc_t *>(tb[k] + so + 16 * t));
            tacc[t + k] = (acc_t){0, 0, 0, 0};
          }
          continue;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
          for (int r = 0; r < 4; ++r) gacc[t + k][r] += tacc[t + k][r] * qv[0];
          *reinterpret_cast<acc_t *>(stg + lcol * C::SP + 16 * k + 4 * krow) = tacc[t + k];
          tacc[t + k] = (acc_t){0, 0, 0, 0};
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const acc_t v = *reinterpret_cast<const acc_t *>(stg + ((lane >> 3) + 8 * h) * C::SP + 4 * j8);
          if (tv[h] && !(STRIP(8192) && t >= MT / 2)) {
            if constexpr (STRIP(16384))  // timing only: the same store instructions carrying a quarter of the bytes
              __builtin_nontemporal_store(v[0], tb[h] + so + 16 * t);
            else
              __builtin_nontemporal_store(v, reinterpret_cast<acc_t *>(tb[h] + so + 16 * t));
          }
        }
      }
      if constexpr (MT % 2 == 1) {
        constexpr int t = MT - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) gacc[t][r] += tacc[t][r] * qv[0];
        *reinterpret_cast<acc_t *>(stg + lcol * C::SP + 4 * krow) = tacc[t];
        tacc[t] = (acc_t){0, 0, 0, 0};
        const acc_t v = *reinterpret_cast<const acc_t *>(stg + (lane >> 2) * C::SP + 4 * (lane & 3));
        if (tv4) __builtin_nontemporal_store(v, reinterpret_cast<acc_t *>(tb4 + so + 16 * t));
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // CALS_DIAG builds: per-stage shader-clock sums (DMA wait, barrier wait, period) over the steady-
  // state stages of the first M block, for waves 0 and 4 of 8 workgroups (tools/ttm_trace.py)
  unsigned long long *trace = nullptr;
  if (DIAG(a.dbg_trace) && m0 == 0 && lane == 0 && (wave & 3) == 0 && (blockIdx.x & 31) == 0 &&
      blockIdx.x < 256 && n_units <= 512)
    trace = a.dbg_trace + ((blockIdx.x >> 5) * 2 + (wave >> 2)) * 2048;

  const bool skip_q3 = a.A - 16 * (nAb - 1) <= 12;  // the last a-block has no valid row in k-step 3
  int buf = 0;
  auto unit_loop = [&]<bool LATE>() {
    unsigned long long dg_vm = 0, dg_bar = 0, dg_per = 0, dg_n = 0, dg_last = 0;  // CALS_DIAG sums
    // fp32 (round 4): the T stores of a flush cost the kernel 8.5 % (stripped build) -- not their bandwidth (0.8 TB/s) but
    // the vmcnt(0) in front of the next barrier, which made the wave wait for their acknowledgement.  Every wave now
    // flushes at the END of the stage (behind the DMA pieces the next barrier is about) and lets that wait cover
    // everything BUT the NST store instructions, the youngest ones; they have a whole further stage to complete.
    constexpr bool SB = (C::ES == 4) && (CALS_TTM_F32_STORES_BEHIND != 0) && !ROLL;
    constexpr int NST = (MT / 2) * 2 + (MT % 2);  // store instructions of one fp32 flush
    static_assert(NST <= 63, "vmcnt is a 6-bit counter");
    bool flushed = false;
    bool pend = false;       // LATE: flush of the previous unit deferred behind this unit's barrier
    int s_pend = 0;
    T q_pend[C::NQ] = {};
    // DMA sources of the unit two stages ahead and the LDS addresses of the next stage are carried
    // across iterations and advanced INSIDE the MFMA stream (right behind the last DMA issue), by
    // additions: between a wave's last MFMA of a stage and the barrier nothing but the DMA wait is
    // left.  (While waves 4-7 walk from their last MFMA to the barrier, waves 0-3 are already parked
    // at it: that tail is dead time for the SIMD.)
    const T *src_slab = Xp + (long long)a.Mp * (16 * ab_2) + slab_stride_s * s_2;
    const T *p_slab = Pt_nb + (long long)(16 * ab_2) * CALS_BN;
    const T *q_row = Qm + s_2;
    const long long src_wrap = slab_stride_s - (long long)a.Mp * 16 * nAb;  // + one a-block step = next s
    const T *const safe_x = Xp + slab_stride_s * s_begin;
    const T *const safe_q = Qm + s_begin;
    unsigned bufb = lds0 + (unsigned)(buf * C::BUF * C::ES), base = bufb + lane_off;
    unsigned bufb_n = bufb, base_n = base;
    int buf_nn = (buf + 2) % 3;
    auto advance = [&]() {  // called once per stage, mid-stream
      ab_2 = ab_2 + 1;
      src_slab += (long long)a.Mp * 16;
      p_slab += 16 * CALS_BN;
      if (ab_2 >= nAb) {
        ab_2 = 0;
        s_2++;
        src_slab += src_wrap;
        p_slab = Pt_nb;
        q_row += 1;
      }
      const int b1 = (buf == 2) ? 0 : buf + 1;
      bufb_n = lds0 + (unsigned)(b1 * C::BUF * C::ES);
      base_n = bufb_n + lane_off;
    };
    for (int iu = 0; iu < n_units; ++iu) {
      // The DMA of unit iu + 2 is issued unconditionally: in the last two stages of the range it
      // re-reads the range's first slab into the buffer nobody consumes any more (waited for at the end
      // of run()), instead of putting a uniform branch in front of every DMA instruction of every stage.
      // (fp64 only: the fp32 kernel at MT 19 measured 4 % slower with it.)
      constexpr bool ALWAYS = (C::ES == 8);
      const bool fetch = iu + 2 < n_units;
      const T *const dma_x = (fetch || !ALWAYS) ? src_slab : safe_x;
      const T *const dma_p = (fetch || !ALWAYS) ? p_slab : Pt_nb;
      const T *const dma_q = (fetch || !ALWAYS) ? q_row : safe_q;
      T *dst = lds + buf_nn * C::BUF;
      const bool dma_qn = !QLAST || ab_2 == nAb - 1;   // (ab_2 still names unit iu + 2: advance() runs behind the issue)
      const bool qstage = !QLAST || ab_c == nAb - 1;    // this stage reads Q

      if constexpr (LATE) {
        // barrier #iu at the start of slab iu: DMA(iu+1) landed everywhere, slab iu-1 is finished by
        // everybody, so DMA(iu+2) may overwrite its buffer.  The T stores of a flush are issued
        // BEHIND this wait, so they never sit in front of it (vmcnt counts stores too).
        unsigned long long d0 = 0, d1 = 0, d2 = 0;
        if (DIAG(trace)) d0 = __builtin_amdgcn_s_memtime();
        if (!STRIP(512)) {
          if (SB && flushed)  // the previous stage ended with a flush: its NST store instructions are the youngest
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST) : "memory");  // vector-memory operations of this wave
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          flushed = false;
        }
        if (DIAG(trace)) d1 = __builtin_amdgcn_s_memtime();
        if (!STRIP(128)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG(trace)) {
          d2 = __builtin_amdgcn_s_memtime();
          if (iu >= 3 && ab_c >= 2) {  // steady-state stages only (not the two behind a T flush)
            dg_vm += d1 - d0;
            dg_bar += d2 - d1;
            dg_per += d2 - dg_last;
            dg_n++;
          }
          dg_last = d2;
        }
        if constexpr (!ROLL) {
          if (pend) {
            flush(s_pend, q_pend);
            pend = false;
          }
        }
      }
      T ring[C::D];
      T qcur[C::NQ];
      T bq[4];
      if constexpr (!ROLL) {
        lds_read_off<T, 0>(qcur[0], bufb + q_lane_off);
        if constexpr (C::NQ == 4) {  // fp64: Q[s, krow + 4r]
          lds_read_off<T, 4 * C::ES>(qcur[1], bufb + q_lane_off);
          lds_read_off<T, 8 * C::ES>(qcur[2], bufb + q_lane_off);
          lds_read_off<T, 12 * C::ES>(qcur[3], bufb + q_lane_off);
        }
      }
      if (!STRIP(256) || iu == 0) {
        lds_read_off<T, 0>(bq[0], bufb + p_lane_off);
        lds_read_off<T, C::PQ * C::ES>(bq[1], bufb + p_lane_off);
        lds_read_off<T, 2 * C::PQ * C::ES>(bq[2], bufb + p_lane_off);
        lds_read_off<T, 3 * C::PQ * C::ES>(bq[3], bufb + p_lane_off);
      }
      P3::template preload<0>(ring, base);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(C::D));  // Q and P landed (D younger reads in flight)
      __builtin_amdgcn_sched_barrier(0);

      // ROLL: `pend` = this is the first stage of a new s and the finished tiles of the previous s are still
      // in tacc.  Tile t is flushed (G += T * Q[s_prev], T -> HBM, T = 0) right in front of the k-step-0 MFMA
      // that starts it anew: the 4 FMAs + 4 stores + 4 moves of a tile sit between two MFMAs instead of 120
      // instructions in a row behind the stage.  q_pend is the ONE set of Q registers:
      // every stage reads its Q[s, :] into it right behind k-step 0 -- i.e. behind the last use of the previous
      // s's values -- so during k-step 0 of a new s it still holds Q[s - 1, :].
      constexpr int QJ = MT - 1;  // the step behind whose ring read the Q reads are issued
      const bool roll = ROLL && pend;
      // STRIP 2048 (timing only): every T store of the rolling flush goes to a 16-KiB window per wave that stays in L2 --
      // the same instructions without the HBM write stream
      const char *const t_s = STRIP(2048) ? reinterpret_cast<const char *>(Tout) + ((blockIdx.x & 255) * 8 + wave) * 16384
                                          : t_wave + ((long long)s_pend * a.Mp) * C::ES;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (
            [&] {
              if constexpr (ROLL && Is < MT) {
                if (roll) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    if (!STRIP(32)) gacc[Is][r] += tacc[Is][r] * q_pend[r];
                    if (st && !(STRIP(8192) && Is >= MT / 2)) {
                      if constexpr (STRIP(16384))  // timing only: the same store instructions carrying half the bytes
                        asm volatile("global_store_dword %0, %1, %2 offset:%3 nt"
                                     :: "v"(tvo[r]), "v"((float)tacc[Is][r]), "s"(t_s), "i"(16 * Is * C::ES) : "memory");
                      else
                        asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt"
                                     :: "v"(tvo[r]), "v"(tacc[Is][r]), "s"(t_s), "i"(16 * Is * C::ES) : "memory");
                    }
                  }
                  // (the MFMA's inline constant C = 0 would save these four moves, but as inline asm with the tile
                  // tied in place the allocator copies tiles through a second register set: 130 spills at MT 10)
                  tacc[Is] = (acc_t){0, 0, 0, 0};
                }
                P3::template step<Is>(tacc, ring, bq, base);
                if constexpr (Is == QJ && !STRIP(64)) if (qstage) {
                  lds_read_off<T, 0>(q_pend[0], bufb + q_lane_off);
                  lds_read_off<T, 4 * C::ES>(q_pend[1], bufb + q_lane_off);
                  lds_read_off<T, 8 * C::ES>(q_pend[2], bufb + q_lane_off);
                  lds_read_off<T, 12 * C::ES>(q_pend[3], bufb + q_lane_off);
                }
              } else if constexpr (ROLL && Is > QJ && Is <= QJ + C::D) {
                // in a Q stage four Q reads were issued behind the ring read this step consumes
                P3::template step<Is, 4>(tacc, ring, bq, base, qstage);
              } else {
                P3::template step<Is, 0>(tacc, ring, bq, base);
              }
              if constexpr (LATE && Is < C::HH) {
                if (ALWAYS || fetch) issue_at.template operator()<Is>(dma_x, dma_p, dma_q, dst, dma_qn);
              }
              if constexpr (LATE && Is == (C::NDMA < C::H - 1 ? C::NDMA : C::H - 1)) advance();
            }(),
            ...);
      }(std::make_integer_sequence<int, C::H>{});

      if constexpr (!LATE) {
        unsigned long long d0 = 0, d1 = 0, d2 = 0;
        if (DIAG(trace)) d0 = __builtin_amdgcn_s_memtime();
#if CALS_TTM_EARLYVM
        // A stage that flushed T (roll) issued its 4 MT stores in its first MT steps, BEHIND this wave's DMA pieces of
        // unit iu + 1 (second half of the previous stage): the in-order counter lets the wait cover the DMA only --
        // the stores, a microsecond old at most, need not be acknowledged before the barrier
        if (!STRIP(512) && C::DW == 8) {
          if (roll && 4 * MT <= 63)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(4 * MT <= 63 ? 4 * MT : 0) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#else
        if (!STRIP(512) && C::DW == 8) {  // (no DMA of its own: no wait)
          if (SB && flushed)
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          flushed = false;
        }
#endif
        if (DIAG(trace)) d1 = __builtin_amdgcn_s_memtime();
        if (!STRIP(128)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG(trace)) {
          d2 = __builtin_amdgcn_s_memtime();
          if (iu >= 3 && ab_c >= 2) {
            dg_vm += d1 - d0;
            dg_bar += d2 - d1;
            dg_per += d2 - dg_last;
            dg_n++;
          }
          dg_last = d2;
        }
      }

      auto second_half = [&]<int I0, int... Is>(std::integer_sequence<int, Is...>) {
        (
            [&] {
              constexpr int J = I0 + Is;  // step H + J
              if constexpr (ROLL && C::H + J <= QJ + C::D) {
                P3::template step<C::H + J, 4>(tacc, ring, bq, base, qstage);
              } else {
                P3::template step<C::H + J, 0>(tacc, ring, bq, base);
              }
              if constexpr (!LATE && J < C::HH && C::DW == 8) {
                if (ALWAYS || fetch) issue_at.template operator()<J>(dma_x, dma_p, dma_q, dst, dma_qn);
              }
              if constexpr (!LATE && J == (C::DW != 8 ? 0 : (C::NDMA < C::N - C::H - 1 ? C::NDMA : C::N - C::H - 1)))
                advance();
            }(),
            ...);
      };
      // k-step 2 (rows 8..11 of the a-block), then k-step 3 (rows 12..15) -- which is all padding in
      // the last a-block of a mode whose size leaves at most 12 rows there (300 = 18 x 16 + 12): its
      // MFMAs would multiply zeros.  (Only when the DMA issue fits k-step 2, i.e. NDMA < MT.)
      second_half.template operator()<0>(std::make_integer_sequence<int, MT>{});
      if (!((C::DW != 8 || C::NDMA < MT) && skip_q3 && ab_c == nAb - 1))
        second_half.template operator()<MT>(std::make_integer_sequence<int, C::N - C::H - MT>{});
      else  // the operand reads already issued for k-step 3 must land before their registers are reused
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

      if constexpr (ROLL) pend = false;
      if (ab_c == nAb - 1) {
        if constexpr (ROLL || (LATE && !SB)) {
          pend = true;
          s_pend = s_c;
          if constexpr (!ROLL) {
#pragma unroll
            for (int r = 0; r < C::NQ; ++r) q_pend[r] = qcur[r];
          }
        } else {
          flush(s_c, qcur);
          flushed = true;
        }
      }
      buf = (buf == 2) ? 0 : buf + 1;
      buf_nn = (buf_nn == 2) ? 0 : buf_nn + 1;
      bufb = bufb_n;
      base = base_n;
      ab_c = ab_c + 1;
      if (ab_c >= nAb) { ab_c = 0; s_c++; }
    }
    if constexpr (ROLL) {  // the last s of the range: nothing overwrites its tiles, flush them here
      if (pend) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const char *const t_s = t_wave + ((long long)s_pend * a.Mp) * C::ES;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gacc[t][r] += tacc[t][r] * q_pend[r];
            if (st)
              asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 nt"
                           :: "v"(tvo[r]), "v"(tacc[t][r]), "s"(t_s), "i"(16 * t * C::ES) : "memory");
          }
        }
      }
    } else if constexpr (LATE) {
      if (pend) flush(s_pend, q_pend);
    }
    if (DIAG(trace)) {  // stamps stay in registers inside the loop (a global store per stage would sit
      trace[0] = dg_vm;  // in front of the vmcnt wait it is meant to measure)
      trace[1] = dg_bar;
      trace[2] = dg_per;
      trace[3] = dg_n;
    }
  };
#if defined(CALS_TTM_PRIO)  // experiment: static issue priority for one wave group (1: waves 4-7, 2: waves 0-3)
  if ((CALS_TTM_PRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
  if (wave < 4 && !DIAG(a.dbg & 2))
    unit_loop.template operator()<false>();
  else
    unit_loop.template operator()<true>();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: G_first partial tile [ldPart x 128] of (nb, tm)
  T *pt = static_cast<T *>(a.partial) + ((long long)(nb * a.T + tm)) * ((long long)a.ldPart * CALS_BN);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (C::ES == 8) {  // transposed tile: 16 lanes = one line of column krow + 4r
        pt[m0 + 16 * t + lcol + (long long)a.ldPart * (wave * 16 + krow + 4 * r)] = gacc[t][r];
      } else {
        const int m = m0 + 16 * t + Acc<T>::row(krow, r);
        pt[m + (long long)a.ldPart * (wave * 16 + lcol)] = gacc[t][r];
      }
    }
  }
}
};

template <int MT, typename T>
__global__ void __launch_bounds__(512, 2) ttm_kernel(const TtmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // Workgroup -> (column block nb, team member tm, M block by).  XCD-aware bijective remap first
  // (workgroups with equal blockIdx % 8 share an XCD and get consecutive p), then p enumerates
  // groups of `nbw` column blocks; inside a group all team members tm, column block fastest.  One
  // XCD's ~grid/8 consecutive p therefore touch few distinct P panels (nbw * Ap * 128 elements fit
  // its 4 MB L2 and are re-read once per s) and few distinct tm, i.e. X slabs, each shared by the
  // nbw workgroups that stream it at the same time.
  const int G = a.grid;
  const int b = blockIdx.x;
  const int xcd = b & 7, q8 = G >> 3, r8 = G & 7;
  const int p = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int g = p / (a.nbw * a.T);
  const int wg = min(a.nbw, a.NB - g * a.nbw);
  const int rem = p - g * a.nbw * a.T;
  const int tm = rem / wg;
  const int nb = g * a.nbw + (rem - tm * wg);
  // One workgroup walks ALL M blocks of its (column block, s range), one after the other: blocks
  // of MT and MT - 1 tiles then cost every workgroup the same (an M block per workgroup left the
  // CUs that drew an MT - 1 block idle for 1/MT of the kernel).
  for (int by = 0; by < a.m_blocks; ++by) {
    if (by > 0) __syncthreads();  // every wave has left the LDS ring of the previous block
    if constexpr (MT > 1) {
      if (by >= a.k_big) {
        TtmBody<MT - 1, T>::run(a, 16 * (a.k_big * MT + (by - a.k_big) * (MT - 1)), tm, nb, lds_raw);
        continue;
      }
    }
    TtmBody<MT, T>::run(a, 16 * by * MT, tm, nb, lds_raw);
  }
}

template <int MT, typename T>
static hipError_t ttm_launch_mt(const TtmArgs &a, hipStream_t st) {
  static AttrOnce attr_once;
  constexpr int lds_bytes = TtmCfg<MT, T>::LDS_BYTES;
  const hipError_t e = attr_once.ensure([] {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&ttm_kernel<MT, T>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  });
  if (e != hipSuccess) return e;
  dim3 grid(a.grid, 1, 1), block(512, 1, 1);
  hipLaunchKernelGGL((ttm_kernel<MT, T>), grid, block, lds_bytes, st, a);
  return hipGetLastError();
}

int ttm_max_mt(int dtype) { return dtype == CALS_F32 ? 20 : 10; }

// fp64: the rolling T flush addresses a wave's 16 columns of T by a 32-bit per-lane byte offset from a
// wave-uniform base: (15 S + 1) Mp elements must stay below 2^32 bytes (the engine plans no pair beyond that)
bool ttm_shape_ok(long long S, long long Mp, int dtype) {
  if (dtype == CALS_F32) return true;
  return (15 * S + 1) * Mp * 8 < (1ll << 32);
}

hipError_t ttm_launch(const TtmArgs &a, hipStream_t st) {
  if (a.MT < 1 || a.MT > ttm_max_mt(a.dtype)) return hipErrorInvalidValue;
  if (a.S >= (1ll << 31) / (a.Ap >> 4 ? a.Ap >> 4 : 1)) return hipErrorInvalidValue;  // 32-bit stage counters
  if (!ttm_shape_ok(a.S, a.Mp, a.dtype)) return hipErrorInvalidValue;
  if (a.dtype == CALS_F32) {
    switch (a.MT) {
#define CASE(N) case N: return ttm_launch_mt<N, float>(a, st);
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11)
      CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
    }
    return hipErrorInvalidValue;
  }
  switch (a.MT) {
#define CASE(N) case N: return ttm_launch_mt<N, double>(a, st);
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Pt[nb][a][128] = P[a, 128 nb + cc] (zero for a >= A or column >= R): the B-operand tiles of the
// TTM as contiguous 16 x 128 blocks (one 1-KiB LDS-DMA piece per row in fp64)
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) pack_pt_kernel(const E *P, long long ldP, int A, int Ap,
                                                      int R, E *Pt) {
  __shared__ E tile[32][33];
  const int a0 = blockIdx.x * 32, c0 = blockIdx.y * 32;  // 32 x 32 transpose tiles
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int a = a0 + tx, c = c0 + j;
    tile[j][tx] = (a < A && c < R) ? P[a + ldP * c] : (E)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int a = a0 + j, c = c0 + tx;
    if (a < Ap) Pt[((long long)(c >> 7) * Ap + a) * CALS_BN + (c & (CALS_BN - 1))] = tile[tx][j];
  }
}

hipError_t pack_pt_launch(const void *P, long long ldP, int A, int Ap, int NB, int R, void *Pt,
                          int dtype, hipStream_t st) {
  const dim3 grid((Ap + 31) / 32, NB * (CALS_BN / 32)), block(256);
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(pack_pt_kernel<float>, grid, block, 0, st, (const float *)P, ldP, A, Ap, R,
                       (float *)Pt);
  else
    hipLaunchKernelGGL(pack_pt_kernel<double>, grid, block, 0, st, (const double *)P, ldP, A, Ap,
                       R, (double *)Pt);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// G_second[s, c] = sum_m T[c][s][m] * F[m, c]: one workgroup per column, one wave per s, fixed
// summation tree (lane-strided partial sums in fp64, then a butterfly) => deterministic.
// ---------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) contract_kernel(const E *Tb, long long S, int Mp, int M,
                                                       const E *F, long long ldF, E *out,
                                                       long long ldOut) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NI = 8;  // register-cached rows of F[:, c] per lane: M <= 512 without re-reads
  double f[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int m = lane + 64 * i;
    f[i] = (m < M) ? (double)F[m + ldF * c] : 0.0;
  }
  const E *Tc = Tb + (long long)c * S * Mp;
  for (long long s = blockIdx.y * 4 + wave; s < S; s += 4 * gridDim.y) {
    const E *row = Tc + s * Mp;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int m = lane + 64 * i;
      if (m < M) acc += (double)row[m] * f[i];
    }
    for (int m = lane + 64 * NI; m < M; m += 64) acc += (double)row[m] * (double)F[m + ldF * c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[s + ldOut * c] = (E)acc;
  }
}

// The same with 16-byte loads and four s rows in flight per wave (more bytes outstanding per CU);
// used when a row fits NI passes of 64 lanes x 16 B.  Summation order per (s, c): lane-strided
// partial sums in fp64, then the butterfly -- fixed, so results are deterministic.
template <typename E, int NI>
__global__ void __launch_bounds__(256) contract4_kernel(const E *Tb, long long S, int Mp, int M,
                                                        const E *F, long long ldF, E *out,
                                                        long long ldOut) {
  constexpr int VE = 16 / (int)sizeof(E);
  typedef E vec __attribute__((ext_vector_type(VE)));
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double f[NI][VE];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int v = 0; v < VE; ++v) {
      const int m = (lane + 64 * i) * VE + v;
      f[i][v] = (m < M) ? (double)F[m + ldF * c] : 0.0;
    }
  const E *Tc = Tb + (long long)c * S * Mp;
  for (long long s0 = 4 * (blockIdx.y * 4 + wave); s0 < S; s0 += 16 * gridDim.y) {
    vec x[4][NI];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long s = (s0 + k < S) ? s0 + k : S - 1;  // clamped rows are computed and dropped
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m0 = (lane + 64 * i) * VE;
        if (m0 < Mp)
          x[k][i] = __builtin_nontemporal_load(reinterpret_cast<const vec *>(Tc + s * Mp + m0));
        else
          x[k][i] = (vec)(E)0;
      }
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int v = 0; v < VE; ++v) acc[k] += (double)x[k][i][v] * f[i][v];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
    if (lane < 4 && s0 + lane < S) {
      const double r = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      out[s0 + lane + ldOut * c] = (E)r;
    }
  }
}

template <typename E>
static bool contract4_try(const E *Tb, long long S, int Mp, int M, const E *F, long long ldF, E *out,
                          long long ldOut, int R, hipStream_t st) {
  constexpr int VE = 16 / (int)sizeof(E);
  const int passes = (Mp + 64 * VE - 1) / (64 * VE);
  if (passes > 4) return false;
  int gy = (int)((S / 4 + 15) / 16);
  if (gy < 1) gy = 1;
  // small problems: as many workgroups as there are row quads (one pass per wave) while the grid stays below four rounds of 8 waves
  // per SIMD of the whole chip -- C2 (640 columns, 100 rows): 2 -> 7 workgroups per column; C3 / C4 are unchanged.
  // The summation of every (s, c) is the same wave-local tree: nothing changes in the results.
  const int gy_max = (int)((S + 15) / 16);
  while (gy < gy_max && (long long)R * (gy + 1) * 4 <= 4 * 8192) gy++;
  const dim3 grid(R, gy), block(256);
  switch (passes) {
    case 1: hipLaunchKernelGGL((contract4_kernel<E, 1>), grid, block, 0, st, Tb, S, Mp, M, F, ldF, out, ldOut); break;
    case 2: hipLaunchKernelGGL((contract4_kernel<E, 2>), grid, block, 0, st, Tb, S, Mp, M, F, ldF, out, ldOut); break;
    case 3: hipLaunchKernelGGL((contract4_kernel<E, 3>), grid, block, 0, st, Tb, S, Mp, M, F, ldF, out, ldOut); break;
    default: hipLaunchKernelGGL((contract4_kernel<E, 4>), grid, block, 0, st, Tb, S, Mp, M, F, ldF, out, ldOut); break;
  }
  return true;
}

hipError_t contract_launch(const void *Tb, long long S, int Mp, int M, const void *F,
                           long long ldF, void *out, long long ldOut, int R, int dtype,
                           hipStream_t st) {
  if (R <= 0) return hipSuccess;
  if (dtype == CALS_F32) {
    if (contract4_try<float>((const float *)Tb, S, Mp, M, (const float *)F, ldF, (float *)out, ldOut, R, st))
      return hipGetLastError();
  } else if (contract4_try<double>((const double *)Tb, S, Mp, M, (const double *)F, ldF,
                                   (double *)out, ldOut, R, st)) {
    return hipGetLastError();
  }
  int gy = (int)((S + 63) / 64);  // >= 16 s values per wave
  if (gy < 1) gy = 1;
  const dim3 grid(R, gy), block(256);
  if (dtype == CALS_F32)
    hipLaunchKernelGGL(contract_kernel<float>, grid, block, 0, st, (const float *)Tb, S, Mp, M,
                       (const float *)F, ldF, (float *)out, ldOut);
  else
    hipLaunchKernelGGL(contract_kernel<double>, grid, block, 0, st, (const double *)Tb, S, Mp, M,
                       (const double *)F, ldF, (double *)out, ldOut);
  return hipGetLastError();
}

}  // namespace calship
