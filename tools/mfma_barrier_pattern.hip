// Microbenchmark: the MFMA loop of mttkrp3/ttm kernels with its per-stage barrier, no DMA, to see what
// the barrier placement alone costs at 2 waves per SIMD (8 waves per workgroup, one workgroup per CU).
//   MODE 0: no barrier            MODE 1: all waves barrier at stage start
//   MODE 2: waves 0-3 barrier mid-stage, waves 4-7 at stage start (the kernels' stagger)
//   MODE 3: as 2, s_setprio 3 around the second wave group     MODE 4: as 2 with s_sleep-free spin? (unused)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -w mfma_barrier_pattern.hip -o mfma_barrier_pattern
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

template <int MT, int LDL, int D, int I0, int I1>
struct Pipe {
  static __device__ __forceinline__ void run(v4d (&acc)[MT], double (&ring)[D], const double (&bq)[4], unsigned base) {
    if constexpr (I0 < I1) {
      constexpr int N = 4 * MT;
      constexpr int outstanding = (D - 1 < N - 1 - I0) ? D - 1 : N - 1 - I0;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(outstanding));
      __builtin_amdgcn_sched_barrier(0);
      constexpr int q = I0 / MT, t = I0 % MT;
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[I0 % D], bq[q], acc[t], 0, 0, 0);
      if constexpr (I0 + D < N) {
        constexpr int qn = (I0 + D) / MT, tn = (I0 + D) % MT;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ring[I0 % D]) : "v"(base), "i"(((4 * qn) * LDL + 16 * tn) * 8));
      }
      Pipe<MT, LDL, D, I0 + 1, I1>::run(acc, ring, bq, base);
    }
  }
  static __device__ __forceinline__ void preload(double (&ring)[D], unsigned base) {
    if constexpr (I0 < I1) {
      constexpr int q = I0 / MT, t = I0 % MT;
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ring[I0 % D]) : "v"(base), "i"(((4 * q) * LDL + 16 * t) * 8));
      Pipe<MT, LDL, D, I0 + 1, I1>::preload(ring, base);
    }
  }
};

#define GLOBAL_AS __attribute__((address_space(1)))
template <int MT, int MODE, int OPT>
__global__ void __launch_bounds__(512, 2) k(double *out, unsigned long long *clk, int iters, const double *src) {
  constexpr int LDL = (MT % 2) ? 16 * MT : 16 * MT + 16, D = 6, N = 4 * MT, H = N / 2;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int i = threadIdx.x; i < 16 * LDL; i += blockDim.x) lds[i] = 1.0 + (i & 7) * 1e-3;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v4d acc[MT];
  for (int t = 0; t < MT; t++) acc[t] = (v4d){0, 0, 0, 0};
  const unsigned base = (unsigned)(size_t)((LDS_AS const char *)(lds + (lane >> 4) * LDL + (lane & 15)));
  double bq[4] = {1.0 + lane * 1e-9, 1.1, 1.2, 1.3};
  v4d gacc[(OPT & 4) ? MT : 1];
  for (auto &g : gacc) g = (v4d){0, 0, 0, 0};
  const unsigned bqaddr = (unsigned)(size_t)((LDS_AS const char *)(lds + (lane >> 4) * 144 + (lane & 15) + wave * 16));
  // stage prologue shared by both loop shapes
  auto stage_start = [&](double (&ring)[D], int it) {
    if constexpr (OPT & 1) {
      double qcur;
      asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(qcur) : "v"(bqaddr));
      asm volatile("ds_read_b64 %0, %1 offset:1152" : "=v"(bq[0]) : "v"(bqaddr));
      asm volatile("ds_read_b64 %0, %1 offset:5760" : "=v"(bq[1]) : "v"(bqaddr));
      asm volatile("ds_read_b64 %0, %1 offset:10368" : "=v"(bq[2]) : "v"(bqaddr));
      asm volatile("ds_read_b64 %0, %1 offset:14976" : "=v"(bq[3]) : "v"(bqaddr));
      Pipe<MT, LDL, D, 0, D>::preload(ring, base);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(D));
      __builtin_amdgcn_sched_barrier(0);
      bq[0] += qcur * 1e-30;
    } else {
      Pipe<MT, LDL, D, 0, D>::preload(ring, base);
    }
  };
  auto dma = [&](int it, int k) {  // piece k of this wave, stage `it`
    if constexpr (OPT & 2) {
      double *dst = lds + 16 * LDL + 4096 + (k * 8 + wave) * 128;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(src + ((size_t)blockIdx.x * 64 + (k * 8 + wave)) * 128 + lane * 2),
                                       (LDS_AS void *)dst, 16, 0, 0);
    }
  };
  auto flush = [&](int it) {
    if constexpr (OPT & 4) {
      if (it % 19 == 18) {
        for (int t = 0; t < MT; t++) {
          for (int r = 0; r < 4; r++) gacc[t][r] += acc[t][r] * bq[r];
          acc[t] = (v4d){0, 0, 0, 0};
        }
      }
    }
  };
  const bool late = (MODE == 1) || (MODE >= 2 && wave >= 4);
  if (MODE == 3 && wave >= 4) __builtin_amdgcn_s_setprio(3);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (late || MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      if (MODE != 0) {
        if (OPT & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      double ring[D];
      stage_start(ring, it);
      Pipe<MT, LDL, D, 0, 1>::run(acc, ring, bq, base); dma(it, 0);
      Pipe<MT, LDL, D, 1, 2>::run(acc, ring, bq, base); dma(it, 1);
      Pipe<MT, LDL, D, 2, 3>::run(acc, ring, bq, base); dma(it, 2);
      Pipe<MT, LDL, D, 3, 4>::run(acc, ring, bq, base); dma(it, 3);
      Pipe<MT, LDL, D, 4, 5>::run(acc, ring, bq, base); dma(it, 4);
      Pipe<MT, LDL, D, 5, 6>::run(acc, ring, bq, base); dma(it, 5);
      Pipe<MT, LDL, D, 6, N>::run(acc, ring, bq, base);
      flush(it);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      double ring[D];
      stage_start(ring, it);
      Pipe<MT, LDL, D, 0, H>::run(acc, ring, bq, base);
      if (OPT & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      Pipe<MT, LDL, D, H, H + 1>::run(acc, ring, bq, base); dma(it, 0);
      Pipe<MT, LDL, D, H + 1, H + 2>::run(acc, ring, bq, base); dma(it, 1);
      Pipe<MT, LDL, D, H + 2, H + 3>::run(acc, ring, bq, base); dma(it, 2);
      Pipe<MT, LDL, D, H + 3, H + 4>::run(acc, ring, bq, base); dma(it, 3);
      Pipe<MT, LDL, D, H + 4, H + 5>::run(acc, ring, bq, base); dma(it, 4);
      Pipe<MT, LDL, D, H + 5, H + 6>::run(acc, ring, bq, base); dma(it, 5);
      Pipe<MT, LDL, D, H + 6, N>::run(acc, ring, bq, base);
      flush(it);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int t = 0; t < MT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (auto &g : gacc) s += g[0] + g[1] + g[2] + g[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MT, int MODE, int OPT>
void run(int blocks, int iters) {
  const int threads = 512;
  double *d;
  unsigned long long *c;
  int waves = blocks * threads / 64;
  hipMalloc(&d, sizeof(double) * blocks * threads);
  hipMalloc(&c, sizeof(unsigned long long) * waves);
  const int lds_bytes = (16 * (16 * MT + 16) + 4096 + 48 * 128) * 8;
  double *src;
  hipMalloc(&src, (size_t)blocks * 64 * 128 * 8 + 4096);
  hipMemset(src, 0, (size_t)blocks * 64 * 128 * 8 + 4096);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MT, MODE, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<MT, MODE, OPT>), dim3(blocks), dim3(threads), lds_bytes, 0, d, c, iters, src);
  hipError_t err = hipDeviceSynchronize();
  if (err != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(err));
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), c, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (int w = 0; w < waves; w++) cyc.push_back((double)h[w] / ((double)iters));
  std::sort(cyc.begin(), cyc.end());
  hipFree(src);
  printf("MT=%2d mode=%d opt=%d: cycles per stage: median %.0f max %.0f; ideal %d -> pipe busy %.1f%%\n", MT, MODE, OPT,
         cyc[waves / 2], cyc[waves - 1], 4 * MT * 64 * 2, 4.0 * MT * 64 * 2 / cyc[waves / 2] * 100);
  hipFree(d);
  hipFree(c);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int cu = p.multiProcessorCount;
  run<10, 2, 0>(cu, 3800); run<10, 2, 1>(cu, 3800); run<10, 2, 2>(cu, 3800); run<10, 2, 3>(cu, 3800);
  run<10, 2, 4>(cu, 3800); run<10, 2, 7>(cu, 3800); run<10, 1, 7>(cu, 3800); run<10, 0, 7>(cu, 3800);
  run<19, 2, 3>(cu, 1900);
  return 0;
}
