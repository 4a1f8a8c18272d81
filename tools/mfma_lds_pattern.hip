// Microbenchmark: the MTTKRP inner-loop pattern in isolation -- [counted lgkmcnt wait, f64 MFMA,
// ds_read_b64 of a later operand] over 19 accumulators, no barriers, no DMA -- to see what the
// pattern itself costs per MFMA at 1 and 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -w mfma_lds_pattern.hip -o mfma_lds_pattern
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

template <int MODE>  // 0: pattern with LDS reads; 1: MFMA only, same accumulators
__global__ void __launch_bounds__(512, 2) k(double *out, unsigned long long *clk, int iters) {
  constexpr int MT = 19, LDL = 304, D = 6;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int i = threadIdx.x; i < 16 * LDL; i += blockDim.x) lds[i] = 1.0 + (i & 7) * 1e-3;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v4d acc[MT];
  for (int t = 0; t < MT; t++) acc[t] = (v4d){0, 0, 0, 0};
  double bq[4] = {1.0 + lane * 1e-9, 1.1, 1.2, 1.3};
  const unsigned base = (unsigned)(size_t)((LDS_AS const char *)(lds + (lane >> 4) * LDL + (lane & 15)));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      double ring[D];
      Pipe<MT, LDL, D, 0, D>::preload(ring, base);
      Pipe<MT, LDL, D, 0, 4 * MT>::run(acc, ring, bq, base);
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[0], bq[q], acc[t], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int t = 0; t < MT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(int blocks, int threads, int iters) {
  double *d;
  unsigned long long *c;
  int waves = blocks * threads / 64;
  hipMalloc(&d, sizeof(double) * blocks * threads);
  hipMalloc(&c, sizeof(unsigned long long) * waves);
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 304 * 8);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 16 * 304 * 8, 0, d, c, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(waves);
  hipMemcpy(h.data(), c, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (int w = 0; w < waves; w++) cyc.push_back((double)h[w] / ((double)iters * 76));
  std::sort(cyc.begin(), cyc.end());
  printf("mode=%d blocks=%d threads=%d (%.1f waves/SIMD): cycles per MFMA per wave: median %.1f max %.1f -> pipe busy %.1f%%\n",
         MODE, blocks, threads, waves / 1024.0, cyc[waves / 2], cyc[waves - 1], 64.0 * (waves / 1024.0) / cyc[waves / 2] * 100);
  hipFree(d);
  hipFree(c);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int cu = p.multiProcessorCount;
  run<1>(cu, 256, 2000);
  run<1>(cu, 512, 2000);
  run<0>(cu, 256, 2000);
  run<0>(cu, 512, 2000);
  return 0;
}
