// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on this device and the clock the chip holds
// under that load (roofline denominator check).
// Build: hipcc --offload-arch=gfx950 -O3 -w mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double *out, unsigned long long *clk, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    clk[2 * w] = t1 - t0;
    clk[2 * w + 1] = r1 - r0;
  }
}

template <int NACC>
void run(int blocks, int threads, int iters) {
  double *d;
  unsigned long long *c;
  int waves = blocks * threads / 64;
  hipMalloc(&d, sizeof(double) * blocks * threads);
  hipMalloc(&c, sizeof(unsigned long long) * 2 * waves);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, c, iters, 1.0, 2.0);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, c, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h(2 * waves);
  hipMemcpy(h.data(), c, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
  std::vector<double> cyc, ghz;
  for (int w = 0; w < waves; w++) {
    cyc.push_back((double)h[2 * w] / ((double)iters * NACC));
    ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  double flops = (double)waves * (double)iters * NACC * 2048.0;
  printf("NACC=%2d blocks=%4d threads=%3d (%.1f waves/SIMD): %.3f ms %.2f TFLOP/s | cycles per MFMA per wave (median) %.1f | clock (median) %.2f GHz\n",
         NACC, blocks, threads, waves / 1024.0, best, flops / best * 1e-9, cyc[waves / 2], ghz[waves / 2]);
  hipFree(d);
  hipFree(c);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int cu = p.multiProcessorCount;
  run<1>(cu, 256, 40000);
  run<2>(cu, 256, 20000);
  run<4>(cu, 256, 20000);
  run<8>(cu, 256, 10000);
  run<16>(cu, 256, 5000);
  run<8>(cu, 512, 10000);
  run<8>(cu, 1024, 5000);
  run<4>(cu, 1024, 10000);
  run<8>(cu / 8, 256, 10000);   // 1/8 of the chip: clock when mostly idle
  return 0;
}
