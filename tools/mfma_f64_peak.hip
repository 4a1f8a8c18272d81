// Microbenchmark: peak rate of v_mfma_f64_16x16x4_f64 on this device (roofline denominator check).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int threads, int iters) {
  double *d;
  hipMalloc(&d, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0, 2.0);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double waves = (double)blocks * threads / 64;
  double flops = waves * (double)iters * NACC * 2048.0;
  printf("NACC=%d blocks=%d threads=%d iters=%d: %.3f ms  %.2f TFLOP/s\n", NACC, blocks, threads, iters,
         best, flops / best * 1e-9);
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run<4>(p.multiProcessorCount, 256, 20000);
  run<8>(p.multiProcessorCount, 256, 10000);
  run<16>(p.multiProcessorCount, 256, 5000);
  run<8>(p.multiProcessorCount, 512, 10000);
  run<8>(p.multiProcessorCount * 2, 256, 10000);
  run<1>(p.multiProcessorCount, 256, 40000);
  return 0;
}
