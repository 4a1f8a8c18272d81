// How should a caller's pageable tensor (216 MB at C3) get to the device?  Times (a) hipMemcpy from pageable memory,
// (b) hipHostRegister + hipMemcpy + hipHostUnregister, (c) a threaded copy into pinned staging chunks + async copies.
// Build: hipcc --offload-arch=gfx950 -O2 -pthread tools/h2d_probe.hip -o tools/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = (size_t)300 * 300 * 300 * 8;
  double *h = (double *)malloc(bytes);
  for (size_t i = 0; i < bytes / 8; i++) h[i] = (double)i;
  void *d;
  hipMalloc(&d, bytes);
  hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);  // warm-up (first touch of everything)
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    double t1 = now();
    printf("pageable hipMemcpy        : %6.1f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) * 1e-9);
  }
  for (int rep = 0; rep < 3; rep++) {
    double t0 = now();
    hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
    double t1 = now();
    hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
    double t2 = now();
    hipHostUnregister(h);
    double t3 = now();
    printf("register %s %6.1f ms + copy %6.1f ms + unregister %6.1f ms = %6.1f ms\n", e == hipSuccess ? "ok" : "FAILED",
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
  }
  const size_t chunk = (size_t)16 << 20;
  const int nbuf = 4;
  char *pin[nbuf];
  for (int b = 0; b < nbuf; b++) hipHostMalloc((void **)&pin[b], chunk, hipHostMallocDefault);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t ev[nbuf];
  for (int b = 0; b < nbuf; b++) hipEventCreate(&ev[b]);
  for (int threads : {1, 2, 4, 8}) {
    double t0 = now();
    size_t off = 0;
    int k = 0;
    while (off < bytes) {
      const size_t n = std::min(chunk, bytes - off);
      const int b = k % nbuf;
      if (k >= nbuf) hipEventSynchronize(ev[b]);
      std::vector<std::thread> th;
      for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
          const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
          memcpy(pin[b] + lo, (char *)h + off + lo, hi - lo);
        });
      for (auto &x : th) x.join();
      hipMemcpyAsync((char *)d + off, pin[b], n, hipMemcpyHostToDevice, st);
      hipEventRecord(ev[b], st);
      off += n;
      k++;
    }
    hipStreamSynchronize(st);
    double t1 = now();
    printf("staged, %d copy thread(s)  : %6.1f ms  %5.1f GB/s\n", threads, (t1 - t0) * 1e3, bytes / (t1 - t0) * 1e-9);
  }
  return 0;
}
