// MFMA / LDS helpers shared by the MFMA kernels (mttkrp_kernel_v3.hip, ttm_kernel.hip).
#ifndef CALS_MFMA_COMMON_H
#define CALS_MFMA_COMMON_H

#include "cals_hip_internal.h"

#include <type_traits>
#include <utility>

namespace calship {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// CALS_DIAG builds keep the in-kernel stamps and the timing-only switches (tools/clock_probe.py,
// tools/time_probe.py); the production build compiles them out of the MFMA loop.
#ifdef CALS_DIAG
#define DIAG(x) (x)
#else
#define DIAG(x) (false)
#endif

template <typename T> struct Acc;
template <> struct Acc<double> {
  typedef v4d type;
  static __device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 C/D layout: lane holds D[row = (lane>>4) + 4*reg][col = lane&15]
  static __device__ __forceinline__ int row(int krow, int reg) { return krow + 4 * reg; }
};
template <> struct Acc<float> {
  typedef v4f type;
  static __device__ __forceinline__ v4f mfma(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // f32 C/D layout: lane holds D[row = 4*(lane>>4) + reg][col = lane&15]
  static __device__ __forceinline__ int row(int krow, int reg) { return 4 * krow + reg; }
};

template <typename T>
static __device__ __forceinline__ void lds_read(T &dst, unsigned addr);
template <>
__device__ __forceinline__ void lds_read<double>(double &dst, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr));
}
template <>
__device__ __forceinline__ void lds_read<float>(float &dst, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(addr));
}
template <typename T, int OFF>
static __device__ __forceinline__ void lds_read_off(T &dst, unsigned addr) {
  if constexpr (std::is_same<T, double>::value)
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
  else
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

}  // namespace calship
#endif
