// Shared device/host helpers for libvita_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/vita_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __builtin_bit_cast(float, ((uint32_t)b) << 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exact-erf GELU (nn.GELU() default, timm Mlp) evaluated with the Abramowitz-Stegun 7.1.26 rational
// approximation of erf (|abs error| <= 1.5e-7, far below the bf16 resolution of the tensors it feeds):
//   Phi(x) = 1 - 0.5 * poly(t) * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt(2))      (x >= 0; mirrored for x < 0)
// ~14 VALU instructions instead of libdevice erff's ~60; exp(-x^2/2) is shared with the density in the gradient.
__device__ __forceinline__ float gelu_half_tail(float x, float& e) {   // returns 0.5 * poly(t) * exp(-x^2/2) = Phi(-|x|)
  const float ax = __builtin_fabsf(x);
  e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);            // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.23164189f, 1.0f));   // p / sqrt(2) = 0.3275911 * 0.70710678
  float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  poly = __builtin_fmaf(t, poly, 1.421413741f);
  poly = __builtin_fmaf(t, poly, -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.254829592f);
  return 0.5f * poly * t * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float e;
  const float tail = gelu_half_tail(x, e);
  return x * (x >= 0.f ? 1.0f - tail : tail);
}
// value and derivative together (the forward fc1 epilogue saves the derivative for the backward pass)
__device__ __forceinline__ void gelu_erf_both(float x, float& g, float& dg) {
  float e;
  const float tail = gelu_half_tail(x, e);
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  g = x * cdf;
  dg = __builtin_fmaf(x * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float e;
  const float tail = gelu_half_tail(x, e);
  const float cdf = x >= 0.f ? 1.0f - tail : tail;
  return __builtin_fmaf(x * 0.3989422804014327f, e, cdf);
}

// physical -> logical index through a segment map (0 = identity); returns -1 when dropped
__device__ __host__ __forceinline__ int64_t seg_map(int64_t i, int32_t seg_phys, int32_t seg_log) {
  if (seg_phys == 0) return i;
  const int64_t q = i / seg_phys, r = i % seg_phys;
  return r < seg_log ? q * seg_log + r : -1;
}

// Host-side set-up that must happen once per device and call site (hipFuncSetAttribute is per device): thread-safe, so
// calls may arrive from the Python main thread and autograd's worker thread at once, and a process that drives several
// GPUs sets the attribute on each of them.
#define VITA_MAX_DEVICES 64
static inline int vita_cur_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= VITA_MAX_DEVICES) d = 0;
  return d;
}
struct VitaPerDeviceOnce { std::once_flag f[VITA_MAX_DEVICES]; };
#define VITA_ONCE_PER_DEVICE(...)                                            \
  do {                                                                       \
    static VitaPerDeviceOnce once__;                                         \
    std::call_once(once__.f[vita_cur_device()], [&]() { __VA_ARGS__; });     \
  } while (0)

// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, lds) once per device for the call site that owns `once`
// (the kernel's address is taken by the caller, outside any lambda: hipcc's host pass otherwise drops the launch
// stubs of all but the first instantiation referenced from one function)
static inline void vita_set_max_lds_once(VitaPerDeviceOnce& once, const void* fn, int lds) {
  std::call_once(once.f[vita_cur_device()], [=]() { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
}

void vita_set_error(const char* what, hipError_t e);
#define VITA_CHECK_LAUNCH(name)                      \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) {                         \
      vita_set_error(name, e__);                     \
      return VITA_ERR_LAUNCH;                        \
    }                                                \
  } while (0)
#define VITA_REQUIRE(cond)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      vita_set_error("invalid argument: " #cond, hipSuccess); \
      return VITA_ERR_INVALID_ARG;                           \
    }                                                        \
  } while (0)
