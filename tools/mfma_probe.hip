// Register-only MFMA ceiling probe (development aid): every wave runs `iters` rounds of NACC independent
// v_mfma_f32_32x32x16_bf16 on registers - no LDS, no memory.  Built by tools_mfma_probe.py into tools_mfma_probe.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_spin(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = s;
}

// the same loop with v_mfma_f32_16x16x32_bf16 (half the FLOPs per instruction, NACC x 2 instructions per round)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void mfma_spin16(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.678f) out[0] = s;
}

extern "C" int mfma_probe(int wgs, int threads, int iters, float* out, void* stream) {
  hipLaunchKernelGGL((mfma_spin<8>), dim3(wgs), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
extern "C" int mfma_probe16(int wgs, int threads, int iters, float* out, void* stream) {
  hipLaunchKernelGGL((mfma_spin16<16>), dim3(wgs), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
