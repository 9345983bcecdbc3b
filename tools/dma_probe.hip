// LDS-DMA / MFMA interference probe (development aid): every wave of an 8-wave workgroup runs `iters` rounds of
// NDMA x 1 KiB `buffer_load ... lds` (or plain 16-B global loads into registers, MODE 2) from an L2-resident region
// followed by `nm` register-only v_mfma_f32_16x16x32_bf16, with at most one round of loads in flight.
// Built by tools/dma_probe.py into tools/dma_probe.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE, int NDMA>     // MODE 0: no loads, 1: LDS-DMA, 2: loads into registers
__global__ __launch_bounds__(512) void probe(const uint16_t* __restrict__ src, int iters, int nm, float* out, int reuse,
                                             uint32_t total_chunks, int issuers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // reuse == 0: 64 regions of 64 KiB, L2-resident.  reuse = R > 0: a fresh 64 KiB chunk every round, shared by the R
  // workgroups of a group on one XCD (1/R of the stream misses L2, like the operand stream of a GEMM)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(src), 0, 0xfffffff0u, 0x00020000);
  const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const uint32_t groups = reuse > 0 ? (gridDim.x / 8 + reuse - 1) / reuse : 1;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 sink = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    uint32_t chunk = blockIdx.x & 63;
    if (reuse > 0) chunk = (((uint32_t)it * groups + slot / reuse) * 8 + xcd) % total_chunks;
    const uint32_t soff = __builtin_amdgcn_readfirstlane(chunk * 65536u);
#pragma unroll
    for (int p = 0; p < NDMA; ++p) {
      // issuers = 8: every wave issues its own piece; 4 / 2: waves 0-3 / 0-1 issue two / four pieces each
      const int per = 8 / issuers;
      if (MODE == 1 && wave < issuers) {
        for (int q = 0; q < per; ++q) {
          const int w = wave * per + q;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (p * 8 + w) * 1024), 16,
                                                   (uint32_t)((p * 8 + w) * 1024 + lane * 16), soff, 0, 0);
        }
      }
      const uint32_t off = (uint32_t)((p * 8 + wave) * 1024 + lane * 16);
      if (MODE == 2) { i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0)); sink ^= v; }
      for (int m = 0; m < nm; m += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
    if (MODE == 1 && issuers == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    if (MODE == 1 && issuers == 4 && wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
    if (MODE == 1 && issuers == 2 && wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.678f || sink[0] == 0x12345678) out[0] = s;
}

extern "C" int dma_probe(int mode, int wgs, int iters, int nm, const void* src, float* out, void* stream, int reuse,
                         unsigned total_chunks, int issuers) {
  const size_t lds = 8 * 8 * 1024;
  if (issuers != 8 && issuers != 4 && issuers != 2) return -1;
#define GO(M) hipLaunchKernelGGL((probe<M, 8>), dim3(wgs), dim3(512), lds, (hipStream_t)stream, (const uint16_t*)src, iters, nm, out, reuse, total_chunks, issuers)
  if (mode == 0) GO(0);
  if (mode == 1) GO(1);
  if (mode == 2) GO(2);
#undef GO
  return (int)hipGetLastError();
}
