// bf16 MFMA GEMM for gfx950 (CDNA4), fp32 accumulate.
//
//   C[m,n] = epi( alpha * sum_k A(m,k) B(k,n) ),  A/B each either k-contiguous or k-strided.
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves, 2x2, 64x64 per wave, 4x4 fragments of
// v_mfma_f32_16x16x32_bf16).  Global -> registers -> LDS staging (issue-early / write-late,
// double-buffered LDS, one barrier per K-tile); out-of-range rows come back as zeros from the
// buffer-resource bounds check, the K tail is predicated per 16-byte chunk.
//   k-contiguous operand : LDS image [128 rows][128 B], 16-B chunk index XOR (row & 7); fragments by
//                          ds_read_b128 (conflict-free for the 16-lane read groups).
//   k-strided operand    : LDS image [64 k-rows][256 B], 32-B chunk index XOR f(k); fragments by two
//                          ds_read_b64_tr_b16 (hardware transpose), conflict-free per 32-lane half.
// The MFMA is issued as D = Bfrag x Afrag so each lane ends up with 4 consecutive n for one m-row
// (8-byte bf16 / 16-byte fp32 stores).
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

struct GemmParams {
  vita_gemm_desc d;
  int tiles_m, tiles_n, k_tiles_per_split;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ---- LDS images ------------------------------------------------------------------------------
// layout 0 (k-contiguous): row in [0,128), 16-B chunk c in [0,8)
__device__ __forceinline__ uint32_t lds_off_l0(int row, int c) { return row * 128 + ((c ^ (row & 7)) << 4); }
// layout 1 (k-strided): k-row r in [0,64), 16-B chunk c in [0,16) along the free dim
__device__ __forceinline__ uint32_t lds_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }
__device__ __forceinline__ uint32_t lds_off_l1(int r, int c) {
  return r * 256 + ((((c >> 1) ^ lds_f(r))) << 5) + ((c & 1) << 4);
}

template <int LAYOUT>
struct Stager {
  u32x4 reg[4];
  __amdgpu_buffer_rsrc_t rs;
  uint32_t voff[4];   // byte offset of this thread's chunk i at k-tile 0
  uint32_t kstep;     // byte increment per k-tile
  int kc[4];          // k index (elements) covered by chunk i at k-tile 0 (layout 0 only)
  uint32_t loff[4];
  int64_t K;

  // base: operand pointer already offset to the batch; free0: first free index of the tile;
  // nfree: total free extent; ld in elements
  __device__ __forceinline__ void init(const uint16_t* base, int64_t free0, int64_t nfree, int64_t Ktot,
                                       int64_t ld, int k_begin, int tid) {
    K = Ktot;
    if (LAYOUT == 0) {
      int64_t rows = nfree - free0; if (rows > BM) rows = BM;
      rs = make_rsrc(base + free0 * ld, (uint32_t)(rows * ld * 2));
      kstep = BK * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i, c = tid & 7;
        kc[i] = k_begin + c * 8;
        voff[i] = (uint32_t)(row * ld * 2 + kc[i] * 2);
        loff[i] = lds_off_l0(row, c);
      }
    } else {
      // whole remaining matrix from column free0: rows k >= K fall outside -> zeros
      // rows < K-1 may wrap into the next row (in bounds, feeds discarded outputs only); the last
      // row ends exactly at nfree so nothing past the operand is ever touched
      int64_t nfree8 = (nfree + 7) & ~(int64_t)7; if (nfree8 > ld) nfree8 = ld;   // whole 16-B chunks
      int64_t bytes = ((Ktot - 1) * ld + (nfree8 - free0)) * 2;
      rs = make_rsrc(base + free0, (uint32_t)bytes);
      kstep = (uint32_t)(BK * ld * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 4) + 16 * i, c = tid & 15;
        kc[i] = 0;
        voff[i] = (uint32_t)(((int64_t)(k_begin + r) * ld + c * 8) * 2);
        loff[i] = lds_off_l1(r, c);
      }
    }
  }
  __device__ __forceinline__ void load(int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i] + kt * kstep, 0, 0);
      if (LAYOUT == 0) {
        if (kc[i] + kt * BK >= K) v = u32x4{0, 0, 0, 0};
      }
      reg[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* lds) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(lds + loff[i]) = reg[i];
  }
};

// fragment for 16 free indices starting at f0 (tile-local), k-step ks (32 k each)
template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int f0, int ks, int lane) {
  if (LAYOUT == 0) {
    const int row = f0 + (lane & 15);
    const int c = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + lds_off_l0(row, c));
  } else {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int r0 = ks * 32 + 8 * g + q;
    const int cc0 = f0 >> 4;
    const uint32_t o0 = r0 * 256 + ((cc0 ^ lds_f(r0)) << 5) + p * 8;
    const int r1 = r0 + 4;
    const uint32_t o1 = r1 * 256 + ((cc0 ^ lds_f(r1)) << 5) + p * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <int LA, int LB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const vita_gemm_desc& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive tiles of one XCD share the A row panel
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / P.tiles_n, tn = tile % P.tiles_n;
  const int split = blockIdx.y;
  const int64_t zo = blockIdx.z / d.batch_inner, zi = blockIdx.z % d.batch_inner;
  const uint16_t* A = (const uint16_t*)d.a + zo * d.a_stride_outer + zi * d.a_stride_inner;
  const uint16_t* B = (const uint16_t*)d.b + zo * d.b_stride_outer + zi * d.b_stride_inner;
  const int64_t c_off = zo * d.c_stride_outer + zi * d.c_stride_inner;

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int kt_total = (int)((d.k + BK - 1) / BK);
  const int kt_begin = split * P.k_tiles_per_split;
  int kt_end = kt_begin + P.k_tiles_per_split; if (kt_end > kt_total) kt_end = kt_total;
  const int nkt = kt_end - kt_begin;

#define LDSA(buf) (smem + (buf) * 32768)
#define LDSB(buf) (smem + (buf) * 32768 + 16384)

  Stager<LA> sa; Stager<LB> sb;
  sa.init(A, m0, d.m, d.k, d.lda, kt_begin * BK, tid);
  sb.init(B, n0, d.n, d.k, d.ldb, kt_begin * BK, tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) {
    sa.load(0); sb.load(0);
    sa.store(LDSA(0)); sb.store(LDSB(0));
  }
  __syncthreads();

  for (int t = 0; t < nkt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nkt) { sa.load(t + 1); sb.load(t + 1); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<LA>(LDSA(cur), wm * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<LB>(LDSB(cur), wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nkt) { sa.store(LDSA(cur ^ 1)); sb.store(LDSB(cur ^ 1)); }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+(lane&15)][n = ..+4*(lane>>4)+0..3]
  const float alpha = d.alpha;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= d.m) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= d.n) continue;   // n is a multiple of 4 and d.n % 4 == 0 unless maps/atomics are used
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * alpha;
      if (d.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < d.n) v[r] += d.bias[n + r];
      }
      if (d.split_k > 1 || d.row_seg_phys || d.col_seg_phys || (d.n & 3)) {
        // scalar path: fp32 C, optional index maps, optional atomic accumulation
        const int64_t lm = seg_map(m, d.row_seg_phys, d.row_seg_log);
        if (lm < 0) continue;
        float* C = (float*)d.c + c_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= d.n) continue;
          const int64_t ln = seg_map(n + r, d.col_seg_phys, d.col_seg_log);
          if (ln < 0) continue;
          if (d.split_k > 1) atomicAdd(C + lm * d.ldc + ln, v[r]);
          else if (d.c_dtype == VITA_DT_F32) C[lm * d.ldc + ln] = v[r];
          else ((uint16_t*)d.c + c_off)[lm * d.ldc + ln] = f32_to_bf16_bits(v[r]);
        }
        continue;
      }
      const int64_t o = c_off + m * d.ldc + n;
      if (d.epilogue == VITA_EPI_GELU) {
        float dg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { float g; gelu_erf_both(v[r], g, dg[r]); v[r] = g; }
        u32x2 u = {pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3])};
        *reinterpret_cast<u32x2*>((uint16_t*)d.aux_out + o) = u;
      } else if (d.epilogue == VITA_EPI_RESIDUAL) {
        const f32x4 res = *reinterpret_cast<const f32x4*>((const float*)d.aux_in + c_off + m * d.ld_aux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += res[r];
      } else if (d.epilogue == VITA_EPI_GELU_BWD) {
        const u32x2 u = *reinterpret_cast<const u32x2*>((const uint16_t*)d.aux_in + c_off + m * d.ld_aux + n);
        v[0] *= bf16_bits_to_f32((uint16_t)(u[0] & 0xffff));
        v[1] *= bf16_bits_to_f32((uint16_t)(u[0] >> 16));
        v[2] *= bf16_bits_to_f32((uint16_t)(u[1] & 0xffff));
        v[3] *= bf16_bits_to_f32((uint16_t)(u[1] >> 16));
      }
      if (d.c_dtype == VITA_DT_F32) {
        *reinterpret_cast<f32x4*>((float*)d.c + o) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        u32x2 u = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>((uint16_t*)d.c + o) = u;
      }
    }
  }
}

// gemm256.hip
bool vita_gemm256_supported(const vita_gemm_desc& d);
double vita_gemm256_cost(const vita_gemm_desc& d, int* nf_out);
double vita_gemm128_cost(const vita_gemm_desc& d);
int vita_gemm256_launch(const vita_gemm_desc& d, int nf, hipStream_t s);

// variant: 0 = choose by shape, 1 = 128x128 kernel, 8 / 10 / 12 / 13 / 14 / 16 = 256 x (16*variant) kernel
extern "C" int vita_gemm_bf16_ex(const vita_gemm_desc* dp, int32_t variant, void* stream) {
  if (!dp) return VITA_ERR_INVALID_ARG;
  vita_gemm_desc d = *dp;
  VITA_REQUIRE(d.a && d.b && d.c);
  VITA_REQUIRE(d.m > 0 && d.n > 0 && d.k > 0);
  VITA_REQUIRE(d.lda % 8 == 0 && d.ldb % 8 == 0);
  VITA_REQUIRE(((uintptr_t)d.a & 15) == 0 && ((uintptr_t)d.b & 15) == 0);
  VITA_REQUIRE((d.a_stride_outer % 8) == 0 && (d.a_stride_inner % 8) == 0);
  VITA_REQUIRE((d.b_stride_outer % 8) == 0 && (d.b_stride_inner % 8) == 0);
  VITA_REQUIRE(d.a_layout == 0 || d.a_layout == 1);
  VITA_REQUIRE(d.b_layout == 0 || d.b_layout == 1);
  VITA_REQUIRE(d.c_dtype == VITA_DT_BF16 || d.c_dtype == VITA_DT_F32);
  if (d.batch_outer < 1) d.batch_outer = 1;
  if (d.batch_inner < 1) d.batch_inner = 1;
  if (d.split_k < 1) d.split_k = 1;
  if (d.alpha == 0.f) d.alpha = 1.f;
  const bool scalar_path = d.split_k > 1 || d.row_seg_phys || d.col_seg_phys || (d.n & 3);
  if (scalar_path) {
    VITA_REQUIRE(d.epilogue == VITA_EPI_NONE);
    if (d.split_k > 1) VITA_REQUIRE(d.c_dtype == VITA_DT_F32);
  } else {
    const int esz = d.c_dtype == VITA_DT_F32 ? 4 : 2;
    VITA_REQUIRE(d.ldc % 4 == 0 && ((uintptr_t)d.c % (4 * esz)) == 0);
    VITA_REQUIRE(d.c_stride_outer % 4 == 0 && d.c_stride_inner % 4 == 0);
    if (d.epilogue == VITA_EPI_GELU) VITA_REQUIRE(d.aux_out && ((uintptr_t)d.aux_out & 7) == 0 && d.c_dtype == VITA_DT_BF16);
    if (d.epilogue == VITA_EPI_RESIDUAL) VITA_REQUIRE(d.aux_in && d.ld_aux % 4 == 0 && ((uintptr_t)d.aux_in & 15) == 0);
    if (d.epilogue == VITA_EPI_GELU_BWD) VITA_REQUIRE(d.aux_in && d.ld_aux % 4 == 0 && ((uintptr_t)d.aux_in & 7) == 0);
  }
  // 32-bit buffer offsets
  const int64_t a_rows = d.a_layout == 0 ? BM : d.k, b_rows = d.b_layout == 0 ? BN : d.k;
  VITA_REQUIRE(a_rows * d.lda * 2 < (int64_t)0xffffffffll && b_rows * d.ldb * 2 < (int64_t)0xffffffffll);

  if (variant != 1) {
    const bool ok = vita_gemm256_supported(d);
    if (variant > 1) {
      VITA_REQUIRE(ok && (variant == 8 || variant == 10 || variant == 12 || variant == 13 || variant == 14 || variant == 16 ||
                          variant == 108 || variant == 110 || variant == 112));
      return vita_gemm256_launch(d, variant, (hipStream_t)stream);
    }
    // large un-batched problems: whichever kernel needs fewer (weighted) rounds of workgroups
    if (ok && d.m >= 512 && d.n >= 208 && d.k >= 256) {
      int nf = 0;
      if (vita_gemm256_cost(d, &nf) <= vita_gemm128_cost(d)) {
        return vita_gemm256_launch(d, nf, (hipStream_t)stream);
      }
    }
  }
  GemmParams P;
  P.d = d;
  P.tiles_m = (int)((d.m + BM - 1) / BM);
  P.tiles_n = (int)((d.n + BN - 1) / BN);
  const int kt_total = (int)((d.k + BK - 1) / BK);
  P.k_tiles_per_split = (kt_total + d.split_k - 1) / d.split_k;
  dim3 grid(P.tiles_m * P.tiles_n, d.split_k, (unsigned)(d.batch_outer * d.batch_inner));
  VITA_REQUIRE((int64_t)P.tiles_m * P.tiles_n < (1ll << 31) && d.batch_outer * d.batch_inner < 65536);
  const size_t lds = 65536;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(LA, LB)                                                                              \
  do {                                                                                              \
    VITA_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_kernel<LA, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((gemm_kernel<LA, LB>), grid, dim3(NTHREADS), lds, s, P);                     \
  } while (0)
  if (d.a_layout == 0 && d.b_layout == 0) LAUNCH(0, 0);
  else if (d.a_layout == 0 && d.b_layout == 1) LAUNCH(0, 1);
  else if (d.a_layout == 1 && d.b_layout == 0) LAUNCH(1, 0);
  else LAUNCH(1, 1);
#undef LAUNCH
  VITA_CHECK_LAUNCH("vita_gemm_bf16");
  return VITA_OK;
}

extern "C" int vita_gemm_bf16(const vita_gemm_desc* dp, void* stream) { return vita_gemm_bf16_ex(dp, 0, stream); }
