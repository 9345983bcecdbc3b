// HBM-bound row kernels of the ViTa hot path (gfx950): LayerNorm fwd/bwd (one wave per row,
// 64-lane shuffle reductions), rank-by-counting argsort + mask, row gather/scatter, im2col of
// 3D+T patches, token assembly (+pos-embed, cls), casts, column sums, materialised-softmax rows,
// weight-shadow packing and Adam.
#include "common.h"

static thread_local char g_err[256] = "";
void vita_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
}
extern "C" const char* vita_last_error(void) { return g_err; }
extern "C" int vita_version(void) { return 100; }

// One wave that keeps its CU busy for `us` microseconds of the constant 100 MHz wall clock (the loop ends on time, whatever
// the core clock does).  vita_amd.nn uses two of them to find out whether two HIP streams run concurrently: streams that
// HIP maps to one hardware queue serialise, and the second stream of the Block backward must not share the main stream's.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int vita_spin_us(int64_t us, void* stream) {
  VITA_REQUIRE(us > 0 && us <= 100000);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100));
  VITA_CHECK_LAUNCH("vita_spin_us");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- LayerNorm
#define LN_MAXV 8   // float4 per lane -> d <= 2048
// R rows per wave: every row's loads are in flight before the first reduction (more bytes in flight per wave
// than one row gives; a one-row-per-wave version reads at 2.0-2.3 TB/s, two rows 2.6-2.7)
#define LN_FWD_R 4
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     void* __restrict__ y, int64_t ldy, int y_f32,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int d, float eps) {
  constexpr int R = LN_FWD_R;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  f32x4 v[R][NV];
  float s[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int64_t row = row0 + k < rows ? row0 + k : rows - 1;      // clamped: loads stay in bounds, stores are guarded
    const float* xr = x + row * ldx;
    s[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      v[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) v[k][i] = *reinterpret_cast<const f32x4*>(xr + c);
    }
  }
  const float inv_d = 1.f / (float)d;
  float mu[R], rs[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) { if (c + r >= d) v[k][i][r] = 0.f; s[k] += v[k][i][r]; }
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) mu[k] = wave_sum(s[k]) * inv_d;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (c + r < d) { const float t = v[k][i][r] - mu[k]; q += t * t; }
    }
    s[k] = q;
  }
#pragma unroll
  for (int k = 0; k < R; ++k) rs[k] = rsqrtf(wave_sum(s[k]) * inv_d + eps);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < R; ++k) if (row0 + k < rows) { mean[row0 + k] = mu[k]; rstd[row0 + k] = rs[k]; }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= ldy) continue;
    float g[4], bt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { g[r] = (c + r < d) ? gamma[c + r] : 0.f; bt[r] = (c + r < d) ? beta[c + r] : 0.f; }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (row0 + k >= rows) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (c + r < d) ? (v[k][i][r] - mu[k]) * rs[k] * g[r] + bt[r] : 0.f;
      if (y_f32) *reinterpret_cast<f32x4*>((float*)y + (row0 + k) * ldy + c) = f32x4{o[0], o[1], o[2], o[3]};
      else *reinterpret_cast<u32x2*>((uint16_t*)y + (row0 + k) * ldy + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
  }
}

extern "C" int vita_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y,
                                  int64_t ldy, int32_t y_dtype, float* mean, float* rstd, int64_t rows,
                                  int64_t d, float eps, void* stream) {
  VITA_REQUIRE(x && gamma && beta && y && mean && rstd);
  VITA_REQUIRE(rows > 0 && d > 0 && d <= LN_MAXV * 256 && ldy <= LN_MAXV * 256);
  VITA_REQUIRE(ldx % 4 == 0 && ldx >= ((d + 3) / 4) * 4 && ldy % 4 == 0 && ldy >= d);
  VITA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
  const int64_t width = ldy > d ? ldy : d;
#define LN_FWD_LAUNCH(NV)                                                                                                     \
  hipLaunchKernelGGL(ln_fwd_kernel<NV>, dim3((unsigned)((rows + 4 * LN_FWD_R - 1) / (4 * LN_FWD_R))), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, \
                     beta, y, ldy, y_dtype == VITA_DT_F32, mean, rstd, rows, (int)d, eps)
  if (width <= 512) LN_FWD_LAUNCH(2);
  else if (width <= 1280) LN_FWD_LAUNCH(5);
  else LN_FWD_LAUNCH(8);
#undef LN_FWD_LAUNCH
  VITA_CHECK_LAUNCH("vita_layernorm_fwd");
  return VITA_OK;
}

#define LN_BWD_MAXBLK 768     // 3 blocks (12 waves, 167 VGPRs each) are resident per CU: 256 x 3 = one full round
extern "C" int64_t vita_layernorm_bwd_nblk(int64_t rows) {
  int64_t n = (rows + 15) / 16;      // at least 4 rows per wave: the per-block partial reduction is not free
  return n < LN_BWD_MAXBLK ? n : LN_BWD_MAXBLK;
}

// NV = float4 per lane (d <= NV*256).  One wave per row; per-lane dgamma/dbeta partials live in
// registers across the rows a wave owns and are reduced over the block's 4 waves through LDS.
template <int NV>
__global__ __launch_bounds__(256, 3) void ln_bwd_kernel(const void* __restrict__ dy, int64_t lddy, int dy_f32,
                                                        const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ dx_in,
                                                        float* __restrict__ dx_out, int64_t lddx, uint16_t* __restrict__ dxb,
                                                        int64_t ld_dxb, float* __restrict__ partial, int64_t rows, int d) {
  __shared__ float red[2][NV * 256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // row addresses stay scalar
  const int nblk = gridDim.x;
  // per-lane column masks: msk[i][r] = 1 for real columns, 0 for pad columns (only the last chunk is mixed)
  f32x4 g[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    ag[i] = ab[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) if (c + r < d) g[i][r] = gamma[c + r];
  }
  const float inv_d = 1.f / (float)d;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)nblk * 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NV], dyv[NV], din[NV];
    float s1 = 0.f, s2 = 0.f;
    // the residual gradient is fetched with x and dy (one memory latency per row, not two)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      din[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dx_in && c < d) din[i] = *reinterpret_cast<const f32x4*>(dx_in + row * lddx + c);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      xh[i] = dyv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
        if (dy_f32) dyv[i] = *reinterpret_cast<const f32x4*>((const float*)dy + row * lddy + c);
        else {
          const u32x2 u = *reinterpret_cast<const u32x2*>((const uint16_t*)dy + row * lddy + c);
          dyv[i] = f32x4{__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                         __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u)};
        }
        // pad columns: gamma is 0 there, so s1/s2 ignore them; dy pads are masked for dgamma/dbeta below
        xh[i] = (xv - mu) * rs;
        const f32x4 t = dyv[i] * g[i];
        s1 += t[0] + t[1] + t[2] + t[3];
        const f32x4 tx = t * xh[i];
        s2 += tx[0] + tx[1] + tx[2] + tx[3];
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= lddx) continue;
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) {
        if (c + 4 > d) {   // mixed chunk: zero the pad lanes of dy so that dbeta / dx pads stay exact zeros
#pragma unroll
          for (int r = 0; r < 4; ++r) if (c + r >= d) { dyv[i][r] = 0.f; xh[i][r] = 0.f; }
        }
        ag[i] += dyv[i] * xh[i];
        ab[i] += dyv[i];
        o = (dyv[i] * g[i] - c1 - xh[i] * c2) * rs;
        if (c + 4 > d) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (c + r >= d) o[r] = 0.f;
        }
        o += din[i];
      }
      *reinterpret_cast<f32x4*>(dx_out + row * lddx + c) = o;
      if (dxb && c < ld_dxb)
        *reinterpret_cast<u32x2*>(dxb + row * ld_dxb + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
  }
  // block reduce over the 4 waves, one wave at a time
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        f32x4 a = ag[i], b = ab[i];
        if (w > 0) { a += *reinterpret_cast<f32x4*>(&red[0][c]); b += *reinterpret_cast<f32x4*>(&red[1][c]); }
        *reinterpret_cast<f32x4*>(&red[0][c]) = a; *reinterpret_cast<f32x4*>(&red[1][c]) = b;
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < d; c += 256) {
    partial[(int64_t)blockIdx.x * d + c] = red[0][c];
    partial[((int64_t)nblk + blockIdx.x) * d + c] = red[1][c];
  }
}

// 64 columns x 4 row groups per block
// grid (ceil(d/64), 2): blockIdx.y = 0 sums the dgamma partials, 1 the dbeta partials; 16 waves take interleaved
// partial rows with four independent accumulators each (the 2 x nblk x d fp32 partials are L2-resident)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int d, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int acc) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const float* p = partial + (int64_t)blockIdx.y * nblk * d + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < d) {
    int k = ry;
    for (; k + 48 < nblk; k += 64) {
      a0 += p[(int64_t)k * d]; a1 += p[(int64_t)(k + 16) * d]; a2 += p[(int64_t)(k + 32) * d]; a3 += p[(int64_t)(k + 48) * d];
    }
    for (; k < nblk; k += 16) a0 += p[(int64_t)k * d];
  }
  red[ry][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ry == 0 && c < d) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][cx];
    float* o = (blockIdx.y ? dbeta : dgamma) + c;      // one writer per element: accumulating stays deterministic
    *o = acc ? *o + t : t;
  }
}

static int ln_bwd_impl(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx, const float* gamma,
                       const float* mean, const float* rstd, const float* dx_in, float* dx_out, int64_t lddx, void* dx_bf16,
                       int64_t ld_dxb, float* dgamma, float* dbeta, float* partial, int64_t rows, int64_t d, void* stream,
                       bool acc = false) {
  VITA_REQUIRE(dy && x && gamma && mean && rstd && dx_out && dgamma && dbeta && partial);
  VITA_REQUIRE(rows > 0 && d > 0 && d <= LN_MAXV * 256 && lddx <= LN_MAXV * 256);
  VITA_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddx >= d && ldx >= ((d + 3) / 4) * 4 && lddy >= ((d + 3) / 4) * 4);
  if (dx_bf16) VITA_REQUIRE(ld_dxb % 4 == 0 && ld_dxb <= lddx);
  const int nblk = (int)vita_layernorm_bwd_nblk(rows);
  const int64_t width = lddx > d ? lddx : d;
#define LN_BWD_LAUNCH(NV)                                                                                                   \
  hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, lddy, dy_dtype == VITA_DT_F32, x, \
                     ldx, gamma, mean, rstd, dx_in, dx_out, lddx, (uint16_t*)dx_bf16, ld_dxb, partial, rows, (int)d)
  if (width <= 256) LN_BWD_LAUNCH(1);
  else if (width <= 512) LN_BWD_LAUNCH(2);
  else if (width <= 1280) LN_BWD_LAUNCH(5);
  else if (width <= 1536) LN_BWD_LAUNCH(6);
  else LN_BWD_LAUNCH(8);
#undef LN_BWD_LAUNCH
  VITA_CHECK_LAUNCH("vita_layernorm_bwd");
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((d + 63) / 64), 2), dim3(1024), 0, (hipStream_t)stream, partial,
                     nblk, (int)d, dgamma, dbeta, acc ? 1 : 0);
  VITA_CHECK_LAUNCH("vita_layernorm_bwd_reduce");
  return VITA_OK;
}
extern "C" int vita_layernorm_bwd(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                                  const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                  float* dx_out, int64_t lddx, void* dx_bf16, int64_t ld_dxb, float* dgamma,
                                  float* dbeta, float* partial, int64_t rows, int64_t d, void* stream) {
  return ln_bwd_impl(dy, lddy, dy_dtype, x, ldx, gamma, mean, rstd, dx_in, dx_out, lddx, dx_bf16, ld_dxb, dgamma, dbeta, partial,
                     rows, d, stream);
}
// the same, with the row sums ADDED to dgamma / dbeta (gradients accumulated straight into an optimizer's buffer)
extern "C" int vita_layernorm_bwd_acc(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                                      const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                      float* dx_out, int64_t lddx, void* dx_bf16, int64_t ld_dxb, float* dgamma,
                                      float* dbeta, float* partial, int64_t rows, int64_t d, void* stream) {
  return ln_bwd_impl(dy, lddy, dy_dtype, x, ldx, gamma, mean, rstd, dx_in, dx_out, lddx, dx_bf16, ld_dxb, dgamma, dbeta, partial,
                     rows, d, stream, true);
}

// The same backward for callers that cannot allocate: the dgamma / dbeta partials ([2][nblk][d] fp32, every element the
// reduce kernel reads is written by the first kernel of the same call, so nothing needs clearing) live in a file-scope
// device array instead of a caller-provided buffer.  Each GPU holds its own copy of the array (12 MiB, until the process
// ends); its device address is looked up once per device, with that device current, and only that address - never the
// array's name, which on the host is a stand-in and not a device address - is passed to the kernels.  Every call on
// one GPU shares that GPU's array: calls that overlap on one GPU (two streams) race.
#define LN_NOALLOC_MAX_D 2048
#define LN_SCRATCH_FLOATS (2 * LN_BWD_MAXBLK * LN_NOALLOC_MAX_D)
static __device__ __align__(256) float ln_bwd_scratch[LN_SCRATCH_FLOATS];
static float* ln_scratch_ptr[VITA_MAX_DEVICES];
extern "C" int vita_layernorm_bwd_noalloc(const void* dy, int64_t lddy, int32_t dy_dtype, const float* x, int64_t ldx,
                                          const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                          float* dx_out, int64_t lddx, void* dx_bf16, int64_t ld_dxb, float* dgamma,
                                          float* dbeta, int64_t rows, int64_t d, void* stream) {
  if (rows <= 0 || d <= 0) {
    vita_set_error("vita_layernorm_bwd_noalloc: rows and d must be positive", hipSuccess);
    return VITA_ERR_INVALID_ARG;
  }
  if (d > LN_NOALLOC_MAX_D || 2 * vita_layernorm_bwd_nblk(rows) * d > LN_SCRATCH_FLOATS) {
    vita_set_error("vita_layernorm_bwd_noalloc: d exceeds the built-in scratch (d <= 2048); use vita_layernorm_bwd with a caller buffer",
                   hipSuccess);
    return VITA_ERR_UNSUPPORTED;
  }
  // The symbol's address is resolved on the FIRST call on a device (that call may also make the runtime load the code
  // object: make it once outside graph capture); a failed lookup is not latched, the next call tries again.
  const int dev = vita_cur_device();
  if (dev < 0 || dev >= VITA_MAX_DEVICES) {
    vita_set_error("vita_layernorm_bwd_noalloc: device index out of range", hipSuccess);
    return VITA_ERR_INVALID_ARG;
  }
  float* scratch_ptr = __atomic_load_n(&ln_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!scratch_ptr) {
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(ln_bwd_scratch));
    if (e != hipSuccess || !p) {
      vita_set_error("vita_layernorm_bwd_noalloc: hipGetSymbolAddress", e);
      return VITA_ERR_LAUNCH;
    }
    scratch_ptr = (float*)p;          // every thread that races here resolves the same per-device address
    __atomic_store_n(&ln_scratch_ptr[dev], scratch_ptr, __ATOMIC_RELEASE);
  }
  return ln_bwd_impl(dy, lddy, dy_dtype, x, ldx, gamma, mean, rstd, dx_in, dx_out, lddx, dx_bf16, ld_dxb, dgamma, dbeta,
                     scratch_ptr, rows, d, stream);
}

// ------------------------------------------------------------------------------------- argsort + mask
// rank[i] = #{j : key_j < key_i  or (key_j == key_i and j < i)}  == ids_restore[i]; ids_shuffle[rank[i]] = i.
// Keys are the order-preserving uint32 image of the float noise; exact and stable by construction.
__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// A block of 16 waves ranks 256 keys: the four waves of quarter q count over the q-th quarter of the row (every lane of
// a wave reads the same LDS address: a broadcast), the four partial counts meet in LDS.  The row is brought into LDS
// once per block with 16-B loads; 4 waves per SIMD hide the LDS latency of this VALU-bound O(l^2) count.
#define ARGSORT_THREADS 1024
__global__ __launch_bounds__(ARGSORT_THREADS) void argsort_rank_kernel(const float* __restrict__ noise, int l, int len_keep,
                                                                       int64_t* __restrict__ ids_shuffle,
                                                                       int64_t* __restrict__ ids_restore, float* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int part[4][256];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);
  const int row = blockIdx.y;
  const float* nr = noise + (int64_t)row * l;
  const int lp = (l + 3) & ~3;
  if ((l & 3) == 0 && (((uintptr_t)nr) & 15) == 0) {
    for (int j4 = threadIdx.x; j4 < (l >> 2); j4 += ARGSORT_THREADS) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(nr + 4 * j4);
      *reinterpret_cast<u32x4*>(keys + 4 * j4) = u32x4{float_key(v[0]), float_key(v[1]), float_key(v[2]), float_key(v[3])};
    }
  } else {
    for (int j = threadIdx.x; j < lp; j += ARGSORT_THREADS) keys[j] = j < l ? float_key(nr[j]) : 0xffffffffu;
  }
  __syncthreads();
  const int il = threadIdx.x & 255, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  const int i0 = blockIdx.x * 256, i = i0 + il;
  const uint32_t ki = i < l ? keys[i] : 0u;
  // this quarter [q0, q1) of the row, in units of 4 keys
  const int per = ((lp >> 2) + 3) / 4;
  int q0 = q * per * 4, q1 = q0 + per * 4; if (q1 > lp) q1 = lp; if (q0 > lp) q0 = lp;
  // [q0, jb): j < every i of this block, [je, q1): j > every i of this block, [jb, je): the block's own keys
  int jb = i0, je = i0 + 256;
  if (jb < q0) jb = q0; if (jb > q1) jb = q1;
  if (je < jb) je = jb; if (je > q1) je = q1;
  int cnt = 0;
  for (int j = q0; j < jb; j += 4) {
    const u32x4 k4 = *reinterpret_cast<const u32x4*>(keys + j);
    cnt += (k4[0] <= ki) + (k4[1] <= ki) + (k4[2] <= ki) + (k4[3] <= ki);
  }
  for (int j = jb; j < je; ++j) { const uint32_t kj = keys[j]; cnt += (kj < ki) || (kj == ki && j < i); }
  for (int j = je; j < q1; j += 4) {
    const u32x4 k4 = *reinterpret_cast<const u32x4*>(keys + j);
    cnt += (k4[0] < ki) + (k4[1] < ki) + (k4[2] < ki) + (k4[3] < ki);
  }
  part[q][il] = cnt;
  __syncthreads();
  if (q == 0 && i < l) {
    cnt = part[0][il] + part[1][il] + part[2][il] + part[3][il];
    ids_restore[(int64_t)row * l + i] = cnt;
    ids_shuffle[(int64_t)row * l + cnt] = i;
    mask[(int64_t)row * l + i] = cnt < len_keep ? 0.f : 1.f;
  }
}

extern "C" int vita_argsort_mask(const float* noise, int64_t n, int64_t l, int64_t len_keep, int64_t* ids_shuffle,
                                 int64_t* ids_restore, float* mask, void* stream) {
  VITA_REQUIRE(noise && ids_shuffle && ids_restore && mask);
  VITA_REQUIRE(n > 0 && n < 65536 && l > 0 && l <= 32768 && len_keep >= 0 && len_keep <= l);
  const size_t lds = (size_t)((l + 3) & ~3) * 4;
  VITA_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)argsort_rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipLaunchKernelGGL(argsort_rank_kernel, dim3((unsigned)((l + 255) / 256), (unsigned)n), dim3(ARGSORT_THREADS), lds, (hipStream_t)stream,
                     noise, (int)l, (int)len_keep, ids_shuffle, ids_restore, mask);
  VITA_CHECK_LAUNCH("vita_argsort_mask");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- gather / scatter
__global__ __launch_bounds__(256) void row_gather_kernel(const float* __restrict__ x, int64_t ldx, int64_t l_in,
                                                         const int64_t* __restrict__ ids, int64_t ld_ids,
                                                         const float* __restrict__ fill, const float* __restrict__ pos,
                                                         float* __restrict__ out, int64_t ldo, int64_t l_out, int d) {
  const int64_t j = blockIdx.x, b = blockIdx.y;
  const int64_t src = ids[b * ld_ids + j];
  float* o = out + (b * l_out + j) * ldo;
  const float* pe = pos ? pos + j * (int64_t)d : nullptr;
  if (src < l_in) {
    const float* s = x + (b * l_in + src) * ldx;
    for (int c = threadIdx.x * 4; c < ldo; c += 1024) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) { v = *reinterpret_cast<const f32x4*>(s + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) { if (c + r >= d) v[r] = 0.f; else if (pe) v[r] += pe[c + r]; } }
      *reinterpret_cast<f32x4*>(o + c) = v;
    }
  } else {
    for (int c = threadIdx.x * 4; c < ldo; c += 1024) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (c + r < d) ? (fill ? fill[c + r] : 0.f) + (pe ? pe[c + r] : 0.f) : 0.f;
      *reinterpret_cast<f32x4*>(o + c) = v;
    }
  }
}
extern "C" int vita_row_gather_f32(const float* x, int64_t ldx, int64_t l_in, const int64_t* ids, int64_t ld_ids,
                                   const float* fill, const float* pos, float* out, int64_t ldo, int64_t l_out,
                                   int64_t nb, int64_t d, void* stream) {
  VITA_REQUIRE(x && ids && out && nb > 0 && nb < 65536 && l_out > 0 && d > 0);
  VITA_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= ((d + 3) / 4) * 4 && ldo >= d);
  VITA_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0);
  hipLaunchKernelGGL(row_gather_kernel, dim3((unsigned)l_out, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     l_in, ids, ld_ids, fill, pos, out, ldo, l_out, (int)d);
  VITA_CHECK_LAUNCH("vita_row_gather_f32");
  return VITA_OK;
}

// One wave per row (16 waves per block, grid-stride over the rows of batch entry blockIdx.y): rows that map into dx are
// copied with 16-B vectors; rows past l_in (the decoder's mask tokens) are summed per lane-owned column in registers,
// reduced over the block's waves in LDS and leave ONE global atomic per column and block - a per-row atomicAdd would
// serialise ~16k updates on each of the d addresses.
#define SCATTER_NV 8     // float4 per lane -> d <= 2048
__global__ __launch_bounds__(1024) void row_scatter_kernel(const float* __restrict__ dout, int64_t ldo, int64_t l_out,
                                                           const int64_t* __restrict__ ids, int64_t ld_ids,
                                                           float* __restrict__ dx, int64_t ldx, int64_t l_in,
                                                           float* __restrict__ dfill, int d) {
  __shared__ float red[SCATTER_NV * 256];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = blockIdx.y;
  f32x4 fill[SCATTER_NV];
#pragma unroll
  for (int i = 0; i < SCATTER_NV; ++i) fill[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int c = threadIdx.x; c < SCATTER_NV * 256; c += 1024) red[c] = 0.f;
  for (int64_t j = (int64_t)blockIdx.x * 16 + wave; j < l_out; j += (int64_t)gridDim.x * 16) {
    const int64_t dst = ids[b * ld_ids + j];
    const float* s = dout + (b * l_out + j) * ldo;
    float* o = dst < l_in ? dx + (b * l_in + dst) * ldx : nullptr;
#pragma unroll
    for (int i = 0; i < SCATTER_NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= ldx) continue;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < d) {
        v = *reinterpret_cast<const f32x4*>(s + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c + r >= d) v[r] = 0.f;
      }
      if (o) *reinterpret_cast<f32x4*>(o + c) = v;
      else fill[i] += v;
    }
  }
  if (dfill == nullptr) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < SCATTER_NV; ++i) {
    const int c = (i * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) if (c + r < d && fill[i][r] != 0.f) atomicAdd(&red[c + r], fill[i][r]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 1024) if (red[c] != 0.f) atomicAdd(dfill + c, red[c]);
}
extern "C" int vita_row_scatter_f32(const float* dout, int64_t ldo, int64_t l_out, const int64_t* ids, int64_t ld_ids,
                                    float* dx, int64_t ldx, int64_t l_in, float* dfill, int64_t nb, int64_t d,
                                    void* stream) {
  VITA_REQUIRE(dout && ids && dx && nb > 0 && nb < 65536 && l_out > 0 && d > 0);
  VITA_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldo >= ((d + 3) / 4) * 4 && ldx >= d && ldx <= SCATTER_NV * 256);
  VITA_REQUIRE(((uintptr_t)dx & 15) == 0 && ((uintptr_t)dout & 15) == 0);
  int64_t chunks = (512 + nb - 1) / nb;                 // about 512 blocks of 16 waves
  if (chunks > (l_out + 15) / 16) chunks = (l_out + 15) / 16;
  hipLaunchKernelGGL(row_scatter_kernel, dim3((unsigned)chunks, (unsigned)nb), dim3(1024), 0, (hipStream_t)stream, dout, ldo,
                     l_out, ids, ld_ids, dx, ldx, l_in, dfill, (int)d);
  VITA_CHECK_LAUNCH("vita_row_scatter_f32");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- im2col + assemble
// one wave per token; unit = 4 consecutive pixels along W (pw % 4 == 0, W % 4 == 0)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, int S, int T, int H, int W, int pt,
                                                     int ph, int pw, const int64_t* __restrict__ ids, int64_t n_tok,
                                                     uint16_t* __restrict__ out, int64_t ldo, int64_t total,
                                                     int64_t out_batch_rows, int64_t out_row0) {
  const int lane = threadIdx.x & 63;
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= total) return;
  const int64_t b = tok / n_tok, j = tok % n_tok;
  const int64_t id = ids ? ids[b * n_tok + j] : j;
  const int t = T / pt, h = H / ph, w = W / pw;
  const int wi = (int)(id % w); int64_t r = id / w;
  const int hi = (int)(r % h); r /= h;
  const int ti = (int)(r % t); const int s = (int)(r / t);
  const float* base = img + (((b * S + s) * T + (int64_t)ti * pt) * H + (int64_t)hi * ph) * W + (int64_t)wi * pw;
  const int P = pt * ph * pw, upr = pw >> 2;   // units per pixel row
  uint16_t* o = out + (b * out_batch_rows + out_row0 + j) * ldo;
  for (int u = lane; u * 4 < ldo; u += 64) {
    const int e = u * 4;
    u32x2 pk = u32x2{0u, 0u};
    if (e < P) {
      const int rowi = u / upr, cu = u % upr;     // rowi = p_t*ph + p_h
      const int p_t = rowi / ph, p_h = rowi % ph;
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((int64_t)p_t * H + p_h) * W + cu * 4);
      pk = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    *reinterpret_cast<u32x2*>(o + e) = pk;
  }
}
extern "C" int vita_im2col_bf16(const float* img, int64_t nb, int64_t S, int64_t T, int64_t H, int64_t W, int64_t pt,
                                int64_t ph, int64_t pw, const int64_t* ids, int64_t n_tok, void* out, int64_t ldo,
                                int64_t out_batch_rows, int64_t out_row0, void* stream) {
  VITA_REQUIRE(img && out && nb > 0 && n_tok > 0);
  VITA_REQUIRE(pw % 4 == 0 && W % 4 == 0 && T % pt == 0 && H % ph == 0 && W % pw == 0);
  VITA_REQUIRE(out_batch_rows >= out_row0 + n_tok && out_row0 >= 0);
  VITA_REQUIRE(ldo % 8 == 0 && ldo >= pt * ph * pw && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 15) == 0);
  const int64_t total = nb * n_tok;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, img, (int)S,
                     (int)T, (int)H, (int)W, (int)pt, (int)ph, (int)pw, ids, n_tok, (uint16_t*)out, ldo, total, out_batch_rows, out_row0);
  VITA_CHECK_LAUNCH("vita_im2col_bf16");
  return VITA_OK;
}

__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ y, int64_t ldy, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const int64_t* __restrict__ ids,
                                                       int64_t n_tok, int d, float* __restrict__ out, int64_t ldo,
                                                       int64_t y_batch_rows) {
  const int64_t r = blockIdx.x, b = blockIdx.y;   // r in [0, 1+n_tok)
  float* o = out + (b * (n_tok + 1) + r) * ldo;
  const float* src; const float* pe = nullptr;
  if (r == 0) { src = cls; if (pos) pe = pos; }
  else {
    src = y + (b * y_batch_rows + (r - 1)) * ldy;
    if (pos) pe = pos + (1 + (ids ? ids[b * n_tok + r - 1] : r - 1)) * (int64_t)d;
  }
  for (int c = threadIdx.x; c < ldo; c += 256) o[c] = c < d ? src[c] + (pe ? pe[c] : 0.f) : 0.f;
}
extern "C" int vita_tokens_assemble(const float* y, int64_t ldy, const float* cls, const float* pos, const int64_t* ids,
                                    int64_t nb, int64_t n_tok, int64_t d, float* out, int64_t ldo, int64_t y_batch_rows,
                                    void* stream) {
  VITA_REQUIRE(y && cls && out && nb > 0 && nb < 65536 && n_tok > 0 && d > 0 && ldo >= d && ldy >= d);
  hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)(n_tok + 1), (unsigned)nb), dim3(256), 0, (hipStream_t)stream, y, ldy,
                     cls, pos, ids, n_tok, (int)d, out, ldo, y_batch_rows);
  VITA_CHECK_LAUNCH("vita_tokens_assemble");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- cast / colsum
// four independent 16-B loads in flight per thread (grid-stride over 4-column units)
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ y,
                                                   int64_t ldy, int64_t rows, int d) {
  const int64_t units = ldy >> 2, total = rows * units, stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
    f32x4 v[4];
    int64_t r[4]; int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u * stride;
      r[u] = i / units; c[u] = (int)(i % units) * 4;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < total && c[u] < d) v[u] = *reinterpret_cast<const f32x4*>(x + r[u] * ldx + c[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * stride >= total) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (c[u] + k >= d) v[u][k] = 0.f;
      *reinterpret_cast<u32x2*>(y + r[u] * ldy + c[u]) = u32x2{pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], v[u][3])};
    }
  }
}
extern "C" int vita_cast_f32_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int64_t d, void* stream) {
  VITA_REQUIRE(x && y && rows > 0 && d > 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= d && ldx >= ((d + 3) / 4) * 4);
  int64_t nb = (rows * (ldy >> 2) + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(cast_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ldx, (uint16_t*)y, ldy, rows, (int)d);
  VITA_CHECK_LAUNCH("vita_cast_f32_bf16");
  return VITA_OK;
}

// Column sums (bias gradients).  A block is 16 column lanes x 16 row lanes: a column lane owns one 16-byte unit (8 bf16 /
// 4 fp32 columns), so a block covers 128 / 64 columns and a wave-load fetches 4 rows x 256 contiguous bytes.  Each thread
// keeps 8 independent 16-B loads in flight over the rows r0 + rl + 16 i of the block's row chunk (the kernel is latency-
// bound: a narrow matrix gives few blocks, so the bytes in flight per thread are what reaches the HBM rate), the 16 row
// lanes are reduced through LDS and every column ends in ONE fp32 atomic per block.
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int64_t ld, int64_t rows, int64_t cols,
                                                     float* __restrict__ out, int64_t rows_per_blk) {
  constexpr int CPT = F32 ? 4 : 8, U = 8;
  __shared__ float red[16][16 * CPT + 4];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = ((int64_t)blockIdx.x * 16 + cl) * CPT;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk; if (r1 > rows) r1 = rows;
  float acc[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
  auto add = [&](const u32x4 v) {
    if constexpr (F32) {
      const f32x4 vf = __builtin_bit_cast(f32x4, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += vf[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __builtin_bit_cast(float, v[j] << 16);
        acc[2 * j + 1] += __builtin_bit_cast(float, v[j] & 0xffff0000u);
      }
    }
  };
  if (c < cols) {
    const char* base = (const char*)x + c * (F32 ? 4 : 2);
    const int64_t rb = ld * (F32 ? 4 : 2);
    int64_t r = r0 + rl;
    for (; r + 16 * (U - 1) < r1; r += 16 * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + (r + 16 * u) * rb));
#pragma unroll
      for (int u = 0; u < U; ++u) add(v[u]);
    }
    for (; r < r1; r += 16) add(*reinterpret_cast<const u32x4*>(base + r * rb));
  }
#pragma unroll
  for (int j = 0; j < CPT; ++j) red[rl][cl * CPT + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 16 * CPT) {
    const int64_t col = (int64_t)blockIdx.x * 16 * CPT + threadIdx.x;
    if (col < cols) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
      atomicAdd(out + col, t);
    }
  }
}
static int colsum_impl(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, bool zero, void* stream);
extern "C" int vita_colsum(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, void* stream) {
  return colsum_impl(x, dtype, ld, rows, cols, out, true, stream);
}
extern "C" int vita_colsum_acc(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, void* stream) {
  return colsum_impl(x, dtype, ld, rows, cols, out, false, stream);
}
static int colsum_impl(const void* x, int32_t dtype, int64_t ld, int64_t rows, int64_t cols, float* out, bool zero, void* stream) {
  const bool f32 = dtype == VITA_DT_F32;
  const int cpt = f32 ? 4 : 8;
  VITA_REQUIRE(x && out && rows > 0 && cols > 0 && ld % cpt == 0 && ld >= ((cols + cpt - 1) / cpt) * cpt);
  VITA_REQUIRE(((uintptr_t)x & 15) == 0);
  if (zero && hipMemsetAsync(out, 0, cols * sizeof(float), (hipStream_t)stream) != hipSuccess) return VITA_ERR_LAUNCH;
  const int64_t colblk = (cols + 16 * cpt - 1) / (16 * cpt);
  // about 2048 blocks in all (8 per CU), at least 128 rows per block (one batch of 8 loads per thread), at most 256 row
  // chunks: every chunk ends in one fp32 atomic per column
  int64_t chunks = (2048 + colblk - 1) / colblk;
  if (chunks > 256) chunks = 256;
  if (chunks > (rows + 127) / 128) chunks = (rows + 127) / 128;
  if (chunks < 1) chunks = 1;
  const int64_t rpb = (rows + chunks - 1) / chunks;
  dim3 grid((unsigned)colblk, (unsigned)chunks);
  if (f32) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out, rpb);
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, out, rpb);
  VITA_CHECK_LAUNCH("vita_colsum");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- softmax rows (materialised attention)
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds_, uint16_t* __restrict__ p,
                                                           int64_t ldp, int n, float scale) {
  __shared__ float red[4];
  const float* sr = s + (int64_t)blockIdx.x * lds_;
  uint16_t* pr = p + (int64_t)blockIdx.x * ldp;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, sr[j]);
  m = block_reduce(m, true, red);
  float z = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) z += __expf((sr[j] - m) * scale);
  z = block_reduce(z, false, red);
  const float inv = 1.f / z;
  for (int j = threadIdx.x; j < ldp; j += 256) pr[j] = j < n ? f32_to_bf16_bits(__expf((sr[j] - m) * scale) * inv) : (uint16_t)0;
}
extern "C" int vita_softmax_rows(const float* s, int64_t lds_, void* p, int64_t ldp, int64_t rows, int64_t n, float scale,
                                 void* stream) {
  VITA_REQUIRE(s && p && rows > 0 && n > 0 && ldp >= n && lds_ >= n && rows < (1ll << 31));
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, lds_, (uint16_t*)p, ldp,
                     (int)n, scale);
  VITA_CHECK_LAUNCH("vita_softmax_rows");
  return VITA_OK;
}
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const uint16_t* __restrict__ p, int64_t ldp,
                                                               const float* __restrict__ dp, int64_t lddp,
                                                               uint16_t* __restrict__ ds, int64_t ldds, int n, float scale) {
  __shared__ float red[4];
  const uint16_t* pr = p + (int64_t)blockIdx.x * ldp;
  const float* dpr = dp + (int64_t)blockIdx.x * lddp;
  uint16_t* dsr = ds + (int64_t)blockIdx.x * ldds;
  float dot = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) dot += bf16_bits_to_f32(pr[j]) * dpr[j];
  dot = block_reduce(dot, false, red);
  for (int j = threadIdx.x; j < ldds; j += 256)
    dsr[j] = j < n ? f32_to_bf16_bits(bf16_bits_to_f32(pr[j]) * (dpr[j] - dot) * scale) : (uint16_t)0;
}
extern "C" int vita_softmax_bwd_rows(const void* p, int64_t ldp, const float* dp, int64_t lddp, void* ds, int64_t ldds,
                                     int64_t rows, int64_t n, float scale, void* stream) {
  VITA_REQUIRE(p && dp && ds && rows > 0 && n > 0 && ldp >= n && lddp >= n && ldds >= n && rows < (1ll << 31));
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)p, ldp,
                     dp, lddp, (uint16_t*)ds, ldds, (int)n, scale);
  VITA_CHECK_LAUNCH("vita_softmax_bwd_rows");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- weight shadows
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int64_t rows_log, int64_t cols_log,
                                                          uint16_t* __restrict__ out, int64_t rows_phys, int64_t ld_phys,
                                                          int rsp, int rsl, int csp, int csl) {
  const int64_t total = rows_phys * ld_phys;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ld_phys, c = i % ld_phys;
    const int64_t lr = seg_map(r, rsp, rsl), lc = seg_map(c, csp, csl);
    float v = 0.f;
    if (lr >= 0 && lr < rows_log && lc >= 0 && lc < cols_log) v = w[lr * cols_log + lc];
    out[i] = f32_to_bf16_bits(v);
  }
}
extern "C" int vita_pack_weight_bf16(const float* w, int64_t rows_log, int64_t cols_log, void* out, int64_t rows_phys,
                                     int64_t ld_phys, int32_t rsp, int32_t rsl, int32_t csp, int32_t csl, void* stream) {
  VITA_REQUIRE(w && out && rows_log > 0 && cols_log > 0 && rows_phys > 0 && ld_phys > 0);
  int64_t nb = (rows_phys * ld_phys + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, rows_log, cols_log,
                     (uint16_t*)out, rows_phys, ld_phys, rsp, rsl, csp, csl);
  VITA_CHECK_LAUNCH("vita_pack_weight_bf16");
  return VITA_OK;
}
__global__ void pack_vec_kernel(const float* __restrict__ v, int64_t n_log, float* __restrict__ out, int64_t n_phys, int sp, int sl) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_phys) return;
  const int64_t li = seg_map(i, sp, sl);
  out[i] = (li >= 0 && li < n_log) ? v[li] : 0.f;
}
extern "C" int vita_pack_vec_f32(const float* v, int64_t n_log, float* out, int64_t n_phys, int32_t sp, int32_t sl, void* stream) {
  VITA_REQUIRE(v && out && n_log > 0 && n_phys > 0);
  hipLaunchKernelGGL(pack_vec_kernel, dim3((unsigned)((n_phys + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, n_log, out, n_phys, sp, sl);
  VITA_CHECK_LAUNCH("vita_pack_vec_f32");
  return VITA_OK;
}
__global__ void unpack_vec_kernel(const float* __restrict__ v, int64_t n_phys, float* __restrict__ out, int64_t n_log, int sp, int sl) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_phys) return;
  const int64_t li = seg_map(i, sp, sl);
  if (li >= 0 && li < n_log) out[li] = v[i];
}
extern "C" int vita_unpack_vec_f32(const float* v, int64_t n_phys, float* out, int64_t n_log, int32_t sp, int32_t sl, void* stream) {
  VITA_REQUIRE(v && out && n_log > 0 && n_phys > 0);
  hipLaunchKernelGGL(unpack_vec_kernel, dim3((unsigned)((n_phys + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, n_phys, out, n_log, sp, sl);
  VITA_CHECK_LAUNCH("vita_unpack_vec_f32");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float gscale, float bc1, float bc2_sqrt) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float gr = gv[r] * gscale + wd * pv[r];
        mv[r] = b1 * mv[r] + (1.f - b1) * gr;
        vv[r] = b2 * vv[r] + (1.f - b2) * gr * gr;
        pv[r] -= (lr / bc1) * mv[r] / (sqrtf(vv[r]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<f32x4*>(p + i) = pv; *reinterpret_cast<f32x4*>(m + i) = mv; *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (int64_t k = i; k < n; ++k) {
        float gr = g[k] * gscale + wd * p[k];
        m[k] = b1 * m[k] + (1.f - b1) * gr;
        v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
        p[k] -= (lr / bc1) * m[k] / (sqrtf(v[k]) / bc2_sqrt + eps);
      }
    }
  }
}
extern "C" int vita_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float grad_scale, int64_t step, void* stream) {
  VITA_REQUIRE(p && g && m && v && n > 0 && step >= 1);
  VITA_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  int64_t nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, grad_scale, bc1, sqrtf(bc2));
  VITA_CHECK_LAUNCH("vita_adam_step");
  return VITA_OK;
}
