// Stage-II glue kernels (gfx950): the tabular tower's 117 per-feature tokenisers in one launch, the mean
// over all tokens that feeds the projection heads, and BatchNorm1d (batch statistics) + ReLU of the SimCLR heads.
// All fp32, HBM / latency-bound; the two Linear layers of a head run on vita_sgemm_f32.
#include "common.h"

// ------------------------------------------------------------------------------------- tabular tokenisers
// out[b][0]     = cls + pos[0]
// out[b][1 + f] = (n_class[f] == 0 ? x[b][f] * w_f + b_f : table_f[(long)x[b][f]]) + pos[1 + f]      (pads zero)
// The per-feature parameters are separate tensors (numerical_fe.i.{weight,bias}, single_c_fe.i.weight,
// multi_c_fe.i.weight - the reference's state_dict): the kernel gets their device addresses in two tables.
__global__ __launch_bounds__(256) void tab_embed_fwd_kernel(const float* __restrict__ x, int F, const int64_t* __restrict__ wptr,
                                                            const int64_t* __restrict__ bptr, const int32_t* __restrict__ n_class,
                                                            const float* __restrict__ cls, const float* __restrict__ pos,
                                                            float* __restrict__ out, int64_t ldo, int D) {
  const int f = blockIdx.x;            // 0 = cls row, 1 + feature otherwise
  const int64_t b = blockIdx.y;
  float* o = out + (b * (F + 1) + f) * ldo;
  const float* pr = pos ? pos + (int64_t)f * D : nullptr;
  const float* w;
  const float* bias = nullptr;
  float xv = 1.f;
  if (f == 0) {
    w = cls;
  } else {
    const int nc = n_class[f - 1];
    const float v = x[b * F + (f - 1)];
    w = reinterpret_cast<const float*>(wptr[f - 1]);
    if (nc == 0) {
      xv = v;
      bias = reinterpret_cast<const float*>(bptr[f - 1]);
    } else {
      int64_t idx = (int64_t)v;                      // x.to(torch.long): truncation
      idx = idx < 0 ? 0 : (idx >= nc ? nc - 1 : idx);  // nn.Embedding raises on an index out of range; the kernel clamps
      w += idx * D;
    }
  }
  for (int c = threadIdx.x; c < ldo; c += 256) {
    float v = 0.f;
    if (c < D) {
      v = w[c] * xv;
      if (bias) v += bias[c];
      if (pr) v += pr[c];
    }
    o[c] = v;
  }
}
extern "C" int vita_tabular_embed_fwd(const float* x, int64_t B, int64_t F, const int64_t* wptr, const int64_t* bptr,
                                      const int32_t* n_class, const float* cls, const float* pos, float* out, int64_t ldo,
                                      int64_t D, void* stream) {
  VITA_REQUIRE(x && wptr && bptr && n_class && cls && out && B > 0 && B < 65536 && F > 0 && D > 0 && ldo >= D);
  hipLaunchKernelGGL(tab_embed_fwd_kernel, dim3((unsigned)(F + 1), (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, (int)F, wptr,
                     bptr, n_class, cls, pos, out, ldo, (int)D);
  VITA_CHECK_LAUNCH("vita_tabular_embed_fwd");
  return VITA_OK;
}
// Backward: every parameter row has ONE owner thread per column, which sums over the batch and ADDS the sum to the
// gradient row (plain read-modify-write, no atomics, fixed order: deterministic).
//   numerical f: dw_f[c] += sum_b x[b][f] dout[b][1+f][c],  db_f[c] += sum_b dout[b][1+f][c]
//   table f    : dtable_f[k][c] += sum_{b: idx(b) == k} dout[b][1+f][c]
//   cls        : dcls[c] += sum_b dout[b][0][c]
// gw / gb: device addresses of the gradient tensors (same shapes as the parameters).
__global__ __launch_bounds__(256) void tab_embed_bwd_kernel(const float* __restrict__ dout, int64_t ldo, const float* __restrict__ x,
                                                            int64_t B, int F, const int32_t* __restrict__ n_class,
                                                            const int64_t* __restrict__ gw, const int64_t* __restrict__ gb,
                                                            float* __restrict__ dcls, int D) {
  const int f = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= D) return;
  const float* d0 = dout + (int64_t)f * ldo + c;
  const int64_t bs = (int64_t)(F + 1) * ldo;
  if (f == 0) {
    if (!dcls) return;
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) s += d0[b * bs];
    dcls[c] += s;
    return;
  }
  const int nc = n_class[f - 1];
  float* w = reinterpret_cast<float*>(gw[f - 1]);
  if (nc == 0) {
    float sw = 0.f, sb = 0.f;
    for (int64_t b = 0; b < B; ++b) {
      const float g = d0[b * bs];
      sw += g * x[b * F + (f - 1)];
      sb += g;
    }
    if (w) w[c] += sw;
    float* bg = reinterpret_cast<float*>(gb[f - 1]);
    if (bg) bg[c] += sb;
    return;
  }
  if (!w) return;
  for (int k = 0; k < nc; ++k) {
    float s = 0.f;
    for (int64_t b = 0; b < B; ++b) {
      int64_t idx = (int64_t)x[b * F + (f - 1)];
      idx = idx < 0 ? 0 : (idx >= nc ? nc - 1 : idx);
      if (idx == k) s += d0[b * bs];
    }
    w[(int64_t)k * D + c] += s;
  }
}
extern "C" int vita_tabular_embed_bwd(const float* dout, int64_t ldo, const float* x, int64_t B, int64_t F, const int32_t* n_class,
                                      const int64_t* gw, const int64_t* gb, float* dcls, int64_t D, void* stream) {
  VITA_REQUIRE(dout && x && n_class && gw && gb && B > 0 && F > 0 && D > 0 && ldo >= D);
  hipLaunchKernelGGL(tab_embed_bwd_kernel, dim3((unsigned)(F + 1), (unsigned)((D + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dout, ldo, x, B, (int)F, n_class, gw, gb, dcls, (int)D);
  VITA_CHECK_LAUNCH("vita_tabular_embed_bwd");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- mean over tokens
// out[b][c] = (1/N) sum_n x[b][n][c]   (y.mean(dim=1), models/MultimodalSimCLR.py:121,133).  A block = 16 column lanes
// (one 16-B unit each) x 16 row lanes over ALL rows of one subject, then an LDS reduce over the row lanes: one writer per
// output element and a fixed summation order, so the result is bit-reproducible (round 2 split the rows over several
// blocks that met in fp32 atomics - run-to-run different sums).  A latency-sized op: Stage II reads 32 x 1153 rows.
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, int64_t ld, int64_t N, int D,
                                                         float* __restrict__ out, float inv_n) {
  __shared__ float red[16][68];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cl) * 4;
  const int64_t b = blockIdx.y;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  if (c < D) {     // ld is a multiple of 4 and columns [D, ld) are zero pads: the 16-B unit that straddles D is safe to read
    const float* base = x + b * N * ld + c;
    int64_t r = rl;
    for (; r + 16 < N; r += 32) {       // two independent loads in flight per thread
      acc0 += *reinterpret_cast<const f32x4*>(base + r * ld);
      acc1 += *reinterpret_cast<const f32x4*>(base + (r + 16) * ld);
    }
    if (r < N) acc0 += *reinterpret_cast<const f32x4*>(base + r * ld);
    acc0 += acc1;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[rl][cl * 4 + j] = acc0[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col < D) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
      out[b * D + col] = t * inv_n;
    }
  }
}
extern "C" int vita_token_mean_fwd(const float* x, int64_t ld, int64_t B, int64_t N, int64_t D, float* out, void* stream) {
  VITA_REQUIRE(x && out && B > 0 && B < 65536 && N > 0 && D > 0 && ld % 4 == 0 && ld >= ((D + 3) / 4) * 4);
  VITA_REQUIRE(((uintptr_t)x & 15) == 0);
  const int64_t colblk = (D + 63) / 64;
  hipLaunchKernelGGL(token_mean_kernel, dim3((unsigned)colblk, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ld, N, (int)D, out,
                     1.0f / (float)N);
  VITA_CHECK_LAUNCH("vita_token_mean_fwd");
  return VITA_OK;
}
// backward: dx[b][n][c] = dy[b][c] / N for c < D, 0 in the pads
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dy, int64_t N, int D, float* __restrict__ dx,
                                                             int64_t ld, int64_t total4, float inv_n) {
  const int64_t l4 = ld >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / l4;
    const int c = (int)(i % l4) * 4;
    const int64_t b = row / N;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = c + j < D ? dy[b * D + c + j] * inv_n : 0.f;
    *reinterpret_cast<f32x4*>(dx + row * ld + c) = v;
  }
}
extern "C" int vita_token_mean_bwd(const float* dy, int64_t B, int64_t N, int64_t D, float* dx, int64_t ld, void* stream) {
  VITA_REQUIRE(dy && dx && B > 0 && N > 0 && D > 0 && ld % 4 == 0 && ld >= D && ((uintptr_t)dx & 15) == 0);
  const int64_t total4 = B * N * (ld >> 2);
  int64_t nb = (total4 + 255) / 256; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(token_mean_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, N, (int)D, dx, ld, total4,
                     1.0f / (float)N);
  VITA_CHECK_LAUNCH("vita_token_mean_bwd");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- BatchNorm1d (+ ReLU)
// nn.BatchNorm1d in training mode over x [B][D]: per-column batch mean and biased variance (two passes, as torch),
// y = (x - mean) * rstd * gamma + beta, optional ReLU; running statistics updated with `momentum` and the UNBIASED
// variance (torch semantics).  A block = 64 columns x 4 row lanes; the row lanes meet in LDS.
__global__ __launch_bounds__(256) void bn_fwd_kernel(const float* __restrict__ x, int64_t B, int D, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int relu, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var, float momentum) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < D;
  float s = 0.f;
  if (on) for (int64_t r = rl; r < B; r += 4) s += x[r * D + c];
  red[rl][cl] = s;
  __syncthreads();
  const float mean = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)B;
  __syncthreads();
  float q = 0.f;
  if (on) for (int64_t r = rl; r < B; r += 4) { const float d = x[r * D + c] - mean; q += d * d; }
  red[rl][cl] = q;
  __syncthreads();
  const float var = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)B;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!on) return;
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  for (int64_t r = rl; r < B; r += 4) {
    float v = (x[r * D + c] - mean) * rstd * g + bt;
    if (relu) v = fmaxf(v, 0.f);
    y[r * D + c] = v;
  }
  if (rl == 0) {
    mean_out[c] = mean; rstd_out[c] = rstd;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (B > 1 ? (float)B / (float)(B - 1) : 1.f);
  }
}
extern "C" int vita_batchnorm_fwd(const float* x, int64_t B, int64_t D, const float* gamma, const float* beta, float eps,
                                  int32_t relu, float* y, float* mean, float* rstd, float* running_mean, float* running_var,
                                  float momentum, void* stream) {
  VITA_REQUIRE(x && y && mean && rstd && B > 0 && D > 0);
  hipLaunchKernelGGL(bn_fwd_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, B, (int)D, gamma, beta,
                     eps, relu, y, mean, rstd, running_mean, running_var, momentum);
  VITA_CHECK_LAUNCH("vita_batchnorm_fwd");
  return VITA_OK;
}
// eval mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta, optional ReLU (elementwise)
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, int64_t total, int D, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int relu, const float* __restrict__ rm,
                                                      const float* __restrict__ rv, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % D);
    float v = (x[i] - rm[c]) / sqrtf(rv[c] + eps) * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}
extern "C" int vita_batchnorm_eval(const float* x, int64_t B, int64_t D, const float* gamma, const float* beta, float eps,
                                   int32_t relu, const float* running_mean, const float* running_var, float* y, void* stream) {
  VITA_REQUIRE(x && y && running_mean && running_var && B > 0 && D > 0);
  int64_t nb = (B * D + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(bn_eval_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, B * D, (int)D, gamma, beta, eps, relu,
                     running_mean, running_var, y);
  VITA_CHECK_LAUNCH("vita_batchnorm_eval");
  return VITA_OK;
}
// backward (training mode).  y: the forward output (its sign is the ReLU mask when relu != 0).
//   dz = dy * [y > 0];  dbeta = sum dz;  dgamma = sum dz * xhat;  dx = gamma * rstd * (dz - dbeta / B - xhat * dgamma / B)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                     int64_t B, int D, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int relu, float* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < D;
  const float mu = on ? mean[c] : 0.f, rs = on ? rstd[c] : 0.f;
  float sb = 0.f, sg = 0.f;
  if (on)
    for (int64_t r = rl; r < B; r += 4) {
      float dz = dy[r * D + c];
      if (relu && !(y[r * D + c] > 0.f)) dz = 0.f;
      sb += dz;
      sg += dz * (x[r * D + c] - mu) * rs;
    }
  red[0][rl][cl] = sb; red[1][rl][cl] = sg;
  __syncthreads();
  if (!on) return;
  const float db = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  const float dg = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  const float g = gamma ? gamma[c] : 1.f, invb = 1.f / (float)B;
  for (int64_t r = rl; r < B; r += 4) {
    float dz = dy[r * D + c];
    if (relu && !(y[r * D + c] > 0.f)) dz = 0.f;
    const float xh = (x[r * D + c] - mu) * rs;
    dx[r * D + c] = g * rs * (dz - db * invb - xh * dg * invb);
  }
  if (rl == 0) {
    if (dgamma) dgamma[c] = dg;
    if (dbeta) dbeta[c] = db;
  }
}
extern "C" int vita_batchnorm_bwd(const float* dy, const float* x, const float* y, int64_t B, int64_t D, const float* gamma,
                                  const float* mean, const float* rstd, int32_t relu, float* dx, float* dgamma, float* dbeta,
                                  void* stream) {
  VITA_REQUIRE(dy && x && y && mean && rstd && dx && B > 0 && D > 0);
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dy, x, y, B, (int)D, gamma,
                     mean, rstd, relu, dx, dgamma, dbeta);
  VITA_CHECK_LAUNCH("vita_batchnorm_bwd");
  return VITA_OK;
}
// out[r][c] += bias[c]  /  column sums into a bias gradient are vita_colsum; this adds a bias after vita_sgemm_f32
__global__ __launch_bounds__(256) void add_bias_kernel(float* __restrict__ out, int64_t total, int D, const float* __restrict__ bias) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] += bias[i % D];
}
extern "C" int vita_add_bias_rows(float* out, int64_t rows, int64_t D, const float* bias, void* stream) {
  VITA_REQUIRE(out && bias && rows > 0 && D > 0);
  int64_t nb = (rows * D + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(add_bias_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, out, rows * D, (int)D, bias);
  VITA_CHECK_LAUNCH("vita_add_bias_rows");
  return VITA_OK;
}
// out[c] = sum_r x[r][c] for a small dense fp32 matrix [rows][D] with any D (no alignment rules): the bias gradient of a
// projection-head Linear.  64 columns x 4 row lanes per block, fixed summation order.
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ x, int64_t rows, int D, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < D) for (int64_t r = rl; r < rows; r += 4) s += x[r * D + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < D) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
extern "C" int vita_colsum_small_f32(const float* x, int64_t rows, int64_t D, float* out, void* stream) {
  VITA_REQUIRE(x && out && rows > 0 && D > 0);
  hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, rows, (int)D, out);
  VITA_CHECK_LAUNCH("vita_colsum_small_f32");
  return VITA_OK;
}
