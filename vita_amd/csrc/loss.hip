// Loss kernels (gfx950): Stage-II CLIP / InfoNCE pieces in fp32 (L2 normalise, small fp32 GEMM,
// row-wise cross entropy with wavefront logsumexp) and the Stage-I sigmoid + MSE against the
// patchified image.
#include "common.h"

// ------------------------------------------------------------------------------------- L2 normalise
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t rows, int d, float eps,
                                                         float* __restrict__ y, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) { const float v = x[row * d + c]; s += v * v; }
  const float nrm = sqrtf(wave_sum(s));
  const float inv = 1.f / fmaxf(nrm, eps);
  if (lane == 0) inv_norm[row] = nrm > eps ? inv : -inv;   // sign flags the clamped branch for bwd
  for (int c = lane; c < d; c += 64) y[row * d + c] = x[row * d + c] * inv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv_norm, int64_t rows, int d,
                                                         float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float inv = inv_norm[row];
  const bool clamped = inv < 0.f; inv = fabsf(inv);
  float dot = 0.f;
  for (int c = lane; c < d; c += 64) dot += dy[row * d + c] * y[row * d + c];
  dot = clamped ? 0.f : wave_sum(dot);
  for (int c = lane; c < d; c += 64) dx[row * d + c] = inv * (dy[row * d + c] - y[row * d + c] * dot);
}
extern "C" int vita_l2_normalize_fwd(const float* x, int64_t rows, int64_t d, float eps, float* y, float* inv_norm, void* stream) {
  VITA_REQUIRE(x && y && inv_norm && rows > 0 && d > 0);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, rows, (int)d, eps, y, inv_norm);
  VITA_CHECK_LAUNCH("vita_l2_normalize_fwd");
  return VITA_OK;
}
extern "C" int vita_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int64_t d, float* dx, void* stream) {
  VITA_REQUIRE(dy && y && inv_norm && dx && rows > 0 && d > 0);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, inv_norm, rows, (int)d, dx);
  VITA_CHECK_LAUNCH("vita_l2_normalize_bwd");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- small fp32 GEMM
// C[m][n] = alpha * sum_k A(m,k) B(k,n) + beta * C ; layouts as vita_gemm_bf16.  (16R)x(16R)x16 tile, RxR per thread:
// R = 4 for large outputs; R = 2 / 1 when the output is small and the reduction long (projection heads: 32 x 1025 x 1025
// is 17 workgroups of 65 serial K-steps with R = 4, 79 us; 130 workgroups with R = 1), so that the grid covers the chip.
template <int R>
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int m, int n, int k, int64_t lda, int64_t ldb, int64_t ldc, int la, int lb,
                                                    float alpha, float beta) {
  constexpr int T = 16 * R;
  __shared__ float As[16][T + 1], Bs[16][T + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  float acc[R][R] = {};
  for (int k0 = 0; k0 < k; k0 += 16) {
    for (int e = threadIdx.x; e < 16 * T; e += 256) {
      int kk, f;
      if (la == 0) { kk = e & 15; f = e >> 4; } else { f = e % T; kk = e / T; }
      float v = 0.f;
      if (m0 + f < m && k0 + kk < k) v = la == 0 ? A[(int64_t)(m0 + f) * lda + k0 + kk] : A[(int64_t)(k0 + kk) * lda + m0 + f];
      As[kk][f] = v;
      if (lb == 0) { kk = e & 15; f = e >> 4; } else { f = e % T; kk = e / T; }
      v = 0.f;
      if (n0 + f < n && k0 + kk < k) v = lb == 0 ? B[(int64_t)(n0 + f) * ldb + k0 + kk] : B[(int64_t)(k0 + kk) * ldb + n0 + f];
      Bs[kk][f] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[R], b[R];
#pragma unroll
      for (int i = 0; i < R; ++i) { a[i] = As[kk][ty * R + i]; b[i] = Bs[kk][tx * R + i]; }
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int r = m0 + ty * R + i, c = n0 + tx * R + j;
      if (r < m && c < n) {
        float* p = C + (int64_t)r * ldc + c;
        *p = alpha * acc[i][j] + (beta != 0.f ? beta * *p : 0.f);
      }
    }
}
extern "C" int vita_sgemm_f32(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, int64_t lda,
                              int64_t ldb, int64_t ldc, int32_t a_layout, int32_t b_layout, float alpha, float beta,
                              void* stream) {
  VITA_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0 && m < (1 << 22) && n < (1 << 22));
  // the largest tile whose grid still has one workgroup per CU; the accumulation order over k is the same for every R
  // (one thread owns an output element and walks k in order), so the result does not depend on the choice
  const int64_t wg4 = ((n + 63) / 64) * ((m + 63) / 64), wg2 = ((n + 31) / 32) * ((m + 31) / 32);
#define SGEMM_LAUNCH(R)                                                                                                        \
  hipLaunchKernelGGL(sgemm_kernel<R>, dim3((unsigned)((n + 16 * R - 1) / (16 * R)), (unsigned)((m + 16 * R - 1) / (16 * R))),   \
                     dim3(256), 0, (hipStream_t)stream, a, b, c, (int)m, (int)n, (int)k, lda, ldb, ldc, a_layout, b_layout,    \
                     alpha, beta)
  if (wg4 >= 256) SGEMM_LAUNCH(4);
  else if (wg2 >= 256) SGEMM_LAUNCH(2);
  else SGEMM_LAUNCH(1);
#undef SGEMM_LAUNCH
  VITA_CHECK_LAUNCH("vita_sgemm_f32");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- row-wise cross entropy
// one wave per row: lse = logsumexp(row); loss += w*(lse - row[label]); dlogits = w*(softmax(row) - onehot)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int64_t ld, int64_t rows, int cols,
                                                      int64_t label_offset, float w, float* __restrict__ loss,
                                                      float* __restrict__ row_lse, float* __restrict__ dlogits, int64_t ldd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* lr = logits + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, lr[c]);
  m = wave_max(m);
  float z = 0.f;
  for (int c = lane; c < cols; c += 64) z += expf(lr[c] - m);
  z = wave_sum(z);
  const float lse = m + logf(z);
  const int64_t label = label_offset + row;
  if (lane == 0) {
    if (row_lse) row_lse[row] = lse;
    atomicAdd(loss, w * (lse - lr[label]));
  }
  if (dlogits)
    for (int c = lane; c < cols; c += 64) dlogits[row * ldd + c] = w * (expf(lr[c] - lse) - (c == label ? 1.f : 0.f));
}
extern "C" int vita_ce_rows(const float* logits, int64_t ld, int64_t rows, int64_t cols, int64_t label_offset, float weight,
                            float* loss, float* row_lse, float* dlogits, int64_t ldd, void* stream) {
  VITA_REQUIRE(logits && loss && rows > 0 && cols > 0 && ld >= cols && label_offset >= 0 && label_offset + rows <= cols);
  if (dlogits) VITA_REQUIRE(ldd >= cols);
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, rows, (int)cols,
                     label_offset, weight, loss, row_lse, dlogits, ldd);
  VITA_CHECK_LAUNCH("vita_ce_rows");
  return VITA_OK;
}

// ------------------------------------------------------------------------------------- sigmoid + MSE vs patchified image
// one wave per row, a block walks rows blockIdx.x*4 + wave, + 4*gridDim.x, ...; the loss leaves one atomic per block
__global__ __launch_bounds__(256) void sigmoid_mse_patch_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ img,
                                                                int S, int T, int H, int W, int pt, int ph, int pw, int64_t L,
                                                                const float* __restrict__ mask, float inv_count, float* __restrict__ pred,
                                                                float* __restrict__ loss, uint16_t* __restrict__ dz, int64_t ld_dz,
                                                                int64_t total_rows) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = pt * ph * pw;
  const int t = T / pt, h = H / ph, w = W / pw;
  const int lim = dz ? (int)ld_dz : P;
  float acc = 0.f;
  for (int64_t rowi = (int64_t)blockIdx.x * 4 + wave; rowi < total_rows; rowi += (int64_t)gridDim.x * 4) {   // row in [0, nb*(1+L))
    const int64_t b = rowi / (L + 1), r = rowi % (L + 1);
    if (r == 0) {   // cls row: no loss, zero gradient
      if (dz) for (int e = lane; e < ld_dz; e += 64) dz[rowi * ld_dz + e] = 0;
      continue;
    }
    const int64_t id = r - 1;
    const int wi = (int)(id % w); int64_t q = id / w;
    const int hi = (int)(q % h); q /= h;
    const int ti = (int)(q % t); const int s = (int)(q / t);
    const float* base = img + (((b * S + s) * T + (int64_t)ti * pt) * H + (int64_t)hi * ph) * W + (int64_t)wi * pw;
    const float wrow = mask ? mask[b * L + id] : 1.f;
    const float* zr = z + rowi * ldz;
    for (int e = lane; e < lim; e += 64) {
      float g = 0.f;
      if (e < P) {
        const int p_w = e % pw, p_h = (e / pw) % ph, p_t = e / (pw * ph);
        const float tgt = base[((int64_t)p_t * H + p_h) * W + p_w];
        const float pr = 1.f / (1.f + __expf(-zr[e]));
        if (pred) pred[(b * L + id) * P + e] = pr;
        const float diff = pr - tgt;
        acc += diff * diff * wrow;
        g = 2.f * diff * pr * (1.f - pr) * wrow * inv_count;
      }
      if (dz) dz[rowi * ld_dz + e] = f32_to_bf16_bits(g);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    if (tot != 0.f) atomicAdd(loss, tot * inv_count);
  }
}
extern "C" int vita_sigmoid_mse_patch(const float* z, int64_t ldz, const float* img, int64_t nb, int64_t S, int64_t T, int64_t H,
                                      int64_t W, int64_t pt, int64_t ph, int64_t pw, const float* mask, float inv_count,
                                      float* pred, float* loss, void* dz_bf16, int64_t ld_dz, void* stream) {
  VITA_REQUIRE(z && img && loss && nb > 0 && T % pt == 0 && H % ph == 0 && W % pw == 0);
  const int64_t L = S * (T / pt) * (H / ph) * (W / pw), P = pt * ph * pw;
  VITA_REQUIRE(ldz >= P);
  if (dz_bf16) VITA_REQUIRE(ld_dz >= P);
  const int64_t total = nb * (L + 1);
  const int64_t nblk = (total + 3) / 4 < 2048 ? (total + 3) / 4 : 2048;
  hipLaunchKernelGGL(sigmoid_mse_patch_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, z, ldz, img, (int)S,
                     (int)T, (int)H, (int)W, (int)pt, (int)ph, (int)pw, L, mask, inv_count, pred, loss, (uint16_t*)dz_bf16, ld_dz, total);
  VITA_CHECK_LAUNCH("vita_sigmoid_mse_patch");
  return VITA_OK;
}
