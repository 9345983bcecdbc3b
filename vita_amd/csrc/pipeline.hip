// Input-pipeline contract of the hot path: the per-subject cine preparation that the reference does
// on the host in Cardiac3DplusTAllAX.__getitem__ (datasets/datasets.py:403-427) with load_im_seg_arr's x=y mirror
// (datasets/datasets.py:169-171) and image_normalization (utils/general.py:139-147), as two HBM-bound kernels:
//
//   stored  lax fp32 [nb][A0][A1][S_lax][T], sax fp32 [nb][A0][A1][S_tot][T]   (npz layout, T fastest)
//   volume  = concat(lax, sax[..., z0 : z0+S_sax, :]) on the slice axis          (datasets.py:414)
//   lo, hi  = min / max over the whole volume of one subject                     (general.py:142-143)
//   out[b][s][k][i][j] = (volume[j][i][s][frames[k]] - lo) / (hi - lo)           (mirror + transpose(2,3,0,1) + frame pick)
//
// Pass 1 (cine_minmax_kernel) reads every element once and leaves per-workgroup partial extrema; pass 2
// (cine_prepare_kernel) reduces the partials (no atomics, no pre-initialised buffer), stages one output row's
// [A][T] slab through LDS so that both the 4*T-byte source chunks and the output rows are read / written as whole
// contiguous pieces, and normalises with the same IEEE fp32 subtract / divide numpy performs: results are bit-exact.
#include "common.h"

#define CINE_NBLK 128      // partial extrema per subject
#define CINE_THREADS 256

struct CineGeom {
  const float* lax;
  const float* sax;
  int64_t lax_subject, sax_subject;     // elements per subject in each stored array
  int A0, A1, s_lax, s_tot, z0, s_sax, T;
};

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(CINE_THREADS) void cine_minmax_kernel(const CineGeom g, float* __restrict__ partial) {
  const int b = blockIdx.y;
  const float* lax = g.lax + (int64_t)b * g.lax_subject;
  const float* sax = g.sax + (int64_t)b * g.sax_subject;
  const int64_t npix = (int64_t)g.A0 * g.A1;
  const int nl = g.s_lax * g.T, ns = g.s_sax * g.T;     // contiguous run per pixel in each array
  float lo = INFINITY, hi = -INFINITY;
  // lax: fully contiguous
  for (int64_t i = (int64_t)blockIdx.x * CINE_THREADS + threadIdx.x; i < npix * nl; i += (int64_t)CINE_NBLK * CINE_THREADS) {
    const float v = lax[i];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  // sax: run of ns elements per pixel starting at slice z0
  for (int64_t i = (int64_t)blockIdx.x * CINE_THREADS + threadIdx.x; i < npix * ns; i += (int64_t)CINE_NBLK * CINE_THREADS) {
    const int64_t pix = i / ns;
    const int r = (int)(i - pix * ns);
    const float v = sax[(pix * g.s_tot + g.z0) * g.T + r];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  lo = wave_min(lo); hi = wave_max(hi);
  __shared__ float red[2][CINE_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = lo; red[1][wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < CINE_THREADS / 64; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
    partial[((int64_t)b * CINE_NBLK + blockIdx.x) * 2] = lo;
    partial[((int64_t)b * CINE_NBLK + blockIdx.x) * 2 + 1] = hi;
  }
}

// grid (A_row, S, nb): one output row family out[b][s][:][i][:]
__global__ __launch_bounds__(CINE_THREADS) void cine_prepare_kernel(const CineGeom g, const int* __restrict__ frames, int n_frames,
                                                                    int mirror, const float* __restrict__ partial,
                                                                    float* __restrict__ minmax, float* __restrict__ out) {
  extern __shared__ float slab[];      // [ncol][TP], TP odd: column-strided reads are bank-conflict free
  __shared__ float lohi[2];
  const int i = blockIdx.x, s = blockIdx.y, b = blockIdx.z;
  const int T = g.T, TP = T | 1;
  const int nrow = mirror ? g.A1 : g.A0, ncol = mirror ? g.A0 : g.A1;
  // extrema of this subject
  if (threadIdx.x < 64) {
    float lo = INFINITY, hi = -INFINITY;
    for (int k = threadIdx.x; k < CINE_NBLK; k += 64) {
      lo = fminf(lo, partial[((int64_t)b * CINE_NBLK + k) * 2]);
      hi = fmaxf(hi, partial[((int64_t)b * CINE_NBLK + k) * 2 + 1]);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (threadIdx.x == 0) {
      lohi[0] = lo; lohi[1] = hi;
      if (i == 0 && s == 0 && minmax) { minmax[2 * b] = lo; minmax[2 * b + 1] = hi; }
    }
  }
  // source: slice s of the concatenated volume
  const bool from_lax = s < g.s_lax;
  const float* src = from_lax ? g.lax + (int64_t)b * g.lax_subject : g.sax + (int64_t)b * g.sax_subject;
  const int ssrc = from_lax ? s : g.z0 + s - g.s_lax, S = from_lax ? g.s_lax : g.s_tot;
  // stored element (a0, a1, ssrc, t) lives at ((a0*A1 + a1)*S + ssrc)*T + t; output row i / column j = (a1, a0) when mirrored
  const int64_t row_off = mirror ? (int64_t)i * S : (int64_t)i * g.A1 * S;
  const int64_t col_stride = mirror ? (int64_t)g.A1 * S : S;
  for (int e = threadIdx.x; e < ncol * T; e += CINE_THREADS) {
    const int j = e / T, t = e - j * T;
    slab[j * TP + t] = src[(row_off + (int64_t)j * col_stride + ssrc) * T + t];
  }
  __syncthreads();
  const float lo = lohi[0], den = lohi[1] - lohi[0];
  float* orow = out + (((int64_t)b * (g.s_lax + g.s_sax) + s) * n_frames * nrow + i) * ncol;
  for (int e = threadIdx.x; e < n_frames * ncol; e += CINE_THREADS) {
    const int k = e / ncol, j = e - k * ncol;
    orow[(int64_t)k * nrow * ncol + j] = (slab[j * TP + frames[k]] - lo) / den;
  }
}

extern "C" int64_t vita_cine_partial_floats(int64_t nb) { return nb * CINE_NBLK * 2; }

extern "C" int vita_cine_prepare(const float* lax, const float* sax, int64_t nb, int64_t a0, int64_t a1, int64_t s_lax,
                                 int64_t s_sax_total, int64_t z0, int64_t s_sax, int64_t T, const int32_t* frames,
                                 int64_t n_frames, int32_t mirror, float* partial, float* minmax, float* out, void* stream) {
  VITA_REQUIRE(lax && sax && frames && partial && out);
  VITA_REQUIRE(nb > 0 && nb < 65536 && a0 > 0 && a1 > 0 && a0 < 32768 && a1 < 32768 && T > 0 && T < 4096);
  VITA_REQUIRE(s_lax > 0 && s_sax > 0 && z0 >= 0 && z0 + s_sax <= s_sax_total && s_lax + s_sax < 65536 && n_frames > 0);
  VITA_REQUIRE(a0 * a1 * s_sax_total * T < (int64_t)1 << 31 && a0 * a1 * s_lax * T < (int64_t)1 << 31);
  const int64_t ncol = mirror ? a0 : a1;
  const size_t lds = (size_t)ncol * (T | 1) * 4;
  VITA_REQUIRE(lds <= 64 * 1024);
  CineGeom g;
  g.lax = lax; g.sax = sax;
  g.lax_subject = a0 * a1 * s_lax * T; g.sax_subject = a0 * a1 * s_sax_total * T;
  g.A0 = (int)a0; g.A1 = (int)a1; g.s_lax = (int)s_lax; g.s_tot = (int)s_sax_total; g.z0 = (int)z0; g.s_sax = (int)s_sax; g.T = (int)T;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cine_minmax_kernel, dim3(CINE_NBLK, (unsigned)nb), dim3(CINE_THREADS), 0, s, g, partial);
  VITA_CHECK_LAUNCH("vita_cine_prepare(minmax)");
  const int64_t nrow = mirror ? a1 : a0;
  hipLaunchKernelGGL(cine_prepare_kernel, dim3((unsigned)nrow, (unsigned)(s_lax + s_sax), (unsigned)nb), dim3(CINE_THREADS), lds, s,
                     g, (const int*)frames, (int)n_frames, (int)mirror, (const float*)partial, minmax, out);
  VITA_CHECK_LAUNCH("vita_cine_prepare");
  return VITA_OK;
}
