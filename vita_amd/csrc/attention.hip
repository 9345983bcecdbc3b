// Fused (flash-style) multi-head attention for gfx950, bf16 MFMA with fp32 online softmax.
// Replaces the materialised q@k^T -> softmax -> @v of timm 0.6.12 Attention.forward (SURVEY
// Appendix A; call sites networks/imaging_encoders.py:36-40, networks/imaging_decoders.py:43-45).
//
// Layout: qkv bf16 [B][N][3][H][HDP] (row stride 3*H*HDP, head dim zero-padded to HDP = 32k),
//         o   bf16 [B][N][H][HDP], lse fp32 [B][H][N] (natural log).
//
// Forward: one workgroup = 4 waves = 128 query rows of one (b,h); each wave owns 32 rows.
//   S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (key on the accumulator rows, query on the lane), so a
//   lane holds 16 scores of ONE query row: the row max / sum need one cross-half exchange only.
//   P^T (bf16, straight from the S^T accumulator registers) is the B operand of O^T = V^T P^T;
//   the V^T fragments come from ds_read_b64_tr_b16 (hardware transpose) of the row-major V tile.
//   K tile rows are padded by 16 B (odd slot count) -> conflict-free ds_read_b128; V tile rows are
//   512 B with the 64-B chunk index XOR (row & 3) -> conflict-free transposed reads.
#include "common.h"

#define ATT_KT 64        // keys per LDS tile
#define ATT_QW 32        // query rows per wave

template <int HDP> struct AttCfg {
  static constexpr int KS = HDP * 2 + 16;                                  // K row stride (bytes)
  static constexpr int VS = (HDP * 2 <= 256) ? 256 : ((HDP * 2 <= 512) ? 512 : 1024);  // V row stride (bytes)
  static constexpr int K_BYTES = ATT_KT * KS;
  static constexpr int V_BYTES = ATT_KT * VS;
  static constexpr int CPR = HDP / 8;                                       // 16-B chunks per row
};

__device__ __forceinline__ uint32_t v_off(int row, int col, int VS) {
  const int byte = col * 2;
  return row * VS + ((((byte >> 6) ^ (row & 3))) << 6) + (byte & 63);
}

__device__ __forceinline__ bf16x8 tr_pair(const char* base, uint32_t o0, uint32_t o1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const float* p) {
  u32x4 u = {pack_bf16x2(p[0], p[1]), pack_bf16x2(p[2], p[3]), pack_bf16x2(p[4], p[5]), pack_bf16x2(p[6], p[7])};
  return __builtin_bit_cast(bf16x8, u);
}

// stage `rows` x HDP bf16 (row-major, row stride ld elements) into an LDS image; rows beyond the
// buffer extent come back as zeros from the bounds check
template <int HDP, bool VIMG, int BATCH = 4>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, int64_t row0, int64_t ld, char* lds, int tid) {
  constexpr int CPR = HDP / 8, TOTAL = ATT_KT * CPR, PER = (TOTAL + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < PER; i0 += BATCH) {
    u32x4 reg[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int e = tid + (i0 + i) * 256;
      const int r = e / CPR, c = e % CPR;
      reg[i] = u32x4{0, 0, 0, 0};
      if (i0 + i < PER && e < TOTAL) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(((row0 + r) * ld + c * 8) * 2), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int e = tid + (i0 + i) * 256;
      const int r = e / CPR, c = e % CPR;
      if (i0 + i < PER && e < TOTAL) {
        const uint32_t off = VIMG ? v_off(r, c * 8, AttCfg<HDP>::VS) : (uint32_t)(r * AttCfg<HDP>::KS + c * 16);
        *reinterpret_cast<u32x4*>(lds + off) = reg[i];
      }
    }
  }
}

template <int HDP, int NKS>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                          float* __restrict__ lse, int N, int H, float c_exp2) {
  using Cfg = AttCfg<HDP>;
  constexpr int NDT = HDP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsK = smem;
  char* ldsV = smem + Cfg::K_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int64_t ld = 3ll * H * HDP;
  const int q0 = blockIdx.x * 128 + wave * ATT_QW;
  const uint16_t* base = qkv + (int64_t)b * N * ld;
  const uint32_t extent = (uint32_t)((((int64_t)N - 1) * ld + HDP) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + h * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (H + h) * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (2 * H + h) * HDP), 0, extent, 0x00020000);

  const int lq = lane & 31, hh = lane >> 5;
  bf16x8 qf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsQ, (uint32_t)((((int64_t)q0 + lq) * ld + 16 * s + 8 * hh) * 2), 0, 0);
    qf[s] = __builtin_bit_cast(bf16x8, v);
  }
  f32x16 oacc[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // transposed-read lane constants
  const int gi = lane >> 4, t = lane & 15, tq = t >> 2, tp = t & 3;
  const int d_in_tile = 16 * (gi & 1) + 4 * tp;   // column offset inside a 32-wide d tile

  const int ntiles = (N + ATT_KT - 1) / ATT_KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    stage_tile<HDP, false>(rsK, (int64_t)kt * ATT_KT, ld, ldsK, tid);
    stage_tile<HDP, true>(rsV, (int64_t)kt * ATT_KT, ld, ldsV, tid);
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kbase = kt * ATT_KT + sub * 32;
      if (kbase >= N) break;
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ldsK + (sub * 32 + lq) * Cfg::KS + (16 * s + 8 * hh) * 2);
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc, 0, 0, 0);
      }
      if (kbase + 32 > N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= N) sacc[r] = -INFINITY;
        }
      }
      float mx = sacc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = exp2f((m_run - m_new) * c_exp2);
      const float mc = m_new * c_exp2;
      float p[16];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { p[r] = exp2f(sacc[r] * c_exp2 - mc); rs += p[r]; }
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      const bf16x8 pf0 = pack8(p), pf1 = pack8(p + 8);
#pragma unroll
      for (int i = 0; i < NDT; ++i) {
        const int col = i * 32 + d_in_tile;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int r0 = sub * 32 + 16 * s2 + 4 * (gi >> 1) + tq;
          const bf16x8 vf = tr_pair(ldsV, v_off(r0, col, Cfg::VS), v_off(r0 + 8, col, Cfg::VS));
          oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, s2 == 0 ? pf0 : pf1, oacc[i], 0, 0, 0);
        }
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  const int q = q0 + lq;
  if (q < N) {
    uint16_t* orow = o + ((int64_t)b * N + q) * ((int64_t)H * HDP) + h * HDP;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int d = i * 32 + 8 * rr + 4 * hh;
        u32x2 u = {pack_bf16x2(oacc[i][4 * rr] * inv, oacc[i][4 * rr + 1] * inv),
                   pack_bf16x2(oacc[i][4 * rr + 2] * inv, oacc[i][4 * rr + 3] * inv)};
        *reinterpret_cast<u32x2*>(orow + d) = u;
      }
    if (hh == 0) lse[((int64_t)b * H + h) * N + q] = (m_run * c_exp2 + log2f(l_tot)) * 0.6931471805599453f;
  }
}

template <int HDP, int NKS>
static int launch_fwd(const void* qkv, void* o, float* lse, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
  using Cfg = AttCfg<HDP>;
  const size_t lds = Cfg::K_BYTES + Cfg::V_BYTES;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<HDP, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL((attn_fwd_kernel<HDP, NKS>), grid, dim3(256), lds, s, (const uint16_t*)qkv, (uint16_t*)o, lse, (int)N, (int)H,
                     scale * 1.4426950408889634f);
  VITA_CHECK_LAUNCH("vita_attn_fwd");
  return VITA_OK;
}

extern "C" int vita_attn_supported(int64_t hd, int64_t hdp) {
  const int64_t nks = (hd + 15) / 16;
  return (hdp == 224 && nks == 13) || (hdp == 64 && nks == 3) || (hdp == 32 && nks == 2) || (hdp == 128 && nks == 8);
}

extern "C" int vita_attn_fwd(const void* qkv, void* o, float* lse, int64_t B, int64_t N, int64_t H, int64_t hd, int64_t hdp,
                             float scale, void* stream) {
  VITA_REQUIRE(qkv && o && lse && B > 0 && N > 0 && H > 0 && B < 65536 && H < 65536);
  VITA_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)o & 7) == 0);
  VITA_REQUIRE(N * 3 * H * hdp * 2 < (int64_t)0xffffffffll);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nks = (hd + 15) / 16;
  if (hdp == 224 && nks == 13) return launch_fwd<224, 13>(qkv, o, lse, B, N, H, scale, s);
  if (hdp == 64 && nks == 3) return launch_fwd<64, 3>(qkv, o, lse, B, N, H, scale, s);
  if (hdp == 32 && nks == 2) return launch_fwd<32, 2>(qkv, o, lse, B, N, H, scale, s);
  if (hdp == 128 && nks == 8) return launch_fwd<128, 8>(qkv, o, lse, B, N, H, scale, s);
  vita_set_error("vita_attn_fwd: unsupported head dim", hipSuccess);
  return VITA_ERR_UNSUPPORTED;
}

// =============================================================================================
// Backward.  P is recomputed from Q, K and the forward's LSE; no N x N tensor is stored.
//   delta[q] = rowsum(dO * O)                                  (attn_delta_kernel)
//   dK, dV : one workgroup = 4 waves = 128 keys of one (b,h), key on the MFMA lane; each wave keeps
//            dK^T and dV^T of its 32 keys in accumulators while the workgroup sweeps 32-row query
//            tiles (Q and dO staged in LDS, double buffered, prefetched through registers).
//   dQ     : one workgroup = 128 query rows, query on the lane; sweeps 64-key K/V tiles.
// Both are deterministic (no atomics).  S and dP accumulators are initialised with -LSE/scale and
// -delta so p = exp2(c*S') and dS = p*dP' need no extra subtraction.
// One LDS image serves row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16):
//   off(row, byte) = row*RS + ((byte>>6 ^ row&3)<<6) + (((byte>>4)&3 ^ (row>>2)&3)<<4) + (byte&15)
template <int HDP> struct BwdCfg {
  static constexpr int RS = (HDP * 2 <= 256) ? 256 : ((HDP * 2 <= 512) ? 512 : 1024);
};
__device__ __forceinline__ uint32_t u_off(int row, int byte, int RS) {
  return row * RS + ((((byte >> 6) ^ (row & 3))) << 6) + (((((byte >> 4) & 3) ^ ((row >> 2) & 3))) << 4) + (byte & 15);
}

__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t* __restrict__ o, const uint16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t BN, int N, int H, int HDP) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // row in [0, B*N)
  if (row >= BN) return;
  const int64_t b = row / N, q = row % N;
  const int cpr = HDP / 4;
  for (int h = 0; h < H; ++h) {
    float acc = 0.f;
    for (int c = lane; c < cpr; c += 64) {
      const u32x2 a = *reinterpret_cast<const u32x2*>(o + (row * H + h) * HDP + c * 4);
      const u32x2 d = *reinterpret_cast<const u32x2*>(dout + (row * H + h) * HDP + c * 4);
      acc += bf16_bits_to_f32((uint16_t)(a[0] & 0xffff)) * bf16_bits_to_f32((uint16_t)(d[0] & 0xffff));
      acc += bf16_bits_to_f32((uint16_t)(a[0] >> 16)) * bf16_bits_to_f32((uint16_t)(d[0] >> 16));
      acc += bf16_bits_to_f32((uint16_t)(a[1] & 0xffff)) * bf16_bits_to_f32((uint16_t)(d[1] & 0xffff));
      acc += bf16_bits_to_f32((uint16_t)(a[1] >> 16)) * bf16_bits_to_f32((uint16_t)(d[1] >> 16));
    }
    acc = wave_sum(acc);
    if (lane == 0) delta[(b * H + h) * N + q] = acc;
  }
}

// stage ROWS x HDP bf16 into the unified image, through registers (load / store split)
template <int HDP, int ROWS>
struct UStager {
  static constexpr int CPR = HDP / 8, TOTAL = ROWS * CPR, PER = (TOTAL + 255) / 256;
  u32x4 reg[PER];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int64_t row0, int64_t ld, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * 256;
      const int r = e / CPR, c = e % CPR;
      reg[i] = u32x4{0, 0, 0, 0};
      if (e < TOTAL) reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(((row0 + r) * ld + c * 8) * 2), 0, 0);
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + i * 256;
      const int r = e / CPR, c = e % CPR;
      if (e < TOTAL) *reinterpret_cast<u32x4*>(lds + u_off(r, c * 16, BwdCfg<HDP>::RS)) = reg[i];
    }
  }
};

template <int HDP, int NKS>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               uint16_t* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int RS = BwdCfg<HDP>::RS, NDT = HDP / 32, QT = 32;
  constexpr int TILE = QT * RS;                 // one image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [buf][ Q image | dO image | lse[32] | delta[32] ]
  constexpr int BUF = 2 * TILE + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int64_t ld = 3ll * H * HDP, ldo = (int64_t)H * HDP;
  const uint16_t* base = qkv + (int64_t)b * N * ld;
  const uint32_t extent = (uint32_t)((((int64_t)N - 1) * ld + HDP) * 2);
  const uint32_t extent_o = (uint32_t)((((int64_t)N - 1) * ldo + HDP) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + h * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (H + h) * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (2 * H + h) * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDO = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dout + (int64_t)b * N * ldo + h * HDP), 0, extent_o, 0x00020000);
  const float* lse_bh = lse + ((int64_t)b * H + h) * N;
  const float* delta_bh = delta + ((int64_t)b * H + h) * N;

  const int lk = lane & 31, hh = lane >> 5;
  const int key = blockIdx.x * 128 + wave * 32 + lk;
  bf16x8 kf[NKS], vf[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    const uint32_t off = (uint32_t)(((int64_t)key * ld + 16 * s + 8 * hh) * 2);
    kf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsK, off, 0, 0));
    vf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsV, off, 0, 0));
  }
  f32x16 dk[NDT], dv[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  const float c_exp2 = scale * 1.4426950408889634f, inv_scale = 1.f / scale;
  const int gi = lane >> 4, t = lane & 15, tq = t >> 2, tp = t & 3;
  const int ntiles = (N + QT - 1) / QT;

  UStager<HDP, QT> sq, sdo;
  float r_ls = 0.f;
  auto load_tile = [&](int tile) {
    sq.load(rsQ, (int64_t)tile * QT, ld, tid);
    sdo.load(rsDO, (int64_t)tile * QT, ldo, tid);
    if (tid < 64) {
      const int qq = tile * QT + (tid & 31);
      r_ls = qq < N ? (tid < 32 ? lse_bh[qq] : delta_bh[qq]) : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    char* bp = smem + buf * BUF;
    sq.store(bp, tid);
    sdo.store(bp + TILE, tid);
    if (tid < 64) reinterpret_cast<float*>(bp + 2 * TILE)[tid] = r_ls;   // [0,32) lse, [32,64) delta
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int cur = tile & 1;
    if (tile + 1 < ntiles) load_tile(tile + 1);
    const char* qimg = smem + cur * BUF;
    const char* doimg = qimg + TILE;
    const float* ls = reinterpret_cast<const float*>(qimg + 2 * TILE);
    f32x16 sacc, dpacc;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(ls + 8 * rr + 4 * hh);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(ls + 32 + 8 * rr + 4 * hh);
#pragma unroll
      for (int k = 0; k < 4; ++k) { sacc[4 * rr + k] = -l4[k] * inv_scale; dpacc[4 * rr + k] = -d4[k]; }
    }
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qimg + u_off(lk, (16 * s + 8 * hh) * 2, RS));
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], sacc, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const bf16x8 da = *reinterpret_cast<const bf16x8*>(doimg + u_off(lk, (16 * s + 8 * hh) * 2, RS));
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[s], dpacc, 0, 0, 0);
    }
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { p[r] = exp2f(sacc[r] * c_exp2); ds[r] = p[r] * dpacc[r]; }
    const bf16x8 pf0 = pack8(p), pf1 = pack8(p + 8), df0 = pack8(ds), df1 = pack8(ds + 8);
#pragma unroll
    for (int i = 0; i < NDT; ++i) {
      const int byte = (i * 32 + 16 * (gi & 1) + 4 * tp) * 2;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int r0 = 16 * s2 + 4 * (gi >> 1) + tq;
        const uint32_t o0 = u_off(r0, byte, RS), o1 = u_off(r0 + 8, byte, RS);
        const bf16x8 doT = tr_pair(doimg, o0, o1);
        dv[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT, s2 == 0 ? pf0 : pf1, dv[i], 0, 0, 0);
        const bf16x8 qT = tr_pair(qimg, o0, o1);
        dk[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT, s2 == 0 ? df0 : df1, dk[i], 0, 0, 0);
      }
    }
    if (tile + 1 < ntiles) store_tile(cur ^ 1);
    __syncthreads();
  }
  if (key < N) {
    uint16_t* krow = dqkv + ((int64_t)b * N + key) * ld + (H + h) * HDP;
    uint16_t* vrow = dqkv + ((int64_t)b * N + key) * ld + (2 * H + h) * HDP;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int d = i * 32 + 8 * rr + 4 * hh;
        *reinterpret_cast<u32x2*>(krow + d) = u32x2{pack_bf16x2(dk[i][4 * rr] * scale, dk[i][4 * rr + 1] * scale),
                                                     pack_bf16x2(dk[i][4 * rr + 2] * scale, dk[i][4 * rr + 3] * scale)};
        *reinterpret_cast<u32x2*>(vrow + d) = u32x2{pack_bf16x2(dv[i][4 * rr], dv[i][4 * rr + 1]),
                                                     pack_bf16x2(dv[i][4 * rr + 2], dv[i][4 * rr + 3])};
      }
  }
}

template <int HDP, int NKS>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             uint16_t* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int RS = BwdCfg<HDP>::RS, NDT = HDP / 32, KT = 64;
  constexpr int TILE = KT * RS, BUF = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z;
  const int64_t ld = 3ll * H * HDP, ldo = (int64_t)H * HDP;
  const uint16_t* base = qkv + (int64_t)b * N * ld;
  const uint32_t extent = (uint32_t)((((int64_t)N - 1) * ld + HDP) * 2);
  const uint32_t extent_o = (uint32_t)((((int64_t)N - 1) * ldo + HDP) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + h * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (H + h) * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base + (2 * H + h) * HDP), 0, extent, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDO = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(dout + (int64_t)b * N * ldo + h * HDP), 0, extent_o, 0x00020000);

  const int lq = lane & 31, hh = lane >> 5;
  const int q = blockIdx.x * 128 + wave * 32 + lq;
  bf16x8 qf[NKS], dof[NKS];
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    qf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, (uint32_t)(((int64_t)q * ld + 16 * s + 8 * hh) * 2), 0, 0));
    dof[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsDO, (uint32_t)(((int64_t)q * ldo + 16 * s + 8 * hh) * 2), 0, 0));
  }
  const float c_exp2 = scale * 1.4426950408889634f;
  const float s_init = q < N ? -lse[((int64_t)b * H + h) * N + q] / scale : 0.f;
  const float d_init = q < N ? -delta[((int64_t)b * H + h) * N + q] : 0.f;
  f32x16 dq[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
  const int gi = lane >> 4, t = lane & 15, tq = t >> 2, tp = t & 3;
  const int ntiles = (N + KT - 1) / KT;

  UStager<HDP, KT> sk, sv;
  sk.load(rsK, 0, ld, tid); sv.load(rsV, 0, ld, tid);
  sk.store(smem, tid); sv.store(smem + TILE, tid);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int cur = tile & 1;
    if (tile + 1 < ntiles) { sk.load(rsK, (int64_t)(tile + 1) * KT, ld, tid); sv.load(rsV, (int64_t)(tile + 1) * KT, ld, tid); }
    const char* kimg = smem + cur * BUF;
    const char* vimg = kimg + TILE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kbase = tile * KT + sub * 32;
      if (kbase < N) {
        f32x16 sacc, dpacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = s_init; dpacc[r] = d_init; }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kimg + u_off(sub * 32 + lq, (16 * s + 8 * hh) * 2, RS));
          sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], sacc, 0, 0, 0);
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const bf16x8 va = *reinterpret_cast<const bf16x8*>(vimg + u_off(sub * 32 + lq, (16 * s + 8 * hh) * 2, RS));
          dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[s], dpacc, 0, 0, 0);
        }
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kbase + (r & 3) + 8 * (r >> 2) + 4 * hh;
          ds[r] = kk < N ? exp2f(sacc[r] * c_exp2) * dpacc[r] : 0.f;
        }
        const bf16x8 df0 = pack8(ds), df1 = pack8(ds + 8);
#pragma unroll
        for (int i = 0; i < NDT; ++i) {
          const int byte = (i * 32 + 16 * (gi & 1) + 4 * tp) * 2;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int r0 = sub * 32 + 16 * s2 + 4 * (gi >> 1) + tq;
            const bf16x8 kT = tr_pair(kimg, u_off(r0, byte, RS), u_off(r0 + 8, byte, RS));
            dq[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT, s2 == 0 ? df0 : df1, dq[i], 0, 0, 0);
          }
        }
      }
    }
    if (tile + 1 < ntiles) { sk.store(smem + (cur ^ 1) * BUF, tid); sv.store(smem + (cur ^ 1) * BUF + TILE, tid); }
    __syncthreads();
  }
  if (q < N) {
    uint16_t* qrow = dqkv + ((int64_t)b * N + q) * ld + h * HDP;
#pragma unroll
    for (int i = 0; i < NDT; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int d = i * 32 + 8 * rr + 4 * hh;
        *reinterpret_cast<u32x2*>(qrow + d) = u32x2{pack_bf16x2(dq[i][4 * rr] * scale, dq[i][4 * rr + 1] * scale),
                                                     pack_bf16x2(dq[i][4 * rr + 2] * scale, dq[i][4 * rr + 3] * scale)};
      }
  }
}

template <int HDP, int NKS>
static int launch_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int64_t B,
                      int64_t N, int64_t H, float scale, hipStream_t s) {
  constexpr int RS = BwdCfg<HDP>::RS;
  const size_t lds_kv = 2 * (2 * 32 * RS + 256), lds_q = 2 * (2 * 64 * RS);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<HDP, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<HDP, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    attr = true;
  }
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((B * N + 3) / 4)), dim3(256), 0, s, (const uint16_t*)o, (const uint16_t*)dout,
                     delta, B * N, (int)N, (int)H, HDP);
  VITA_CHECK_LAUNCH("vita_attn_bwd(delta)");
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HDP, NKS>), grid, dim3(256), lds_kv, s, (const uint16_t*)qkv, (const uint16_t*)dout, lse,
                     delta, (uint16_t*)dqkv, (int)N, (int)H, scale);
  VITA_CHECK_LAUNCH("vita_attn_bwd(dkdv)");
  hipLaunchKernelGGL((attn_bwd_dq_kernel<HDP, NKS>), grid, dim3(256), lds_q, s, (const uint16_t*)qkv, (const uint16_t*)dout, lse, delta,
                     (uint16_t*)dqkv, (int)N, (int)H, scale);
  VITA_CHECK_LAUNCH("vita_attn_bwd(dq)");
  return VITA_OK;
}

extern "C" int vita_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                             int64_t B, int64_t N, int64_t H, int64_t hd, int64_t hdp, float scale, void* stream) {
  VITA_REQUIRE(qkv && o && dout && lse && delta && dqkv && B > 0 && N > 0 && H > 0 && B < 65536 && H < 65536);
  VITA_REQUIRE((((uintptr_t)qkv | (uintptr_t)dout | (uintptr_t)dqkv | (uintptr_t)o) & 15) == 0);
  VITA_REQUIRE(N * 3 * H * hdp * 2 < (int64_t)0xffffffffll);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nks = (hd + 15) / 16;
  if (hdp == 224 && nks == 13) return launch_bwd<224, 13>(qkv, o, dout, lse, delta, dqkv, B, N, H, scale, s);
  if (hdp == 64 && nks == 3) return launch_bwd<64, 3>(qkv, o, dout, lse, delta, dqkv, B, N, H, scale, s);
  if (hdp == 32 && nks == 2) return launch_bwd<32, 2>(qkv, o, dout, lse, delta, dqkv, B, N, H, scale, s);
  if (hdp == 128 && nks == 8) return launch_bwd<128, 8>(qkv, o, dout, lse, delta, dqkv, B, N, H, scale, s);
  vita_set_error("vita_attn_bwd: unsupported head dim", hipSuccess);
  return VITA_ERR_UNSUPPORTED;
}
