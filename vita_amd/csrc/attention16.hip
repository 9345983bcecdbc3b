// Fused attention on v_mfma_f32_16x16x32_bf16 tiles (gfx950).  Same algorithm, interfaces, ring / LDS-DMA structure and
// results as the 32x32x16 kernels of attention.hip (timm 0.6.12 Attention.forward; call sites
// networks/imaging_encoders.py:36-40, networks/imaging_decoders.py:43-45), re-tiled for the other MFMA shape: per FLOP it
// reads half the accumulator bytes and twice the operand bytes, 20 % less register-file traffic in total, and the chip -
// which holds its clock down under these kernels ("Power limit") - sustains a higher clock with it.
//
// Geometry of one wave: 32 queries = two 16-query blocks; a K/V tile = 32 keys = two 16-key blocks.
//   S^T[key][q]  = K Q^T     : A = K rows from LDS (lane: key l&15, 8 head-dim values 32s + 8g ..), B = Q fragments in
//                              registers (lane: query l&15, the same 8 head-dim values); g = lane >> 4.  The head dim is
//                              swept in steps of 32 (HDP / 32 steps; 7 for the 205-wide heads).
//   accumulator              : lane holds query l&15 and keys 4g .. 4g+3 of a 16-key block.
//   O^T[d][q]   += V^T P^T   : K index = the 32 keys.  B = P^T straight from the accumulators: slots 8g + j <- key 4g + j
//                              of key block 0, slots 8g + 4 + j <- key 16 + 4g + j of key block 1; A = V^T by two
//                              transposed LDS reads per 16-column block with exactly that key order (rows 4g .. 4g+3 and
//                              16 + 4g .. 16 + 4g + 3).  Only ceil(hd / 16) column blocks are computed (13 = 208 columns
//                              for hd = 205, against 7 x 32 = 224 with the 32x32 shape); the pad block is written as zeros.
// LDS image: rows of RS bytes, 16-byte chunk index XOR xf(row) with
//     xf(row) = (((row & 7) ^ ((row >> 1) & 4)) << 1) | ((row >> 3) & 1)          (row mod 16)
// which makes BOTH access kinds conflict-free for the 16x16x32 operand maps: a ds_read_b128 row read (its four 16-lane
// groups {0-3,12-15,20-27}, ... see rows {0-3,12-15} at one chunk and rows {4-11} at the next: the XOR values of either
// row set are closed under ^1) and a ds_read_b64_tr_b16 block read (a 32-lane half covers 8 consecutive rows x 32 bytes:
// xf >> 1 is a bijection on every aligned group of 8 rows).
#include "attn_common.h"

__device__ __forceinline__ int xf(int row) { return ((((row & 7) ^ ((row >> 1) & 4)) << 1) | ((row >> 3) & 1)); }

// Per-lane source offsets of a wave's LDS-DMA pieces of a 32-row image in the xf layout (see DmaOff in attn_common.h).
template <int HDP, int NW>
struct DmaOff16 {
  static constexpr int NI = 32 / AttCfg<HDP>::RPI / NW;
  uint32_t vb[NI];
  __device__ __forceinline__ void init(int64_t ld, int wave, int lane) {
    using C = AttCfg<HDP>;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int piece = wave * NI + j;
      const int row = piece * C::RPI + lane / C::CPR;
      const int clog = (lane % C::CPR) ^ xf(row);
      vb[j] = (uint32_t)(row * ld * 2 + clog * 16);
    }
  }
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, uint32_t tile_off, char* lds_img, int wave, int j) const {
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds_img + (wave * NI + j) * 1024), 16, vb[j] + tile_off, 0, 0, 0);
  }
};

// Per-lane LDS byte offsets inside one 32-row image (xf layout), every access = base register + immediate:
//   row read, 32-wide k-step s, 16-row block kb : rb[s & 3] + 256 * (s >> 2) + kb * 16 * RS
//   transposed read, column block dt, half u    : tb[dt & 7] + 256 * (dt >> 3) + u * 16 * RS
template <int HDP>
struct LdsAddr16 {
  static constexpr int RS = AttCfg<HDP>::RS;
  static constexpr int NK = HDP / 32;
  static constexpr int NRB = NK < 4 ? NK : 4;
  static constexpr int NTB = (HDP / 16) < 8 ? (HDP / 16) : 8;
  uint32_t rb[NRB];
  uint32_t tb[NTB];
  __device__ __forceinline__ void init(int lane) {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < NRB; ++j) rb[j] = l15 * RS + (((4 * j + g) ^ xf(l15)) << 4);
    const int q = l15 >> 2, p = l15 & 3, r = 4 * g + q;
#pragma unroll
    for (int j = 0; j < NTB; ++j) tb[j] = r * RS + (((2 * j + (p >> 1)) ^ xf(r)) << 4) + 8 * (p & 1);
  }
  __device__ __forceinline__ void advance_rows(int bytes) {
#pragma unroll
    for (int j = 0; j < NRB; ++j) rb[j] += bytes;
  }
  __device__ __forceinline__ void advance_tr(int bytes) {
#pragma unroll
    for (int j = 0; j < NTB; ++j) tb[j] += bytes;
  }
  __device__ __forceinline__ void advance(int bytes) { advance_rows(bytes); advance_tr(bytes); }
  // A operand of a [16 rows] x [32-wide k-step] product: this lane's row l&15 of block kb, columns 32 s + 8 g .. + 7
  __device__ __forceinline__ bf16x8 row(const char* img, int s, int kb) const {
    return *reinterpret_cast<const bf16x8*>(img + rb[s & 3] + 256 * (s >> 2) + kb * 16 * RS);
  }
  // A operand [16 columns of block DT] x [32 rows in the accumulator's key order], inline asm (see LdsAddr::tr_asm)
  template <int DT, int IMG>
  __device__ __forceinline__ bf16x8 tr_asm() const {
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    v2i lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(tb[DT & 7]), "n"(IMG + 256 * (DT >> 3)) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(tb[DT & 7]), "n"(IMG + 256 * (DT >> 3) + 16 * RS) : "memory");
    v4i v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

// max / sum over the four 16-lane groups of a wave (the 16 lanes of a group hold different queries)
__device__ __forceinline__ float xg_max(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  float m = fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
  const uint32_t v = __builtin_bit_cast(uint32_t, m);
  auto t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)t[0]), __builtin_bit_cast(float, (uint32_t)t[1]));
}
__device__ __forceinline__ float xg_sum(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  float m = __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
  const uint32_t v = __builtin_bit_cast(uint32_t, m);
  auto t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __builtin_bit_cast(float, (uint32_t)t[0]) + __builtin_bit_cast(float, (uint32_t)t[1]);
}
__device__ __forceinline__ bf16x8 pack44(const f32x4& a, const f32x4& b) {
  u32x4 u = {cvt_pk(a[0], a[1]), cvt_pk(a[2], a[3]), cvt_pk(b[0], b[1]), cvt_pk(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, u);
}
#ifndef ATT16_THR
#define ATT16_THR 8.0f      // see the lazy reference maximum in attn_fwd16_kernel
#endif
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
// The younger four waves of a workgroup (one per SIMD, beside an older one) run at this priority: at equal priority the older
// wave of a SIMD wins every issue arbitration and the younger one is the critical path of every tile (phase stamps, DESIGN.md
// section 4); N = 23 041 forward 3.67 -> 3.61 ms, N = 6 913 0.391 -> 0.383 ms (priority 1 and 3 alike).
#ifndef ATT16_YOUNG_PRIO
#define ATT16_YOUNG_PRIO 1
#endif
#define ATT16_NSLOT ATT_NSLOT

// ============================================================================================ forward
// ND = ceil(hd / 16): output column blocks that are computed; same arguments as attn_fwd_kernel.
template <int HDP, int ND, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd16_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                                float* __restrict__ lse, float* __restrict__ part, int wg_base,
                                                                int nsplit, int tiles_per_split, int N, int H, float c_exp2) {
  using C = AttCfg<HDP>;
  constexpr int RS = C::RS, NK = HDP / 32, BUF = 2 * C::TILE, NSLOT = ATT16_NSLOT;
  constexpr int NI = 32 / C::RPI / NW;                          // DMA pieces per wave per image
  constexpr int DPT = 2 * NI;                                   // DMA instructions per wave per tile
  constexpr int NA = 2 * NK;                                    // A fragments of the S chain (k-step x key block)
  extern __shared__ __attribute__((aligned(16))) char smem[];   // ring of NSLOT x [K image | V image]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int QB = (N + NW * 32 - 1) / (NW * 32);
  const int lin = nsplit == 1 ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : wg_base + (int)blockIdx.x / nsplit;
  const int sp = (int)blockIdx.x % nsplit;
  const int bx = lin % QB, h = (lin / QB) % H, b = lin / (QB * H);
  const int64_t ld = 3ll * H * HDP;
  const int q0 = bx * (NW * 32) + wave * 32;
  const uint16_t* base = qkv + (int64_t)b * N * ld;
  const uint32_t extent = (uint32_t)((((int64_t)N - 1) * ld + HDP) * 2);
  const __amdgpu_buffer_rsrc_t rsQ = mk_rsrc(base + h * HDP, extent);
  const __amdgpu_buffer_rsrc_t rsK = mk_rsrc(base + (H + h) * HDP, extent);
  const __amdgpu_buffer_rsrc_t rsV = mk_rsrc(base + (2 * H + h) * HDP, extent);

  lds_zero<HDP>(smem, NSLOT * BUF, tid);
  const int l15 = lane & 15, g = lane >> 4;
  const bool active = __builtin_amdgcn_readfirstlane(q0) < N;     // waves past the last row only feed the ring
  bf16x8 qf[2][NK];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int s = 0; s < NK; ++s)
      qf[qb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
          rsQ, (uint32_t)((((int64_t)q0 + 16 * qb + l15) * ld + 32 * s + 8 * g) * 2), 0, 0));
  f32x4 oacc[ND][2];
#pragma unroll
  for (int i = 0; i < ND; ++i) { oacc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // m_run: reference maximum of each query block's row (raw score units, identical in the four lanes that share a query),
  // mc = m_run * c_exp2, m_thr = m_run + ATT16_THR / c_exp2: a score above it forces an update.  Row sums come from the matrix
  // pipe: lsum^T[.][q] += ones[16 x 32 keys] P^T, every row of the 16 x 16 tile = sum over the tile's keys of the bf16
  // probabilities the PV product uses (no vector adds, no cross-lane reduction).
  float m_run[2] = {-INFINITY, -INFINITY}, mc[2] = {-INFINITY, -INFINITY}, m_thr[2] = {-INFINITY, -INFINITY};
  const float thr_raw = ATT16_THR / c_exp2;
  f32x4 lsum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

  const int ntiles = (N + 31) / 32;
  const int t0 = sp * tiles_per_split;
  const int t1 = t0 + tiles_per_split < ntiles ? t0 + tiles_per_split : ntiles;
  LdsAddr16<HDP> la;
  la.init(lane);
  DmaOff16<HDP, NW> off;
  off.init(ld, wave, lane);
  auto issue_piece = [&](int tile, int buf, int p) {
    const int tcl = tile < t1 ? tile : ntiles;
    const uint32_t toff = (uint32_t)tcl * (uint32_t)(32 * ld * 2);
    if (p < NI) off.issue(rsK, toff, smem + buf * BUF, wave, p);
    else off.issue(rsV, toff, smem + buf * BUF + C::TILE, wave, p - NI);
  };
  auto issue = [&](int tile, int buf) {
#pragma unroll
    for (int p = 0; p < 2 * NI; ++p) issue_piece(tile, buf, p);
  };
  __syncthreads();
  issue(t0, 0); issue(t0 + 1, 1);
  dma_wait_n<DPT>();               // tile t0 has landed
  ring_barrier();
  int slot = 0;
#if ATT_STAMP
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = stamp_now();
  const unsigned long long st_begin = st_last;
#endif
  bf16x8 pf0, pf1;           // P^T of the current tile, packed for the PV product (both query blocks)
  // S chain of tile kt (K rows at la.rb) with the LDS-DMA pieces of tile `itile` -> ring slot `islot` issued between its
  // MFMAs, then the softmax of the tile -> pf0 / pf1
  auto s_phase = [&](int kt, int islot, int itile) __attribute__((always_inline)) {
    const int kbase = kt * 32;
    f32x4 sacc[2][2];                                   // [key block][query block]
#pragma unroll
    for (int i = 0; i < 4; ++i) sacc[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // A fragment i = (k-step i >> 1, key block i & 1), read DEPTH fragments ahead; each feeds two MFMAs (both query blocks)
      constexpr int DEPTH = 4;
      bf16x8 a[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH && i < NA; ++i) a[i] = la.row(smem, i >> 1, i & 1);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        sacc[i & 1][0] = MFMA16(a[i % DEPTH], qf[0][i >> 1], sacc[i & 1][0]);
        sacc[i & 1][1] = MFMA16(a[i % DEPTH], qf[1][i >> 1], sacc[i & 1][1]);
        if (i + DEPTH < NA) a[i % DEPTH] = la.row(smem, (i + DEPTH) >> 1, (i + DEPTH) & 1);
#pragma unroll
        for (int p = i; p < 2 * NI; p += NA) issue_piece(itile, islot, p);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(1);
    if (kbase + 32 > N) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kbase + 16 * kb + 4 * g + j >= N) { sacc[kb][0][j] = -INFINITY; sacc[kb][1][j] = -INFINITY; }
    }
    // Lazy reference maximum: every lane takes the maximum of its own 8 scores per query; only when some lane of the wave
    // sees a score more than ATT16_THR (in the exp2 domain) above its query's reference is the exact row maximum taken
    // across the four lane groups and O / the row sums rescaled.  Probabilities are then bounded by 2^THR instead of 1,
    // which changes nothing for bf16 / fp32 (a common power-of-two-like factor of a row cancels in O / l).
    float lm[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      lm[qb] = fmaxf(fmaxf(fmaxf(sacc[0][qb][0], sacc[0][qb][1]), fmaxf(sacc[0][qb][2], sacc[0][qb][3])),
                     fmaxf(fmaxf(sacc[1][qb][0], sacc[1][qb][1]), fmaxf(sacc[1][qb][2], sacc[1][qb][3])));
    if (__any(lm[0] > m_thr[0] || lm[1] > m_thr[1])) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float m_new = fmaxf(m_run[qb], xg_max(lm[qb]));
        const float a = fast_exp2((m_run[qb] - m_new) * c_exp2);
        lsum[qb] *= a;
#pragma unroll
        for (int i = 0; i < ND; ++i) oacc[i][qb] *= a;
        m_run[qb] = m_new;
        mc[qb] = m_new * c_exp2;
        m_thr[qb] = m_new + thr_raw;
      }
    }
    f32x4 p[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) p[kb][qb][j] = fast_exp2(__builtin_fmaf(sacc[kb][qb][j], c_exp2, -mc[qb]));
    pf0 = pack44(p[0][0], p[1][0]); pf1 = pack44(p[0][1], p[1][1]);
#if ATT_STAMP
    asm volatile("" :: "v"(pf0), "v"(pf1));
    STAMP(2);
#endif
  };
  // O^T += V^T P^T of the tile whose V image la.tb addresses; row sums on the matrix pipe
  auto pv_phase = [&]() __attribute__((always_inline)) {
    bf16x8 fr[3];      // transposed V fragments two column blocks ahead of their MFMAs
    fr[0] = la.template tr_asm<0, C::TILE>();
    if constexpr (ND > 1) fr[1] = la.template tr_asm<1, C::TILE>();
    static_for<ND>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + 2 < ND) {
        fr[(i + 2) % 3] = la.template tr_asm<i + 2, C::TILE>();
        tr_wait<4>();
      } else if constexpr (i + 1 < ND) {
        tr_wait<2>();
      } else {
        tr_wait<0>();
      }
      oacc[i][0] = MFMA16(fr[i % 3], pf0, oacc[i][0]);
      oacc[i][1] = MFMA16(fr[i % 3], pf1, oacc[i][1]);
      if constexpr (i == 0) { lsum[0] = MFMA16(ones, pf0, lsum[0]); lsum[1] = MFMA16(ones, pf1, lsum[1]); }
      __builtin_amdgcn_sched_barrier(0);
    });
    STAMP(3);
  };
  if (ATT16_YOUNG_PRIO && wave >= NW / 2) __builtin_amdgcn_s_setprio(ATT16_YOUNG_PRIO);
  {
    for (int kt = t0; kt < t1; ++kt) {
      int pre = slot + 2; if (pre >= NSLOT) pre -= NSLOT;     // slot of tile kt+2
      STAMP(0);
      if (active) {
        s_phase(kt, pre, kt + 2);
        pv_phase();
      } else {
        issue(kt + 2, pre);          // waves past the last query row only feed the ring
      }
      slot = slot + 1 == NSLOT ? 0 : slot + 1;
      la.advance(slot == 0 ? -(NSLOT - 1) * BUF : BUF);
      dma_wait_n<DPT>();             // tile kt+1 has landed
      STAMP(4);
      ring_barrier();
      STAMP(5);
    }
  }
#if ATT_STAMP
  if (lane == 0 && nsplit == 1 && blockIdx.x < ATT_STAMP_WG) {
    st_sum[6] = st_last - st_begin; st_sum[7] = (unsigned long long)(t1 - t0);
    for (int k = 0; k < 8; ++k) att_stamp[((size_t)blockIdx.x * 8 + wave) * 8 + k] = st_sum[k];
  }
#endif
  dma_wait();                      // zero-filled prefetches past the last tile
  float inv[2], l2[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l_tot = lsum[qb][0];
    inv[qb] = 1.f / l_tot;
    l2[qb] = m_run[qb] * c_exp2 + log2f(l_tot);
  }
  if (part != nullptr) {
    // part[(tail workgroup * nsplit + split)]: [NW*32 queries][HDP] fp32 normalised output, then [NW*32] log2-sum-exp
    float* pw_ = part + ((int64_t)(lin - wg_base) * nsplit + sp) * ((NW * 32) * (HDP + 1));
    if (active) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float* prow = pw_ + (int64_t)(wave * 32 + 16 * qb + l15) * HDP;
#pragma unroll
        for (int i = 0; i < ND; ++i) *reinterpret_cast<f32x4*>(prow + 16 * i + 4 * g) = oacc[i][qb] * inv[qb];
        if constexpr (16 * ND < HDP) *reinterpret_cast<f32x4*>(prow + 16 * ND + 4 * g) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g == 0) pw_[(NW * 32) * HDP + wave * 32 + 16 * qb + l15] = l2[qb];
      }
    }
  } else {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = q0 + 16 * qb + l15;
      if (q < N) {
        uint16_t* orow = o + ((int64_t)b * N + q) * ((int64_t)H * HDP) + h * HDP;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const f32x4 v = oacc[i][qb] * inv[qb];
          *reinterpret_cast<u32x2*>(orow + 16 * i + 4 * g) = u32x2{cvt_pk(v[0], v[1]), cvt_pk(v[2], v[3])};
        }
        if constexpr (16 * ND < HDP) *reinterpret_cast<u32x2*>(orow + 16 * ND + 4 * g) = u32x2{0u, 0u};
        if (g == 0) lse[((int64_t)b * H + h) * N + q] = l2[qb] * 0.6931471805599453f;
      }
    }
  }
}

template __global__ void attn_fwd16_kernel<224, 13, 8>(const uint16_t*, uint16_t*, float*, float*, int, int, int, int, int, float);
template __global__ void attn_fwd16_kernel<64, 3, 8>(const uint16_t*, uint16_t*, float*, float*, int, int, int, int, int, float);
template __global__ void attn_fwd16_kernel<32, 2, 8>(const uint16_t*, uint16_t*, float*, float*, int, int, int, int, int, float);
template __global__ void attn_fwd16_kernel<128, 8, 8>(const uint16_t*, uint16_t*, float*, float*, int, int, int, int, int, float);

// ============================================================================================ host
template <int HDP, int ND>
static int launch_fwd16(const void* qkv, void* o, float* lse, void* ws, int64_t B, int64_t N, int64_t H, float scale, hipStream_t s) {
  constexpr int NW = 8;
  static const bool no_tail = att_env_flag("VITA_ATTN_NO_TAIL_SPLIT");
  const size_t lds = ATT16_NSLOT * 2 * AttCfg<HDP>::TILE;
  const void* fn = (const void*)attn_fwd16_kernel<HDP, ND, NW>;
  VITA_ONCE_PER_DEVICE((void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto kern = attn_fwd16_kernel<HDP, ND, NW>;
  const int64_t L = B * H * ((N + NW * 32 - 1) / (NW * 32));
  const int ntiles = (int)((N + 31) / 32);
  int64_t l_main = L; int nsplit = 1, per = ntiles;
  if (ws != nullptr && !no_tail) att_tail_plan(L, ntiles, att_num_cu(), ATT_FWD_SLOTS, &l_main, &nsplit, &per);
  const float c = scale * 1.4426950408889634f;
  if (l_main > 0)
    hipLaunchKernelGGL(kern, dim3((unsigned)l_main), dim3(NW * 64), lds, s, (const uint16_t*)qkv,
                       (uint16_t*)o, lse, (float*)nullptr, 0, 1, ntiles, (int)N, (int)H, c);
  if (l_main < L) {
    const int nwg = (int)(L - l_main);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nwg * nsplit)), dim3(NW * 64), lds, s, (const uint16_t*)qkv,
                       (uint16_t*)o, lse, (float*)ws, (int)l_main, nsplit, per, (int)N, (int)H, c);
    att_launch_fwd_combine((const float*)ws, o, lse, (int)l_main, nwg, nsplit, (int)N, (int)H, HDP, NW * 32, s);
  }
  VITA_CHECK_LAUNCH("vita_attn_fwd(16x16x32)");
  return VITA_OK;
}

int att16_launch_fwd(int hdp, int nks, const void* qkv, void* o, float* lse, void* ws, int64_t B, int64_t N, int64_t H, float scale,
                     hipStream_t s) {
  if (hdp == 224 && nks == 13) return launch_fwd16<224, 13>(qkv, o, lse, ws, B, N, H, scale, s);
  if (hdp == 64 && nks == 3) return launch_fwd16<64, 3>(qkv, o, lse, ws, B, N, H, scale, s);
  if (hdp == 32 && nks == 2) return launch_fwd16<32, 2>(qkv, o, lse, ws, B, N, H, scale, s);
  if (hdp == 128 && nks == 8) return launch_fwd16<128, 8>(qkv, o, lse, ws, B, N, H, scale, s);
  return VITA_ERR_UNSUPPORTED;
}
