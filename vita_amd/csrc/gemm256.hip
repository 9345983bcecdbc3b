// Large-tile bf16 MFMA GEMM for gfx950: 256 x (NF*16) x 64 tiles, 8 waves (4 along M x 2 along N),
// one workgroup per CU.  Same contract as gemm_kernel (gemm.hip); used for the big un-batched GEMMs of
// the transformer blocks (nn.Linear forward / dgrad / wgrad).
//
//   * operands go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, bounds-checked: rows past
//     the operand and K tails arrive as zeros); the XOR swizzle is applied to the per-lane SOURCE
//     address because the DMA writes LDS linearly
//   * two 64 KiB LDS stages; the DMA of K-tile t+2 is issued right after the barrier that retires
//     K-tile t, so it has a whole K-tile of MFMAs (2 waves per SIMD) to land
//   * fragments are double-buffered in registers: the ds_reads of the next 32-deep k-step are issued
//     under the MFMAs of the current one; one barrier per K-tile
//   * NF*16 = 208 / 224 / 256 columns so that N = 1040, 4160 (x208) and 1120, 3360 (x224) tile exactly
//   * k-contiguous operand: LDS image [rows][128 B], 16-B chunk ^ (row & 7), ds_read_b128
//     k-strided operand   : LDS image [64 k][512 B], 32-B chunk ^ f(k), ds_read_b64_tr_b16
//   * grouped XCD-aware raster: the 32 tiles running on one XCD form 8 M-panels x 4 N-panels
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifndef G2_DBG
#define G2_DBG 0
#endif
#define G2_BM 256
#define G2_BK 64
#define G2_STAGE 65536
#define G2_BOFF 32768
#define G2_THREADS 512
#define G2_OOB 0x7ffffff0u
#define G2_NARROW_OVERHEAD 3.0   // fixed per-tile cost of the 128..192-column tiles in the cost model's units

struct Gemm2Params {
  vita_gemm_desc d;
  int tiles_m, tiles_n, k_tiles_per_split;
  int group_m;      // M-panels per raster group (the workgroups of a group share an XCD's L2)
};

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g2_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t g2_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// ---- LDS-DMA loader of one operand's K-tile image -------------------------------------------------
// FREE = free extent of the tile (256 for A, NF*16 for B).  Every wave issues PPW = 4 pieces of 1 KiB per
// K-tile; pieces past the image write zeros into the unused tail of the 32 KiB region.
__device__ __forceinline__ uint32_t g2_sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

template <int LAYOUT, int FREE, int NWAVE = 8>
struct G2Loader {
  static constexpr int NPIECE = LAYOUT == 0 ? FREE / 8 : 32;
  static constexpr int PPW = (NPIECE + NWAVE - 1) / NWAVE;     // 4 with 8 waves; the region holds PPW * NWAVE KiB
  static constexpr int BYTES = PPW * NWAVE * 1024;
  uint32_t voff[PPW];
  uint32_t base_lo, base_hi, bytes, kstep;   // wave-uniform (SGPRs)
  int kchunk;   // LAYOUT 0: k element offset of this lane's 16-B chunk inside a K-tile
  int K;

  __device__ __forceinline__ void init(const uint16_t* op, int64_t free0, int64_t nfree, int64_t Ktot, int64_t ld,
                                       int wave, int lane) {
    K = (int)Ktot;
    uint64_t bp;
    if (LAYOUT == 0) {
      int64_t rows = nfree - free0; if (rows > FREE) rows = FREE;
      bp = (uint64_t)(op + free0 * ld);
      bytes = g2_sgpr((uint32_t)(rows * ld * 2));
      kstep = G2_BK * 2;
      const int clog = (lane & 7) ^ (lane >> 3);
      kchunk = clog * 8;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int p = wave + NWAVE * i;
        const int row = p * 8 + (lane >> 3);
        voff[i] = p < NPIECE ? (uint32_t)(row * ld * 2 + clog * 16) : G2_OOB;
      }
    } else {
      int64_t nfree8 = (nfree + 7) & ~(int64_t)7; if (nfree8 > ld) nfree8 = ld;
      bp = (uint64_t)(op + free0);
      bytes = g2_sgpr((uint32_t)(((Ktot - 1) * ld + (nfree8 - free0)) * 2));
      kstep = g2_sgpr((uint32_t)(G2_BK * ld * 2));
      kchunk = 0;
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int p = wave + NWAVE * i;
        const int krow = 2 * p + (lane >> 5);
        const int pc = (lane & 31) >> 1, half = lane & 1;
        const int lc = pc ^ (int)g2_f(krow);
        const int col = lc * 16 + half * 8;
        const bool ok = col < FREE && free0 + col < nfree8;
        voff[i] = ok ? (uint32_t)((int64_t)krow * ld * 2 + col * 2) : G2_OOB;
      }
    }
    base_lo = g2_sgpr((uint32_t)bp); base_hi = g2_sgpr((uint32_t)(bp >> 32));
  }
  // resource descriptor of K-tile kt (absolute index): base advanced, extent shrunk, all in SGPRs
  __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(int kt) const {
    const uint32_t adv = g2_sgpr((uint32_t)kt * kstep);
    const uint64_t bp = (((uint64_t)base_hi << 32) | base_lo) + adv;
    const uint32_t rem = g2_sgpr(adv < bytes ? bytes - adv : 0u);
    return g2_rsrc((const void*)bp, rem);
  }
  __device__ __forceinline__ bool keep(int kt) const { return LAYOUT != 0 || (kt * G2_BK + kchunk < K); }
  // piece i (0..3) of this wave; img = LDS address of this operand's 32 KiB region in the target stage
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, bool kp, int i, char* img, int wave, bool in_loop = false) const {
    if ((G2_DBG & 8) && in_loop) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(img + (wave + NWAVE * i) * 1024), 16, kp ? voff[i] : G2_OOB, 0, 0, 0);
  }
};

// ---- fragment readers -------------------------------------------------------------------------------
// NFR fragments (16 free indices each) starting at tile-local fragment index f_first.
template <int LAYOUT, int NFR>
struct G2Frags {
  // LAYOUT 0: b[ks] ; LAYOUT 1: b[j]
  uint32_t b[LAYOUT == 0 ? 2 : NFR];
  __device__ __forceinline__ void init(int f_first, uint32_t img_off, int lane) {
    if (LAYOUT == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        b[ks] = img_off + (f_first * 16 + (lane & 15)) * 128 + ((((ks * 4 + (lane >> 4)) ^ (lane & 7))) << 4);
    } else {
      const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
      const int fl = q | ((g & 1) << 2);
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const int fi = f_first + j;
        b[j] = img_off + (8 * g + q) * 512 + ((fi ^ fl) << 5) + p * 8;
      }
    }
  }
  // delta = +stage bytes when moving from stage 0 to stage 1, -stage bytes back
  __device__ __forceinline__ void toggle(int delta) {
#pragma unroll
    for (int j = 0; j < (LAYOUT == 0 ? 2 : NFR); ++j) b[j] += delta;
  }
  // ASM: issue the LDS reads as inline asm.  hipcc puts s_waitcnt vmcnt(0) in front of every
  // ds_read_b64_tr_b16 builtin while an LDS-DMA is in flight (it cannot tell the DMA's LDS write from the
  // read), which would serialise the pipeline; the asm form is invisible to that pass, so kernels that use
  // it wait for their fragments with an explicit lgkmcnt(0) before the MFMAs that consume them.
  template <int KS, bool ASM, int J>
  __device__ __forceinline__ bf16x8 read(const char* smem) const {
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    if constexpr (LAYOUT == 0) {
      if constexpr (!ASM) {
        return *reinterpret_cast<const bf16x8*>(smem + b[KS] + J * 2048);
      } else {
        v4i v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(b[KS]), "n"(J * 2048) : "memory");
        return __builtin_bit_cast(bf16x8, v);
      }
    } else {
      v2i lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(b[J]), "n"(KS * 16384) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(b[J]), "n"(KS * 16384 + 2048) : "memory");
      v4i v = {lo[0], lo[1], hi[0], hi[1]};
      return __builtin_bit_cast(bf16x8, v);
    }
  }
  // fragments [LO, HI) of k-step KS into dst
  template <int KS, bool ASM, int LO, int HI, int N>
  __device__ __forceinline__ void read_range(const char* smem, bf16x8 (&dst)[N]) const {
    if constexpr (LO < HI && LO < N) {
      dst[LO] = read<KS, ASM, LO>(smem);
      read_range<KS, ASM, LO + 1, HI, N>(smem, dst);
    }
  }
};

// s_waitcnt vmcnt(0) lgkmcnt(0) as a builtin (the compiler's own counters stay in step), then a raw
// barrier: a __syncthreads() would be the same here, but this keeps the intent explicit
__device__ __forceinline__ void g2_sync_all() {
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);   // nothing (e.g. a register-only MFMA fed by asm reads) moves above the wait
}

// NF = tile width in 16-column fragments; waves 0-3 own the first NFW = ceil(NF/2) fragments, waves 4-7
// the rest (the two waves of a SIMD are w and w+4, so an odd NF still balances per SIMD).
// ATOMIC: fp32 C accumulated with atomics through the segment maps (split-K weight gradients); the MFMA
// is then issued un-swapped so that one accumulator register covers 4 rows x 64 contiguous bytes.
// NWM = waves along M: 4 -> 256-row tiles, 8 waves, two 64 KiB stages, one workgroup per CU; 2 -> 128-row tiles, 4 waves,
// stages of 16 KiB (A) + NF * 2 KiB (B) (NF <= 12: 80 KiB), so that TWO workgroups share a CU and one's epilogue store
// tail runs under the other's main loop (k-contiguous operands only).
template <int LA, int LB, int NF, bool ATOMIC, int NWM>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256_kernel(const Gemm2Params P) {
  constexpr int NFW = (NF + 1) / 2;
  constexpr bool ASMR = LA == 1 || LB == 1;
  constexpr int NWAVE = 2 * NWM, BM = 64 * NWM;
  static_assert(NWM == 4 || (LA == 0 && LB == 0 && !ATOMIC), "the 128-row variant handles k-contiguous operands only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const vita_gemm_desc& d = P.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % NWM, wn = wave / NWM;
  const int nfw = wn == 0 ? NFW : NF - NFW;

  // grouped, XCD-aware raster
  int tm, tn;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GM = P.group_m;
    const int per_group = GM * P.tiles_n;
    const int g = l / per_group, in_g = l - g * per_group;
    int gm = P.tiles_m - g * GM; if (gm > GM) gm = GM;
    tm = g * GM + in_g % gm;
    tn = in_g / gm;
  }
  const int split = blockIdx.y;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * (NF * 16);
  const int kt_total = (int)((d.k + G2_BK - 1) / G2_BK);
  const int kt_begin = split * P.k_tiles_per_split;
  int kt_end = kt_begin + P.k_tiles_per_split; if (kt_end > kt_total) kt_end = kt_total;
  const int nkt = kt_end - kt_begin;

  typedef G2Loader<LA, BM, NWAVE> LoaderA;
  typedef G2Loader<LB, NF * 16, NWAVE> LoaderB;
  constexpr int BOFF = LoaderA::BYTES, STAGE = BOFF + LoaderB::BYTES;   // 32 KiB + 32 KiB with 8 waves
  constexpr int PA = LoaderA::PPW, PB = LoaderB::PPW, NDMA = PA + PB;
  LoaderA la;
  LoaderB lb;
  la.init((const uint16_t*)d.a, m0, d.m, d.k, d.lda, wave, lane);
  lb.init((const uint16_t*)d.b, n0, d.n, d.k, d.ldb, wave, lane);
  G2Frags<LA, 4> fa;
  G2Frags<LB, NFW> fb;
  fa.init(wm * 4, 0, lane);
  fb.init(wn * NFW, BOFF, lane);

  f32x4 acc[4][NFW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NFW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 a0[4], b0[NFW], a1[4], b1[NFW];
#define G2_MFMA_GROUP(J, AF, BF)                                                                          \
  if (NF % 2 == 0 || (J) < NFW - 1 || wn == 0) {                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
      acc[i][J] = ATOMIC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[i], BF[J], acc[i][J], 0, 0, 0)      \
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[J], AF[i], acc[i][J], 0, 0, 0);     \
  }
  // B fragments [lo, hi) of k-step KS into BF
#define G2_READ_B(KS, BF, LO, HI) fb.template read_range<KS, ASMR, LO, HI, NFW>(smem, BF)
#define G2_READ_A(KS, AF) fa.template read_range<KS, ASMR, 0, 4, 4>(smem, AF)
  // in-loop reads; timing-only builds (G2_DBG, results INVALID) drop them: 1 = B fragments, 2 = A fragments;
  // 4 = no per-K-tile barrier (wait kept), 16 = neither wait nor barrier, 8 = no in-loop LDS-DMA issue
#define G2_LREAD_B(KS, BF, LO, HI) do { if (!(G2_DBG & 1)) { G2_READ_B(KS, BF, LO, HI); } } while (0)
#define G2_LREAD_A(KS, AF) do { if (!(G2_DBG & 2)) { G2_READ_A(KS, AF); } } while (0)
  // asm-read kernels: every fragment read issued so far has landed (the compiler cannot see those reads)
#define G2_WAIT_FRAGS() do { if (ASMR) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } } while (0)
  // phase 1: MFMAs of k-step 0 of the current K-tile; the fragments of its k-step 1 are read underneath
#define G2_PHASE1()                                                  \
  G2_WAIT_FRAGS();                                                   \
  _Pragma("unroll") for (int j = 0; j < NFW; ++j) {                  \
    if (j == 0) { G2_LREAD_A(1, a1); G2_LREAD_B(1, b1, 0, 1); }      \
    if (j == 1) { G2_LREAD_B(1, b1, 1, 4); }                         \
    if (j == 2) { G2_LREAD_B(1, b1, 4, NFW); }                       \
    G2_MFMA_GROUP(j, a0, b0);                                        \
    __builtin_amdgcn_sched_barrier(0);                               \
  }

  if (nkt > 0) {
    {
      const __amdgpu_buffer_rsrc_t ra = la.rsrc(kt_begin), rb = lb.rsrc(kt_begin);
      const bool ka = la.keep(kt_begin), kb = lb.keep(kt_begin);
#pragma unroll
      for (int i = 0; i < PA; ++i) la.issue(ra, ka, i, smem, wave);
#pragma unroll
      for (int i = 0; i < PB; ++i) lb.issue(rb, kb, i, smem + BOFF, wave);
    }
    g2_sync_all();
    {
      // K-tile 1 (a K-tile past the end reads zeros / in-bounds bytes that are never used)
      const __amdgpu_buffer_rsrc_t ra = la.rsrc(kt_begin + 1), rb = lb.rsrc(kt_begin + 1);
      const bool ka = la.keep(kt_begin + 1), kb = lb.keep(kt_begin + 1);
#pragma unroll
      for (int i = 0; i < PA; ++i) la.issue(ra, ka, i, smem + STAGE, wave);
#pragma unroll
      for (int i = 0; i < PB; ++i) lb.issue(rb, kb, i, smem + STAGE + BOFF, wave);
    }
    G2_READ_A(0, a0); G2_READ_B(0, b0, 0, NFW);
    if (G2_DBG) {
      G2_WAIT_FRAGS();
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = a0[i];
#pragma unroll
      for (int j = 0; j < NFW; ++j) b1[j] = b0[j];
    }
    for (int t = 0; t + 1 < nkt; ++t) {
      G2_PHASE1();
      // K-tile t+1 has landed (its DMA was issued one K-tile ago) - for every wave; every wave is also
      // done reading this stage, which K-tile t+2 may now overwrite
      if (G2_DBG & 4) { __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_sched_barrier(0); }     // wait, no barrier
      else if (!(G2_DBG & 16)) g2_sync_all();                                                       // 16: neither
      fa.toggle((t & 1) ? -STAGE : STAGE); fb.toggle((t & 1) ? -STAGE : STAGE);
      const int kt2 = kt_begin + t + 2;
      const __amdgpu_buffer_rsrc_t ra = la.rsrc(kt2), rb = lb.rsrc(kt2);
      const bool ka = la.keep(kt2), kb = lb.keep(kt2);
      char* st = smem + (t & 1) * STAGE;
      // ---- phase 2: MFMAs of k-step 1; DMA of K-tile t+2 and the reads of (t+1, k-step 0) spread between them
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        if constexpr (NFW >= 7 && NWM == 4) {
          if (j == 0) { la.issue(ra, ka, 0, st, wave, true); la.issue(ra, ka, 1, st, wave, true); }
          if (j == 1) la.issue(ra, ka, 2, st, wave, true);
          if (j == 2) la.issue(ra, ka, 3, st, wave, true);
          if (j == 3) lb.issue(rb, kb, 0, st + BOFF, wave, true);
          if (j == 4) lb.issue(rb, kb, 1, st + BOFF, wave, true);
          if (j == 5) lb.issue(rb, kb, 2, st + BOFF, wave, true);
          if (j == 6) lb.issue(rb, kb, 3, st + BOFF, wave, true);
          if (j == 2) { G2_LREAD_A(0, a0); }
          if (j == 3) { G2_LREAD_B(0, b0, 0, 3); }
          if (j == 4) { G2_LREAD_B(0, b0, 3, NFW); }
        } else {
          // narrow tiles (NFW = 4..6 MFMA groups per k-step): DMA instruction i of the NDMA goes to group i*NFW/NDMA,
          // the fragment reads of the next K-tile's first k-step to the first three groups
#pragma unroll
          for (int i = 0; i < NDMA; ++i)
            if ((i * NFW) / NDMA == j) {
              if (i < PA) la.issue(ra, ka, i, st, wave, true); else lb.issue(rb, kb, i - PA, st + BOFF, wave, true);
            }
          if (j == 0) { G2_LREAD_A(0, a0); }
          if (j == 1) { G2_LREAD_B(0, b0, 0, 2); }
          if (j == 2) { G2_LREAD_B(0, b0, 2, NFW); }
        }
        G2_MFMA_GROUP(j, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // last K-tile
    G2_PHASE1();
    G2_WAIT_FRAGS();
#pragma unroll
    for (int j = 0; j < NFW; ++j) { G2_MFMA_GROUP(j, a1, b1); }
  }
#undef G2_PHASE1
#undef G2_WAIT_FRAGS
#undef G2_MFMA_GROUP
#undef G2_READ_A
#undef G2_READ_B
#undef G2_LREAD_A
#undef G2_LREAD_B

  // ---- epilogue: each wave stages its 64 x (nfw*16) fp32 block through a private 16 KiB LDS slice, 32
  // rows at a time, and writes whole rows (2 rows x up to 32 lanes x 16 B per wave-instruction)
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_s_barrier();           // every wave is done with the operand stages
  constexpr int RS = NFW * 16 + 4;        // staging row stride in floats (pad keeps ds_write_b128 conflict-free)
  float* stg = reinterpret_cast<float*>(smem + wave * (32 * RS * 4));
  const float alpha = d.alpha;
  const int ncol = nfw * 16;
  const int64_t nw0 = n0 + wn * NFW * 16;
  // bf16 C (and bf16 aux) with 16-B aligned rows: the wide-store epilogue (uniform over the launch)
  const bool wide_bf16 = !ATOMIC && d.c_dtype == VITA_DT_BF16 && d.epilogue != VITA_EPI_RESIDUAL && (d.n & 7) == 0 && (d.ldc & 7) == 0 &&
                         ((uintptr_t)d.c & 15) == 0 && (d.bias == nullptr || ((uintptr_t)d.bias & 15) == 0) &&
                         (d.epilogue == VITA_EPI_NONE ||
                          (d.epilogue == VITA_EPI_GELU && ((uintptr_t)d.aux_out & 15) == 0) ||
                          (d.epilogue == VITA_EPI_GELU_BWD && (d.ld_aux & 7) == 0 && ((uintptr_t)d.aux_in & 15) == 0));
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (ATOMIC) {
      // lane holds C[m = 4*(lane>>4) + r][n = lane&15] of each fragment
#pragma unroll
      for (int il = 0; il < 2; ++il)
#pragma unroll
        for (int j = 0; j < NFW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stg[(il * 16 + 4 * (lane >> 4) + r) * RS + j * 16 + (lane & 15)] = acc[pass * 2 + il][j][r];
    } else {
      // lane holds C[m = lane&15][n = 4*(lane>>4) + 0..3]
#pragma unroll
      for (int il = 0; il < 2; ++il)
#pragma unroll
        for (int j = 0; j < NFW; ++j)
          *reinterpret_cast<f32x4*>(stg + (il * 16 + (lane & 15)) * RS + j * 16 + 4 * (lane >> 4)) = acc[pass * 2 + il][j];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slice is wave-private, no barrier needed
    const int64_t mrow0 = m0 + wm * 64 + pass * 32;
    if (ATOMIC) {
      float* C = (float*)d.c;
      // columns lane and lane+64 of the block; their logical column index is fixed over the rows
      int64_t lnc[2]; bool okc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        const int64_t n = nw0 + c;
        okc[h] = c < ncol && n < d.n;
        lnc[h] = okc[h] ? seg_map(n, d.col_seg_phys, d.col_seg_log) : -1;
        okc[h] = okc[h] && lnc[h] >= 0;
      }
      // logical row index, advanced incrementally (one division per pass)
      int64_t rq = 0, rr = mrow0;
      if (d.row_seg_phys) { rq = mrow0 / d.row_seg_phys; rr = mrow0 % d.row_seg_phys; }
#pragma nounroll
      for (int row = 0; row < 32; ++row) {
        const int64_t m = mrow0 + row;
        const bool okr = m < d.m && (d.row_seg_phys == 0 || rr < d.row_seg_log);
        const int64_t lm = d.row_seg_phys ? rq * d.row_seg_log + rr : m;
        if (okr) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (!okc[h]) continue;
            const float v = stg[row * RS + lane + 64 * h] * alpha;
            if (d.split_k > 1) atomicAdd(C + lm * d.ldc + lnc[h], v);
            else C[lm * d.ldc + lnc[h]] = v;
          }
        }
        if (d.row_seg_phys) { if (++rr == d.row_seg_phys) { rr = 0; ++rq; } }
      }
    } else if (wide_bf16) {
      // bf16 output, 16-B stores: a lane owns 8 columns, a wave-instruction writes 4 rows x 16 lanes x 16 B.  The
      // epilogue is bound by the CU's store ISSUE rate (MI355X_MICROARCH.md, store tail), so half as many store
      // instructions as the 8-B form below is half the tail.
      const int c8 = lane & 15;
      const int64_t n = nw0 + c8 * 8;
      const bool okc = c8 * 8 < ncol && n < d.n;      // d.n % 8 == 0 on this path
      const int64_t nl = okc ? n : 0;
      f32x4 bv0 = f32x4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
      if (okc && d.bias) { bv0 = *reinterpret_cast<const f32x4*>(d.bias + n); bv1 = *reinterpret_cast<const f32x4*>(d.bias + n + 4); }
      auto runw = [&](auto tag) {
        constexpr int EPI = decltype(tag)::value;
        u32x4 ux[8];
        if (EPI == VITA_EPI_GELU_BWD) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            int64_t m = mrow0 + 4 * u + (lane >> 4);
            if (m >= d.m) m = d.m - 1;
            ux[u] = *reinterpret_cast<const u32x4*>((const uint16_t*)d.aux_in + m * d.ld_aux + nl);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = 4 * u + (lane >> 4);
          const int64_t m = mrow0 + row;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 8) * alpha + bv0;
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * RS + c8 * 8 + 4) * alpha + bv1;
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          const bool ok = okc && m < d.m;
          const int64_t o = m * d.ldc + n;
          if (EPI == VITA_EPI_GELU) {
            float dg[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { float g, dgr; gelu_erf_both(v[r], g, dgr); v[r] = g; dg[r] = dgr; }
            if (ok) *reinterpret_cast<u32x4*>((uint16_t*)d.aux_out + o) =
                u32x4{pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3]), pack_bf16x2(dg[4], dg[5]), pack_bf16x2(dg[6], dg[7])};
          } else if (EPI == VITA_EPI_GELU_BWD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[2 * r] *= __builtin_bit_cast(float, ux[u][r] << 16);
              v[2 * r + 1] *= __builtin_bit_cast(float, ux[u][r] & 0xffff0000u);
            }
          }
          if (ok) *reinterpret_cast<u32x4*>((uint16_t*)d.c + o) =
              u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        }
      };
      switch (d.epilogue) {
        case VITA_EPI_GELU: runw(std::integral_constant<int, VITA_EPI_GELU>{}); break;
        case VITA_EPI_GELU_BWD: runw(std::integral_constant<int, VITA_EPI_GELU_BWD>{}); break;
        default: runw(std::integral_constant<int, VITA_EPI_NONE>{}); break;
      }
    } else {
      const int c4 = lane & 31;
      const int64_t n = nw0 + c4 * 4;
      const bool okc = c4 * 4 < ncol && n < d.n;      // d.n % 4 == 0 on this path
      const int64_t nl = okc ? n : 0;                 // loads always touch valid addresses; only stores are predicated
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (okc && d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n);
      // 8 rows-pairs per batch: the batch's global loads (residual / saved pre-activation) are all issued
      // before the first use, so their latency is paid once per batch instead of once per row
      auto run = [&](auto tag) {
        constexpr int EPI = decltype(tag)::value;
#pragma nounroll
        for (int it0 = 0; it0 < 16; it0 += 8) {
          f32x4 res[8];
          u32x2 ux[8];
          if (EPI == VITA_EPI_RESIDUAL || EPI == VITA_EPI_GELU_BWD) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              int64_t m = mrow0 + 2 * (it0 + u) + (lane >> 5);
              if (m >= d.m) m = d.m - 1;
              if (EPI == VITA_EPI_RESIDUAL) res[u] = *reinterpret_cast<const f32x4*>((const float*)d.aux_in + m * d.ld_aux + nl);
              else ux[u] = *reinterpret_cast<const u32x2*>((const uint16_t*)d.aux_in + m * d.ld_aux + nl);
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int row = 2 * (it0 + u) + (lane >> 5);
            const int64_t m = mrow0 + row;
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * RS + c4 * 4) * alpha + bv;
            const bool ok = okc && m < d.m;
            const int64_t o = m * d.ldc + n;
            if (EPI == VITA_EPI_GELU) {
              f32x4 dg;
#pragma unroll
              for (int r = 0; r < 4; ++r) { float g, dgr; gelu_erf_both(v[r], g, dgr); v[r] = g; dg[r] = dgr; }
              if (ok) *reinterpret_cast<u32x2*>((uint16_t*)d.aux_out + o) = u32x2{pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3])};
            } else if (EPI == VITA_EPI_RESIDUAL) {
              v += res[u];
            } else if (EPI == VITA_EPI_GELU_BWD) {
              v[0] *= __builtin_bit_cast(float, ux[u][0] << 16);
              v[1] *= __builtin_bit_cast(float, ux[u][0] & 0xffff0000u);
              v[2] *= __builtin_bit_cast(float, ux[u][1] << 16);
              v[3] *= __builtin_bit_cast(float, ux[u][1] & 0xffff0000u);
            }
            if (ok) {
              if (d.c_dtype == VITA_DT_F32) *reinterpret_cast<f32x4*>((float*)d.c + o) = v;
              else *reinterpret_cast<u32x2*>((uint16_t*)d.c + o) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
          }
        }
      };
      switch (d.epilogue) {
        case VITA_EPI_GELU: run(std::integral_constant<int, VITA_EPI_GELU>{}); break;
        case VITA_EPI_RESIDUAL: run(std::integral_constant<int, VITA_EPI_RESIDUAL>{}); break;
        case VITA_EPI_GELU_BWD: run(std::integral_constant<int, VITA_EPI_GELU_BWD>{}); break;
        default: run(std::integral_constant<int, VITA_EPI_NONE>{}); break;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // staging reads done before the next pass overwrites the slice
  }
}

// ---- host -------------------------------------------------------------------------------------------
template <int LA, int LB, int NF, bool ATOMIC, int NWM = 4>
static int g2_launch(const vita_gemm_desc& d, hipStream_t s) {
  constexpr int BM = 64 * NWM, NWAVE = 2 * NWM;
  Gemm2Params P;
  P.d = d;
  P.tiles_m = (int)((d.m + BM - 1) / BM);
  P.tiles_n = (int)((d.n + NF * 16 - 1) / (NF * 16));
  P.group_m = 8;    // measured flat between 4 and 8 on the model's shapes, worse at 2, 16 and 32
  const int kt_total = (int)((d.k + G2_BK - 1) / G2_BK);
  P.k_tiles_per_split = (kt_total + d.split_k - 1) / d.split_k;
  const int splits = (kt_total + P.k_tiles_per_split - 1) / P.k_tiles_per_split;
  dim3 grid((unsigned)(P.tiles_m * P.tiles_n), (unsigned)splits, 1);
  // = 2 * (G2Loader<LA, BM, NWAVE>::BYTES + G2Loader<LB, NF * 16, NWAVE>::BYTES), spelled out: naming the device-side
  // loader class here makes hipcc's host pass drop launch stubs
  constexpr int npa = LA == 0 ? BM / 8 : 32, npb = LB == 0 ? NF * 2 : 32;
  constexpr size_t stage_bytes = 2 * 1024 * (size_t)(((npa + NWAVE - 1) / NWAVE + (npb + NWAVE - 1) / NWAVE) * NWAVE);
  constexpr size_t epi_bytes = NWAVE * 32 * (((NF + 1) / 2) * 16 + 4) * 4;
  const size_t lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static VitaPerDeviceOnce once;
  vita_set_max_lds_once(once, (const void*)gemm256_kernel<LA, LB, NF, ATOMIC, NWM>, (int)lds);
  hipLaunchKernelGGL((gemm256_kernel<LA, LB, NF, ATOMIC, NWM>), grid, dim3(NWM * 128), lds, s, P);
  VITA_CHECK_LAUNCH("vita_gemm_bf16(256)");
  return VITA_OK;
}

// Cost model (arbitrary units, one unit ~ 16 output columns of a 256-row tile over the whole K):
// rounds of 256 workgroups x (tile width + fixed per-tile overhead).  *nf_out = best tile width.
bool vita_gemm128x_supported(const vita_gemm_desc& d);
double vita_gemm256_cost(const vita_gemm_desc& d, int* nf_out) {
  const int64_t tm = (d.m + G2_BM - 1) / G2_BM;
  const int kt_total = (int)((d.k + G2_BK - 1) / G2_BK);
  const int per = (kt_total + d.split_k - 1) / d.split_k, splits = (kt_total + per - 1) / per;
  const int cand[6] = {13, 14, 16, 8, 10, 12};
  // The 128..192-column tiles pay where a GEMM has the chip to itself and its 208..256-column grid ends in a thin last round
  // (forward GEMMs with N = 1088 at M = 13826: proj 76 -> 66 us, fc2 171 -> 149 us).  The dgrad GEMMs (k-strided B) run
  // beside the weight-gradient GEMMs of the second stream, which already fill such a round: there the narrow tiles
  // measured +1.5 ms per step, so they are not candidates.
  static const bool wide_only = getenv("VITA_GEMM_WIDE_ONLY") && getenv("VITA_GEMM_WIDE_ONLY")[0] == '1';
  static const bool no_128 = getenv("VITA_GEMM_NO128") && getenv("VITA_GEMM_NO128")[0] == '1';
  const int ncand = (wide_only || d.a_layout != 0 || d.b_layout != 0) ? 3 : 6;
  double best = 0; int best_nf = 0;
  for (int c = 0; c < ncand; ++c) {
    const int64_t tn = (d.n + cand[c] * 16 - 1) / (cand[c] * 16);
    const int64_t rounds = (tm * tn * splits + 255) / 256;
    // fixed part per tile.  The 208 / 224-column tiles also carry their lower main-loop efficiency (8192^3: 1201 / 1217
    // against 1403 TFLOP/s at 256 columns, i.e. 13 -> 15.2 and 14 -> 16.1 units of work): without it the model preferred
    // 224 columns wherever they saved a round (fc1 at M = 46082: 576 us against 518 us with 256); step -0.5 % at T = 50,
    // -2.5 % in Stage II.  (Narrow tiles for the dgrad GEMMs re-measured in round 2: still +2.5 % on the step.)
    const double fixed = cand[c] < 13 ? G2_NARROW_OVERHEAD : cand[c] == 13 ? 4.2 : cand[c] == 14 ? 4.1 : 2.0;
    const double cost = (double)rounds * (cand[c] + fixed) / splits;
    if (best_nf == 0 || cost < best) { best = cost; best_nf = cand[c]; }
  }
  // 128-row tiles, two workgroups per CU (512 slots; a slot advances at half a CU's rate, a tile has half the rows: one
  // round of 512 costs what a round of 256-row tiles of the same width costs, fixed part 2.5 units - fitted at
  // M = 13826 / 46082, N = 1088..4160: qkv 116.7 -> 109.8 us, proj 65.7 -> 58.2, fc2 137.4 -> 118.9 at M = 13826)
  if (!no_128 && !wide_only && vita_gemm128x_supported(d)) {
    const int64_t tm1 = (d.m + 127) / 128;
    for (int nf = 8; nf <= 12; nf += 2) {
      const int64_t tn = (d.n + nf * 16 - 1) / (nf * 16);
      const int64_t rounds = (tm1 * tn + 511) / 512;
      const double cost = (double)rounds * (nf + 2.5);
      if (cost < 0.97 * best) { best = cost; best_nf = 100 + nf; }
    }
  }
  if (nf_out) *nf_out = best_nf;
  return best;
}
// the 128x128 kernel in the same units: a CU works through ceil(tiles / 256) of them, one costing about 8.75 units
// (two co-resident tiles share a CU: a full round of 512 costs 15-19 units on the model's shapes - M = 13826: proj
// 18.7, fc2 19, dln2 16.4, dln1 15.2, do 16, measured against the 256-row kernel's rounds)
double vita_gemm128_cost(const vita_gemm_desc& d) {
  const int64_t t = ((d.m + 127) / 128) * ((d.n + 127) / 128) * d.split_k;
  return (double)((t + 255) / 256) * 8.75 / d.split_k;
}
#if G2_DBG
// marker of a timing-only build (vita_amd/_lib.py warns when it finds it)
extern "C" int vita_debug_build_gemm(void) { return G2_DBG; }
#endif
// Whether the large-tile kernel can run this descriptor (already validated / normalised by vita_gemm_bf16).
bool vita_gemm256_supported(const vita_gemm_desc& d) {
  if (d.batch_outer * d.batch_inner != 1) return false;
  if (d.a_layout == 1 && d.b_layout == 0) return false;
  const bool scalar_path = d.split_k > 1 || d.row_seg_phys || d.col_seg_phys || (d.n & 3);
  if (scalar_path) { if (!(d.a_layout == 1 && d.b_layout == 1) || d.c_dtype != VITA_DT_F32) return false; }
  else if (d.bias && ((uintptr_t)d.bias & 15)) return false;
  const int64_t a_rows = d.a_layout == 0 ? G2_BM : d.k, b_rows = d.b_layout == 0 ? 256 : d.k;
  if (a_rows * d.lda * 2 >= (int64_t)0x7fffff00ll || b_rows * d.ldb * 2 >= (int64_t)0x7fffff00ll) return false;
  return true;
}

// whether the 128-row, two-workgroups-per-CU variant (nf = 100 + tile width in fragments) can run this descriptor
bool vita_gemm128x_supported(const vita_gemm_desc& d) {
  const bool scalar_path = d.split_k > 1 || d.row_seg_phys || d.col_seg_phys || (d.n & 3);
  return vita_gemm256_supported(d) && !scalar_path && d.a_layout == 0 && d.b_layout == 0;
}

int vita_gemm256_launch(const vita_gemm_desc& d, int nf, hipStream_t s) {
  const bool scalar_path = d.split_k > 1 || d.row_seg_phys || d.col_seg_phys || (d.n & 3);
  if (nf == 0) (void)vita_gemm256_cost(d, &nf);
  if (nf > 100) {
    if (!vita_gemm128x_supported(d)) {
      vita_set_error("vita_gemm_bf16: the 128-row kernel needs k-contiguous operands and a plain epilogue path", hipSuccess);
      return VITA_ERR_UNSUPPORTED;
    }
    if (nf == 108) return g2_launch<0, 0, 8, false, 2>(d, s);
    if (nf == 110) return g2_launch<0, 0, 10, false, 2>(d, s);
    if (nf == 112) return g2_launch<0, 0, 12, false, 2>(d, s);
  }
#define G2_CASE(LA, LB, AT)                                      \
  do {                                                           \
    if (nf == 8) return g2_launch<LA, LB, 8, AT>(d, s);          \
    if (nf == 10) return g2_launch<LA, LB, 10, AT>(d, s);        \
    if (nf == 12) return g2_launch<LA, LB, 12, AT>(d, s);        \
    if (nf == 13) return g2_launch<LA, LB, 13, AT>(d, s);        \
    if (nf == 14) return g2_launch<LA, LB, 14, AT>(d, s);        \
    if (nf == 16) return g2_launch<LA, LB, 16, AT>(d, s);        \
  } while (0)
  if (scalar_path) G2_CASE(1, 1, true);
  else if (d.a_layout == 0 && d.b_layout == 0) G2_CASE(0, 0, false);
  else if (d.a_layout == 0 && d.b_layout == 1) G2_CASE(0, 1, false);
  else if (d.a_layout == 1 && d.b_layout == 1) G2_CASE(1, 1, false);
#undef G2_CASE
  vita_set_error("vita_gemm_bf16: no large-tile kernel for this descriptor", hipSuccess);
  return VITA_ERR_UNSUPPORTED;
}

// Explicit instantiation of every kernel the launcher can pick: hipcc's host pass (ROCm 7.2) drops the launch stubs of
// most implicit instantiations of this template (only the first one referenced from a function survived).
#define G2_INST(LA, LB, AT, NWM)                                               \
  template __global__ void gemm256_kernel<LA, LB, 8, AT, NWM>(const Gemm2Params);  \
  template __global__ void gemm256_kernel<LA, LB, 10, AT, NWM>(const Gemm2Params); \
  template __global__ void gemm256_kernel<LA, LB, 12, AT, NWM>(const Gemm2Params);
#define G2_INST_WIDE(LA, LB, AT)                                               \
  template __global__ void gemm256_kernel<LA, LB, 13, AT, 4>(const Gemm2Params);   \
  template __global__ void gemm256_kernel<LA, LB, 14, AT, 4>(const Gemm2Params);   \
  template __global__ void gemm256_kernel<LA, LB, 16, AT, 4>(const Gemm2Params);
G2_INST(0, 0, false, 4) G2_INST_WIDE(0, 0, false)
G2_INST(0, 1, false, 4) G2_INST_WIDE(0, 1, false)
G2_INST(1, 1, false, 4) G2_INST_WIDE(1, 1, false)
G2_INST(1, 1, true, 4) G2_INST_WIDE(1, 1, true)
G2_INST(0, 0, false, 2)
#undef G2_INST
#undef G2_INST_WIDE
