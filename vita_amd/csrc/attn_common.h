// Device helpers shared by the fused-attention kernels (attention.hip: 32x32x16 MFMA tiles; attention16.hip:
// 16x16x32 tiles): LDS image geometry, LDS-DMA pieces, counted waits, transposed LDS reads.
#pragma once
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

// Timing-only dissection builds of the dK/dV kernel (tools/attn_dissect.sh; results are INVALID when non-zero):
// 1 = no dS stores, 2 = no in-loop LDS-DMA issue, 4 = no per-tile barrier, 8 = no exp, 16 = no P hand-off through LDS,
// 32 = no in-loop wait for tile arrival, 64 = LDS operand reads of the chains cut to the first 3 (rows) / 4 (transposed),
// 256 = Q / dO LDS-DMA for every other tile only, 512 = two 16x16x32 MFMAs in place of each 32x32x16 (see mfma32_or_proxy)
#ifndef ATT_DBG
#define ATT_DBG 0
#endif

template <int HDP> struct AttCfg {
  static constexpr int RS = (HDP * 2 <= 256) ? 256 : 512;   // LDS row stride in bytes
  static constexpr int CPR = RS / 16;                        // 16-B chunks per LDS row (incl. pad)
  static constexpr int RPI = 1024 / RS;                      // rows per DMA wave-instruction
  static constexpr int TILE = 32 * RS;                       // one 32-row image
  static constexpr int NDT = HDP / 32;
};

__device__ __forceinline__ int xr(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ uint32_t u_off(int row, int byte, int RS) {
  return row * RS + ((((byte >> 4) ^ xr(row))) << 4) + (byte & 15);
}

__device__ __forceinline__ bf16x8 tr_pair(const char* base, uint32_t o0, uint32_t o1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + o1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {   // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ bf16x8 pack8(const float* p) {
  u32x4 u = {cvt_pk(p[0], p[1]), cvt_pk(p[2], p[3]), cvt_pk(p[4], p[5]), cvt_pk(p[6], p[7])};
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // bare v_exp_f32

// Per-lane LDS byte offsets inside one 32-row image, arranged so that every access in the tile loop is
// base register + compile-time immediate (no address arithmetic in the loop):
//   row read, k-step s      : rb[s & 7] + 256 * (s >> 3)
//   transposed read (i,s2,u): tb[i & 3][u] + 256 * (i >> 2) + (16*s2 + 8*u) * RS     (u = second half)
template <int HDP, int NKS>
struct LdsAddr {
  static constexpr int RS = AttCfg<HDP>::RS;
  static constexpr int NRB = NKS < 8 ? NKS : 8;
  static constexpr int NTB = AttCfg<HDP>::NDT < 4 ? AttCfg<HDP>::NDT : 4;
  uint32_t rb[NRB];
  uint32_t tb[NTB][2];
  __device__ __forceinline__ void init(int lane) {
    const int lrow = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NRB; ++j) rb[j] = lrow * RS + ((((2 * j + hh) ^ xr(lrow)) & 15) << 4);
    const int gi = lane >> 4, t = lane & 15, tq = t >> 2, tp = t & 3;
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int xlow = ((gi >> 1) + 2 * u) & 3;
        tb[j][u] = (4 * (gi >> 1) + tq) * RS + ((4 * (j ^ tq) + ((2 * (gi & 1) + (tp >> 1)) ^ xlow)) << 4) + 8 * (tp & 1);
      }
  }
  __device__ __forceinline__ void toggle(int bytes) {   // switch ring buffer (bytes = power of two)
#pragma unroll
    for (int j = 0; j < NRB; ++j) rb[j] ^= bytes;
#pragma unroll
    for (int j = 0; j < NTB; ++j) { tb[j][0] ^= bytes; tb[j][1] ^= bytes; }
  }
  __device__ __forceinline__ void advance_rows(int bytes) {   // row-read bases to another ring slot
#pragma unroll
    for (int j = 0; j < NRB; ++j) rb[j] += bytes;
  }
  __device__ __forceinline__ void advance_tr(int bytes) {     // transposed-read bases to another ring slot
#pragma unroll
    for (int j = 0; j < NTB; ++j) { tb[j][0] += bytes; tb[j][1] += bytes; }
  }
  __device__ __forceinline__ void advance(int bytes) { advance_rows(bytes); advance_tr(bytes); }
  __device__ __forceinline__ bf16x8 row(const char* img, int s) const {
    return *reinterpret_cast<const bf16x8*>(img + rb[s & 7] + 256 * (s >> 3));
  }
  __device__ __forceinline__ bf16x8 tr(const char* img, int i, int s2) const {
    return tr_pair(img, tb[i & 3][0] + 256 * (i >> 2) + (16 * s2) * RS, tb[i & 3][1] + 256 * (i >> 2) + (16 * s2 + 8) * RS);
  }
  // The same read as inline asm (IMG = byte offset of the image inside a ring slot; dynamic LDS starts at
  // LDS address 0 in these kernels).  hipcc puts s_waitcnt vmcnt(0) in front of the ds_read_b64_tr_b16
  // builtin whenever an LDS-DMA is in flight; the asm form is invisible to that pass, so the caller waits
  // for the data with a counted lgkmcnt (tr_wait) followed by a sched_barrier.
  template <int I, int S2, int IMG>
  __device__ __forceinline__ bf16x8 tr_asm() const {
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    v2i lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(tb[I & 3][0]), "n"(IMG + 256 * (I >> 2) + (16 * S2) * RS) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(tb[I & 3][1]), "n"(IMG + 256 * (I >> 2) + (16 * S2 + 8) * RS) : "memory");
    v4i v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
  }
};
// wait until at most N LDS operations of this wave are outstanding, and keep the MFMAs that consume
// asm-read fragments behind the wait
template <int N> __device__ __forceinline__ void tr_wait() {
  __builtin_amdgcn_s_waitcnt(0xc07f | (N << 8));
  __builtin_amdgcn_sched_barrier(0);
}
// an empty asm that uses every accumulator tile as an AGPR operand
template <int N> __device__ __forceinline__ void pin_agpr(f32x16 (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+a"(acc[i]));
}
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}


// Two independent row-read -> MFMA accumulation chains (S and dP), software pipelined DEPTH k-steps
// ahead so the ds_read_b128 latency is covered by the other chain's MFMAs; the order is pinned with
// sched_barrier because hipcc otherwise serialises read -> wait -> MFMA when registers are tight.
template <int HDP, int NKS, int DEPTH, class F>
__device__ __forceinline__ void dual_chain(f32x16& acc0, f32x16& acc1, const char* img0, const char* img1,
                                           const LdsAddr<HDP, NKS>& la, const bf16x8 (&b0)[NKS], const bf16x8 (&b1)[NKS],
                                           F&& between) {
  bf16x8 a0[DEPTH], a1[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH && j < NKS; ++j) { a0[j] = la.row(img0, j); a1[j] = la.row(img1, j); }
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s % DEPTH], b0[s], acc0, 0, 0, 0);
    if (s + DEPTH < NKS) a0[s % DEPTH] = la.row(img0, s + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s % DEPTH], b1[s], acc1, 0, 0, 0);
    if (s + DEPTH < NKS) a1[s % DEPTH] = la.row(img1, s + DEPTH);
    between(s);     // e.g. one LDS-DMA piece of a later tile: its issue time hides under the two MFMAs
  }
  __builtin_amdgcn_sched_barrier(0);
}


// ATT_DBG & 512 (timing only, results INVALID): every 32x32x16 MFMA of the dK/dV tile loop replaced by two 16x16x32 MFMAs on
// the same operand registers - the same FLOPs, operand reads and issue slots a 16x16x32 re-tiling would have - to price
// that re-tiling before building it.
__device__ __forceinline__ f32x16 mfma32_or_proxy(const bf16x8 a, const bf16x8 b, f32x16 acc) {
  if constexpr (ATT_DBG & 512) {
    f32x4 lo = {acc[0], acc[1], acc[2], acc[3]}, hi = {acc[4], acc[5], acc[6], acc[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, hi, 0, 0, 0);
    acc[0] = lo[0]; acc[1] = lo[1]; acc[2] = lo[2]; acc[3] = lo[3];
    acc[4] = hi[0]; acc[5] = hi[1]; acc[6] = hi[2]; acc[7] = hi[3];
    return acc;
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}

template <int HDP, int NKS, int DEPTH, class F>
__device__ __forceinline__ void single_chain(f32x16& acc, const char* img, const LdsAddr<HDP, NKS>& la, const bf16x8 (&b)[NKS],
                                             F&& between) {
  bf16x8 a[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH && j < NKS; ++j) a[j] = la.row(img, j);
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    __builtin_amdgcn_sched_barrier(0);
    acc = mfma32_or_proxy(a[s % DEPTH], b[s], acc);
    if (s + DEPTH < NKS && !(ATT_DBG & 64)) a[s % DEPTH] = la.row(img, s + DEPTH);
    between(s);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// DMA one 32-row image (rows row0.. of a [*, ld] bf16 matrix seen through rs) into lds_img.
// NW waves, each issues NI = 32/RPI/NW wave-instructions ("pieces") of 1 KiB; dma_piece issues piece j.
template <int HDP, int NW = 4>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, int row0, int64_t ld, char* lds_img, int wave, int lane, int j) {
  using C = AttCfg<HDP>;
  constexpr int NI = 32 / C::RPI / NW;
  typedef __attribute__((address_space(3))) void lds_void;
  const int piece = wave * NI + j;
  const int row = piece * C::RPI + lane / C::CPR;
  const int cphys = lane % C::CPR;
  const int clog = cphys ^ xr(row);
  const uint32_t voff = (uint32_t)(((int64_t)(row0 + row) * ld) * 2 + clog * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds_img + piece * 1024), 16, voff, 0, 0, 0);
}
template <int HDP, int NW = 4>
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rs, int row0, int64_t ld, char* lds_img, int wave, int lane) {
  constexpr int NI = 32 / AttCfg<HDP>::RPI / NW;
#pragma unroll
  for (int j = 0; j < NI; ++j) dma_piece<HDP, NW>(rs, row0, ld, lds_img, wave, lane, j);
}
// Per-lane source offsets of a wave's NI pieces of a 32-row image, computed once; a tile is then
// base offset (wave-uniform) + vb[j]: one VALU add per piece instead of the 64-bit address arithmetic.
template <int HDP, int NW = 4>
struct DmaOff {
  static constexpr int NI = 32 / AttCfg<HDP>::RPI / NW;
  uint32_t vb[NI];
  __device__ __forceinline__ void init(int64_t ld, int wave, int lane) {
    using C = AttCfg<HDP>;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int piece = wave * NI + j;
      const int row = piece * C::RPI + lane / C::CPR;
      const int clog = (lane % C::CPR) ^ xr(row);
      vb[j] = (uint32_t)(row * ld * 2 + clog * 16);
    }
  }
  // tile_off = byte offset of the tile's first row (wave-uniform)
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, uint32_t tile_off, char* lds_img, int wave, int j) const {
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds_img + (wave * NI + j) * 1024), 16, vb[j] + tile_off, 0, 0, 0);
  }
};
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's youngest vector-memory operations are still in flight
template <int N> __device__ __forceinline__ void dma_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// barrier that does not drain the LDS-DMA queue (a __syncthreads() would add vmcnt(0))
__device__ __forceinline__ void ring_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// Workgroup id -> work item so that the ids an XCD receives (id % 8, round-robin dispatch) form one contiguous range of
// work items: the workgroups sharing an XCD's L2 then sweep the same head's K/V (resp. Q/dO) stream.
__device__ __forceinline__ int xcd_contiguous(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
#define ATT_NSLOT 3
#define ATT_BWD_NSLOT 4   // the software-pipelined dK/dV kernel keeps tiles t .. t+3 in the ring
#define ATT_DQDS_NSLOT 5  // ring depth of the streaming dQ kernel (K image + 8 dS tiles = 32 KiB per slot at hdp 224: all 160 KiB)

template <int HDP>
__device__ __forceinline__ void lds_zero(char* smem, int bytes, int tid) {
  for (int o = tid * 16; o < bytes; o += (int)blockDim.x * 16) *reinterpret_cast<u32x4*>(smem + o) = u32x4{0, 0, 0, 0};
}

// ATT_STAMP=1: diagnostic build - every wave sums the shader cycles of each phase of its tile loop (s_memtime) and leaves
// the sums in att_stamp[workgroup][wave][8]; read them with vita_attn_debug_stamps().  Never set in the shipped library.
#ifndef ATT_STAMP
#define ATT_STAMP 0
#endif
#if ATT_STAMP
#define ATT_STAMP_WG 4096
static __device__ unsigned long long att_stamp[ATT_STAMP_WG * 8 * 8];
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define STAMP(k) do { const unsigned long long n__ = stamp_now(); st_sum[k] += n__ - st_last; st_last = n__; } while (0)
#ifndef ATT_STAMP_READER
#define ATT_STAMP_READER vita_attn_debug_stamps
#endif
extern "C" int ATT_STAMP_READER(unsigned long long* host_out, int n_wg) {
  if (n_wg > ATT_STAMP_WG) n_wg = ATT_STAMP_WG;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(att_stamp), (size_t)n_wg * 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define STAMP(k) do { } while (0)
#endif

// ============================================================================================ host helpers
#define ATT_MAX_CU 320
static inline bool att_env_flag(const char* name) {
  const char* e = getenv(name);
  return e && e[0] == '1';
}
static inline int att_num_cu() {
  static int n[VITA_MAX_DEVICES];          // per device; written once under the per-device once flag
  const int dev = vita_cur_device();
  VITA_ONCE_PER_DEVICE({
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n[dev] = v > ATT_MAX_CU ? ATT_MAX_CU : v;
  });
  return n[dev];
}
// These grids run one 8-wave workgroup per CU and every workgroup of a launch does the same work, so a grid of L
// workgroups costs ceil(L / CUs) rounds.  When the last round is less than 3/4 full its `rem` workgroups are launched
// separately with their sweep cut into nsplit ranges (>= 8 tiles per range, rem * nsplit <= slots partial results).  The
// pieces may themselves take several rounds: the split that minimises ceil(rem * nsplit / CUs) / nsplit wins (142 left-over
// workgroups of 721 tiles: 3 ranges = 426 pieces = 2 rounds of a third = 0.67 of a round instead of a whole one).
static inline void att_tail_plan(int64_t L, int64_t ntiles, int ncu, int64_t slots, int64_t* l_main, int* nsplit, int* tiles_per_split) {
  *l_main = L; *nsplit = 1; *tiles_per_split = (int)ntiles;
  const int64_t rem = L % ncu;
  if (rem == 0 || rem * 4 > (int64_t)ncu * 3) return;
  double best = 0.95;                      // an unsplit tail costs one round; a split must save at least 5 % of it
  int64_t best_per = 0;
  for (int64_t ns = 2; ns <= ntiles / 8 && rem * ns <= slots; ++ns) {
    const int64_t per = (ntiles + ns - 1) / ns, nse = (ntiles + per - 1) / per;
    const double cost = (double)((rem * nse + ncu - 1) / ncu) * (double)per / (double)ntiles + 0.004 * (double)nse;
    if (cost < best) { best = cost; best_per = per; }
  }
  if (best_per == 0) return;
  *l_main = L - rem; *nsplit = (int)((ntiles + best_per - 1) / best_per); *tiles_per_split = (int)best_per;
}
#define ATT_FWD_SLOTS 640      // partial outputs of the forward tail launch (256 query rows each)
static inline int64_t att_fwd_part_bytes(int64_t hdp) { return (int64_t)ATT_FWD_SLOTS * 256 * (hdp + 1) * 4; }


// entry points of attention16.hip (16x16x32 MFMA tiles) and the shared tail-combine launch of attention.hip
int att16_launch_fwd(int hdp, int nks, const void* qkv, void* o, float* lse, void* ws, int64_t B, int64_t N, int64_t H, float scale,
                     hipStream_t s);
void att_launch_fwd_combine(const float* part, void* o, float* lse, int wg_base, int nwg, int nsplit, int N, int H, int hdp, int qbs,
                            hipStream_t s);
