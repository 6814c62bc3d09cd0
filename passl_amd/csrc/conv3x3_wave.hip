// 3x3 / stride-1 / pad-1 convolution with 64 input and 64 output channels, bf16 (the conv2 layers of ResNet-50's first
// stage, forward and data gradient) — one WAVE per 8 x 8 output patch, no workgroup barrier in the loop (gfx950).
//
// Why another kernel (round 6).  The LDS-DMA ring kernel (conv_igemm_ring.hip) takes 95-120 us for this layer against a
// traffic floor of 38 us (206 MB at the 5.4 TB/s a mixed read / write stream reaches on this chip,
// profiles/r06_store_probe.txt) and an MFMA floor of 30 us: it fetches every input pixel nine times through L2 (216 KB
// of LDS-DMA per 128-row tile), re-stages the weights for every tile, and a tile's life is setup + operand wait + nine
// barrier-separated K-tiles + an LDS-staged epilogue that nothing overlaps (2.6 + 0.8 + 4.2 + 2.6 us).
// Rounds 4-5 measured three spatially tiled forms of the same WORKGROUP-per-tile structure; none beat it.  Here the
// structure itself goes:
//
//  * the nine weight taps (64 x 576 bf16 = 72 KB) are staged ONCE per workgroup and stay in LDS (rows of 128 B, the
//    ring kernel's XOR swizzle); workgroups are persistent (one per CU);
//  * each of a workgroup's four waves (one per SIMD, up to 512 registers) owns a private 16 KB halo buffer and walks
//    its own list of 8 x 8 patches: [write the 10 x 10 x 64 halo of patch t from staging registers into LDS] ->
//    [request the halo of patch t+1 into the staging registers: in flight under everything below] -> [9 taps x 2
//    k-steps x 16 MFMAs straight out of LDS: a tap is an immediate offset, the zero padding is physically in the halo]
//    -> [epilogue from the accumulators: affine / ReLU, bf16 pack, v_permlane32_swap to 16-byte pieces, stores,
//    statistics in registers].  No s_barrier, no LDS-DMA wait, no LDS-staged output: the four waves drift apart and
//    cover each other's epilogues and load waits;
//  * halo pixel pitch 160 B (8 data + 2 pad 16-byte slots) and the lane -> pixel map {lanes 0-3, 12-15: first row of a
//    fragment's 2 x 8 pixels, lanes 4-11: second row} put the 16 lanes of every ds_read_b128 service group on 16 different
//    bank quads for every tap (10 c mod 16 runs through the even residues; the other k-group of the group sits one slot
//    further: the odd ones);
//  * arithmetic: taps in (r, s) order, channels ascending, fp32 accumulation in the MFMA — the ring kernel's order:
//    the stored bf16 output is bit-identical to it.  The fused statistics (forward: shifted sums per 128 rows; data
//    gradient: BatchNorm-backward sums of the producing layer, csrc/igemm_epi.h) are summed in a different, equally
//    fixed order: deterministic, equal to the staged epilogue's to fp32 rounding.  A slab row is two consecutive patches
//    (128 pixels) — bn_finalize only needs every row to hold 128 pixels, not which ones.
//
// Envelope (passl_conv3x3_wave_try returns PASSL_EUNSUPPORTED outside it and the ring kernel takes the launch):
// R = S = 3, stride 1, pad 1, C = NCOLS = 64, bf16 in / out, dense NHWC operands, IH and IW multiples of 8, no residual;
// epilogues: affine (+ReLU), forward statistics, BatchNorm-backward statistics with the ReLU mask recomputed from y
// (bnb_relu 0 or 2).  Option conv3x3_wave = 0 / 1 (passl_hip_set_option, PASSL_CONV3X3_WAVE).
#include <stdlib.h>
#include <string.h>
#include "igemm_dma.h"
#include "igemm_epi.h"

namespace w3 {

using ring::bf16x8_t;
using ring::u32x4;
using ring::FastDiv;
using ring::fdiv;
using ring::make_fastdiv;
using ring::kOOB;

constexpr int kC = 64;                     // input channels = output channels
constexpr int kPix = 160;                  // bytes per halo pixel: 128 data + 32 pad
constexpr int kHaloW = 10;                 // 8 + 2 pixels per halo row
// ROWS = output rows of a wave's patch (8 columns wide): 8 -> four waves per workgroup (one per SIMD), 4 -> eight waves
// (two per SIMD: one wave's epilogue / halo staging under the other's MFMAs; the 32-pixel patches re-read more halo)
template <int ROWS> struct Geo {
  static constexpr int kWaves = ROWS == 8 ? 4 : 8;
  static constexpr int kThreads = kWaves * 64;
  static constexpr int kHaloPx = (ROWS + 2) * kHaloW;
  static constexpr int kHaloStride = (kHaloPx * kPix + 255) / 256 * 256;      // per-wave halo buffer
  static constexpr int kLoads = (kHaloPx * 8 + 63) / 64;                        // 16-byte halo chunks per lane
  static constexpr int kFrags = ROWS / 2;                                       // 2 x 8-pixel fragments per patch
  static constexpr int kPerSlab = 128 / (ROWS * 8);                             // patches per 128-pixel slab row
};
constexpr int kTapBytes = kC * 128;        // one tap of the weights: 64 rows x 128 B
constexpr int kWBytes = 9 * kTapBytes;     // 73 728

struct Params {
  const char* a;
  const char* b;
  char* y;
  const float* scale;
  const float* shift;
  float* stats;
  int stats_tiles;
  const char* bnb_y;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_scale;
  const float* bnb_shift;
  float* bnb_partial;
  int bnb_relu, bnb_tile_off;
  uint32_t a_bytes;
  int N, IH, IW, PH, PW;                   // PH x PW patches per image
  int npatches, nslabs;                    // slab row = 128 pixels = 2 (ROWS 8) / 4 (ROWS 4) consecutive patches
  int relu;
  int dbg;                                 // ablation (conv3x3_wave_dbg): 1 no statistics, 2 no epilogue, 4 no MFMA loop, 8 no halo loads
  FastDiv d_pp, d_pw;                      // / (PH * PW), / PW
};

template <int OFF>
__device__ __forceinline__ u32x4 lds_rd(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// MODE 0: affine (+ReLU); 1: + forward statistics; 2: BatchNorm-backward statistics of the producing layer.
// AFFINE / RELU are compile-time: as run-time branches they cost the epilogue 512 accumulator-register moves and 32
// loads per patch (the compiler keeps every branch's result in the accumulator file)
template <int MODE, bool AFFINE, bool RELU, int ROWS>
__global__ void __launch_bounds__(Geo<ROWS>::kThreads, ROWS == 8 ? 1 : 2) conv3x3_wave_kernel(const Params p) {
  typedef Geo<ROWS> G;
  constexpr int kThreads = G::kThreads, kLoads = G::kLoads, kHaloStride = G::kHaloStride, FI = G::kFrags;
  extern __shared__ __attribute__((aligned(256))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);

  // ---- per-lane geometry of the halo staging: chunk q = lane + 64 i -> halo pixel q >> 3, 16-byte piece q & 7
  char* halo = smem + kWBytes + wave * kHaloStride;
  const uint32_t halo0 = lds0 + kWBytes + wave * kHaloStride;
  int rel[kLoads];                          // byte offset of the chunk relative to the patch's first pixel
  int hyx[kLoads];                          // (hy << 8) | hx, or -1 for the 32 chunks past the halo
  uint32_t hdst[kLoads];
#pragma unroll
  for (int i = 0; i < kLoads; ++i) {
    const int q = lane + 64 * i;
    const int hq = q >> 3;
    const int hy = hq / kHaloW, hx = hq - hy * kHaloW;
    rel[i] = ((hy - 1) * p.IW + (hx - 1)) * (kC * 2) + (q & 7) * 16;
    hyx[i] = hq < G::kHaloPx ? ((hy << 8) | hx) : -1;
    hdst[i] = (uint32_t)(hq * kPix + (q & 7) * 16);
  }
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, p.a_bytes, 0x00020000);

  // ---- fragment read addresses.  Pixel of lane l15 inside a fragment's 2 x 8 block: lanes 0-3 / 12-15 -> first row,
  // columns 0-3 / 4-7; lanes 4-11 -> second row, columns 0-7
  const int fr = (l15 >= 4 && l15 < 12) ? 1 : 0;
  const int fc = l15 < 4 ? l15 : (l15 < 12 ? l15 - 4 : l15 - 8);
  const uint32_t a_rd = halo0 + (uint32_t)((fr * kHaloW + fc) * kPix + l4 * 16);
  // weights: fragment row of lane l15 with bits 2 and 3 exchanged (the 8-phase kernel's direct epilogue): accumulator
  // register r of lane (l15, l4) is output channel {0, 8, 4, 12}[l4] + r of its 16-channel fragment
  const int pl = (l15 & 3) | ((l15 & 4) << 1) | ((l15 & 8) >> 1);
  uint32_t b_rd[2], b_rd_hi[2];             // (the ds_read offset field holds 16 bits: taps 5-8 get a base of their own)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    b_rd[ks] = lds0 + (uint32_t)(pl * 128 + (((ks * 4 + l4) ^ ((pl >> 1) & 7)) << 4));
    b_rd_hi[ks] = b_rd[ks] + 5 * kTapBytes;
  }

  // ---- this wave's slab rows: workgroup iteration u = blockIdx.x + k * gridDim.x through the XCD-aware map, four
  // consecutive slab rows (eight consecutive patches) per workgroup iteration
  const int units = (p.nslabs + G::kWaves - 1) / G::kWaves;
  auto unit_of = [&](int v) __attribute__((always_inline)) {
    const int xcd = v & 7, local = v >> 3;
    const int q = units >> 3, r = units & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
  };
  int v = blockIdx.x;
  auto slab_of = [&](int vv) __attribute__((always_inline)) { return vv < units ? unit_of(vv) * G::kWaves + wave : p.nslabs; };
  int slab = slab_of(v);
  int patch = slab * G::kPerSlab;           // the patch being MULTIPLIED
  const bool has_work = patch < p.npatches;

  // ---- request a patch's halo into the staging registers
  u32x4 stage[kLoads];
  auto request = [&](int pt) __attribute__((always_inline)) {
    const int n = fdiv(pt, p.d_pp);
    const int rem = pt - n * (p.PH * p.PW);
    const int py = fdiv(rem, p.d_pw), px = rem - py * p.PW;
    const uint32_t base = (uint32_t)(((n * p.IH + py * ROWS) * p.IW + px * 8) * (kC * 2));
    const int y0 = py * ROWS - 1, x0 = px * 8 - 1;
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
      const bool ok = hyx[i] >= 0 && (uint32_t)(y0 + (hyx[i] >> 8)) < (uint32_t)p.IH &&
                      (uint32_t)(x0 + (hyx[i] & 255)) < (uint32_t)p.IW;
      const uint32_t off = ok ? base + (uint32_t)rel[i] : kOOB;       // out of range -> the descriptor returns zeros
      stage[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, off, 0, 0));
    }
  };
  auto stage_to_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kLoads; ++i)
      if (hyx[i] >= 0) *reinterpret_cast<u32x4*>(halo + hdst[i]) = stage[i];
  };

  // per-lane statistics of the slab row in flight (MODE 1 / 2): 2 column halves x 8 columns
  float s0[8], s1[8], sh[8];
  // the 8 consecutive output channels this lane stores (after the epilogue's trade: see there)
  const int mycol = fr * 32 + l4 * 8;

  if (has_work) request(patch);             // the first halo travels while the weights are staged

  // ---- the nine weight taps, once: global [col][tap][c] -> LDS [tap][col][128 B], slot ^= (col >> 1) & 7
  for (int q = tid; q < 9 * kC * 8; q += kThreads) {
    const int chunk = q & 7, row = (q >> 3) % kC, tap = (q >> 3) / kC;
    const uint4 v = *reinterpret_cast<const uint4*>(p.b + ((size_t)row * 9 + tap) * 128 + chunk * 16);
    *reinterpret_cast<uint4*>(smem + tap * kTapBytes + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = v;
  }
  __syncthreads();                          // the only workgroup barrier of the kernel
  if (!has_work) return;

  for (;;) {
    const bool first_of_slab = (patch % G::kPerSlab) == 0;
    const bool last_of_slab = (patch % G::kPerSlab) == G::kPerSlab - 1 || patch + 1 >= p.npatches;
    // ---- halo of this patch: staging registers -> LDS (the previous patch's fragment reads are retired: lgkmcnt(0) at
    // the end of its last step; LDS operations of one wave execute in order)
    stage_to_lds();
    // ---- the patch after this one (same slab row, or the first of this wave's next slab row)
    int npatch = patch + 1, nv = v;
    if (last_of_slab) {
      nv = v + gridDim.x;
      npatch = slab_of(nv) * G::kPerSlab;
    }
    const bool has_next = npatch < p.npatches;
    if (has_next && !(p.dbg & 8)) request(npatch);          // in flight until the top of the next iteration

    // ---- output coordinates of this patch (and, MODE 2, the BatchNorm-input rows the epilogue needs: requested now)
    const int n = fdiv(patch, p.d_pp);
    const int rem = patch - n * (p.PH * p.PW);
    const int py = fdiv(rem, p.d_pw), px = rem - py * p.PW;
    uint32_t orow[FI][2];                     // element offset of the pixel (row rr of fragment i, this lane's column)
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
        orow[i][rr] = (uint32_t)(((n * p.IH + py * ROWS + 2 * i + rr) * p.IW + px * 8 + fc) * kC);
    uint4 ybn[MODE == 2 ? FI : 1][2];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
          ybn[i][rr] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bnb_y) + orow[i][rr] + mycol);
    }

    // ---- 9 taps x 2 k-steps: 8 fragment reads + 16 MFMAs per step, the reads of step t+1 under the MFMAs of step t
    f32x4 acc[FI][4];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af[3][FI], bf[3][4];            // three fragment sets: the reads run TWO steps ahead of the MFMAs
    auto read_step = [&](auto SET, auto STEP) __attribute__((always_inline)) {
      constexpr int S_ = decltype(SET)::value, T_ = decltype(STEP)::value;
      constexpr int tap = T_ >> 1, ks = T_ & 1;
      constexpr int r = tap / 3, s = tap % 3;
      constexpr int AOFF = (r * kHaloW + s) * kPix + ks * 64;
      af[S_][0] = lds_rd<AOFF>(a_rd);
      af[S_][1] = lds_rd<AOFF + 2 * kHaloW * kPix>(a_rd);
      if constexpr (FI == 4) {
        af[S_][2] = lds_rd<AOFF + 4 * kHaloW * kPix>(a_rd);
        af[S_][3] = lds_rd<AOFF + 6 * kHaloW * kPix>(a_rd);
      }
      constexpr int BOFF = (tap < 5 ? tap : tap - 5) * kTapBytes;
      const uint32_t bb = tap < 5 ? b_rd[ks] : b_rd_hi[ks];
      bf[S_][0] = lds_rd<BOFF>(bb);
      bf[S_][1] = lds_rd<BOFF + 16 * 128>(bb);
      bf[S_][2] = lds_rd<BOFF + 32 * 128>(bb);
      bf[S_][3] = lds_rd<BOFF + 48 * 128>(bb);
    };
    auto mma_step = [&](auto SET) __attribute__((always_inline)) {
      constexpr int S_ = decltype(SET)::value;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[S_][j]),
                                                              __builtin_bit_cast(bf16x8_t, af[S_][i]), acc[i][j], 0, 0, 0);
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the halo writes above are in LDS
    // Reads two steps ahead: with one step of lead the wait at the end of a step still saw part of the LDS latency (8
    // MFMAs = 136 clocks of cover at ROWS 4).  LDS operations of a wave return in order, so `lgkmcnt(NRD)` — NRD = reads
    // per step — means "everything but the newest step's reads has arrived".
    constexpr int NRD = FI + 4;
    read_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    read_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](auto STEP) __attribute__((always_inline)) {
      constexpr int T_ = decltype(STEP)::value;
      if constexpr (T_ + 2 < 18)
        read_step(std::integral_constant<int, (T_ + 2) % 3>{}, std::integral_constant<int, T_ + 2>{});
      __builtin_amdgcn_sched_barrier(0);      // the reads of step t+2 go out BEFORE the MFMAs of step t (hipcc otherwise
                                              // reuses the operand registers and issues them behind the MFMAs)
      mma_step(std::integral_constant<int, T_ % 3>{});
      __builtin_amdgcn_sched_barrier(0);      // ... and the wait comes BEHIND the MFMAs, not among them
      if constexpr (T_ + 2 < 18) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");    // step t+1 has landed
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    if (!(p.dbg & 4)) {
    step(std::integral_constant<int, 0>{});  step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});  step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{});  step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{});  step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});  step(std::integral_constant<int, 9>{});
    step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
    step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{});
    step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
    step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 17>{});
    }

    // ---- epilogue from the accumulators.  Fragment pair jh = (j 2jh, 2jh + 1): after the affine and the bf16 pack,
    // v_permlane32_swap leaves lane (l15, l4) with the 16-byte piece l4 of its pixel's 64-byte half jh.  A store
    // instruction made of those would write 64 B of sixteen different pixels (half lines: measured 36 of the kernel's
    // 95 us).  So the two rows of a fragment trade halves first: lane l15 and lane l15 ^ 4 hold the same column of the
    // fragment's first / second row (lanes 0-3, 12-15 / 4-11); the first-row lanes give away their half 1 and keep
    // half 0 of BOTH rows, the second-row lanes keep half 1 of both — and each of the two store instructions per
    // fragment writes one complete patch row: 8 pixels x 128 B = 1 KB of consecutive memory.  A lane then only ever
    // sees ONE set of 8 output channels: first-row lanes channels l4*8 .., second-row lanes 32 + l4*8 ...
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.y);
    if (!(p.dbg & 2)) {
    if (first_of_slab) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; sh[e] = 0.f; }
    }
    const int pcol = ((l4 & 1) << 3) | ((l4 & 2) << 1);              // {0, 8, 4, 12}[l4]: this lane's own columns, pre-swap
    float4 asc[4], ash[4];                   // per fragment j: scale / shift of this lane's 4 channels
    if constexpr (AFFINE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asc[j] = p.scale ? *reinterpret_cast<const float4*>(p.scale + j * 16 + pcol) : make_float4(1.f, 1.f, 1.f, 1.f);
        ash[j] = p.shift ? *reinterpret_cast<const float4*>(p.shift + j * 16 + pcol) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float c0[8], c1[8], c2[8], c3[8];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        c0[e] = p.bnb_mean[mycol + e]; c1[e] = p.bnb_invstd[mycol + e];
        c2[e] = p.bnb_relu == 2 ? p.bnb_scale[mycol + e] : 0.f;
        c3[e] = p.bnb_relu == 2 ? p.bnb_shift[mycol + e] : 1.f;          // relu 0: y * 0 + 1 > 0 always
      }
    }
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      uint4 half[2];
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        uint32_t w[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          f32x4 a = acc[i][jh * 2 + f];
          if constexpr (AFFINE) {
            const float4 sc = asc[jh * 2 + f], sf = ash[jh * 2 + f];
            a[0] = a[0] * sc.x + sf.x; a[1] = a[1] * sc.y + sf.y;
            a[2] = a[2] * sc.z + sf.z; a[3] = a[3] * sc.w + sf.w;
          }
          if constexpr (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
          }
          w[f][0] = pack2bf(a[0], a[1]);
          w[f][1] = pack2bf(a[2], a[3]);
        }
        const auto x0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto x1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        half[jh] = make_uint4(x0[0], x1[0], x0[1], x1[1]);
      }
      // trade: first-row lanes send half 1, second-row lanes send half 0, to lane ^ 4
      uint4 snd;                                 // (component-wise selects: a select of whole arrays went through scratch)
      snd.x = fr ? half[0].x : half[1].x; snd.y = fr ? half[0].y : half[1].y;
      snd.z = fr ? half[0].z : half[1].z; snd.w = fr ? half[0].w : half[1].w;
      uint4 rcv;
      // lane ^ 4: ds_swizzle in bit mode (and 0x1f, or 0, xor 4) — no address register, no LDS memory
      rcv.x = (uint32_t)__builtin_amdgcn_ds_swizzle((int)snd.x, 0x101F); rcv.y = (uint32_t)__builtin_amdgcn_ds_swizzle((int)snd.y, 0x101F);
      rcv.z = (uint32_t)__builtin_amdgcn_ds_swizzle((int)snd.z, 0x101F); rcv.w = (uint32_t)__builtin_amdgcn_ds_swizzle((int)snd.w, 0x101F);
      uint4 out0, out1;                          // this lane's 16 bytes of the fragment's first / second row pixel
      out0.x = fr ? rcv.x : half[0].x; out0.y = fr ? rcv.y : half[0].y; out0.z = fr ? rcv.z : half[0].z; out0.w = fr ? rcv.w : half[0].w;
      out1.x = fr ? half[1].x : rcv.x; out1.y = fr ? half[1].y : rcv.y; out1.z = fr ? half[1].z : rcv.z; out1.w = fr ? half[1].w : rcv.w;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        uint4 vv = rr == 0 ? out0 : out1;
        if (MODE == 1 && !(p.dbg & 1)) {
          float x[8];
          epi::unpack8(vv, x);
          if (first_of_slab && i == 0 && rr == 0) {
            // shift of the slab row = what the first lane of this lane's channel set stores here (any stored value of
            // the row serves: bn_finalize re-centres on it)
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[e] = __shfl(x[e], (lane & 48) | (fr << 2), 64);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = x[e] - sh[e];
            s0[e] += d;
            s1[e] += d * d;
          }
        }
        if constexpr (MODE == 2) {
          float g[8], yv[8];
          epi::unpack8(vv, g);
          epi::unpack8(ybn[i][rr], yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            g[e] = (yv[e] * c2[e] + c3[e]) > 0.f ? g[e] : 0.f;
            s0[e] += g[e];
            s1[e] += g[e] * (yv[e] - c0[e]) * c1[e];
          }
          vv = epi::pack8(g);
        }
        *reinterpret_cast<uint4*>(yb + orow[i][rr] + mycol) = vv;
      }
    }
    if constexpr (MODE != 0) {
      if (last_of_slab) {
        // ---- the slab row is complete: fold the 8 lanes that share this lane's channels (same l4, same row class:
        // quad swaps and the row mirror stay inside {0-3, 12-15} and inside {4-11}), fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_mirror (lane i <-> 15 - i: {0-3} <-> {15-12}, {4-7} <-> {11-8})
          s0[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0[e]), 0xB1, 0xf, 0xf, true));
          s1[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[e]), 0xB1, 0xf, 0xf, true));
          s0[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0[e]), 0x4E, 0xf, 0xf, true));
          s1[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[e]), 0x4E, 0xf, 0xf, true));
          s0[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0[e]), 0x140, 0xf, 0xf, true));
          s1[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1[e]), 0x140, 0xf, 0xf, true));
        }
        if (l15 == (fr << 2)) {                  // lanes 0 / 4 of every 16-lane row: one writer per channel set
          const int t = patch / G::kPerSlab;
          float* slab_row = MODE == 1 ? p.stats + ((size_t)t * kC + mycol) * 2
                                      : p.bnb_partial + ((size_t)(p.bnb_tile_off + t) * kC + mycol) * 2;
#pragma unroll
          for (int e = 0; e < 8; e += 2)
            *reinterpret_cast<float4*>(slab_row + e * 2) = make_float4(s0[e], s1[e], s0[e + 1], s1[e + 1]);
          if constexpr (MODE == 1) {
            float* sp = p.stats + (size_t)p.stats_tiles * kC * 2 + (size_t)t * kC + mycol;
            *reinterpret_cast<float4*>(sp) = make_float4(sh[0], sh[1], sh[2], sh[3]);
            *reinterpret_cast<float4*>(sp + 4) = make_float4(sh[4], sh[5], sh[6], sh[7]);
          }
        }
      }
    }
    }
    if (!has_next) return;
    patch = npatch;
    v = nv;
  }
}

template <int ROWS, int MODE, bool AFFINE = false, bool RELU = false>
static int launch(const Params& p, hipStream_t st) {
  typedef Geo<ROWS> G;
  constexpr int kLds = kWBytes + G::kWaves * G::kHaloStride;      // 138 240 (ROWS 8) / 151 552 (ROWS 4)
  static bool attr_set = false;
  static int cus = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wave_kernel<MODE, AFFINE, RELU, ROWS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
      cus = 256;
    cus &= ~7;                              // a multiple of 8: a workgroup's iterations stay on its XCD's range
    attr_set = true;
  }
  const int units = (p.nslabs + G::kWaves - 1) / G::kWaves;
  const int grid = units < cus ? units : cus;
  hipLaunchKernelGGL((conv3x3_wave_kernel<MODE, AFFINE, RELU, ROWS>), dim3(grid), dim3(G::kThreads), kLds, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

template <int ROWS>
static int dispatch(const passl_conv_desc* d, Params& p, hipStream_t st) {
  p.PH = d->IH / ROWS; p.PW = d->IW / 8;
  p.npatches = d->N * p.PH * p.PW;
  p.nslabs = (p.npatches + Geo<ROWS>::kPerSlab - 1) / Geo<ROWS>::kPerSlab;
  p.stats_tiles = p.nslabs;                 // = ceil(M / 128): what the caller sized the slab for
  if (d->stats && d->stats_tiles != p.nslabs) return PASSL_EINVAL;
  p.d_pp = make_fastdiv((uint32_t)(p.PH * p.PW));
  p.d_pw = make_fastdiv((uint32_t)p.PW);
  if (d->stats) return launch<ROWS, 1>(p, st);
  if (d->bnb_partial) return launch<ROWS, 2>(p, st);
  const bool affine = d->scale || d->shift;
  if (affine) return d->relu ? launch<ROWS, 0, true, true>(p, st) : launch<ROWS, 0, true, false>(p, st);
  return d->relu ? launch<ROWS, 0, false, true>(p, st) : launch<ROWS, 0, false, false>(p, st);
}

}  // namespace w3

static int g_w3 = -1, g_w3_dbg = 0, g_w3_rows = 4, g_w3_modes = 7;
int passl_conv3x3_wave_option(const char* name, int value) {
  if (!strcmp(name, "conv3x3_wave_dbg")) { g_w3_dbg = value; return PASSL_OK; }
  // which launches take the kernel (in-step experiments): bit 0 = plain / affine / ReLU epilogues (the key encoder's
  // folded BatchNorm, evaluation), bit 1 = forward with fused statistics, bit 2 = data gradient with the BatchNorm-backward sums
  if (!strcmp(name, "conv3x3_wave_modes")) { g_w3_modes = value & 7; return PASSL_OK; }
  if (!strcmp(name, "conv3x3_wave_rows")) {          // 8: four waves x (8 x 8 patches); 4: eight waves x (4 x 8 patches)
    if (value != 4 && value != 8) return PASSL_EINVAL;
    g_w3_rows = value;
    return PASSL_OK;
  }
  if (strcmp(name, "conv3x3_wave")) return PASSL_EINVAL;
  g_w3 = value != 0;
  return PASSL_OK;
}

int passl_conv3x3_wave_try(const passl_conv_desc* d, hipStream_t st) {
  if (g_w3 < 0) {
    const char* e = getenv("PASSL_CONV3X3_WAVE");
    g_w3 = e ? (atoi(e) != 0) : 1;
  }
  if (!g_w3) return PASSL_EUNSUPPORTED;
  if (d->dtype != PASSL_BF16 || d->out_f32 || d->residual) return PASSL_EUNSUPPORTED;
  if (d->R != 3 || d->S != 3 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1) return PASSL_EUNSUPPORTED;
  if (d->C != w3::kC || d->NCOLS != w3::kC || d->IH != d->OP || d->IW != d->OQ) return PASSL_EUNSUPPORTED;
  if ((d->IH & 7) || (d->IW & 7) || d->IH > 248 || d->IW > 248) return PASSL_EUNSUPPORTED;
  if (d->a_sw != d->C || d->a_sh != (int64_t)d->IW * d->C || d->a_sn != (int64_t)d->IH * d->IW * d->C ||
      d->y_sw != d->NCOLS || d->y_sh != (int64_t)d->OQ * d->NCOLS || d->y_sn != (int64_t)d->OP * d->OQ * d->NCOLS)
    return PASSL_EUNSUPPORTED;
  const int64_t a_bytes = (int64_t)d->N * d->IH * d->IW * d->C * 2;
  if (a_bytes >= 0x7ffffff0ll) return PASSL_EUNSUPPORTED;           // 32-bit buffer offsets, also for the output
  if (d->stats && d->bnb_partial) return PASSL_EUNSUPPORTED;
  if (!((g_w3_modes >> (d->stats ? 1 : d->bnb_partial ? 2 : 0)) & 1)) return PASSL_EUNSUPPORTED;
  if (d->bnb_partial && (d->relu || d->scale || d->shift || (d->bnb_relu != 0 && d->bnb_relu != 2) || d->bnb2_partial))
    return PASSL_EUNSUPPORTED;
  if (d->stats && (d->relu || d->scale || d->shift)) return PASSL_EUNSUPPORTED;
  w3::Params p;
  p.a = reinterpret_cast<const char*>(d->a);
  p.b = reinterpret_cast<const char*>(d->b);
  p.y = reinterpret_cast<char*>(d->y);
  p.scale = d->scale; p.shift = d->shift;
  p.stats = d->stats;
  p.bnb_y = reinterpret_cast<const char*>(d->bnb_y);
  p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd; p.bnb_scale = d->bnb_scale; p.bnb_shift = d->bnb_shift;
  p.bnb_partial = d->bnb_partial; p.bnb_relu = d->bnb_relu; p.bnb_tile_off = d->bnb_tile_off;
  p.a_bytes = (uint32_t)a_bytes;
  p.N = d->N; p.IH = d->IH; p.IW = d->IW;
  p.relu = d->relu;
  p.dbg = g_w3_dbg;
  return g_w3_rows == 8 ? w3::dispatch<8>(d, p, st) : w3::dispatch<4>(d, p, st);
}
