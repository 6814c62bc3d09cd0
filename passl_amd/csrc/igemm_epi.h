// bf16 output epilogue shared by igemm_kernel (conv_igemm.hip) and igemm_ring_kernel
// (conv_igemm_ring.hip): accumulators -> LDS (packed bf16) -> coalesced 16-byte row stores with the
// fused per-channel affine, residual add, ReLU and — fixed-order, atomic-free — either
//
//  (a) forward BatchNorm statistics of the STORED values: per 128-row output tile t and column c
//        stats[t][c][0..1] = sum (v - s), sum (v - s)^2,   s = shifts[t][c] = the tile's first-row value
//      (shifted sums: no E[x^2] - mean^2 cancellation; bn_finalize re-centres them in fp64), or
//  (b) the BatchNorm BACKWARD statistics of the layer whose output gradient this launch produces
//      (a data-gradient conv): with y = that BatchNorm's input tile,  g = value * relu_mask(y),
//        bnb_partial[tile_off + t][c][0..1] = sum g, sum g * (y - mean[c]) * invstd[c]
//      and g (not the unmasked gradient) is what gets stored — so bn_bwd_reduce never runs and
//      bn_bwd_apply needs neither the mask nor a separate residual-gradient output.
//
// Every slab row is written by exactly one workgroup (plain stores), the cross-tile reduction
// happens in bn_finalize / bn_bwd_finalize in a fixed order: bit-reproducible run to run.
#pragma once
#include "common.h"

namespace epi {

// tuning switches of the register-staged kernel (PASSL_IGEMM_DBG, conv_igemm.hip: Params::dbg); 0 for the other kernels
template <typename P> __device__ __forceinline__ auto dbg_of(const P& p, int) -> decltype(p.dbg) { return p.dbg; }
template <typename P> __device__ __forceinline__ int dbg_of(const P&, long) { return 0; }

__device__ __forceinline__ void unpack8(const uint4 v, float (&a)[8]) {
  a[0] = __uint_as_float(v.x << 16); a[1] = __uint_as_float(v.x & 0xffff0000u);
  a[2] = __uint_as_float(v.y << 16); a[3] = __uint_as_float(v.y & 0xffff0000u);
  a[4] = __uint_as_float(v.z << 16); a[5] = __uint_as_float(v.z & 0xffff0000u);
  a[6] = __uint_as_float(v.w << 16); a[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&a)[8]) {
  return make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(a[4], a[5]), pack2bf(a[6], a[7]));
}

// P needs: y, scale, shift, res, relu, NCOLS, stats, bnb_y, bnb_mask, bnb_mean, bnb_invstd, bnb_scale,
// bnb_shift, bnb_partial, bnb_relu, bnb_tile_off, M (rows) and stats_tiles (= ceil(M / 128)).
// smem: the kernel's dynamic LDS (main-loop tiles are dead); rowoff[BM]: element offset of each
// output row (-1 = out of range).  mt = index of this workgroup's 128-row tile.
// Column of accumulator fragment j: (j / FNH) * CH + wn * WN + (j % FNH) * 16 (+ 4 * (lane >> 4)); the
// defaults (FNH = FN, CH = 0) are one contiguous WN-wide strip per wave, the 8-phase kernel's waves own one
// 32-column strip in each 128-column half of the tile (FNH = 2, CH = 128).
// BNB2 (compile-time: its own kernel instantiations only): a second BatchNorm fed by the same gradient
// (passl_conv_desc.bnb2_*): one more row load and one more accumulator in the row loop, a second fold through LDS.
template <int BM, int BN, int NTHREADS, int FM, int FN, int WM, int WN, bool LEAN = false, int FNH = FN, int CH = 0,
          bool BNB2 = false, typename P>
__device__ __forceinline__ void epilogue_bf16(const P& p, char* smem, const int64_t* rowoff,
                                              const f32x4 (&acc)[FM][FN], int wm, int wn, int lane,
                                              int tid, int n0, int mt) {
  static_assert(BM == 128, "slab rows are per 128-row tile");
  constexpr int LDOB = BN + 8;             // bf16 epilogue pitch (elements): 16-byte aligned rows
  constexpr int CPR = BN / 8;              // 8-column chunks per row
  constexpr int NT = BM * CPR / NTHREADS;  // chunks per thread
  static_assert(NTHREADS % CPR == 0, "a thread must keep its column chunk across iterations");
  const int l15 = lane & 15, l4 = lane >> 4;
  const int frow = l15;                                // this lane's row inside a 16-row fragment

  // ---- phase 1: affine (+ReLU when nothing else follows) on the fp32 accumulators; each lane packs
  // its 4 consecutive channels and writes 8 bytes (ds_write_b64) into out[BM][LDOB]
  char* outc = smem;
  const bool bnb = !LEAN && p.bnb_partial != nullptr;
  const bool has_res = !LEAN && p.res != nullptr;
  const bool relu_now = p.relu && !has_res;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = (j / FNH) * CH + wn * WN + (j % FNH) * 16 + l4 * 4;
    const int gcol = n0 + col;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gcol < p.NCOLS) {
      if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + gcol);
      if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + gcol);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      f32x4 a = acc[i][j];
      a[0] = a[0] * sc.x + sh.x; a[1] = a[1] * sc.y + sh.y;
      a[2] = a[2] * sc.z + sh.z; a[3] = a[3] * sc.w + sh.w;
      if (relu_now) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
      }
      const int row = wm * WM + i * 16 + frow;
      *reinterpret_cast<uint2*>(outc + row * (LDOB * 2) + col * 2) =
          make_uint2(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]));
    }
  }
  // this thread's phase-2 chunks: residual rows / BatchNorm-input rows / mask bytes are requested
  // before the barrier (the accumulators are dead by now) and consumed after it
  const int cc = tid % CPR;
  const int gcol = n0 + cc * 8;
  const bool col_ok = gcol < p.NCOLS;
  uint4 rres[NT], ry[NT];
  uint32_t rmask[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    rres[t] = make_uint4(0, 0, 0, 0);
    ry[t] = make_uint4(0, 0, 0, 0);
    rmask[t] = 0xffu;
  }
  if (has_res || bnb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = (tid + t * NTHREADS) / CPR;
      const int64_t roff = rowoff[row];
      if (roff < 0 || !col_ok) continue;
      if (has_res)
        rres[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.res) + roff + gcol);
      if (bnb) {
        ry[t] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bnb_y) + roff + gcol);
        if (p.bnb_relu == 3) rmask[t] = p.bnb_mask[(roff + gcol) >> 3];
      }
    }
  }
  __syncthreads();
  const bf16_t* outb = reinterpret_cast<const bf16_t*>(smem);

  // per-column constants of the statistics
  float c0[8], c1[8], c2[8], c3[8];
  const bool fstats = p.stats != nullptr;
#pragma unroll
  for (int e = 0; e < 8; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
  if (fstats) {
    // shift = the tile's first row (always in range: m0 < M), as stored
    unpack8(*reinterpret_cast<const uint4*>(outb + cc * 8), c0);
  } else if (bnb && col_ok) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { c0[e] = p.bnb_mean[gcol + e]; c1[e] = p.bnb_invstd[gcol + e]; }
    if (p.bnb_relu == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { c2[e] = p.bnb_scale[gcol + e]; c3[e] = p.bnb_shift[gcol + e]; }
    }
  }

  float s0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // BNB2: sum g * xhat of the second BatchNorm
  float c4[8], c5[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { c4[e] = 0.f; c5[e] = 0.f; }
  if constexpr (BNB2) {
    if (bnb && col_ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { c4[e] = p.bnb2_mean[gcol + e]; c5[e] = p.bnb2_invstd[gcol + e]; }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = (tid + t * NTHREADS) / CPR;
    const int64_t roff = rowoff[row];
    if (roff < 0 || !col_ok) continue;
    const int64_t o = roff + gcol;
    uint4 v = *reinterpret_cast<const uint4*>(outb + row * LDOB + cc * 8);
    if (has_res) {
      float a[8], rr[8];
      unpack8(v, a);
      unpack8(rres[t], rr);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] += rr[e];
        if (p.relu) a[e] = fmaxf(a[e], 0.f);
      }
      v = pack8(a);
    }
    if (bnb) {
      float g[8], yv[8];
      unpack8(v, g);          // the bf16-rounded gradient: what a separate pass would have read
      unpack8(ry[t], yv);
      if (p.bnb_relu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (yv[e] * c2[e] + c3[e]) > 0.f ? g[e] : 0.f;
      } else if (p.bnb_relu == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = ((rmask[t] >> e) & 1u) ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s0[e] += g[e];
        s1[e] += g[e] * (yv[e] - c0[e]) * c1[e];
      }
      if constexpr (BNB2) {
        // (loaded here, not with the other rows in front of the barrier: 32 more live registers there push the
        // instantiation from 3 to 2 workgroups per CU)
        float y2[8];
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bnb2_y) + o), y2);
#pragma unroll
        for (int e = 0; e < 8; ++e) s2[e] += g[e] * (y2[e] - c4[e]) * c5[e];
      }
      v = pack8(g);           // exact: g holds bf16 values or zeros
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + o) = v;
    if (fstats && !(dbg_of(p, 0) & 32)) {
      float x[8];
      unpack8(v, x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = x[e] - c0[e];
        s0[e] += d;
        s1[e] += d * d;
      }
    }
  }
  if (!fstats && !bnb) return;
  if (dbg_of(p, 0) & 16) return;          // (ablation: no cross-thread reduction, no slab)

  // threads tid, tid+CPR, ... share a column chunk: each writes its 16 partial values to
  // red[tid / CPR][BN][2] (the output tile is dead by now), 2*BN threads add the NTHREADS/CPR
  // partials in a fixed order and store ONE slab entry per (column, statistic) of the tile.
  constexpr int J = NTHREADS / CPR;
  __syncthreads();                       // every thread is done reading the output tile
  float* red = reinterpret_cast<float*>(smem);
  {
    float* dst = red + (tid / CPR) * (BN * 2) + cc * 16;
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      *reinterpret_cast<float4*>(dst + e * 2) = make_float4(s0[e], s1[e], s0[e + 1], s1[e + 1]);
  }
  if (fstats && tid < CPR && col_ok) {
    // shifts[t][c] live behind the sums: stats + stats_tiles * NCOLS * 2
    float* sp = p.stats + (int64_t)p.stats_tiles * p.NCOLS * 2 + (int64_t)mt * p.NCOLS + gcol;
    *reinterpret_cast<float4*>(sp) = make_float4(c0[0], c0[1], c0[2], c0[3]);
    *reinterpret_cast<float4*>(sp + 4) = make_float4(c0[4], c0[5], c0[6], c0[7]);
  }
  __syncthreads();
  const int64_t trow = fstats ? (int64_t)mt : (int64_t)p.bnb_tile_off + mt;
  if (tid < 2 * BN && n0 + (tid >> 1) < p.NCOLS) {
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) a += red[j * (BN * 2) + tid];
    float* slab = fstats ? p.stats : p.bnb_partial;
    slab[(trow * p.NCOLS + n0 + (tid >> 1)) * 2 + (tid & 1)] = a;
  }
  if constexpr (BNB2) {
    if (!bnb) return;
    // the second layer's slab row: (sum g, sum g * xhat2) folded the same way
    __syncthreads();
    {
      float* dst = red + (tid / CPR) * (BN * 2) + cc * 16;
#pragma unroll
      for (int e = 0; e < 8; e += 2)
        *reinterpret_cast<float4*>(dst + e * 2) = make_float4(s0[e], s2[e], s0[e + 1], s2[e + 1]);
    }
    __syncthreads();
    if (tid < 2 * BN && n0 + (tid >> 1) < p.NCOLS) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) a += red[j * (BN * 2) + tid];
      p.bnb2_partial[(trow * p.NCOLS + n0 + (tid >> 1)) * 2 + (tid & 1)] = a;
    }
  }
}

}  // namespace epi
