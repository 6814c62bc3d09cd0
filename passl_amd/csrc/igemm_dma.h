// Shared pieces of the LDS-DMA implicit-GEMM kernels (conv_igemm_ring.hip: 128-row tiles, 2-stage ring;
// conv_igemm_8p.hip: 256 x 256 tiles, 8-phase schedule): launch parameters, fast division, the
// inline-asm fragment read.  gfx950 only.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.h"

namespace ring {

constexpr int kRowBytes = 128;
constexpr uint32_t kOOB = 0x7ffffff0u;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct FastDiv { uint32_t mul, sh1, sh2; };
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f = {0, 0, 0};
  if (d > 1) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = 1;
    f.sh2 = l - 1;
  }
  return f;
}
__device__ __forceinline__ int fdiv(int n, const FastDiv f) {
  const uint32_t t = __umulhi(f.mul, (uint32_t)n);
  return (int)((t + (((uint32_t)n - t) >> f.sh1)) >> f.sh2);
}

struct Params {
  const char* a;
  const char* b;
  char* y;
  const float* scale;
  const float* shift;
  const char* res;
  float* stats;      // fused BatchNorm statistics slab (igemm_epi.h)
  int stats_tiles;
  const char* bnb_y;
  const uint8_t* bnb_mask;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_scale;
  const float* bnb_shift;
  float* bnb_partial;
  int bnb_relu, bnb_tile_off;
  const char* bnb2_y;         // a second BatchNorm fed by the same gradient (igemm_epi.h: BNB2 instantiation only)
  const float* bnb2_mean;
  const float* bnb2_invstd;
  float* bnb2_partial;
  int64_t a_total;            // bytes of the whole A tensor (may exceed 32 bits: every workgroup rebases its buffer)
  uint32_t b_bytes;
  int M, NCOLS, KDIM;
  int OP, OQ, S, C, IH, IW, sh, sw, ph, pw;
  int a_sn2, a_sh2, a_sw2;      // BYTE strides of A (fit 32 bits, checked on the host)
  int64_t y_sn, y_sh, y_sw;
  int relu, dense;
  int tiles_n, ntiles;
  FastDiv d_opq, d_oq, d_tn;
};

template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// Host side: descriptor -> Params for BM x BN output tiles.  false = outside the LDS-DMA kernels' envelope
// (bf16 operands and output, C a multiple of 64, operands addressable through 32-bit buffer offsets).
static inline bool fill_params(const passl_conv_desc* d, int bm, int bn, Params& p) {
  if (d->dtype != PASSL_BF16 || d->out_f32) return false;
  if ((d->C % 64) != 0) return false;
  const int64_t M64 = (int64_t)d->N * d->OP * d->OQ;
  const int64_t K64 = (int64_t)d->R * d->S * d->C;
  const int64_t a_bytes = (int64_t)d->N * d->a_sn * 2;
  const int64_t b_bytes = (int64_t)d->NCOLS * K64 * 2;
  const int64_t lim = 0x7ffffff0ll;
  if (a_bytes <= 0 || b_bytes >= lim) return false;
  {
    // every workgroup addresses A relative to the first image its (<= 256-row) tile touches: the images a
    // tile can span must fit 31 bits of byte offset (the tensor itself may be far larger)
    const int64_t opq = (int64_t)d->OP * d->OQ;
    const int64_t span_images = 256 / opq + 2;
    if (d->a_sn * 2 * span_images >= lim || d->a_sn * 2 >= lim) return false;
  }
  const int tiles_n = (d->NCOLS + bn - 1) / bn;
  const int64_t tiles_m = (M64 + bm - 1) / bm;
  if (tiles_m * tiles_n > 0x7fffffffll) return false;
  p.a = reinterpret_cast<const char*>(d->a);
  p.b = reinterpret_cast<const char*>(d->b);
  p.y = reinterpret_cast<char*>(d->y);
  p.scale = d->scale; p.shift = d->shift;
  p.res = reinterpret_cast<const char*>(d->residual);
  p.stats = d->stats; p.stats_tiles = (int)((M64 + 127) / 128);
  p.bnb_y = reinterpret_cast<const char*>(d->bnb_y); p.bnb_mask = d->bnb_mask;
  p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd;
  p.bnb_scale = d->bnb_scale; p.bnb_shift = d->bnb_shift;
  p.bnb_partial = d->bnb_partial; p.bnb_relu = d->bnb_relu; p.bnb_tile_off = d->bnb_tile_off;
  p.bnb2_y = reinterpret_cast<const char*>(d->bnb2_y); p.bnb2_mean = d->bnb2_mean; p.bnb2_invstd = d->bnb2_invstd;
  p.bnb2_partial = d->bnb2_partial;
  p.a_total = a_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.M = (int)M64; p.NCOLS = d->NCOLS; p.KDIM = (int)K64;
  p.OP = d->OP; p.OQ = d->OQ; p.S = d->S; p.C = d->C;
  p.IH = d->IH; p.IW = d->IW; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw;
  p.a_sn2 = (int)(d->a_sn * 2); p.a_sh2 = (int)(d->a_sh * 2); p.a_sw2 = (int)(d->a_sw * 2);
  p.y_sn = d->y_sn; p.y_sh = d->y_sh; p.y_sw = d->y_sw;
  p.relu = d->relu;
  p.dense = d->R == 1 && d->S == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 &&
            d->IH == d->OP && d->IW == d->OQ && d->a_sw == d->C &&
            d->a_sh == (int64_t)d->IW * d->C && d->a_sn == (int64_t)d->IH * d->IW * d->C &&
            d->y_sw == d->NCOLS && d->y_sh == (int64_t)d->OQ * d->NCOLS &&
            d->y_sn == (int64_t)d->OP * d->OQ * d->NCOLS;
  p.tiles_n = tiles_n;
  p.ntiles = (int)(tiles_m * tiles_n);
  p.d_opq = make_fastdiv((uint32_t)(d->OP * d->OQ));
  p.d_oq = make_fastdiv((uint32_t)d->OQ);
  p.d_tn = make_fastdiv((uint32_t)tiles_n);
  return true;
}

}  // namespace ring
