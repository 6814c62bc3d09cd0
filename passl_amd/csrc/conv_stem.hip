// ResNet stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels), bf16, gfx950.
//
// Same contract and operand layout as the stem launch of passl_hip_conv_igemm (plan.stem_desc):
//   A = zero-padded NHWC image with 4 channels, [N][Hp][Wp][4] bf16 (Wp even);
//   B = packed filter [64][7][8][4] bf16 (tap 7 and channel 3 carry zero weights): K = 7 x 32;
//   Y[n][oh][ow][64] = epi( sum_{r,s,c} A[n][2 oh + r][2 ow + s][c] * B[k][r][s][c] ).
// The generic implicit-GEMM kernel gathers that K = 224 reduction in 64-byte pieces (a 16-byte chunk
// per lane, 7 filter rows, integer divisions per chunk): 303 us per pass at batch 256, against
// ~85 us for its 520 MB of HBM traffic.  Here the workgroup is a SPATIAL tile instead:
//
//  * one workgroup (4 waves, 2 x 2) = 8 x 16 output pixels x 64 channels; its input patch
//    (21 rows x 38 pixels x 4 channels = 6.4 KB) is copied to LDS once with 16-byte row-contiguous
//    loads, every input pixel is fetched from L2/HBM once per tile (1.56x halo) instead of 12 times;
//  * K is walked one filter row r at a time (7 x v_mfma_f32_16x16x32_bf16 k-steps): the 8 k-values
//    of lane (pixel l15, group l4) are taps s = 2 l4, 2 l4 + 1 x 4 channels = the two adjacent input
//    pixels (2 ow + 2 l4, +1) = ONE aligned ds_read_b128 (pixel pitch 8 B, even pixel index); the 16
//    lanes of a group read 16 consecutive 16-byte slots (conflict-free), neighbouring (ow, l4) pairs
//    that share an address broadcast;
//  * the packed filter (28 KB) is copied to LDS ONCE per workgroup (bank-conflict-free 464-byte row
//    pitch): the launch is persistent (768 workgroups = 3 per CU by LDS, tiles dealt round-robin inside
//    each XCD's range).  One workgroup per tile with the weight fragments fetched from L2 into
//    registers spends its time in those loads (16 cache lines per wave instruction: 198 us per pass),
//    keeping them in registers across a persistent loop spills (the epilogue needs the registers);
//  * accumulators follow igemm_kernel's layout, so the shared epilogue (igemm_epi.h: affine / ReLU /
//    fused BatchNorm statistics slab per 128-pixel tile) is reused unchanged.  A statistics slab row is
//    one 8 x 16 tile here (any partition into 128-element groups serves bn_finalize).
//
// Used for OP % 8 == 0, OQ % 16 == 0 (224^2 inputs: 112 x 112); everything else stays on igemm_kernel.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "igemm_epi.h"

namespace stem {

constexpr int kThreads = 256;
constexpr int TH = 8, TW = 16;                 // output tile (pixels)
constexpr int PR = 2 * TH + 5;                 // 21 patch rows
constexpr int PW = 2 * TW + 6;                 // 38 patch pixels per row (37 + the zero-weight 8th tap)
constexpr int PROW = PW * 8;                   // 304 bytes per patch row = 19 x 16 B
constexpr int PCH = PROW / 16;                 // 19 chunks per row
constexpr int BM = TH * TW, BN = 64;
constexpr int KROW = 32;                       // k-values per filter row (8 taps x 4 channels)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

struct Params {
  const char* a;
  const char* b;
  char* y;
  const float* scale;
  const float* shift;
  const char* res;
  float* stats;
  int stats_tiles;
  const char* bnb_y;
  const uint8_t* bnb_mask;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_scale;
  const float* bnb_shift;
  float* bnb_partial;
  int bnb_relu, bnb_tile_off;
  int M, NCOLS;
  int OP, OQ;
  int64_t a_sn, a_sh;        // elements: image pitch, padded-row pitch
  int relu;
  int tiles_x, tiles_y;      // OQ / 16, OP / 8
  int ntiles;
};

__global__ void __launch_bounds__(kThreads, 3) stem_kernel(const Params p) {
  constexpr int WM = 64, WN = 32, FM = 4, FN = 2;
  constexpr int LDOB = BN + 8;
  constexpr int PATCH_BYTES = PR * PROW;                       // 6384
  constexpr int EPI_BYTES = BM * LDOB * 2;                     // 18432
  constexpr int RED_BYTES = (kThreads / (BN / 8)) * BN * 2 * 4;  // statistics fold: 16 KB
  constexpr int MAIN = EPI_BYTES > RED_BYTES ? EPI_BYTES : RED_BYTES;
  static_assert(PATCH_BYTES <= MAIN, "the epilogue tile aliases the patch");
  // weights [64][7 x 32] bf16 with a 464-byte row pitch: the 16 lanes of a ds_read_b128 group read 16
  // different rows at the same column -> (116 * row) mod 64 dwords hits 16 distinct 4-bank groups
  constexpr int WPITCH = 7 * KROW * 2 + 16;
  constexpr int W_BYTES = BN * WPITCH;                         // 29,696
  __shared__ __attribute__((aligned(16))) char smem[MAIN + BM * 8 + W_BYTES];
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + MAIN);
  char* wlds = smem + MAIN + BM * 8;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // ---- weights -> LDS, once per workgroup (coalesced 16-byte chunks: 28 per row)
  for (int c = tid; c < BN * 28; c += kThreads) {
    const int row = c / 28, ch = c - row * 28;
    *reinterpret_cast<uint4*>(wlds + row * WPITCH + ch * 16) =
        *reinterpret_cast<const uint4*>(p.b + ((int64_t)row * (7 * KROW * 2) + ch * 16));
  }
  // fragment of filter row r, N-fragment j: B[col = wn*32 + j*16 + l15][k = r*32 + l4*8 .. +8]
  const int b0 = (wn * WN + l15) * WPITCH + l4 * 16;

  // persistent: the tiles of one XCD (a contiguous range: neighbouring tiles share halo pixels in its
  // L2) are dealt round-robin to its gridDim.x / 8 workgroups
  int tile, tile_end;
  const int tile_step = (int)(gridDim.x >> 3);
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile = start + local;
    tile_end = start + q + (xcd < r ? 1 : 0);
  }
  // pixel of fragment i, lane l15: tile row wm*64 + i*16 + l15 -> (oh_l = wm*4 + i, ow_l = l15)
  const int a0 = ((2 * (wm * 4)) * PW + 2 * l15 + 2 * l4) * 8;

  for (; tile < tile_end; tile += tile_step) {
    // tile -> (image, band, column block)
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    const int n = t2 / p.tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;

    // ---- input patch -> LDS (row-contiguous 16-byte chunks)
    const char* src = p.a + ((int64_t)n * p.a_sn + (int64_t)(2 * oh0) * p.a_sh + (int64_t)(2 * ow0) * 4) * 2;
    uint4 pv[(PR * PCH + kThreads - 1) / kThreads];
#pragma unroll
    for (int it = 0; it < (PR * PCH + kThreads - 1) / kThreads; ++it) {
      const int c = tid + it * kThreads;
      const int cc = c < PR * PCH ? c : PR * PCH - 1;       // branch-free: both loads in flight together
      const int pr = cc / PCH, pc = cc - pr * PCH;
      pv[it] = *reinterpret_cast<const uint4*>(src + (int64_t)pr * p.a_sh * 2 + pc * 16);
    }
#pragma unroll
    for (int it = 0; it < (PR * PCH + kThreads - 1) / kThreads; ++it) {
      const int c = tid + it * kThreads;
      if (c < PR * PCH) {
        const int pr = c / PCH, pc = c - pr * PCH;
        *reinterpret_cast<uint4*>(smem + pr * PROW + pc * 16) = pv[it];
      }
    }
    // output row offsets (elements): tile row t <-> pixel (oh0 + t / 16, ow0 + t % 16)
    if (tid < BM) {
      const int oh = oh0 + (tid >> 4), ow = ow0 + (tid & 15);
      rowoff[tid] = (((int64_t)n * p.OP + oh) * p.OQ + ow) * p.NCOLS;
    }
    __syncthreads();

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      bf16x8_t af[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + a0 + (2 * i + r) * PROW));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        wf[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(wlds + b0 + j * 16 * WPITCH + r * (KROW * 2)));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // operands swapped like igemm_kernel: acc[i][j][e] = C[pixel wm*64 + i*16 + l15][col wn*32 + j*16 + l4*4 + e]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();      // every wave is done with the patch: the epilogue tile aliases it

    epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN>(p, smem, rowoff, acc, wm, wn, lane, tid, 0, tile);
    __syncthreads();      // the epilogue's LDS reads are done before the next patch is written
  }
}

}  // namespace stem

// PASSL_EUNSUPPORTED when the descriptor is not the bf16 stem launch this kernel covers (the caller
// then uses igemm_kernel); the descriptor has already been validated by passl_hip_conv_igemm.
static int g_stem_enabled = -1;

// passl_hip_set_option("stem_kernel", 0/1)   (runtime.hip dispatches)
int passl_stem_option(const char* name, int value) {
  if (strcmp(name, "stem_kernel")) return PASSL_EINVAL;
  g_stem_enabled = value != 0;
  return PASSL_OK;
}

int passl_stem_try(const passl_conv_desc* d, hipStream_t st) {
  if (g_stem_enabled < 0) {
    const char* e = getenv("PASSL_STEM_KERNEL");
    g_stem_enabled = e ? atoi(e) : 1;
  }
  const int enabled = g_stem_enabled;
  if (!enabled || d->dtype != PASSL_BF16 || d->out_f32 || d->bnb_partial || d->residual) return PASSL_EUNSUPPORTED;
  if (d->R != 7 || d->S != 1 || d->C != 32 || d->sh != 2 || d->sw != 1 || d->ph != 0 || d->pw != 0 ||
      d->NCOLS != 64 || d->a_sw != 8)
    return PASSL_EUNSUPPORTED;
  if ((d->OP % stem::TH) || (d->OQ % stem::TW) || (d->a_sh & 7) || (d->a_sn & 7)) return PASSL_EUNSUPPORTED;
  // the patch of the last tile must lie inside the padded image: rows 2 (OP-1) + 6, pixels 2 (OQ-1) + 7
  if (2 * (d->OP - 1) + 7 > d->IH || (int64_t)(2 * (d->OQ - 1) + 8) * 4 > d->a_sh) return PASSL_EUNSUPPORTED;
  // dense NHWC output
  if (d->y_sw != d->NCOLS || d->y_sh != (int64_t)d->OQ * d->NCOLS || d->y_sn != (int64_t)d->OP * d->OQ * d->NCOLS)
    return PASSL_EUNSUPPORTED;
  const int64_t M64 = (int64_t)d->N * d->OP * d->OQ;
  stem::Params p;
  p.a = reinterpret_cast<const char*>(d->a);
  p.b = reinterpret_cast<const char*>(d->b);
  p.y = reinterpret_cast<char*>(d->y);
  p.scale = d->scale; p.shift = d->shift;
  p.res = nullptr;
  p.stats = d->stats;
  p.stats_tiles = (int)(M64 / stem::BM);
  p.bnb_y = nullptr; p.bnb_mask = nullptr; p.bnb_mean = nullptr; p.bnb_invstd = nullptr;
  p.bnb_scale = nullptr; p.bnb_shift = nullptr; p.bnb_partial = nullptr; p.bnb_relu = 0; p.bnb_tile_off = 0;
  p.M = (int)M64; p.NCOLS = d->NCOLS;
  p.OP = d->OP; p.OQ = d->OQ;
  p.a_sn = d->a_sn; p.a_sh = d->a_sh;
  p.relu = d->relu;
  p.tiles_x = d->OQ / stem::TW; p.tiles_y = d->OP / stem::TH;
  if (d->stats && d->stats_tiles != p.stats_tiles) return PASSL_EINVAL;
  const int64_t ntiles = (int64_t)d->N * p.tiles_x * p.tiles_y;
  if (ntiles > 0x7fffffff) return PASSL_EUNSUPPORTED;
  p.ntiles = (int)ntiles;
  // 3 workgroups per CU (49 KB of LDS each), fewer when the launch is small; a multiple of the 8 XCDs
  int64_t grid = ntiles < 768 ? ((ntiles + 7) / 8) * 8 : 768;
  hipLaunchKernelGGL(stem::stem_kernel, dim3((unsigned)grid), dim3(stem::kThreads), 0, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}
