// Index arithmetic of the spatially tiled 3x3 weight-gradient kernel (conv_wgrad_halo.inc, opt-in), free of device
// intrinsics: compiled into the kernel and into the host emulator (tests/emu/wgrad_halo_emu.cpp).
//
//   dW[oc][(r, s), c] = sum over output pixels m of dY[m][oc] * X[m @ (r, s)][c]        (3x3, stride 1, pad 1)
//
// The reduction runs over pixels; a k-tile is one 8 x 8 PATCH of an image (patches in the global order n, py, px;
// images whose sides are not multiples of 8 are covered by ceil(IH / 8) x ceil(IW / 8) patches, the pixels of a patch
// that lie outside the image are fetched as zeros — 31 % more k-tiles at 28 x 28, 14 x 14 and 7 x 7), i.e. 64 reduction
// rows = two 16x16x32 MFMA k-steps of four patch rows each.  Per k-tile
// the workgroup stages
//   * dY of the patch:  [64 pixels][64 oc], 128-byte rows, the transposing-read swizzle of wgrad_pipe_kernel, and
//   * X of the patch WITH ITS HALO, once, for all nine taps: [10 rows][12 columns][64 c + pad] — pixel (y, x) of the
//     patch and tap (r, s) sit at halo row (y + r) * 12 + (x + s); zero padding outside the image is fetched through
//     the buffer bounds check.  Row pitch 160 B (five 32-byte units) and 12 columns per patch row make the eight row
//     segments a 32-lane group of ds_read_b64_tr_b16 touches (rows L..L+3 and L+12..L+15, 32 bytes each) start at
//     eight different multiples of 8 banks: conflict-free for every tap and fragment.
// A workgroup owns a 64 (oc) x 64 (c) block of dW for ALL NINE TAPS (36 accumulator fragments per wave): per k-tile
// 8 + 20 LDS-DMA pieces feed 72 MFMAs per wave, against 24 pieces for 16 MFMAs in the per-tap kernel.
#pragma once
#include <stdint.h>
#include "halo_geom.h"      // FDiv, fdiv, kNoSrc, HALO_HD

namespace wgh {

using halo::FDiv;
using halo::fdiv;
using halo::kNoSrc;

constexpr int kPW = 12;                 // LDS columns per patch row (10 used)
constexpr int kRows = 10 * kPW;         // 120 halo rows
constexpr int kSlots = 10;              // 16-byte slots per halo row: 8 channel chunks + 2 pad
constexpr int kPitch = kSlots * 16;     // 160 bytes
constexpr int kHaloPieces = 20;         // 1 KB LDS-DMA pieces per k-tile: 1200 slots -> 18.75, rounded to 5 per wave
constexpr int kDyPieces = 8;            // 64 pixels x 128 bytes
constexpr int kDyBytes = kDyPieces * 1024, kHaloBytes = kHaloPieces * 1024, kStage = kDyBytes + kHaloBytes;

struct Geom {
  int N, IH, IW, C, NCOLS;
  int PXN, PN, npatches;               // patches per image row, per image, in all
  int a_sn2, a_sh2, a_sw2;             // BYTE strides of X
  int dy_pitch;                        // bytes per dY row
  FDiv d_pn, d_pxn;
};

// image and origin of global patch gp
HALO_HD void patch_origin(const Geom& g, int gp, int& n, int& y0, int& x0) {
  n = fdiv(gp, g.d_pn);
  const int rem = gp - n * g.PN;
  const int py = fdiv(rem, g.d_pxn);
  y0 = py * 8;
  x0 = (rem - py * g.PXN) * 8;
}

// transposing-read swizzle of a [rows][64 channels] tile with 128-byte rows (wgrad_pipe_kernel: hswz<8>)
HALO_HD int hswz8(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// dY piece q (0..7) of patch gp: rows 8q .. 8q+7 of the k-tile (row kk = 8 * y + x), lane -> row 8q + lane / 8, LDS
// slot lane % 8, SOURCE chunk (lane % 8) ^ (hswz8(row) << 1).  Byte offset into dY, kNoSrc past the last patch /
// column.  oc0 = first output channel of the workgroup's block.
HALO_HD uint32_t dy_src(const Geom& g, int gp, int oc0, int q, int lane) {
  if (gp >= g.npatches) return kNoSrc;
  const int kk = q * 8 + lane / 8;
  const int chunk = (lane % 8) ^ (hswz8(kk) << 1);
  const int oc = oc0 + chunk * 8;
  if (oc >= g.NCOLS) return kNoSrc;
  int n, y0, x0;
  patch_origin(g, gp, n, y0, x0);
  if (y0 + (kk >> 3) >= g.IH || x0 + (kk & 7) >= g.IW) return kNoSrc;      // patch pixel outside the image
  const int m = n * g.IH * g.IW + (y0 + (kk >> 3)) * g.IW + x0 + (kk & 7);
  return (uint32_t)m * (uint32_t)g.dy_pitch + (uint32_t)oc * 2u;
}

// halo piece q (0..19) of patch gp, channel block c0: byte offset into X or kNoSrc
HALO_HD uint32_t halo_src(const Geom& g, int gp, int c0, int q, int lane) {
  if (gp >= g.npatches) return kNoSrc;
  const int p = q * 64 + lane;
  const int hrow = p / kSlots;
  const int cpos = p - hrow * kSlots;
  if (cpos >= 8 || hrow >= kRows) return kNoSrc;
  const int hy = hrow / kPW, hx = hrow - hy * kPW;
  if (hx >= 10) return kNoSrc;
  const int c = c0 + cpos * 8;
  if (c >= g.C) return kNoSrc;
  int n, y0, x0;
  patch_origin(g, gp, n, y0, x0);
  const int ih = y0 + hy - 1, iw = x0 + hx - 1;
  if (ih < 0 || ih >= g.IH || iw < 0 || iw >= g.IW) return kNoSrc;
  return (uint32_t)n * (uint32_t)g.a_sn2 + (uint32_t)ih * (uint32_t)g.a_sh2 + (uint32_t)iw * (uint32_t)g.a_sw2 +
         (uint32_t)c * 2u;
}

// ---- the kernel's split of those offsets into a lane-static part and a per-patch scalar part
// dY: offset = patch_m(n, y0, x0) * dy_pitch + dy_static(q, lane)  if  dy_inside(y0, x0, yx),  yx = the lane's pixel
HALO_HD uint32_t dy_static(const Geom& g, int oc0, int q, int lane, int& yx) {
  const int kk = q * 8 + lane / 8;
  const int chunk = (lane % 8) ^ (hswz8(kk) << 1);
  const int oc = oc0 + chunk * 8;
  yx = (kk >> 3) | ((kk & 7) << 8);
  if (oc >= g.NCOLS) return kNoSrc;
  return (uint32_t)((kk >> 3) * g.IW + (kk & 7)) * (uint32_t)g.dy_pitch + (uint32_t)oc * 2u;
}
HALO_HD bool dy_inside(const Geom& g, int y0, int x0, int yx) { return y0 + (yx & 255) < g.IH && x0 + (yx >> 8) < g.IW; }
HALO_HD int patch_m(const Geom& g, int n, int y0, int x0) { return n * g.IH * g.IW + y0 * g.IW + x0; }
// X: offset = patch_x(n, y0, x0) + halo_static(q, lane)  if  halo_inside(y0, x0, hyx);  hyx = halo (row, column) of
// the lane, or -1 for lanes that never fetch (pad slot, unused row / column, channel >= C)
HALO_HD uint32_t halo_static(const Geom& g, int c0, int q, int lane, int& hyx) {
  const int p = q * 64 + lane;
  const int hrow = p / kSlots;
  const int cpos = p - hrow * kSlots;
  const int hy = hrow / kPW, hx = hrow - hy * kPW;
  const int c = c0 + cpos * 8;
  if (cpos >= 8 || hrow >= kRows || hx >= 10 || c >= g.C) { hyx = -1; return 0; }
  hyx = hy | (hx << 8);
  return (uint32_t)((hy - 1) * g.a_sh2) + (uint32_t)((hx - 1) * g.a_sw2) + (uint32_t)c * 2u;     // wraps for hy, hx = 0
}
HALO_HD bool halo_inside(const Geom& g, int y0, int x0, int hyx) {
  return hyx >= 0 && (uint32_t)(y0 + (hyx & 255) - 1) < (uint32_t)g.IH && (uint32_t)(x0 + (hyx >> 8) - 1) < (uint32_t)g.IW;
}
HALO_HD uint32_t patch_x(const Geom& g, int n, int y0, int x0) {
  return (uint32_t)n * (uint32_t)g.a_sn2 + (uint32_t)y0 * (uint32_t)g.a_sh2 + (uint32_t)x0 * (uint32_t)g.a_sw2;
}

// The form for images whose sides are multiples of 8 (no overhang) — the form that was run on hardware in round 4
// (exact; 64->64 @56: 77 us): validity of a halo fetch is four border flags against the patch's position.
HALO_HD uint32_t dy_static_tiled(const Geom& g, int oc0, int q, int lane) {
  const int kk = q * 8 + lane / 8;
  const int chunk = (lane % 8) ^ (hswz8(kk) << 1);
  const int oc = oc0 + chunk * 8;
  if (oc >= g.NCOLS) return kNoSrc;
  return (uint32_t)((kk >> 3) * g.IW + (kk & 7)) * (uint32_t)g.dy_pitch + (uint32_t)oc * 2u;
}
// flags: 1 top halo row, 2 bottom, 4 left column, 8 right, 16 never valid (pad slot / unused row / channel >= C)
HALO_HD uint32_t halo_static_tiled(const Geom& g, int c0, int q, int lane, int& flags) {
  const int p = q * 64 + lane;
  const int hrow = p / kSlots;
  const int cpos = p - hrow * kSlots;
  const int hy = hrow / kPW, hx = hrow - hy * kPW;
  const int c = c0 + cpos * 8;
  flags = 0;
  if (cpos >= 8 || hrow >= kRows || hx >= 10 || c >= g.C) { flags = 16; return 0; }
  if (hy == 0) flags |= 1;
  if (hy == 9) flags |= 2;
  if (hx == 0) flags |= 4;
  if (hx == 9) flags |= 8;
  return (uint32_t)((hy - 1) * g.a_sh2) + (uint32_t)((hx - 1) * g.a_sw2) + (uint32_t)c * 2u;     // wraps for hy, hx = 0
}
HALO_HD int edge_mask(const Geom& g, int y0, int x0) {
  return 16 | (y0 == 0 ? 1 : 0) | (y0 == g.IH - 8 ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 == g.IW - 8 ? 8 : 0);
}

// ds_read_b64_tr_b16 addresses (bytes inside a stage).  Lane = (g = lane >> 4, r4 = (lane >> 2) & 3, c4 = lane & 3);
// a read covers reduction rows 8 g + 4 half + r4 of a 32-row k-step and hands lane l15 column l15 of a 16-column block.
//   dY fragment (16 oc starting at oc_local): k-step ks, half h
HALO_HD uint32_t dy_frag(int oc_local, int ks, int h, int lane) {
  const int g = lane >> 4, r4 = (lane >> 2) & 3, c4 = lane & 3;
  const int row = 32 * ks + 8 * g + 4 * h + r4;
  const int pair = oc_local >> 4;
  return (uint32_t)(row * 128 + ((pair ^ hswz8(row)) << 5) + c4 * 8);
}
//   X fragment (16 channels starting at c_local), tap (r, s): patch row y = 4 ks + g, x = 4 h + r4
HALO_HD uint32_t x_frag_lane(int lane) {                     // the lane-dependent part
  const int g = lane >> 4, r4 = (lane >> 2) & 3, c4 = lane & 3;
  return (uint32_t)((g * kPW + r4) * kPitch + c4 * 8);
}
HALO_HD constexpr uint32_t x_frag_const(int c_local, int ks, int h, int r, int s) {   // the compile-time part
  return (uint32_t)(kDyBytes + ((4 * ks + r) * kPW + 4 * h + s) * kPitch + c_local * 2);
}

}  // namespace wgh
