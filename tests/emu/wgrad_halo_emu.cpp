// Host emulator of the spatially tiled 3x3 weight-gradient kernel (passl_amd/csrc/conv_wgrad_halo.inc), compiled and
// run by tests/test_halo_geometry.py.  Executes the kernel's address arithmetic lane by lane with the functions of
// passl_amd/csrc/wgrad_halo_geom.h — LDS-DMA pieces (the lane-static + per-patch split the kernel uses, cross-checked
// against the direct formulas), ds_read_b64_tr_b16 addresses WITH the instruction's 16-lane transposition, the MFMA
// operand layout, the accumulator -> dW map — and compares with a direct weight gradient on integer data.  Also the
// bank property of the halo layout for the transposing reads.  Time (waits, barriers, ring) is not modelled.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../passl_amd/csrc/wgrad_halo_geom.h"

using namespace wgh;

static uint32_t mix(uint64_t i, uint32_t seed) {
  uint64_t z = i * 0x9E3779B97F4A7C15ull + ((uint64_t)seed << 32 | 0x7F4A7C15u);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)(z >> 33);
}
static int16_t ival(uint64_t i, uint32_t seed) { return (int16_t)((int)(mix(i, seed) % 31u) - 15); }

// ds_read_b64_tr_b16: every lane gives the address of 4 consecutive 16-bit elements; inside each group of 16 lanes,
// lane i receives element (i & 3) of the source lanes e * 4 + (i >> 2), e = 0..3
static void tr_read(const std::vector<char>& lds, const uint32_t (&addr)[64], int16_t (&out)[64][4]) {
  int16_t src[64][4];
  for (int l = 0; l < 64; ++l) memcpy(src[l], &lds[addr[l]], 8);
  for (int l = 0; l < 64; ++l) {
    const int grp = l & ~15, i = l & 15;
    for (int e = 0; e < 4; ++e) out[l][e] = src[grp + e * 4 + (i >> 2)][i & 3];
  }
}

struct Case { int N, H, W, C, K, splits; };

static long run(const Case& cs) {
  const int N = cs.N, IH = cs.H, IW = cs.W, C = cs.C, K = cs.K;
  const long M = (long)N * IH * IW, KDIM = 9L * C;
  std::vector<int16_t> X((size_t)M * C), DY((size_t)M * K);
  for (size_t i = 0; i < X.size(); ++i) X[i] = ival(i, 11u);
  for (size_t i = 0; i < DY.size(); ++i) DY[i] = ival(i, 29u);
  const char* Xb = reinterpret_cast<const char*>(X.data());
  const char* Db = reinterpret_cast<const char*>(DY.data());
  const uint32_t x_bytes = (uint32_t)(X.size() * 2), dy_bytes = (uint32_t)(DY.size() * 2);
  Geom g;
  g.N = N; g.IH = IH; g.IW = IW; g.C = C; g.NCOLS = K;
  g.PXN = (IW + 7) / 8; g.PN = ((IH + 7) / 8) * g.PXN; g.npatches = N * g.PN;
  g.a_sw2 = C * 2; g.a_sh2 = IW * C * 2; g.a_sn2 = IH * IW * C * 2; g.dy_pitch = K * 2;
  g.d_pn = halo::make_fdiv((uint32_t)g.PN); g.d_pxn = halo::make_fdiv((uint32_t)g.PXN);
  int splits = cs.splits;
  const int nk_total = g.npatches;
  if (splits > nk_total) splits = nk_total;
  const int tps = (nk_total + splits - 1) / splits;
  splits = (nk_total + tps - 1) / tps;
  const int grid_j = (C + 63) / 64, grid_oc = (K + 63) / 64;
  std::vector<int64_t> dw((size_t)K * KDIM, 0);
  std::vector<char> stage(kStage);
  long split_mismatch = 0;
  for (int bz = 0; bz < splits; ++bz)
    for (int by = 0; by < grid_oc; ++by)
      for (int bx = 0; bx < grid_j; ++bx) {
        const int oc0 = by * 64, c0 = bx * 64;
        const int kt_begin = bz * tps, kt_end = kt_begin + tps < nk_total ? kt_begin + tps : nk_total;
        // acc[wave][lane][tap][i][j][e]
        std::vector<int64_t> acc((size_t)4 * 64 * 9 * 2 * 2 * 4, 0);
        int pn, py0, px0;
        patch_origin(g, kt_begin, pn, py0, px0);
        for (int gp = kt_begin; gp < kt_end; ++gp) {
          // ---- DMA pieces: the kernel's static + per-patch form, checked against the direct formulas
          const uint32_t dy_base = (uint32_t)patch_m(g, pn, py0, px0) * (uint32_t)g.dy_pitch;
          const uint32_t x_base = patch_x(g, pn, py0, px0);
          const bool tiled = (IH % 8) == 0 && (IW % 8) == 0;       // the kernel's OVERHANG = false instantiation
          const int emask = edge_mask(g, py0, px0);
          for (int q = 0; q < kDyPieces; ++q)
            for (int lane = 0; lane < 64; ++lane) {
              int yx = 0;
              const uint32_t st = tiled ? dy_static_tiled(g, oc0, q, lane) : dy_static(g, oc0, q, lane, yx);
              const uint32_t off = (st == kNoSrc || (!tiled && !dy_inside(g, py0, px0, yx))) ? kNoSrc : st + dy_base;
              if (off != dy_src(g, gp, oc0, q, lane)) ++split_mismatch;
              char* dst = &stage[(size_t)q * 1024 + lane * 16];
              if (off == kNoSrc || off + 16 > dy_bytes) memset(dst, 0, 16); else memcpy(dst, Db + off, 16);
            }
          for (int q = 0; q < kHaloPieces; ++q)
            for (int lane = 0; lane < 64; ++lane) {
              int hyx;
              const uint32_t st = tiled ? halo_static_tiled(g, c0, q, lane, hyx) : halo_static(g, c0, q, lane, hyx);
              const uint32_t off = tiled ? ((hyx & emask) ? kNoSrc : st + x_base) : (halo_inside(g, py0, px0, hyx) ? st + x_base : kNoSrc);
              if (off != halo_src(g, gp, c0, q, lane)) ++split_mismatch;
              char* dst = &stage[(size_t)kDyBytes + (size_t)q * 1024 + lane * 16];
              if (off == kNoSrc || off + 16 > x_bytes) memset(dst, 0, 16); else memcpy(dst, Xb + off, 16);
            }
          // ---- fragments and MFMAs
          for (int wave = 0; wave < 4; ++wave) {
            const int wm = wave >> 1, wn = wave & 1;
            for (int ks = 0; ks < 2; ++ks) {
              int16_t af[2][64][8];
              for (int i = 0; i < 2; ++i)
                for (int h = 0; h < 2; ++h) {
                  uint32_t ad[64];
                  int16_t o[64][4];
                  for (int lane = 0; lane < 64; ++lane) ad[lane] = dy_frag(wm * 32 + i * 16, 0, h, lane) + (uint32_t)(32 * 128 * ks);
                  tr_read(stage, ad, o);
                  for (int lane = 0; lane < 64; ++lane) memcpy(&af[i][lane][4 * h], o[lane], 8);
                }
              for (int tap = 0; tap < 9; ++tap)
                for (int j = 0; j < 2; ++j) {
                  int16_t bf[64][8];
                  for (int h = 0; h < 2; ++h) {
                    uint32_t ad[64];
                    int16_t o[64][4];
                    for (int lane = 0; lane < 64; ++lane)
                      ad[lane] = x_frag_lane(lane) + (uint32_t)(wn * 64) + x_frag_const(j * 16, ks, h, tap / 3, tap % 3);
                    tr_read(stage, ad, o);
                    for (int lane = 0; lane < 64; ++lane) memcpy(&bf[lane][4 * h], o[lane], 8);
                  }
                  for (int i = 0; i < 2; ++i)
                    for (int lane = 0; lane < 64; ++lane) {
                      const int c15 = lane & 15, r4 = lane >> 4;
                      for (int e = 0; e < 4; ++e) {
                        const int drow = 4 * r4 + e;          // A row (oc), B column c15 (channel)
                        int64_t sum = 0;
                        for (int kg = 0; kg < 4; ++kg)
                          for (int x = 0; x < 8; ++x) sum += (int64_t)af[i][kg * 16 + drow][x] * (int64_t)bf[kg * 16 + c15][x];
                        acc[(((((size_t)wave * 64 + lane) * 9 + tap) * 2 + i) * 2 + j) * 4 + e] += sum;
                      }
                    }
                }
            }
          }
          px0 += 8;
          if (px0 >= IW) { px0 = 0; py0 += 8; if (py0 >= IH) { py0 = 0; ++pn; } }
        }
        for (int wave = 0; wave < 4; ++wave)
          for (int lane = 0; lane < 64; ++lane)
            for (int tap = 0; tap < 9; ++tap)
              for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                  for (int e = 0; e < 4; ++e) {
                    const int oc = oc0 + (wave >> 1) * 32 + i * 16 + (lane >> 4) * 4 + e;
                    const int c = c0 + (wave & 1) * 32 + j * 16 + (lane & 15);
                    if (oc < K && c < C)
                      dw[(size_t)oc * KDIM + (size_t)tap * C + c] += acc[(((((size_t)wave * 64 + lane) * 9 + tap) * 2 + i) * 2 + j) * 4 + e];
                  }
      }
  long bad = 0;
  for (int k = 0; k < K && bad < 5; ++k)
    for (int r = 0; r < 3; ++r)
      for (int s = 0; s < 3; ++s)
        for (int c = 0; c < C; ++c) {
          int64_t sum = 0;
          for (long m = 0; m < M; ++m) {
            const int n = (int)(m / (IH * IW)), rem = (int)(m % (IH * IW)), op = rem / IW, oq = rem % IW;
            const int ih = op + r - 1, iw = oq + s - 1;
            if (ih < 0 || ih >= IH || iw < 0 || iw >= IW) continue;
            sum += (int64_t)DY[(size_t)m * K + k] * (int64_t)X[(((size_t)n * IH + ih) * IW + iw) * C + c];
          }
          if (dw[(size_t)k * KDIM + (size_t)(r * 3 + s) * C + c] != sum) {
            if (bad < 5) printf("  dw[oc %d][tap %d,%d][c %d] want %lld got %lld\n", k, r, s, c, (long long)sum,
                                (long long)dw[(size_t)k * KDIM + (size_t)(r * 3 + s) * C + c]);
            ++bad;
          }
        }
  printf("N=%d %dx%d C=%d K=%d slices %d: %s%s\n", N, IH, IW, C, K, splits, bad ? "WRONG" : "exact",
         split_mismatch ? " (static/per-patch split differs from the direct offsets)" : "");
  return bad + split_mismatch;
}

// banks of the halo layout under ds_read_b64_tr_b16: a 32-lane half (lanes 0-31 / 32-63) is serviced together, each
// lane reads 8 bytes, bank = (address / 4) % 64
static int bank_check() {
  int conflicts = 0;
  for (int tap = 0; tap < 9; ++tap)
    for (int ks = 0; ks < 2; ++ks)
      for (int h = 0; h < 2; ++h)
        for (int c_local = 0; c_local < 64; c_local += 16)
          for (int half = 0; half < 2; ++half) {
            int seen[64] = {0};
            for (int l = 0; l < 32; ++l) {
              const int lane = half * 32 + l;
              const uint32_t ad = x_frag_lane(lane) + x_frag_const(c_local, ks, h, tap / 3, tap % 3);
              for (int w = 0; w < 2; ++w) if (seen[(ad / 4 + w) % 64]++) ++conflicts;
            }
          }
  printf("bank check, halo rows of %d B x %d columns under the transposing read: %d conflicts\n", kPitch, kPW, conflicts);
  return conflicts;
}

int main() {
  long bad = bank_check();
  bad += run({2, 8, 8, 64, 64, 1});
  bad += run({3, 16, 8, 64, 64, 4});
  bad += run({2, 8, 24, 128, 64, 3});
  bad += run({1, 16, 16, 96, 72, 2});
  bad += run({1, 56, 56, 64, 64, 7});
  bad += run({2, 28, 28, 64, 64, 5});           // patches overhang the image
  bad += run({3, 14, 14, 128, 64, 2});
  bad += run({5, 7, 7, 64, 128, 3});
  bad += run({2, 9, 20, 72, 64, 4});
  printf(bad ? "EMULATION FAILED\n" : "EMULATION OK\n");
  return bad ? 1 : 0;
}
