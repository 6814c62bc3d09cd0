// Host emulator of conv_igemm_halo.hip's data path (tests/test_halo_geometry.py compiles and runs it with g++).
//
// It executes, lane by lane, what the kernel does with ADDRESSES — the LDS-DMA pieces of the halo and of the
// weight tiles (lane -> source offset, LDS written lane-linearly), the ds_read_b128 fragment addresses, the
// operand layout of v_mfma_f32_16x16x32_bf16 and the epilogue's accumulator -> (row, column) map — using the very
// functions the kernel is compiled from (passl_amd/csrc/halo_geom.h), and compares the result with a direct
// convolution on integer data (exact).  It does not model time: waits, barriers and buffer reuse are the ring
// kernel's proven protocol and are checked on the GPU (tools/kbench check igemm_halo=1).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../passl_amd/csrc/halo_geom.h"

using namespace halo;

static uint32_t mix(uint64_t i, uint32_t seed) {
  uint64_t z = i * 0x9E3779B97F4A7C15ull + ((uint64_t)seed << 32 | 0x7F4A7C15u);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)(z >> 33);
}
static int16_t ival(uint64_t i, uint32_t seed) { return (int16_t)((int)(mix(i, seed) % 31u) - 15); }

static int swz32(int row) { return (4 - ((row >> 2) & 3)) & 3; }

struct Case { int N, H, W, C, K; };

// one launch; returns the number of wrong outputs
template <int BN, int CK, bool P2D = false>
static long run(const Case& cs) {
  constexpr int RB = CK * 2, KS = CK / 32, CPRW = RB / 16, RPI = 1024 / RB, WAVES = 4;
  constexpr int WAVES_N = BN == 128 ? 2 : 1, WAVES_M = WAVES / WAVES_N;
  constexpr int WM = 128 / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
  constexpr int B_BYTES = BN * RB, NIB = B_BYTES / 1024 / WAVES;
  const int N = cs.N, IH = cs.H, IW = cs.W, C = cs.C, K = cs.K;
  const long M = (long)N * IH * IW, KDIM = 9L * C;
  std::vector<int16_t> A((size_t)N * IH * IW * C), B((size_t)K * KDIM);
  for (size_t i = 0; i < A.size(); ++i) A[i] = ival(i, 11u);
  for (size_t i = 0; i < B.size(); ++i) B[i] = ival(i, 23u);
  const char* Ab = reinterpret_cast<const char*>(A.data());
  const char* Bb = reinterpret_cast<const char*>(B.data());
  const uint32_t a_bytes = (uint32_t)(A.size() * 2), b_bytes = (uint32_t)(B.size() * 2);

  Geom g;
  g.N = N; g.IH = IH; g.IW = IW; g.PW = IW + 2; g.PH1 = IH + 1; g.opq = IH * IW; g.M = (int)M;
  g.a_sw2 = C * 2; g.a_sh2 = IW * C * 2; g.a_sn2 = IH * IW * C * 2;
  g.hrows = P2D ? kHaloRows2 : halo_rows(IH, IW);
  g.nq = (g.hrows * (CPRW + 1) + 63) / 64;
  Geom2 g2;
  g2.N = N; g2.IH = IH; g2.IW = IW; g2.PXN = IW / 8; g2.PN = (IH / 8) * (IW / 8); g2.npatches = N * g2.PN;
  g2.a_sn2 = g.a_sn2; g2.a_sh2 = g.a_sh2; g2.a_sw2 = g.a_sw2;
  if (P2D) { g2.d_pn = make_fdiv((uint32_t)g2.PN); g2.d_pxn = make_fdiv((uint32_t)g2.PXN); }
  g.d_opq = make_fdiv((uint32_t)g.opq); g.d_iw = make_fdiv((uint32_t)IW);
  g.d_pw = make_fdiv((uint32_t)g.PW); g.d_ph1 = make_fdiv((uint32_t)g.PH1);
  const int nchunks = C / CK;
  const int tiles_m = P2D ? (g2.npatches + 1) / 2 : (int)((M + 127) / 128), tiles_n = (K + BN - 1) / BN;
  if (P2D && tiles_m != (int)((M + 127) / 128)) { printf("tile count mismatch\n"); return 1; }

  std::vector<int32_t> out((size_t)M * K, INT32_MIN);
  std::vector<char> hal((size_t)g.nq * 1024), wt(B_BYTES);
  long halo_rows_used_max = 0;
  for (int mt = 0; mt < tiles_m; ++mt)
    for (int nt = 0; nt < tiles_n; ++nt) {
      const int m0 = mt * 128, n0 = nt * BN;
      const int pbase = P2D ? 0 : padded_index(g, m0) - g.PW - 1;
      // accumulators: [wave][lane][i][j][r]
      std::vector<int32_t> acc((size_t)WAVES * 64 * FM * FN * 4, 0);
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        // ---- halo pieces (buffer_load_dwordx4 ... lds: lane-linear 16-byte writes, out-of-range -> zeros)
        for (int q = 0; q < g.nq; ++q)
          for (int lane = 0; lane < 64; ++lane) {
            uint32_t off = P2D ? halo_src2<CPRW>(g2, mt, q, lane) : halo_src<CPRW>(g, pbase, q, lane);
            char* dst = &hal[(size_t)q * 1024 + lane * 16];
            if (off != kNoSrc) off += (uint32_t)(chunk * RB);
            if (off == kNoSrc || off + 16 > a_bytes) memset(dst, 0, 16);
            else memcpy(dst, Ab + off, 16);
          }
        for (int tap = 0; tap < 9; ++tap) {
          const int r = tap / 3, s = tap % 3;
          const uint32_t koff = (uint32_t)((tap * C + chunk * CK) * 2);
          // ---- weight pieces
          for (int wave = 0; wave < WAVES; ++wave)
            for (int i = 0; i < NIB; ++i)
              for (int lane = 0; lane < 64; ++lane) {
                const int piece = i * WAVES + wave;
                const int row = piece * RPI + lane / CPRW;
                const int col = n0 + row;
                const uint32_t chunkb = (uint32_t)(((lane % CPRW) ^ (CK == 64 ? ((row >> 1) & 7) : swz32(row))) * 16);
                char* dst = &wt[(size_t)piece * 1024 + lane * 16];
                if (col >= K) { memset(dst, 0, 16); continue; }
                const uint32_t off = (uint32_t)col * (uint32_t)(KDIM * 2) + chunkb + koff;
                if (off + 16 > b_bytes) memset(dst, 0, 16); else memcpy(dst, Bb + off, 16);
              }
          // ---- fragments + MFMA
          for (int wave = 0; wave < WAVES; ++wave) {
            const int wm = wave / WAVES_N, wn = wave % WAVES_N;
            for (int ks = 0; ks < KS; ++ks)
              for (int i = 0; i < FM; ++i)
                for (int j = 0; j < FN; ++j) {
                  int16_t pix[64][8], wgt[64][8];
                  for (int lane = 0; lane < 64; ++lane) {
                    const int l15 = lane & 15, l4 = lane >> 4;
                    const uint32_t ad = (P2D ? a_frag_base2<CPRW>(wm * WM + i * 16, l15, l4) + tap_bytes2<CPRW>(r, s)
                                             : a_frag_base<CPRW>(g, m0, wm * WM + i * 16, l15, l4) + tap_bytes<CPRW>(g, r, s)) +
                                        (uint32_t)(64 * ks);
                    if (ad + 16 > hal.size()) { printf("halo read out of the buffer: %u\n", ad); return -1; }
                    if ((long)(ad / ((CPRW + 1) * 16)) > halo_rows_used_max) halo_rows_used_max = ad / ((CPRW + 1) * 16);
                    memcpy(pix[lane], &hal[ad], 16);
                    const int rb = wn * WN + l15;
                    const uint32_t bd = (uint32_t)(rb * RB + ((CK == 64 ? ((ks * 4 + l4) ^ ((rb >> 1) & 7)) : (l4 ^ swz32(rb))) << 4)) +
                                        (uint32_t)(j * 16 * RB);
                    memcpy(wgt[lane], &wt[bd], 16);
                  }
                  // D[row][col] += sum_k Aop[row][k] * Bop[k][col]; Aop = weights (lane l15 = row, l4 = k-group),
                  // Bop = pixels (lane l15 = col); D of lane (l15', l4'), element e: row 4 l4' + e, col l15'
                  for (int lane = 0; lane < 64; ++lane) {
                    const int c15 = lane & 15, r4 = lane >> 4;
                    for (int e = 0; e < 4; ++e) {
                      const int drow = 4 * r4 + e;
                      int32_t sum = 0;
                      for (int kg = 0; kg < 4; ++kg)
                        for (int x = 0; x < 8; ++x) sum += (int32_t)wgt[kg * 16 + drow][x] * (int32_t)pix[kg * 16 + c15][x];
                      acc[((((size_t)wave * 64 + lane) * FM + i) * FN + j) * 4 + e] += sum;
                    }
                  }
                }
          }
        }
      }
      // ---- epilogue map (igemm_epi.h with PERM): row = wm*WM + i*16 + sigma(l15), col = wn*WN + j*16 + l4*4 + e
      for (int wave = 0; wave < WAVES; ++wave) {
        const int wm = wave / WAVES_N, wn = wave % WAVES_N;
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < FM; ++i)
            for (int j = 0; j < FN; ++j)
              for (int e = 0; e < 4; ++e) {
                const int row = wm * WM + i * 16 + sigma(lane & 15), col = wn * WN + j * 16 + (lane >> 4) * 4 + e;
                long m = m0 + row; const int gc = n0 + col;
                if (P2D) {
                  int n, oy, ox;
                  m = out_pixel2(g2, mt, row, n, oy, ox) ? (long)n * IH * IW + (long)oy * IW + ox : M;
                }
                if (m < M && gc < K) out[(size_t)m * K + gc] = acc[((((size_t)wave * 64 + lane) * FM + i) * FN + j) * 4 + e];
              }
      }
    }
  // ---- direct convolution
  long bad = 0;
  for (long m = 0; m < M; ++m) {
    const int n = (int)(m / (IH * IW)), rem = (int)(m % (IH * IW)), op = rem / IW, oq = rem % IW;
    for (int k = 0; k < K; ++k) {
      int32_t sum = 0;
      for (int r = 0; r < 3; ++r) {
        const int ih = op + r - 1;
        if (ih < 0 || ih >= IH) continue;
        for (int s = 0; s < 3; ++s) {
          const int iw = oq + s - 1;
          if (iw < 0 || iw >= IW) continue;
          const int16_t* ap = &A[(((size_t)n * IH + ih) * IW + iw) * C];
          const int16_t* bp = &B[(size_t)k * KDIM + (size_t)(r * 3 + s) * C];
          for (int c = 0; c < C; ++c) sum += (int32_t)ap[c] * (int32_t)bp[c];
        }
      }
      if (out[(size_t)m * K + k] != sum) {
        if (bad < 5) printf("  m=%ld (n=%d op=%d oq=%d) col=%d want %d got %d\n", m, n, op, oq, k, sum, out[(size_t)m * K + k]);
        ++bad;
      }
    }
  }
  printf("%sN=%d %dx%d C=%d K=%d BN=%d CK=%d: hrows %d (highest row read %ld) nq %d -> %s\n", P2D ? "2-D tiles " : "", N, IH, IW, C, K, BN, CK, g.hrows,
         halo_rows_used_max, g.nq, bad ? "WRONG" : "exact");
  if (halo_rows_used_max >= g.hrows) { printf("  a fragment read went past the staged halo rows\n"); ++bad; }
  return bad;
}

// The persistent form (igemm_halo_pw_kernel: C = 64, K <= 64, 2-D tiles): weights of all nine taps resident, the halo
// sources in their lane-static + patch-origin form (cross-checked against halo_src2), tile ranges per workgroup.
static long run_pw(const Case& cs, int nwg) {
  constexpr int CPRW = 8, RB = 128, BN = 64, FM = 2, FN = 4, WM = 32, TAP_BYTES = BN * RB, W_BYTES = 9 * TAP_BYTES;
  const int N = cs.N, IH = cs.H, IW = cs.W, C = cs.C, K = cs.K;
  if (C != 64 || K > 64) return 1;
  const long M = (long)N * IH * IW, KDIM = 9L * C;
  std::vector<int16_t> A((size_t)N * IH * IW * C), B((size_t)K * KDIM);
  for (size_t i = 0; i < A.size(); ++i) A[i] = ival(i, 11u);
  for (size_t i = 0; i < B.size(); ++i) B[i] = ival(i, 23u);
  const char* Ab = reinterpret_cast<const char*>(A.data());
  const char* Bb = reinterpret_cast<const char*>(B.data());
  const uint32_t a_bytes = (uint32_t)(A.size() * 2), b_bytes = (uint32_t)(B.size() * 2);
  Geom2 g2;
  g2.N = N; g2.IH = IH; g2.IW = IW; g2.PXN = IW / 8; g2.PN = (IH / 8) * (IW / 8); g2.npatches = N * g2.PN;
  g2.a_sw2 = C * 2; g2.a_sh2 = IW * C * 2; g2.a_sn2 = IH * IW * C * 2;
  g2.d_pn = make_fdiv((uint32_t)g2.PN); g2.d_pxn = make_fdiv((uint32_t)g2.PXN);
  const int nq = (kHaloRows2 * (CPRW + 1) + 63) / 64, ntiles = (g2.npatches + 1) / 2;
  std::vector<int32_t> out((size_t)M * K, INT32_MIN);
  std::vector<char> wts(W_BYTES), hal((size_t)nq * 1024);
  // weights: piece = tap * 8 + r8, lane -> row 8 r8 + lane / 8, source chunk (lane % 8) ^ ((row >> 1) & 7)
  for (int piece = 0; piece < 72; ++piece)
    for (int lane = 0; lane < 64; ++lane) {
      const int tap = piece >> 3, row = (piece & 7) * 8 + lane / CPRW;
      const uint32_t chunk = (uint32_t)(((lane % CPRW) ^ ((row >> 1) & 7)) * 16);
      char* dst = &wts[(size_t)piece * 1024 + lane * 16];
      const uint32_t off = (uint32_t)row * (uint32_t)(KDIM * 2) + (uint32_t)(tap * C * 2) + chunk;
      if (row >= K || off + 16 > b_bytes) memset(dst, 0, 16); else memcpy(dst, Bb + off, 16);
    }
  long mism = 0, covered = 0;
  for (int wg = 0; wg < nwg; ++wg) {
    const int t_begin = (int)(((int64_t)wg * ntiles) / nwg), t_end = (int)(((int64_t)(wg + 1) * ntiles) / nwg);
    for (int t = t_begin; t < t_end; ++t) {
      ++covered;
      int n0, y0, x0, n1, y1, x1;
      const bool ok0 = patch_origin(g2, 2 * t, n0, y0, x0), ok1 = patch_origin(g2, 2 * t + 1, n1, y1, x1);
      const uint32_t base0 = patch_base2(g2, n0, y0, x0), base1 = patch_base2(g2, n1, y1, x1);
      for (int q = 0; q < nq; ++q)
        for (int lane = 0; lane < 64; ++lane) {
          int byx;
          const uint32_t st = halo_static2<CPRW>(g2, q, lane, byx);
          const bool b = byx & 1;
          const bool in = halo_inside2(g2, b ? y1 : y0, b ? x1 : x0, b ? ok1 : ok0, byx);
          const uint32_t off = in ? st + (b ? base1 : base0) : kNoSrc;
          if (off != halo_src2<CPRW>(g2, t, q, lane)) ++mism;
          char* dst = &hal[(size_t)q * 1024 + lane * 16];
          if (off == kNoSrc || off + 16 > a_bytes) memset(dst, 0, 16); else memcpy(dst, Ab + off, 16);
        }
      std::vector<int32_t> acc((size_t)4 * 64 * FM * FN * 4, 0);
      for (int tap = 0; tap < 9; ++tap)
        for (int wave = 0; wave < 4; ++wave)
          for (int ks = 0; ks < 2; ++ks)
            for (int i = 0; i < FM; ++i)
              for (int j = 0; j < FN; ++j) {
                int16_t pix[64][8], wgt[64][8];
                for (int lane = 0; lane < 64; ++lane) {
                  const int l15 = lane & 15, l4 = lane >> 4;
                  const uint32_t ad = a_frag_base2<CPRW>(wave * WM + i * 16, l15, l4) + tap_bytes2<CPRW>(tap / 3, tap % 3) + (uint32_t)(64 * ks);
                  memcpy(pix[lane], &hal[ad], 16);
                  const uint32_t base = (uint32_t)(l15 * RB + (((ks * 4 + l4) ^ ((l15 >> 1) & 7)) << 4)) + (tap < 4 ? 0u : (uint32_t)(4 * TAP_BYTES));
                  const uint32_t bd = base + (uint32_t)((tap < 4 ? tap : tap - 4) * TAP_BYTES + j * 16 * RB);
                  memcpy(wgt[lane], &wts[bd], 16);
                }
                for (int lane = 0; lane < 64; ++lane)
                  for (int e = 0; e < 4; ++e) {
                    int32_t sum = 0;
                    for (int kg = 0; kg < 4; ++kg)
                      for (int x = 0; x < 8; ++x) sum += (int32_t)wgt[kg * 16 + 4 * (lane >> 4) + e][x] * (int32_t)pix[kg * 16 + (lane & 15)][x];
                    acc[((((size_t)wave * 64 + lane) * FM + i) * FN + j) * 4 + e] += sum;
                  }
              }
      for (int wave = 0; wave < 4; ++wave)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < FM; ++i)
            for (int j = 0; j < FN; ++j)
              for (int e = 0; e < 4; ++e) {
                const int row = wave * WM + i * 16 + sigma(lane & 15), col = j * 16 + (lane >> 4) * 4 + e;
                int rb, ry, rx, n, py0, px0;
                patch_row(row, rb, ry, rx);
                if (!patch_origin(g2, 2 * t + rb, n, py0, px0) || col >= K) continue;
                out[((size_t)n * IH * IW + (size_t)(py0 + ry) * IW + px0 + rx) * K + col] = acc[((((size_t)wave * 64 + lane) * FM + i) * FN + j) * 4 + e];
              }
    }
  }
  long bad = 0;
  for (long m = 0; m < M; ++m) {
    const int n = (int)(m / (IH * IW)), rem = (int)(m % (IH * IW)), op = rem / IW, oq = rem % IW;
    for (int k = 0; k < K; ++k) {
      int32_t sum = 0;
      for (int r = 0; r < 3; ++r) {
        const int ih = op + r - 1;
        if (ih < 0 || ih >= IH) continue;
        for (int s = 0; s < 3; ++s) {
          const int iw = oq + s - 1;
          if (iw < 0 || iw >= IW) continue;
          const int16_t* ap = &A[(((size_t)n * IH + ih) * IW + iw) * C];
          const int16_t* bp = &B[(size_t)k * KDIM + (size_t)(r * 3 + s) * C];
          for (int c = 0; c < C; ++c) sum += (int32_t)ap[c] * (int32_t)bp[c];
        }
      }
      if (out[(size_t)m * K + k] != sum) ++bad;
    }
  }
  printf("persistent form N=%d %dx%d C=%d K=%d, %d workgroups: %ld tiles of %d -> %s%s\n", N, IH, IW, C, K, nwg, covered, ntiles,
         bad ? "WRONG" : "exact", mism ? " (split halo sources differ from halo_src2)" : "");
  return bad + mism + (covered != ntiles);
}

// bank check: the 16 lanes of every ds_read_b128 service group must hit 16 different 16-byte slots of the 256-byte
// bank row, for every tile-relative start row and tap offset (rows of one fragment consecutive)
template <int CPRW>
static int bank_check() {
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int conflicts = 0;
  for (int start = 0; start < 64; ++start)
    for (int gi = 0; gi < 4; ++gi) {
      int seen[16] = {0};
      for (int x = 0; x < 16; ++x) {
        const int lane = groups[gi][x], l15 = lane & 15, l4 = lane >> 4;
        const uint32_t ad = (uint32_t)(start + sigma(l15)) * ((CPRW + 1) * 16) + (uint32_t)swap01(l4) * 16u;
        const int slot = (ad / 16) % 16;
        if (seen[slot]++) ++conflicts;
      }
    }
  printf("bank check, pitch %d B: %d conflicts over 64 start rows x 4 lane groups\n", (CPRW + 1) * 16, conflicts);
  return conflicts;
}

// the same for the 2-D tiles: a fragment = patch rows yy and yy + 4, every tap offset
template <int CPRW>
static int bank_check2() {
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int conflicts = 0;
  for (int frag = 0; frag < 8; ++frag)
    for (int tap = 0; tap < 9; ++tap)
      for (int gi = 0; gi < 4; ++gi) {
        int seen[16] = {0};
        for (int x = 0; x < 16; ++x) {
          const int lane = groups[gi][x];
          const uint32_t ad = a_frag_base2<CPRW>(frag * 16, lane & 15, lane >> 4) + tap_bytes2<CPRW>(tap / 3, tap % 3);
          if (seen[(ad / 16) % 16]++) ++conflicts;
        }
      }
  printf("bank check, 2-D tiles, pitch %d B: %d conflicts over 8 fragments x 9 taps x 4 lane groups\n", (CPRW + 1) * 16, conflicts);
  return conflicts;
}

int main() {
  long bad = 0;
  bad += bank_check<8>();
  bad += bank_check<4>();
  bad += bank_check2<8>();
  bad += bank_check2<4>();
  bad += run<64, 64, true>({3, 56, 56, 64, 64});
  bad += run<64, 32, true>({1, 56, 56, 64, 40});
  bad += run<128, 32, true>({3, 8, 24, 96, 136});
  bad += run<128, 64, true>({5, 16, 8, 128, 72});
  bad += run<64, 64, true>({1, 8, 8, 64, 64});
  bad += run_pw({3, 56, 56, 64, 64}, 7);
  bad += run_pw({5, 8, 16, 64, 40}, 3);
  bad += run_pw({1, 8, 8, 64, 64}, 4);
  { int seen = 0; for (int i = 0; i < 16; ++i) seen |= 1 << sigma(i); if (seen != 0xffff) { printf("sigma is not a permutation\n"); ++bad; } }
  for (int c = 0; c < 8; ++c) if (swap01(swap01(c)) != c) { printf("swap01 is not an involution\n"); ++bad; }
  // images that tiles cross (9408 rows = 73.5 tiles), odd widths, one image smaller than a tile, several chunks
  bad += run<64, 64>({3, 56, 56, 64, 64});
  bad += run<64, 32>({1, 56, 56, 64, 64});
  bad += run<128, 32>({2, 28, 28, 128, 128});
  bad += run<128, 64>({2, 28, 28, 128, 128});
  bad += run<128, 32>({4, 14, 14, 256, 128});
  bad += run<128, 64>({9, 7, 7, 192, 136});
  bad += run<64, 64>({5, 9, 20, 64, 40});
  bad += run<128, 32>({3, 20, 9, 96, 72});
  bad += run<64, 64>({130, 1, 1, 64, 8});
  bad += run<64, 32>({2, 3, 50, 32, 64});
  printf(bad ? "EMULATION FAILED\n" : "EMULATION OK\n");
  return bad ? 1 : 0;
}
