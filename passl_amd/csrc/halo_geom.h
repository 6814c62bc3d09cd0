// Index arithmetic of the spatially tiled 3x3 kernel (conv_igemm_halo.hip), kept free of device intrinsics so
// that the SAME functions are compiled into the kernel and into the host emulator
// (tests/emu/halo_emu.cpp, run by tests/test_halo_geometry.py on the CPU): the emulator stages a tile through an
// LDS image with these functions, reads the MFMA operands back with them and compares with a direct
// convolution — every address the kernel forms is checked without a GPU.
//
// The tile.  128 consecutive output rows m0 .. m0+127 of the flattened (n, op, oq) order (the contract of the
// shared epilogue and of the statistics slabs).  For a 3x3 / stride 1 / pad 1 convolution over a dense NHWC
// image the input pixel of output row m and tap (r, s) is the pixel of row m moved by (r-1, s-1).  The input
// pixels a tile needs are staged ONCE per 64-channel chunk as a run of rows of a ZERO-PADDED image:
//
//   padded index   P(n, ih, iw) = (n * (IH + 1) + ih + 1) * (IW + 2) + iw + 1
//
// i.e. one zero column left and right of every image row and ONE zero row between consecutive images (the
// bottom padding of image n is the top padding of image n+1).  In that numbering tap (r, s) of output row m is
// at P(m) + (r-1) * (IW+2) + (s-1): a constant distance — no per-tap gather, no per-tap bounds test, the zero
// padding is physically in LDS (those rows are fetched with an out-of-range buffer offset).
//
//   LDS halo row of padded index P:  P - pbase,   pbase = P(m0) - (IW+2) - 1
//   LDS halo row of (output row m, tap r, s):  P(m) - P(m0) + r * (IW+2) + s
//
// Layout of a halo row: 8 (CK = 64) or 4 (CK = 32) 16-byte channel chunks and one 16-byte pad, i.e. a pitch of
// 144 / 80 bytes = 9 / 5 bank slots: an ODD number, so that 16 rows at any distance of a multiple of 2 fall on 8
// different slots of one parity.  ds_read_b128 services a wave in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32): each group holds the 16 pixel rows of a fragment once, 8 of them with the k-group
// of l4 = 0 and 8 with that of l4 = 1.  With
//   * lane l15 -> pixel row sigma(l15) of the fragment: lanes {0-3, 12-15} take the EVEN rows, {4-11} the odd
//     ones (the epilogue stores with the same map), and
//   * k-group g (8 channels) of a k-step at chunk position swap01(g): groups 0 and 1 sit TWO slots apart,
// the 16 lanes of a group hit 16 different slots for EVERY start row — which a XOR swizzle cannot give for
// rows at an arbitrary (tap-dependent) offset.  The weight operand keeps the ring kernel's layout; lane l4 of
// either operand holds the same 8 channels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HALO_HD __host__ __device__ __forceinline__
#else
#define HALO_HD static inline
#endif

namespace halo {

constexpr int kBM = 128;
constexpr uint32_t kNoSrc = 0x7ffffff0u;        // buffer offset that the bounds check turns into zeros

// lane l15 of an MFMA pixel fragment -> row of the fragment (a permutation of 0..15)
HALO_HD int sigma(int l15) { return l15 < 4 ? 2 * l15 : (l15 < 12 ? 2 * (l15 - 4) + 1 : 2 * (l15 - 8)); }
// chunk <-> position inside a halo row (an involution): bits 0 and 1 of the chunk index exchanged
HALO_HD int swap01(int c) { return (c & ~3) | ((c & 1) << 1) | ((c >> 1) & 1); }

// division by a run-time constant as multiply-high + shifts (the setup code divides per lane and per DMA piece)
struct FDiv { uint32_t mul, sh1, sh2; };
static inline FDiv make_fdiv(uint32_t d) {
  FDiv f = {0, 0, 0};
  if (d > 1) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = 1;
    f.sh2 = l - 1;
  }
  return f;
}
HALO_HD int fdiv(int n, const FDiv f) {
  const uint32_t t = (uint32_t)(((uint64_t)f.mul * (uint64_t)(uint32_t)n) >> 32);
  return (int)((t + (((uint32_t)n - t) >> f.sh1)) >> f.sh2);
}

struct Geom {
  int N, IH, IW;          // image (= output) extent: stride 1, pad 1
  int PW;                 // IW + 2
  int PH1;                // IH + 1
  int opq;                // IH * IW
  int M;                  // N * opq
  int a_sn2, a_sh2, a_sw2;   // BYTE strides of the input
  int hrows;              // halo rows per tile (worst case over tiles)
  int nq;                 // 1 KB LDS-DMA pieces per halo chunk = ceil(hrows * (CPRW + 1) / 64)
  FDiv d_opq, d_iw, d_pw, d_ph1;
};

// padded index of output row m (m < M)
HALO_HD int padded_index(const Geom& g, int m) {
  const int n = fdiv(m, g.d_opq);
  const int rem = m - n * g.opq;
  const int op = fdiv(rem, g.d_iw);
  const int oq = rem - op * g.IW;
  return (n * g.PH1 + op + 1) * g.PW + oq + 1;
}

// worst-case halo rows of a 128-row tile: 127 steps cross at most ceil(127 / IW) image rows (+2 each) and
// ceil(127 / opq) images (+PW each), plus one padded row and one column on either side
HALO_HD int halo_rows(int IH, int IW) {
  const int PW = IW + 2, opq = IH * IW;
  const int rc = (127 + IW - 1) / IW, ic = (127 + opq - 1) / opq;
  return 127 + 2 * rc + PW * ic + 2 * PW + 3;
}

// LDS-DMA piece q (1 KB = 64 lanes x 16 B) of a halo chunk: what lane `lane` fetches.  CPRW = 16-byte chunks per
// row of CK channels.  Returns the byte offset of the source relative to the input tensor (channel chunk 0 of the
// 64/32-channel slice: the caller adds 2 * c0) or kNoSrc for padding / pad slot / rows past the tile's halo.
template <int CPRW>
HALO_HD uint32_t halo_src(const Geom& g, int pbase, int q, int lane) {
  const int p = q * 64 + lane;
  const int hrow = p / (CPRW + 1);
  const int cpos = p - hrow * (CPRW + 1);
  if (cpos == CPRW || hrow >= g.hrows) return kNoSrc;
  const int P = pbase + hrow;
  const int prow = fdiv(P, g.d_pw);
  const int pcol = P - prow * g.PW;
  const int n = fdiv(prow, g.d_ph1);
  const int ihp = prow - n * g.PH1;
  if (pcol < 1 || pcol > g.IW || ihp < 1 || n >= g.N) return kNoSrc;
  return (uint32_t)n * (uint32_t)g.a_sn2 + (uint32_t)(ihp - 1) * (uint32_t)g.a_sh2 +
         (uint32_t)(pcol - 1) * (uint32_t)g.a_sw2 + (uint32_t)swap01(cpos) * 16u;
}

// byte address (relative to the halo buffer) of the 16 bytes lane (l15, l4) reads for fragment rows
// frag_row0 .. frag_row0+15 of the tile, k-step 0, tap (0, 0); the caller adds tap_bytes() and 64 * ks
template <int CPRW>
HALO_HD uint32_t a_frag_base(const Geom& g, int m0, int frag_row0, int l15, int l4) {
  int m = m0 + frag_row0 + sigma(l15);
  if (m > g.M - 1) m = g.M - 1;                    // ragged last tile: any in-range row (never stored)
  const int hrow = padded_index(g, m) - padded_index(g, m0);
  return (uint32_t)hrow * (uint32_t)((CPRW + 1) * 16) + (uint32_t)swap01(l4) * 16u;
}
template <int CPRW>
HALO_HD uint32_t tap_bytes(const Geom& g, int r, int s) { return (uint32_t)(r * g.PW + s) * (uint32_t)((CPRW + 1) * 16); }

// ------------------------------------------------------------------------------------------------------------
// 2-D tiles (IH and IW multiples of 8).  The statistics slabs and the epilogue only need every tile to hold 128
// output rows — not consecutive ones — so a tile may as well be two 8 x 8 PATCHES (consecutive in the global patch
// order n, py, px): their halos are two 10 x 10 blocks = 200 LDS rows instead of up to 310, no divisions by run-time
// numbers per lane, and three workgroups per CU at 56 x 56 x 64.
//
//   tile row r' (0..127):  patch b = r' >> 6,  y = ((r' >> 4) & 3) + 4 * ((r' >> 3) & 1),  x = r' & 7
//   LDS halo row of (r', tap r, s):  b * 100 + (y + r) * 10 + (x + s)
//
// i.e. a 16-row fragment holds the patch rows yy and yy + 4: their 16 LDS rows have 16 different residues mod 16
// (40 = 8 mod 16), even x on even rows — the same lane map sigma and the same slot argument as above.
struct Geom2 {
  int N, IH, IW;
  int PXN, PN;              // patches per image row, per image
  int npatches;             // N * PN
  int a_sn2, a_sh2, a_sw2;
  FDiv d_pn, d_pxn;
};
constexpr int kHaloRows2 = 200;

// pixel of tile row r' inside its patch
HALO_HD void patch_row(int r, int& b, int& y, int& x) {
  b = r >> 6;
  y = ((r >> 4) & 3) + 4 * ((r >> 3) & 1);
  x = r & 7;
}
// image and patch origin of global patch gp; false past the last patch
HALO_HD bool patch_origin(const Geom2& g, int gp, int& n, int& y0, int& x0) {
  n = fdiv(gp, g.d_pn);
  const int rem = gp - n * g.PN;
  const int py = fdiv(rem, g.d_pxn);
  y0 = py * 8;
  x0 = (rem - py * g.PXN) * 8;
  return gp < g.npatches;
}
template <int CPRW>
HALO_HD uint32_t halo_src2(const Geom2& g, int tile, int q, int lane) {
  const int p = q * 64 + lane;
  const int hrow = p / (CPRW + 1);
  const int cpos = p - hrow * (CPRW + 1);
  if (cpos == CPRW || hrow >= kHaloRows2) return kNoSrc;
  const int b = hrow >= 100 ? 1 : 0;
  const int hr = hrow - 100 * b;
  const int hy = hr / 10, hx = hr - 10 * hy;
  int n, y0, x0;
  if (!patch_origin(g, 2 * tile + b, n, y0, x0)) return kNoSrc;
  const int ih = y0 + hy - 1, iw = x0 + hx - 1;
  if (ih < 0 || ih >= g.IH || iw < 0 || iw >= g.IW) return kNoSrc;
  return (uint32_t)n * (uint32_t)g.a_sn2 + (uint32_t)ih * (uint32_t)g.a_sh2 + (uint32_t)iw * (uint32_t)g.a_sw2 +
         (uint32_t)swap01(cpos) * 16u;
}
template <int CPRW>
HALO_HD uint32_t a_frag_base2(int frag_row0, int l15, int l4) {
  int b, y, x;
  patch_row(frag_row0 + sigma(l15), b, y, x);
  return (uint32_t)(b * 100 + y * 10 + x) * (uint32_t)((CPRW + 1) * 16) + (uint32_t)swap01(l4) * 16u;
}
template <int CPRW>
HALO_HD uint32_t tap_bytes2(int r, int s) { return (uint32_t)(r * 10 + s) * (uint32_t)((CPRW + 1) * 16); }
// The persistent form (igemm_halo_pw_kernel) splits halo_src2 into a lane-static part and two patch origins per
// tile: offset = patch_base[b] + halo_static2(q, lane), valid iff halo_inside2(...);  byx = b | hy << 8 | hx << 16, or -1
// for lanes that never fetch (pad slot, rows past the two 10 x 10 blocks).
template <int CPRW>
HALO_HD uint32_t halo_static2(const Geom2& g, int q, int lane, int& byx) {
  const int p = q * 64 + lane;
  const int hrow = p / (CPRW + 1);
  const int cpos = p - hrow * (CPRW + 1);
  if (cpos == CPRW || hrow >= kHaloRows2) { byx = -1; return 0; }
  const int b = hrow >= 100 ? 1 : 0;
  const int hr = hrow - 100 * b;
  const int hy = hr / 10, hx = hr - 10 * hy;
  byx = b | (hy << 8) | (hx << 16);
  return (uint32_t)((hy - 1) * g.a_sh2) + (uint32_t)((hx - 1) * g.a_sw2) + (uint32_t)swap01(cpos) * 16u;   // wraps for hy, hx = 0
}
HALO_HD uint32_t patch_base2(const Geom2& g, int n, int y0, int x0) {
  return (uint32_t)n * (uint32_t)g.a_sn2 + (uint32_t)y0 * (uint32_t)g.a_sh2 + (uint32_t)x0 * (uint32_t)g.a_sw2;
}
// y0 / x0 / ok = origin and existence of the patch the lane's halo row belongs to
HALO_HD bool halo_inside2(const Geom2& g, int y0, int x0, bool ok, int byx) {
  return byx >= 0 && ok && (uint32_t)(y0 + ((byx >> 8) & 255) - 1) < (uint32_t)g.IH &&
         (uint32_t)(x0 + (byx >> 16) - 1) < (uint32_t)g.IW;
}

// flattened output row (n, oy, ox) of tile row r'; false when the patch does not exist (ragged last tile)
HALO_HD bool out_pixel2(const Geom2& g, int tile, int r, int& n, int& oy, int& ox) {
  int b, y, x, y0, x0;
  patch_row(r, b, y, x);
  const bool ok = patch_origin(g, 2 * tile + b, n, y0, x0);
  oy = y0 + y;
  ox = x0 + x;
  return ok;
}

}  // namespace halo
