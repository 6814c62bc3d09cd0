// Implicit-GEMM 3x3 convolution with a spatially tiled input operand (gfx950), OPT-IN
// (passl_hip_set_option("igemm_halo", 1) / PASSL_IGEMM_HALO=1; tools/kbench checks and times it).
//
// Same contract as igemm_ring_kernel (conv_igemm_ring.hip) for 3x3 / stride 1 / pad 1 layers over a dense NHWC
// input: Y[m][col] = epi(sum_{r,s,c} A[m @ (r,s)][c] * B[col][r][s][c]), 128-row output tiles of the
// flattened (n, op, oq) order, the shared epilogue of igemm_epi.h (affine, residual, ReLU, fused BatchNorm
// statistics / BatchNorm-backward statistics).  What differs is how the input operand reaches the MFMAs.
//
// The ring kernel stages the im2col operand tap by tap: every input pixel of a tile travels L2 -> LDS NINE
// times (profiles/r04_conv_layers.txt: the stage-1/2 layers run at 620-720 TFLOP/s with 14 TB/s of LDS-DMA
// traffic and idle MFMAs).  Here the pixels a tile needs are staged ONCE per 64-channel chunk — a run of rows
// of the zero-padded image, halo_geom.h — and the nine taps read their fragments from that one LDS image at a
// constant row distance: a tap costs a weight tile (BN x 128 B) and nothing else.  LDS-DMA bytes per tile and
// chunk at 56 x 56 / 64 channels: 9 x 16 KB + 9 x 8 KB = 216 KB before, 44 KB + 72 KB = 116 KB now.
//
//  * workgroup = 4 waves, 128 x BN output tile (BN = 128: 2 x 2 waves of 64 x 64; BN = 64: 4 x 1 of 32 x 64);
//  * LDS: two weight stages (BN x 128 B each) | one or two halo buffers (two when C > 64: the next chunk is
//    staged while the current one is multiplied) | 128 output row offsets;
//  * K order: chunk-major, then tap (r, s): k-tile t = chunk * 9 + tap; fp32 accumulation over all of them;
//  * per k-tile the ring protocol with two stages: vmcnt(0) -> barrier -> issue weights of tile t+2 (+ one
//    slice of the next chunk's halo) -> read the fragments of tile t+1 into the other register set -> 16 / 8
//    MFMAs per k-step of tile t -> lgkmcnt(0).  Fragment reads are inline-asm ds_read_b128 (hipcc would order
//    compiler-visible LDS reads behind every outstanding LDS-DMA);
//  * the halo layout (odd row pitch, lane -> row map sigma, k-group position swap) is conflict-free for every
//    tap: halo_geom.h.  The epilogue is told about sigma (PERM = true).
#include "igemm_dma.h"
#include "igemm_epi.h"
#include <stdio.h>
#include "halo_geom.h"

namespace halo {

using ring::bf16x8_t;
using ring::u32x4;
using ring::lds_read_b128;

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
  uint32_t a_bytes, b_bytes;
  int M, NCOLS, KDIM, C;
  int64_t y_sn, y_sh, y_sw;
  int relu;
  int tiles_n, ntiles;
  int nchunks;                 // C / CK
  uint32_t hbytes;             // bytes of one halo buffer = g.nq * 1024
  FDiv d_tn;
  Geom g;
  Geom2 g2;                    // 2-D tiles (P2D): two 8 x 8 patches per tile
  unsigned long long* stamps;  // NULL, or 8 time stamps (100 MHz) per tile: igemm_halo_dbg
};

// 64-byte weight rows (CK = 32): the ring kernel's slot rotation (conv_igemm_ring.hip: swz32)
__device__ __forceinline__ int swz32(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <int BN, int CK, int MINB, int STAGES, bool P2D>
__global__ void __launch_bounds__(256, MINB) igemm_halo_kernel(const Params p) {
  constexpr int BM = kBM;
  constexpr int RB = CK * 2;                 // bytes of a weight-tile row = of the channel chunk of a halo row
  constexpr int KS = CK / 32;                // 16x16x32 MFMA k-steps per k-tile
  constexpr int CPRW = RB / 16;              // 16-byte chunks per row
  constexpr int RPI = 1024 / RB;             // weight rows per DMA piece
  static_assert(CK == 64 || CK == 32, "channel chunk");
  constexpr int kThreads = 256, WAVES = 4;
  constexpr int WAVES_N = BN == 128 ? 2 : 1, WAVES_M = WAVES / WAVES_N;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int FM = WM / 16, FN = WN / 16;
  static_assert(FN == 4 && (FM == 4 || FM == 2), "wave tile is 64 or 32 rows x 64 columns");
  constexpr int B_BYTES = BN * RB;
  constexpr int NIB = B_BYTES / 1024 / WAVES;        // weight DMA pieces per wave per k-tile (4, 2 or 1)
  static_assert(NIB >= 1, "a k-tile of weights is at least one DMA piece per wave");
  // halo DMA pieces per wave, at most (host: nq <= 4 * NHI): up to 310 rows of 9 / 5 slots, or the 200 rows of two patches
  constexpr int NHI = P2D ? (CK == 64 ? 8 : 4) : (CK == 64 ? 11 : 7);
  static_assert(STAGES >= 2 && STAGES <= 4, "vmcnt bookkeeping covers 2 to 4 weight stages");
  // the NEXT chunk's halo is issued in slices behind the weights of taps 0 .. NSL-1; the last slice must be older
  // than the (STAGES - 2) weight tiles the counted waits leave in flight when the chunk's last tap is reached
  constexpr int NSL = 10 - STAGES;
  constexpr int PIECE = (NHI + NSL - 1) / NSL;
  constexpr int HOFF = STAGES * B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const int nbuf = p.nchunks > 1 ? 2 : 1;
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + HOFF + nbuf * p.hbytes);

  // ---- XCD-aware tile mapping (as the ring kernel: an XCD works on a contiguous range of tiles, so the halo
  // rows two neighbouring tiles share are found in that XCD's L2)
  int tile;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    tile = start + local;
  }
  const int mt = fdiv(tile, p.d_tn), nt = tile - mt * p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l15 = lane & 15, l4 = lane >> 4;
  const Geom& g = p.g;
  auto stamp = [&](int k) { if (p.stamps && tid == 0) p.stamps[(size_t)tile * 8 + k] = __builtin_amdgcn_s_memrealtime(); };
  stamp(0);

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, p.b_bytes, 0x00020000);

  // ---- static DMA geometry
  const int pbase = P2D ? 0 : padded_index(g, m0) - g.PW - 1;
  uint32_t h_src[NHI];                       // halo piece q = i * 4 + wave: this lane's source (channel chunk 0)
#pragma unroll
  for (int i = 0; i < NHI; ++i) {
    const int q = i * WAVES + wave;
    if constexpr (P2D) h_src[i] = q < g.nq ? halo_src2<CPRW>(p.g2, mt, q, lane) : kNoSrc;
    else h_src[i] = q < g.nq ? halo_src<CPRW>(g, pbase, q, lane) : kNoSrc;
  }
  uint32_t b_off[NIB];                       // weight piece i: rows 8 (i*4 + wave) .., the ring kernel's swizzle
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int row = (i * WAVES + wave) * RPI + lane / CPRW;
    const int col = n0 + row;
    const uint32_t chunk = (uint32_t)(((lane % CPRW) ^ (CK == 64 ? ((row >> 1) & 7) : swz32(row))) * 16);
    b_off[i] = col < p.NCOLS ? (uint32_t)col * (uint32_t)(p.KDIM * 2) + chunk : kNoSrc;
  }
  if (tid < BM) {                            // output row offsets (elements) for the epilogue; -1 = out of range
    const int m = m0 + tid;
    int64_t off = -1;
    if constexpr (P2D) {
      int n, oy, ox;
      if (out_pixel2(p.g2, mt, tid, n, oy, ox)) off = (int64_t)n * p.y_sn + (int64_t)oy * p.y_sh + (int64_t)ox * p.y_sw;
    } else if (m < p.M) {
      const int n = fdiv(m, g.d_opq);
      const int rem = m - n * g.opq;
      const int op = fdiv(rem, g.d_iw);
      const int oq = rem - op * g.IW;
      off = (int64_t)n * p.y_sn + (int64_t)op * p.y_sh + (int64_t)oq * p.y_sw;
    }
    rowoff[tid] = off;
  }

  auto issue_halo = [&](auto I, int chunk) {               // piece I of this wave, channel chunk `chunk`
    constexpr int i = decltype(I)::value;
    const int q = i * WAVES + wave;
    if (q < g.nq) {
      const uint32_t off = h_src[i] == kNoSrc ? kNoSrc : h_src[i] + (uint32_t)(chunk * RB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_a, (__attribute__((address_space(3))) void*)(smem + HOFF + (chunk & 1) * p.hbytes + q * 1024), 16, off, 0, 0, 0);
    }
  };

  const int nk = 9 * p.nchunks;
  // weights of the NEXT k-tile to issue: tile index, its tap and chunk (wave-uniform)
  int w_t = 0, w_tap = 0, w_chunk = 0;
  auto issue_b = [&]() {
    char* Bb = smem + (w_t % STAGES) * B_BYTES;
    const uint32_t koff = (uint32_t)((w_tap * p.C + w_chunk * CK) * 2);   // byte offset of the k-tile in a weight row
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const uint32_t off = b_off[i] == kNoSrc ? kNoSrc : b_off[i] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_b, (__attribute__((address_space(3))) void*)(Bb + (i * WAVES + wave) * 1024), 16, off, 0, 0, 0);
    }
    ++w_t;
    if (++w_tap == 9) { w_tap = 0; ++w_chunk; }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read addresses
  uint32_t a_rd[FM];                         // halo: per fragment (its 16 rows need not be 16 halo rows apart)
#pragma unroll
  for (int i = 0; i < FM; ++i)
    a_rd[i] = P2D ? a_frag_base2<CPRW>(wm * WM + i * 16, l15, l4) : a_frag_base<CPRW>(g, m0, wm * WM + i * 16, l15, l4);
  uint32_t b_rd[KS];                         // weights: the ring kernel's addresses
  {
    const int rb = wn * WN + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      b_rd[ks] = (uint32_t)(rb * RB + ((CK == 64 ? ((ks * 4 + l4) ^ ((rb >> 1) & 7)) : (l4 ^ swz32(rb))) << 4));
  }

  u32x4 af[2][KS][FM], bfr[2][KS][FN];
  // tile whose fragments are read next: its tap (r, s) and chunk
  int f_t = 0, f_r = 0, f_s = 0, f_chunk = 0;
  auto read_frags = [&](auto SET) {
    constexpr int S_ = decltype(SET)::value;
    const uint32_t hb = lds0 + (uint32_t)HOFF + (uint32_t)(f_chunk & 1) * p.hbytes +
                        (P2D ? tap_bytes2<CPRW>(f_r, f_s) : tap_bytes<CPRW>(g, f_r, f_s));
    const uint32_t sb = lds0 + (uint32_t)((f_t % STAGES) * B_BYTES);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const uint32_t ad = hb + a_rd[i];
      af[S_][0][i] = lds_read_b128<0>(ad);
      if constexpr (KS == 2) af[S_][1][i] = lds_read_b128<64>(ad);     // chunk positions 4..7: k-step 1
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bfr[S_][ks][0] = lds_read_b128<0>(sb + b_rd[ks]);
      bfr[S_][ks][1] = lds_read_b128<16 * RB>(sb + b_rd[ks]);
      bfr[S_][ks][2] = lds_read_b128<32 * RB>(sb + b_rd[ks]);
      bfr[S_][ks][3] = lds_read_b128<48 * RB>(sb + b_rd[ks]);
    }
    ++f_t;
    if (++f_s == 3) { f_s = 0; if (++f_r == 3) { f_r = 0; ++f_chunk; } }
  };

  // Iteration t (k-tile t = chunk * 9 + tap).  Before it: weights of tiles <= t+1 issued, the halo of chunk(t)
  // resident, tile t's fragments in register set t & 1.  vmcnt(0): this wave's weights of tile t+1 (and its
  // slices of the next halo) landed -> barrier: everybody's did, and everybody finished reading weight stage
  // t & 1 (tile t, read during iteration t-1) -> issue weights of tile t+2 into that stage -> taps 0..7 of a chunk
  // also issue one slice of the NEXT chunk's halo into the other halo buffer (last read for tile 9*chunk - 1,
  // i.e. during iteration 9*chunk - 2: free since the barrier of iteration 9*chunk - 1; complete — vmcnt(0) +
  // barrier of iteration 9*chunk + 8 — before the first fragment read of the next chunk in that iteration).
  int c_tap = 0, c_chunk = 0;                // tap / chunk of tile t
  auto iteration = [&](auto SET, int t) {
    constexpr int S_ = decltype(SET)::value;
    if (t + 1 < nk) {
      // weights of tile t+1 landed: the min(STAGES - 2, nk - 2 - t) tiles issued after it may stay in flight (halo
      // slices sit between the weight tiles in issue order: the count is conservative, never short)
      if (STAGES >= 4 && t + 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIB) : "memory");
      else if (STAGES >= 3 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + STAGES < nk) issue_b();
      if (c_chunk + 1 < p.nchunks && c_tap < NSL) {
        // slice c_tap: pieces c_tap * PIECE .. + PIECE - 1 (PIECE = 2: compile-time indices by a switch)
        switch (c_tap) {
#define PASSL_HALO_SLICE(T)                                                                              \
  case T:                                                                                                \
    if constexpr (T * PIECE < NHI) issue_halo(std::integral_constant<int, (T * PIECE < NHI ? T * PIECE : 0)>{}, c_chunk + 1);          \
    if constexpr (T * PIECE + 1 < NHI && PIECE > 1)                                                      \
      issue_halo(std::integral_constant<int, (T * PIECE + 1 < NHI ? T * PIECE + 1 : 0)>{}, c_chunk + 1);  \
    break;
          PASSL_HALO_SLICE(0) PASSL_HALO_SLICE(1) PASSL_HALO_SLICE(2) PASSL_HALO_SLICE(3)
          PASSL_HALO_SLICE(4) PASSL_HALO_SLICE(5) PASSL_HALO_SLICE(6) PASSL_HALO_SLICE(7)
#undef PASSL_HALO_SLICE
          default: break;
        }
      }
      read_frags(std::integral_constant<int, 1 - S_>{});
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // operands swapped (A := weight fragment): acc[i][j][r] = C[row sigma(l15)][col = .. + l4*4 + r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, bfr[S_][ks][j]), __builtin_bit_cast(bf16x8_t, af[S_][ks][i]),
              acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (++c_tap == 9) { c_tap = 0; ++c_chunk; }
  };

  stamp(1);
  // ---- prologue: the whole halo of chunk 0, weights of tiles 0 .. STAGES-1 (nk >= 9)
  static_assert(PIECE <= 2 && NSL <= 8, "the slice switch issues at most two pieces in taps 0..7");
#define PASSL_HALO_ALL(I) if constexpr (I < NHI) issue_halo(std::integral_constant<int, (I < NHI ? I : 0)>{}, 0);
  PASSL_HALO_ALL(0) PASSL_HALO_ALL(1) PASSL_HALO_ALL(2) PASSL_HALO_ALL(3) PASSL_HALO_ALL(4) PASSL_HALO_ALL(5)
  PASSL_HALO_ALL(6) PASSL_HALO_ALL(7) PASSL_HALO_ALL(8) PASSL_HALO_ALL(9) PASSL_HALO_ALL(10)
#undef PASSL_HALO_ALL
  static_assert(NHI <= 11, "the prologue lists the pieces explicitly");
#pragma unroll
  for (int t = 0; t < STAGES; ++t) issue_b();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * NIB) : "memory");   // halo + tile 0 landed; newer tiles in flight
  __builtin_amdgcn_s_barrier();
  stamp(2);
  read_frags(std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    iteration(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) iteration(std::integral_constant<int, 1>{}, kt + 1);
  }
  __syncthreads();      // all fragment reads done before the LDS is reused as the output tile
  stamp(3);

  epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN, false, FN, 0, true, true>(
      p, smem, rowoff, acc, wm, wn, lane, tid, n0, mt, p.stamps ? p.stamps + (size_t)tile * 8 : nullptr);
  stamp(4);
}

// ------------------------------------------------------------------------------------------------------------
// Persistent form for the stage-1 shape (C = 64, NCOLS <= 64, 2-D tiles; option igemm_halo = 2).  What the time
// stamps of the kernel above say: of the 10 us a tile spends in a workgroup, 0.96 us are MFMAs; 2.6 us
// are setup, 0.8 us waiting for the halo, and in the main loop every k-tile pays for its weight DMA and a barrier.
// Here a workgroup stays on its CU and walks a contiguous range of tiles:
//   * all nine weight taps (9 x 64 rows x 128 B = 72 KB) are staged ONCE per workgroup;
//   * the halo of tile T+1 (two 10 x 10 blocks, 29 KB) is fetched while tile T is multiplied and stored: two
//     buffers; its source offsets are a lane-static part plus two patch origins per tile (no divisions per lane);
//   * the main loop of a tile is 9 taps x 16 MFMAs per wave with NO DMA and NO barrier in it (fragment reads of tap
//     t+1 overlap the MFMAs of tap t);
//   * the shared epilogue runs out of the tile's own halo buffer, which is dead by then.
// LDS: 72 KB weights | 2 x 29 KB halo | 1 KB row offsets = 131 KB: one workgroup (4 waves) per CU.
__global__ void __launch_bounds__(256, 1) igemm_halo_pw_kernel(const Params p) {
  constexpr int BM = kBM, BN = 64, RB = 128, CPRW = 8;
  constexpr int kThreads = 256, WAVES = 4;
  constexpr int WM = 32, WN = 64, FM = 2, FN = 4, KS = 2;
  constexpr int TAP_BYTES = BN * RB;             // 8 KB of weights per tap
  constexpr int W_BYTES = 9 * TAP_BYTES;
  constexpr int NHI = 8;                         // halo pieces per wave (29 per tile)
  constexpr int NWI = 9 * BN * RB / 1024 / WAVES;   // weight pieces per wave (18)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + W_BYTES + 2 * p.hbytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const Geom2& g2 = p.g2;

  // this workgroup's tiles: a contiguous range (neighbouring tiles share halo rows: same L2, one after the other)
  const int t_begin = (int)(((int64_t)blockIdx.x * p.ntiles) / gridDim.x);
  const int t_end = (int)(((int64_t)(blockIdx.x + 1) * p.ntiles) / gridDim.x);
  if (t_begin >= t_end) return;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, p.a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, p.b_bytes, 0x00020000);

  // ---- the weights, once: piece = tap * 8 + r8 covers rows 8 r8 .. 8 r8 + 7 of tap `tap` (the ring kernel's row swizzle)
#pragma unroll
  for (int i = 0; i < NWI; ++i) {
    const int piece = i * WAVES + wave;
    const int tap = piece >> 3, row = (piece & 7) * 8 + lane / CPRW;
    const uint32_t chunk = (uint32_t)(((lane % CPRW) ^ ((row >> 1) & 7)) * 16);
    const uint32_t off = row < p.NCOLS ? (uint32_t)row * (uint32_t)(p.KDIM * 2) + (uint32_t)(tap * p.C * 2) + chunk : kNoSrc;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, off, 0, 0, 0);
  }

  // ---- lane-static geometry
  uint32_t h_st[NHI];
  int h_byx[NHI];
#pragma unroll
  for (int i = 0; i < NHI; ++i) h_st[i] = halo_static2<CPRW>(g2, i * WAVES + wave, lane, h_byx[i]);
  uint32_t a_rd[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_rd[i] = a_frag_base2<CPRW>(wave * WM + i * 16, l15, l4);
  uint32_t b_rd[KS][2];                       // [k-step][taps 0..3 | taps 4..8]: the immediate offset field is 16 bits
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    b_rd[ks][0] = lds0 + (uint32_t)(l15 * RB + (((ks * 4 + l4) ^ ((l15 >> 1) & 7)) << 4));
    b_rd[ks][1] = b_rd[ks][0] + (uint32_t)(4 * TAP_BYTES);
  }
  int rb_, ry_, rx_;                          // this thread's output row of a tile (threads 0..127)
  patch_row(tid & 127, rb_, ry_, rx_);

  auto issue_halo = [&](int t, int buf) {
    int n0, y0, x0, n1, y1, x1;
    const bool ok0 = patch_origin(g2, 2 * t, n0, y0, x0), ok1 = patch_origin(g2, 2 * t + 1, n1, y1, x1);
    const uint32_t base0 = patch_base2(g2, n0, y0, x0), base1 = patch_base2(g2, n1, y1, x1);
#pragma unroll
    for (int i = 0; i < NHI; ++i) {
      const int q = i * WAVES + wave;
      if (q < p.g.nq) {
        const bool b = h_byx[i] & 1;
        const bool in = halo_inside2(g2, b ? y1 : y0, b ? x1 : x0, b ? ok1 : ok0, h_byx[i]);
        const uint32_t off = in ? h_st[i] + (b ? base1 : base0) : kNoSrc;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs_a, (__attribute__((address_space(3))) void*)(smem + W_BYTES + buf * p.hbytes + q * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[FM][FN];
  u32x4 af[2][KS][FM], bfr[2][KS][FN];
  auto read_frags = [&](auto TAP, auto SET, uint32_t hb) {
    constexpr int T_ = decltype(TAP)::value, S_ = decltype(SET)::value;
    constexpr int TB = (int)((T_ / 3) * 10 + (T_ % 3)) * (CPRW + 1) * 16;        // tap_bytes2
    constexpr int WB = (T_ < 4 ? T_ : T_ - 4) * TAP_BYTES;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      af[S_][0][i] = lds_read_b128<TB>(hb + a_rd[i]);
      af[S_][1][i] = lds_read_b128<TB + 64>(hb + a_rd[i]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t wb = b_rd[ks][T_ < 4 ? 0 : 1];
      bfr[S_][ks][0] = lds_read_b128<WB>(wb);
      bfr[S_][ks][1] = lds_read_b128<WB + 16 * RB>(wb);
      bfr[S_][ks][2] = lds_read_b128<WB + 32 * RB>(wb);
      bfr[S_][ks][3] = lds_read_b128<WB + 48 * RB>(wb);
    }
  };
  auto mfma_tap = [&](auto SET) {
    constexpr int S_ = decltype(SET)::value;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, bfr[S_][ks][j]), __builtin_bit_cast(bf16x8_t, af[S_][ks][i]), acc[i][j], 0, 0, 0);
  };

  issue_halo(t_begin, 0);
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    // everything this wave has in flight (the weights, the halo of tile t) landed -> barrier: everybody's did, and
    // everybody is through the epilogue of tile t-1 (its scratch = the buffer the NEXT halo goes to)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t_end) issue_halo(t + 1, buf ^ 1);
    if (tid < BM) {
      int n, y0, x0;
      const bool ok = patch_origin(g2, 2 * t + rb_, n, y0, x0);
      rowoff[tid] = ok ? (int64_t)n * p.y_sn + (int64_t)(y0 + ry_) * p.y_sh + (int64_t)(x0 + rx_) * p.y_sw : (int64_t)-1;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t hb = lds0 + (uint32_t)W_BYTES + (uint32_t)buf * p.hbytes;
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, hb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#define PASSL_PW_TAP(T)                                                                                          \
    if constexpr (T < 8) read_frags(std::integral_constant<int, (T < 8 ? T + 1 : 8)>{}, std::integral_constant<int, (T + 1) & 1>{}, hb); \
    mfma_tap(std::integral_constant<int, T & 1>{});                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
    __builtin_amdgcn_sched_barrier(0);
    PASSL_PW_TAP(0) PASSL_PW_TAP(1) PASSL_PW_TAP(2) PASSL_PW_TAP(3) PASSL_PW_TAP(4)
    PASSL_PW_TAP(5) PASSL_PW_TAP(6) PASSL_PW_TAP(7) PASSL_PW_TAP(8)
#undef PASSL_PW_TAP
    __syncthreads();      // all fragment reads of the tile done: its halo buffer becomes the epilogue's scratch
    epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN, false, FN, 0, true>(p, smem + W_BYTES + buf * p.hbytes, rowoff, acc,
                                                                            wave, 0, lane, tid, 0, t);
  }
}

template <int BN, int CK, int MINB, int STAGES, bool P2D>
int launch(const Params& p, int lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_halo_kernel<BN, CK, MINB, STAGES, P2D>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_halo_kernel<BN, CK, MINB, STAGES, P2D>), dim3(p.ntiles), dim3(256), lds, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

}  // namespace halo

static int g_halo_enabled = -1, g_halo_max_c = 128, g_halo_ck = 0, g_halo_stages = 2, g_halo_2d = 1;
// igemm_halo_dbg: 1 = the next launches write 5 time stamps per tile (kernel entry, setup done, halo landed, main
// loop done, epilogue done; s_memrealtime, 100 MHz); 2 = print what the LAST launch took per phase and switch off
static unsigned long long* g_halo_stamps = nullptr;
static int g_halo_stamp_tiles = 0, g_halo_dbg = 0;
static void halo_report() {
  if (!g_halo_stamps || g_halo_stamp_tiles <= 0) return;
  (void)hipDeviceSynchronize();
  const int n = g_halo_stamp_tiles;
  unsigned long long* h = (unsigned long long*)malloc((size_t)n * 64);
  (void)hipMemcpy(h, g_halo_stamps, (size_t)n * 64, hipMemcpyDeviceToHost);
  double ph[4] = {0, 0, 0, 0}, ep[3] = {0, 0, 0};
  unsigned long long first = ~0ull, last = 0;
  for (int t = 0; t < n; ++t) {
    for (int k = 0; k < 4; ++k) ph[k] += (double)(h[t * 8 + k + 1] - h[t * 8 + k]);
    ep[0] += (double)(h[t * 8 + 5] - h[t * 8 + 3]);      // accumulators -> LDS, barrier
    ep[1] += (double)(h[t * 8 + 6] - h[t * 8 + 5]);      // row stores (+ statistics accumulation)
    ep[2] += (double)(h[t * 8 + 4] - h[t * 8 + 6]);      // statistics: two barriers, fixed-order reduction, slab store
    if (h[t * 8] < first) first = h[t * 8];
    if (h[t * 8 + 4] > last) last = h[t * 8 + 4];
  }
  fprintf(stderr, "halo stamps over %d tiles (us per tile): setup %.2f, halo landed %.2f, main loop %.2f, epilogue %.2f "
                  "(to LDS + barrier %.2f, row stores %.2f, statistics %.2f); whole launch %.1f us => %.2f tiles in flight on average\n",
          n, ph[0] / n / 100.0, ph[1] / n / 100.0, ph[2] / n / 100.0, ph[3] / n / 100.0, ep[0] / n / 100.0, ep[1] / n / 100.0,
          ep[2] / n / 100.0, (double)(last - first) / 100.0, (ph[0] + ph[1] + ph[2] + ph[3]) / (double)(last - first));
  free(h);
}

// passl_hip_set_option("igemm_halo", 0/1) / ("igemm_halo_max_c", n) / ("igemm_halo_ck", 0|32|64)   (runtime.hip)
int passl_igemm_halo_option(const char* name, int value) {
  // 0 off (default), 1 on, 2 = on, and the persistent weights-resident form for C = 64, NCOLS <= 64 with 2-D tiles
  if (!strcmp(name, "igemm_halo")) { g_halo_enabled = value < 0 ? 0 : (value > 2 ? 2 : value); return PASSL_OK; }
  if (!strcmp(name, "igemm_halo_max_c")) { g_halo_max_c = value; return PASSL_OK; }
  if (!strcmp(name, "igemm_halo_dbg")) {
    if (value == 2) halo_report();
    g_halo_dbg = value == 1;
    return PASSL_OK;
  }
  if (!strcmp(name, "igemm_halo_2d")) { g_halo_2d = value != 0; return PASSL_OK; }   // two 8 x 8 patches per tile when IH, IW % 8 == 0
  if (!strcmp(name, "igemm_halo_stages")) {      // depth of the weight ring
    if (value < 2 || value > 4) return PASSL_EINVAL;
    g_halo_stages = value;
    return PASSL_OK;
  }
  if (!strcmp(name, "igemm_halo_ck")) {          // channels per halo chunk: 0 = 64 for C = 64, 32 beyond
    if (value != 0 && value != 32 && value != 64) return PASSL_EINVAL;
    g_halo_ck = value;
    return PASSL_OK;
  }
  return PASSL_EINVAL;
}

// Returns PASSL_EUNSUPPORTED when switched off (the default) or outside the envelope: the caller goes on to the
// 8-phase / ring / register-staged kernels.  The descriptor has been validated by passl_hip_conv_igemm.
int passl_igemm_halo_try(const passl_conv_desc* d, hipStream_t st) {
  if (g_halo_enabled < 0) {
    const char* e = getenv("PASSL_IGEMM_HALO");
    g_halo_enabled = e ? atoi(e) : 0;
    const char* c = getenv("PASSL_IGEMM_HALO_MAX_C");
    if (c) g_halo_max_c = atoi(c);
  }
  if (!g_halo_enabled) return PASSL_EUNSUPPORTED;
  // one 64-channel chunk needs one halo buffer; from two chunks on there are two buffers, and 32-channel chunks
  // keep three workgroups per CU (2 x 18 KB + 2 x 8 KB at 28 x 28 / BN = 128)
  const int CK = g_halo_ck ? g_halo_ck : (d->C == 64 ? 64 : 32);
  if (d->dtype != PASSL_BF16 || d->out_f32) return PASSL_EUNSUPPORTED;
  if (d->R != 3 || d->S != 3 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1) return PASSL_EUNSUPPORTED;
  if (d->IH != d->OP || d->IW != d->OQ) return PASSL_EUNSUPPORTED;
  if ((d->C % CK) != 0 || d->C > g_halo_max_c) return PASSL_EUNSUPPORTED;
  const int64_t lim = 0x7ffffff0ll;
  const int64_t a_bytes = (int64_t)d->N * d->a_sn * 2;
  const int64_t K64 = 9ll * d->C;
  const int64_t b_bytes = (int64_t)d->NCOLS * K64 * 2;
  if (a_bytes <= 0 || a_bytes >= lim || b_bytes >= lim) return PASSL_EUNSUPPORTED;   // one descriptor per operand
  // the halo addresses rows as n * a_sn + ih * a_sh + iw * a_sw: any strides, but the last pixel must be inside
  if ((int64_t)(d->IH - 1) * d->a_sh + (int64_t)(d->IW - 1) * d->a_sw + d->C > d->a_sn) return PASSL_EUNSUPPORTED;
  const int64_t M64 = (int64_t)d->N * d->OP * d->OQ;
  const int bn = d->NCOLS <= 64 ? 64 : 128;
  const int tiles_n = (d->NCOLS + bn - 1) / bn;
  const int64_t tiles_m = (M64 + 127) / 128;
  if (tiles_m * tiles_n > 0x7fffffffll) return PASSL_EUNSUPPORTED;
  // padded indices stay far below 2^31: N * (IH + 1) * (IW + 2)
  if ((int64_t)(d->N + 1) * (d->IH + 1) * (d->IW + 2) >= lim) return PASSL_EUNSUPPORTED;

  halo::Params p;
  p.a = reinterpret_cast<const char*>(d->a);
  p.b = reinterpret_cast<const char*>(d->b);
  p.y = reinterpret_cast<char*>(d->y);
  p.scale = d->scale; p.shift = d->shift;
  p.res = reinterpret_cast<const char*>(d->residual);
  p.stats = d->stats; p.stats_tiles = (int)tiles_m;
  p.bnb_y = reinterpret_cast<const char*>(d->bnb_y); p.bnb_mask = d->bnb_mask;
  p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd;
  p.bnb_scale = d->bnb_scale; p.bnb_shift = d->bnb_shift;
  p.bnb_partial = d->bnb_partial; p.bnb_relu = d->bnb_relu; p.bnb_tile_off = d->bnb_tile_off;
  p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.M = (int)M64; p.NCOLS = d->NCOLS; p.KDIM = (int)K64; p.C = d->C;
  p.y_sn = d->y_sn; p.y_sh = d->y_sh; p.y_sw = d->y_sw;
  p.relu = d->relu;
  p.tiles_n = tiles_n; p.ntiles = (int)(tiles_m * tiles_n);
  p.nchunks = d->C / CK;
  p.d_tn = halo::make_fdiv((uint32_t)tiles_n);
  halo::Geom& g = p.g;
  g.N = d->N; g.IH = d->IH; g.IW = d->IW; g.PW = d->IW + 2; g.PH1 = d->IH + 1;
  g.opq = d->IH * d->IW; g.M = (int)M64;
  g.a_sn2 = (int)(d->a_sn * 2); g.a_sh2 = (int)(d->a_sh * 2); g.a_sw2 = (int)(d->a_sw * 2);
  const bool p2d = g_halo_2d && (d->IH % 8) == 0 && (d->IW % 8) == 0;
  g.hrows = p2d ? halo::kHaloRows2 : halo::halo_rows(d->IH, d->IW);
  g.nq = (g.hrows * (CK / 8 + 1) + 63) / 64;
  const int nhi = p2d ? (CK == 64 ? 8 : 4) : (CK == 64 ? 11 : 7);
  halo::Geom2& g2 = p.g2;
  g2.N = d->N; g2.IH = d->IH; g2.IW = d->IW;
  g2.PXN = p2d ? d->IW / 8 : 1; g2.PN = p2d ? (d->IH / 8) * (d->IW / 8) : 1; g2.npatches = d->N * g2.PN;
  g2.a_sn2 = g.a_sn2; g2.a_sh2 = g.a_sh2; g2.a_sw2 = g.a_sw2;
  g2.d_pn = halo::make_fdiv((uint32_t)g2.PN); g2.d_pxn = halo::make_fdiv((uint32_t)g2.PXN);
  g.d_opq = halo::make_fdiv((uint32_t)g.opq); g.d_iw = halo::make_fdiv((uint32_t)g.IW);
  g.d_pw = halo::make_fdiv((uint32_t)g.PW); g.d_ph1 = halo::make_fdiv((uint32_t)g.PH1);
  if (g.nq > 4 * nhi) return PASSL_EUNSUPPORTED;       // pieces per wave the kernel unrolls (images wider than ~56 columns)
  p.hbytes = (uint32_t)g.nq * 1024u;
  p.stamps = nullptr;
  if (g_halo_dbg) {
    if (g_halo_stamp_tiles < p.ntiles) {
      if (g_halo_stamps) (void)hipFree(g_halo_stamps);
      g_halo_stamps = nullptr;
      if (hipMalloc((void**)&g_halo_stamps, (size_t)p.ntiles * 64) != hipSuccess) g_halo_stamps = nullptr;
    }
    g_halo_stamp_tiles = g_halo_stamps ? p.ntiles : 0;
    p.stamps = g_halo_stamps;
  }
  if (g_halo_enabled == 2 && p2d && CK == 64 && d->C == 64 && d->NCOLS <= 64) {
    static int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      int dev = 0;
      (void)hipGetDevice(&dev);
      cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int lds_pw = 9 * 64 * 128 + 2 * (int)p.hbytes + 1024;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo::igemm_halo_pw_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    const int grid = p.ntiles < cus ? p.ntiles : cus;
    hipLaunchKernelGGL(halo::igemm_halo_pw_kernel, dim3(grid), dim3(256), lds_pw, st, p);
    return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
  }
  const int nbuf = p.nchunks > 1 ? 2 : 1;
  const int stages = g_halo_stages;
  const int lds = stages * bn * CK * 2 + nbuf * (int)p.hbytes + 1024;
  if (lds > 160 * 1024) return PASSL_EUNSUPPORTED;
  // the output tile of the epilogue (128 x (BN + 8) bf16) and its 16 KB reduction scratch live below the row offsets
  if (128 * (bn + 8) * 2 > lds - 1024 || 16 * 1024 > lds - 1024) return PASSL_EUNSUPPORTED;
  // three workgroups per CU when the LDS allows it (the register budget follows: 168 VGPRs; the 128 x 128 tile with
  // 64-channel k-tiles needs more and stays at two)
  const bool three = lds * 3 <= 160 * 1024 && !(bn == 128 && CK == 64);
#define PASSL_HALO_GO(BN_, CK_, MB_, ST_) \
  (p2d ? halo::launch<BN_, CK_, MB_, ST_, true>(p, lds, st) : halo::launch<BN_, CK_, MB_, ST_, false>(p, lds, st))
#define PASSL_HALO_ST(BN_, CK_, MB_) \
  (stages == 2 ? PASSL_HALO_GO(BN_, CK_, MB_, 2) : stages == 3 ? PASSL_HALO_GO(BN_, CK_, MB_, 3) : PASSL_HALO_GO(BN_, CK_, MB_, 4))
  if (bn == 128 && CK == 64) return PASSL_HALO_ST(128, 64, 2);
  if (bn == 128) return three ? PASSL_HALO_ST(128, 32, 3) : PASSL_HALO_ST(128, 32, 2);
  if (CK == 64) return three ? PASSL_HALO_ST(64, 64, 3) : PASSL_HALO_ST(64, 64, 2);
  return three ? PASSL_HALO_ST(64, 32, 3) : PASSL_HALO_ST(64, 32, 2);
#undef PASSL_HALO_ST
#undef PASSL_HALO_GO
}
