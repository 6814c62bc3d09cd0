// Implicit-GEMM convolution / Linear, bf16, 256 x 256 output tiles with an 8-phase schedule (gfx950).
//
// Same contract and the same bits as igemm_ring_kernel (conv_igemm_ring.hip): Y[m][col] =
// epi(sum_k A[m][k] B[col][k]), m = (n, op, oq) gathered from an NHWC tensor, k = (r, s, c) walked in
// 64-element K-tiles, fp32 accumulation in ascending k, the shared epilogue of igemm_epi.h (affine,
// residual, ReLU, fused BatchNorm statistics per 128-row half).  What differs is the schedule — the
// structure /opt/skills/guides/cdna_hip_programming.md gives for large GEMMs on this chip:
//
//  * one workgroup per CU: 512 threads = 8 waves as 2 (M) x 4 (N), two waves per SIMD; a wave owns a
//    64-row strip in EACH 128-row half of the A tile and a 32-column strip in EACH 128-column half of
//    the B tile, i.e. four 64 x 32 quadrants (A-half i, B-half j) = 32 accumulator fragments (128 VGPRs);
//  * LDS = 2 K-tile buffers x 4 half-tiles (A0, A1, B0, B1; 128 rows x 128 B = 16 KB each), filled by
//    `buffer_load_dwordx4 ... lds` (2 instructions per wave per half-tile), rows XOR-swizzled through the
//    SOURCE address exactly as in the ring kernel; fragments come out with inline-asm ds_read_b128;
//  * a K-tile is FOUR phases, one quadrant (16 MFMAs per wave) each:
//        q0: read B0 (4) + A0 (8)   issue A1(t+1)   MFMA A0 x B0
//        q1: read B1 (4)            issue B0(t+2)   MFMA A0 x B1
//        q2: read A1 (8)            issue A0(t+2)   MFMA A1 x B1
//        q3: -                      issue B1(t+2)   MFMA A1 x B0      + the only vmcnt wait of the K-tile
//    and a phase is {fragment reads, one half-tile of DMA} | barrier | {16 MFMAs at raised priority} |
//    barrier.  The two wave groups (wr = 0 / 1: one wave of each per SIMD) run ONE BARRIER APART, so that on
//    every SIMD one wave multiplies while the other reads LDS and issues DMA;
//  * hazards (segment s = the interval between barriers s-1 and s; group 0 reads phase p in segment 2p and
//    multiplies in 2p+1, group 1 one segment later):
//      RAW  every wave waits `vmcnt(6)` (three half-tiles of tile t+2 may stay in flight) before the first
//           barrier of q3: its own DMA of tile t+1 has landed; all waves have passed that wait one barrier
//           later, the first read of tile t+1 is two barriers later;
//      WAR  a half-tile is re-staged two phases after its last read (the read is retired by the lgkmcnt(0)
//           behind the phase's first barrier, i.e. before the second barrier of that phase for group 0 and
//           the first barrier of the next phase for group 1), or ONE phase after for B0, whose 4 reads are
//           issued first in q0 and retired by `lgkmcnt(8)` in front of the barrier.
//
//  * two forms.  <DIRECT = false>: one tile per workgroup and the LDS-staged epilogue of igemm_epi.h (needed by
//    the fused statistics).  <DIRECT = true>: PERSISTENT — a workgroup walks its tiles and treats (tile, K-tile)
//    as ONE stream of K-tiles, so the DMA pipeline never drains: while the last two K-tiles of a tile are
//    multiplied the first two of the next tile are already being staged (the DMA geometry registers are
//    switched to the next tile right after the last A1 issue of the current one), and the epilogue goes from
//    the accumulators straight to global memory (affine, pack, v_permlane32_swap to 16-byte pieces, residual,
//    ReLU — the same arithmetic in the same order as the staged epilogue: same bits) with stores that drain
//    under the next tile's main loop.  The weight fragments of this form are read with operand rows 4-7 and
//    8-11 exchanged so that lanes l and l+32 hold adjacent 4-column groups.
//
// Dispatched by passl_hip_conv_igemm ahead of the ring kernel when a cost model says that the 256 x 256 tiles fill
// the 256 CUs well enough (passl_igemm_8p_try).
#include "igemm_dma.h"
#include "igemm_epi.h"

namespace g8 {

using ring::Params;
using ring::bf16x8_t;
using ring::u32x4;
using ring::fdiv;
using ring::kOOB;
using ring::lds_read_b128;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int HALF = 16384;                 // one half-tile: 128 rows x 128 bytes
constexpr int BUF = 2 * HALF;               // per operand and K-tile buffer: two half-tiles
constexpr int A_REGION = 0, B_REGION = 4 * HALF;
constexpr int LDS_TILES = 8 * HALF;         // 128 KB
constexpr int LDS_BYTES = LDS_TILES + BM * 8;

template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef std::integral_constant<int, 0> I0;
typedef std::integral_constant<int, 1> I1;

// DENSE (compile-time): the operand is a plain [M, C] matrix (Linear / 1x1 stride-1 convolution) — the filter-tap
// walk, the per-tap bounds checks and the image decomposition of a row drop out of the instruction stream and of the
// scalar register file (the general form keeps ~20 more scalars live and spills 69 of them to lanes:
// profiles/r03_kernel_resources.txt).  Same arithmetic, same order, same bits (profiles/r04_8p_dense_ab.txt).
// passl_hip_set_option("igemm_8p_dense", v) / PASSL_IGEMM_8P_DENSE=v: 0 off, 1 the persistent form (default),
// 2 also the staged (fused-statistics) form.
template <bool DIRECT, bool DENSE = false>
__global__ void __launch_bounds__(kThreads) igemm_8p_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + LDS_TILES);     // staged epilogue only
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int opq = DENSE ? 1 : p.OP * p.OQ;
  const int nk = p.KDIM / BK;

  // ---- tiles of this workgroup: virtual ids bid, bid + grid, ... through the XCD-aware map (bijective for
  // any ntiles; the grid is a multiple of 8 or equals ntiles, so a workgroup's tiles stay on its XCD's range),
  // column tiles fastest
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int cnt = DIRECT ? (p.ntiles - 1 - bid) / nwg + 1 : 1;
  auto tile_of = [&](int v, int& m0_, int& n0_) __attribute__((always_inline)) {
    const int xcd = v & 7, local = v >> 3;
    const int q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int tile = start + local;
    const int mt = fdiv(tile, p.d_tn);
    m0_ = mt * BM;
    n0_ = (tile - mt * p.tiles_n) * BN;
  };

  // ---- DMA geometry of the tile being STAGED.  Half-tile h, instruction i of this wave covers half rows
  // (i*8 + wave)*8 .. +7: lane -> row + (lane >> 3), LDS slot lane & 7, SOURCE chunk (lane & 7) ^ ((row >> 1) & 7).
  uint32_t a_base[4], b_off[4];
  int ih0[4], iw0[4];
  __amdgpu_buffer_rsrc_t rs_a;
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, p.b_bytes, 0x00020000);
  int r0 = 0, s0 = 0, c0 = 0;       // next K-tile of the A0 stream (wave-uniform taps)
  int r1 = 0, s1 = 0, c1 = 0;       // next K-tile of the A1 stream
  auto set_geometry = [&](int m0, int n0) __attribute__((always_inline)) {
    // A beyond 2 GB: the descriptor starts at the first image (dense: row) of this tile (see the ring kernel)
    const int nb = (DENSE || p.dense) ? 0 : fdiv(m0 < p.M ? m0 : p.M - 1, p.d_opq);
    const int64_t a_off0 = (DENSE || p.dense) ? (int64_t)m0 * (p.C * 2) : (int64_t)nb * p.a_sn2;
    const int64_t a_left = p.a_total - a_off0;
    rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a) + a_off0, 0,
                                             (uint32_t)(a_left < 0x7ffffff0ll ? a_left : 0x7ffffff0ll), 0x00020000);
    // branch-free per lane (selects): the arrays stay in registers
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int hrow = ((x & 1) * 8 + wave) * 8 + (lane >> 3);
      const int m = m0 + (x >> 1) * 128 + hrow;
      const uint32_t chunk = (uint32_t)(((lane & 7) ^ ((hrow >> 1) & 7)) * 16);
      const bool valid = m < p.M;
      const int mc = valid ? m : p.M - 1;
      uint32_t ab;
      int ih = 0, iw = 0;
      if (DENSE || p.dense) {             // wave-uniform
        ab = (uint32_t)(mc - m0) * (uint32_t)(p.C * 2) + chunk;
      } else if constexpr (!DENSE) {
        const int n = fdiv(mc, p.d_opq);
        const int rem = mc - n * opq;
        const int op = fdiv(rem, p.d_oq);
        const int oq = rem - op * p.OQ;
        ih = op * p.sh - p.ph;
        iw = oq * p.sw - p.pw;
        ab = (uint32_t)(n - nb) * (uint32_t)p.a_sn2 + (uint32_t)(ih * p.a_sh2) + (uint32_t)(iw * p.a_sw2) + chunk;
      }
      if constexpr (DENSE) {
        a_base[x] = valid ? ab : kOOB;        // rows past M read zeros
      } else {
        a_base[x] = valid ? ab : 0u;
        ih0[x] = valid ? ih : -(1 << 28);     // fails every tap's bounds check: the DMA reads zeros
        iw0[x] = valid ? iw : 0;
      }
      const int col = n0 + (x >> 1) * 128 + hrow;
      b_off[x] = col < p.NCOLS ? (uint32_t)col * (uint32_t)(p.KDIM * 2) + chunk : kOOB;
    }
    r0 = s0 = c0 = 0;
    r1 = s1 = c1 = 0;
  };

  int cm0, cn0;                     // the tile being MULTIPLIED
  tile_of(bid, cm0, cn0);
  set_geometry(cm0, cn0);
  if constexpr (!DIRECT) {
    if (tid < BM) {
      const int m = cm0 + tid;
      int64_t off = -1;
      if (m < p.M) {
        if (DENSE || p.dense) {
          off = (int64_t)m * p.NCOLS;
        } else if constexpr (!DENSE) {
          const int n = fdiv(m, p.d_opq);
          const int rem = m - n * opq;
          const int op = fdiv(rem, p.d_oq);
          const int oq = rem - op * p.OQ;
          off = (int64_t)n * p.y_sn + (int64_t)op * p.y_sh + (int64_t)oq * p.y_sw;
        }
      }
      rowoff[tid] = off;
    }
  }

  // ---- DMA issue.  The A0 and A1 half-tiles of one K-tile go out in different phases: each stream walks
  // the (r, s, c0) taps on its own
  auto issue_a = [&](auto PB, auto H, int& tr, int& ts, int& tc) __attribute__((always_inline)) {
    constexpr int PB_ = decltype(PB)::value, H_ = decltype(H)::value;
    const uint32_t tap = DENSE ? (uint32_t)(tc * 2) : (uint32_t)(tr * p.a_sh2 + ts * p.a_sw2 + tc * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int x = H_ * 2 + i;
      uint32_t off;
      if constexpr (DENSE) {
        off = a_base[x] == kOOB ? kOOB : a_base[x] + tap;
      } else {
        const bool ok = (uint32_t)(ih0[x] + tr) < (uint32_t)p.IH && (uint32_t)(iw0[x] + ts) < (uint32_t)p.IW;
        off = ok ? a_base[x] + tap : kOOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_a, (__attribute__((address_space(3))) void*)(smem + A_REGION + PB_ * BUF + H_ * HALF + (i * 8 + wave) * 1024),
          16, off, 0, 0, 0);
    }
    tc += BK;                     // (DENSE: one tap — every stream issues exactly nk K-tiles between two set_geometry)
    if constexpr (!DENSE) {
      if (tc == p.C) { tc = 0; if (++ts == p.S) { ts = 0; ++tr; } }
    }
  };
  auto issue_b = [&](auto PB, auto H, int t) __attribute__((always_inline)) {
    constexpr int PB_ = decltype(PB)::value, H_ = decltype(H)::value;
    const uint32_t koff = (uint32_t)t * (uint32_t)(BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int x = H_ * 2 + i;
      const uint32_t off = b_off[x] == kOOB ? kOOB : b_off[x] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_b, (__attribute__((address_space(3))) void*)(smem + B_REGION + PB_ * BUF + H_ * HALF + (i * 8 + wave) * 1024),
          16, off, 0, 0, 0);
    }
  };

  // ---- fragment read addresses: row r of a half, k-step ks: slot (ks*4 + l4) ^ ((r >> 1) & 7); the other
  // fragments / halves / buffers are immediate offsets (fragment +16 rows = 2048 B keeps the swizzle term).
  // DIRECT: operand rows 4-7 <-> 8-11 of every weight fragment are exchanged (bits 2 and 3 of the lane's row),
  // so that accumulator register r of lane (l15, l4) is column {0, 8, 4, 12}[l4] + r of the fragment
  uint32_t a_rd[2], b_rd[2];
  {
    const int pl = DIRECT ? ((l15 & 3) | ((l15 & 4) << 1) | ((l15 & 8) >> 1)) : l15;
    const int ra = wr * 64 + l15, rb = wc * 32 + pl;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_rd[ks] = lds0 + (uint32_t)(A_REGION + ra * 128 + (((ks * 4 + l4) ^ ((ra >> 1) & 7)) << 4));
      b_rd[ks] = lds0 + (uint32_t)(B_REGION + rb * 128 + (((ks * 4 + l4) ^ ((rb >> 1) & 7)) << 4));
    }
  }

  f32x4 acc[2][4][4];              // [A half][A fragment][B half * 2 + B fragment]
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  u32x4 af[2][4];                  // [k-step][fragment] of the current A half
  u32x4 bf0[2][2], bf1[2][2];      // B0 / B1: [k-step][fragment]

  auto read_a = [&](auto PB, auto H) __attribute__((always_inline)) {
    constexpr int OFF = decltype(PB)::value * BUF + decltype(H)::value * HALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      af[ks][0] = lds_read_b128<OFF>(a_rd[ks]);
      af[ks][1] = lds_read_b128<OFF + 2048>(a_rd[ks]);
      af[ks][2] = lds_read_b128<OFF + 4096>(a_rd[ks]);
      af[ks][3] = lds_read_b128<OFF + 6144>(a_rd[ks]);
    }
  };
  auto read_b = [&](auto PB, auto H, u32x4 (&bf)[2][2]) {
    constexpr int OFF = decltype(PB)::value * BUF + decltype(H)::value * HALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf[ks][0] = lds_read_b128<OFF>(b_rd[ks]);
      bf[ks][1] = lds_read_b128<OFF + 2048>(b_rd[ks]);
    }
  };
  // one quadrant: 16 MFMAs, operands swapped (first operand := weight fragment) so that
  // acc[..][r] = C[row = .. + l15][col = .. + l4*4 + r] (igemm_epi.h's layout; DIRECT: see b_rd)
  auto mma = [&](auto AH, auto BH, const u32x4 (&bf)[2][2]) {
    constexpr int AH_ = decltype(AH)::value, BH_ = decltype(BH)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[AH_][i][BH_ * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, bf[ks][j]), __builtin_bit_cast(bf16x8_t, af[ks][i]),
              acc[AH_][i][BH_ * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // segment boundaries: nothing may be scheduled across them
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- DIRECT epilogue of the tile at (m0, n0): accumulators -> global memory, no LDS, no barrier.
  // Per (A half h, fragment i, B half jh): this lane's two fragments f = 0 / 1 hold columns
  // jh*128 + wc*32 + f*16 + {0, 8, 4, 12}[l4] + 0..3 of row h*128 + wr*64 + i*16 + l15; after the affine and the
  // bf16 pack, v_permlane32_swap(f0 regs, f1 regs) leaves lanes 0-31 with 8 consecutive columns of f = 0 and
  // lanes 32-63 with 8 consecutive columns of f = 1: ONE 16-byte store per lane.
  auto epilogue_direct = [&](auto HAS_RES, int m0, int n0) __attribute__((always_inline)) {
    constexpr bool kRes = decltype(HAS_RES)::value;
    const bool relu_now = p.relu && !kRes;
    const int pcol = ((l4 & 1) << 3) | ((l4 & 2) << 1);               // {0, 8, 4, 12}[l4]
    const int scol = (lane >> 5) * 16 + (l4 & 1) * 8;                  // store column inside the 32-wide strip
    bf16_t* yb = reinterpret_cast<bf16_t*>(p.y);
    const bf16_t* rb = reinterpret_cast<const bf16_t*>(p.res);
    // element offsets of this lane's 8 rows (-1: row out of range)
    int64_t roff[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + h * 128 + wr * 64 + i * 16 + l15;
        const int mc = m < p.M ? m : p.M - 1;
        int64_t o;
        if (DENSE || p.dense) {
          o = (int64_t)mc * p.NCOLS;
        } else if constexpr (!DENSE) {
          const int n = fdiv(mc, p.d_opq);
          const int rem = mc - n * opq;
          const int op = fdiv(rem, p.d_oq);
          const int oq = rem - op * p.OQ;
          o = (int64_t)n * p.y_sn + (int64_t)op * p.y_sh + (int64_t)oq * p.y_sw;
        }
        roff[h][i] = m < p.M ? o : -1;
      }
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      const int gcol = n0 + jh * 128 + wc * 32 + scol;
      const bool col_ok = gcol < p.NCOLS;
      // the loads of this column half first, back to back and unconditional (clamped addresses; a load behind a
      // per-element branch costs a full vmcnt(0) each): the affine of this lane's own (pre-swap) columns and the
      // residual pieces
      float4 sc[2], sh[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        int gc = n0 + jh * 128 + wc * 32 + f * 16 + pcol;
        gc = gc < p.NCOLS ? gc : 0;                                    // ragged last column tile: any valid address
        sc[f] = make_float4(1.f, 1.f, 1.f, 1.f);
        sh[f] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc[f] = *reinterpret_cast<const float4*>(p.scale + gc);
        if (p.shift) sh[f] = *reinterpret_cast<const float4*>(p.shift + gc);
      }
      uint4 rres[2][4];
      if constexpr (kRes) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t o = (roff[h][i] >= 0 && col_ok) ? roff[h][i] + gcol : 0;
            rres[h][i] = *reinterpret_cast<const uint4*>(rb + o);
          }
      }
      // ONE compiler-visible vmcnt(0) per column half, behind its last load (it also retires the DMA of the next
      // K-tiles, which the main loop needs next anyway, and — second half — the first half's stores).  Only stores
      // follow, so the compiler's waitcnt pass carries no pending load into the next K-tile; it would otherwise
      // drain the DMA pipeline with a vmcnt(0) in front of every register it re-uses there.
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          uint32_t w[2][2];
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            f32x4 a = acc[h][i][jh * 2 + f];
            a[0] = a[0] * sc[f].x + sh[f].x; a[1] = a[1] * sc[f].y + sh[f].y;
            a[2] = a[2] * sc[f].z + sh[f].z; a[3] = a[3] * sc[f].w + sh[f].w;
            if (relu_now) {
#pragma unroll
              for (int r = 0; r < 4; ++r) a[r] = fmaxf(a[r], 0.f);
            }
            w[f][0] = pack2bf(a[0], a[1]);
            w[f][1] = pack2bf(a[2], a[3]);
          }
          const auto x0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
          const auto x1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
          uint4 v = make_uint4(x0[0], x1[0], x0[1], x1[1]);
          if constexpr (kRes) {
            float a8[8], r8[8];
            epi::unpack8(v, a8);
            epi::unpack8(rres[h][i], r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              a8[e] += r8[e];
              if (p.relu) a8[e] = fmaxf(a8[e], 0.f);
            }
            v = epi::pack8(a8);
          }
          if (roff[h][i] >= 0 && col_ok) *reinterpret_cast<uint4*>(yb + roff[h][i] + gcol) = v;
        }
    }
  };

  // ---- the stream of K-tiles.  g = global K-tile index of this workgroup (G in total), t = its index inside
  // the tile being multiplied, kb = index inside the tile being STAGED of the next B0 / A0 / B1 to go out
  const int G = cnt * nk;
  int t = 0, kb = 2, staged = 0;     // staged: how many tiles' geometry has been set up so far, minus one

  auto ktile = [&](auto PB, int g) __attribute__((always_inline)) {
    typedef std::integral_constant<int, 1 - decltype(PB)::value> PO;    // the other buffer
    const bool more1 = g + 1 < G, more2 = g + 2 < G;
    // ---- q0: A0 x B0
    read_b(PB, I0{}, bf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(PB, I0{});
    if (more1) issue_a(PO{}, I1{}, r1, s1, c1);             // A1(g+1)
    wait_lgkm<8>();                                        // the four B0 reads are retired: B0 may be re-staged in q1
    bar();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I0{}, bf0);
    bar();
    // ---- q1: A0 x B1
    read_b(PB, I1{}, bf1);
    if constexpr (DIRECT) {
      // everything staged from here on belongs to the next tile: switch the DMA geometry
      if (more2 && kb == nk) {
        ++staged;
        int nm0, nn0;
        tile_of(bid + staged * nwg, nm0, nn0);
        set_geometry(nm0, nn0);
        kb = 0;
      }
    }
    if (more2) issue_b(PB, I0{}, kb);                       // B0(g+2)
    bar();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(I0{}, I1{}, bf1);
    bar();
    // ---- q2: A1 x B1
    read_a(PB, I1{});
    if (more2) issue_a(PB, I0{}, r0, s0, c0);              // A0(g+2)
    bar();
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    mma(I1{}, I1{}, bf1);
    bar();
    // ---- q3: A1 x B0; K-tile g+1 must have landed (A1(g+1) is the oldest DMA still counted)
    if (more2) {
      issue_b(PB, I1{}, kb);                                // B1(g+2)
      ++kb;
      wait_vm<6>();
    } else {
      wait_vm<0>();
    }
    bar();
    mma(I1{}, I0{}, bf0);
    bar();
    if constexpr (DIRECT) {
      if (++t == nk) {
        if (p.res) epilogue_direct(std::true_type{}, cm0, cn0);
        else epilogue_direct(std::false_type{}, cm0, cn0);
        t = 0;
        if (more1) {
          zero_acc();
          tile_of(bid + (g + 1) / nk * nwg, cm0, cn0);
        }
      }
    }
  };

  // ---- prologue: K-tile 0 complete, then the three half-tiles of K-tile 1 that q3 of a "K-tile -1" would have
  // issued (B0, A0, B1 — the steady-state order), so that the loop starts in its steady state (nk >= 2)
  issue_b(I0{}, I0{}, 0);
  issue_a(I0{}, I0{}, r0, s0, c0);
  issue_b(I0{}, I1{}, 0);
  issue_a(I0{}, I1{}, r1, s1, c1);
  if (G > 1) {
    issue_b(I1{}, I0{}, 1);
    issue_a(I1{}, I0{}, r0, s0, c0);
    issue_b(I1{}, I1{}, 1);
    wait_vm<6>();
  } else {
    wait_vm<0>();
  }
  bar();
  if (wr == 1) bar();               // the second wave group runs one barrier behind the first
  for (int g = 0; g < G; g += 2) {
    ktile(I0{}, g);
    if (g + 1 < G) ktile(I1{}, g + 1);
  }
  if (wr == 0) bar();

  if constexpr (!DIRECT) {
    __syncthreads();                // every fragment read is done: the tile buffers become the output staging area
    // ---- staged epilogue: the two 128-row halves one after the other through igemm_epi.h (statistics slabs are
    // per 128-row tile: half h of tile mt is slab row 2*mt + h)
    const int mt = cm0 / BM;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (cm0 + h * 128 < p.M) {
        epi::epilogue_bf16<128, BN, kThreads, 4, 4, 64, 32, false, 2, 128>(p, smem, rowoff + h * 128, acc[h], wr, wc,
                                                                          lane, tid, cn0, mt * 2 + h);
      }
      __syncthreads();
    }
  }
}

template <bool DIRECT, bool DENSE = false>
static int launch(const Params& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_8p_kernel<DIRECT, DENSE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  // persistent form: one workgroup per CU (a multiple of 8 so that a workgroup's tiles stay on one XCD's range)
  const int grid = DIRECT && p.ntiles > 256 ? 256 : p.ntiles;
  hipLaunchKernelGGL((igemm_8p_kernel<DIRECT, DENSE>), dim3(grid), dim3(kThreads), LDS_BYTES, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

}  // namespace g8

// igemm_8p: 0 = never, 1 = when the cost model below prefers it (default), 2 = whenever the launch is inside
// the kernel's envelope.  The model (fitted to profiles/r03_8p_vs_ring.txt, times in 0.01 us):
//   8-phase: one workgroup per CU, rounds of 256 tiles in lockstep: ceil(tiles / 256) x (nk x tk + te)
//   ring:    two workgroups per CU drifting apart:                  rounds(tiles / 512) x (nk x rtk + rte)
// with nk = 64-element K-tiles, tk / rtk the time of one K-tile and te / rte the exposed prologue + epilogue of
// a tile (the staged 8-phase form's is larger: nothing else runs on the CU while a tile is stored; the persistent
// form — igemm_8p_direct, default on — hides most of it: te_direct).
static int g_8p_mode = -1, g_8p_min_nk = 8, g_8p_tk = 145, g_8p_te = 1000, g_8p_ted = 900, g_8p_rtk = 112,
           g_8p_rte = 420, g_8p_margin = 100, g_8p_direct = 1, g_8p_dense = -1;

int passl_igemm_8p_option(const char* name, int value) {
  if (!strcmp(name, "igemm_8p")) {
    if (value < 0 || value > 2) return PASSL_EINVAL;
    g_8p_mode = value;
    return PASSL_OK;
  }
  if (!strcmp(name, "igemm_8p_direct")) { g_8p_direct = value != 0; return PASSL_OK; }
  if (!strcmp(name, "igemm_8p_dense")) {              // 0 off (default), 1 persistent form, 2 also the staged form
    if (value < 0 || value > 2) return PASSL_EINVAL;
    g_8p_dense = value;
    return PASSL_OK;
  }
  int* slot = !strcmp(name, "igemm_8p_min_nk") ? &g_8p_min_nk : !strcmp(name, "igemm_8p_tk") ? &g_8p_tk :
              !strcmp(name, "igemm_8p_te") ? &g_8p_te : !strcmp(name, "igemm_8p_te_direct") ? &g_8p_ted : !strcmp(name, "igemm_8p_ring_tk") ? &g_8p_rtk :
              !strcmp(name, "igemm_8p_ring_te") ? &g_8p_rte : !strcmp(name, "igemm_8p_margin") ? &g_8p_margin : nullptr;
  if (!slot) return PASSL_EINVAL;
  if (value <= 0) return PASSL_EINVAL;
  *slot = value;
  return PASSL_OK;
}

int passl_igemm_8p_try(const passl_conv_desc* d, hipStream_t st) {
  if (g_8p_mode < 0) {
    const char* e = getenv("PASSL_IGEMM_8P");
    g_8p_mode = e ? atoi(e) : 1;
    if (g_8p_mode < 0 || g_8p_mode > 2) g_8p_mode = 1;
  }
  if (g_8p_mode == 0) return PASSL_EUNSUPPORTED;
  ring::Params p;
  if (!ring::fill_params(d, g8::BM, g8::BN, p)) return PASSL_EUNSUPPORTED;
  const int nk = p.KDIM / g8::BK;
  // the persistent form stores from the accumulators: launches with fused statistics keep the staged epilogue
  const bool direct = g_8p_direct && nk >= 2 && !d->stats && !d->bnb_partial;
  if (g_8p_mode == 1) {
    // short reductions: igemm_kernel's territory (also inside the step: taking the K = 256 launches of stages 3 / 4
    // here — 4 K-tiles, <= 800 tiles — measured 24.4 vs 24.1 ms per MoCo step, profiles/r05_negative_results.txt)
    if (nk < g_8p_min_nk) return PASSL_EUNSUPPORTED;
    const int64_t t8 = p.ntiles;
    const int64_t tr = ((int64_t)(p.M + 127) / 128) * ((d->NCOLS + 127) / 128);
    const double time8 = (double)((t8 + 255) / 256) * ((double)nk * g_8p_tk + (direct ? g_8p_ted : g_8p_te));
    // the ring kernel's workgroups drift apart, so large launches cost tiles / 512 "rounds"; a short launch
    // pays for its last, partly filled round (measured: 1.53 rounds cost 1.8, 3.06 cost 3.7)
    double rr = (double)tr / 512.0;
    if (rr < 4.0) rr = (double)((int64_t)(rr * 2.0 + 0.999)) * 0.5;
    const double timer = (rr < 1.0 ? 1.0 : rr) * ((double)nk * g_8p_rtk + g_8p_rte);
    if (time8 * g_8p_margin >= timer * 100.0) return PASSL_EUNSUPPORTED;
  }
  if (g_8p_dense < 0) {
    const char* e = getenv("PASSL_IGEMM_8P_DENSE");
    g_8p_dense = e ? atoi(e) : 1;
    if (g_8p_dense < 0 || g_8p_dense > 2) g_8p_dense = 1;
  }
  // the matrix-operand specialisation.  1 (default since round 4): every plain-matrix launch of the persistent form —
  // the ViT Linears and the 1x1 stride-1 convolutions without fused statistics; bit-identical to the general form on
  // every shape of scratch/ab_8p_dense.py incl. ragged tiles and 4-13 % faster on 8 of the 9 shapes this kernel takes
  // (the K = 512 -> 2048 Linear is 6 % slower; profiles/r04_8p_dense_ab.txt), GPU suite green with it.  2 (opt-in): also the launches with fused statistics
  // (staged form) — exact as well, but faster on only 2 of 4 R50 shapes (256->1024 @14 is 8 % slower).
  if (g_8p_dense >= 1 && p.dense && direct) return g8::launch<true, true>(p, st);
  if (g_8p_dense >= 2 && p.dense && !direct) return g8::launch<false, true>(p, st);
  return direct ? g8::launch<true>(p, st) : g8::launch<false>(p, st);
}
