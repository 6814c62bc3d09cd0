// Implicit-GEMM convolution, bf16, large tiles + LDS-DMA ring (gfx950).
//
// Same contract as igemm_kernel (conv_igemm.hip): Y[m][col] = epi(sum_k A[m][k] B[col][k]) with
// m = (n,op,oq) gathered from an NHWC tensor, k = (r,s,c), fused affine / residual / ReLU /
// BatchNorm statistics in the epilogue.  What differs is how the operands reach the MFMAs:
//
//  * block tile 256 x BN (BN = 128: 8 waves as 4 x 2; BN = 64: 8 waves as 8 x 1), K-tile 64;
//    one workgroup per CU (512 threads, two waves per SIMD);
//  * staging by `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging VGPRs, no ds_write; the
//    im2col gather is the per-lane buffer offset, padding / ragged rows use an out-of-range
//    offset that the buffer bounds check turns into zeros; the XOR slot swizzle of the tile rows
//    is applied to the SOURCE chunk each lane fetches (LDS-DMA writes lane-linear);
//  * a 3-stage LDS ring (3 x (256 + BN) x 128 B): while tile t is multiplied, tiles t+1 and t+2
//    are in flight.  hipcc orders every LDS read it can see behind ALL pending LDS-DMA
//    (s_waitcnt vmcnt(0)), which would collapse the ring to depth 1, so the fragment reads are
//    inline-asm ds_read_b128 and the waits are explicit: counted `s_waitcnt vmcnt(IPT)` (IPT =
//    DMA instructions per tile per wave: the newest tile stays in flight across the barrier),
//    raw s_barrier, `s_waitcnt lgkmcnt(0)` + sched_barrier before the MFMAs.
//  * ring protocol per K-tile t (stage t % 3):  wait own DMA of tile t -> barrier (all waves'
//    tile-t DMA landed AND all waves finished reading stage (t-1) % 3) -> issue DMA of tile t+2
//    into stage (t-1) % 3 -> read fragments of tile t -> MFMA.
//
// Used when the launch has enough 256-row tiles to fill the chip and the operands are addressable
// with 32-bit byte offsets; passl_hip_conv_igemm falls back to igemm_kernel otherwise.
#include "igemm_dma.h"
#include "igemm_epi.h"

namespace ring {

// BK = K-elements per ring stage.  64: LDS rows of 128 B, 8 chunks, slot ^= (row >> 1) & 7.
// 32: LDS rows of 64 B, 4 chunks, slot ^= g((row >> 2) & 3) with g = {0, 3, 2, 1}: for every one of
// the four 16-lane groups that ds_read_b128 services together ({0-3,12-15,20-27}, {4-11,16-19,28-31}
// and the same in the upper half-wave) the 16 lanes then fall on 16 different 16-byte bank quads —
// the 4 rows that share a 64-byte bank range carry 4 different physical chunks.  BK = 32 with 4
// stages keeps the 64 KB ring (2 workgroups per CU) but has up to 3 half-tiles in flight.
__device__ __forceinline__ int swz32(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <int BM, int BN, int STAGES, int BK = 64, bool BNB2 = false>
__global__ void __launch_bounds__(BM * 2, (STAGES * (BM + BN) * BK * 2 + BM * 8 <= 53 * 1024) ? 3 : 2)
    igemm_ring_kernel(const Params p) {
  constexpr int RB = BK * 2;                // LDS row bytes
  constexpr int KS = BK / 32;               // 16x16x32 MFMA k-steps per stage
  constexpr int RPI = 1024 / RB;            // tile rows per DMA instruction (8 or 16)
  constexpr int CPRW = RB / 16;             // 16-byte chunks per row (8 or 4)
  static_assert(BK == 64 || BK == 32, "ring stage depth");
  constexpr int kThreads = BM * 2;          // 4 waves for 128-row tiles, 8 for 256-row tiles
  constexpr int WAVES = kThreads / 64;
  constexpr int WAVES_N = BN == 128 ? 2 : 1;
  constexpr int WAVES_M = WAVES / WAVES_N;
  constexpr int WM = BM / WAVES_M;          // 64 or 32 rows per wave
  constexpr int WN = BN / WAVES_N;          // 64 columns per wave
  constexpr int FM = WM / 16, FN = WN / 16;
  static_assert(FN == 4 && (FM == 4 || FM == 2), "wave tile is 64 or 32 rows x 64 columns");
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NIA = A_BYTES / 1024 / WAVES;   // A DMA instructions per wave per tile (4)
  constexpr int NIB = B_BYTES / 1024 / WAVES;   // 1, 2 or 4
  constexpr int IPT = NIA + NIB;
  constexpr int LDOB = BN + 8;              // bf16 epilogue pitch (elements)
  constexpr int CPR = BN / 8;
  static_assert(BM * LDOB * 2 <= STAGES * STAGE, "epilogue tile must fit the ring");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + STAGES * STAGE);
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) char*)smem);

  // ---- XCD-aware tile mapping (bijective for any ntiles)
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
  const int opq = p.OP * p.OQ;

  // A may be larger than a buffer descriptor can address (32-bit byte offsets): the descriptor of this
  // workgroup starts at the first image (dense: first row) its tile touches, lane offsets are relative to
  // that; the host guarantees that one tile's span stays below 2 GB
  const int nb = p.dense ? 0 : fdiv(m0 < p.M ? m0 : p.M - 1, p.d_opq);
  const int64_t a_off0 = p.dense ? (int64_t)m0 * (p.C * 2) : (int64_t)nb * p.a_sn2;
  const int64_t a_left = p.a_total - a_off0;
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.a) + a_off0, 0, (uint32_t)(a_left < 0x7ffffff0ll ? a_left : 0x7ffffff0ll), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.b), 0, p.b_bytes, 0x00020000);

  // ---- static DMA geometry.  Instruction q = i*WAVES + wave covers tile rows 8q .. 8q+7;
  // lane -> row 8q + (lane >> 3), LDS slot lane & 7, SOURCE chunk (lane & 7) ^ ((row >> 1) & 7).
  // A byte offset of (row, tap r/s, channel c0) = a_base + r*a_sh2 + s*a_sw2 + 2*c0 with
  // a_base = n*a_sn2 + ih0*a_sh2 + iw0*a_sw2 + chunk (may be "negative" mod 2^32 for padding
  // taps: those lanes are redirected to the out-of-range offset), valid iff
  // 0 <= ih0 + r < IH and 0 <= iw0 + s < IW (unsigned compares; invalid rows carry ih0 = -2^28).
  uint32_t a_base[NIA];
  int ih0[NIA], iw0[NIA];
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int row = (i * WAVES + wave) * RPI + lane / CPRW;
    const int m = m0 + row;
    const uint32_t chunk = (uint32_t)(((lane % CPRW) ^ (BK == 64 ? ((row >> 1) & 7) : swz32(row))) * 16);
    if (m < p.M) {
      if (p.dense) {
        a_base[i] = (uint32_t)(m - m0) * (uint32_t)(p.C * 2) + chunk;
        ih0[i] = 0; iw0[i] = 0;
      } else {
        const int n = fdiv(m, p.d_opq);
        const int rem = m - n * opq;
        const int op = fdiv(rem, p.d_oq);
        const int oq = rem - op * p.OQ;
        ih0[i] = op * p.sh - p.ph;
        iw0[i] = oq * p.sw - p.pw;
        a_base[i] = (uint32_t)(n - nb) * (uint32_t)p.a_sn2 + (uint32_t)(ih0[i] * p.a_sh2) +
                    (uint32_t)(iw0[i] * p.a_sw2) + chunk;
      }
    } else {
      a_base[i] = 0; ih0[i] = -(1 << 28); iw0[i] = 0;
    }
  }
  uint32_t b_off[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int row = (i * WAVES + wave) * RPI + lane / CPRW;
    const int col = n0 + row;
    const uint32_t chunk = (uint32_t)(((lane % CPRW) ^ (BK == 64 ? ((row >> 1) & 7) : swz32(row))) * 16);
    b_off[i] = col < p.NCOLS ? (uint32_t)col * (uint32_t)(p.KDIM * 2) + chunk : kOOB;
  }
  // output row offsets (elements) for the epilogue; -1 = row out of range
  if (tid < BM) {
    const int m = m0 + tid;
    int64_t off = -1;
    if (m < p.M) {
      if (p.dense) {
        off = (int64_t)m * p.NCOLS;
      } else {
        const int n = fdiv(m, p.d_opq);
        const int rem = m - n * opq;
        const int op = fdiv(rem, p.d_oq);
        const int oq = rem - op * p.OQ;
        off = (int64_t)n * p.y_sn + (int64_t)op * p.y_sh + (int64_t)oq * p.y_sw;
      }
    }
    rowoff[tid] = off;
  }

  const int nk = p.KDIM / BK;
  // (r, s, c0) of the NEXT tile to issue, advanced incrementally (wave-uniform scalars)
  int ir = 0, is = 0, ic = 0;
  int issued = 0;
  auto issue_tile = [&]() {
    char* Ab = smem + (issued % STAGES) * STAGE;
    char* Bb = Ab + A_BYTES;
    const uint32_t koff = (uint32_t)issued * (uint32_t)RB;   // byte offset of the K-tile in a B row
    const uint32_t tap = (uint32_t)(ir * p.a_sh2 + is * p.a_sw2 + ic * 2);   // wave-uniform
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const bool ok = (uint32_t)(ih0[i] + ir) < (uint32_t)p.IH && (uint32_t)(iw0[i] + is) < (uint32_t)p.IW;
      const uint32_t off = ok ? a_base[i] + tap : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_a, (__attribute__((address_space(3))) void*)(Ab + (i * WAVES + wave) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const uint32_t off = b_off[i] == kOOB ? kOOB : b_off[i] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_b, (__attribute__((address_space(3))) void*)(Bb + (i * WAVES + wave) * 1024), 16, off, 0, 0, 0);
    }
    ++issued;
    ic += BK;
    if (ic == p.C) { ic = 0; if (++is == p.S) { is = 0; ++ir; } }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (bytes from the stage base): row r, slot (ks*4 + l4) ^ ((r >> 1) & 7);
  // rows of the other fragments differ by multiples of 16 -> same swizzle term, immediate offsets
  uint32_t a_rd[KS], b_rd[KS];
  {
    const int ra = wm * WM + l15, rb = wn * WN + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int sa = BK == 64 ? ((ks * 4 + l4) ^ ((ra >> 1) & 7)) : (l4 ^ swz32(ra));
      const int sbb = BK == 64 ? ((ks * 4 + l4) ^ ((rb >> 1) & 7)) : (l4 ^ swz32(rb));
      a_rd[ks] = (uint32_t)(ra * RB + (sa << 4));
      b_rd[ks] = (uint32_t)(A_BYTES + rb * RB + (sbb << 4));
    }
  }

  // two register sets of fragments: tile t is multiplied from one set while tile t+1 is read
  // from LDS into the other (the reads overlap the MFMAs; waited at the end of the iteration)
  u32x4 af[2][KS][FM], bfr[2][KS][FN];
  auto read_frags = [&](auto SET, int t) {
    constexpr int S_ = decltype(SET)::value;
    const uint32_t sb = lds0 + (uint32_t)((t % STAGES) * STAGE);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {           // fragment f sits 16 rows = 16 * RB bytes further
      af[S_][ks][0] = lds_read_b128<0>(sb + a_rd[ks]);
      af[S_][ks][1] = lds_read_b128<16 * RB>(sb + a_rd[ks]);
      if constexpr (FM == 4) {
        af[S_][ks][2] = lds_read_b128<32 * RB>(sb + a_rd[ks]);
        af[S_][ks][3] = lds_read_b128<48 * RB>(sb + a_rd[ks]);
      }
      bfr[S_][ks][0] = lds_read_b128<0>(sb + b_rd[ks]);
      bfr[S_][ks][1] = lds_read_b128<16 * RB>(sb + b_rd[ks]);
      bfr[S_][ks][2] = lds_read_b128<32 * RB>(sb + b_rd[ks]);
      bfr[S_][ks][3] = lds_read_b128<48 * RB>(sb + b_rd[ks]);
    }
  };
  // Ring protocol.  Before iteration t: tiles <= t+STAGES-1 are issued, tile t sits in
  // registers (set t & 1).  Iteration t: wait own DMA of tile t+1 (the newest tile may stay in
  // flight) -> barrier (everybody's tile t+1 landed; everybody finished reading stage t % STAGES
  // in iteration t-1) -> issue tile t+STAGES into stage t % STAGES -> start reading tile t+1 into
  // the other register set -> MFMAs of tile t -> lgkmcnt(0).
  auto iteration = [&](auto SET, int t) {
    constexpr int S_ = decltype(SET)::value;
    if (t + 1 < nk) {
      // tiles issued after t+1 so far: min(STAGES - 2, nk - 2 - t)
      if (STAGES >= 4 && t + 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
      else if (STAGES >= 3 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + STAGES < nk) issue_tile();
      read_frags(std::integral_constant<int, 1 - S_>{}, t + 1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // operands swapped (A := weight fragment): acc[i][j][r] = C[row][col = .. + l4*4 + r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, bfr[S_][ks][j]), __builtin_bit_cast(bf16x8_t, af[S_][ks][i]),
              acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  static_assert(STAGES >= 2 && STAGES <= 4, "vmcnt bookkeeping covers 2 to 4 stages");
  for (int t = 0; t < STAGES && t < nk; ++t) issue_tile();
  // tile 0 landed: at most min(STAGES, nk) - 1 newer tiles may stay in flight
  if (nk >= STAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * IPT) : "memory");
  else if (nk == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(std::integral_constant<int, 0>{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    iteration(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) iteration(std::integral_constant<int, 1>{}, kt + 1);
  }
  __syncthreads();      // all fragment reads done before the ring is reused as the output tile

  // ---- epilogue (shared with igemm_kernel's bf16 path): igemm_epi.h
  if constexpr (BM == 128) {
    epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN, false, FN, 0, BNB2>(p, smem, rowoff, acc, wm, wn, lane, tid, n0, mt);
  } else {
    // 256-row tiles (experimental igemm_ring_bm = 256): plain epilogue, no fused statistics
    // (passl_igemm_ring_try refuses stats / bnb launches for this tile shape)
    char* outc = smem;
    const bool relu_now = p.relu && !p.res;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = wn * WN + j * 16 + l4 * 4;
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
        const int row = wm * WM + i * 16 + l15;
        *reinterpret_cast<uint2*>(outc + row * (LDOB * 2) + col * 2) =
            make_uint2(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]));
      }
    }
    __syncthreads();
    const bf16_t* outb = reinterpret_cast<const bf16_t*>(smem);
    constexpr int NT = BM * CPR / kThreads;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int chunk = tid + t * kThreads;
      const int row = chunk / CPR, cc = chunk - row * CPR;
      const int gcol = n0 + cc * 8;
      const int64_t roff = rowoff[row];
      if (roff < 0 || gcol >= p.NCOLS) continue;
      const int64_t o = roff + gcol;
      uint4 v = *reinterpret_cast<const uint4*>(outb + row * LDOB + cc * 8);
      if (p.res) {
        float a[8], rr[8];
        epi::unpack8(v, a);
        epi::unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.res) + o), rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a[e] += rr[e];
          if (p.relu) a[e] = fmaxf(a[e], 0.f);
        }
        v = epi::pack8(a);
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.y) + o) = v;
    }
  }
}

template <int BM, int BN, int STAGES, int BK = 64, bool BNB2 = false>
int launch(const Params& p, hipStream_t st) {
  constexpr int LDS = STAGES * (BM + BN) * BK * 2 + BM * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_ring_kernel<BM, BN, STAGES, BK, BNB2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_ring_kernel<BM, BN, STAGES, BK, BNB2>), dim3(p.ntiles), dim3(BM * 2), LDS, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

}  // namespace ring

// Returns PASSL_EUNSUPPORTED when the descriptor is outside this kernel's envelope (the caller
// then uses igemm_kernel); the descriptor has already been validated by passl_hip_conv_igemm.
static int g_ring_enabled = -1, g_ring_min_tiles = 1, g_ring_bm = 128, g_ring_min_nk = 8, g_ring_bk = 64,
           g_ring_stages32 = 4;

// passl_hip_set_option("igemm_ring", 0/1) / ("igemm_ring_min_tiles", n)   (runtime.hip dispatches)
int passl_igemm_ring_option(const char* name, int value) {
  if (!strcmp(name, "igemm_ring")) { g_ring_enabled = value != 0; return PASSL_OK; }
  if (!strcmp(name, "igemm_ring_min_tiles")) { g_ring_min_tiles = value; return PASSL_OK; }
  if (!strcmp(name, "igemm_ring_min_nk")) { g_ring_min_nk = value; return PASSL_OK; }
  if (!strcmp(name, "igemm_ring_stages32")) {       // ring depth of the BK = 32 variant: 3 (3 WG/CU) or 4
    if (value != 3 && value != 4) return PASSL_EINVAL;
    g_ring_stages32 = value;
    return PASSL_OK;
  }
  if (!strcmp(name, "igemm_ring_bk")) {
    if (value != 64 && value != 32) return PASSL_EINVAL;
    g_ring_bk = value;
    return PASSL_OK;
  }
  if (!strcmp(name, "igemm_ring_bm")) {
    if (value != 128 && value != 256) return PASSL_EINVAL;
    g_ring_bm = value;
    return PASSL_OK;
  }
  return PASSL_EINVAL;
}

int passl_igemm_ring_try(const passl_conv_desc* d, hipStream_t st) {
  if (g_ring_enabled < 0) {
    const char* e = getenv("PASSL_IGEMM_RING");
    g_ring_enabled = e ? atoi(e) : 1;
    const char* t = getenv("PASSL_IGEMM_RING_MIN_TILES");
    if (t) g_ring_min_tiles = atoi(t);
  }
  const int enabled = g_ring_enabled, min_tiles = g_ring_min_tiles;
  if (!enabled) return PASSL_EUNSUPPORTED;
  // short reductions are dominated by the prologue/epilogue: igemm_kernel's single-stage variant
  // (3 workgroups per CU) wins there (measured per layer: profiles/r01_ring_vs_igemm_bs256.txt)
  if ((int64_t)d->R * d->S * d->C < 64ll * g_ring_min_nk) return PASSL_EUNSUPPORTED;
  const int bn = d->NCOLS <= 64 ? 64 : 128;
  const int bm = g_ring_bm;
  if (bm != 128 && (d->stats || d->bnb_partial)) return PASSL_EUNSUPPORTED;   // slabs are per 128-row tile
  ring::Params p;
  if (!ring::fill_params(d, bm, bn, p)) return PASSL_EUNSUPPORTED;
  if (p.ntiles < min_tiles) return PASSL_EUNSUPPORTED;
  if (bm == 256) return bn == 64 ? ring::launch<256, 64, 3>(p, st) : ring::launch<256, 128, 3>(p, st);
  if (g_ring_bk == 32 && g_ring_stages32 == 3)
    return bn == 64 ? ring::launch<128, 64, 3, 32>(p, st) : ring::launch<128, 128, 3, 32>(p, st);
  if (g_ring_bk == 32) return bn == 64 ? ring::launch<128, 64, 4, 32>(p, st) : ring::launch<128, 128, 4, 32>(p, st);
  return bn == 64 ? ring::launch<128, 64, 2>(p, st) : ring::launch<128, 128, 2>(p, st);
}

// The two-BatchNorm instantiation (passl_conv_desc.bnb2_*; the caller has checked: dense 1x1, bf16, >= 8 K-tiles, NCOLS > 64)
int passl_igemm_ring_bnb2(const passl_conv_desc* d, hipStream_t st) {
  ring::Params p;
  if (!ring::fill_params(d, 128, 128, p)) return PASSL_EUNSUPPORTED;
  return ring::launch<128, 128, 2, 64, true>(p, st);
}
