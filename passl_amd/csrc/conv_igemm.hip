// Implicit-GEMM convolution for gfx950 (MFMA), NHWC activations x [K][R][S][C] weights.
//
//   Y[m][col] = epi( sum_k A[m][k] * B[col][k] ),  m = (n,op,oq), k = (r,s,c) with c fastest
//
// One kernel serves forward convs, data-gradient convs (conv over dy with repacked weights,
// optionally writing a strided sub-lattice of dx) and Linear layers.
//
// Tiling (per 256-thread workgroup = 4 waves, 2x2):
//   block tile  BM x BN  (128x128 or 128x64), K-tile = 128 BYTES of k per row
//               (64 bf16 / 32 fp32 elements) so the staging code is dtype independent
//   wave tile   64 x BN/2  =  4 x (BN/32) MFMA 16x16 accumulator fragments
//   bf16: v_mfma_f32_16x16x32_bf16 (2 k-steps / tile);  fp32: v_mfma_f32_16x16x4_f32 (8 k-steps)
// LDS: double-buffered A and B tiles, rows of 8 x 16-byte slots, slot index XOR-swizzled with
//   (row>>1)&7 so that the 16-lane groups of ds_read_b128 hit 16 distinct slots of the 256-byte
//   bank row (conflict-free), global->register->LDS staging with the next tile's loads issued
//   before the current tile's MFMAs (one barrier per K-tile).
// Epilogue: accumulators -> LDS -> coalesced 16/32-byte row stores with the fused per-channel
//   affine (BatchNorm in inference form), residual add and ReLU.  bf16 outputs are staged as
//   packed bf16 pairs (34 KB instead of 66 KB of LDS); fp32 outputs / the fp32 path stage fp32.
// Pipelining: STAGES=2 double-buffers the LDS tiles (one barrier per K-tile, 64 KB of LDS, 2
//   workgroups per CU); STAGES=1 keeps ONE LDS tile and prefetches the next K-tile into registers
//   while the MFMAs run (two barriers per K-tile, 35 KB of LDS, 3 workgroups per CU) — the better
//   trade for the short-K, store-heavy 1x1 convolutions that are HBM/latency- rather than
//   MFMA-bound.
// Grid: one workgroup per output tile, N-tiles fastest, remapped so that every XCD (private L2)
//   walks a contiguous range of tiles: the N-tiles that share an A row-panel hit the same L2.
// PERSIST (round 6; dense 1x1 launches, bf16): a workgroup walks tiles bid, bid + grid, ... of its XCD's range.  A
//   wave cannot retire before its stores are acknowledged (s_endpgm waits for vmcnt = 0), and in a write-saturated
//   launch an acknowledgement takes ~5 us: with one tile per workgroup a CU never has more than (workgroups per CU) x
//   32 KB of stores in flight and its slots sit idle for most of a tile's life (64->256 @56: stores alone 89 us =
//   4.6 TB/s against a plain fill's 6.8, profiles/r04_kbench_halo_experiment.txt; 3 -> 4 workgroups per CU bought 14 %).
//   The persistent form never waits for a store: the first K-tile of the NEXT tile is requested (into the staging
//   registers) before the current tile's epilogue, whose stores then drain under the next tile's loads and MFMAs.
//   Same tiles, same arithmetic in the same order, same epilogue: bit-identical output (kbench check igemm_persist=1).
//   MEASURED (profiles/r06_kbench_persist.txt): exact, and 10-26 % SLOWER than one tile per workgroup — the staging registers
//   that stay live across the epilogue cost the kernel half its workgroups per CU (229 VGPRs: 2 instead of 3-4), and the
//   store phase is bound by how many waves issue stores, not by their acknowledgements.  OPT-IN (igemm_persist=1).
#include <stdlib.h>
#include "common.h"
#include "igemm_epi.h"
#include "prof.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRowBytes = 128;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// exact n / d for all 32-bit n by multiply-high (Granlund-Montgomery); d is launch-invariant
struct FastDiv {
  uint32_t mul, sh1, sh2;
};
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
  float* stats;      // fused BatchNorm statistics slab (igemm_epi.h): [stats_tiles][NCOLS][2] + shifts
  int stats_tiles;   // ceil(M / 128)
  // BatchNorm-backward statistics fused into a data-gradient launch (igemm_epi.h)
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
  int M, NCOLS, KDIM;
  int OP, OQ, R, S, C, IH, IW, sh, sw, ph, pw;
  int64_t a_sn, a_sh, a_sw;
  int64_t y_sn, y_sh, y_sw;
  int relu, out_f32;
  int tiles_n, ntiles;
  FastDiv d_opq, d_oq, d_tn;
  int dbg;   // ablation switches for tuning runs (PASSL_IGEMM_DBG): 1 no stores, 2 no epilogue, 4 no A loads, 8 no MFMA
};

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// DENSE: 1x1 / stride 1 / no padding with dense A and Y: row m lives at m*C resp. m*NCOLS, no
// (n,op,oq) decomposition at all.
template <typename T, int BM, int BN, bool GENERIC, int STAGES, bool EPI32, bool DENSE, bool LEAN = false, bool BNB2 = false,
          bool PERSIST = false>
__global__ void __launch_bounds__(kThreads, PERSIST ? 2 : (LEAN ? 4 : ((STAGES == 1 && !EPI32) ? 3 : 2)))
    igemm_kernel(const Params p) {
  static_assert(!PERSIST || (DENSE && STAGES == 1 && !EPI32 && !GENERIC), "persistent form: dense bf16 launches");
  constexpr int ES = sizeof(T);
  constexpr int VEC = 16 / ES;          // elements per 16-byte slot
  constexpr int BK = kRowBytes / ES;    // elements per K-tile
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int ACH = BM * 8 / kThreads;  // A chunks per thread
  constexpr int BCH = BN * 8 / kThreads;  // B chunks per thread
  constexpr int A_BYTES = BM * kRowBytes, B_BYTES = BN * kRowBytes;
  constexpr int LDO = BN + 4;    // fp32 epilogue pitch (floats)
  constexpr int LDOB = BN + 8;   // bf16 epilogue pitch (elements): 272-byte rows, 16-byte aligned

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // layout: [A0][A1][B0][B1] ; epilogue reuses the front as float out[BM][LDO]; row offsets at the end
  char* As = smem;
  char* Bs = smem + STAGES * A_BYTES;
  constexpr int STAGE_BYTES = STAGES * (A_BYTES + B_BYTES);
  constexpr int EPI_BYTES = EPI32 ? BM * LDO * 4 : BM * LDOB * 2;
  constexpr int MAIN_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  int64_t* rowoff = reinterpret_cast<int64_t*>(smem + MAIN_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int slot = tid & 7;
  const int opq = p.OP * p.OQ;

  // ---- XCD-aware tile mapping (bijective for any ntiles): virtual id v -> tile; a persistent workgroup takes
  // v = bid, bid + grid, ... (the grid is a multiple of 8, so all of them lie in its XCD's range)
  auto tile_of = [&](int v) __attribute__((always_inline)) {
    const int xcd = v & 7, local = v >> 3;
    const int q = p.ntiles >> 3, r = p.ntiles & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
  };
  int vt = blockIdx.x;
  int tile = tile_of(vt);
  int mt = fdiv(tile, p.d_tn), nt = tile - mt * p.tiles_n;
  int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread A rows of the tile being LOADED (PERSIST: re-derived for the next tile before its first loads)
  int64_t a_base[ACH];
  int ih0[ACH], iw0[ACH];
  int ld_n0 = n0;                          // column origin of the B tile being loaded
  auto set_rows = [&](int m0_) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int row = (tid >> 3) + 32 * i;
      const int m = m0_ + row;
      if (DENSE) {
        a_base[i] = (int64_t)m * p.C;
        ih0[i] = m < p.M ? 0 : -(1 << 28);
        iw0[i] = 0;
      } else if (m < p.M) {
        const int n = fdiv(m, p.d_opq);
        const int rem = m - n * opq;
        const int op = fdiv(rem, p.d_oq);
        const int oq = rem - op * p.OQ;
        a_base[i] = (int64_t)n * p.a_sn;
        ih0[i] = op * p.sh - p.ph;
        iw0[i] = oq * p.sw - p.pw;
      } else {
        a_base[i] = 0;
        ih0[i] = -(1 << 28);
        iw0[i] = 0;
      }
    }
  };
  set_rows(m0);
  // output row offsets (elements) for the epilogue; -1 = row out of range
  auto set_rowoff = [&](int m0_) __attribute__((always_inline)) {
    if (tid < BM) {
      const int m = m0_ + tid;
      int64_t off = -1;
      if (m < p.M) {
        if (DENSE) {
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
  };
  set_rowoff(m0);

  uint4 ra[ACH], rb[BCH];
  const int nk = LEAN ? 1 : (p.KDIM + BK - 1) / BK;     // LEAN: single K-tile launches only

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    int r, s, c;
    bool kvalid = true;
    if (GENERIC) {
      const int kc = k0 + slot * VEC;
      kvalid = kc < p.KDIM;
      const int rs = kc / p.C;
      c = kc - rs * p.C;
      r = rs / p.S;
      s = rs - r * p.S;
    } else {
      const int rs = k0 / p.C;           // wave-uniform
      c = k0 - rs * p.C + slot * VEC;
      r = rs / p.S;
      s = rs - r * p.S;
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int ih = ih0[i] + r, iw = iw0[i] + s;
      const bool ok = DENSE ? (ih0[i] >= 0)
                            : (kvalid && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok && !(p.dbg & 4)) {
        const int64_t off = DENSE ? a_base[i] + c
                                  : a_base[i] + (int64_t)ih * p.a_sh + (int64_t)iw * p.a_sw + c;
        v = *reinterpret_cast<const uint4*>(p.a + off * ES);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int col = ld_n0 + (tid >> 3) + 32 * j;
      const int kc = k0 + slot * VEC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (col < p.NCOLS && kc < p.KDIM)
        v = *reinterpret_cast<const uint4*>(p.b + ((int64_t)col * p.KDIM + kc) * ES);
      rb[j] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int row = (tid >> 3) + 32 * i;
      *reinterpret_cast<uint4*>(As + buf * A_BYTES + row * kRowBytes + (swz(row, slot) << 4)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int row = (tid >> 3) + 32 * j;
      *reinterpret_cast<uint4*>(Bs + buf * B_BYTES + row * kRowBytes + (swz(row, slot) << 4)) = rb[j];
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, l4 = lane >> 4;
  auto compute_tile = [&](const char* Ab, const char* Bb) {
    if constexpr (ES == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WM + i * 16 + l15;
          const uint4 v = *reinterpret_cast<const uint4*>(Ab + row * kRowBytes + (swz(row, ks * 4 + l4) << 4));
          af[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * WN + j * 16 + l15;
          const uint4 v = *reinterpret_cast<const uint4*>(Bb + row * kRowBytes + (swz(row, ks * 4 + l4) << 4));
          bfr[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            // operands swapped (A := weight fragment, B := activation fragment) so that a lane
            // ends up with 4 CONSECUTIVE output channels of one pixel:
            //   acc[i][j][r] = C[row = wm*WM + i*16 + l15][col = wn*WN + j*16 + l4*4 + r]
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WM + i * 16 + l15;
          af[i] = *reinterpret_cast<const float*>(Ab + row * kRowBytes + (swz(row, ks) << 4) + (l4 << 2));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * WN + j * 16 + l15;
          bfr[j] = *reinterpret_cast<const float*>(Bb + row * kRowBytes + (swz(row, ks) << 4) + (l4 << 2));
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
  };

  load_tile(0);
  if constexpr (PERSIST) {
    const int G = gridDim.x;
    for (;;) {
      const int nv = vt + G;
      const bool has_next = nv < p.ntiles;                 // workgroup-uniform
      int nmt = 0, nnt = 0;
      for (int kt = 0; kt < nk; ++kt) {
        store_tile(0);
        __syncthreads();
        if (kt + 1 < nk) {
          load_tile(kt + 1);                               // in flight while the MFMAs run
        } else if (has_next) {
          // the NEXT tile's first K-tile: requested now, consumed after this tile's epilogue — whose stores are
          // issued behind these loads and never waited for
          const int ntile = tile_of(nv);
          nmt = fdiv(ntile, p.d_tn);
          nnt = ntile - nmt * p.tiles_n;
          set_rows(nmt * BM);
          ld_n0 = nnt * BN;
          load_tile(0);
        }
        if (!(p.dbg & 8)) compute_tile(As, Bs);
        __syncthreads();
      }
      if (!(p.dbg & 2))
        epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN, LEAN, FN, 0, BNB2>(p, smem, rowoff, acc, wm, wn, lane, tid, n0, mt);
      if (!has_next) return;
      __syncthreads();           // the staging area and the row table are rewritten for the next tile
      vt = nv; mt = nmt; nt = nnt; m0 = mt * BM; n0 = nt * BN;
      set_rowoff(m0);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else if constexpr (STAGES == 2) {
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_tile(kt + 1);
      compute_tile(As + buf * A_BYTES, Bs + buf * B_BYTES);
      if (kt + 1 < nk) store_tile(buf ^ 1);
      __syncthreads();
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      store_tile(0);
      __syncthreads();
      if (kt + 1 < nk) load_tile(kt + 1);      // in flight while the MFMAs run
      if (!(p.dbg & 8)) compute_tile(As, Bs);
      __syncthreads();
    }
  }

  constexpr int CPR = BN / 8;  // 8-column chunks per row
  if (p.dbg & 2) return;
  if constexpr (EPI32) {
    // ---- epilogue phase 1: accumulators -> LDS fp32 tile
    float* out = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wm * WM + i * 16 + l15;
        const int col = wn * WN + j * 16 + l4 * 4;
        *reinterpret_cast<f32x4*>(out + row * LDO + col) = acc[i][j];
      }
    __syncthreads();
    // ---- phase 2: coalesced row stores, 8 columns per thread-chunk
#pragma unroll
    for (int t = 0; t < BM * CPR / kThreads; ++t) {
      const int chunk = tid + t * kThreads;
      const int row = chunk / CPR, cc = chunk - row * CPR;
      const int gcol = n0 + cc * 8;
      const int64_t roff = rowoff[row];
      if (roff < 0 || gcol >= p.NCOLS) continue;
      float v[8];
      const float4 v0 = *reinterpret_cast<const float4*>(out + row * LDO + cc * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(out + row * LDO + cc * 8 + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
      v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      if (p.scale) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.scale[gcol + e];
      }
      if (p.shift) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.shift[gcol + e];
      }
      const int64_t o = roff + gcol;
      if (p.out_f32) {
        if (p.res) {
          float rr[8];
          ElemTraits<float>::load8(reinterpret_cast<const float*>(p.res) + o, rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rr[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        ElemTraits<float>::store8(reinterpret_cast<float*>(p.y) + o, v);
      } else {
        if (p.res) {
          float rr[8];
          ElemTraits<T>::load8(reinterpret_cast<const T*>(p.res) + o, rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rr[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        ElemTraits<T>::store8(reinterpret_cast<T*>(p.y) + o, v);
      }
    }
  } else {
    // ---- bf16 epilogue (shared with the ring kernel): igemm_epi.h
    epi::epilogue_bf16<BM, BN, kThreads, FM, FN, WM, WN, LEAN, FN, 0, BNB2>(p, smem, rowoff, acc, wm, wn, lane, tid, n0, mt);
  }
}

static int g_persist = -1, g_persist_grid = 0;

template <typename T, int BM, int BN, bool GENERIC, int STAGES, bool EPI32, bool DENSE, bool LEAN = false, bool BNB2 = false,
          bool PERSIST = false>
int launch(const Params& p, hipStream_t st) {
  constexpr int STAGE = STAGES * (BM + BN) * kRowBytes;
  constexpr int EPI = EPI32 ? BM * (BN + 4) * 4 : BM * (BN + 8) * 2;
  constexpr int LDS = (STAGE > EPI ? STAGE : EPI) + BM * 8;
  const void* fn = reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, GENERIC, STAGES, EPI32, DENSE, LEAN, BNB2, PERSIST>);
  static bool attr_set = false;
  static int wg_slots = 0;               // PERSIST: resident workgroups of this instantiation on the whole device
  if (!attr_set) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (PERSIST) {
      int occ = 0, dev = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kThreads, LDS) != hipSuccess || occ < 1) occ = 1;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
        cus = 256;
      wg_slots = (occ * cus) & ~7;        // a multiple of 8: a workgroup's tiles stay on its XCD's range
    }
    attr_set = true;
  }
  const int slots = (PERSIST && g_persist_grid >= 8) ? g_persist_grid : wg_slots;
  const int grid = (PERSIST && p.ntiles > slots) ? slots : p.ntiles;
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, GENERIC, STAGES, EPI32, DENSE, LEAN, BNB2, PERSIST>), dim3(grid),
                     dim3(kThreads), LDS, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

// igemm_persist: 0 = one tile per workgroup (rounds 1-5), 1 = the persistent form for dense bf16 launches whose tiles
// outnumber the resident workgroups (passl_hip_set_option("igemm_persist", v) / PASSL_IGEMM_PERSIST)
}  // namespace
int passl_igemm_persist_option(int value) { g_persist = value != 0; return PASSL_OK; }
// igemm_persist_grid: 0 = as many workgroups as the device holds (default); n = that many (tests: small launches walk several tiles)
int passl_igemm_persist_grid_option(int value) { g_persist_grid = value < 0 ? 0 : (value & ~7); return PASSL_OK; }
namespace {
static bool persist_on() {
  if (g_persist < 0) {
    const char* e = getenv("PASSL_IGEMM_PERSIST");
    g_persist = e ? (atoi(e) != 0) : 0;      // measured SLOWER (profiles/r06_negative_results.txt #1): opt-in
  }
  return g_persist != 0;
}

// K-tiles up to which the single-LDS-stage variant is used (tunable: PASSL_IGEMM_NK1)
int nk1_threshold() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PASSL_IGEMM_NK1");
    v = e ? atoi(e) : 24;
  }
  return v;
}

template <typename T, int BN>
int dispatch(const Params& p, bool generic, bool out_f32, bool dense, int nk, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    return generic ? launch<T, 128, BN, true, 2, true, false>(p, st)
                   : launch<T, 128, BN, false, 2, true, false>(p, st);
  } else {
    if (out_f32)
      return generic ? PASSL_EUNSUPPORTED : launch<T, 128, BN, false, 2, true, false>(p, st);
    if (generic) return launch<T, 128, BN, true, 2, false, false>(p, st);
    const bool one = nk <= nk1_threshold();
    // LEAN: single-K-tile dense launches without residual / BatchNorm-backward work in the epilogue (the
    // K = 64 forward 1x1 layers of stage 1) fit 116 VGPRs -> 4 workgroups per CU instead of 3: these
    // launches are bound by how many operand waits / epilogues a CU has in flight;
    // 64->256 @56: 117.6 -> 101.1 us, with statistics 152.9 -> 127.0 us.  PASSL_IGEMM_LEAN=0 disables.
    static const bool lean_on = !(getenv("PASSL_IGEMM_LEAN") && atoi(getenv("PASSL_IGEMM_LEAN")) == 0);
    if (dense && lean_on && nk == 1 && !p.res && !p.bnb_partial)
      return persist_on() ? launch<T, 128, BN, false, 1, false, true, true, false, true>(p, st)
                          : launch<T, 128, BN, false, 1, false, true, true>(p, st);
    if (dense && one && persist_on()) return launch<T, 128, BN, false, 1, false, true, false, false, true>(p, st);
    if (dense)
      return one ? launch<T, 128, BN, false, 1, false, true>(p, st)
                 : launch<T, 128, BN, false, 2, false, true>(p, st);
    return one ? launch<T, 128, BN, false, 1, false, false>(p, st)
               : launch<T, 128, BN, false, 2, false, false>(p, st);
  }
}

}  // namespace

int passl_igemm_ring_try(const passl_conv_desc* d, hipStream_t st);   // conv_igemm_ring.hip
int passl_igemm_ring_bnb2(const passl_conv_desc* d, hipStream_t st);  // ... its two-BatchNorm instantiation
int passl_igemm_8p_try(const passl_conv_desc* d, hipStream_t st);     // conv_igemm_8p.hip
int passl_stem_try(const passl_conv_desc* d, hipStream_t st);         // conv_stem.hip
int passl_conv3x3_wave_try(const passl_conv_desc* d, hipStream_t st); // conv3x3_wave.hip

static int g_last_kernel = -1;
extern "C" int passl_hip_last_igemm_kernel(void) { return g_last_kernel; }

extern "C" int passl_hip_conv_igemm(const passl_conv_desc* d, passl_stream_t stream) {
  if (!d || !d->a || !d->b || !d->y) return PASSL_EINVAL;
  // fused BN statistics (forward or backward) live in the bf16-output epilogue only
  if (d->stats && (d->dtype != PASSL_BF16 || d->out_f32 || d->residual || d->bnb_partial ||
                   !aligned16(d->stats)))
    return PASSL_EUNSUPPORTED;
  if (d->bnb_partial &&
      (d->dtype != PASSL_BF16 || d->out_f32 || d->relu || !d->bnb_y || !d->bnb_mean || !d->bnb_invstd ||
       !aligned16(d->bnb_y) || d->bnb_tile_off < 0 || (d->bnb_relu != 0 && d->bnb_relu != 2 && d->bnb_relu != 3) ||
       (d->bnb_relu == 2 && (!d->bnb_scale || !d->bnb_shift)) || (d->bnb_relu == 3 && !d->bnb_mask)))
    return PASSL_EINVAL;
  if (d->N <= 0 || d->OP <= 0 || d->OQ <= 0 || d->NCOLS <= 0 || d->R <= 0 || d->S <= 0 ||
      d->C <= 0 || d->IH <= 0 || d->IW <= 0)
    return PASSL_EINVAL;
  if (d->dtype != PASSL_F32 && d->dtype != PASSL_BF16) return PASSL_EUNSUPPORTED;
  const int es = d->dtype == PASSL_BF16 ? 2 : 4;
  const int vec = 16 / es, bk = 128 / es;
  if ((d->C % vec) || (d->NCOLS & 7)) return PASSL_EINVAL;
  if ((d->a_sn % vec) || (d->a_sh % vec) || (d->a_sw % vec)) return PASSL_EINVAL;
  if ((d->y_sn & 7) || (d->y_sh & 7) || (d->y_sw & 7)) return PASSL_EINVAL;
  if (!aligned16(d->a) || !aligned16(d->b) || !aligned16(d->y) ||
      (d->residual && !aligned16(d->residual)) || (d->scale && !aligned16(d->scale)) ||
      (d->shift && !aligned16(d->shift)))
    return PASSL_EINVAL;
  const int64_t M64 = (int64_t)d->N * d->OP * d->OQ;
  const int64_t K64 = (int64_t)d->R * d->S * d->C;
  if (M64 > 0x7fffffff || K64 > 0x7fffffff) return PASSL_EINVAL;

  Params p;
  p.a = reinterpret_cast<const char*>(d->a);
  p.b = reinterpret_cast<const char*>(d->b);
  p.y = reinterpret_cast<char*>(d->y);
  p.scale = d->scale; p.shift = d->shift;
  p.res = reinterpret_cast<const char*>(d->residual);
  p.stats = d->stats;
  p.bnb_y = reinterpret_cast<const char*>(d->bnb_y); p.bnb_mask = d->bnb_mask;
  p.bnb_mean = d->bnb_mean; p.bnb_invstd = d->bnb_invstd;
  p.bnb_scale = d->bnb_scale; p.bnb_shift = d->bnb_shift;
  p.bnb_partial = d->bnb_partial; p.bnb_relu = d->bnb_relu; p.bnb_tile_off = d->bnb_tile_off;
  p.bnb2_y = reinterpret_cast<const char*>(d->bnb2_y); p.bnb2_mean = d->bnb2_mean; p.bnb2_invstd = d->bnb2_invstd;
  p.bnb2_partial = d->bnb2_partial;
  p.M = (int)M64; p.NCOLS = d->NCOLS; p.KDIM = (int)K64;
  p.OP = d->OP; p.OQ = d->OQ; p.R = d->R; p.S = d->S; p.C = d->C;
  p.IH = d->IH; p.IW = d->IW; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw;
  p.a_sn = d->a_sn; p.a_sh = d->a_sh; p.a_sw = d->a_sw;
  p.y_sn = d->y_sn; p.y_sh = d->y_sh; p.y_sw = d->y_sw;
  p.relu = d->relu; p.out_f32 = d->out_f32;
  {
    static int dbg = -1;
    static int dyn = -1;
    if (dyn < 0) dyn = getenv("PASSL_IGEMM_DBG_DYNAMIC") ? 1 : 0;
    if (dbg < 0 || dyn) { const char* e = getenv("PASSL_IGEMM_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  const bool generic = (d->C % bk) != 0;
  const bool narrow = d->NCOLS <= 64;
  const int bn = narrow ? 64 : 128;
  p.tiles_n = (d->NCOLS + bn - 1) / bn;
  const int tiles_m = (p.M + 127) / 128;
  p.stats_tiles = tiles_m;
  if (d->stats && d->stats_tiles != tiles_m) return PASSL_EINVAL;
  p.ntiles = tiles_m * p.tiles_n;
  p.d_opq = make_fastdiv((uint32_t)(d->OP * d->OQ));
  p.d_oq = make_fastdiv((uint32_t)d->OQ);
  p.d_tn = make_fastdiv((uint32_t)p.tiles_n);
  const bool dense = d->R == 1 && d->S == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 &&
                     d->pw == 0 && d->IH == d->OP && d->IW == d->OQ && d->a_sw == d->C &&
                     d->a_sh == (int64_t)d->IW * d->C && d->a_sn == (int64_t)d->IH * d->IW * d->C &&
                     d->y_sw == d->NCOLS && d->y_sh == (int64_t)d->OQ * d->NCOLS &&
                     d->y_sn == (int64_t)d->OP * d->OQ * d->NCOLS;
  hipStream_t st = as_stream(stream);
  // algorithmic work of this launch (bench.py's roofline): FLOPs of the real taps (the stem's padded
  // 8th tap / 4th channel do not count) and HBM bytes = every operand element once
  const int es_out = d->out_f32 ? 4 : es;
  const double kreal = (d->R == 7 && d->S == 1 && d->C == 32) ? 147.0 : (double)K64;
  const double w_flops = 2.0 * (double)M64 * d->NCOLS * kreal;
  double in_px = (double)d->N * d->IH * d->IW;
  if ((double)M64 * d->R * d->S < in_px) in_px = (double)M64 * d->R * d->S;     // strided 1x1: a sub-lattice
  const double w_bytes = in_px * d->C * es + (double)d->NCOLS * K64 * es +
                         (double)M64 * d->NCOLS * es_out * (1 + (d->residual ? 1 : 0)) +
                         (d->bnb_partial ? (double)M64 * d->NCOLS * es : 0.0);
  if (d->bnb2_partial) {
    // two BatchNorm layers behind one gradient (passl_conv_desc.bnb2_*): its own instantiations of the register-staged
    // kernel (reductions below 8 K-tiles) and of the ring kernel, dense 1x1 launches only
    if (!d->bnb_partial || !d->bnb2_y || !d->bnb2_mean || !d->bnb2_invstd || !aligned16(d->bnb2_y)) return PASSL_EINVAL;
    if (!dense || generic || narrow || d->dtype != PASSL_BF16) return PASSL_EUNSUPPORTED;
    const int nk = (p.KDIM + bk - 1) / bk;
    if (nk >= 8) {
      passl_prof_begin(0, st);
      const int rc2 = passl_igemm_ring_bnb2(d, st);
      passl_prof_work(0, w_flops, w_bytes + (double)M64 * d->NCOLS * es);
      passl_prof_end(0, st);
      g_last_kernel = 1;
      return rc2;
    }
    if (nk > nk1_threshold()) return PASSL_EUNSUPPORTED;
    passl_prof_begin(2, st);
    const int rc2 = persist_on() ? launch<bf16_t, 128, 128, false, 1, false, true, false, true, true>(p, st)
                                 : launch<bf16_t, 128, 128, false, 1, false, true, false, true>(p, st);
    passl_prof_work(2, w_flops, w_bytes + (double)M64 * d->NCOLS * es);
    passl_prof_end(2, st);
    g_last_kernel = 0;
    return rc2;
  }
  passl_prof_begin(0, st);
  int rc = passl_conv3x3_wave_try(d, st);          // 64 -> 64 3x3 / stride 1: one wave per 8 x 8 patch, weights resident
  if (rc != PASSL_EUNSUPPORTED) {
    passl_prof_work(0, w_flops, w_bytes);
    passl_prof_end(0, st);
    g_last_kernel = 4;
    return rc;
  }
  rc = passl_igemm_8p_try(d, st);                  // 256 x 256 tiles, 8-phase schedule: wide, deep GEMMs
  if (rc != PASSL_EUNSUPPORTED) {
    passl_prof_retag(0, 3);
    passl_prof_work(3, w_flops, w_bytes);
    passl_prof_end(3, st);
    g_last_kernel = 3;
    return rc;
  }
  rc = passl_igemm_ring_try(d, st);            // 128-row tiles, LDS-DMA ring kernel when it applies
  if (rc != PASSL_EUNSUPPORTED) {
    passl_prof_work(0, w_flops, w_bytes);
    passl_prof_end(0, st);
    g_last_kernel = 1;
    return rc;
  }
  passl_prof_retag(0, 2);
  rc = passl_stem_try(d, st);                  // the spatially tiled stem kernel when it applies
  g_last_kernel = rc == PASSL_EUNSUPPORTED ? 0 : 2;
  if (rc == PASSL_EUNSUPPORTED) {
    const int nk = (p.KDIM + bk - 1) / bk;
    const bool of32 = d->out_f32 != 0;
    if (d->dtype == PASSL_BF16)
      rc = narrow ? dispatch<bf16_t, 64>(p, generic, of32, dense, nk, st) : dispatch<bf16_t, 128>(p, generic, of32, dense, nk, st);
    else
      rc = narrow ? dispatch<float, 64>(p, generic, of32, dense, nk, st) : dispatch<float, 128>(p, generic, of32, dense, nk, st);
  }
  passl_prof_work(2, w_flops, w_bytes);
  passl_prof_end(2, st);
  return rc;
}
