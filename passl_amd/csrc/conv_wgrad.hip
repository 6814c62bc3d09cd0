// Weight-gradient implicit GEMM for gfx950 (MFMA).
//
//   dW[oc][j] += sum_m dY[m][oc] * A[m][j],   m = (n,op,oq) (reduction),  j = (r,s,c)
//
// Both operands are stored with the REDUCTION index m as the slow dimension (NHWC rows), while
// MFMA fragments want 8 consecutive reduction elements per lane.  bf16: rows m, m+1 are loaded
// as two 16-byte channel vectors per lane, interleaved in registers and written to LDS
// transposed ([channel][m], ds_write_b32 of an (m, m+1) pair), so fragment reads are the same
// conflict-free ds_read_b128 as in the forward kernel.  fp32: v_mfma_f32_16x16x4_f32 takes one
// k per lane, so the natural [m][channel] tile is read directly (row pitch padded by 64 B).
//
// Block tile BMo x BNo (64/128 each), 4 waves 2x2, reduction tile = 64 (bf16) / 32 (fp32) rows.
// The reduction over M is split over blockIdx.z; partial tiles are accumulated with fp32
// global atomics (dW is zeroed by the caller).  Per-row gather info (image base, ih0, iw0) for
// the next reduction tile is computed by 64 threads into a double-buffered LDS table.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "wgrad_halo_geom.h"

namespace {

constexpr int kThreads = 256;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

struct WParams {
  const char* a;
  const char* dy;
  float* dw;
  int M, NCOLS, KDIM;
  int OP, OQ, R, S, C, IH, IW, sh, sw, ph, pw;
  int64_t a_sn, a_sh, a_sw, dy_ld;
  int tiles_per_split, nk_total;
  float* ws;        // split-M partial tiles [splits][NCOLS][KDIM] (fixed-order reduction afterwards) or null
  int nsplits;      // number of M-slices of this launch
  int dbg;   // tuning switches (PASSL_WGRAD_DBG): 1 = skip the epilogue stores
  int grid_j, grid_oc;   // tile counts of the 1-D launch of wgrad_dma_kernel
};

struct RowInfo {
  int64_t base;   // n * a_sn, or -1 if the row is out of range
  int ih0, iw0;
};

__device__ __forceinline__ int swz(int row, int slot) {
  return slot ^ (((row >> 1) ^ (row >> 3)) & 7);
}

// One element of a partial tile.  Deterministic forms: a single M-slice owns dW (plain
// read-modify-write), several slices write their tile into slab `bz` of the workspace (summed in slice
// order by slab_reduce_kernel afterwards).  There is no atomic form: a multi-slice launch without a workspace
// is rejected by the host code (PASSL_EINVAL).
__device__ __forceinline__ void wg_store(const WParams& p, int bz, int oc, int jj, float v) {
  const int64_t o = (int64_t)oc * p.KDIM + jj;
  if (p.ws) p.ws[(int64_t)bz * p.NCOLS * p.KDIM + o] = v;
  else p.dw[o] += v;
}

template <typename T, int BMo, int BNo>
__global__ void __launch_bounds__(kThreads, 2) wgrad_kernel(const WParams p) {
  constexpr int ES = sizeof(T);
  constexpr bool BF = ES == 2;
  constexpr int BKM = BF ? 64 : 32;                       // reduction rows per tile
  constexpr int WM = BMo / 2, WN = BNo / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  // bf16: transposed tiles [ch][64 m] -> 128-byte rows.  fp32: natural [32 m][ch] with padded pitch.
  constexpr int PITCH_A = BF ? 128 : BMo * 4 + 64;
  constexpr int PITCH_B = BF ? 128 : BNo * 4 + 64;
  constexpr int A_BYTES = BF ? BMo * 128 : BKM * PITCH_A;
  constexpr int B_BYTES = BF ? BNo * 128 : BKM * PITCH_B;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + 2 * A_BYTES;
  RowInfo* rinfo = reinterpret_cast<RowInfo*>(smem + 2 * (A_BYTES + B_BYTES));  // [2][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int oc0 = blockIdx.y * BMo;
  const int j0 = blockIdx.x * BNo;
  const int opq = p.OP * p.OQ;
  const int kt_begin = blockIdx.z * p.tiles_per_split;
  int kt_end = kt_begin + p.tiles_per_split;
  if (kt_end > p.nk_total) kt_end = p.nk_total;
  if (kt_begin >= kt_end) return;

  auto fill_rowinfo = [&](int kt, int buf) {
    if (tid < BKM) {
      const int m = kt * BKM + tid;
      RowInfo ri;
      if (m < p.M) {
        const int n = m / opq;
        const int rem = m - n * opq;
        const int op = rem / p.OQ;
        const int oq = rem - op * p.OQ;
        ri.base = (int64_t)n * p.a_sn;
        ri.ih0 = op * p.sh - p.ph;
        ri.iw0 = oq * p.sw - p.pw;
      } else {
        ri.base = -1; ri.ih0 = 0; ri.iw0 = 0;
      }
      rinfo[buf * 64 + tid] = ri;
    }
  };

  // ---- per-thread static column decomposition
  // bf16: work item = (channel chunk of 8, row pair).  fp32: (channel chunk of 4, row).
  constexpr int VEC = 16 / ES;
  constexpr int A_CPR = BMo / VEC;                      // chunks per m-row of the dy tile
  constexpr int B_CPR = BNo / VEC;
  constexpr int A_ITEMS = BF ? A_CPR * (BKM / 2) : A_CPR * BKM;
  constexpr int B_ITEMS = BF ? B_CPR * (BKM / 2) : B_CPR * BKM;
  constexpr int A_IT = (A_ITEMS + kThreads - 1) / kThreads;
  constexpr int B_IT = (B_ITEMS + kThreads - 1) / kThreads;
  constexpr int NLD = BF ? 2 : 1;                       // global loads per item

  // x-operand column info per item (r, s, c are fixed for the whole reduction)
  int b_r[B_IT], b_s[B_IT], b_c[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int item = tid + i * kThreads;
    const int cc = item % B_CPR;
    const int j = j0 + cc * VEC;
    b_ok[i] = (item < B_ITEMS) && (j < p.KDIM);
    const int rs = j / p.C;
    b_c[i] = j - rs * p.C;
    b_r[i] = rs / p.S;
    b_s[i] = rs - b_r[i] * p.S;
  }

  uint4 ra[A_IT][NLD], rb[B_IT][NLD];

  auto load_tile = [&](int kt, int ibuf) {
    const int mbase = kt * BKM;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int item = tid + i * kThreads;
      const int cc = item % A_CPR;
      const int mr = item / A_CPR;                      // row (fp32) or row pair (bf16)
      const int oc = oc0 + cc * VEC;
#pragma unroll
      for (int h = 0; h < NLD; ++h) {
        const int m = mbase + (BF ? 2 * mr + h : mr);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (item < A_ITEMS && m < p.M && oc < p.NCOLS)
          v = *reinterpret_cast<const uint4*>(p.dy + ((int64_t)m * p.dy_ld + oc) * ES);
        ra[i][h] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int item = tid + i * kThreads;
      const int mr = item / B_CPR;
#pragma unroll
      for (int h = 0; h < NLD; ++h) {
        const int lr = BF ? 2 * mr + h : mr;            // row within the tile
        uint4 v = make_uint4(0, 0, 0, 0);
        if (b_ok[i]) {
          const RowInfo ri = rinfo[ibuf * 64 + lr];
          const int ih = ri.ih0 + b_r[i], iw = ri.iw0 + b_s[i];
          if (ri.base >= 0 && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) {
            const int64_t off = ri.base + (int64_t)ih * p.a_sh + (int64_t)iw * p.a_sw + b_c[i];
            v = *reinterpret_cast<const uint4*>(p.a + off * ES);
          }
        }
        rb[i][h] = v;
      }
    }
  };

  auto store_one = [&](char* base, int pitch, int cc, int mr, const uint4 (&v)[NLD]) {
    if constexpr (BF) {
      const uint32_t a0[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
      const uint32_t a1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t lo = (e & 1) ? (a0[e >> 1] >> 16) : (a0[e >> 1] & 0xffffu);
        const uint32_t hi = (e & 1) ? (a1[e >> 1] & 0xffff0000u) : (a1[e >> 1] << 16);
        const int ch = cc * 8 + e;
        *reinterpret_cast<uint32_t*>(base + ch * 128 + (swz(ch, mr >> 2) << 4) + ((mr & 3) << 2)) = lo | hi;
      }
    } else {
      *reinterpret_cast<uint4*>(base + mr * pitch + cc * 16) = v[0];
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int item = tid + i * kThreads;
      if (item < A_ITEMS) store_one(As + buf * A_BYTES, PITCH_A, item % A_CPR, item / A_CPR, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int item = tid + i * kThreads;
      if (item < B_ITEMS) store_one(Bs + buf * B_BYTES, PITCH_B, item % B_CPR, item / B_CPR, rb[i]);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fill_rowinfo(kt_begin, 0);
  __syncthreads();
  load_tile(kt_begin, 0);
  if (kt_begin + 1 < kt_end) fill_rowinfo(kt_begin + 1, 1);
  store_tile(0);
  __syncthreads();

  const int l15 = lane & 15, l4 = lane >> 4;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int it = kt - kt_begin;
    const int buf = it & 1;
    if (kt + 1 < kt_end) load_tile(kt + 1, buf ^ 1);
    if (kt + 2 < kt_end) fill_rowinfo(kt + 2, buf);
    const char* Ab = As + buf * A_BYTES;
    const char* Bb = Bs + buf * B_BYTES;
    if constexpr (BF) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WM + i * 16 + l15;
          const uint4 v = *reinterpret_cast<const uint4*>(Ab + row * 128 + (swz(row, ks * 4 + l4) << 4));
          af[i] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = wn * WN + j * 16 + l15;
          const uint4 v = *reinterpret_cast<const uint4*>(Bb + row * 128 + (swz(row, ks * 4 + l4) << 4));
          bfr[j] = __builtin_bit_cast(bf16x8_t, v);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        float af[FM], bfr[FN];
        const int mrow = ks * 4 + l4;
#pragma unroll
        for (int i = 0; i < FM; ++i)
          af[i] = *reinterpret_cast<const float*>(Ab + mrow * PITCH_A + (wm * WM + i * 16 + l15) * 4);
#pragma unroll
        for (int j = 0; j < FN; ++j)
          bfr[j] = *reinterpret_cast<const float*>(Bb + mrow * PITCH_B + (wn * WN + j * 16 + l15) * 4);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < kt_end) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: partial tile -> dW[oc][j] / workspace slab (wg_store)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = oc0 + wm * WM + i * 16 + l4 * 4 + r;
        const int jj = j0 + wn * WN + j * 16 + l15;
        if (oc < p.NCOLS && jj < p.KDIM) wg_store(p, blockIdx.z, oc, jj, acc[i][j][r]);
      }
}

// number of M-slices the launch functions actually used (they may shrink the request); read by
// passl_hip_conv_wgrad right after the dispatch to size the fixed-order reduction
thread_local int t_eff_splits = 1;

template <typename T, int BMo, int BNo>
int launch(const WParams& p0, int splits, hipStream_t st) {
  WParams p = p0;
  p.nsplits = splits;
  t_eff_splits = splits;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int BKM = BF ? 64 : 32;
  constexpr int A_BYTES = BF ? BMo * 128 : BKM * (BMo * 4 + 64);
  constexpr int B_BYTES = BF ? BNo * 128 : BKM * (BNo * 4 + 64);
  constexpr int LDS = 2 * (A_BYTES + B_BYTES) + 2 * 64 * (int)sizeof(RowInfo);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BMo, BNo>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  dim3 grid((p.KDIM + BNo - 1) / BNo, (p.NCOLS + BMo - 1) / BMo, splits);
  hipLaunchKernelGGL((wgrad_kernel<T, BMo, BNo>), grid, dim3(kThreads), LDS, st, p);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

template <typename T>
int dispatch(const WParams& p, int splits, hipStream_t st) {
  const bool m64 = p.NCOLS <= 64, n64 = p.KDIM <= 64;
  if (m64 && n64) return launch<T, 64, 64>(p, splits, st);
  if (m64) return launch<T, 64, 128>(p, splits, st);
  if (n64) return launch<T, 128, 64>(p, splits, st);
  return launch<T, 128, 128>(p, splits, st);
}


// ------------------------------------------------------------------------------------------------
// bf16 fast path: LDS-DMA staging + hardware transpose reads.
//
// Both operands keep their NATURAL layout in LDS ([64 reduction rows m][channels], 16-byte chunks
// XOR-swizzled within a row), written by `buffer_load_dwordx4 ... lds` (no VGPR staging, no
// ds_write, out-of-range lanes fetch zeros through the buffer bounds check), and the MFMA
// fragments (8 consecutive m per lane) come from two `ds_read_b64_tr_b16` each: a 16-lane group
// addresses a [4 m][16 ch] block (lane j -> row j>>2, 4 channels at 4*(j&3)) and lane i receives
// channel i of the 4 rows.  Swizzle: the 32-byte chunk pair index is XORed with h(row), chosen so
// that the 8 row segments a 32-lane group touches fall on disjoint banks (see hswz).
// Pipeline: 2 LDS stages (64 KB) -> 2 workgroups per CU; the DMA of tile t+1 is in flight while
// tile t is multiplied; one barrier per tile.
struct RowInfo2 {
  uint32_t base;   // byte offset of image n
  int ih0, iw0;
  int valid;
};

template <int CPR>   // 16-byte chunks per LDS row (16: 128 channels, 8: 64 channels)
__device__ __forceinline__ int hswz(int row) {
  return CPR == 16 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}

constexpr uint32_t kOOB = 0x7ffffff0u;   // beyond every buffer's num_records -> the load returns 0

template <int BMo, int BNo>
__global__ void __launch_bounds__(kThreads, 2) wgrad_dma_kernel(const WParams p, uint32_t a_bytes,
                                                                uint32_t dy_bytes) {
  constexpr int BKM = 64;
  constexpr int WM = BMo / 2, WN = BNo / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int CPA = BMo / 8, CPB = BNo / 8;          // chunks per row
  constexpr int PA = BMo * 2, PB = BNo * 2;            // row pitch (bytes)
  constexpr int A_BYTES = BKM * PA, B_BYTES = BKM * PB;
  constexpr int NIA = A_BYTES / 1024 / 4, NIB = B_BYTES / 1024 / 4;   // DMA instructions per wave
  constexpr int RPA = 1024 / PA, RPB = 1024 / PB;      // rows per DMA instruction
  constexpr int STAGE = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  RowInfo2* rinfo = reinterpret_cast<RowInfo2*>(smem + 2 * STAGE);   // [2][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid, XCD-aware: hardware block b runs on XCD b % 8 (private L2); give every XCD a
  // contiguous range of virtual blocks, j-tiles fastest, so that the j-tiles sharing one dy tile
  // (same oc-tile, same M-slice) and the oc-tiles sharing one x tile hit the same L2.  Pure speed:
  // any placement is correct (bijective for every grid size).
  int vb;
  {
    const int bid = blockIdx.x, nb = gridDim.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int bx = vb % p.grid_j, by = (vb / p.grid_j) % p.grid_oc, bz = vb / (p.grid_j * p.grid_oc);
  const int oc0 = by * BMo;
  const int j0 = bx * BNo;
  const int opq = p.OP * p.OQ;
  const int kt_begin = bz * p.tiles_per_split;
  int kt_end = kt_begin + p.tiles_per_split;
  if (kt_end > p.nk_total) kt_end = p.nk_total;
  if (kt_begin >= kt_end) return;

  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.a), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dy), 0, dy_bytes, 0x00020000);

  // ---- static per-thread DMA geometry (tile invariant: rows advance by 64 = 0 mod 16)
  int a_row[NIA]; uint32_t a_col[NIA];                 // dy tile
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int q = i * 4 + wave;
    const int row = q * RPA + lane / CPA;
    const int chunk = (lane % CPA) ^ (hswz<CPA>(row) << 1);
    const int oc = oc0 + chunk * 8;
    a_row[i] = row;
    a_col[i] = oc < p.NCOLS ? (uint32_t)oc * 2u : kOOB;
  }
  int b_row[NIB], b_r[NIB], b_s[NIB]; uint32_t b_c[NIB];   // x tile
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int q = i * 4 + wave;
    const int row = q * RPB + lane / CPB;
    const int chunk = (lane % CPB) ^ (hswz<CPB>(row) << 1);
    const int j = j0 + chunk * 8;
    const int rs = j / p.C;
    b_row[i] = row;
    b_r[i] = rs / p.S;
    b_s[i] = rs - b_r[i] * p.S;
    b_c[i] = j < p.KDIM ? (uint32_t)(j - rs * p.C) * 2u : kOOB;
  }

  auto fill_rowinfo = [&](int kt, int buf) {
    if (tid < BKM) {
      const int m = kt * BKM + tid;
      RowInfo2 ri;
      if (m < p.M) {
        const int n = m / opq;
        const int rem = m - n * opq;
        const int op = rem / p.OQ;
        const int oq = rem - op * p.OQ;
        ri.base = (uint32_t)((int64_t)n * p.a_sn * 2);
        ri.ih0 = op * p.sh - p.ph;
        ri.iw0 = oq * p.sw - p.pw;
        ri.valid = 1;
      } else {
        ri.base = 0; ri.ih0 = 0; ri.iw0 = 0; ri.valid = 0;
      }
      rinfo[buf * 64 + tid] = ri;
    }
  };

  auto issue_tile = [&](int kt, int sbuf, int ibuf) {
    char* Ab = smem + sbuf * STAGE;
    char* Bb = Ab + A_BYTES;
    const int mbase = kt * BKM;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int m = mbase + a_row[i];
      uint32_t off = kOOB;
      if (m < p.M && a_col[i] != kOOB) off = (uint32_t)m * (uint32_t)(p.dy_ld * 2) + a_col[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_dy, (__attribute__((address_space(3))) void*)(Ab + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const RowInfo2 ri = rinfo[ibuf * 64 + b_row[i]];
      const int ih = ri.ih0 + b_r[i], iw = ri.iw0 + b_s[i];
      uint32_t off = kOOB;
      if (ri.valid && b_c[i] != kOOB && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
        off = ri.base + (uint32_t)(ih * (int)(p.a_sh * 2) + iw * (int)(p.a_sw * 2)) + b_c[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_a, (__attribute__((address_space(3))) void*)(Bb + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  fill_rowinfo(kt_begin, 0);
  __syncthreads();
  issue_tile(kt_begin, 0, 0);
  if (kt_begin + 1 < kt_end) fill_rowinfo(kt_begin + 1, 1);

  // transpose-read addressing: 16-lane group g reads rows 8g + 4*half + (lane>>2)&3, lane's own
  // 8 bytes sit at channel 4*(lane&3) of the fragment's 16-channel block
  const int l15 = lane & 15, l4 = lane >> 4;
  const int r4 = (lane >> 2) & 3, c4 = lane & 3;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* Ab = smem + buf * STAGE;
    const char* Bb = Ab + A_BYTES;
    // ALL fragment reads of this tile are issued BEFORE the next tile's DMA: hipcc orders every
    // LDS read that follows an LDS-DMA behind a full `s_waitcnt vmcnt(0)` (it cannot prove the
    // two touch different stages), which would serialise the DMA with the MFMAs.
    bf16x8_t af[2][FM], bfr[2][FN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = ks * 32 + 8 * l4 + 4 * h + r4;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int pair = (wm * WM) / 16 + i;
          const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4_t*)(Ab + row * PA + ((pair ^ hswz<CPA>(row)) << 5) + c4 * 8));
          af[ks][i][4 * h + 0] = v[0]; af[ks][i][4 * h + 1] = v[1];
          af[ks][i][4 * h + 2] = v[2]; af[ks][i][4 * h + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int pair = (wn * WN) / 16 + j;
          const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4_t*)(Bb + row * PB + ((pair ^ hswz<CPB>(row)) << 5) + c4 * 8));
          bfr[ks][j][4 * h + 0] = v[0]; bfr[ks][j][4 * h + 1] = v[1];
          bfr[ks][j][4 * h + 2] = v[2]; bfr[ks][j][4 * h + 3] = v[3];
        }
      }
    }
    if (kt + 1 < kt_end) issue_tile(kt + 1, buf ^ 1, buf ^ 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    if (kt + 2 < kt_end) fill_rowinfo(kt + 2, buf);
  }

  // ---- epilogue: partial tile -> dW[oc][j] / workspace slab (wg_store)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = oc0 + wm * WM + i * 16 + l4 * 4 + r;
        const int jj = j0 + wn * WN + j * 16 + l15;
        if (oc < p.NCOLS && jj < p.KDIM && !(p.dbg & 1)) wg_store(p, bz, oc, jj, acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------------------
// Dense variant (1x1 stride-1 unpadded convolutions over a contiguous NHWC tensor and every Linear
// layer: x row m = bytes [m*C*2, (m+1)*C*2)): a 4-stage ring of 32-row tiles with register
// double-buffering of the MFMA fragments.  Compared with wgrad_dma_kernel: no row-info table
// (offsets are linear in m), three tiles in flight per workgroup instead of one (the 64 KB ring is
// cut into 4 x 16 KB), the DMA of tile t+4 is issued as soon as the barrier of iteration t proves
// stage t % 4 free, and the fragments of tile t+1 are fetched while tile t is multiplied.  The
// transposing LDS reads are inline asm: hipcc orders any compiler-visible LDS read behind a full
// `s_waitcnt vmcnt(0)` whenever an LDS-DMA is outstanding, which would drain the ring every iteration.
template <int OFF>
__device__ __forceinline__ uint2 lds_read_tr64(uint32_t addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int BMo, int BNo, int BKM, int NST, bool DENSE>
__global__ void __launch_bounds__(kThreads, 2) wgrad_pipe_kernel(const WParams p, int64_t a_total,
                                                                 int64_t dy_total) {
  constexpr int KS = BKM / 32;                         // 16x16x32 MFMA k-steps per stage
  static_assert((BKM == 32 || BKM == 64) && NST >= 2 && NST <= 4, "ring shape");
  constexpr int WM = BMo / 2, WN = BNo / 2;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int CPA = BMo / 8, CPB = BNo / 8;          // 16-byte chunks per row
  constexpr int PA = BMo * 2, PB = BNo * 2;            // row pitch (bytes)
  constexpr int A_BYTES = BKM * PA, B_BYTES = BKM * PB;
  constexpr int NIA = A_BYTES / 1024 / 4, NIB = B_BYTES / 1024 / 4;   // DMA instructions per wave
  constexpr int RPA = 1024 / PA, RPB = 1024 / PB;      // rows per DMA instruction
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int IPT = NIA + NIB;
  static_assert(NIA >= 1 && NIB >= 1, "tile too narrow for one DMA instruction per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  // general (gathered) x rows: per-row image base / input origin, 4 slots of BKM rows, slot = tile % 4
  RowInfo2* rinfo = reinterpret_cast<RowInfo2*>(smem + NST * STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int vb;
  {
    const int bid = blockIdx.x, nb = gridDim.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int q = nb >> 3, r = nb & 7;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  }
  const int bx = vb % p.grid_j, by = (vb / p.grid_j) % p.grid_oc, bz = vb / (p.grid_j * p.grid_oc);
  const int oc0 = by * BMo;
  const int j0 = bx * BNo;
  const int opq = p.OP * p.OQ;
  const int kt_begin = bz * p.tiles_per_split;
  int kt_end = kt_begin + p.tiles_per_split;
  if (kt_end > p.nk_total) kt_end = p.nk_total;
  if (kt_begin >= kt_end) return;
  const int nk = kt_end - kt_begin;

  // Both operands may exceed what a buffer descriptor addresses (32-bit byte offsets): this workgroup's
  // descriptors start at the first row (x gathered: the first image) of ITS M-slice and every lane offset is
  // relative to that; the host sizes the slices so that one slice spans < 2 GB of each operand.
  const int m_begin = kt_begin * BKM;
  const int nb0 = m_begin / opq;
  const uint32_t dy_pitch = (uint32_t)(p.dy_ld * 2), x_pitch = (uint32_t)(p.C * 2);
  const int64_t dy_off0 = (int64_t)m_begin * dy_pitch;
  const int64_t a_off0 = DENSE ? (int64_t)m_begin * x_pitch : (int64_t)nb0 * p.a_sn * 2;
  const int64_t dy_left = dy_total - dy_off0, a_left = a_total - a_off0;
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.a) + a_off0, 0, (uint32_t)(a_left < 0x7ffffff0ll ? a_left : 0x7ffffff0ll), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(p.dy) + dy_off0, 0, (uint32_t)(dy_left < 0x7ffffff0ll ? dy_left : 0x7ffffff0ll), 0x00020000);

  // ---- static per-thread DMA geometry (rows advance by 32 per tile: the swizzle term is invariant)
  int a_row[NIA]; uint32_t a_col[NIA];                 // dy tile
#pragma unroll
  for (int i = 0; i < NIA; ++i) {
    const int q = i * 4 + wave;
    const int row = q * RPA + lane / CPA;
    const int chunk = (lane % CPA) ^ (hswz<CPA>(row) << 1);
    const int oc = oc0 + chunk * 8;
    a_row[i] = row;
    a_col[i] = oc < p.NCOLS ? (uint32_t)oc * 2u : kOOB;
  }
  int b_row[NIB], b_r[NIB], b_s[NIB]; uint32_t b_col[NIB];   // x tile
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int q = i * 4 + wave;
    const int row = q * RPB + lane / CPB;
    const int chunk = (lane % CPB) ^ (hswz<CPB>(row) << 1);
    const int j = j0 + chunk * 8;
    b_row[i] = row;
    if (DENSE) {
      b_r[i] = 0; b_s[i] = 0;
      b_col[i] = j < p.KDIM ? (uint32_t)j * 2u : kOOB;
    } else {
      const int rs = j / p.C;
      b_r[i] = rs / p.S;
      b_s[i] = rs - b_r[i] * p.S;
      b_col[i] = j < p.KDIM ? (uint32_t)(j - rs * p.C) * 2u : kOOB;
    }
  }
  auto fill_rowinfo = [&](int t) {                     // tile t (relative), slot t & 3
    if (!DENSE && tid < BKM) {
      const int m = (kt_begin + t) * BKM + tid;
      RowInfo2 ri;
      if (m < p.M) {
        const int n = m / opq;
        const int rem = m - n * opq;
        const int op = rem / p.OQ;
        const int oq = rem - op * p.OQ;
        ri.base = (uint32_t)((int64_t)(n - nb0) * p.a_sn * 2);
        ri.ih0 = op * p.sh - p.ph;
        ri.iw0 = oq * p.sw - p.pw;
        ri.valid = 1;
      } else {
        ri.base = 0; ri.ih0 = 0; ri.iw0 = 0; ri.valid = 0;
      }
      rinfo[(t & 3) * 64 + tid] = ri;
    }
  };

  int issued = 0;                                      // tiles issued so far (relative to kt_begin)
  auto issue_tile = [&]() {
    char* Ab = smem + (issued % NST) * STAGE;
    char* Bb = Ab + A_BYTES;
    const int mbase = (kt_begin + issued) * BKM;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int m = mbase + a_row[i];
      const uint32_t off = (m < p.M && a_col[i] != kOOB) ? (uint32_t)(m - m_begin) * dy_pitch + a_col[i] : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_dy, (__attribute__((address_space(3))) void*)(Ab + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      uint32_t off = kOOB;
      if (DENSE) {
        const int m = mbase + b_row[i];
        if (m < p.M && b_col[i] != kOOB) off = (uint32_t)(m - m_begin) * x_pitch + b_col[i];
      } else {
        // read right after the iteration's `vmcnt(0)` + barrier: no DMA is outstanding, so the
        // compiler's conservative wait in front of this LDS read costs nothing
        const RowInfo2 ri = rinfo[(issued & 3) * 64 + b_row[i]];
        const int ih = ri.ih0 + b_r[i], iw = ri.iw0 + b_s[i];
        if (ri.valid && b_col[i] != kOOB && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW)
          off = ri.base + (uint32_t)(ih * (int)(p.a_sh * 2) + iw * (int)(p.a_sw * 2)) + b_col[i];
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_a, (__attribute__((address_space(3))) void*)(Bb + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
    }
    ++issued;
  };

  // ---- transposing fragment reads: 16-lane group g = l4 reads rows 8g + 4*half + (lane>>2)&3, the
  // lane's own 8 bytes sit at channel 4*(lane&3) of the fragment's 16-channel block
  const int l15 = lane & 15, l4 = lane >> 4;
  const int r4 = (lane >> 2) & 3, c4 = lane & 3;
  // rows 32 further down (second k-step of a 64-row stage) keep the swizzle term (hswz uses row bits
  // 0, 1 and 3 only): they are reached with the immediate offset 32 * pitch
  uint32_t a_rd[2][FM], b_rd[2][FN];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * l4 + 4 * h + r4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int pair = (wm * WM) / 16 + i;
      a_rd[h][i] = (uint32_t)(row * PA + ((pair ^ hswz<CPA>(row)) << 5) + c4 * 8);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int pair = (wn * WN) / 16 + j;
      b_rd[h][j] = (uint32_t)(A_BYTES + row * PB + ((pair ^ hswz<CPB>(row)) << 5) + c4 * 8);
    }
  }
  uint4 af[2][KS][FM], bfr[2][KS][FN];
  auto read_frags = [&](auto SET, int t) {
    constexpr int S_ = decltype(SET)::value;
    const uint32_t sb = lds0 + (uint32_t)((t % NST) * STAGE);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const uint2 lo = lds_read_tr64<0>(sb + a_rd[0][i]), hi = lds_read_tr64<0>(sb + a_rd[1][i]);
      af[S_][0][i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if constexpr (KS == 2) {
        const uint2 lo2 = lds_read_tr64<32 * PA>(sb + a_rd[0][i]), hi2 = lds_read_tr64<32 * PA>(sb + a_rd[1][i]);
        af[S_][1][i] = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
      }
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const uint2 lo = lds_read_tr64<0>(sb + b_rd[0][j]), hi = lds_read_tr64<0>(sb + b_rd[1][j]);
      bfr[S_][0][j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if constexpr (KS == 2) {
        const uint2 lo2 = lds_read_tr64<32 * PB>(sb + b_rd[0][j]), hi2 = lds_read_tr64<32 * PB>(sb + b_rd[1][j]);
        bfr[S_][1][j] = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Ring protocol (cf. conv_igemm_ring.hip).  Before iteration t: tiles <= t+3 issued, tile t in the
  // register set t & 1.  Iteration t: wait for the own DMAs of tile t+1 (up to two newer tiles stay
  // in flight) -> barrier (everybody's tile t+1 landed; everybody finished reading stage t % 4 during
  // iteration t-1) -> issue tile t+4 into stage t % 4 -> fetch tile t+1 into the other register set
  // -> MFMAs of tile t -> lgkmcnt(0).
  auto iteration = [&](auto SET, int t) {
    constexpr int S_ = decltype(SET)::value;
    if (t + 1 < nk) {
      // tiles issued after t+1 so far: min(NST - 2, nk - 2 - t)
      if (NST >= 4 && t + 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
      else if (NST >= 3 && t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + NST < nk) issue_tile();
      read_frags(std::integral_constant<int, 1 - S_>{}, t + 1);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[S_][ks][i]),
                                                              __builtin_bit_cast(bf16x8_t, bfr[S_][ks][j]),
                                                              acc[i][j], 0, 0, 0);
    // row info of tile t+NST+1: its slot held tile t+NST-3's, consumed when that tile was issued.
    // Written BEFORE the lgkmcnt(0) below so that the store has completed when this wave reaches the
    // next iteration's barrier (raw s_barrier carries no LDS wait of its own).
    if (!DENSE && t + NST + 1 < nk) fill_rowinfo(t + NST + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  if (!DENSE) {
    static_assert(DENSE || NST == 2, "the gathered variant's row-info ring (4 slots) is sized for 2 stages");
    for (int t = 0; t < NST + 1 && t < nk; ++t) fill_rowinfo(t);
    __syncthreads();
  }
  for (int t = 0; t < NST && t < nk; ++t) issue_tile();
  // tile 0 landed: min(NST, nk) - 1 newer tiles may stay in flight
  if (NST >= 4 && nk >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPT) : "memory");
  else if (NST >= 3 && nk >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
  else if (nk >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(std::integral_constant<int, 0>{}, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < nk; kt += 2) {
    iteration(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) iteration(std::integral_constant<int, 1>{}, kt + 1);
  }

  // ---- epilogue: partial tile -> dW[oc][j] / workspace slab (wg_store)
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = oc0 + wm * WM + i * 16 + l4 * 4 + r;
        const int jj = j0 + wn * WN + j * 16 + l15;
        if (oc < p.NCOLS && jj < p.KDIM && !(p.dbg & 1)) wg_store(p, bz, oc, jj, acc[i][j][r]);
      }
}

template <int BMo, int BNo, int BKM, int NST, bool DENSE>
int launch_pipe(const WParams& p, int splits, int64_t a_bytes, int64_t dy_bytes, hipStream_t st) {
  constexpr int LDS = NST * BKM * (BMo + BNo) * 2 + (DENSE ? 0 : 4 * 64 * (int)sizeof(RowInfo2));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pipe_kernel<BMo, BNo, BKM, NST, DENSE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  WParams q = p;
  q.nk_total = (p.M + BKM - 1) / BKM;
  if (splits > q.nk_total) splits = q.nk_total;
  q.tiles_per_split = (q.nk_total + splits - 1) / splits;
  splits = (q.nk_total + q.tiles_per_split - 1) / q.tiles_per_split;
  q.grid_j = (p.KDIM + BNo - 1) / BNo;
  q.grid_oc = (p.NCOLS + BMo - 1) / BMo;
  q.nsplits = splits;
  {
    // one M-slice must stay inside a 2 GB window of each operand (descriptors are rebased per slice)
    const int64_t lim = 0x7ffffff0ll;
    const int64_t rows = (int64_t)q.tiles_per_split * BKM;
    const int64_t opq = (int64_t)p.OP * p.OQ;
    const int64_t x_span = DENSE ? rows * p.C * 2 : (rows / opq + 2) * p.a_sn * 2;
    if (rows * p.dy_ld * 2 >= lim || x_span >= lim) return PASSL_EUNSUPPORTED;
  }
  t_eff_splits = splits;
  hipLaunchKernelGGL((wgrad_pipe_kernel<BMo, BNo, BKM, NST, DENSE>), dim3(q.grid_j * q.grid_oc * splits),
                     dim3(kThreads), LDS, st, q, a_bytes, dy_bytes);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

template <int BMo, int BNo>
int launch_dma(const WParams& p, int splits, uint32_t a_bytes, uint32_t dy_bytes, hipStream_t st) {
  constexpr int LDS = 2 * 64 * (BMo + BNo) * 2 + 2 * 64 * (int)sizeof(RowInfo2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<BMo, BNo>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  WParams q = p;
  q.grid_j = (p.KDIM + BNo - 1) / BNo;
  q.grid_oc = (p.NCOLS + BMo - 1) / BMo;
  q.nsplits = splits;
  t_eff_splits = splits;
  hipLaunchKernelGGL((wgrad_dma_kernel<BMo, BNo>), dim3(q.grid_j * q.grid_oc * splits), dim3(kThreads),
                     LDS, st, q, a_bytes, dy_bytes);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

#include "conv_wgrad_halo.inc"

int g_wgrad_tile = 0;     // 0 = by shape; 1 = 64x64, 2 = 64x128, 3 = 128x64, 4 = 128x128 (experiments)

// wgrad_pipe_kernel (register double-buffered fragments): 2 = 2 x 64-row stages for every shape (default),
// 3 = same, 1 = 4 x 32-row stages for dense 128x128 shapes only, 0 = wgrad_dma_kernel everywhere
int g_wgrad_pipe = 2;

// x is a contiguous [M][C] matrix: 1x1, stride 1, no padding over a dense NHWC tensor (or a Linear)
bool dense_rows(const WParams& p) {
  return p.R == 1 && p.S == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.OP == p.IH &&
         p.OQ == p.IW && p.a_sw == p.C && p.a_sh == (int64_t)p.IW * p.C &&
         p.a_sn == (int64_t)p.IH * p.IW * p.C;
}

int dispatch_dma(const WParams& p, int splits, int64_t a_bytes, int64_t dy_bytes, hipStream_t st) {
  const int64_t lim32 = 0x7ffffff0ll;
  const bool small = a_bytes < lim32 && dy_bytes < lim32;     // the first-generation kernel addresses globally
  if (g_wgrad_pipe && g_wgrad_tile == 0) {
    const bool m64 = p.NCOLS <= 64, n64 = p.KDIM <= 64;
    if (g_wgrad_pipe == 1 && dense_rows(p) && !m64 && !n64)
      return launch_pipe<128, 128, 32, 4, true>(p, splits, a_bytes, dy_bytes, st);
    if (dense_rows(p)) {
      if (m64 && n64) return launch_pipe<64, 64, 64, 2, true>(p, splits, a_bytes, dy_bytes, st);
      if (m64) return launch_pipe<64, 128, 64, 2, true>(p, splits, a_bytes, dy_bytes, st);
      if (n64) return launch_pipe<128, 64, 64, 2, true>(p, splits, a_bytes, dy_bytes, st);
      return launch_pipe<128, 128, 64, 2, true>(p, splits, a_bytes, dy_bytes, st);
    }
    if (g_wgrad_pipe >= 2) {
      if (m64 && n64) return launch_pipe<64, 64, 64, 2, false>(p, splits, a_bytes, dy_bytes, st);
      if (m64) return launch_pipe<64, 128, 64, 2, false>(p, splits, a_bytes, dy_bytes, st);
      if (n64) return launch_pipe<128, 64, 64, 2, false>(p, splits, a_bytes, dy_bytes, st);
      return launch_pipe<128, 128, 64, 2, false>(p, splits, a_bytes, dy_bytes, st);
    }
  }
  if (!small) return PASSL_EUNSUPPORTED;
  const uint32_t a32 = (uint32_t)a_bytes, dy32 = (uint32_t)dy_bytes;
  if (g_wgrad_tile == 1) return launch_dma<64, 64>(p, splits, a32, dy32, st);
  if (g_wgrad_tile == 2) return launch_dma<64, 128>(p, splits, a32, dy32, st);
  if (g_wgrad_tile == 3) return launch_dma<128, 64>(p, splits, a32, dy32, st);
  if (g_wgrad_tile == 4) return launch_dma<128, 128>(p, splits, a32, dy32, st);
  const bool m64 = p.NCOLS <= 64, n64 = p.KDIM <= 64;
  if (m64 && n64) return launch_dma<64, 64>(p, splits, a32, dy32, st);
  if (m64) return launch_dma<64, 128>(p, splits, a32, dy32, st);
  if (n64) return launch_dma<128, 64>(p, splits, a32, dy32, st);
  return launch_dma<128, 128>(p, splits, a32, dy32, st);
}

}  // namespace

int passl_slab_reduce_launch(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                             hipStream_t st);   // flat.hip

int passl_wgrad_option(const char* name, int value) {
  if (strcmp(name, "wgrad_tile") == 0) { g_wgrad_tile = value; return PASSL_OK; }
  if (strcmp(name, "wgrad_pipe") == 0) { g_wgrad_pipe = value; return PASSL_OK; }
  // spatially tiled 3x3 kernel (opt-in): 1 = images whose sides are multiples of 8, 2 = every 3x3 / stride-1 layer
  if (strcmp(name, "wgrad_halo") == 0) { g_wgrad_halo = value < 0 ? 0 : (value > 2 ? 2 : value); return PASSL_OK; }
  if (strcmp(name, "wgrad_halo_stages") == 0) {
    if (value != 2 && value != 3) return PASSL_EINVAL;
    g_wgrad_halo_nst = value;
    return PASSL_OK;
  }
  return PASSL_EINVAL;
}

extern "C" int passl_hip_conv_wgrad(const passl_wgrad_desc* d, passl_stream_t stream) {
  if (!d || !d->a || !d->dy || !d->dw) return PASSL_EINVAL;
  if (d->N <= 0 || d->OP <= 0 || d->OQ <= 0 || d->NCOLS <= 0 || d->R <= 0 || d->S <= 0 ||
      d->C <= 0 || d->IH <= 0 || d->IW <= 0 || d->splits <= 0)
    return PASSL_EINVAL;
  if (d->dtype != PASSL_F32 && d->dtype != PASSL_BF16) return PASSL_EUNSUPPORTED;
  const int es = d->dtype == PASSL_BF16 ? 2 : 4;
  const int vec = 16 / es;
  if ((d->C % vec) || (d->NCOLS % vec) || (d->dy_ld % vec)) return PASSL_EINVAL;
  if ((d->a_sn % vec) || (d->a_sh % vec) || (d->a_sw % vec)) return PASSL_EINVAL;
  if (!aligned16(d->a) || !aligned16(d->dy) || (reinterpret_cast<uintptr_t>(d->dw) & 3))
    return PASSL_EINVAL;
  const int64_t M64 = (int64_t)d->N * d->OP * d->OQ;
  const int64_t K64 = (int64_t)d->R * d->S * d->C;
  if (M64 > 0x7fffffff || K64 > 0x7fffffff) return PASSL_EINVAL;
  WParams p;
  p.a = reinterpret_cast<const char*>(d->a);
  p.dy = reinterpret_cast<const char*>(d->dy);
  p.dw = d->dw;
  p.M = (int)M64; p.NCOLS = d->NCOLS; p.KDIM = (int)K64;
  p.OP = d->OP; p.OQ = d->OQ; p.R = d->R; p.S = d->S; p.C = d->C;
  p.IH = d->IH; p.IW = d->IW; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw;
  p.a_sn = d->a_sn; p.a_sh = d->a_sh; p.a_sw = d->a_sw; p.dy_ld = d->dy_ld;
  p.ws = nullptr; p.nsplits = 1;
  const int bkm = d->dtype == PASSL_BF16 ? 64 : 32;
  p.nk_total = (p.M + bkm - 1) / bkm;
  int splits = d->splits;
  if (splits > p.nk_total) splits = p.nk_total;
  p.tiles_per_split = (p.nk_total + splits - 1) / splits;
  splits = (p.nk_total + p.tiles_per_split - 1) / p.tiles_per_split;
  {
    static int dyn = -1;
    static int dbg = 0;
    if (dyn < 0) dyn = getenv("PASSL_WGRAD_DBG_DYNAMIC") ? 1 : 0;
    if (dyn || dbg == 0) { const char* e = getenv("PASSL_WGRAD_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
  }
  const int64_t n_out = (int64_t)d->NCOLS * K64;
  if (splits > 1 && !d->ws) return PASSL_EINVAL;      // partial tiles need their slabs (no fp32 atomics)
  if (splits > 1 && d->ws) {
    if (!aligned16(d->ws) || !aligned16(d->dw) || d->ws_floats < (int64_t)splits * n_out) return PASSL_EINVAL;
    p.ws = d->ws;
  }
  hipStream_t st = as_stream(stream);
  passl_prof_begin(1, st);
  int rc;
  // bf16: LDS-DMA + transpose-read kernel when both operands are addressable with 32-bit byte
  // offsets (buffer addressing); the register-staged kernel otherwise and for fp32
  static int use_dma = -1;
  if (use_dma < 0) { const char* e = getenv("PASSL_WGRAD_DMA"); use_dma = e ? atoi(e) : 1; }
  const int64_t a_bytes = ((int64_t)d->N * d->a_sn) * 2;
  const int64_t dy_bytes = (M64 * d->dy_ld) * 2;
  const int64_t lim = 0x7ffffff0ll;
  (void)lim;
  rc = PASSL_EUNSUPPORTED;
  if (d->dtype == PASSL_BF16 && use_dma && a_bytes > 0)
    rc = wgrad_halo_try(p, splits, a_bytes, dy_bytes, st);     // opt-in; EUNSUPPORTED otherwise
  if (rc == PASSL_EUNSUPPORTED && d->dtype == PASSL_BF16 && use_dma && a_bytes > 0)
    rc = dispatch_dma(p, splits, a_bytes, dy_bytes, st);       // EUNSUPPORTED: a slice spans >= 2 GB
  if (rc == PASSL_EUNSUPPORTED)
    rc = d->dtype == PASSL_BF16 ? dispatch<bf16_t>(p, splits, st) : dispatch<float>(p, splits, st);
  // several slices wrote slabs: dw += slab 0 + slab 1 + ... in slice order (one slice wrote dw itself)
  if (rc == PASSL_OK && p.ws && t_eff_splits > 1)
    rc = passl_slab_reduce_launch(p.ws, p.dw, n_out, t_eff_splits, 1, st);
  else if (rc == PASSL_OK && p.ws && t_eff_splits == 1)
    rc = passl_slab_reduce_launch(p.ws, p.dw, n_out, 1, 1, st);
  passl_prof_work(1, 2.0 * (double)M64 * d->NCOLS * (double)K64,
                  ((double)d->N * d->IH * d->IW * d->C + (double)M64 * d->NCOLS) * (d->dtype == PASSL_BF16 ? 2 : 4) +
                      (double)d->NCOLS * K64 * 4);
  passl_prof_end(1, st);
  return rc;
}
