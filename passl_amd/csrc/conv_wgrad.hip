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
#include "common.h"
#include "prof.h"

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
};

struct RowInfo {
  int64_t base;   // n * a_sn, or -1 if the row is out of range
  int ih0, iw0;
};

__device__ __forceinline__ int swz(int row, int slot) {
  return slot ^ (((row >> 1) ^ (row >> 3)) & 7);
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

  // ---- epilogue: fp32 atomics into dW[oc][j]
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oc = oc0 + wm * WM + i * 16 + l4 * 4 + r;
        const int jj = j0 + wn * WN + j * 16 + l15;
        if (oc < p.NCOLS && jj < p.KDIM) atomicAdd(p.dw + (int64_t)oc * p.KDIM + jj, acc[i][j][r]);
      }
}

template <typename T, int BMo, int BNo>
int launch(const WParams& p, int splits, hipStream_t st) {
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

}  // namespace

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
  const int bkm = d->dtype == PASSL_BF16 ? 64 : 32;
  p.nk_total = (p.M + bkm - 1) / bkm;
  int splits = d->splits;
  if (splits > p.nk_total) splits = p.nk_total;
  p.tiles_per_split = (p.nk_total + splits - 1) / splits;
  splits = (p.nk_total + p.tiles_per_split - 1) / p.tiles_per_split;
  hipStream_t st = as_stream(stream);
  passl_prof_begin(1, st);
  const int rc = d->dtype == PASSL_BF16 ? dispatch<bf16_t>(p, splits, st) : dispatch<float>(p, splits, st);
  passl_prof_end(1, st);
  return rc;
}
