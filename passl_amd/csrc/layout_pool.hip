// Input layout conversion, pooling and column sums (all HBM-bound, NHWC, 16 B per lane).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// x fp32 [N,C,H,W] -> y [N,Hp,Wp,Cp]; one thread per output pixel (Cp <= 8 channels).
template <typename T>
__global__ void __launch_bounds__(kThreads) nchw_to_nhwc_pad_kernel(
    const float* __restrict__ x, T* __restrict__ y, int N, int C, int H, int W, int pad, int Hp,
    int Wp, int Cp) {
  const int64_t total = (int64_t)N * Hp * Wp;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int wp = (int)(i % Wp);
    const int hp = (int)((i / Wp) % Hp);
    const int n = (int)(i / ((int64_t)Wp * Hp));
    const int h = hp - pad, w = wp - pad;
    const bool in = (h >= 0) && (h < H) && (w >= 0) && (w < W);
    T* o = y + i * Cp;
    for (int c = 0; c < Cp; ++c) {
      float v = 0.f;
      if (in && c < C) v = x[(((int64_t)n * C + c) * H + h) * W + w];
      ElemTraits<T>::st(o + c, v);
    }
  }
}

// The stem's case (Cp = 4, bf16: an output pixel is 8 bytes): one thread per pixel, ONE 8-byte store instead of four
// 2-byte stores, 32-bit index arithmetic, no grid-stride loop (r05: 112 -> ~50 us per 256 x 3 x 224 x 224 batch).
__global__ void __launch_bounds__(kThreads) nchw_to_nhwc_pad4_bf16_kernel(
    const float* __restrict__ x, bf16_t* __restrict__ y, uint32_t total, int C, int H, int W, int pad, int Hp, int Wp) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int wp = (int)(i % (uint32_t)Wp);
  const uint32_t t = i / (uint32_t)Wp;
  const int hp = (int)(t % (uint32_t)Hp);
  const int n = (int)(t / (uint32_t)Hp);
  const int h = hp - pad, w = wp - pad;
  const bool in = (h >= 0) && (h < H) && (w >= 0) && (w < W);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (in) {
    const float* px = x + ((int64_t)n * C * H + h) * W + w;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) v[c] = px[(int64_t)c * H * W];
  }
  *reinterpret_cast<uint2*>(y + (int64_t)i * 4) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

// 3x3 s2 p1 max pool; one thread per (n,p,q, 8-channel chunk), one pass per thread (no grid-stride
// loop).  Branch-free: the 9 taps are loaded from CLAMPED coordinates back to back (9 x 16 B in flight
// per lane) and an out-of-range tap simply never wins.
template <typename T>
__global__ void __launch_bounds__(kThreads) maxpool_fwd_kernel(const T* __restrict__ x,
                                                               T* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int N,
                                                               int H, int W, int C, int P, int Q) {
  const uint32_t cc = (uint32_t)C >> 3;
  const uint32_t total = (uint32_t)N * P * Q * cc;       // < 2^31 (checked on the host): 32-bit index math
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cc);
  uint32_t t = i / cc;
  const int q = (int)(t % (uint32_t)Q); t /= (uint32_t)Q;
  const int p = (int)(t % (uint32_t)P);
  const int n = (int)(t / (uint32_t)P);
  float v[9][8];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      int h = p * 2 - 1 + r, w = q * 2 - 1 + s;
      h = h < 0 ? 0 : (h >= H ? H - 1 : h);
      w = w < 0 ? 0 : (w >= W ? W - 1 : w);
      ElemTraits<T>::load8(x + (((int64_t)n * H + h) * W + w) * C + c8 * 8, v[r * 3 + s]);
    }
  float best[8];
  int bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
  bool first = true;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int h = p * 2 - 1 + r, w = q * 2 - 1 + s;
      const bool valid = h >= 0 && h < H && w >= 0 && w < W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // first max wins (strictly greater replaces), NaN propagates like torch
        const float u = v[r * 3 + s][e];
        if (valid && (first || u > best[e] || u != u)) { best[e] = u; bi[e] = r * 3 + s; }
      }
      first = first && !valid;
    }
  const int64_t o = (((int64_t)n * P + p) * Q + q) * C + c8 * 8;
  ElemTraits<T>::store8(y + o, best);
  uint2 packed;
  packed.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  packed.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
  *reinterpret_cast<uint2*>(idx + o) = packed;
}

// gather-style backward, no atomics: one thread per 2 x 2 block of input pixels (rows 2p', 2p'+1,
// columns 2q', 2q'+1) x 8 channels.  Row 2p' lies only in window row p' (tap r = 1); row 2p'+1 in
// p' (r = 2) and p'+1 (r = 0); the same along w: the block needs the 2 x 2 windows (p'+a, q'+b), each
// loaded ONCE (index bytes + gradient, branch-free from clamped coordinates) for four outputs.  Every
// output adds its windows in the order (p0,q0), (p0,q1), (p1,q0), (p1,q1) = ascending (p, q).
template <typename T>
__global__ void __launch_bounds__(kThreads) maxpool_bwd_kernel(const T* __restrict__ dy,
                                                               const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int N, int H,
                                                               int W, int C, int P, int Q) {
  const uint32_t cc = (uint32_t)C >> 3;
  const uint32_t H2 = (uint32_t)(H + 1) >> 1, W2 = (uint32_t)(W + 1) >> 1;
  const uint32_t total = (uint32_t)N * H2 * W2 * cc;     // < 2^31 (checked on the host): 32-bit index math
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cc);
  uint32_t t = i / cc;
  const int q2 = (int)(t % W2); t /= W2;
  const int p2 = (int)(t % H2);
  const int n = (int)(t / H2);
  uint2 packed[2][2];
  float g[2][2][8];
  bool pv[2], qv[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    pv[a] = p2 + a < P;
    qv[a] = q2 + a < Q;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int pc = pv[a] ? p2 + a : P - 1, qc = qv[b] ? q2 + b : Q - 1;
      const int64_t o = (((int64_t)n * P + pc) * Q + qc) * C + c8 * 8;
      packed[a][b] = *reinterpret_cast<const uint2*>(idx + o);
      ElemTraits<T>::load8(dy + o, g[a][b]);
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      const int h = 2 * p2 + dh, w = 2 * q2 + dw;
      if (h >= H || w >= W) continue;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int a = 0; a <= dh; ++a)
#pragma unroll
        for (int b = 0; b <= dw; ++b) {
          const int r = dh == 0 ? 1 : (a == 0 ? 2 : 0);
          const int sx = dw == 0 ? 1 : (b == 0 ? 2 : 0);
          const int tap = r * 3 + sx;
          const bool valid = pv[a] && qv[b];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t word = e < 4 ? packed[a][b].x : packed[a][b].y;
            const int bsel = (int)((word >> (8 * (e & 3))) & 0xffu);
            if (valid && bsel == tap) acc[e] += g[a][b][e];
          }
        }
      ElemTraits<T>::store8(dx + (((int64_t)n * H + h) * W + w) * C + c8 * 8, acc);
    }
}

// y[n][c] = mean_hw x[n][hw][c]; one thread per (n, 8-channel chunk)
template <typename T>
__global__ void __launch_bounds__(kThreads) avgpool_fwd_kernel(const T* __restrict__ x,
                                                               T* __restrict__ y, int N, int HW,
                                                               int C) {
  const int cc = C >> 3;
  const int64_t total = (int64_t)N * cc;
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cc);
  const int n = (int)(i / cc);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = 0; p < HW; ++p) {
    float v[8];
    ElemTraits<T>::load8(x + ((int64_t)n * HW + p) * C + c8 * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
  const float inv = 1.0f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= inv;
  ElemTraits<T>::store8(y + (int64_t)n * C + c8 * 8, acc);
}

template <typename T>
__global__ void __launch_bounds__(kThreads) avgpool_bwd_kernel(const T* __restrict__ dy,
                                                               T* __restrict__ dx, int N, int HW,
                                                               int C) {
  const int cc = C >> 3;
  const int64_t total = (int64_t)N * HW * cc;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % cc);
    const int n = (int)(i / ((int64_t)cc * HW));
    float g[8];
    ElemTraits<T>::load8(dy + (int64_t)n * C + c8 * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] *= inv;
    ElemTraits<T>::store8(dx + i * 8, g);
  }
}

// column sums of x[M][C].  A block covers 32 consecutive 8-channel chunks (512 contiguous bytes per
// row in bf16) x one slab of rows: 32 chunk columns x 8 row lanes, coalesced row reads, LDS reduce over
// the row lanes.  mode 0: out[col] = s (single slab), 1: out[col] += s (single slab), 2: the slab's
// partial goes to ws[slab][C] (summed in slab order by slab_reduce_kernel), 3: fp32 atomics into out
// (no workspace given: order-dependent rounding).
template <typename T>
__global__ void __launch_bounds__(kThreads) colsum_kernel(const T* __restrict__ x,
                                                          float* __restrict__ out, int64_t M,
                                                          int C, int rows_per_block, int mode,
                                                          float* __restrict__ ws) {
  __shared__ float red[8][32 * 8 + 8];
  const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cc) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    for (int64_t m = r0 + rl; m < r1; m += 8) {
      float v[8];
      ElemTraits<T>::load8(x + m * C + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cc * 8 + e] = acc[e];
  __syncthreads();
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col < C) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) s += red[l][threadIdx.x];
    if (mode == 0) out[col] = s;
    else if (mode == 1) out[col] += s;
    else ws[(int64_t)blockIdx.y * C + col] = s;           // mode 2: slab, added in order afterwards
  }
}

// dx = dy * (y > 0)
template <typename T>
__global__ void __launch_bounds__(kThreads) relu_bwd_kernel(const T* __restrict__ dy,
                                                            const T* __restrict__ y,
                                                            T* __restrict__ dx, int64_t nchunks) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nchunks; i += stride) {
    float g[8], v[8];
    ElemTraits<T>::load8(dy + i * 8, g);
    ElemTraits<T>::load8(y + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = v[e] > 0.f ? g[e] : 0.f;
    ElemTraits<T>::store8(dx + i * 8, g);
  }
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define DISPATCH_DTYPE(dtype, ...)                          \
  if ((dtype) == PASSL_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == PASSL_F32) { using T = float; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

extern "C" int passl_hip_nchw_to_nhwc_pad(const float* x, void* y, int N, int C, int H, int W,
                                          int pad, int Wp, int Cp, int dtype,
                                          passl_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || pad < 0 || Wp < W + 2 * pad || Cp < C ||
      Cp > 8)
    return PASSL_EINVAL;
  const int Hp = H + 2 * pad;
  const int64_t total = (int64_t)N * Hp * Wp;
  if (dtype == PASSL_BF16 && Cp == 4 && C <= 4 && total <= 0x7fffffffll && (reinterpret_cast<uintptr_t>(y) & 7) == 0) {
    hipLaunchKernelGGL(nchw_to_nhwc_pad4_bf16_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads),
                       0, as_stream(stream), x, reinterpret_cast<bf16_t*>(y), (uint32_t)total, C, H, W, pad, Hp, Wp);
    PASSL_RETURN_IF_LAUNCH_FAILED();
    return PASSL_OK;
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel<T>, dim3(grid_for(total)),
                                           dim3(kThreads), 0, as_stream(stream), x,
                                           reinterpret_cast<T*>(y), N, C, H, W, pad, Hp, Wp, Cp);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H,
                                          int W, int C, int dtype, passl_stream_t stream) {
  if (!x || !y || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || !aligned16(x) ||
      !aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7))
    return PASSL_EINVAL;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * P * Q * (C >> 3);
  if (total > 0x7fffffffll) return PASSL_EUNSUPPORTED;       // the kernels index with 32 bits
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3((unsigned)((total + kThreads - 1) / kThreads)),
                                           dim3(kThreads), 0, as_stream(stream),
                                           reinterpret_cast<const T*>(x), reinterpret_cast<T*>(y),
                                           idx, N, H, W, C, P, Q);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N,
                                          int H, int W, int C, int dtype, passl_stream_t stream) {
  if (!dy || !dx || !idx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || !aligned16(dy) ||
      !aligned16(dx))
    return PASSL_EINVAL;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3);   // 2 x 2 input pixels per thread
  if (total > 0x7fffffffll) return PASSL_EUNSUPPORTED;       // the kernels index with 32 bits
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3((unsigned)((total + kThreads - 1) / kThreads)),
                                           dim3(kThreads), 0, as_stream(stream),
                                           reinterpret_cast<const T*>(dy), idx,
                                           reinterpret_cast<T*>(dx), N, H, W, C, P, Q);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype,
                                     passl_stream_t stream) {
  if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || !aligned16(x) || !aligned16(y))
    return PASSL_EINVAL;
  const int64_t total = (int64_t)N * (C >> 3);
  const int grid = (int)((total + kThreads - 1) / kThreads);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(grid), dim3(kThreads), 0,
                                           as_stream(stream), reinterpret_cast<const T*>(x),
                                           reinterpret_cast<T*>(y), N, HW, C);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype,
                                     passl_stream_t stream) {
  if (!dy || !dx || N <= 0 || HW <= 0 || C <= 0 || (C & 7) || !aligned16(dy) || !aligned16(dx))
    return PASSL_EINVAL;
  const int64_t total = (int64_t)N * HW * (C >> 3);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(grid_for(total)),
                                           dim3(kThreads), 0, as_stream(stream),
                                           reinterpret_cast<const T*>(dy),
                                           reinterpret_cast<T*>(dx), N, HW, C);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

int passl_slab_reduce_launch(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                             hipStream_t st);   // flat.hip

static int colsum_impl(const void* x, float* out, int64_t M, int C, int dtype, bool accumulate,
                       float* ws, int64_t ws_floats, passl_stream_t stream) {
  if (!x || !out || M <= 0 || C <= 0 || (C & 7) || !aligned16(x)) return PASSL_EINVAL;
  // ~64 rows per block: enough blocks in flight for the short-sequence ViT shapes (M = 6400 ... 50432);
  // 256-row slabs at M >= 32768
  const int per = M >= 32768 ? 256 : 64;
  int slabs = (int)((M + per - 1) / per);
  if (slabs > 1024) slabs = 1024;
  const int rows = (int)((M + slabs - 1) / slabs);
  slabs = (int)((M + rows - 1) / rows);              // every slab holds rows
  int mode;
  if (slabs == 1) mode = accumulate ? 1 : 0;
  else if (ws) {
    if (!aligned16(ws) || !aligned16(out) || ws_floats < (int64_t)slabs * C) return PASSL_EINVAL;
    mode = 2;
  } else {
    return PASSL_EINVAL;                                // several row slabs need the workspace (no fp32 atomics)
  }
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(colsum_kernel<T>, dim3((C + 255) / 256, slabs),
                                           dim3(kThreads), 0, as_stream(stream),
                                           reinterpret_cast<const T*>(x), out, M, C, rows, mode, ws);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  if (mode == 2) return passl_slab_reduce_launch(ws, out, C, slabs, accumulate ? 1 : 0, as_stream(stream));
  return PASSL_OK;
}

extern "C" int passl_hip_colsum(const void* x, float* out, int64_t M, int C, int dtype, float* ws,
                                int64_t ws_floats, passl_stream_t stream) {
  return colsum_impl(x, out, M, C, dtype, false, ws, ws_floats, stream);
}

extern "C" int passl_hip_colsum_acc(const void* x, float* out, int64_t M, int C, int dtype, float* ws,
                                    int64_t ws_floats, passl_stream_t stream) {
  return colsum_impl(x, out, M, C, dtype, true, ws, ws_floats, stream);
}

extern "C" int passl_hip_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype,
                                  passl_stream_t stream) {
  if (!dy || !y || !dx || n <= 0 || (n & 7) || !aligned16(dy) || !aligned16(y) || !aligned16(dx))
    return PASSL_EINVAL;
  const int64_t nchunks = n >> 3;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(grid_for(nchunks)),
                                           dim3(kThreads), 0, as_stream(stream),
                                           reinterpret_cast<const T*>(dy),
                                           reinterpret_cast<const T*>(y),
                                           reinterpret_cast<T*>(dx), nchunks);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
