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

// 3x3 s2 p1 max pool; one thread per (n,p,q, 8-channel chunk).
template <typename T>
__global__ void __launch_bounds__(kThreads) maxpool_fwd_kernel(const T* __restrict__ x,
                                                               T* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int N,
                                                               int H, int W, int C, int P, int Q) {
  const int cc = C >> 3;
  const int64_t total = (int64_t)N * P * Q * cc;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % cc);
    int64_t t = i / cc;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P);
    const int n = (int)(t / P);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      const int h = p * 2 - 1 + r;
      if (h < 0 || h >= H) continue;
      for (int s = 0; s < 3; ++s) {
        const int w = q * 2 - 1 + s;
        if (w < 0 || w >= W) continue;
        float v[8];
        ElemTraits<T>::load8(x + (((int64_t)n * H + h) * W + w) * C + c8 * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          // first max wins (strictly greater replaces), NaN propagates like torch
          if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
        }
        first = false;
      }
    }
    const int64_t o = (((int64_t)n * P + p) * Q + q) * C + c8 * 8;
    ElemTraits<T>::store8(y + o, best);
    uint2 packed;
    packed.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    packed.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + o) = packed;
  }
}

// gather-style backward: one thread per input (n,h,w, 8-channel chunk); no atomics.
template <typename T>
__global__ void __launch_bounds__(kThreads) maxpool_bwd_kernel(const T* __restrict__ dy,
                                                               const uint8_t* __restrict__ idx,
                                                               T* __restrict__ dx, int N, int H,
                                                               int W, int C, int P, int Q) {
  const int cc = C >> 3;
  const int64_t total = (int64_t)N * H * W * cc;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % cc);
    int64_t t = i / cc;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows (p,q) with p*2-1+r == h  ->  r = h+1-2p in [0,3)
    for (int p = h / 2; p <= (h + 1) / 2 && p < P; ++p) {
      const int r = h + 1 - 2 * p;
      if (r < 0 || r > 2) continue;
      for (int q = w / 2; q <= (w + 1) / 2 && q < Q; ++q) {
        const int s = w + 1 - 2 * q;
        if (s < 0 || s > 2) continue;
        const int64_t o = (((int64_t)n * P + p) * Q + q) * C + c8 * 8;
        const uint2 packed = *reinterpret_cast<const uint2*>(idx + o);
        float g[8];
        ElemTraits<T>::load8(dy + o, g);
        const int tap = r * 3 + s;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t word = e < 4 ? packed.x : packed.y;
          const int b = (int)((word >> (8 * (e & 3))) & 0xffu);
          if (b == tap) acc[e] += g[e];
        }
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
    else if (mode == 2) ws[(int64_t)blockIdx.y * C + col] = s;
    else atomicAdd(out + col, s);
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
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total)),
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
  const int64_t total = (int64_t)N * H * W * (C >> 3);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total)),
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
    mode = 3;
    if (!accumulate &&
        hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, as_stream(stream)) != hipSuccess)
      return PASSL_ELAUNCH;
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
