// ResNet stem: training-mode BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max-pool as ONE pass, forward and backward.
//
// Unfused, the stem of a 224^2 batch of 256 moves the [N,112,112,64] tensor seven times after the convolution:
// bn_apply (read y, write z), max-pool (read z), and backward max-pool (write dz), bn_bwd_reduce (read dz, y),
// bn_bwd_apply (read dz, y, write dy) — 2.9 GB, and the three backward passes are the LAST thing on the step's main
// chain (profiles/r05_trace_chain_*.txt: 195 + 299 + 239 us with nothing left to overlap).  z and dz exist only
// between two passes of this file, so they are never written:
//
//   forward   out[n,p,q,c] = max over the window of z,  z = T(relu(y * scale + shift))       reads y, writes out + idx
//   backward  dz[n,h,w,c]  = T(sum of dout over the windows whose arg-max is (h, w))         (idx: 1 byte per output)
//             g = dz where y * scale + shift > 0, else 0
//     reduce  partial[b][c] = (sum g, sum g * (y - mean) * invstd)     -> passl_hip_bn_bwd_finalize -> coef
//     apply   dy = A g + B y + C                                                              writes dy
//
// The arithmetic of every element is the unfused kernels' (bn.hip: bn_apply_tile_kernel, bn_reduce_kernel<.,1>,
// bn_bwd_apply_tile_kernel; layout_pool.hip: maxpool_fwd / maxpool_bwd): the same expressions, z and dz rounded to
// the storage type T exactly where the unfused path stores them, first maximum wins in (r, s) order, windows added in
// ascending (p, q).  out, idx and dy therefore equal the unfused path's bit for bit given the same coefficients; the
// per-channel sums are added in a different (fixed) order, i.e. agree to fp32 rounding.
//
// Replaces paddle.nn.BatchNorm2D + ReLU + MaxPool2D(3, 2, 1) of the stem (resnetimagenet.py:196-198) in training mode.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;              // 2 x 2 input blocks per thread of the reduce pass: slab rows = items / 2048

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <typename T>
__global__ void __launch_bounds__(kThreads) bn_relu_maxpool_fwd_kernel(const T* __restrict__ x,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift,
                                                                       T* __restrict__ y, uint8_t* __restrict__ idx,
                                                                       int N, int H, int W, int C, int P, int Q) {
  const uint32_t cc = (uint32_t)C >> 3;
  const uint32_t total = (uint32_t)N * P * Q * cc;       // < 2^31 (checked on the host): 32-bit index math
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cc);
  uint32_t t = i / cc;
  const int q = (int)(t % (uint32_t)Q); t /= (uint32_t)Q;
  const int p = (int)(t % (uint32_t)P);
  const int n = (int)(t / (uint32_t)P);
  float sc[8], sh[8];
  {
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c8 * 8), s1 = *reinterpret_cast<const float4*>(scale + c8 * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + c8 * 8), b1 = *reinterpret_cast<const float4*>(shift + c8 * 8 + 4);
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
  }
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
        // z as bn_apply stores it (affine, ReLU, rounded to T); first max wins, NaN propagates (maxpool_fwd_kernel)
        const float u = round_to<T>(fmaxf(v[r * 3 + s][e] * sc[e] + sh[e], 0.f));
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

// g of one 2 x 2 block of input pixels (rows 2 p2, 2 p2 + 1, columns 2 q2, 2 q2 + 1) x 8 channels, and the block's
// y values: maxpool_bwd_kernel's gather (each of the 2 x 2 windows (p2 + a, q2 + b) loaded once; an output adds its
// windows in ascending (p, q)), the sum rounded to T (= the dz the unfused path stores), then the ReLU mask
// recomputed from y (bn.hip: apply_relu_mask, relu = 2).
template <typename T>
__device__ __forceinline__ void block_grad(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                           const T* __restrict__ x, const float (&sc)[8], const float (&sh)[8], int n,
                                           int p2, int q2, int c8, int H, int W, int C, int P, int Q,
                                           float (&g)[2][2][8], float (&xv)[2][2][8], bool (&ok)[2][2]) {
  uint2 packed[2][2];
  float gw[2][2][8];
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
      ElemTraits<T>::load8(dy + o, gw[a][b]);
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      const int h = 2 * p2 + dh, w = 2 * q2 + dw;
      ok[dh][dw] = h < H && w < W;
      const int hc = h < H ? h : H - 1, wc = w < W ? w : W - 1;
      ElemTraits<T>::load8(x + (((int64_t)n * H + hc) * W + wc) * C + c8 * 8, xv[dh][dw]);
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
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
            if (valid && bsel == tap) acc[e] += gw[a][b][e];
          }
        }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dz = round_to<T>(acc[e]);
        g[dh][dw][e] = (xv[dh][dw][e] * sc[e] + sh[e]) > 0.f ? dz : 0.f;
      }
    }
}

__device__ __forceinline__ void load_cols(const float* __restrict__ p, int c8, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p + c8 * 8), b = *reinterpret_cast<const float4*>(p + c8 * 8 + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// partial[b][c][0..1] = sum g, sum g * (y - mean) * invstd over the kUnroll * 256 items of block b (item = one 2 x 2
// pixel block x 8 channels; a thread keeps its channel chunk: 256 % (C / 8) == 0)
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_relu_maxpool_bwd_reduce_kernel(
    const T* __restrict__ dy, const uint8_t* __restrict__ idx, const T* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift,
    float* __restrict__ partial, int N, int H, int W, int C, int P, int Q) {
  __shared__ float red[kThreads][16];
  const uint32_t cc = (uint32_t)C >> 3;
  const uint32_t H2 = (uint32_t)(H + 1) >> 1, W2 = (uint32_t)(W + 1) >> 1;
  const uint32_t total = (uint32_t)N * H2 * W2 * cc;
  const int c8 = (int)(threadIdx.x % cc);
  float sc[8], sh[8], mu[8], is[8];
  load_cols(scale, c8, sc); load_cols(shift, c8, sh); load_cols(mean, c8, mu); load_cols(invstd, c8, is);
  float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int u = 0; u < kUnroll; ++u) {
    const uint32_t i = (blockIdx.x * kUnroll + u) * kThreads + threadIdx.x;
    if (i >= total) break;
    uint32_t t = i / cc;
    const int q2 = (int)(t % W2); t /= W2;
    const int p2 = (int)(t % H2);
    const int n = (int)(t / H2);
    float g[2][2][8], xv[2][2][8];
    bool ok[2][2];
    block_grad<T>(dy, idx, x, sc, sh, n, p2, q2, c8, H, W, C, P, Q, g, xv, ok);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        if (!ok[dh][dw]) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a0[e] += g[dh][dw][e];
          a1[e] += g[dh][dw][e] * (xv[dh][dw][e] - mu[e]) * is[e];
        }
      }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = a0[e]; red[threadIdx.x][8 + e] = a1[e]; }
  __syncthreads();
  if (threadIdx.x < cc) {                  // column chunk threadIdx.x: the threads t = l * cc + threadIdx.x, in order
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t l = 0; l < kThreads / cc; ++l) {
      const uint32_t t = l * cc + threadIdx.x;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] += red[t][e]; s1[e] += red[t][8 + e]; }
    }
    float* o = partial + ((int64_t)blockIdx.x * C + threadIdx.x * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) *reinterpret_cast<float4*>(o + e * 2) = make_float4(s0[e], s1[e], s0[e + 1], s1[e + 1]);
  }
}

// dy = A g + B y + C (bn_bwd_apply_tile_kernel's expression), one thread per 2 x 2 pixel block x 8 channels
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_relu_maxpool_bwd_apply_kernel(
    const T* __restrict__ dy, const uint8_t* __restrict__ idx, const T* __restrict__ x, const float* __restrict__ coef,
    const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ dx, int N, int H, int W, int C,
    int P, int Q) {
  const uint32_t cc = (uint32_t)C >> 3;
  const uint32_t H2 = (uint32_t)(H + 1) >> 1, W2 = (uint32_t)(W + 1) >> 1;
  const uint32_t total = (uint32_t)N * H2 * W2 * cc;
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % cc);
  uint32_t t = i / cc;
  const int q2 = (int)(t % W2); t /= W2;
  const int p2 = (int)(t % H2);
  const int n = (int)(t / H2);
  float sc[8], sh[8], cA[8], cB[8], cC[8];
  load_cols(scale, c8, sc); load_cols(shift, c8, sh);
  load_cols(coef, c8, cA); load_cols(coef + C, c8, cB); load_cols(coef + 2 * C, c8, cC);
  float g[2][2][8], xv[2][2][8];
  bool ok[2][2];
  block_grad<T>(dy, idx, x, sc, sh, n, p2, q2, c8, H, W, C, P, Q, g, xv, ok);
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      if (!ok[dh][dw]) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = cA[e] * g[dh][dw][e] + cB[e] * xv[dh][dw][e] + cC[e];
      ElemTraits<T>::store8(dx + (((int64_t)n * H + 2 * p2 + dh) * W + 2 * q2 + dw) * C + c8 * 8, o);
    }
}

bool shape_ok(int N, int H, int W, int C, int64_t* out_items, int64_t* in_items) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (kThreads % (C >> 3)) != 0) return false;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  *out_items = (int64_t)N * P * Q * (C >> 3);
  *in_items = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3);
  return *out_items <= 0x7fffffffll && *in_items <= 0x7fffffffll && (int64_t)N * H * W * C <= 0x7fffffff0ll;
}

}  // namespace

#define DISPATCH_DTYPE(dtype, ...)                          \
  if ((dtype) == PASSL_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == PASSL_F32) { using T = float; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

extern "C" int passl_hip_bn_relu_maxpool_blocks(int N, int H, int W, int C) {
  int64_t oi, ii;
  if (!shape_ok(N, H, W, C, &oi, &ii)) return 0;
  return (int)((ii + (int64_t)kThreads * kUnroll - 1) / ((int64_t)kThreads * kUnroll));
}

extern "C" int passl_hip_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y,
                                             uint8_t* idx, int N, int H, int W, int C, int dtype,
                                             passl_stream_t stream) {
  int64_t oi, ii;
  if (!x || !scale || !shift || !y || !idx || !aligned16(x) || !aligned16(y) || !aligned16(scale) || !aligned16(shift) ||
      (reinterpret_cast<uintptr_t>(idx) & 7))
    return PASSL_EINVAL;
  if (!shape_ok(N, H, W, C, &oi, &ii)) return PASSL_EUNSUPPORTED;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<T>, dim3((unsigned)((oi + kThreads - 1) / kThreads)),
                                           dim3(kThreads), 0, as_stream(stream), reinterpret_cast<const T*>(x), scale,
                                           shift, reinterpret_cast<T*>(y), idx, N, H, W, C, P, Q);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_relu_maxpool_bwd_reduce(const void* dy, const uint8_t* idx, const void* x, const float* mean,
                                                    const float* invstd, const float* scale, const float* shift,
                                                    float* partial, int nblocks, int N, int H, int W, int C, int dtype,
                                                    passl_stream_t stream) {
  int64_t oi, ii;
  if (!dy || !idx || !x || !mean || !invstd || !scale || !shift || !partial || !aligned16(dy) || !aligned16(x) ||
      !aligned16(mean) || !aligned16(invstd) || !aligned16(scale) || !aligned16(shift) || !aligned16(partial) ||
      (reinterpret_cast<uintptr_t>(idx) & 7))
    return PASSL_EINVAL;
  if (!shape_ok(N, H, W, C, &oi, &ii)) return PASSL_EUNSUPPORTED;
  if (nblocks != passl_hip_bn_relu_maxpool_blocks(N, H, W, C)) return PASSL_EINVAL;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_relu_maxpool_bwd_reduce_kernel<T>, dim3((unsigned)nblocks), dim3(kThreads), 0,
                                           as_stream(stream), reinterpret_cast<const T*>(dy), idx,
                                           reinterpret_cast<const T*>(x), mean, invstd, scale, shift, partial, N, H, W, C,
                                           P, Q);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_relu_maxpool_bwd_apply(const void* dy, const uint8_t* idx, const void* x, const float* coef,
                                                   const float* scale, const float* shift, void* dx, int N, int H, int W,
                                                   int C, int dtype, passl_stream_t stream) {
  int64_t oi, ii;
  if (!dy || !idx || !x || !coef || !scale || !shift || !dx || !aligned16(dy) || !aligned16(x) || !aligned16(dx) ||
      !aligned16(coef) || !aligned16(scale) || !aligned16(shift) || (reinterpret_cast<uintptr_t>(idx) & 7))
    return PASSL_EINVAL;
  if (!shape_ok(N, H, W, C, &oi, &ii)) return PASSL_EUNSUPPORTED;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_kernel<T>, dim3((unsigned)((ii + kThreads - 1) / kThreads)),
                                           dim3(kThreads), 0, as_stream(stream), reinterpret_cast<const T*>(dy), idx,
                                           reinterpret_cast<const T*>(x), coef, scale, shift, reinterpret_cast<T*>(dx), N,
                                           H, W, C, P, Q);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
