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
#include <string.h>
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

// ------------------------------------------------------------------------------------------------
// Reduce pass, second form (default): the same arithmetic per element, organised for the machine.
//
// The first form above (a workgroup = 2048 items, every thread loads, WAITS, computes; ~710 VALU instructions per item,
// 159 registers + 41 spilled SGPRs) runs at 179 us alone = 3.2 TB/s of its 565 MB.  Here a workgroup walks over its
// share of the items with the 12 loads of item t + 1 in flight while item t is computed (two register sets of RAW
// words, unpacked where they are used), the per-channel coefficients sit in LDS (one ds_read_b128 per channel pair
// instead of 32 registers), offsets are 32-bit with multiply-high division, selects are branch-free (`a && b ? x : y`
// compiled to 44 exec-mask branches per item), and the per-thread sums are folded through LDS once per workgroup
// (slab rows = workgroups, not items / 2048): ~620 instructions per item, 175 registers, 117 us = 4.8 TB/s
// (`tools/kbench pooltime`, profiles/r05_kbench_pool.txt; 512 / 768 / 1024 / 1536 / 3136 workgroups: 127 / 118 / 117 /
// 122 / 129 us).  Every g is the first form's bit for bit: measured with an apply pass built on the same pair_grad
// (dx identical on every `kbench poolcheck` case, profiles/r05_kbench_pool.txt); the slab holds the same sums grouped
// differently (fp32 rounding; poolcheck compares the column sums of both forms).
//
// The APPLY pass stays in the first form: the same treatment measured 197-219 us against 205 — 976 MB with 42 % writes
// is at the memory system's mixed read / write rate (4.8 TB/s) either way.
namespace v2 {

struct FDiv { uint32_t mul, sh1, sh2; };                 // division by a run-time constant (halo_geom.h: same scheme)
FDiv make_fdiv(uint32_t d) {
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
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FDiv f) {
  const uint32_t t = __umulhi(f.mul, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct Geo {
  int H, W, C, P, Q;
  uint32_t H2, W2, cshift, total;      // 2 x 2 blocks per column / row, log2(C / 8), items = N * H2 * W2 * (C / 8)
  uint32_t chunks, per_wg;             // 256-item chunks in all and per workgroup
  FDiv dW2, dH2;
};

template <typename T> constexpr int kWords = 2 * (int)sizeof(T);      // 32-bit words of 8 channels: 4 (bf16), 8 (fp32)

template <typename T>
__device__ __forceinline__ void load_raw(const T* p, uint32_t (&w)[kWords<T>]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  if constexpr (sizeof(T) == 4) {
    const uint4 b = *(reinterpret_cast<const uint4*>(p) + 1);
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  }
}
template <typename T>
__device__ __forceinline__ float elem(const uint32_t (&w)[kWords<T>], int e) {
  if constexpr (sizeof(T) == 2) return __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
  else return __uint_as_float(w[e]);
}

// what a thread holds of one item (a 2 x 2 block of input pixels x 8 channels), as loaded
template <typename T>
struct Item {
  uint32_t gw[2][2][kWords<T>];        // output gradients of the windows (p2 + a, q2 + b)
  uint32_t ix[2][2][2];                // their arg-max bytes
  uint32_t xv[2][2][kWords<T>];        // y of the pixels (2 p2 + dh, 2 q2 + dw)
  uint32_t xb;                         // element offset of pixel (0, 0)
  uint32_t flags;                      // bit a * 2 + b: window inside the output; bit 4 + dh * 2 + dw: pixel inside the input
};

template <typename T>
__device__ __forceinline__ const T* at(const T* base, uint32_t elem_off) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(elem_off * (uint32_t)sizeof(T)));
}

// Element offsets are 32-bit (the host takes this form for tensors below 2^30 elements only).  A window / pixel beyond
// the edge re-reads its in-range neighbour (any mapped address will do) and is masked by its flag.
template <typename T>
__device__ __forceinline__ void load_item(Item<T>& it, const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                          const T* __restrict__ x, const Geo& g, uint32_t i, int c8) {
  const bool live = i < g.total;
  const uint32_t pix = (live ? i : g.total - 1) >> g.cshift;
  const uint32_t t1 = fdiv(pix, g.dW2);
  const uint32_t q2 = pix - t1 * g.W2;
  const uint32_t n = fdiv(t1, g.dH2);
  const uint32_t p2 = t1 - n * g.H2;
  const uint32_t C = (uint32_t)g.C;
  const bool pv = p2 + 1 < (uint32_t)g.P, qv = q2 + 1 < (uint32_t)g.Q;
  const bool hv = 2 * p2 + 1 < (uint32_t)g.H, wv = 2 * q2 + 1 < (uint32_t)g.W;
  const uint32_t ob = ((n * (uint32_t)g.P + p2) * (uint32_t)g.Q + q2) * C + (uint32_t)c8 * 8u;
  const uint32_t oa = pv ? (uint32_t)g.Q * C : 0u, oq = qv ? C : 0u;
  const uint32_t xb = ((n * (uint32_t)g.H + 2 * p2) * (uint32_t)g.W + 2 * q2) * C + (uint32_t)c8 * 8u;
  const uint32_t xa = hv ? (uint32_t)g.W * C : 0u, xq = wv ? C : 0u;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const uint32_t o = ob + (a ? oa : 0u) + (b ? oq : 0u);
      load_raw<T>(at<T>(dy, o), it.gw[a][b]);
      const uint2 t = *reinterpret_cast<const uint2*>(idx + (size_t)o);
      it.ix[a][b][0] = t.x; it.ix[a][b][1] = t.y;
    }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) load_raw<T>(at<T>(x, xb + (dh ? xa : 0u) + (dw ? xq : 0u)), it.xv[dh][dw]);
  it.xb = xb;
  const uint32_t l = live ? 1u : 0u, p1 = pv ? l : 0u, q1 = qv ? l : 0u, h1 = hv ? l : 0u, w1 = wv ? l : 0u;
  it.flags = l | (q1 << 1) | (p1 << 2) | ((p1 & q1) << 3) | (l << 4) | (w1 << 5) | (h1 << 6) | ((h1 & w1) << 7);
}

// channels 2 k, 2 k + 1 of the item: g (the masked, rounded pool gradient) and y of its four pixels
template <typename T>
__device__ __forceinline__ void pair_grad(const Item<T>& it, const uint32_t (&ix)[2][2][2], int k, const float (&sc)[2],
                                          const float (&sh)[2], float (&g)[2][2][2], float (&xf)[2][2][2]) {
  float w[2][2][2];
  uint32_t t[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        w[a][b][j] = elem<T>(it.gw[a][b], 2 * k + j);
        t[a][b][j] = __builtin_amdgcn_ubfe(ix[a][b][k >> 1], (uint32_t)(16 * (k & 1) + 8 * j), 8u);
      }
#define PASSL_SEL(a, b, tap) (t[a][b][j] == (tap) ? w[a][b][j] : 0.f)
  float d[2][2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    // block_grad's sums: windows in ascending (a, b); tap = r * 3 + s of the pixel inside the window
    d[0][0][j] = 0.f + PASSL_SEL(0, 0, 4u);
    d[0][1][j] = (0.f + PASSL_SEL(0, 0, 5u)) + PASSL_SEL(0, 1, 3u);
    d[1][0][j] = (0.f + PASSL_SEL(0, 0, 7u)) + PASSL_SEL(1, 0, 1u);
    d[1][1][j] = (((0.f + PASSL_SEL(0, 0, 8u)) + PASSL_SEL(0, 1, 6u)) + PASSL_SEL(1, 0, 2u)) + PASSL_SEL(1, 1, 0u);
  }
#undef PASSL_SEL
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      if constexpr (sizeof(T) == 2) {
        if (dh | dw) {                              // (0, 0) has one window: its value is a T already
          const uint32_t pk = pack2bf(d[dh][dw][0], d[dh][dw][1]);
          d[dh][dw][0] = __uint_as_float(pk << 16);
          d[dh][dw][1] = __uint_as_float(pk & 0xffff0000u);
        }
      }
      // (a pixel outside the input is no window's arg-max — the forward pass never selects an invalid tap — so its
      // d is zero without looking at the pixel's flag)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        xf[dh][dw][j] = elem<T>(it.xv[dh][dw], 2 * k + j);
        g[dh][dw][j] = (xf[dh][dw][j] * sc[j] + sh[j]) > 0.f ? d[dh][dw][j] : 0.f;
      }
    }
}

template <typename T>
__device__ __forceinline__ void masked_idx(const Item<T>& it, uint32_t (&ix)[2][2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const bool v = (it.flags >> (a * 2 + b)) & 1u;      // a window outside the output matches no tap
      ix[a][b][0] = v ? it.ix[a][b][0] : 0xffffffffu;
      ix[a][b][1] = v ? it.ix[a][b][1] : 0xffffffffu;
    }
}

// LDS table of per-channel coefficients: tab[(c / 2) * NV + v] = float4 of vector v for channels c, c + 1
template <int NV>
__device__ __forceinline__ void fill_table(float4* tab, int C, const float* const (&src)[NV * 2]) {
  for (int c2 = threadIdx.x; c2 < C / 2; c2 += kThreads)
#pragma unroll
    for (int v = 0; v < NV; ++v)
      tab[c2 * NV + v] = make_float4(src[2 * v][2 * c2], src[2 * v][2 * c2 + 1], src[2 * v + 1][2 * c2], src[2 * v + 1][2 * c2 + 1]);
  __syncthreads();
}

template <typename T>
__global__ void __launch_bounds__(kThreads, 2) bwd_reduce_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                              const T* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              float* __restrict__ partial, const Geo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tab = reinterpret_cast<float4*>(smem);                      // [C / 2][2]: (sc, sc, sh, sh), (mu, mu, is, is)
  {
    const float* const src[4] = {scale, shift, mean, invstd};
    fill_table<2>(tab, g.C, src);
  }
  const int c8 = (int)(threadIdx.x & ((1u << g.cshift) - 1u));
  float a0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint32_t c_begin = blockIdx.x * g.per_wg;
  uint32_t c_end = c_begin + g.per_wg;
  if (c_end > g.chunks) c_end = g.chunks;
  auto consume = [&](const Item<T>& it) {
    uint32_t ix[2][2][2];
    masked_idx<T>(it, ix);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 t0 = tab[(c8 * 4 + k) * 2], t1 = tab[(c8 * 4 + k) * 2 + 1];
      const float sc[2] = {t0.x, t0.y}, sh[2] = {t0.z, t0.w}, mu[2] = {t1.x, t1.y}, is[2] = {t1.z, t1.w};
      float gr[2][2][2], xf[2][2][2];
      pair_grad<T>(it, ix, k, sc, sh, gr, xf);
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            a0[2 * k + j] += gr[dh][dw][j];
            a1[2 * k + j] += gr[dh][dw][j] * (xf[dh][dw][j] - mu[j]) * is[j];
          }
    }
  };
  if (c_begin < c_end) {
    Item<T> A, B;
    load_item<T>(A, dy, idx, x, g, c_begin * kThreads + threadIdx.x, c8);
    for (uint32_t c = c_begin; c < c_end; c += 2) {
      const bool more = c + 1 < c_end;
      if (more) load_item<T>(B, dy, idx, x, g, (c + 1) * kThreads + threadIdx.x, c8);
      consume(A);
      if (more) {
        if (c + 2 < c_end) load_item<T>(A, dy, idx, x, g, (c + 2) * kThreads + threadIdx.x, c8);
        consume(B);
      }
    }
  }
  // fold the threads of a channel chunk in thread order (a pixel outside the input contributed zeros)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                         // [256][16] (the table is dead)
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = a0[e]; red[threadIdx.x * 16 + 8 + e] = a1[e]; }
  __syncthreads();
  const uint32_t cc = 1u << g.cshift;
  if (threadIdx.x < cc) {
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t l = 0; l < kThreads / cc; ++l) {
      const uint32_t t = l * cc + threadIdx.x;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] += red[t * 16 + e]; s1[e] += red[t * 16 + 8 + e]; }
    }
    float* o = partial + ((int64_t)blockIdx.x * g.C + threadIdx.x * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) *reinterpret_cast<float4*>(o + e * 2) = make_float4(s0[e], s1[e], s0[e + 1], s1[e + 1]);
  }
}

}  // namespace v2

// 1 (default): the second form of the reduce pass, 0: the first (the reference `kbench poolcheck` compares against,
// and the form of tensors of 2^30 elements and more).  g_pool_wgs: workgroups of a second-form launch (2 per CU are
// resident; 1024 = two rounds measured best).
int g_pool_form = 1;
int g_pool_wgs = 1024;

bool shape_ok(int N, int H, int W, int C, int64_t* out_items, int64_t* in_items) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (kThreads % (C >> 3)) != 0) return false;
  const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
  *out_items = (int64_t)N * P * Q * (C >> 3);
  *in_items = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3);
  return *out_items <= 0x7fffffffll && *in_items <= 0x7fffffffll && (int64_t)N * H * W * C <= 0x7fffffff0ll;
}

v2::Geo make_geo(int N, int H, int W, int C, int64_t in_items) {
  v2::Geo g;
  g.H = H; g.W = W; g.C = C;
  g.P = (H + 2 - 3) / 2 + 1; g.Q = (W + 2 - 3) / 2 + 1;
  g.H2 = (uint32_t)(H + 1) >> 1; g.W2 = (uint32_t)(W + 1) >> 1;
  g.cshift = 0;
  while ((8u << g.cshift) < (uint32_t)C) ++g.cshift;
  g.total = (uint32_t)in_items;
  g.chunks = (uint32_t)((in_items + kThreads - 1) / kThreads);
  const uint32_t wgs = (uint32_t)(g_pool_wgs > 0 ? g_pool_wgs : 1);
  g.per_wg = (g.chunks + wgs - 1) / wgs;
  g.dW2 = v2::make_fdiv(g.W2); g.dH2 = v2::make_fdiv(g.H2);
  (void)N;
  return g;
}
inline int geo_blocks(const v2::Geo& g) { return (int)((g.chunks + g.per_wg - 1) / g.per_wg); }
// the second form addresses with 32-bit element offsets
inline bool second_form(int N, int H, int W, int C) { return g_pool_form == 1 && (int64_t)N * H * W * C < (1ll << 30); }

}  // namespace

int passl_pool_option(const char* name, int value) {
  if (strcmp(name, "stem_pool_form") == 0) {
    if (value != 0 && value != 1) return PASSL_EINVAL;
    g_pool_form = value;
    return PASSL_OK;
  }
  if (strcmp(name, "stem_pool_wgs") == 0) {
    if (value < 1 || value > 65536) return PASSL_EINVAL;
    g_pool_wgs = value;
    return PASSL_OK;
  }
  return PASSL_EINVAL;
}

#define DISPATCH_DTYPE(dtype, ...)                          \
  if ((dtype) == PASSL_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == PASSL_F32) { using T = float; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

extern "C" int passl_hip_bn_relu_maxpool_blocks(int N, int H, int W, int C) {
  int64_t oi, ii;
  if (!shape_ok(N, H, W, C, &oi, &ii)) return 0;
  if (second_form(N, H, W, C)) return geo_blocks(make_geo(N, H, W, C, ii));
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
  if (second_form(N, H, W, C)) {
    const v2::Geo g = make_geo(N, H, W, C, ii);
    const unsigned lds = (unsigned)(C * 16 > 16384 ? C * 16 : 16384);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(v2::bwd_reduce_kernel<T>, dim3((unsigned)nblocks), dim3(kThreads), lds,
                                             as_stream(stream), reinterpret_cast<const T*>(dy), idx,
                                             reinterpret_cast<const T*>(x), mean, invstd, scale, shift, partial, g);)
    PASSL_RETURN_IF_LAUNCH_FAILED();
    return PASSL_OK;
  }
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
