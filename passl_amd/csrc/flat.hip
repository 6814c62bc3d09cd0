// Flat-buffer (multi-tensor) kernels: key-encoder EMA, momentum-SGD, casts, weight packing.
// All are HBM-bound streaming kernels: 16-byte accesses per lane, grid-stride, ~2048 blocks.
#include "common.h"

namespace {

constexpr int kThreads = 256;

static inline int grid_for(int64_t n_vec) {
  int64_t b = (n_vec + kThreads - 1) / kThreads;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// k = k*m + q*(1-m); optional bf16 copy.   12 (+2) bytes per element.
template <bool LP>
__global__ void __launch_bounds__(kThreads) ema_kernel(float* __restrict__ k,
                                                       const float* __restrict__ q,
                                                       bf16_t* __restrict__ klp, int64_t n,
                                                       float m) {
  const float om = 1.0f - m;
  const int64_t nv = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += stride) {
    float4 kv = reinterpret_cast<float4*>(k)[i];
    const float4 qv = reinterpret_cast<const float4*>(q)[i];
    kv.x = kv.x * m + qv.x * om;
    kv.y = kv.y * m + qv.y * om;
    kv.z = kv.z * m + qv.z * om;
    kv.w = kv.w * m + qv.w * om;
    reinterpret_cast<float4*>(k)[i] = kv;
    if (LP) reinterpret_cast<uint2*>(klp)[i] = make_uint2(pack2bf(kv.x, kv.y), pack2bf(kv.z, kv.w));
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nv << 2) + threadIdx.x;
    const float v = k[i] * m + q[i] * om;
    k[i] = v;
    if (LP) klp[i] = f2bf(v);
  }
}

// out[i] (+)= sum_{z < slabs} ws[z*n + i], summed in a FIXED order: the deterministic replacement of
// "every producer atomically adds its partial tile".  A block covers 64 consecutive float4 columns
// (1 KB per slab row) x ZL slab lanes; lane zl adds slabs zl, zl+ZL, ... in ascending order, the
// ZL partial sums are combined through LDS in lane order.  n % 4 == 0.
template <int ZL>
__global__ void __launch_bounds__(kThreads) slab_reduce_kernel(const float* __restrict__ ws,
                                                               float* __restrict__ out, int64_t n4,
                                                               int slabs, int accumulate) {
  constexpr int COLS = kThreads / ZL;
  __shared__ float4 red[ZL][COLS];
  const int col = threadIdx.x % COLS, zl = threadIdx.x / COLS;
  const int64_t i = (int64_t)blockIdx.x * COLS + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    for (int z = zl; z < slabs; z += ZL) {
      const float4 v = reinterpret_cast<const float4*>(ws)[(int64_t)z * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (ZL > 1) {
    red[zl][col] = a;
    __syncthreads();
    if (zl != 0) return;
#pragma unroll
    for (int l = 1; l < ZL; ++l) {
      const float4 v = red[l][col];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  if (i < n4) {
    float4* o = reinterpret_cast<float4*>(out) + i;
    if (accumulate) {
      const float4 v = *o;
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *o = a;
  }
}

// g' = g*gs + wd*p; v = mu*v + g'; p -= lr*v.   20 bytes per element.
__global__ void __launch_bounds__(kThreads) sgd_kernel(float* __restrict__ p,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ v, int64_t n, float lr_val,
                                                       const float* __restrict__ hyper, float mu, float wd,
                                                       float gs) {
  const float lr = hyper ? hyper[0] : lr_val;       // device-resident learning rate: HIP-graph replays (passl_hip.h)
  const int64_t nv = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    vv.x = mu * vv.x + (gv.x * gs + wd * pv.x);
    vv.y = mu * vv.y + (gv.y * gs + wd * pv.y);
    vv.z = mu * vv.z + (gv.z * gs + wd * pv.z);
    vv.w = mu * vv.w + (gv.w * gs + wd * pv.w);
    pv.x -= lr * vv.x;
    pv.y -= lr * vv.y;
    pv.z -= lr * vv.z;
    pv.w -= lr * vv.w;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nv << 2) + threadIdx.x;
    const float vv = mu * v[i] + (g[i] * gs + wd * p[i]);
    v[i] = vv;
    p[i] -= lr * vv;
  }
}

__global__ void __launch_bounds__(kThreads) cast_kernel(const float* __restrict__ s,
                                                        bf16_t* __restrict__ d, int64_t n) {
  const int64_t nv = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(s)[2 * i];
    const float4 b = reinterpret_cast<const float4*>(s)[2 * i + 1];
    reinterpret_cast<uint4*>(d)[i] =
        make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const int64_t i = (nv << 3) + threadIdx.x;
    d[i] = f2bf(s[i]);
  }
}

// ---- LARS (multi-tensor): one block = one chunk (<= kLarsChunk elements) of ONE parameter tensor.
constexpr int kLarsChunk = 4096;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial[block] = (sum p^2, sum (g*gs)^2) over the block's chunk.  No atomics: the per-parameter
// norms must be BIT-IDENTICAL on every data-parallel rank (identical inputs after the gradient
// all-reduce), otherwise the replicas drift apart through the local learning rates.
__global__ void __launch_bounds__(kThreads) lars_norm_kernel(
    const float* __restrict__ p, const float* __restrict__ g, const int64_t* __restrict__ blk_off,
    const int32_t* __restrict__ blk_len, float gs, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t off = blk_off[blockIdx.x];
  const int len = blk_len[blockIdx.x];
  float sp = 0.f, sg = 0.f;
  for (int i = threadIdx.x * 4; i < len; i += kThreads * 4) {
    if (i + 4 <= len) {
      const float4 pv = *reinterpret_cast<const float4*>(p + off + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + off + i);
      sp += pv.x * pv.x + pv.y * pv.y + pv.z * pv.z + pv.w * pv.w;
      sg += (gv.x * gv.x + gv.y * gv.y + gv.z * gv.z + gv.w * gv.w) * gs * gs;
    } else {
      for (int e = i; e < len; ++e) { sp += p[off + e] * p[off + e]; sg += g[off + e] * g[off + e] * gs * gs; }
    }
  }
  sp = block_sum(sp, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sp;
    partial[2 * blockIdx.x + 1] = sg;
  }
}

// norms[seg] = fixed-order sum of the partials of the segment's blocks (blk_seg is ascending: the
// blocks of one parameter are consecutive).  One workgroup per segment.
__global__ void __launch_bounds__(kThreads) lars_seg_reduce_kernel(
    const float* __restrict__ partial, const int32_t* __restrict__ blk_seg, int n_blocks,
    float* __restrict__ norms) {
  __shared__ float red[4];
  __shared__ int range[2];
  const int seg = blockIdx.x;
  if (threadIdx.x < 2) {                      // lower bound of seg (+ threadIdx.x)
    const int key = seg + (int)threadIdx.x;
    int lo = 0, hi = n_blocks;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (blk_seg[mid] < key) lo = mid + 1; else hi = mid;
    }
    range[threadIdx.x] = lo;
  }
  __syncthreads();
  float sp = 0.f, sg = 0.f;
  for (int b = range[0] + threadIdx.x; b < range[1]; b += kThreads) {
    sp += partial[2 * b];
    sg += partial[2 * b + 1];
  }
  sp = block_sum(sp, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) {
    norms[2 * seg] = sp;
    norms[2 * seg + 1] = sg;
  }
}

// lars_momentum op:  local_lr = lr*coeff*|p| / (|g| + wd*|p| + eps)  if wd > 0, |p| > 0, |g| > 0
//                    (else lr);   v = mu*v + local_lr*(g + wd*p);   p -= v
__global__ void __launch_bounds__(kThreads) lars_update_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
    const int64_t* __restrict__ blk_off, const int32_t* __restrict__ blk_len,
    const int32_t* __restrict__ blk_seg, const float* __restrict__ seg_wd,
    const float* __restrict__ norms, float lr_val, const float* __restrict__ hyper, float mu, float coeff,
    float eps, float gs) {
  const float lr = hyper ? hyper[0] : lr_val;
  const int seg = blk_seg[blockIdx.x];
  const float wd = seg_wd[seg];
  const float pn = sqrtf(norms[2 * seg]), gn = sqrtf(norms[2 * seg + 1]);
  float llr = lr;
  if (wd > 0.f && pn > 0.f && gn > 0.f) llr = lr * coeff * pn / (gn + wd * pn + eps);
  const int64_t off = blk_off[blockIdx.x];
  const int len = blk_len[blockIdx.x];
  for (int i = threadIdx.x * 4; i < len; i += kThreads * 4) {
    if (i + 4 <= len) {
      float4 pv = *reinterpret_cast<float4*>(p + off + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + off + i);
      float4 vv = *reinterpret_cast<float4*>(v + off + i);
      vv.x = mu * vv.x + llr * (gv.x * gs + wd * pv.x);
      vv.y = mu * vv.y + llr * (gv.y * gs + wd * pv.y);
      vv.z = mu * vv.z + llr * (gv.z * gs + wd * pv.z);
      vv.w = mu * vv.w + llr * (gv.w * gs + wd * pv.w);
      pv.x -= vv.x; pv.y -= vv.y; pv.z -= vv.z; pv.w -= vv.w;
      *reinterpret_cast<float4*>(v + off + i) = vv;
      *reinterpret_cast<float4*>(p + off + i) = pv;
    } else {
      for (int e = i; e < len; ++e) {
        const float vv = mu * v[off + e] + llr * (g[off + e] * gs + wd * p[off + e]);
        v[off + e] = vv;
        p[off + e] -= vv;
      }
    }
  }
}

// MomentumLARC (passl/optimizer/momentum_larc.py:56-111), per parameter tensor:
//   if |p| != 0 and |g| != 0:  a = trust*|p| / (|g| + |p|*wd + eps)  [clip: a = min(a / lr, 1)],  g' = a*(g + wd*p)
//   else                     :  g' = g  (no weight decay)
//   v = mu*v + g';   p -= lr*v
__global__ void __launch_bounds__(kThreads) larc_update_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
    const int64_t* __restrict__ blk_off, const int32_t* __restrict__ blk_len,
    const int32_t* __restrict__ blk_seg, const float* __restrict__ seg_wd,
    const float* __restrict__ norms, const float* __restrict__ hyper, float mu, float trust,
    float eps, int clip, float gs) {
  const float lr = hyper[0];
  const int seg = blk_seg[blockIdx.x];
  const float pn = sqrtf(norms[2 * seg]), gn = sqrtf(norms[2 * seg + 1]);
  float a = 1.f, wd = 0.f;
  if (pn != 0.f && gn != 0.f) {
    wd = seg_wd[seg];
    a = trust * pn / (gn + pn * wd + eps);
    if (clip) a = fminf(a / lr, 1.f);
  }
  const int64_t off = blk_off[blockIdx.x];
  const int len = blk_len[blockIdx.x];
  for (int i = threadIdx.x * 4; i < len; i += kThreads * 4) {
    if (i + 4 <= len) {
      float4 pv = *reinterpret_cast<float4*>(p + off + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + off + i);
      float4 vv = *reinterpret_cast<float4*>(v + off + i);
      vv.x = mu * vv.x + a * (gv.x * gs + wd * pv.x);
      vv.y = mu * vv.y + a * (gv.y * gs + wd * pv.y);
      vv.z = mu * vv.z + a * (gv.z * gs + wd * pv.z);
      vv.w = mu * vv.w + a * (gv.w * gs + wd * pv.w);
      pv.x -= lr * vv.x; pv.y -= lr * vv.y; pv.z -= lr * vv.z; pv.w -= lr * vv.w;
      *reinterpret_cast<float4*>(v + off + i) = vv;
      *reinterpret_cast<float4*>(p + off + i) = pv;
    } else {
      for (int e = i; e < len; ++e) {
        const float vv = mu * v[off + e] + a * (g[off + e] * gs + wd * p[off + e]);
        v[off + e] = vv;
        p[off + e] -= lr * vv;
      }
    }
  }
}

// One block = up to 1024 consecutive destination elements of one job.
template <typename T>
__global__ void __launch_bounds__(kThreads) pack_kernel(const float* __restrict__ src,
                                                        T* __restrict__ dst,
                                                        const passl_pack_job* __restrict__ jobs,
                                                        const int32_t* __restrict__ block_job,
                                                        const int32_t* __restrict__ block_start) {
  const passl_pack_job j = jobs[block_job[blockIdx.x]];
  if (j.transpose == 2) {
    // 2-D transpose of a Linear / 1x1 weight: dst[c][k] = src[k][c]; block_start = tile index over
    // 32 x 32 tiles; coalesced rows on both sides through a padded LDS tile
    __shared__ float tile[32][33];
    const int tiles_k = (j.K + 31) >> 5;
    const int tk = block_start[blockIdx.x] % tiles_k, tc = block_start[blockIdx.x] / tiles_k;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = src + j.src_off;
    T* d = dst + j.dst_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = tk * 32 + ty + 8 * i, c = tc * 32 + tx;
      tile[ty + 8 * i][tx] = (k < j.K && c < j.C) ? s[(int64_t)k * j.C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tc * 32 + ty + 8 * i, k = tk * 32 + tx;
      if (c < j.C && k < j.K) ElemTraits<T>::st(d + (int64_t)c * j.K + k, tile[tx][ty + 8 * i]);
    }
    return;
  }
  if (j.transpose == 3) {
    // straight cast of a contiguous weight (forward operand of a Linear / unpadded conv): 4 per thread
    const int64_t total = (int64_t)j.K * j.R * j.S * j.C;
    const int64_t e = (int64_t)block_start[blockIdx.x] + threadIdx.x * 4;
    const float* s = src + j.src_off;
    T* d = dst + j.dst_off;
    if (e + 4 <= total) {
      const float4 v = *reinterpret_cast<const float4*>(s + e);
      ElemTraits<T>::st(d + e, v.x); ElemTraits<T>::st(d + e + 1, v.y);
      ElemTraits<T>::st(d + e + 2, v.z); ElemTraits<T>::st(d + e + 3, v.w);
    } else {
      for (int64_t q = e; q < total; ++q) ElemTraits<T>::st(d + q, s[q]);
    }
    return;
  }
  const int inner_src = j.transpose ? j.K : j.C;            // logical innermost dim of dst
  const int inner = j.c_pad > 0 ? j.c_pad : inner_src;      // padded width
  const int outer = j.transpose ? j.C : j.K;
  const int64_t total = (int64_t)outer * j.TR * j.TS * inner;
  const float* s = src + j.src_off;
  T* d = dst + j.dst_off;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t e = (int64_t)block_start[blockIdx.x] + it * kThreads + threadIdx.x;
    if (e >= total) break;
    int64_t t = e;
    const int in = (int)(t % inner); t /= inner;
    const int ts = (int)(t % j.TS); t /= j.TS;
    const int tr = (int)(t % j.TR); t /= j.TR;
    const int out = (int)t;
    float v = 0.f;
    if (in < inner_src) {
      const int k = j.transpose ? in : out;
      const int c = j.transpose ? out : in;
      const int r = j.r_base + tr * j.r_step;
      const int sx = j.s_base + ts * j.s_step;
      if (r >= 0 && r < j.R && sx >= 0 && sx < j.S)
        v = s[(((int64_t)k * j.R + r) * j.S + sx) * j.C + c];
    }
    ElemTraits<T>::st(d + e, v);
  }
}

}  // namespace

// shared with conv_wgrad.hip / head.hip / layout_pool.hip (same library)
int passl_slab_reduce_launch(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                             hipStream_t st) {
  const int64_t n4 = n >> 2;
  if (slabs <= 4) {
    hipLaunchKernelGGL(slab_reduce_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(kThreads), 0, st,
                       ws, out, n4, slabs, accumulate);
  } else if (slabs <= 32) {
    hipLaunchKernelGGL(slab_reduce_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(kThreads), 0, st,
                       ws, out, n4, slabs, accumulate);
  } else {
    hipLaunchKernelGGL(slab_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(kThreads), 0, st,
                       ws, out, n4, slabs, accumulate);
  }
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

extern "C" int passl_hip_slab_reduce(const float* ws, float* out, int64_t n, int slabs,
                                     int accumulate, passl_stream_t stream) {
  if (!ws || !out || n <= 0 || (n & 3) || slabs <= 0 || !aligned16(ws) || !aligned16(out))
    return PASSL_EINVAL;
  return passl_slab_reduce_launch(ws, out, n, slabs, accumulate, as_stream(stream));
}

extern "C" int passl_hip_ema_update(float* k, const float* q, void* k_lp, int64_t n, float m,
                                    passl_stream_t stream) {
  if (!k || !q || n < 0 || !aligned16(k) || !aligned16(q) || (k_lp && !aligned16(k_lp)))
    return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  const int grid = grid_for(n >> 2);
  if (k_lp)
    hipLaunchKernelGGL(ema_kernel<true>, dim3(grid), dim3(kThreads), 0, as_stream(stream), k, q,
                       reinterpret_cast<bf16_t*>(k_lp), n, m);
  else
    hipLaunchKernelGGL(ema_kernel<false>, dim3(grid), dim3(kThreads), 0, as_stream(stream), k, q,
                       nullptr, n, m);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

__global__ void __launch_bounds__(kThreads) bn_fold_kernel(const float* __restrict__ flat,
                                                           const int64_t* __restrict__ gi,
                                                           const int64_t* __restrict__ bi,
                                                           const int64_t* __restrict__ mi,
                                                           const int64_t* __restrict__ vi, int64_t n,
                                                           float eps, float* __restrict__ scale,
                                                           float* __restrict__ shift) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const float sc = flat[gi[i]] * rsqrtf(flat[vi[i]] + eps);
  scale[i] = sc;
  shift[i] = flat[bi[i]] - flat[mi[i]] * sc;
}

extern "C" int passl_hip_bn_fold(const float* flat, const int64_t* gamma_idx, const int64_t* beta_idx,
                                 const int64_t* mean_idx, const int64_t* var_idx, int64_t n, float eps,
                                 float* scale, float* shift, passl_stream_t stream) {
  if (!flat || !gamma_idx || !beta_idx || !mean_idx || !var_idx || !scale || !shift || n <= 0)
    return PASSL_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     as_stream(stream), flat, gamma_idx, beta_idx, mean_idx, var_idx, n, eps, scale, shift);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_momentum_sgd(float* p, const float* g, float* v, int64_t n, float lr,
                                      float mu, float wd, float grad_scale,
                                      passl_stream_t stream) {
  if (!p || !g || !v || n < 0 || !aligned16(p) || !aligned16(g) || !aligned16(v))
    return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n >> 2)), dim3(kThreads), 0, as_stream(stream), p,
                     g, v, n, lr, (const float*)nullptr, mu, wd, grad_scale);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_momentum_sgd_dev(float* p, const float* g, float* v, int64_t n, const float* hyper,
                                          float mu, float wd, float grad_scale, passl_stream_t stream) {
  if (!p || !g || !v || !hyper || n < 0 || !aligned16(p) || !aligned16(g) || !aligned16(v))
    return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n >> 2)), dim3(kThreads), 0, as_stream(stream), p,
                     g, v, n, 0.f, hyper, mu, wd, grad_scale);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_cast_f32_to_bf16(const float* src, void* dst, int64_t n,
                                          passl_stream_t stream) {
  if (!src || !dst || n < 0 || !aligned16(src) || !aligned16(dst)) return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n >> 3)), dim3(kThreads), 0, as_stream(stream),
                     src, reinterpret_cast<bf16_t*>(dst), n);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_pack_weights(const float* src, void* dst, int dtype,
                                      const passl_pack_job* jobs, const int32_t* block_job,
                                      const int32_t* block_start, int n_blocks,
                                      passl_stream_t stream) {
  if (!src || !dst || !jobs || !block_job || !block_start || n_blocks < 0) return PASSL_EINVAL;
  if (n_blocks == 0) return PASSL_OK;
  if (dtype == PASSL_BF16)
    hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(n_blocks), dim3(kThreads), 0, as_stream(stream),
                       src, reinterpret_cast<bf16_t*>(dst), jobs, block_job, block_start);
  else if (dtype == PASSL_F32)
    hipLaunchKernelGGL(pack_kernel<float>, dim3(n_blocks), dim3(kThreads), 0, as_stream(stream),
                       src, reinterpret_cast<float*>(dst), jobs, block_job, block_start);
  else
    return PASSL_EUNSUPPORTED;
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

static int lars_impl(float* p, const float* g, float* v, const int64_t* blk_off,
                                       const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                       const float* seg_wd, int n_seg, float* norms, float lr, const float* hyper,
                                       float mu, float lars_coeff, float epsilon, float grad_scale,
                                       passl_stream_t stream) {
  if (!p || !g || !v || !blk_off || !blk_len || !blk_seg || !seg_wd || !norms || n_blocks < 0 ||
      n_seg <= 0 || !aligned16(p) || !aligned16(g) || !aligned16(v))
    return PASSL_EINVAL;
  if (n_blocks == 0) return PASSL_OK;
  hipStream_t st = as_stream(stream);
  // caller-provided workspace: norms[n_seg][2] followed by the per-block partial sums [n_blocks][2]
  float* partial = norms + 2 * (size_t)n_seg;
  hipLaunchKernelGGL(lars_norm_kernel, dim3(n_blocks), dim3(kThreads), 0, st, p, g, blk_off, blk_len,
                     grad_scale, partial);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(lars_seg_reduce_kernel, dim3(n_seg), dim3(kThreads), 0, st, partial, blk_seg,
                     (int)n_blocks, norms);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(lars_update_kernel, dim3(n_blocks), dim3(kThreads), 0, st, p, g, v, blk_off,
                     blk_len, blk_seg, seg_wd, norms, lr, hyper, mu, lars_coeff, epsilon, grad_scale);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_lars_momentum(float* p, const float* g, float* v, const int64_t* blk_off,
                                       const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                       const float* seg_wd, int n_seg, float* norms, float lr,
                                       float mu, float lars_coeff, float epsilon, float grad_scale,
                                       passl_stream_t stream) {
  return lars_impl(p, g, v, blk_off, blk_len, blk_seg, n_blocks, seg_wd, n_seg, norms, lr, nullptr, mu, lars_coeff,
                   epsilon, grad_scale, stream);
}

extern "C" int passl_hip_lars_momentum_dev(float* p, const float* g, float* v, const int64_t* blk_off,
                                           const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                           const float* seg_wd, int n_seg, float* norms, const float* hyper,
                                           float mu, float lars_coeff, float epsilon, float grad_scale,
                                           passl_stream_t stream) {
  if (!hyper) return PASSL_EINVAL;
  return lars_impl(p, g, v, blk_off, blk_len, blk_seg, n_blocks, seg_wd, n_seg, norms, 0.f, hyper, mu, lars_coeff,
                   epsilon, grad_scale, stream);
}

extern "C" int passl_hip_larc_momentum_dev(float* p, const float* g, float* v, const int64_t* blk_off,
                                           const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                           const float* seg_wd, int n_seg, float* norms, const float* hyper,
                                           float mu, float trust_coefficient, float epsilon, int clip,
                                           float grad_scale, passl_stream_t stream) {
  if (!p || !g || !v || !blk_off || !blk_len || !blk_seg || !seg_wd || !norms || !hyper || n_blocks < 0 ||
      n_seg <= 0 || !aligned16(p) || !aligned16(g) || !aligned16(v))
    return PASSL_EINVAL;
  if (n_blocks == 0) return PASSL_OK;
  hipStream_t st = as_stream(stream);
  float* partial = norms + 2 * (size_t)n_seg;          // as passl_hip_lars_momentum: same workspace layout
  hipLaunchKernelGGL(lars_norm_kernel, dim3(n_blocks), dim3(kThreads), 0, st, p, g, blk_off, blk_len,
                     grad_scale, partial);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(lars_seg_reduce_kernel, dim3(n_seg), dim3(kThreads), 0, st, partial, blk_seg,
                     (int)n_blocks, norms);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(larc_update_kernel, dim3(n_blocks), dim3(kThreads), 0, st, p, g, v, blk_off,
                     blk_len, blk_seg, seg_wd, norms, hyper, mu, trust_coefficient, epsilon, clip, grad_scale);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
