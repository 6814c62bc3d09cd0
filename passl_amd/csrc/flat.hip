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

// g' = g*gs + wd*p; v = mu*v + g'; p -= lr*v.   20 bytes per element.
__global__ void __launch_bounds__(kThreads) sgd_kernel(float* __restrict__ p,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ v, int64_t n, float lr,
                                                       float mu, float wd, float gs) {
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

// One block = up to 1024 consecutive destination elements of one job.
template <typename T>
__global__ void __launch_bounds__(kThreads) pack_kernel(const float* __restrict__ src,
                                                        T* __restrict__ dst,
                                                        const passl_pack_job* __restrict__ jobs,
                                                        const int32_t* __restrict__ block_job,
                                                        const int32_t* __restrict__ block_start) {
  const passl_pack_job j = jobs[block_job[blockIdx.x]];
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

extern "C" int passl_hip_momentum_sgd(float* p, const float* g, float* v, int64_t n, float lr,
                                      float mu, float wd, float grad_scale,
                                      passl_stream_t stream) {
  if (!p || !g || !v || n < 0 || !aligned16(p) || !aligned16(g) || !aligned16(v))
    return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n >> 2)), dim3(kThreads), 0, as_stream(stream), p,
                     g, v, n, lr, mu, wd, grad_scale);
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
