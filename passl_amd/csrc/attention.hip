// Fused multi-head self-attention for short ViT sequences (MAE: 50 / 197 tokens) on gfx950.
//
// Reference: Attention.forward, passl_v110/modeling/backbones/mae.py:141-155 (= passl/models/
// vision_transformer.py:142-156):  softmax(q k^T * d^-0.5) v  per (image, head), q/k/v sliced from
// the fused qkv projection [B, T, 3, H, d].  The T x T score matrix never reaches HBM.
//
// One workgroup per (image, head).  The whole K and V (or Q and dO) of the head live in LDS as
// fp32 [T_pad][d+4]; a wave owns 16 query rows (or 16 key columns) at a time.  Scores come from a
// "swapped" exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: A-operand = swept tile, B-operand = own
// tile) so that a lane holds 4 scores of ONE own row per 16-wide tile: row max / sum are in-lane
// plus two wave shuffles (lanes l, l^16, l^32 share a row).  The probability tile is then already
// in the A-operand layout of the second MFMA (P x V, dS x K, P^T x dO, dS^T x Q).
// Backward = two sweeps: own query rows x swept keys -> dQ;  own key columns x swept queries ->
// dK, dV (P is recomputed from the saved row log-sum-exp; delta_i = dO_i . O_i).
// `causal` (CLIP text tower, passl_v110/modeling/backbones/clip.py:284-286: additive triu(-inf, 1)
// mask): key j is visible to query i iff j <= i; fully masked tiles are skipped.
// Limits: d in {32, 64}, T <= 208 (13 tiles) — the MAE pre-training shapes; larger T needs a
// KV-tiled (flash-style) variant.  The kernels in THIS file use exact-fp32 MFMA (bf16 inputs are
// converted when staged): they serve fp32 activations (the parity dtype); bf16 activations take
// the bf16-MFMA kernels of attention_bf16.hip.
#include <stdlib.h>
#include "common.h"

// attention_bf16.hip: bf16-MFMA kernels for bf16 activations
int passl_attn_bf16_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, int DH, float scale,
                        int causal, hipStream_t st);
int passl_attn_bf16_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                        int B, int Tn, int H, int DH, float scale, int causal, hipStream_t st);

namespace {

// PASSL_ATTN_F32MFMA=1 routes bf16 activations through the exact-fp32-MFMA kernels (A/B runs)
bool use_bf16_mfma() {
  static const bool v = [] {
    const char* e = getenv("PASSL_ATTN_F32MFMA");
    return !(e && e[0] == '1');
  }();
  return v;
}

constexpr int kThreads = 256;
constexpr int kMaxTiles = 13;
constexpr float kNeg = -1e30f;

__device__ __forceinline__ float shx(float v, int m) { return __shfl_xor(v, m, 64); }

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// element (b, t, which, h, 0) of qkv [B, T, 3, H, DH]
template <int DH>
__device__ __forceinline__ int64_t qkv_off(int b, int t, int which, int h, int Tn, int H) {
  return ((((int64_t)b * Tn + t) * 3 + which) * H + h) * DH;
}

// stage rows 0..Tn-1 (DH elements each, row stride `rs`) to fp32 LDS [Tpad][DH + 4]; rows >= Tn = 0
template <typename T, int DH>
__device__ __forceinline__ void stage(const T* __restrict__ base, int64_t rs, int Tn, int Tpad,
                                      float* lds) {
  constexpr int P = DH + 4;
  for (int i = threadIdx.x; i < Tpad * (DH / 4); i += kThreads) {
    const int r = i / (DH / 4), c = (i % (DH / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Tn) {
      const T* p = base + (int64_t)r * rs + c;
      v = make_float4(ldf(p), ldf(p + 1), ldf(p + 2), ldf(p + 3));
    }
    *reinterpret_cast<float4*>(lds + r * P + c) = v;
  }
}

// lane (l15, l4): DH/4 consecutive elements [l4*DH/4, ...) of row (tile*16 + l15) of an LDS matrix
template <int DH>
__device__ __forceinline__ void lds_row(const float* lds, int row, int l4, float (&reg)[DH / 4]) {
  constexpr int P = DH + 4;
#pragma unroll
  for (int v = 0; v < DH / 16; ++v) {
    const float4 t = *reinterpret_cast<const float4*>(lds + row * P + l4 * (DH / 4) + v * 4);
    reg[v * 4] = t.x; reg[v * 4 + 1] = t.y; reg[v * 4 + 2] = t.z; reg[v * 4 + 3] = t.w;
  }
}

template <typename T, int DH>
__device__ __forceinline__ void glb_row(const T* __restrict__ p, bool valid, int l4, float (&reg)[DH / 4]) {
#pragma unroll
  for (int v = 0; v < DH / 4; ++v) reg[v] = valid ? ldf(p + l4 * (DH / 4) + v) : 0.f;
}

// acc[r] = own[row l15] . swept[index 4*l4 + r]
template <int DH>
__device__ __forceinline__ f32x4 dot_tile(const float (&swept)[DH / 4], const float (&own)[DH / 4]) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DH / 4; ++ks)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(swept[ks], own[ks], acc, 0, 0, 0);
  return acc;
}

// out[own = 4*l4' + r'][d = jd*16 + l15] += sum_{r} coef[r] (own l15, swept 4*l4 + r) * M[swept][d]
template <int DH>
__device__ __forceinline__ void accum_tile(const float (&coef)[4], const float* lds, int tile, int l15,
                                           int l4, f32x4 (&o)[DH / 16]) {
  constexpr int P = DH + 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* rp = lds + (tile * 16 + l4 * 4 + r) * P + l15;
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd)
      o[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(coef[r], rp[jd * 16], o[jd], 0, 0, 0);
  }
}

// ------------------------------------------------------------------ forward
template <typename T, int DH>
__global__ void __launch_bounds__(kThreads) attn_fwd_kernel(const T* __restrict__ qkv,
                                                            T* __restrict__ out,
                                                            float* __restrict__ lse, int Tn, int H,
                                                            float scale, int causal) {
  constexpr int P = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = nt * 16;
  float* Ks = lds;
  float* Vs = lds + Tpad * P;
  const int64_t rs = (int64_t)3 * H * DH;
  stage<T, DH>(qkv + qkv_off<DH>(b, 0, 1, h, Tn, H), rs, Tn, Tpad, Ks);
  stage<T, DH>(qkv + qkv_off<DH>(b, 0, 2, h, Tn, H), rs, Tn, Tpad, Vs);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int rb = wave; rb < nt; rb += 4) {
    const int row = rb * 16 + l15;
    float q[DH / 4];
    glb_row<T, DH>(qkv + qkv_off<DH>(b, row < Tn ? row : 0, 0, h, Tn, H), row < Tn, l4, q);
    float s[kMaxTiles][4];
    float m = kNeg;
    const int ntc = causal ? rb + 1 : nt;        // causal: key tiles beyond the diagonal are all masked
    const int lim = causal ? min(row, Tn - 1) : Tn - 1;   // last visible key of this query row
#pragma unroll
    for (int ct = 0; ct < kMaxTiles; ++ct) {
      if (ct < ntc) {
        float kr[DH / 4];
        lds_row<DH>(Ks, ct * 16 + l15, l4, kr);
        const f32x4 a = dot_tile<DH>(kr, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[ct][r] = (ct * 16 + l4 * 4 + r <= lim) ? a[r] * scale : kNeg;
          m = fmaxf(m, s[ct][r]);
        }
      }
    }
    m = fmaxf(m, shx(m, 16));
    m = fmaxf(m, shx(m, 32));
    float z = 0.f;
#pragma unroll
    for (int ct = 0; ct < kMaxTiles; ++ct)
      if (ct < ntc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[ct][r] = __expf(s[ct][r] - m); z += s[ct][r]; }
      }
    z += shx(z, 16);
    z += shx(z, 32);
    const float inv = 1.0f / z;
    f32x4 o[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) o[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < kMaxTiles; ++ct)
      if (ct < ntc) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = s[ct][r] * inv;
        accum_tile<DH>(p, Vs, ct, l15, l4, o);
      }
    if (row < Tn && l4 == 0) lse[((int64_t)b * H + h) * Tn + row] = m + __logf(z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rb * 16 + l4 * 4 + r;
      if (orow < Tn) {
        T* op = out + (((int64_t)b * Tn + orow) * H + h) * DH + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) stf(op + jd * 16, o[jd][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ backward, sweep 1: dQ
template <typename T, int DH>
__global__ void __launch_bounds__(kThreads) attn_bwd_q_kernel(
    const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
    const float* __restrict__ lse, T* __restrict__ dqkv, int Tn, int H, float scale, int causal) {
  constexpr int P = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = nt * 16;
  float* Ks = lds;
  float* Vs = lds + Tpad * P;
  const int64_t rs = (int64_t)3 * H * DH;
  stage<T, DH>(qkv + qkv_off<DH>(b, 0, 1, h, Tn, H), rs, Tn, Tpad, Ks);
  stage<T, DH>(qkv + qkv_off<DH>(b, 0, 2, h, Tn, H), rs, Tn, Tpad, Vs);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int rb = wave; rb < nt; rb += 4) {
    const int row = rb * 16 + l15;
    const bool rv = row < Tn;
    const int rr = rv ? row : 0;
    float q[DH / 4], dor[DH / 4], orow_[DH / 4];
    glb_row<T, DH>(qkv + qkv_off<DH>(b, rr, 0, h, Tn, H), rv, l4, q);
    const int64_t oo = (((int64_t)b * Tn + rr) * H + h) * DH;
    glb_row<T, DH>(dout + oo, rv, l4, dor);
    glb_row<T, DH>(out + oo, rv, l4, orow_);
    float delta = 0.f;
#pragma unroll
    for (int v = 0; v < DH / 4; ++v) delta += dor[v] * orow_[v];
    delta += shx(delta, 16);
    delta += shx(delta, 32);
    const float l = rv ? lse[((int64_t)b * H + h) * Tn + row] : 0.f;
    f32x4 dq[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) dq[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntc = causal ? rb + 1 : nt;
    const int lim = causal ? min(row, Tn - 1) : Tn - 1;
    for (int ct = 0; ct < ntc; ++ct) {
      float kr[DH / 4], vr[DH / 4];
      lds_row<DH>(Ks, ct * 16 + l15, l4, kr);
      lds_row<DH>(Vs, ct * 16 + l15, l4, vr);
      const f32x4 s = dot_tile<DH>(kr, q), dp = dot_tile<DH>(vr, dor);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool cv = rv && (ct * 16 + l4 * 4 + r <= lim);
        const float p = cv ? __expf(s[r] * scale - l) : 0.f;
        ds[r] = p * (dp[r] - delta) * scale;
      }
      accum_tile<DH>(ds, Ks, ct, l15, l4, dq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rb * 16 + l4 * 4 + r;
      if (orow < Tn) {
        T* op = dqkv + qkv_off<DH>(b, orow, 0, h, Tn, H) + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) stf(op + jd * 16, dq[jd][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ backward, sweep 2: dK, dV
template <typename T, int DH>
__global__ void __launch_bounds__(kThreads) attn_bwd_kv_kernel(
    const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
    const float* __restrict__ lse, T* __restrict__ dqkv, int Tn, int H, float scale, int causal) {
  constexpr int P = DH + 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = nt * 16;
  float* Qs = lds;
  float* Ds = lds + Tpad * P;                 // dO
  float* Ls = lds + 2 * Tpad * P;             // lse[Tpad]
  float* Dl = Ls + Tpad;                      // delta[Tpad]
  stage<T, DH>(qkv + qkv_off<DH>(b, 0, 0, h, Tn, H), (int64_t)3 * H * DH, Tn, Tpad, Qs);
  stage<T, DH>(dout + (((int64_t)b * Tn) * H + h) * DH, (int64_t)H * DH, Tn, Tpad, Ds);
  __syncthreads();
  for (int t = threadIdx.x; t < Tpad; t += kThreads) {
    float d = 0.f, l = 0.f;
    if (t < Tn) {
      const T* op = out + (((int64_t)b * Tn + t) * H + h) * DH;
      for (int c = 0; c < DH; ++c) d += Ds[t * P + c] * ldf(op + c);
      l = lse[((int64_t)b * H + h) * Tn + t];
    }
    Ls[t] = l;
    Dl[t] = d;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int cb = wave; cb < nt; cb += 4) {
    const int col = cb * 16 + l15;
    const bool cv = col < Tn;
    float kown[DH / 4], vown[DH / 4];
    glb_row<T, DH>(qkv + qkv_off<DH>(b, cv ? col : 0, 1, h, Tn, H), cv, l4, kown);
    glb_row<T, DH>(qkv + qkv_off<DH>(b, cv ? col : 0, 2, h, Tn, H), cv, l4, vown);
    f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) { dk[jd] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[jd] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int rt = causal ? cb : 0; rt < nt; ++rt) {      // causal: query rows before the key see nothing
      float qr[DH / 4], dr[DH / 4];
      lds_row<DH>(Qs, rt * 16 + l15, l4, qr);
      lds_row<DH>(Ds, rt * 16 + l15, l4, dr);
      const f32x4 s = dot_tile<DH>(qr, kown), dp = dot_tile<DH>(dr, vown);
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + l4 * 4 + r;
        const bool ok = cv && row < Tn && (!causal || row >= col);
        p[r] = ok ? __expf(s[r] * scale - Ls[row]) : 0.f;
        ds[r] = p[r] * (dp[r] - Dl[row]) * scale;
      }
      accum_tile<DH>(p, Ds, rt, l15, l4, dv);
      accum_tile<DH>(ds, Qs, rt, l15, l4, dk);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ocol = cb * 16 + l4 * 4 + r;
      if (ocol < Tn) {
        T* kp = dqkv + qkv_off<DH>(b, ocol, 1, h, Tn, H) + l15;
        T* vp = dqkv + qkv_off<DH>(b, ocol, 2, h, Tn, H) + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) { stf(kp + jd * 16, dk[jd][r]); stf(vp + jd * 16, dv[jd][r]); }
      }
    }
  }
}

template <typename T, int DH>
int launch_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, float scale, int causal,
               hipStream_t st) {
  const int Tpad = (Tn + 15) / 16 * 16;
  const int ldsb = 2 * Tpad * (DH + 4) * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<T, DH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 208 * (DH + 4) * 4);
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<T, DH>), dim3(B * H), dim3(kThreads), ldsb, st,
                     reinterpret_cast<const T*>(qkv), reinterpret_cast<T*>(out), lse, Tn, H, scale, causal);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

template <typename T, int DH>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
               int Tn, int H, float scale, int causal, hipStream_t st) {
  const int Tpad = (Tn + 15) / 16 * 16;
  const int lds1 = 2 * Tpad * (DH + 4) * 4;
  const int lds2 = lds1 + 2 * Tpad * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<T, DH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 208 * (DH + 4) * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_kernel<T, DH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * 208 * (DH + 4) * 4 + 2 * 208 * 4);
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_q_kernel<T, DH>), dim3(B * H), dim3(kThreads), lds1, st,
                     reinterpret_cast<const T*>(qkv), reinterpret_cast<const T*>(out),
                     reinterpret_cast<const T*>(dout), lse, reinterpret_cast<T*>(dqkv), Tn, H, scale, causal);
  if (hipGetLastError() != hipSuccess) return PASSL_ELAUNCH;
  hipLaunchKernelGGL((attn_bwd_kv_kernel<T, DH>), dim3(B * H), dim3(kThreads), lds2, st,
                     reinterpret_cast<const T*>(qkv), reinterpret_cast<const T*>(out),
                     reinterpret_cast<const T*>(dout), lse, reinterpret_cast<T*>(dqkv), Tn, H, scale, causal);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

bool shape_ok(int B, int Tn, int H, int DH) {
  return B > 0 && H > 0 && Tn > 0 && Tn <= 16 * kMaxTiles && (DH == 32 || DH == 64);
}

}  // namespace

extern "C" int passl_hip_attention_fwd(const void* qkv, void* out, float* lse, int B, int T_, int H,
                                       int DH, float scale, int causal, int dtype, passl_stream_t stream) {
  if (!qkv || !out || !lse) return PASSL_EINVAL;
  if (!shape_ok(B, T_, H, DH)) return PASSL_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (dtype == PASSL_BF16 && use_bf16_mfma() && aligned16(qkv) && aligned16(out))
    return passl_attn_bf16_fwd(qkv, out, lse, B, T_, H, DH, scale, causal, st);
  if (dtype == PASSL_BF16)
    return DH == 64 ? launch_fwd<bf16_t, 64>(qkv, out, lse, B, T_, H, scale, causal, st)
                    : launch_fwd<bf16_t, 32>(qkv, out, lse, B, T_, H, scale, causal, st);
  if (dtype == PASSL_F32)
    return DH == 64 ? launch_fwd<float, 64>(qkv, out, lse, B, T_, H, scale, causal, st)
                    : launch_fwd<float, 32>(qkv, out, lse, B, T_, H, scale, causal, st);
  return PASSL_EUNSUPPORTED;
}

extern "C" int passl_hip_attention_bwd(const void* qkv, const void* out, const void* dout,
                                       const float* lse, void* dqkv, int B, int T_, int H, int DH,
                                       float scale, int causal, int dtype, passl_stream_t stream) {
  if (!qkv || !out || !dout || !lse || !dqkv) return PASSL_EINVAL;
  if (!shape_ok(B, T_, H, DH)) return PASSL_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (dtype == PASSL_BF16 && use_bf16_mfma() && aligned16(qkv) && aligned16(out) && aligned16(dout) &&
      aligned16(dqkv))
    return passl_attn_bf16_bwd(qkv, out, dout, lse, dqkv, B, T_, H, DH, scale, causal, st);
  if (dtype == PASSL_BF16)
    return DH == 64 ? launch_bwd<bf16_t, 64>(qkv, out, dout, lse, dqkv, B, T_, H, scale, causal, st)
                    : launch_bwd<bf16_t, 32>(qkv, out, dout, lse, dqkv, B, T_, H, scale, causal, st);
  if (dtype == PASSL_F32)
    return DH == 64 ? launch_bwd<float, 64>(qkv, out, dout, lse, dqkv, B, T_, H, scale, causal, st)
                    : launch_bwd<float, 32>(qkv, out, dout, lse, dqkv, B, T_, H, scale, causal, st);
  return PASSL_EUNSUPPORTED;
}
