// bf16-MFMA variant of the fused short-sequence attention (see attention.hip for the algorithm, the
// reference call sites and the limits); used when the activations are bf16.
//
// Same decomposition — one workgroup per (image, head), the head's whole K and V (backward: Q and
// dO) resident in LDS, a wave owns 16 query rows (or key columns) at a time, "swapped" products so
// that a lane holds 4 scores of ONE own row per 16-wide tile — but the operands stay bf16:
//   * LDS holds bf16 rows [T_pad32][d + 8] (16-byte padded pitch): half the footprint of the fp32
//     staging, 2 workgroups per CU;
//   * scores S^T = K Q^T and dP^T = V dO^T: v_mfma_f32_16x16x32_bf16, A-operand = one ds_read_b128
//     per 32 channels of the swept row, B-operand = the own row's 16-byte global loads;
//   * P V, dS K, P^T dO, dS^T Q: the lane's 2 x 4 fp32 coefficients of a PAIR of tiles are packed
//     to one bf16x8 A-operand (k-slot e < 4 <-> row 4*l4 + e of the first tile, e >= 4 <-> the
//     second); the B-operand comes from the row-major LDS tile through two ds_read_b64_tr_b16
//     (transposing reads: a 16-lane group fetches a [4 rows][16 channels] block and lane i
//     receives channel i of the 4 rows) — no transposed copy of V / K / dO / Q is ever made.
// Softmax statistics, exp, delta and all accumulators are fp32; P and dS are rounded to bf16 only
// as MFMA operands.  8x fewer MFMA issue slots than the exact-fp32 kernels.
#include <stdlib.h>
#include "common.h"

namespace abf {

constexpr int kThreads = 256;
constexpr int kMaxTiles = 14;             // T <= 208 -> 13 tiles, padded to an even count
constexpr float kNeg = -1e30f;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
#define ABF_LDS3(p) ((__attribute__((address_space(3))) bf16x4_t*)(p))

__device__ __forceinline__ float shx(float v, int m) { return __shfl_xor(v, m, 64); }

template <int DH>
__device__ __forceinline__ int64_t qkv_off(int b, int t, int which, int h, int Tn, int H) {
  return ((((int64_t)b * Tn + t) * 3 + which) * H + h) * DH;
}

// rows [0, Tn) of a strided bf16 matrix -> LDS [Tpad][DH + 8]; rows >= Tn are zero
template <int DH, int NTH>
__device__ __forceinline__ void stage(const bf16_t* __restrict__ base, int64_t rs, int Tn, int Tpad,
                                      bf16_t* lds) {
  constexpr int P = DH + 8, CH = DH / 8;
  for (int i = threadIdx.x; i < Tpad * CH; i += NTH) {
    const int r = i / CH, c = (i % CH) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < Tn) v = *reinterpret_cast<const uint4*>(base + (int64_t)r * rs + c);
    *reinterpret_cast<uint4*>(lds + r * P + c) = v;
  }
}

// own row fragments: channels [32 s + 8 l4, +8) for s < DH/32
template <int DH>
__device__ __forceinline__ void glb_frags(const bf16_t* __restrict__ p, bool valid, int l4,
                                          bf16x8_t (&f)[DH / 32]) {
#pragma unroll
  for (int s = 0; s < DH / 32; ++s) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (valid) v = *reinterpret_cast<const uint4*>(p + 32 * s + 8 * l4);
    f[s] = __builtin_bit_cast(bf16x8_t, v);
  }
}

// acc[r] = own[row l15] . swept[row tile*16 + 4*l4 + r]
template <int DH>
__device__ __forceinline__ f32x4 dot_tile(const bf16_t* lds, int tile, const bf16x8_t (&own)[DH / 32],
                                          int l15, int l4) {
  constexpr int P = DH + 8;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < DH / 32; ++s) {
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(lds + (tile * 16 + l15) * P + 32 * s + 8 * l4);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, own[s], acc, 0, 0, 0);
  }
  return acc;
}

__device__ __forceinline__ bf16x8_t pack8(const float (&c0)[4], const float (&c1)[4]) {
  const uint4 v = make_uint4(pack2bf(c0[0], c0[1]), pack2bf(c0[2], c0[3]), pack2bf(c1[0], c1[1]),
                             pack2bf(c1[2], c1[3]));
  return __builtin_bit_cast(bf16x8_t, v);
}

// o[own = 4*l4' + r'][d = jd*16 + l15] += sum over the 32 rows of tiles (t0, t0+1) of
//   coef(own l15, row) * M[row][d];   c0 / c1 = the lane's coefficients for rows 4*l4 + r of each tile
template <int DH>
__device__ __forceinline__ void accum_pair(const float (&c0)[4], const float (&c1)[4], const bf16_t* lds,
                                           int t0, int lane, f32x4 (&o)[DH / 16]) {
  constexpr int P = DH + 8;
  const bf16x8_t a = pack8(c0, c1);
  const int j = lane & 15, l4 = lane >> 4;
  const bf16_t* base = lds + (t0 * 16 + 4 * l4 + (j >> 2)) * P + 4 * (j & 3);
#pragma unroll
  for (int jd = 0; jd < DH / 16; ++jd) {
    const bf16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(ABF_LDS3(base + jd * 16));
    const bf16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(ABF_LDS3(base + 16 * P + jd * 16));
    bf16x8_t b;
    b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3];
    b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
    o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, o[jd], 0, 0, 0);
  }
}

// ------------------------------------------------------------------ forward
template <int DH, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                 bf16_t* __restrict__ out,
                                                                 float* __restrict__ lse, int Tn, int H,
                                                                 float scale, int causal) {
  constexpr int P = DH + 8;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = ((Tn + 31) >> 5) << 5;
  bf16_t* Ks = smem;
  bf16_t* Vs = smem + Tpad * P;
  const int64_t rs = (int64_t)3 * H * DH;
  stage<DH, NW * 64>(qkv + qkv_off<DH>(b, 0, 1, h, Tn, H), rs, Tn, Tpad, Ks);
  stage<DH, NW * 64>(qkv + qkv_off<DH>(b, 0, 2, h, Tn, H), rs, Tn, Tpad, Vs);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int rb = wave; rb < nt; rb += NW) {
    const int row = rb * 16 + l15;
    bf16x8_t q[DH / 32];
    glb_frags<DH>(qkv + qkv_off<DH>(b, row < Tn ? row : 0, 0, h, Tn, H), row < Tn, l4, q);
    float s[kMaxTiles][4];
    float m = kNeg;
    const int ntc = causal ? rb + 1 : nt;
    const int lim = causal ? min(row, Tn - 1) : Tn - 1;
#pragma unroll
    for (int ct = 0; ct < kMaxTiles; ++ct) {
      if (ct < ntc) {
        const f32x4 a = dot_tile<DH>(Ks, ct, q, l15, l4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[ct][r] = (ct * 16 + l4 * 4 + r <= lim) ? a[r] * scale : kNeg;
          m = fmaxf(m, s[ct][r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[ct][r] = kNeg;
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
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) s[ct][r] = 0.f;
      }
    z += shx(z, 16);
    z += shx(z, 32);
    const float inv = 1.0f / z;
    f32x4 o[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) o[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cp = 0; cp < kMaxTiles / 2; ++cp)
      if (2 * cp < ntc) {
        float p0[4], p1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0[r] = s[2 * cp][r] * inv; p1[r] = s[2 * cp + 1][r] * inv; }
        accum_pair<DH>(p0, p1, Vs, 2 * cp, lane, o);
      }
    if (row < Tn && l4 == 0) lse[((int64_t)b * H + h) * Tn + row] = m + __logf(z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rb * 16 + l4 * 4 + r;
      if (orow < Tn) {
        bf16_t* op = out + (((int64_t)b * Tn + orow) * H + h) * DH + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) op[jd * 16] = f2bf(o[jd][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ backward, sweep 1: dQ
template <int DH, int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_q_bf16_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int Tn, int H, float scale, int causal) {
  constexpr int P = DH + 8;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = ((Tn + 31) >> 5) << 5;
  bf16_t* Ks = smem;
  bf16_t* Vs = smem + Tpad * P;
  const int64_t rs = (int64_t)3 * H * DH;
  stage<DH, NW * 64>(qkv + qkv_off<DH>(b, 0, 1, h, Tn, H), rs, Tn, Tpad, Ks);
  stage<DH, NW * 64>(qkv + qkv_off<DH>(b, 0, 2, h, Tn, H), rs, Tn, Tpad, Vs);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int rb = wave; rb < nt; rb += NW) {
    const int row = rb * 16 + l15;
    const bool rv = row < Tn;
    const int rr = rv ? row : 0;
    bf16x8_t q[DH / 32], dor[DH / 32], orw[DH / 32];
    glb_frags<DH>(qkv + qkv_off<DH>(b, rr, 0, h, Tn, H), rv, l4, q);
    const int64_t oo = (((int64_t)b * Tn + rr) * H + h) * DH;
    glb_frags<DH>(dout + oo, rv, l4, dor);
    glb_frags<DH>(out + oo, rv, l4, orw);
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += (float)dor[s][e] * (float)orw[s][e];
    delta += shx(delta, 16);
    delta += shx(delta, 32);
    const float l = rv ? lse[((int64_t)b * H + h) * Tn + row] : 0.f;
    f32x4 dq[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) dq[jd] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntc = causal ? rb + 1 : nt;
    const int lim = causal ? min(row, Tn - 1) : Tn - 1;
    for (int cp = 0; 2 * cp < ntc; ++cp) {
      float ds[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ct = 2 * cp + u;
        if (ct < ntc) {
          const f32x4 s = dot_tile<DH>(Ks, ct, q, l15, l4), dp = dot_tile<DH>(Vs, ct, dor, l15, l4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool cv = rv && (ct * 16 + l4 * 4 + r <= lim);
            const float p = cv ? __expf(s[r] * scale - l) : 0.f;
            ds[u][r] = p * (dp[r] - delta) * scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[u][r] = 0.f;
        }
      }
      accum_pair<DH>(ds[0], ds[1], Ks, 2 * cp, lane, dq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rb * 16 + l4 * 4 + r;
      if (orow < Tn) {
        bf16_t* op = dqkv + qkv_off<DH>(b, orow, 0, h, Tn, H) + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) op[jd * 16] = f2bf(dq[jd][r]);
      }
    }
  }
}

// ------------------------------------------------------------------ backward, sweep 2: dK, dV
template <int DH, int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_kv_bf16_kernel(
    const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout,
    const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int Tn, int H, float scale, int causal) {
  constexpr int P = DH + 8;
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int nt = (Tn + 15) >> 4, Tpad = ((Tn + 31) >> 5) << 5;
  bf16_t* Qs = smem;
  bf16_t* Ds = smem + Tpad * P;                                    // dO
  float* Ls = reinterpret_cast<float*>(smem + 2 * Tpad * P);       // lse[Tpad]
  float* Dl = Ls + Tpad;                                           // delta[Tpad]
  stage<DH, NW * 64>(qkv + qkv_off<DH>(b, 0, 0, h, Tn, H), (int64_t)3 * H * DH, Tn, Tpad, Qs);
  stage<DH, NW * 64>(dout + (((int64_t)b * Tn) * H + h) * DH, (int64_t)H * DH, Tn, Tpad, Ds);
  __syncthreads();
  for (int t = threadIdx.x; t < Tpad; t += NW * 64) {
    float d = 0.f, l = 0.f;
    if (t < Tn) {
      const bf16_t* op = out + (((int64_t)b * Tn + t) * H + h) * DH;
#pragma unroll
      for (int c = 0; c < DH; c += 8) {
        float a[8], o8[8];
        ElemTraits<bf16_t>::load8(Ds + t * P + c, a);
        ElemTraits<bf16_t>::load8(op + c, o8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += a[e] * o8[e];
      }
      l = lse[((int64_t)b * H + h) * Tn + t];
    }
    Ls[t] = l;
    Dl[t] = d;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ntp = Tpad >> 4;                                      // even
  for (int cb = wave; cb < nt; cb += NW) {
    const int col = cb * 16 + l15;
    const bool cv = col < Tn;
    bf16x8_t kown[DH / 32], vown[DH / 32];
    glb_frags<DH>(qkv + qkv_off<DH>(b, cv ? col : 0, 1, h, Tn, H), cv, l4, kown);
    glb_frags<DH>(qkv + qkv_off<DH>(b, cv ? col : 0, 2, h, Tn, H), cv, l4, vown);
    f32x4 dk[DH / 16], dv[DH / 16];
#pragma unroll
    for (int jd = 0; jd < DH / 16; ++jd) { dk[jd] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[jd] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int rp = causal ? (cb >> 1) : 0; 2 * rp < ntp; ++rp) {    // causal: rows before the key see nothing
      float p[2][4], ds[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int rt = 2 * rp + u;
        const f32x4 s = dot_tile<DH>(Qs, rt, kown, l15, l4), dp = dot_tile<DH>(Ds, rt, vown, l15, l4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + l4 * 4 + r;
          const bool ok = cv && row < Tn && (!causal || row >= col);
          p[u][r] = ok ? __expf(s[r] * scale - Ls[row]) : 0.f;
          ds[u][r] = p[u][r] * (dp[r] - Dl[row]) * scale;
        }
      }
      accum_pair<DH>(p[0], p[1], Ds, 2 * rp, lane, dv);
      accum_pair<DH>(ds[0], ds[1], Qs, 2 * rp, lane, dk);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ocol = cb * 16 + l4 * 4 + r;
      if (ocol < Tn) {
        bf16_t* kp = dqkv + qkv_off<DH>(b, ocol, 1, h, Tn, H) + l15;
        bf16_t* vp = dqkv + qkv_off<DH>(b, ocol, 2, h, Tn, H) + l15;
#pragma unroll
        for (int jd = 0; jd < DH / 16; ++jd) { kp[jd * 16] = f2bf(dk[jd][r]); vp[jd * 16] = f2bf(dv[jd][r]); }
      }
    }
  }
}

constexpr int max_lds(int DH) { return 2 * 224 * (DH + 8) * 2 + 2 * 224 * 4; }

// PASSL_ATTN_WAVES=4 / 8 forces the workgroup size (A/B runs); default: 8 waves from 8 row tiles on
inline bool eight_waves(int Tn, bool backward) {
  static const int forced = [] { const char* e = getenv("PASSL_ATTN_WAVES"); return e ? atoi(e) : 0; }();
  if (forced == 4) return false;
  if (forced == 8) return true;
  // measured (scratch/bench_attn.py): the forward gains from 8 waves at every benchmark shape (faster staging even
  // when half the waves have no row tile), the backward only from 8 row tiles on
  return backward ? (Tn + 15) / 16 >= 8 : true;
}

// NW = waves per workgroup: a head with >= 8 row tiles (T > 112) gets 8 waves — its K / V staging (64 KB at
// T = 197, d = 64) allows only 2 workgroups per CU, and 8 waves per CU cannot hide the global-load latency of the
// per-tile query fragments; shorter sequences keep 4 (more workgroups per CU fit anyway).
template <int DH, int NW>
int launch_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, float scale, int causal,
               hipStream_t st) {
  const int Tpad = (Tn + 31) / 32 * 32;
  const int ldsb = 2 * Tpad * (DH + 8) * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_bf16_kernel<DH, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(DH));
    attr = true;
  }
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<DH, NW>), dim3(B * H), dim3(NW * 64), ldsb, st,
                     reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<bf16_t*>(out), lse, Tn, H,
                     scale, causal);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

template <int DH, int NW>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
               int Tn, int H, float scale, int causal, hipStream_t st) {
  const int Tpad = (Tn + 31) / 32 * 32;
  const int lds1 = 2 * Tpad * (DH + 8) * 2;
  const int lds2 = lds1 + 2 * Tpad * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_bf16_kernel<DH, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(DH));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_bf16_kernel<DH, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds(DH));
    attr = true;
  }
  hipLaunchKernelGGL((attn_bwd_q_bf16_kernel<DH, NW>), dim3(B * H), dim3(NW * 64), lds1, st,
                     reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<const bf16_t*>(out),
                     reinterpret_cast<const bf16_t*>(dout), lse, reinterpret_cast<bf16_t*>(dqkv), Tn, H,
                     scale, causal);
  if (hipGetLastError() != hipSuccess) return PASSL_ELAUNCH;
  hipLaunchKernelGGL((attn_bwd_kv_bf16_kernel<DH, NW>), dim3(B * H), dim3(NW * 64), lds2, st,
                     reinterpret_cast<const bf16_t*>(qkv), reinterpret_cast<const bf16_t*>(out),
                     reinterpret_cast<const bf16_t*>(dout), lse, reinterpret_cast<bf16_t*>(dqkv), Tn, H,
                     scale, causal);
  return hipGetLastError() == hipSuccess ? PASSL_OK : PASSL_ELAUNCH;
}

}  // namespace abf

// entry points used by attention.hip (shapes already validated there); 16-byte aligned rows required
int passl_attn_bf16_fwd(const void* qkv, void* out, float* lse, int B, int Tn, int H, int DH, float scale,
                        int causal, hipStream_t st) {
  const bool wide = abf::eight_waves(Tn, false);
  if (DH == 64)
    return wide ? abf::launch_fwd<64, 8>(qkv, out, lse, B, Tn, H, scale, causal, st)
                : abf::launch_fwd<64, 4>(qkv, out, lse, B, Tn, H, scale, causal, st);
  return wide ? abf::launch_fwd<32, 8>(qkv, out, lse, B, Tn, H, scale, causal, st)
              : abf::launch_fwd<32, 4>(qkv, out, lse, B, Tn, H, scale, causal, st);
}

int passl_attn_bf16_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                        int B, int Tn, int H, int DH, float scale, int causal, hipStream_t st) {
  const bool wide = abf::eight_waves(Tn, true);
  if (DH == 64)
    return wide ? abf::launch_bwd<64, 8>(qkv, out, dout, lse, dqkv, B, Tn, H, scale, causal, st)
                : abf::launch_bwd<64, 4>(qkv, out, dout, lse, dqkv, B, Tn, H, scale, causal, st);
  return wide ? abf::launch_bwd<32, 8>(qkv, out, dout, lse, dqkv, B, Tn, H, scale, causal, st)
              : abf::launch_bwd<32, 4>(qkv, out, dout, lse, dqkv, B, Tn, H, scale, causal, st);
}
