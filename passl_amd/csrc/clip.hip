// CLIP kernels for gfx950: QuickGELU, token embedding (+ positional) gather / scatter-add, row
// gather / scatter (class token, EOT token), EOT index (argmax of the token ids) and the symmetric
// image<->text cross-entropy over the scaled cosine similarities (forward and backward).
//
// Reference call sites:
//   passl_v110/modeling/backbones/base_transformer.py:25-28   QuickGELU  x * sigmoid(1.702 x)
//   passl_v110/modeling/backbones/clip.py:294-311             encode_text: token_embedding(text) +
//       positional_embedding ... x[i][argmax(text[i])] @ text_projection
//   passl_v110/modeling/backbones/clip.py:317-336             forward: features / ||features||,
//       image_logits = (exp(s) I) T^T, text_logits = (exp(s) T) I^T, logit_scale.clip_(-4.6, 4.6)
//   passl_v110/modeling/heads/clip_head.py:24-36              loss = CE(image_logits) + CE(text_logits)
//
// Everything here is small next to the towers' GEMMs (B x B logits with B = the per-GPU batch,
// D = 512): the kernels are written for exact fp32 arithmetic and coalesced access, the
// similarity / gradient products use v_mfma_f32_16x16x4_f32 straight from global memory (the
// operands are L2-resident).  text_logits is the transpose of image_logits (identical up to the
// rounding of exp(s) * x), so ONE logits matrix L is formed: image CE runs over its rows, text CE
// over its columns.  Backward:  G = g/B (softmax_rows(L) + softmax_cols(L) - 2 I),
//   d I^ = exp(s) G T^,  d T^ = exp(s) G^T I^,  d s = sum(G .* L),  d x = (d x^ - x^ <x^, d x^>) / ||x||.
#include <climits>
#include "common.h"

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(int64_t n, int per_block = kThreads) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > 65535 * 4) g = 65535 * 4;
  return (unsigned)(g < 1 ? 1 : g);
}

#define CLIP_DISPATCH(dtype, ...)                              \
  if ((dtype) == PASSL_BF16) { typedef bf16_t T; __VA_ARGS__ } \
  else if ((dtype) == PASSL_F32) { typedef float T; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

// ------------------------------------------------------------------ QuickGELU
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// tile form (see gelu_kernel, vit.hip): U x 256 consecutive chunks per workgroup, loads back to back
constexpr int kEltU = 4;

template <typename T, bool BWD>
__global__ void __launch_bounds__(kThreads) quick_gelu_kernel(const T* __restrict__ x,
                                                              const T* __restrict__ dy,
                                                              T* __restrict__ out, int64_t nchunks) {
  const int64_t base = (int64_t)blockIdx.x * (kThreads * kEltU) + threadIdx.x;
  float v[kEltU][8], d[kEltU][8];
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    const int64_t ic = i < nchunks ? i : nchunks - 1;
    ElemTraits<T>::load8(x + ic * 8, v[u]);
    if (BWD) ElemTraits<T>::load8(dy + ic * 8, d[u]);
  }
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    if (i >= nchunks) break;
    float o[8];
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = sigm(1.702f * v[u][e]);
        o[e] = d[u][e] * (sg + 1.702f * v[u][e] * sg * (1.0f - sg));
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[u][e] * sigm(1.702f * v[u][e]);
    }
    ElemTraits<T>::store8(out + i * 8, o);
  }
}

// ------------------------------------------------------------------ token embedding
// out[(b*Tn + t)][c] = E[text[b][t]][c] + pos[t][c]   (fp32 tables, output in the compute dtype)
template <typename T>
__global__ void __launch_bounds__(kThreads) embed_fwd_kernel(const int64_t* __restrict__ text,
                                                             const float* __restrict__ E,
                                                             const float* __restrict__ pos,
                                                             T* __restrict__ out, int64_t rows, int Tn,
                                                             int C, int vocab) {
  const int chunks = C >> 3;
  const int64_t total = rows * chunks;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / chunks;
    const int c = (int)(i % chunks) * 8;
    const int64_t tok = text[row];
    const int t = (int)(row % Tn);
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, p[8];
    if (tok >= 0 && tok < vocab) ElemTraits<float>::load8(E + tok * C + c, e);
    ElemTraits<float>::load8(pos + (int64_t)t * C + c, p);
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] += p[k];
    ElemTraits<T>::store8(out + row * C + c, e);
  }
}

// Token-embedding gradient dE[text[b][t]] += dout[b][t]: a scatter-add with arbitrary collisions (every padded
// position of every caption hits the pad token's row).  Floating-point atomics would make the sum depend on the
// order in which the hardware serialises them; instead every addend is converted to 64-bit fixed point with ONE
// power-of-two scale for the whole launch and accumulated with INTEGER atomics — integer addition is associative,
// so the result is independent of the order, bit-reproducible, and exact (the final fp32 value is the correctly
// rounded sum).  Scale: 2^(61 - e) with 2^e > max|dout| * rows, so the total cannot overflow; an addend keeps all
// 24 mantissa bits unless it is more than ~2^37 times smaller than the largest one.
//   pass 1  embed_absmax_kernel   max|dout| via atomicMax on the float's bit pattern (order-independent)
//   pass 2  embed_scatter_kernel  fixed-point atomics into acc[vocab][C] (int64), touched[tok] = 1;
//                                 per-(position, image-slab) column sums of dout into the dpos slabs
//   pass 3  embed_flush_kernel    dE[tok] += acc[tok] * 2^-(61-e) for touched tokens; acc and touched are cleared
// acc / touched must be zero on entry and are zero again on return (a persistent workspace of the caller).
template <typename T>
__global__ void __launch_bounds__(kThreads) embed_absmax_kernel(const T* __restrict__ dout, int64_t nchunks,
                                                                uint32_t* __restrict__ amax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * kThreads) {
    float v[8];
    ElemTraits<T>::load8(dout + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax, __float_as_uint(m));   // non-negative floats order as uints
}

__device__ __forceinline__ int embed_scale_exp(uint32_t amax_bits, int64_t rows) {
  int e;
  frexpf(__uint_as_float(amax_bits) * (float)rows, &e);      // amax * rows < 2^e
  return min(61 - e, 120);                                   // (2^s must stay a finite float for tiny gradients)
}

template <typename T>
__global__ void __launch_bounds__(kThreads) embed_scatter_kernel(const int64_t* __restrict__ text,
                                                                 const T* __restrict__ dout,
                                                                 const uint32_t* __restrict__ amax,
                                                                 unsigned long long* __restrict__ acc,
                                                                 int32_t* __restrict__ touched,
                                                                 float* __restrict__ pos_slab, int B, int Tn,
                                                                 int C, int vocab) {
  extern __shared__ float red[];                       // [rows_par][C]
  const int chunks = C >> 3;
  const int rows_par = kThreads / chunks;
  const int ch = threadIdx.x % chunks, rl = threadIdx.x / chunks;
  const int t = blockIdx.x;
  const uint32_t mx = *amax;
  const float scale = mx ? ldexpf(1.0f, embed_scale_exp(mx, (int64_t)B * Tn)) : 0.f;
  float accp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rows_par) {
    // run-length accumulation: captions are zero-padded, so at most positions the ids of consecutive
    // images repeat (the pad id) — their gradients are summed (exactly, in fixed point) in registers first
    long long run[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t run_tok = -1;
    for (int b = blockIdx.y * rows_par + rl; b < B; b += gridDim.y * rows_par) {
      const int64_t row = (int64_t)b * Tn + t;
      float v[8];
      ElemTraits<T>::load8(dout + row * C + ch * 8, v);
      const int64_t tok = text[row];
#pragma unroll
      for (int k = 0; k < 8; ++k) accp[k] += v[k];
      if (tok != run_tok) {
        if (run_tok >= 0 && run_tok < vocab) {
          unsigned long long* dst = acc + run_tok * C + ch * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) atomicAdd(dst + k, (unsigned long long)run[k]);
          if (ch == 0) touched[run_tok] = 1;
        }
        run_tok = tok;
#pragma unroll
        for (int k = 0; k < 8; ++k) run[k] = (long long)(v[k] * scale);     // power-of-two scale: exact
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) run[k] += (long long)(v[k] * scale);
      }
    }
    if (run_tok >= 0 && run_tok < vocab) {
      unsigned long long* dst = acc + run_tok * C + ch * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(dst + k, (unsigned long long)run[k]);
      if (ch == 0) touched[run_tok] = 1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[rl * C + ch * 8 + k] = accp[k];
  }
  __syncthreads();
  float* slab = pos_slab + ((int64_t)blockIdx.y * Tn + t) * C;          // [slabs][Tn][C]
  for (int c = threadIdx.x; c < C; c += kThreads) {
    float s = 0.f;
    for (int r = 0; r < rows_par; ++r) s += red[r * C + c];
    slab[c] = s;
  }
}

// one wave per vocabulary row
__global__ void __launch_bounds__(kThreads) embed_flush_kernel(unsigned long long* __restrict__ acc,
                                                               int32_t* __restrict__ touched,
                                                               const uint32_t* __restrict__ amax,
                                                               float* __restrict__ dE, int vocab, int C,
                                                               int64_t rows) {
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= vocab || !touched[tok]) return;
  const double inv = ldexp(1.0, -embed_scale_exp(*amax, rows));
  for (int c = threadIdx.x & 63; c < C; c += 64) {
    const int64_t o = (int64_t)tok * C + c;
    dE[o] += (float)((double)(long long)acc[o] * inv);
    acc[o] = 0ull;
  }
  if ((threadIdx.x & 63) == 0) touched[tok] = 0;
}

// ------------------------------------------------------------------ row gather / scatter
template <typename T>
__global__ void __launch_bounds__(kThreads) gather_rows_kernel(const T* __restrict__ x,
                                                               const int32_t* __restrict__ idx,
                                                               T* __restrict__ out, int n, int C) {
  constexpr int V = ElemTraits<T>::VEC;
  const int chunks = C / V;
  const int64_t total = (int64_t)n * chunks;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int r = (int)(i / chunks), c = (int)(i % chunks) * V;
    *reinterpret_cast<uint4*>(out + (int64_t)r * C + c) =
        *reinterpret_cast<const uint4*>(x + (int64_t)idx[r] * C + c);
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads) scatter_rows_kernel(const T* __restrict__ dout,
                                                                const int32_t* __restrict__ idx,
                                                                T* __restrict__ dx, int n, int C) {
  constexpr int V = ElemTraits<T>::VEC;
  const int chunks = C / V;
  const int64_t total = (int64_t)n * chunks;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int r = (int)(i / chunks), c = (int)(i % chunks) * V;
    *reinterpret_cast<uint4*>(dx + (int64_t)idx[r] * C + c) =
        *reinterpret_cast<const uint4*>(dout + (int64_t)r * C + c);
  }
}

// idx[b] = b*Tn + argmax_t text[b][t] (first maximum); one wave per row
__global__ void __launch_bounds__(kThreads) eot_index_kernel(const int64_t* __restrict__ text, int B,
                                                             int Tn, int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  long long best = LLONG_MIN;
  int bi = 0x7fffffff;
  for (int t = lane; t < Tn; t += 64) {
    const long long v = text[(int64_t)b * Tn + t];
    if (v > best) { best = v; bi = t; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) idx[b] = b * Tn + bi;
}

// ------------------------------------------------------------------ contrastive loss
// y = x / ||x||, inv[row] = 1 / ||x||      (one wave per row, rows of both feature sets)
__global__ void __launch_bounds__(kThreads) clip_norm_kernel(const float* __restrict__ xi,
                                                             const float* __restrict__ xt,
                                                             float* __restrict__ yi,
                                                             float* __restrict__ yt,
                                                             float* __restrict__ inv, int B, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B) return;
  const float* x = row < B ? xi + (int64_t)row * D : xt + (int64_t)(row - B) * D;
  float* y = row < B ? yi + (int64_t)row * D : yt + (int64_t)(row - B) * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(x + c);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss);
  if (lane == 0) inv[row] = r;
  for (int c = lane * 4; c < D; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(x + c);
    v.x *= r; v.y *= r; v.z *= r; v.w *= r;
    *reinterpret_cast<float4*>(y + c) = v;
  }
}

// dx = (dy - y <y, dy>) * inv   for both feature sets
__global__ void __launch_bounds__(kThreads) clip_norm_bwd_kernel(
    const float* __restrict__ yi, const float* __restrict__ yt, const float* dyi, const float* dyt,
    const float* __restrict__ inv, float* dxi, float* dxt, int B, int D) {   // dx may alias dy
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= 2 * B) return;
  const int64_t o = (int64_t)(row < B ? row : row - B) * D;
  const float* y = (row < B ? yi : yt) + o;
  const float* dy = (row < B ? dyi : dyt) + o;
  float* dx = (row < B ? dxi : dxt) + o;
  float dot = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(y + c), d = *reinterpret_cast<const float4*>(dy + c);
    dot += a.x * d.x + a.y * d.y + a.z * d.z + a.w * d.w;
  }
  dot = wave_sum(dot);
  const float r = inv[row];
  for (int c = lane * 4; c < D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(y + c), d = *reinterpret_cast<const float4*>(dy + c);
    float4 v;
    v.x = (d.x - a.x * dot) * r; v.y = (d.y - a.y * dot) * r;
    v.z = (d.z - a.z * dot) * r; v.w = (d.w - a.w * dot) * r;
    *reinterpret_cast<float4*>(dx + c) = v;
  }
}

// C[M][N] = alpha * A[M][K] . Bm[N][K]^T   (both operands row-major over K; one wave per 16x16
// tile; lane (l15, l4) feeds k = 16*kc + 4*l4 + r at MFMA step r of chunk kc — the same
// permutation of k on both operands).  M, N multiples of 16 are not required (guards), K % 16 == 0.
__global__ void __launch_bounds__(kThreads) gemm_nt_kernel(const float* __restrict__ A,
                                                           const float* __restrict__ Bm,
                                                           float* __restrict__ Cm, int M, int N, int K,
                                                           const float* __restrict__ alpha_p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tn = (N + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ((M + 15) >> 4) * tn) return;
  const int i0 = (tile / tn) * 16, j0 = (tile % tn) * 16;
  const bool av = i0 + l15 < M, bv = j0 + l15 < N;
  const float* ap = A + (int64_t)(av ? i0 + l15 : 0) * K + l4 * 4;
  const float* bp = Bm + (int64_t)(bv ? j0 + l15 : 0) * K + l4 * 4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; k += 16) {
    float4 a = *reinterpret_cast<const float4*>(ap + k);
    float4 b = *reinterpret_cast<const float4*>(bp + k);
    if (!av) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!bv) b = make_float4(0.f, 0.f, 0.f, 0.f);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  }
  const float alpha = *alpha_p;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + l4 * 4 + r;
    if (i < M && bv) Cm[(int64_t)i * N + j0 + l15] = alpha * acc[r];
  }
}

// C[M][N] = alpha * op(G)[M][K] . X[K][N],  op(G) = G (TRANS = false, G is [M][K]) or G^T
// (TRANS = true, G is [K][M]).  K % 4 need not hold (guards).
template <bool TRANS>
__global__ void __launch_bounds__(kThreads) gemm_gx_kernel(const float* __restrict__ G,
                                                           const float* __restrict__ X,
                                                           float* __restrict__ Cm, int M, int N, int K,
                                                           const float* __restrict__ alpha_p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tn = (N + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ((M + 15) >> 4) * tn) return;
  const int i0 = (tile / tn) * 16, j0 = (tile % tn) * 16;
  const bool av = i0 + l15 < M, bv = j0 + l15 < N;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + l4;
    float a = 0.f, b = 0.f;
    if (k < K) {
      if (av) a = TRANS ? G[(int64_t)k * M + i0 + l15] : G[(int64_t)(i0 + l15) * K + k];
      if (bv) b = X[(int64_t)k * N + j0 + l15];
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  const float alpha = *alpha_p;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + l4 * 4 + r;
    if (i < M && bv) Cm[(int64_t)i * N + j0 + l15] = alpha * acc[r];
  }
}

// alpha = exp(s) as used by this step's logits, then s <- clip(s) in place (clip.py:309-311,331)
__global__ void clip_scale_kernel(float* __restrict__ logit_scale, float* __restrict__ alpha,
                                  float clip_lo, float clip_hi) {
  const float s = *logit_scale;
  *alpha = expf(s);
  *logit_scale = fminf(fmaxf(s, clip_lo), clip_hi);
}

// stats[0..B) = row log-sum-exp, stats[B..2B) = column log-sum-exp of L [B][B];
// out = {img_loss (rows), text_loss (columns), loss = their sum}, labels = arange(B)
__global__ void __launch_bounds__(kThreads) clip_ce_kernel(const float* __restrict__ Lm, int B,
                                                           float* __restrict__ stats,
                                                           float* __restrict__ terms) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrow_blocks = (B + 3) / 4;
  const float invB = 1.0f / (float)B;
  if ((int)blockIdx.x < nrow_blocks) {                    // rows: one wave per row
    const int i = blockIdx.x * 4 + wave;
    if (i >= B) return;
    const float* r = Lm + (int64_t)i * B;
    float m = -INFINITY;
    for (int j = lane; j < B; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float z = 0.f;
    for (int j = lane; j < B; j += 64) z += __expf(r[j] - m);
    z = wave_sum(z);
    if (lane == 0) {
      const float l = m + __logf(z);
      stats[i] = l;
      terms[i] = (l - r[i]) * invB;
    }
  } else {                                                // columns: one thread per column
    const int j = (blockIdx.x - nrow_blocks) * kThreads + threadIdx.x;
    if (j >= B) return;
    float m = -INFINITY;
    for (int i = 0; i < B; ++i) m = fmaxf(m, Lm[(int64_t)i * B + j]);
    float z = 0.f;
    for (int i = 0; i < B; ++i) z += __expf(Lm[(int64_t)i * B + j] - m);
    const float l = m + __logf(z);
    stats[B + j] = l;
    terms[B + j] = (l - Lm[(int64_t)j * B + j]) * invB;
  }
}

// out = {sum terms[0..B), sum terms[B..2B), their sum}, each in one fixed order (wave 0: rows, wave 1: columns)
__global__ void __launch_bounds__(128) clip_ce_finish_kernel(const float* __restrict__ terms, int B,
                                                             float* __restrict__ out) {
  __shared__ float part[2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float a = 0.f;
  for (int i = lane; i < B; i += 64) a += terms[w * B + i];
  a = wave_sum(a);
  if (lane == 0) part[w] = a;
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = part[0]; out[1] = part[1]; out[2] = part[0] + part[1]; }
}

// G[i][j] = g/B (exp(L - lse_row[i]) + exp(L - lse_col[j]) - 2 [i == j])
__global__ void __launch_bounds__(kThreads) clip_grad_kernel(const float* __restrict__ Lm,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gloss, int B,
                                                             float* __restrict__ G) {
  const int64_t total = (int64_t)B * B;
  const float k = *gloss / (float)B;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kThreads) {
    const int i = (int)(e / B), j = (int)(e % B);
    const float l = Lm[e];
    G[e] = k * (__expf(l - stats[i]) + __expf(l - stats[B + j]) - (i == j ? 2.0f : 0.0f));
  }
}

// partial[b] = sum over block b's grid-stride elements of a[e] * b[e]  (fixed order inside the block)
__global__ void __launch_bounds__(kThreads) dot_partial_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b, int64_t total,
                                                               float* __restrict__ partial) {
  __shared__ float part[4];
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kThreads)
    acc += a[e] * b[e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// out += partial[0] + partial[1] + ... (block order): the atomic-free end of a dot product
__global__ void dot_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += partial[i];
  *out += s;
}

}  // namespace

extern "C" int passl_hip_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype,
                                        passl_stream_t stream) {
  if (!x || !y || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(y)) return PASSL_EINVAL;
  CLIP_DISPATCH(dtype, hipLaunchKernelGGL((quick_gelu_kernel<T, false>), dim3((unsigned)(((n >> 3) + kThreads * kEltU - 1) / (kThreads * kEltU))),
                                          dim3(kThreads), 0, as_stream(stream),
                                          reinterpret_cast<const T*>(x), nullptr,
                                          reinterpret_cast<T*>(y), n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype,
                                        passl_stream_t stream) {
  if (!dy || !x || !dx || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(dy) || !aligned16(dx))
    return PASSL_EINVAL;
  CLIP_DISPATCH(dtype, hipLaunchKernelGGL((quick_gelu_kernel<T, true>), dim3((unsigned)(((n >> 3) + kThreads * kEltU - 1) / (kThreads * kEltU))),
                                          dim3(kThreads), 0, as_stream(stream),
                                          reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(dy),
                                          reinterpret_cast<T*>(dx), n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_embed_fwd(const int64_t* text, const float* table, const float* pos, void* out,
                                   int B, int T_, int C, int vocab, int dtype, passl_stream_t stream) {
  if (!text || !table || !pos || !out || B <= 0 || T_ <= 0 || C <= 0 || (C & 7) || vocab <= 0 ||
      !aligned16(table) || !aligned16(pos) || !aligned16(out))
    return PASSL_EINVAL;
  const int64_t rows = (int64_t)B * T_;
  CLIP_DISPATCH(dtype, hipLaunchKernelGGL(embed_fwd_kernel<T>, dim3(grid_for(rows * (C >> 3))),
                                          dim3(kThreads), 0, as_stream(stream), text, table, pos,
                                          reinterpret_cast<T*>(out), rows, T_, C, vocab);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

int passl_slab_reduce_launch(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                             hipStream_t st);   // flat.hip

static inline int embed_bwd_slabs(int B, int C) {
  const int rows_par = kThreads / (C >> 3);
  int slabs = (B + rows_par * 32 - 1) / (rows_par * 32);  // ~32 images per thread (long pad runs)
  return slabs < 1 ? 1 : (slabs > 64 ? 64 : slabs);
}

// bytes of the PERSISTENT workspace (zero on first use, left zero by every call): int64 acc[vocab][C],
// int32 touched[vocab] (+ padding), uint32 amax
extern "C" int64_t passl_hip_embed_bwd_acc_bytes(int vocab, int C) {
  if (vocab <= 0 || C <= 0) return 0;
  return (int64_t)vocab * C * 8 + (((int64_t)vocab * 4 + 15) / 16) * 16 + 16;
}

// floats of the per-call scratch (position-gradient slabs)
extern "C" int64_t passl_hip_embed_bwd_ws_floats(int B, int T_, int C) {
  if (B <= 0 || T_ <= 0 || C <= 0 || (C & 7) || (C >> 3) > kThreads) return 0;
  return (int64_t)embed_bwd_slabs(B, C) * T_ * C;
}

extern "C" int passl_hip_embed_bwd(const int64_t* text, const void* dout, float* dtable, float* dpos,
                                   int B, int T_, int C, int vocab, int dtype, void* acc, int64_t acc_bytes,
                                   float* ws, int64_t ws_floats, passl_stream_t stream) {
  if (!text || !dout || !dtable || !dpos || !acc || !ws || B <= 0 || T_ <= 0 || C <= 0 || (C & 7) ||
      vocab <= 0 || !aligned16(dout) || !aligned16(acc) || !aligned16(ws) || !aligned16(dpos))
    return PASSL_EINVAL;
  const int chunks = C >> 3;
  if (chunks > kThreads) return PASSL_EUNSUPPORTED;       // C <= 2048
  if (acc_bytes < passl_hip_embed_bwd_acc_bytes(vocab, C) || ws_floats < passl_hip_embed_bwd_ws_floats(B, T_, C))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  const int rows_par = kThreads / chunks;
  const int slabs = embed_bwd_slabs(B, C);
  unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(acc);
  int32_t* touched = reinterpret_cast<int32_t*>(acc64 + (int64_t)vocab * C);
  uint32_t* amax = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(touched) + (((int64_t)vocab * 4 + 15) / 16) * 16);
  const int64_t nchunks = (int64_t)B * T_ * chunks;
  if (passl_rec::memset_async(amax, 0, sizeof(uint32_t), st) != hipSuccess) return PASSL_ELAUNCH;
  CLIP_DISPATCH(dtype,
                hipLaunchKernelGGL(embed_absmax_kernel<T>, dim3(grid_for(nchunks)), dim3(kThreads), 0, st,
                                   reinterpret_cast<const T*>(dout), nchunks, amax);
                hipLaunchKernelGGL(embed_scatter_kernel<T>, dim3(T_, slabs), dim3(kThreads),
                                   (size_t)rows_par * C * sizeof(float), st, text,
                                   reinterpret_cast<const T*>(dout), amax, acc64, touched, ws, B, T_, C, vocab);)
  hipLaunchKernelGGL(embed_flush_kernel, dim3((vocab + 3) / 4), dim3(kThreads), 0, st, acc64, touched, amax,
                     dtable, vocab, C, (int64_t)B * T_);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return passl_slab_reduce_launch(ws, dpos, (int64_t)T_ * C, slabs, 1, st);
}

extern "C" int passl_hip_gather_rows(const void* x, const int32_t* idx, void* out, int n, int C,
                                     int dtype, passl_stream_t stream) {
  if (!x || !idx || !out || n <= 0 || C <= 0 || (C & 7) || !aligned16(x) || !aligned16(out))
    return PASSL_EINVAL;
  CLIP_DISPATCH(dtype, hipLaunchKernelGGL(gather_rows_kernel<T>,
                                          dim3(grid_for((int64_t)n * (C / ElemTraits<T>::VEC))),
                                          dim3(kThreads), 0, as_stream(stream),
                                          reinterpret_cast<const T*>(x), idx, reinterpret_cast<T*>(out),
                                          n, C);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_scatter_rows(const void* dout, const int32_t* idx, void* dx, int n,
                                      int64_t rows_total, int C, int dtype, passl_stream_t stream) {
  if (!dout || !idx || !dx || n <= 0 || rows_total < n || C <= 0 || (C & 7) || !aligned16(dout) ||
      !aligned16(dx))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  const size_t esz = dtype == PASSL_BF16 ? 2 : 4;
  if (dtype != PASSL_BF16 && dtype != PASSL_F32) return PASSL_EUNSUPPORTED;
  if (passl_rec::memset_async(dx, 0, (size_t)rows_total * C * esz, st) != hipSuccess) return PASSL_ELAUNCH;
  CLIP_DISPATCH(dtype, hipLaunchKernelGGL(scatter_rows_kernel<T>,
                                          dim3(grid_for((int64_t)n * (C / ElemTraits<T>::VEC))),
                                          dim3(kThreads), 0, st, reinterpret_cast<const T*>(dout), idx,
                                          reinterpret_cast<T*>(dx), n, C);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_eot_index(const int64_t* text, int B, int T_, int32_t* idx,
                                   passl_stream_t stream) {
  if (!text || !idx || B <= 0 || T_ <= 0) return PASSL_EINVAL;
  hipLaunchKernelGGL(eot_index_kernel, dim3((B + 3) / 4), dim3(kThreads), 0, as_stream(stream), text, B,
                     T_, idx);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// ws: fp32 workspace of passl_hip_clip_logits_ws_floats(B, D) floats, kept by the caller from the
// forward to the backward call:  [I^ B*D][T^ B*D][1/|x| 2B][alpha 8]
extern "C" int64_t passl_hip_clip_logits_ws_floats(int B, int D) {
  return 2 * (int64_t)B * D + 2 * (int64_t)B + 8;
}

extern "C" int passl_hip_clip_logits_fwd(const float* img, const float* txt, float* logit_scale, int B,
                                         int D, float clip_lo, float clip_hi, float* ws, float* logits,
                                         passl_stream_t stream) {
  if (!img || !txt || !logit_scale || !ws || !logits || B <= 0 || D <= 0 || (D & 15) ||
      !aligned16(img) || !aligned16(txt) || !aligned16(ws))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  float* In = ws;
  float* Tn = In + (int64_t)B * D;
  float* inv = Tn + (int64_t)B * D;
  float* alpha = inv + 2 * (int64_t)B;
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, st, logit_scale, alpha, clip_lo, clip_hi);
  hipLaunchKernelGGL(clip_norm_kernel, dim3((2 * B + 3) / 4), dim3(kThreads), 0, st, img, txt, In, Tn,
                     inv, B, D);
  const int tiles = ((B + 15) / 16) * ((B + 15) / 16);
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((tiles + 3) / 4), dim3(kThreads), 0, st, In, Tn, logits, B, B,
                     D, alpha);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// scratch: 256 floats (block partials of d logit_scale = sum dL o L, added in block order)
extern "C" int passl_hip_clip_logits_bwd(const float* dlogits, const float* logits, const float* ws,
                                         int B, int D, float* dimg, float* dtxt, float* dlogit_scale,
                                         float* scratch, passl_stream_t stream) {
  if (!dlogits || !logits || !ws || !dimg || !dtxt || !dlogit_scale || !scratch || B <= 0 || D <= 0 ||
      (D & 15) || !aligned16(ws) || !aligned16(dimg) || !aligned16(dtxt))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  const float* In = ws;
  const float* Tn = In + (int64_t)B * D;
  const float* inv = Tn + (int64_t)B * D;
  const float* alpha = inv + 2 * (int64_t)B;
  int dblocks = grid_for((int64_t)B * B);
  if (dblocks > 256) dblocks = 256;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(dblocks), dim3(kThreads), 0, st, dlogits, logits, (int64_t)B * B,
                     scratch);
  hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(1), 0, st, scratch, dblocks, dlogit_scale);
  // d I^ -> dimg, d T^ -> dtxt (then the normalisation backward in place)
  const int tiles = ((B + 15) / 16) * ((D + 15) / 16);
  hipLaunchKernelGGL(gemm_gx_kernel<false>, dim3((tiles + 3) / 4), dim3(kThreads), 0, st, dlogits, Tn,
                     dimg, B, D, B, alpha);
  hipLaunchKernelGGL(gemm_gx_kernel<true>, dim3((tiles + 3) / 4), dim3(kThreads), 0, st, dlogits, In,
                     dtxt, B, D, B, alpha);
  hipLaunchKernelGGL(clip_norm_bwd_kernel, dim3((2 * B + 3) / 4), dim3(kThreads), 0, st, In, Tn, dimg,
                     dtxt, inv, dimg, dtxt, B, D);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// ws: 2 * B floats (the per-row and per-column loss terms)
extern "C" int passl_hip_clip_ce_fwd(const float* logits, int B, float* lse, float* out, float* ws,
                                     int64_t ws_floats, passl_stream_t stream) {
  if (!logits || !lse || !out || !ws || B <= 0 || ws_floats < 2 * (int64_t)B) return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(clip_ce_kernel, dim3((B + 3) / 4 + (B + kThreads - 1) / kThreads), dim3(kThreads),
                     0, st, logits, B, lse, ws);
  hipLaunchKernelGGL(clip_ce_finish_kernel, dim3(1), dim3(128), 0, st, ws, B, out);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_clip_ce_bwd(const float* logits, const float* lse, const float* gloss, int B,
                                     float* dlogits, passl_stream_t stream) {
  if (!logits || !lse || !gloss || !dlogits || B <= 0) return PASSL_EINVAL;
  hipLaunchKernelGGL(clip_grad_kernel, dim3(grid_for((int64_t)B * B)), dim3(kThreads), 0,
                     as_stream(stream), logits, lse, gloss, B, dlogits);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// ---- building blocks of the cross-rank CLIP InfoNCE (rectangular logits [B][W*B]): the same kernels
// the square single-rank path uses, exposed one by one
extern "C" int passl_hip_clip_scale(float* logit_scale, float* alpha, float clip_lo, float clip_hi,
                                    passl_stream_t stream) {
  if (!logit_scale || !alpha) return PASSL_EINVAL;
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, as_stream(stream), logit_scale, alpha, clip_lo,
                     clip_hi);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_gemm_f32_nt(const float* A, const float* Bm, float* Cm, int M, int N, int K,
                                     const float* alpha, passl_stream_t stream) {
  if (!A || !Bm || !Cm || !alpha || M <= 0 || N <= 0 || K <= 0 || (K & 15) || !aligned16(A) || !aligned16(Bm))
    return PASSL_EINVAL;
  const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
  hipLaunchKernelGGL(gemm_nt_kernel, dim3((tiles + 3) / 4), dim3(kThreads), 0, as_stream(stream), A, Bm, Cm,
                     M, N, K, alpha);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_gemm_f32_gx(const float* G, const float* X, float* Cm, int M, int N, int K,
                                     int trans, const float* alpha, passl_stream_t stream) {
  if (!G || !X || !Cm || !alpha || M <= 0 || N <= 0 || K <= 0) return PASSL_EINVAL;
  const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
  if (trans)
    hipLaunchKernelGGL(gemm_gx_kernel<true>, dim3((tiles + 3) / 4), dim3(kThreads), 0, as_stream(stream), G,
                       X, Cm, M, N, K, alpha);
  else
    hipLaunchKernelGGL(gemm_gx_kernel<false>, dim3((tiles + 3) / 4), dim3(kThreads), 0, as_stream(stream), G,
                       X, Cm, M, N, K, alpha);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_dot_acc(const float* a, const float* b, int64_t n, float* out, float* ws,
                                 passl_stream_t stream) {
  if (!a || !b || !out || !ws || n <= 0) return PASSL_EINVAL;
  int blocks = grid_for(n);
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(blocks), dim3(kThreads), 0, as_stream(stream), a, b, n, ws);
  hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(1), 0, as_stream(stream), ws, blocks, out);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
