// ViT / MAE streaming kernels for gfx950: LayerNorm, GELU, MAE token plumbing (masking ranks,
// keep-gather + cls/pos, decoder unshuffle + mask tokens), patchify, masked-patch reconstruction
// loss, multi-tensor AdamW.  All HBM-bound: 16-byte accesses per lane, one wave per row where a
// row reduction is needed (wave shuffles), fp32 arithmetic, activations in `dtype`.
//
// Reference call sites: passl_v110/modeling/backbones/mae.py (= passl/models/mae.py):
//   LayerNorm/GELU  :61-85,158-189      random_masking :461-488     forward_encoder :490-510
//   forward_decoder :512-539            patchify/forward_loss :433-445,541-557
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]) {
  ElemTraits<T>::load8(p, v);
}

// ------------------------------------------------------------------ LayerNorm (one wave per row)
template <typename T>
__global__ void __launch_bounds__(kThreads) layernorm_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (int64_t)row * C;
  float s = 0.f, ss = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    float v[8];
    ld8(xr + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s += v[e]; ss += v[e] * v[e]; }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  const float mu = s / (float)C;
  float var = ss / (float)C - mu * mu;
  var = var < 0.f ? 0.f : var;
  const float rs = rsqrtf(var + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  T* yr = y + (int64_t)row * C;
  for (int c = lane * 8; c < C; c += 512) {
    float v[8], g[8], b[8];
    ld8(xr + c, v);
    ElemTraits<float>::load8(gamma + c, g);
    ElemTraits<float>::load8(beta + c, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (v[e] - mu) * rs * g[e] + b[e];
    ElemTraits<T>::store8(yr + c, v);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  The column sums of dy*xhat
// and dy are kept in registers (a lane owns the same column chunks for every row its wave visits),
// combined across the 4 waves through LDS and written to slab `blockIdx.x` of the workspace ([nb][2][C]);
// ln_param_reduce_kernel then adds the slabs in block order into dgamma / dbeta (no atomics: the sums are
// bit-reproducible from run to run).
constexpr int kLnMaxChunks = 4;            // C <= 2048

// NCH = number of 512-column chunks a lane owns (C <= 512 * NCH).  A wave handles RPI rows per
// iteration (row, row + 4, ...): all rows' x / dy chunks are fetched back to back and kept in registers
// for the second pass — the kernel is latency- not bandwidth-bound on the short ViT rows, a single
// row per wave-iteration with a re-read left it at a third of the streaming rate.
template <typename T, int NCH>
__global__ void __launch_bounds__(kThreads) layernorm_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const T* __restrict__ dres,
    T* __restrict__ dx, float* __restrict__ slab, int M, int C, int rows_per_block) {
  extern __shared__ float col[];            // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float ag[NCH][8], ab[NCH][8], gm[NCH][8];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lane * 8 + k * 512;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; gm[k][e] = 0.f; }
    if (c < C) ElemTraits<float>::load8(gamma + c, gm[k]);
  }
  constexpr int RPI = NCH == 1 ? 4 : 2;
  for (int row = r0 + wave; row < r1; row += 4 * RPI) {
    int rows[RPI];
    bool ok[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) { rows[u] = row + 4 * u; ok[u] = rows[u] < r1; }
    float xv[RPI][NCH][8], dv[RPI][NCH][8], rv[RPI][NCH][8];    // rv: residual-branch gradient
    float mu[RPI], rs[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      const int rsafe = ok[u] ? rows[u] : r0;
      const float mu_l = mean[rsafe], rs_l = rstd[rsafe];       // unconditional loads, selected below
      mu[u] = ok[u] ? mu_l : 0.f;
      rs[u] = ok[u] ? rs_l : 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = lane * 8 + k * 512;
        // branch-free: a load behind `if (ok && c < C)` compiles to load - s_waitcnt vmcnt(0) - next load;
        // out-of-range lanes read a valid address (row r0 / column 0) and are zeroed by a select
        const bool live = ok[u] && c < C;
        const int64_t off = (int64_t)(ok[u] ? rows[u] : r0) * C + (c < C ? c : 0);
        ld8(x + off, xv[u][k]);
        ld8(dy + off, dv[u][k]);
        // fetched with the other operands: a load issued after the row reductions would put a
        // second full memory round trip on the critical path of every iteration
        ld8((dres ? dres : dy) + off, rv[u][k]);          // no residual: a second (L1-resident) read of dy, unused
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xv[u][k][e] = live ? xv[u][k][e] : 0.f;
          dv[u][k][e] = live ? dv[u][k][e] : 0.f;
        }
      }
    }
    float s1[RPI], s2[RPI];
#pragma unroll
    for (int u = 0; u < RPI; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
#pragma unroll
    for (int u = 0; u < RPI; ++u)
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = lane * 8 + k * 512;
        if (c < C) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (xv[u][k][e] - mu[u]) * rs[u], gg = dv[u][k][e] * gm[k][e];
            xv[u][k][e] = xh;                 // rows beyond r1 carry rs = 0, dv = 0: no contribution
            s1[u] += gg;
            s2[u] += gg * xh;
            ag[k][e] += dv[u][k][e] * xh;
            ab[k][e] += dv[u][k][e];
          }
        }
      }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      s1[u] = wave_sum(s1[u]) / (float)C;
      s2[u] = wave_sum(s2[u]) / (float)C;
    }
#pragma unroll
    for (int u = 0; u < RPI; ++u) {
      if (!ok[u]) continue;
      T* dr = dx + (int64_t)rows[u] * C;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c = lane * 8 + k * 512;
        if (c < C) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rs[u] * (dv[u][k][e] * gm[k][e] - s1[u] - xv[u][k][e] * s2[u]);
          if (dres) {                       // gradient of the residual branch that forked off x
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += rv[u][k][e];
          }
          ElemTraits<T>::store8(dr + c, o);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        col[(wave * 2 + 0) * C + c + e] = ag[k][e];
        col[(wave * 2 + 1) * C + c + e] = ab[k][e];
      }
    }
  }
  __syncthreads();
  float* mine = slab + (int64_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < C; c += kThreads) {
    mine[c] = col[c] + col[2 * C + c] + col[4 * C + c] + col[6 * C + c];
    mine[C + c] = col[C + c] + col[3 * C + c] + col[5 * C + c] + col[7 * C + c];
  }
}

// dgamma[c] += sum_b slab[b][0][c], dbeta[c] += sum_b slab[b][1][c]: 16 columns x 16 slab lanes per block; lane z
// adds slabs z, z+16, ... in ascending order, the 16 partial sums are combined in lane order.
__global__ void __launch_bounds__(kThreads) ln_param_reduce_kernel(const float* __restrict__ slab, int nb, int C,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta) {
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, z = threadIdx.x >> 4;
  const int c2 = blockIdx.x * 16 + cl;                    // column of the [2C] slab row
  float a = 0.f;
  if (c2 < 2 * C)
    for (int b = z; b < nb; b += 16) a += slab[(int64_t)b * 2 * C + c2];
  red[z][cl] = a;
  __syncthreads();
  if (z != 0 || c2 >= 2 * C) return;
#pragma unroll
  for (int l = 1; l < 16; ++l) a += red[l][cl];
  if (c2 < C) dgamma[c2] += a;
  else dbeta[c2 - C] += a;
}

// ------------------------------------------------------------------ GELU (exact erf form)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  return cdf + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

// Tile form (as the BatchNorm streaming kernels, bn.hip): a workgroup owns U x 256 consecutive 16-byte
// chunks, a lane the chunks base + u * 256; all loads are issued back to back, branch-free (a lane past
// the end re-reads the last chunk), no loop.
constexpr int kEltU = 4;

template <typename T, bool BWD>
__global__ void __launch_bounds__(kThreads) gelu_kernel(const T* __restrict__ x,
                                                        const T* __restrict__ dy,
                                                        T* __restrict__ out, int64_t nchunks) {
  const int64_t base = (int64_t)blockIdx.x * (kThreads * kEltU) + threadIdx.x;
  float v[kEltU][8], d[kEltU][8];
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    const int64_t ic = i < nchunks ? i : nchunks - 1;
    ld8(x + ic * 8, v[u]);
    if (BWD) ld8(dy + ic * 8, d[u]);
  }
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    if (i >= nchunks) break;
    float o[8];
    if (BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = d[u][e] * gelu_grad_f(v[u][e]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = gelu_f(v[u][e]);
    }
    ElemTraits<T>::store8(out + i * 8, o);
  }
}

// tanh / its gradient (the `representation_size` head of the v2 VisionTransformer: tanh(head0(x)),
// passl/models/vision_transformer.py:340-343); same streaming form as gelu_kernel
template <typename T, bool BWD>
__global__ void __launch_bounds__(kThreads) tanh_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        T* __restrict__ out, int64_t nchunks) {
  const int64_t base = (int64_t)blockIdx.x * (kThreads * kEltU) + threadIdx.x;
  float v[kEltU][8], d[kEltU][8];
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    const int64_t ic = i < nchunks ? i : nchunks - 1;
    ld8(x + ic * 8, v[u]);
    if (BWD) ld8(dy + ic * 8, d[u]);
  }
#pragma unroll
  for (int u = 0; u < kEltU; ++u) {
    const int64_t i = base + u * kThreads;
    if (i >= nchunks) break;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = tanhf(v[u][e]);
      o[e] = BWD ? d[u][e] * (1.0f - t * t) : t;
    }
    ElemTraits<T>::store8(out + i * 8, o);
  }
}

// ------------------------------------------------------------------ MAE masking
// rank[i] = #{j : noise[j] < noise[i] or (== and j < i)}  (= ids_restore of argsort(argsort));
// ids_keep[rank] = i for rank < K; mask[i] = rank >= K.   One block per row, L <= 4096.
__global__ void __launch_bounds__(kThreads) mae_mask_kernel(const float* __restrict__ noise, int L,
                                                            int K, int32_t* __restrict__ ids_keep,
                                                            int32_t* __restrict__ ids_restore,
                                                            float* __restrict__ mask) {
  extern __shared__ float nz[];
  const float* nr = noise + (int64_t)blockIdx.x * L;
  for (int i = threadIdx.x; i < L; i += kThreads) nz[i] = nr[i];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += kThreads) {
    const float v = nz[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += (nz[j] < v || (nz[j] == v && j < i)) ? 1 : 0;
    ids_restore[(int64_t)blockIdx.x * L + i] = rank;
    mask[(int64_t)blockIdx.x * L + i] = rank >= K ? 1.f : 0.f;
    if (rank < K) ids_keep[(int64_t)blockIdx.x * K + rank] = i;
  }
}

// encoder input: out[b,0] = cls + pos[0]; out[b,1+k] = x[b, ids_keep[b,k]] + pos[1 + ids_keep[b,k]]
template <typename T>
__global__ void __launch_bounds__(kThreads) mae_gather_kernel(
    const T* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
    const int32_t* __restrict__ ids_keep, T* __restrict__ out, int B, int L, int K, int D) {
  const int cpr = D >> 3;
  const int64_t total = (int64_t)B * (K + 1) * cpr;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(i % cpr) * 8;
    const int64_t tok = i / cpr;
    const int t = (int)(tok % (K + 1)), b = (int)(tok / (K + 1));
    float v[8], p[8];
    int src = 0;
    if (t == 0) {
      ElemTraits<float>::load8(cls + c, v);
    } else {
      src = ids_keep[(int64_t)b * K + t - 1];
      ld8(x + ((int64_t)b * L + src) * D + c, v);
      src += 1;
    }
    ElemTraits<float>::load8(pos + (int64_t)src * D + c, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += p[e];
    ElemTraits<T>::store8(out + tok * D + c, v);
  }
}

// backward of the gather: dx[b, l] = dout[b, 1 + rank] if rank < K else 0 (rank = ids_restore)
template <typename T>
__global__ void __launch_bounds__(kThreads) mae_gather_bwd_kernel(
    const T* __restrict__ dout, const int32_t* __restrict__ ids_restore, T* __restrict__ dx,
    int B, int L, int K, int D) {
  const int cpr = D >> 3;
  const int64_t total = (int64_t)B * L * cpr;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(i % cpr) * 8;
    const int64_t tok = i / cpr;
    const int l = (int)(tok % L), b = (int)(tok / L);
    float v[8];
    const int rank = ids_restore[(int64_t)b * L + l];
    if (rank < K) ld8(dout + ((int64_t)b * (K + 1) + 1 + rank) * D + c, v);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    ElemTraits<T>::store8(dx + ((int64_t)b * L + l) * D + c, v);
  }
}

// dcls[c] += sum_b dout[b, 0, c] (row stride `stride` elements): 32 columns x 8 image lanes per block, lane z adds
// images z, z+8, ... in ascending order, the 8 partial sums are combined in lane order (no atomics).
template <typename T>
__global__ void __launch_bounds__(kThreads) cls_grad_kernel(const T* __restrict__ dout, float* __restrict__ dcls,
                                                            int B, int64_t stride, int D) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, z = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (c < D)
    for (int b = z; b < B; b += 8) a += ElemTraits<T>::ld(dout + (int64_t)b * stride + c);
  red[z][cl] = a;
  __syncthreads();
  if (z != 0 || c >= D) return;
#pragma unroll
  for (int l = 1; l < 8; ++l) a += red[l][cl];
  dcls[c] += a;
}

// decoder input: out[b,0] = x[b,0] + pos[0]; out[b,1+l] = (r < K ? x[b,1+r] : mask_token) + pos[1+l]
template <typename T>
__global__ void __launch_bounds__(kThreads) mae_unshuffle_kernel(
    const T* __restrict__ x, const float* __restrict__ mask_token, const float* __restrict__ pos,
    const int32_t* __restrict__ ids_restore, T* __restrict__ out, int B, int L, int K, int D) {
  const int cpr = D >> 3;
  const int64_t total = (int64_t)B * (L + 1) * cpr;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int c = (int)(i % cpr) * 8;
    const int64_t tok = i / cpr;
    const int t = (int)(tok % (L + 1)), b = (int)(tok / (L + 1));
    float v[8], p[8];
    int r = 0;
    if (t > 0) r = ids_restore[(int64_t)b * L + t - 1] + 1;
    if (r <= K) ld8(x + ((int64_t)b * (K + 1) + r) * D + c, v);
    else ElemTraits<float>::load8(mask_token + c, v);
    ElemTraits<float>::load8(pos + (int64_t)t * D + c, p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += p[e];
    ElemTraits<T>::store8(out + tok * D + c, v);
  }
}

// backward: dx[b,0] = dout[b,0]; dx[b,1+r] = dout[b,1+l] where ids_restore[b,l] = r < K (the kept
// token k sits at position l = ids_keep[b,k]); dmask_token += sum over masked positions.
// A thread keeps ONE column chunk for all the tokens it visits (block = 32 chunks x 8 token lanes,
// grid.y = token slabs), so the mask-token gradient is accumulated in registers, reduced over the
// token lanes in LDS and written to slab `blockIdx.y` of the workspace ([slabs][D]); the slabs are added in order
// afterwards (passl_slab_reduce_launch): no atomics.
template <typename T>
__global__ void __launch_bounds__(kThreads) mae_unshuffle_bwd_kernel(
    const T* __restrict__ dout, const int32_t* __restrict__ ids_keep,
    const int32_t* __restrict__ ids_restore, T* __restrict__ dx, float* __restrict__ dmask,
    int B, int L, int K, int D, int toks_per_block) {
  __shared__ float red[8][32 * 8 + 8];
  const int cc = threadIdx.x & 31, tl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cc) * 8;
  const int64_t ntok = (int64_t)B * (L + 1);
  const int64_t t0 = (int64_t)blockIdx.y * toks_per_block;
  int64_t t1 = t0 + toks_per_block;
  if (t1 > ntok) t1 = ntok;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < D) {
    for (int64_t tok = t0 + tl; tok < t1; tok += 8) {
      const int t = (int)(tok % (L + 1)), b = (int)(tok / (L + 1));
      float v[8];
      if (t <= K) {       // destination row t of dx: source position in dout
        const int src = t == 0 ? 0 : 1 + ids_keep[(int64_t)b * K + t - 1];
        ld8(dout + ((int64_t)b * (L + 1) + src) * D + c, v);
        ElemTraits<T>::store8(dx + ((int64_t)b * (K + 1) + t) * D + c, v);
      }
      if (t > 0 && ids_restore[(int64_t)b * L + t - 1] >= K) {
        ld8(dout + tok * D + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tl][cc * 8 + e] = acc[e];
  __syncthreads();
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col < D) {
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) s += red[l][threadIdx.x];
    dmask[(int64_t)blockIdx.y * D + col] = s;             // (the workspace slab)
  }
}

// ------------------------------------------------------------------ patchify
// imgs fp32 NCHW [B,C,H,W] -> out [B*L, p*p*C] with column order (ph, pw, c): the K-order of the
// patch-embed weight stored [Cout][p][p][C] and of MAE.patchify ('nchpwq->nhwpqc')
template <typename T>
__global__ void __launch_bounds__(kThreads) patchify_kernel(const float* __restrict__ img,
                                                            T* __restrict__ out, int B, int C, int H,
                                                            int W, int p) {
  const int gh = H / p, gw = W / p;
  const int P = p * p * C;
  const int64_t total = (int64_t)B * gh * gw * P;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int col = (int)(i % P);
    const int64_t patch = i / P;
    const int c = col % C, pw = (col / C) % p, ph = col / (C * p);
    const int w_ = (int)(patch % gw), h_ = (int)((patch / gw) % gh), b = (int)(patch / (gw * gh));
    const float v = img[(((int64_t)b * C + c) * H + h_ * p + ph) * W + w_ * p + pw];
    ElemTraits<T>::st(out + i, v);
  }
}

// ------------------------------------------------------------------ masked-patch loss
// one wave per patch.  pred fp32 [B, L+1, P] (row 0 of every image = cls, skipped), target from the
// image: optional per-patch (mean, unbiased var) normalisation, loss = sum_l mask * mean_P (pred -
// target)^2 / denom.  BWD writes dpred (zeros on cls rows and kept patches).
template <bool BWD>
__global__ void __launch_bounds__(kThreads) mae_loss_kernel(
    const float* __restrict__ img, const float* __restrict__ pred, const float* __restrict__ mask,
    const float* __restrict__ gscale, float* __restrict__ out, float* __restrict__ dpred, int B, int C,
    int H, int W, int p, int norm_pix, float inv_denom) {
  const int gh = H / p, gw = W / p, L = gh * gw, P = p * p * C;
  const int lane = threadIdx.x & 63;
  const int64_t patch = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (patch >= (int64_t)B * (L + 1)) return;
  const int t = (int)(patch % (L + 1)), b = (int)(patch / (L + 1));
  float* dr = BWD ? dpred + patch * P : nullptr;
  const float m = t == 0 ? 0.f : mask[(int64_t)b * L + t - 1];
  if (m == 0.f) {
    if (BWD) for (int i = lane; i < P; i += 64) dr[i] = 0.f;
    else if (lane == 0) out[patch] = 0.f;
    return;
  }
  const int l = t - 1, w_ = l % gw, h_ = l / gw;
  // first pass: patch statistics
  float s = 0.f, ss = 0.f;
  for (int i = lane; i < P; i += 64) {
    const int c = i % C, pw = (i / C) % p, ph = i / (C * p);
    const float v = img[(((int64_t)b * C + c) * H + h_ * p + ph) * W + w_ * p + pw];
    s += v; ss += v * v;
  }
  s = wave_sum(s); ss = wave_sum(ss);
  float mu = 0.f, rs = 1.f;
  if (norm_pix) {
    mu = s / (float)P;
    float var = (ss - (float)P * mu * mu) / (float)(P - 1);      // unbiased (paddle var default)
    var = var < 0.f ? 0.f : var;
    rs = rsqrtf(var + 1e-6f);
  }
  const float* pr = pred + patch * P;
  const float g = BWD ? (gscale ? *gscale : 1.f) * 2.f * inv_denom / (float)P : 0.f;
  float acc = 0.f;
  for (int i = lane; i < P; i += 64) {
    const int c = i % C, pw = (i / C) % p, ph = i / (C * p);
    const float v = img[(((int64_t)b * C + c) * H + h_ * p + ph) * W + w_ * p + pw];
    const float d = pr[i] - (v - mu) * rs;
    if (BWD) dr[i] = g * d;
    else acc += d * d;
  }
  if (!BWD) {
    acc = wave_sum(acc);
    if (lane == 0) out[patch] = acc / (float)P * inv_denom;        // (the per-patch workspace)
  }
}

// out[0] = sum_i v[i] in ONE fixed order: thread t adds elements t, t+256, ...; wave shuffles; 4 wave sums in order
__global__ void __launch_bounds__(kThreads) ordered_sum_kernel(const float* __restrict__ v, int64_t n,
                                                               float* __restrict__ out) {
  __shared__ float part[4];
  float a = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) a += v[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = part[0] + part[1] + part[2] + part[3];
}

// ------------------------------------------------------------------ AdamW (flat buffer)
// p *= 1 - lr*wd; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr_t * m / (sqrt(v) + eps_t),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps_t = eps*sqrt(1-b2^t)
__global__ void __launch_bounds__(kThreads) adamw_kernel(float* __restrict__ p,
                                                         const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         int64_t n, float lr, float b1p, float b2p,
                                                         const float* __restrict__ hyper, float wd, float eps,
                                                         float b1, float b2, float gs) {
  // the step-dependent scalars come by value or from device memory (hyper = {lr, beta1^t, beta2^t}: HIP-graph
  // replays); the derived ones are computed here either way (correctly rounded sqrt / divide: same bits as on the host)
  if (hyper) { lr = hyper[0]; b1p = hyper[1]; b2p = hyper[2]; }
  const float c2 = sqrtf(1.f - b2p);
  const float decay = 1.f - lr * wd;
  const float lr_t = lr * c2 / (1.f - b1p);
  const float eps_t = eps * c2;
  const int64_t nv = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nv; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; const float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gp[e] * gs;
      mp[e] = b1 * mp[e] + (1.f - b1) * gg;
      vp[e] = b2 * vp[e] + (1.f - b2) * gg * gg;
      pp[e] = pp[e] * decay - lr_t * (mp[e] / (sqrtf(vp[e]) + eps_t));
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nv << 2) + threadIdx.x;
    const float gg = g[i] * gs;
    m[i] = b1 * m[i] + (1.f - b1) * gg;
    v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
    p[i] = p[i] * decay - lr_t * (m[i] / (sqrtf(v[i]) + eps_t));
  }
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
static inline unsigned elt_grid(int64_t nchunks) {
  return (unsigned)((nchunks + kThreads * kEltU - 1) / (kThreads * kEltU));
}

}  // namespace

#define VIT_DISPATCH(dtype, ...)                                \
  if ((dtype) == PASSL_BF16) { using T = bf16_t; __VA_ARGS__ }  \
  else if ((dtype) == PASSL_F32) { using T = float; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

extern "C" int passl_hip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                       float* mean, float* rstd, int64_t M, int C, float eps,
                                       int dtype, passl_stream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || M <= 0 || C <= 0 || (C & 7) ||
      !aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3((unsigned)((M + 3) / 4)),
                                         dim3(kThreads), 0, as_stream(stream),
                                         reinterpret_cast<const T*>(x), gamma, beta,
                                         reinterpret_cast<T*>(y), mean, rstd, (int)M, C, eps);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

static inline void ln_bwd_blocks(int64_t M, int& rows, int& nb) {
  // 16..64 rows per block: >= 2 blocks per CU for the short-sequence shapes (CLIP: M = 6400 / 9856)
  rows = (int)(M / 1024);
  rows = rows < 16 ? 16 : (rows > 64 ? 64 : rows);
  nb = (int)((M + rows - 1) / rows);
}

extern "C" int64_t passl_hip_layernorm_bwd_ws_floats(int64_t M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rows, nb;
  ln_bwd_blocks(M, rows, nb);
  return (int64_t)nb * 2 * C;
}

extern "C" int passl_hip_layernorm_bwd(const void* dy, const void* x, const float* gamma,
                                       const float* mean, const float* rstd, const void* dres,
                                       void* dx, float* dgamma, float* dbeta, int64_t M, int C,
                                       int dtype, float* ws, int64_t ws_floats, passl_stream_t stream) {
  // dgamma == dbeta == NULL: leave the per-block partial sums in ws; passl_hip_layernorm_param_reduce folds them later
  // (on another stream: the fold is a latency-bound launch that the input-gradient chain does not depend on)
  const bool defer = !dgamma && !dbeta;
  if (!dy || !x || !gamma || !mean || !rstd || !dx || (!defer && (!dgamma || !dbeta)) || !ws || M <= 0 || C <= 0 ||
      (C & 7) || C > 512 * kLnMaxChunks || !aligned16(dy) || !aligned16(x) || !aligned16(dx) ||
      !aligned16(gamma) || (dres && !aligned16(dres)) || ws_floats < passl_hip_layernorm_bwd_ws_floats(M, C))
    return PASSL_EINVAL;
  int rows, nb;
  ln_bwd_blocks(M, rows, nb);
#define LN_BWD_LAUNCH(NCH)                                                                          \
  VIT_DISPATCH(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH>), dim3(nb), dim3(kThreads),  \
                                         8 * C * sizeof(float), as_stream(stream),                  \
                                         reinterpret_cast<const T*>(dy), reinterpret_cast<const T*>(x), \
                                         gamma, mean, rstd, reinterpret_cast<const T*>(dres),       \
                                         reinterpret_cast<T*>(dx), ws, (int)M, C, rows);)
  if (C <= 512) { LN_BWD_LAUNCH(1) }
  else if (C <= 1024) { LN_BWD_LAUNCH(2) }
  else { LN_BWD_LAUNCH(4) }
#undef LN_BWD_LAUNCH
  if (!defer)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 15) / 16), dim3(kThreads), 0, as_stream(stream), ws, nb,
                       C, dgamma, dbeta);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_layernorm_param_reduce(const float* ws, int64_t M, int C, float* dgamma, float* dbeta,
                                                passl_stream_t stream) {
  if (!ws || !dgamma || !dbeta || M <= 0 || C <= 0 || (C & 7) || C > 512 * kLnMaxChunks) return PASSL_EINVAL;
  int rows, nb;
  ln_bwd_blocks(M, rows, nb);
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * C + 15) / 16), dim3(kThreads), 0, as_stream(stream), ws, nb,
                     C, dgamma, dbeta);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_gelu_fwd(const void* x, void* y, int64_t n, int dtype, passl_stream_t stream) {
  if (!x || !y || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(y)) return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL((gelu_kernel<T, false>), dim3(elt_grid(n >> 3)), dim3(kThreads),
                                         0, as_stream(stream), reinterpret_cast<const T*>(x), nullptr,
                                         reinterpret_cast<T*>(y), n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype,
                                  passl_stream_t stream) {
  if (!dy || !x || !dx || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(dy) || !aligned16(dx))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL((gelu_kernel<T, true>), dim3(elt_grid(n >> 3)), dim3(kThreads),
                                         0, as_stream(stream), reinterpret_cast<const T*>(x),
                                         reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx),
                                         n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_tanh_fwd(const void* x, void* y, int64_t n, int dtype, passl_stream_t stream) {
  if (!x || !y || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(y)) return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL((tanh_kernel<T, false>), dim3(elt_grid(n >> 3)), dim3(kThreads),
                                         0, as_stream(stream), reinterpret_cast<const T*>(x), nullptr,
                                         reinterpret_cast<T*>(y), n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_tanh_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype,
                                  passl_stream_t stream) {
  if (!dy || !x || !dx || n <= 0 || (n & 7) || !aligned16(x) || !aligned16(dy) || !aligned16(dx))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL((tanh_kernel<T, true>), dim3(elt_grid(n >> 3)), dim3(kThreads),
                                         0, as_stream(stream), reinterpret_cast<const T*>(x),
                                         reinterpret_cast<const T*>(dy), reinterpret_cast<T*>(dx),
                                         n >> 3);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_mae_mask(const float* noise, int B, int L, int len_keep, int32_t* ids_keep,
                                  int32_t* ids_restore, float* mask, passl_stream_t stream) {
  if (!noise || !ids_keep || !ids_restore || !mask || B <= 0 || L <= 0 || L > 4096 || len_keep <= 0 ||
      len_keep > L)
    return PASSL_EINVAL;
  hipLaunchKernelGGL(mae_mask_kernel, dim3(B), dim3(kThreads), L * sizeof(float), as_stream(stream),
                     noise, L, len_keep, ids_keep, ids_restore, mask);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_mae_gather(const void* x, const float* cls, const float* pos,
                                    const int32_t* ids_keep, void* out, int B, int L, int K, int D,
                                    int dtype, passl_stream_t stream) {
  if (!x || !cls || !pos || !ids_keep || !out || B <= 0 || L <= 0 || K <= 0 || K > L || D <= 0 || (D & 7))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(mae_gather_kernel<T>, dim3(grid_for((int64_t)B * (K + 1) * (D >> 3))),
                                         dim3(kThreads), 0, as_stream(stream),
                                         reinterpret_cast<const T*>(x), cls, pos, ids_keep,
                                         reinterpret_cast<T*>(out), B, L, K, D);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_mae_gather_bwd(const void* dout, const int32_t* ids_restore, void* dx,
                                        float* dcls, int B, int L, int K, int D, int dtype,
                                        passl_stream_t stream) {
  if (!dout || !ids_restore || !dx || !dcls || B <= 0 || L <= 0 || K <= 0 || K > L || D <= 0 || (D & 7))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(mae_gather_bwd_kernel<T>,
                                         dim3(grid_for((int64_t)B * L * (D >> 3))), dim3(kThreads), 0,
                                         as_stream(stream), reinterpret_cast<const T*>(dout), ids_restore,
                                         reinterpret_cast<T*>(dx), B, L, K, D);
               hipLaunchKernelGGL(cls_grad_kernel<T>, dim3((D + 31) / 32), dim3(kThreads), 0, as_stream(stream),
                                  reinterpret_cast<const T*>(dout), dcls, B, (int64_t)(K + 1) * D, D);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_mae_unshuffle(const void* x, const float* mask_token, const float* pos,
                                       const int32_t* ids_restore, void* out, int B, int L, int K,
                                       int D, int dtype, passl_stream_t stream) {
  if (!x || !mask_token || !pos || !ids_restore || !out || B <= 0 || L <= 0 || K <= 0 || K > L ||
      D <= 0 || (D & 7))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(mae_unshuffle_kernel<T>,
                                         dim3(grid_for((int64_t)B * (L + 1) * (D >> 3))), dim3(kThreads), 0,
                                         as_stream(stream), reinterpret_cast<const T*>(x), mask_token, pos,
                                         ids_restore, reinterpret_cast<T*>(out), B, L, K, D);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

int passl_slab_reduce_launch(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                             hipStream_t st);   // flat.hip

// workspace: 1024 * D floats cover every shape
extern "C" int passl_hip_mae_unshuffle_bwd(const void* dout, const int32_t* ids_keep,
                                           const int32_t* ids_restore, void* dx, float* dmask_token,
                                           int B, int L, int K, int D, int dtype, float* ws,
                                           int64_t ws_floats, passl_stream_t stream) {
  if (!dout || !ids_keep || !ids_restore || !dx || !dmask_token || !ws || B <= 0 || L <= 0 || K <= 0 ||
      K > L || D <= 0 || (D & 7) || !aligned16(ws) || !aligned16(dmask_token))
    return PASSL_EINVAL;
  const int64_t ntok = (int64_t)B * (L + 1);
  int slabs = (int)((ntok + 127) / 128);
  if (slabs > 1024) slabs = 1024;
  const int tpb = (int)((ntok + slabs - 1) / slabs);
  slabs = (int)((ntok + tpb - 1) / tpb);
  if (ws_floats < (int64_t)slabs * D) return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(mae_unshuffle_bwd_kernel<T>, dim3((D + 255) / 256, slabs),
                                         dim3(kThreads), 0, as_stream(stream),
                                         reinterpret_cast<const T*>(dout), ids_keep, ids_restore,
                                         reinterpret_cast<T*>(dx), ws, B, L, K, D, tpb);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return passl_slab_reduce_launch(ws, dmask_token, D, slabs, 1, as_stream(stream));
}

extern "C" int passl_hip_patchify(const float* img, void* out, int B, int C, int H, int W, int p,
                                  int dtype, passl_stream_t stream) {
  if (!img || !out || B <= 0 || C <= 0 || p <= 0 || H <= 0 || W <= 0 || (H % p) || (W % p))
    return PASSL_EINVAL;
  VIT_DISPATCH(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(grid_for((int64_t)B * C * H * W)),
                                         dim3(kThreads), 0, as_stream(stream), img,
                                         reinterpret_cast<T*>(out), B, C, H, W, p);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// workspace: one float per (image, token) row = B * (L + 1)
extern "C" int passl_hip_mae_loss_fwd(const float* img, const float* pred, const float* mask,
                                      float* loss, int B, int C, int H, int W, int p, int norm_pix,
                                      float denom, float* ws, int64_t ws_floats, passl_stream_t stream) {
  if (!img || !pred || !mask || !loss || !ws || B <= 0 || C <= 0 || p <= 0 || (H % p) || (W % p) ||
      !(denom > 0.f))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t rows = (int64_t)B * ((H / p) * (W / p) + 1);
  if (ws_floats < rows) return PASSL_EINVAL;
  hipLaunchKernelGGL(mae_loss_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0, st, img,
                     pred, mask, nullptr, ws, nullptr, B, C, H, W, p, norm_pix, 1.0f / denom);
  hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(kThreads), 0, st, ws, rows, loss);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_mae_loss_bwd(const float* img, const float* pred, const float* mask,
                                      const float* gscale, float* dpred, int B, int C, int H, int W,
                                      int p, int norm_pix, float denom, passl_stream_t stream) {
  if (!img || !pred || !mask || !dpred || B <= 0 || C <= 0 || p <= 0 || (H % p) || (W % p) || !(denom > 0.f))
    return PASSL_EINVAL;
  const int64_t rows = (int64_t)B * ((H / p) * (W / p) + 1);
  hipLaunchKernelGGL(mae_loss_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(kThreads), 0,
                     as_stream(stream), img, pred, mask, gscale, nullptr, dpred, B, C, H, W, p, norm_pix,
                     1.0f / denom);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

static int adamw_impl(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1_pow,
                      float beta2_pow, const float* hyper, float beta1, float beta2, float epsilon,
                      float weight_decay, float grad_scale, passl_stream_t stream) {
  if (!p || !g || !m || !v || n < 0 || !aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v))
    return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  int64_t b = ((n >> 2) + kThreads - 1) / kThreads;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)b), dim3(kThreads), 0, as_stream(stream), p, g, m, v, n,
                     lr, beta1_pow, beta2_pow, hyper, weight_decay, epsilon, beta1, beta2, grad_scale);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                               float beta1, float beta2, float epsilon, float weight_decay,
                               float beta1_pow, float beta2_pow, float grad_scale,
                               passl_stream_t stream) {
  return adamw_impl(p, g, m, v, n, lr, beta1_pow, beta2_pow, nullptr, beta1, beta2, epsilon, weight_decay,
                    grad_scale, stream);
}

extern "C" int passl_hip_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                   float beta1, float beta2, float epsilon, float weight_decay, float grad_scale,
                                   passl_stream_t stream) {
  if (!hyper) return PASSL_EINVAL;
  return adamw_impl(p, g, m, v, n, 0.f, 0.f, 0.f, hyper, beta1, beta2, epsilon, weight_decay, grad_scale, stream);
}
