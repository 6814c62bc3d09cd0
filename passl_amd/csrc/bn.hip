// Training-mode BatchNorm (+ReLU, +residual add) over NHWC rows x[M][C], forward and backward.
// HBM-bound streaming kernels.  Per-channel reductions are two-level: each block reduces a row
// slab in registers + LDS and writes a partial (no atomics, deterministic); a small finalize
// kernel combines the partials in fp64 in a fixed order.  The partials may also come from the
// epilogue of the convolution that produced x (forward) or dz (backward): csrc/igemm_epi.h.
//
// Forward statistics are SHIFTED sums: partial[b][c] = (sum (x - s), sum (x - s)^2) with
// s = shifts[b][c] = the first row of slab b (stored behind the sums), so the variance never
// comes from E[x^2] - mean^2 of large numbers; the finalize kernel re-centres every slab on
// slab 0's shift in fp64.
//
// Algorithmic bytes per activation (bf16): stats 2, apply 4 (+2 with residual),
// bwd_reduce 6, bwd_apply 8 (+2 with residual).
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "common.h"

namespace {

constexpr int kThreads = 256;

// Thread layout for the reductions: the C/8 "chunk columns" (8 channels = one 16 B bf16 vector)
// are spread over threadIdx % cols; the remaining threads stride over rows.
// C/8 may exceed 256 (not on this path: C <= 2048) -> column loop.

// ReLU mask sources of the backward kernels (relu argument):
//   0 none   1 z > 0 (z tensor)   2 x*scale + shift > 0 (recomputed; BN+ReLU without residual)
//   3 bit mask written by bn_apply (bit e of byte i <-> element 8*i + e)
template <typename T>
__device__ __forceinline__ void apply_relu_mask(float (&g)[8], const float (&xv)[8], int relu,
                                                const void* zsrc, int64_t chunk,
                                                const float* sc, const float* sh) {
  if (relu == 1) {
    float zz[8];
    ElemTraits<T>::load8(reinterpret_cast<const T*>(zsrc) + chunk * 8, zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
  } else if (relu == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
  } else if (relu == 3) {
    const uint32_t bits = reinterpret_cast<const uint8_t*>(zsrc)[chunk];
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = ((bits >> e) & 1u) ? g[e] : 0.f;
  }
}

template <typename T, int MODE>
// MODE 0: (sum x, sum x^2)      MODE 1: (sum g, sum g*xhat) with g = dz * relu-mask
__global__ void __launch_bounds__(kThreads) bn_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ dz, const void* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ partial,
    int64_t M, int C, int rows_per_block, int relu) {
  extern __shared__ float red[];  // [kThreads][16]
  const int cols = C >> 3;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  for (int cbase = 0; cbase < cols; cbase += kThreads) {
    const int ncol = (cols - cbase) < kThreads ? (cols - cbase) : kThreads;  // columns this pass
    const int lanes = kThreads / ncol;                                        // row lanes (>=1)
    const int col = cbase + (threadIdx.x % ncol);
    const int rl = threadIdx.x / ncol;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    float mu[8], is[8], sc[8], sh[8];
    if (MODE == 0) {
      // shift = the slab's first row (mu doubles as the shift)
      if (row0 < M) ElemTraits<T>::load8(x + row0 * C + col * 8, mu);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) mu[e] = 0.f;
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { mu[e] = mean[col * 8 + e]; is[e] = invstd[col * 8 + e]; }
      if (relu == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = scale[col * 8 + e]; sh[e] = shift[col * 8 + e]; }
      }
    }
    if (rl < lanes) {
      // 4 rows per trip: their loads are issued back to back (branch-free, a row past the slab
      // re-reads the last one) and accumulated in row order — the same sums as one row per trip
      constexpr int U = 4;
      for (int64_t r = row0 + rl; r < row1; r += (int64_t)U * lanes) {
        float v[U][8], g[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t rr = r + (int64_t)u * lanes;
          const int64_t rc = rr < row1 ? rr : row1 - 1;
          ElemTraits<T>::load8(x + rc * C + col * 8, v[u]);
          if (MODE == 1) ElemTraits<T>::load8(dz + rc * C + col * 8, g[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t rr = r + (int64_t)u * lanes;
          if (rr >= row1) break;
          if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[u][e] - mu[e]; a0[e] += d; a1[e] += d * d; }
          } else {
            apply_relu_mask<T>(g[u], v[u], relu, z, rr * cols + col, sc, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += g[u][e]; a1[e] += g[u][e] * (v[u][e] - mu[e]) * is[e]; }
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[threadIdx.x * 16 + e] = a0[e];
      red[threadIdx.x * 16 + 8 + e] = a1[e];
    }
    __syncthreads();
    // thread t < ncol sums its column over the row lanes
    if (threadIdx.x < ncol) {
      float s0[8], s1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
      for (int l = 0; l < lanes; ++l) {
        const int t = l * ncol + threadIdx.x;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += red[t * 16 + e]; s1[e] += red[t * 16 + 8 + e]; }
      }
      float* o = partial + ((int64_t)blockIdx.x * C + col * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e * 2] = s0[e]; o[e * 2 + 1] = s1[e]; }
      if (MODE == 0) {
        float* sp = partial + (int64_t)gridDim.x * C * 2 + (int64_t)blockIdx.x * C + col * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) sp[e] = mu[e];
      }
    }
  }
}

// ---- fixed-order fp64 combine of a partial slab [nblocks][C][2] — ONE launch for any slab height.
// Until round 4 tall slabs (>= 512 rows) took two launches (bn_combine: 16 row segments per channel group, then
// bn_finalize over the segment totals) and each thread walked its rows with one dependent load pair per trip.  Inside
// the training step these launches sit on the main chain between a convolution and the streaming pass that needs
// their result, 53 + 53 times per MoCo step, and under load every launch and every dependent memory round trip of
// such a latency-bound kernel costs microseconds: 15.6 + 11.2 us plus two launch gaps per tall BatchNorm and
// direction (profiles/r04_bench_bs256_bf16_kernel_stats.txt).  Now:
//  * a 256-thread block owns 8 channels (64 contiguous bytes of every slab row) and one SEGMENT of rows: 4 threads x
//    2 channels (one 16-byte load of sums + one 8-byte load of shifts per row) x 64 row lanes, the 16 (10 with shifts)
//    rows of a lane in flight at once = ONE round trip per segment of 1024 (640) rows; rows of a lane are added in
//    ascending order, lanes folded with xor shuffles (4, 8, 16, 32), the 4 waves through LDS in wave order.  Light
//    blocks on purpose: a 1024-thread block with 128 registers needs a whole idle CU and waited 44 us on average for
//    one inside the step (profiles/r05_trace_chain_1024thread_finalize.txt);
//  * several segments (tall slabs) are combined by the LAST block of the channel group to arrive: a block publishes
//    its four fp64 segment totals with 8-byte agent-scope atomic stores (write-through `sc1`), waits for them, adds 1
//    to the group's counter with a returning agent-scope atomic; the block that reads nseg - 1 back re-zeroes the
//    counter, reads all segment totals (agent-scope atomic loads) and adds them IN SEGMENT ORDER — who is last varies,
//    the arithmetic does not (bit-reproducible).  Every segment is centred on slab 0's shift, so totals simply add.
//    The protocol (payload, drained vmcnt, flag; consumer loads control-dependent on the flag's value) is the one
//    MI355X_MICROARCH.md lists as "8-B agent atomics both sides"; tools/kbench finstress runs it under load.
constexpr int kFinThreads = 256;
constexpr int kFinCh = 8;                        // channels per block
constexpr int kFinRowLanes = kFinThreads / 4;
constexpr int kFinSegMax = 16;                   // row segments of a tall slab (scratch: passl_hip_bn_partial_floats)
constexpr int kFinSegRowsFwd = kFinRowLanes * 10, kFinSegRowsBwd = kFinRowLanes * 16;

template <bool SHIFTED>
__device__ __forceinline__ void combine_slab(const float* __restrict__ partial, int nblocks, int C,
                                             int64_t M, int rows_per_block, int c, bool c_ok, int b0, int b1,
                                             double (&t1)[2], double (&t2)[2], float (&g0)[2]) {
  constexpr int kFinBatch = SHIFTED ? 10 : 16;
  __shared__ double red[kFinThreads / 64][4][4];
  const int rl = threadIdx.x >> 2;
  const float* shifts = partial + (int64_t)nblocks * C * 2;
  double a1[2] = {0.0, 0.0}, a2[2] = {0.0, 0.0};
  g0[0] = g0[1] = 0.f;
  if (c_ok) {
    if (SHIFTED) {                                       // slab 0 always holds rows
      const float2 g = *reinterpret_cast<const float2*>(shifts + c);
      g0[0] = g.x; g0[1] = g.y;
    }
    for (int bb = b0 + rl; bb < b1; bb += kFinRowLanes * kFinBatch) {
      float4 p[kFinBatch];
      float2 sh[kFinBatch];
#pragma unroll
      for (int i = 0; i < kFinBatch; ++i) {              // every load of the batch before the first use
        const int b = bb + i * kFinRowLanes;
        const int bc = b < b1 ? b : b1 - 1;
        p[i] = *reinterpret_cast<const float4*>(partial + ((int64_t)bc * C + c) * 2);
        sh[i] = SHIFTED ? *reinterpret_cast<const float2*>(shifts + (int64_t)bc * C + c) : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < kFinBatch; ++i) {
        const int b = bb + i * kFinRowLanes;
        if (b >= b1) break;
        if (SHIFTED) {
          int64_t n = M - (int64_t)b * rows_per_block;
          if (n > rows_per_block) n = rows_per_block;
          if (n <= 0) continue;
          const double d0 = (double)sh[i].x - (double)g0[0], d1 = (double)sh[i].y - (double)g0[1];
          a1[0] += (double)p[i].x + (double)n * d0;
          a2[0] += (double)p[i].y + 2.0 * d0 * (double)p[i].x + (double)n * d0 * d0;
          a1[1] += (double)p[i].z + (double)n * d1;
          a2[1] += (double)p[i].w + 2.0 * d1 * (double)p[i].z + (double)n * d1 * d1;
        } else {
          a1[0] += (double)p[i].x; a2[0] += (double)p[i].y;
          a1[1] += (double)p[i].z; a2[1] += (double)p[i].w;
        }
      }
    }
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      a1[e] += __shfl_xor(a1[e], o, 64);
      a2[e] += __shfl_xor(a2[e], o, 64);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 4) { red[wave][lane][0] = a1[0]; red[wave][lane][1] = a2[0]; red[wave][lane][2] = a1[1]; red[wave][lane][3] = a2[1]; }
  __syncthreads();
  t1[0] = t1[1] = t2[0] = t2[1] = 0.0;
  if (threadIdx.x < 4) {
    for (int w = 0; w < kFinThreads / 64; ++w) {
      t1[0] += red[w][threadIdx.x][0]; t2[0] += red[w][threadIdx.x][1];
      t1[1] += red[w][threadIdx.x][2]; t2[1] += red[w][threadIdx.x][3];
    }
  }
}

__device__ __forceinline__ void fin_st8(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fin_ld8(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Grid (C / 8, nseg).  Returns true in the threads that hold the totals of the WHOLE slab for channels c, c + 1:
// threads 0..3 of the only block (nseg == 1) or of the last block of the channel group to arrive.
template <bool SHIFTED>
__device__ __forceinline__ bool slab_totals(const float* __restrict__ partial, int nblocks, int C, int64_t M,
                                            int rows_per_block, int seg_rows, double* __restrict__ scratch,
                                            int* __restrict__ counters, int c, double (&t1)[2], double (&t2)[2],
                                            float (&g0)[2]) {
  const int nseg = gridDim.y;
  const int b0 = blockIdx.y * seg_rows;
  const int b1 = (b0 + seg_rows) < nblocks ? (b0 + seg_rows) : nblocks;
  combine_slab<SHIFTED>(partial, nblocks, C, M, rows_per_block, c, c < C, b0, b1, t1, t2, g0);
  const bool own = threadIdx.x < 4 && c < C;
  if (nseg == 1) return own;
  if (threadIdx.x >= 64) return false;                   // wave 0 carries the hand-off
  if (own) {
    double* s = scratch + ((int64_t)blockIdx.y * C + c) * 2;
    fin_st8(s, t1[0]); fin_st8(s + 1, t2[0]); fin_st8(s + 2, t1[1]); fin_st8(s + 3, t2[1]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the totals are written through before the flag moves
  int old = 0;
  if (threadIdx.x == 0) old = __hip_atomic_fetch_add(counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != nseg - 1) return false;
  if (threadIdx.x == 0) __hip_atomic_store(counters + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!own) return false;
  t1[0] = t1[1] = t2[0] = t2[1] = 0.0;
  for (int s0 = 0; s0 < nseg; s0 += 8) {                 // loads of 8 segments in flight, added in segment order
    double v[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sg = (s0 + i) < nseg ? (s0 + i) : nseg - 1;
      const double* s = scratch + ((int64_t)sg * C + c) * 2;
      v[i][0] = fin_ld8(s); v[i][1] = fin_ld8(s + 1); v[i][2] = fin_ld8(s + 2); v[i][3] = fin_ld8(s + 3);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (s0 + i >= nseg) break;
      t1[0] += v[i][0]; t2[0] += v[i][1]; t1[1] += v[i][2]; t2[1] += v[i][3];
    }
  }
  return true;
}

__global__ void __launch_bounds__(kFinThreads) bn_finalize_kernel(
    const float* __restrict__ partial, int nblocks, int64_t M, int C, int rows_per_block, int seg_rows,
    double* __restrict__ scratch, int* __restrict__ counters,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, float momentum, float eps, float* __restrict__ mean,
    float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * kFinCh + (threadIdx.x & 3) * 2;
  // the per-channel parameters travel with the slab loads (one round trip), not behind the reduction: under load
  // every dependent round trip of this latency-bound launch costs microseconds of the step's main chain
  float ga[2] = {0.f, 0.f}, be[2] = {0.f, 0.f}, rm[2] = {0.f, 0.f}, rv[2] = {0.f, 0.f};
  if (threadIdx.x < 4 && c < C) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ga[e] = gamma[c + e]; be[e] = beta[c + e];
      if (rmean) { rm[e] = rmean[c + e]; rv[e] = rvar[c + e]; }
    }
  }
  double t1[2], t2[2];
  float g0[2];
  if (!slab_totals<true>(partial, nblocks, C, M, rows_per_block, seg_rows, scratch, counters, c, t1, t2, g0)) return;
  const double inv_m = 1.0 / (double)M;             // one fp64 division instead of three
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const double dm = t1[e] * inv_m;                // mean - g0
    const double mu = (double)g0[e] + dm;
    double var = t2[e] * inv_m - dm * dm;           // biased; centred on a sample value: no cancellation
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c + e] = (float)mu;
    invstd[c + e] = is;
    const float sc = ga[e] * is;
    scale[c + e] = sc;
    shift[c + e] = be[e] - (float)mu * sc;
    if (rmean) {
      rmean[c + e] = momentum * rm[e] + (1.0f - momentum) * (float)mu;
      rvar[c + e] = momentum * rv[e] + (1.0f - momentum) * (float)var;
    }
  }
}

__global__ void __launch_bounds__(kFinThreads) bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int nblocks, int64_t M, int C, int seg_rows,
    double* __restrict__ scratch, int* __restrict__ counters,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ coef) {
  const int c = blockIdx.x * kFinCh + (threadIdx.x & 3) * 2;
  float ga[2] = {0.f, 0.f}, mu[2] = {0.f, 0.f}, is[2] = {0.f, 0.f}, dg[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
  if (threadIdx.x < 4 && c < C) {          // with the slab loads, not behind the reduction (bn_finalize_kernel)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ga[e] = gamma[c + e]; mu[e] = mean[c + e]; is[e] = invstd[c + e]; dg[e] = dgamma[c + e]; db[e] = dbeta[c + e];
    }
  }
  double sg[2], sgx[2];
  float unused[2];
  if (!slab_totals<false>(partial, nblocks, C, M, 0, seg_rows, scratch, counters, c, sg, sgx, unused)) return;
  const double inv_m = 1.0 / (double)M;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    dbeta[c + e] = db[e] + (float)sg[e];      // accumulate: the flat gradient buffer is zeroed by clear_grad()
    dgamma[c + e] = dg[e] + (float)sgx[e];
    // dx = gamma*invstd*( g - sg/M - xhat*sgx/M ),  xhat = (x-mean)*invstd
    const double isd = (double)is[e];
    const double gi = (double)ga[e] * isd;
    const double B = -gi * isd * sgx[e] * inv_m;
    const double Cc = -gi * sg[e] * inv_m - B * (double)mu[e];
    coef[c + e] = (float)gi;
    coef[C + c + e] = (float)B;
    coef[2 * C + c + e] = (float)Cc;
  }
}

// ------------------------------------------------------------------ cross-rank (Sync) BatchNorm pieces
// SyncBatchNorm (reference passl/models/simsiam.py:160-162: nn.SyncBatchNorm.convert_sync_batchnorm under data
// parallelism) = BatchNorm over the batches of ALL ranks.  Each rank folds its slab to fp64 moments
// {mean, M2 = sum (x - mean)^2, n}; the moments of all ranks are gathered (3 C doubles per rank) and combined in RANK
// ORDER with Chan's update (deterministic, no cancellation); backward the same way with {sum g, sum g xhat}.
__global__ void __launch_bounds__(kFinThreads) bn_moments_kernel(
    const float* __restrict__ partial, int nblocks, int64_t M, int C, int rows_per_block, int seg_rows,
    double* __restrict__ scratch, int* __restrict__ counters, double* __restrict__ mom) {
  const int c = blockIdx.x * kFinCh + (threadIdx.x & 3) * 2;
  double t1[2], t2[2];
  float g0[2];
  if (!slab_totals<true>(partial, nblocks, C, M, rows_per_block, seg_rows, scratch, counters, c, t1, t2, g0)) return;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const double dm = t1[e] / (double)M;
    double m2 = t2[e] - t1[e] * dm;                  // sum (x - mean)^2, centred on a sample value first
    if (m2 < 0.0) m2 = 0.0;
    mom[c + e] = (double)g0[e] + dm;
    mom[C + c + e] = m2;
    mom[2 * C + c + e] = (double)M;
  }
}

__global__ void __launch_bounds__(256) bn_finalize_moments_kernel(
    const double* __restrict__ mom_all, int world, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
    float eps, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
    float* __restrict__ shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, mu = 0.0, m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* m = mom_all + (int64_t)r * 3 * C;
    const double nb = m[2 * C + c], mb = m[c], qb = m[C + c];
    if (nb <= 0.0) continue;
    const double nn = n + nb, d = mb - mu;
    m2 += qb + d * d * n * nb / nn;
    mu += d * nb / nn;
    n = nn;
  }
  const double var = n > 0.0 ? m2 / n : 0.0;         // biased
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mu * sc;
  if (rmean) {
    rmean[c] = momentum * rmean[c] + (1.0f - momentum) * (float)mu;
    rvar[c] = momentum * rvar[c] + (1.0f - momentum) * (float)var;
  }
}

__global__ void __launch_bounds__(kFinThreads) bn_bwd_sums_kernel(
    const float* __restrict__ partial, int nblocks, int64_t M, int C, int seg_rows, double* __restrict__ scratch,
    int* __restrict__ counters, double* __restrict__ sums) {
  const int c = blockIdx.x * kFinCh + (threadIdx.x & 3) * 2;
  double sg[2], sgx[2];
  float unused[2];
  if (!slab_totals<false>(partial, nblocks, C, M, 0, seg_rows, scratch, counters, c, sg, sgx, unused)) return;
  sums[c] = sg[0]; sums[c + 1] = sg[1];
  sums[C + c] = sgx[0]; sums[C + c + 1] = sgx[1];
}

// dgamma / dbeta accumulate THIS rank's sums (the data-parallel reducer averages parameter gradients afterwards);
// the input-gradient coefficients use the totals over all ranks and the global row count
__global__ void __launch_bounds__(256) bn_bwd_finalize_sums_kernel(
    const double* __restrict__ sums_all, int world, int rank, double m_total, int C,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double sg = 0.0, sgx = 0.0;
  for (int r = 0; r < world; ++r) {
    sg += sums_all[(int64_t)r * 2 * C + c];
    sgx += sums_all[(int64_t)r * 2 * C + C + c];
  }
  dbeta[c] += (float)sums_all[(int64_t)rank * 2 * C + c];
  dgamma[c] += (float)sums_all[(int64_t)rank * 2 * C + C + c];
  const double gi = (double)gamma[c] * (double)invstd[c];
  const double inv_m = 1.0 / m_total;
  const double B = -gi * (double)invstd[c] * sgx * inv_m;
  const double Cc = -gi * sg * inv_m - B * (double)mean[c];
  coef[c] = (float)gi;
  coef[C + c] = (float)B;
  coef[2 * C + c] = (float)Cc;
}

// z = relu?(x*scale + shift + res)
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_apply_kernel(const T* __restrict__ x,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const T* __restrict__ res,
                                                            T* __restrict__ z,
                                                            uint8_t* __restrict__ mask,
                                                            int64_t nchunks, int cols, int relu) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nchunks; i += stride) {
    const int col = (int)(i % cols);
    float v[8];
    ElemTraits<T>::load8(x + i * 8, v);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + col * 8);
    const float4 s1 = *reinterpret_cast<const float4*>(scale + col * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(shift + col * 8);
    const float4 b1 = *reinterpret_cast<const float4*>(shift + col * 8 + 4);
    v[0] = v[0] * s0.x + b0.x; v[1] = v[1] * s0.y + b0.y;
    v[2] = v[2] * s0.z + b0.z; v[3] = v[3] * s0.w + b0.w;
    v[4] = v[4] * s1.x + b1.x; v[5] = v[5] * s1.y + b1.y;
    v[6] = v[6] * s1.z + b1.z; v[7] = v[7] * s1.w + b1.w;
    if (res) {
      float r[8];
      ElemTraits<T>::load8(res + i * 8, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (relu) {
      if (mask) {
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (v[e] > 0.f ? 1u : 0u) << e;
        mask[i] = (uint8_t)bits;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    ElemTraits<T>::store8(z + i * 8, v);
  }
}

// dx = A*g + B*x + Cc ; dres = g
template <typename T>
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_kernel(
    const T* __restrict__ dz, const void* __restrict__ z, const T* __restrict__ x,
    const float* __restrict__ coef, const float* __restrict__ scale,
    const float* __restrict__ shift, T* __restrict__ dx, T* __restrict__ dres, int64_t nchunks,
    int cols, int C, int relu) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nchunks; i += stride) {
    const int col = (int)(i % cols);
    float g[8], v[8];
    ElemTraits<T>::load8(dz + i * 8, g);
    ElemTraits<T>::load8(x + i * 8, v);
    if (relu == 2) {
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = scale[col * 8 + e]; sh[e] = shift[col * 8 + e]; }
      apply_relu_mask<T>(g, v, 2, nullptr, i, sc, sh);
    } else if (relu) {
      apply_relu_mask<T>(g, v, relu, z, i, nullptr, nullptr);
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col * 8 + e;
      o[e] = coef[c] * g[e] + coef[C + c] * v[e] + coef[2 * C + c];
    }
    ElemTraits<T>::store8(dx + i * 8, o);
    if (dres) ElemTraits<T>::store8(dres + i * 8, g);
  }
}

// ---- streaming kernels, tile form (the default when 256 % (C/8) == 0, i.e. every power-of-two C up
// to 2048).  One workgroup owns U * 256 CONSECUTIVE 16-byte chunks, a lane takes chunks
// base + u * 256: every lane keeps ONE channel group (per-channel constants loaded once, ahead of the
// data), the U data loads of a lane are issued back to back (U x 16 B in flight per lane instead of
// one load per grid-stride iteration), and there is no loop.  g_stream_variant: 0 = the grid-stride
// kernels above, U = 2 / 4 / 8 otherwise (passl_hip_set_option("bn_stream_unroll", U)).
template <typename T, int U>
__global__ void __launch_bounds__(kThreads) bn_apply_tile_kernel(const T* __restrict__ x,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const T* __restrict__ res,
                                                                 T* __restrict__ z,
                                                                 uint8_t* __restrict__ mask,
                                                                 int64_t nchunks, int cols, int relu) {
  const int64_t base = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
  const int col = (int)base & (cols - 1);      // cols divides 256: a power of two
  const float4 s0 = *reinterpret_cast<const float4*>(scale + col * 8);
  const float4 s1 = *reinterpret_cast<const float4*>(scale + col * 8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(shift + col * 8);
  const float4 b1 = *reinterpret_cast<const float4*>(shift + col * 8 + 4);
  const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  const float sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  // branch-free loads (a lane past the end re-reads the last chunk): all U loads are in flight together
  float v[U][8], r[U][8];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kThreads;
    ElemTraits<T>::load8(x + (i < nchunks ? i : nchunks - 1) * 8, v[u]);
  }
  if (res) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * kThreads;
      ElemTraits<T>::load8(res + (i < nchunks ? i : nchunks - 1) * 8, r[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kThreads;
    if (i >= nchunks) break;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[u][e] = v[u][e] * sc[e] + sh[e];
    if (res) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] += r[u][e];
    }
    if (relu) {
      if (mask) {
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (v[u][e] > 0.f ? 1u : 0u) << e;
        mask[i] = (uint8_t)bits;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = fmaxf(v[u][e], 0.f);
    }
    ElemTraits<T>::store8(z + i * 8, v[u]);
  }
}

template <typename T, int U>
__global__ void __launch_bounds__(kThreads) bn_bwd_apply_tile_kernel(
    const T* __restrict__ dz, const void* __restrict__ z, const T* __restrict__ x,
    const float* __restrict__ coef, const float* __restrict__ scale,
    const float* __restrict__ shift, T* __restrict__ dx, T* __restrict__ dres, int64_t nchunks,
    int cols, int C, int relu) {
  const int64_t base = (int64_t)blockIdx.x * (kThreads * U) + threadIdx.x;
  const int col = (int)base & (cols - 1);      // cols divides 256: a power of two
  float cA[8], cB[8], cC[8], sc[8], sh[8];
  {
    const float4* ca = reinterpret_cast<const float4*>(coef + col * 8);
    const float4* cb = reinterpret_cast<const float4*>(coef + C + col * 8);
    const float4* cc = reinterpret_cast<const float4*>(coef + 2 * C + col * 8);
    const float4 a0 = ca[0], a1 = ca[1], b0 = cb[0], b1 = cb[1], c0 = cc[0], c1 = cc[1];
    cA[0] = a0.x; cA[1] = a0.y; cA[2] = a0.z; cA[3] = a0.w; cA[4] = a1.x; cA[5] = a1.y; cA[6] = a1.z; cA[7] = a1.w;
    cB[0] = b0.x; cB[1] = b0.y; cB[2] = b0.z; cB[3] = b0.w; cB[4] = b1.x; cB[5] = b1.y; cB[6] = b1.z; cB[7] = b1.w;
    cC[0] = c0.x; cC[1] = c0.y; cC[2] = c0.z; cC[3] = c0.w; cC[4] = c1.x; cC[5] = c1.y; cC[6] = c1.z; cC[7] = c1.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = relu == 2 ? scale[col * 8 + e] : 0.f;
    sh[e] = relu == 2 ? shift[col * 8 + e] : 0.f;
  }
  float g[U][8], v[U][8];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kThreads;
    const int64_t ic = i < nchunks ? i : nchunks - 1;
    ElemTraits<T>::load8(dz + ic * 8, g[u]);
    ElemTraits<T>::load8(x + ic * 8, v[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kThreads;
    if (i >= nchunks) break;
    if (relu) apply_relu_mask<T>(g[u], v[u], relu, z, i, sc, sh);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = cA[e] * g[u][e] + cB[e] * v[u][e] + cC[e];
    ElemTraits<T>::store8(dx + i * 8, o);
    if (dres) ElemTraits<T>::store8(dres + i * 8, g[u]);
  }
}

static int g_stream_unroll = -1;
static inline int stream_unroll() {
  if (g_stream_unroll < 0) {
    const char* e = getenv("PASSL_BN_STREAM_UNROLL");
    g_stream_unroll = e ? atoi(e) : 4;
  }
  return g_stream_unroll;
}

static inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

// passl_hip_set_option("bn_stream_unroll", 0 | 2 | 4 | 8)   (runtime.hip dispatches)
int passl_bn_option(const char* name, int value) {
  if (!strcmp(name, "bn_stream_unroll")) {
    if (value != 0 && value != 2 && value != 4 && value != 8) return PASSL_EINVAL;
    g_stream_unroll = value;
    return PASSL_OK;
  }
  return PASSL_EINVAL;
}

#define DISPATCH_DTYPE(dtype, ...)                          \
  if ((dtype) == PASSL_BF16) { using T = bf16_t; __VA_ARGS__ } \
  else if ((dtype) == PASSL_F32) { using T = float; __VA_ARGS__ } \
  else return PASSL_EUNSUPPORTED;

// row segments of a slab: one per kFinSegRows* rows (one round trip of loads per block), at most kFinSegMax
static int fin_segments(int nblocks, bool shifted, int* seg_rows) {
  const int per = shifted ? kFinSegRowsFwd : kFinSegRowsBwd;
  int nseg = (nblocks + per - 1) / per;
  if (nseg > kFinSegMax) nseg = kFinSegMax;
  *seg_rows = (nblocks + nseg - 1) / nseg;
  return (nblocks + *seg_rows - 1) / *seg_rows;
}

// Arrival counters of the multi-segment finalize launches: a slice (one int per 8-channel group) of a library-owned
// ring of ints that is ALL ZERO whenever no launch is using a slice — the block that completes a counter re-zeroes it.
// A slice is handed out again after 64 K ints of later requests (a MoCo step asks for ~2 K).
static int* fin_counters(int groups) {
  // one pool per device (a pointer of another device's pool would be foreign memory: round-5 advisor finding); zeroed
  // with a synchronous memset FOLLOWED by a device synchronisation, so that no launch on any (non-blocking) stream can
  // run before the zeros are in place; a failed allocation / memset leaves no half-initialised pool behind
  constexpr int kMaxDev = 16;
  constexpr int64_t kPoolInts = 1 << 16;
  static std::mutex mu;
  static int* pool[kMaxDev] = {};
  static int64_t next[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!pool[dev]) {
    int* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), kPoolInts * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, kPoolInts * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    pool[dev] = p;
    next[dev] = 0;
  }
  if (next[dev] + groups > kPoolInts) next[dev] = 0;
  int* r = pool[dev] + next[dev];
  next[dev] += (groups + 3) & ~3;
  return r;
}

extern "C" int64_t passl_hip_bn_partial_floats(int nblocks, int C, int shifted) {
  if (nblocks <= 0 || C <= 0) return 0;
  return (int64_t)nblocks * C * (shifted ? 3 : 2) + (int64_t)kFinSegMax * C * 4;
}

extern "C" int passl_hip_bn_stats(const void* x, float* partial, int64_t M, int C, int nblocks,
                                  int dtype, passl_stream_t stream) {
  if (!x || !partial || M <= 0 || C <= 0 || (C & 7) || nblocks <= 0 || !aligned16(x))
    return PASSL_EINVAL;
  const int rows = (int)((M + nblocks - 1) / nblocks);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T, 0>), dim3(nblocks), dim3(kThreads),
                                           kThreads * 16 * sizeof(float), as_stream(stream),
                                           reinterpret_cast<const T*>(x), nullptr, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, partial, M, C, rows, 0);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_finalize(const float* partial, int nblocks, int64_t M, int C,
                                     int rows_per_block, const float* gamma, const float* beta,
                                     float* running_mean,
                                     float* running_var, float momentum, float eps, float* mean,
                                     float* invstd, float* scale, float* shift,
                                     passl_stream_t stream) {
  if (!partial || !gamma || !beta || !mean || !invstd || !scale || !shift || M <= 0 || C <= 0 ||
      (C & 7) || nblocks <= 0 || rows_per_block <= 0 || (int64_t)nblocks * rows_per_block < M ||
      (running_mean && !running_var) || !aligned16(partial))
    return PASSL_EINVAL;
  int seg_rows = 0;
  const int nseg = fin_segments(nblocks, true, &seg_rows);
  // the segment scratch lives behind the slab (passl_hip_bn_partial_floats sizes the buffer)
  double* scratch = reinterpret_cast<double*>(const_cast<float*>(partial) + (int64_t)nblocks * C * 3);
  int* counters = nseg > 1 ? fin_counters(C / kFinCh) : nullptr;
  if (nseg > 1 && !counters) return PASSL_ELAUNCH;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / kFinCh, nseg), dim3(kFinThreads), 0, as_stream(stream),
                     partial, nblocks, M, C, rows_per_block, seg_rows, scratch, counters, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, invstd, scale, shift);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_apply(const void* x, const float* scale, const float* shift,
                                  const void* residual, void* z, uint8_t* relu_mask, int64_t M,
                                  int C, int relu, int dtype, passl_stream_t stream) {
  if (!x || !scale || !shift || !z || M <= 0 || C <= 0 || (C & 7) || !aligned16(x) ||
      !aligned16(z) || (residual && !aligned16(residual)) || !aligned16(scale) || !aligned16(shift))
    return PASSL_EINVAL;
  if (relu_mask && !relu) return PASSL_EINVAL;
  const int64_t nchunks = M * (C >> 3);
  const int U = (kThreads % (C >> 3)) == 0 ? stream_unroll() : 0;
#define PASSL_BN_APPLY_TILE(UU)                                                                          \
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_apply_tile_kernel<T, UU>),                               \
                                           dim3((unsigned)((nchunks + kThreads * UU - 1) / (kThreads * UU))), \
                                           dim3(kThreads), 0, as_stream(stream),                        \
                                           reinterpret_cast<const T*>(x), scale, shift,                 \
                                           reinterpret_cast<const T*>(residual),                        \
                                           reinterpret_cast<T*>(z), relu_mask, nchunks, C >> 3, relu);)
  if (U == 2) { PASSL_BN_APPLY_TILE(2) }
  else if (U == 4) { PASSL_BN_APPLY_TILE(4) }
  else if (U == 8) { PASSL_BN_APPLY_TILE(8) }
  else {
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(grid_for(nchunks)),
                                             dim3(kThreads), 0, as_stream(stream),
                                             reinterpret_cast<const T*>(x), scale, shift,
                                             reinterpret_cast<const T*>(residual),
                                             reinterpret_cast<T*>(z), relu_mask, nchunks, C >> 3,
                                             relu);)
  }
#undef PASSL_BN_APPLY_TILE
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

static bool relu_args_ok(int relu, const void* z, const float* scale, const float* shift) {
  if (relu < 0 || relu > 3) return false;
  if (relu == 1 && (!z || !aligned16(z))) return false;
  if (relu == 2 && (!scale || !shift)) return false;
  if (relu == 3 && !z) return false;
  return true;
}

extern "C" int passl_hip_bn_bwd_reduce(const void* dz, const void* z, const void* x,
                                       const float* mean, const float* invstd,
                                       const float* scale, const float* shift, float* partial,
                                       int64_t M, int C, int nblocks, int relu, int dtype,
                                       passl_stream_t stream) {
  if (!dz || !x || !mean || !invstd || !partial || !relu_args_ok(relu, z, scale, shift) ||
      M <= 0 || C <= 0 || (C & 7) || nblocks <= 0 || !aligned16(dz) || !aligned16(x))
    return PASSL_EINVAL;
  const int rows = (int)((M + nblocks - 1) / nblocks);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_reduce_kernel<T, 1>), dim3(nblocks), dim3(kThreads),
                                           kThreads * 16 * sizeof(float), as_stream(stream),
                                           reinterpret_cast<const T*>(x),
                                           reinterpret_cast<const T*>(dz), z, mean, invstd, scale,
                                           shift, partial, M, C, rows, relu);)
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_bwd_finalize(const float* partial, int nblocks, int64_t M, int C,
                                         const float* gamma, const float* mean,
                                         const float* invstd, float* dgamma, float* dbeta,
                                         float* coef, passl_stream_t stream) {
  if (!partial || !gamma || !mean || !invstd || !dgamma || !dbeta || !coef || M <= 0 || C <= 0 ||
      (C & 7) || nblocks <= 0 || !aligned16(partial))
    return PASSL_EINVAL;
  int seg_rows = 0;
  const int nseg = fin_segments(nblocks, false, &seg_rows);
  double* scratch = reinterpret_cast<double*>(const_cast<float*>(partial) + (int64_t)nblocks * C * 2);
  int* counters = nseg > 1 ? fin_counters(C / kFinCh) : nullptr;
  if (nseg > 1 && !counters) return PASSL_ELAUNCH;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C / kFinCh, nseg), dim3(kFinThreads), 0, as_stream(stream),
                     partial, nblocks, M, C, seg_rows, scratch, counters, gamma, mean, invstd, dgamma, dbeta, coef);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

// ---- cross-rank BatchNorm (see the kernels above): mom / sums are caller-owned fp64 buffers
extern "C" int passl_hip_bn_moments(const float* partial, int nblocks, int64_t M, int C, int rows_per_block,
                                    double* mom, passl_stream_t stream) {
  if (!partial || !mom || M <= 0 || C <= 0 || (C & 7) || nblocks <= 0 || rows_per_block <= 0 ||
      (int64_t)nblocks * rows_per_block < M || !aligned16(partial))
    return PASSL_EINVAL;
  int seg_rows = 0;
  const int nseg = fin_segments(nblocks, true, &seg_rows);
  double* scratch = reinterpret_cast<double*>(const_cast<float*>(partial) + (int64_t)nblocks * C * 3);
  int* counters = nseg > 1 ? fin_counters(C / kFinCh) : nullptr;
  if (nseg > 1 && !counters) return PASSL_ELAUNCH;
  hipLaunchKernelGGL(bn_moments_kernel, dim3(C / kFinCh, nseg), dim3(kFinThreads), 0, as_stream(stream),
                     partial, nblocks, M, C, rows_per_block, seg_rows, scratch, counters, mom);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_finalize_moments(const double* mom_all, int world, int C, const float* gamma,
                                             const float* beta, float* running_mean, float* running_var,
                                             float momentum, float eps, float* mean, float* invstd,
                                             float* scale, float* shift, passl_stream_t stream) {
  if (!mom_all || !gamma || !beta || !mean || !invstd || !scale || !shift || world <= 0 || C <= 0 ||
      (running_mean && !running_var))
    return PASSL_EINVAL;
  hipLaunchKernelGGL(bn_finalize_moments_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), mom_all,
                     world, C, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_bwd_sums(const float* partial, int nblocks, int64_t M, int C, double* sums,
                                     passl_stream_t stream) {
  if (!partial || !sums || M <= 0 || C <= 0 || (C & 7) || nblocks <= 0 || !aligned16(partial))
    return PASSL_EINVAL;
  int seg_rows = 0;
  const int nseg = fin_segments(nblocks, false, &seg_rows);
  double* scratch = reinterpret_cast<double*>(const_cast<float*>(partial) + (int64_t)nblocks * C * 2);
  int* counters = nseg > 1 ? fin_counters(C / kFinCh) : nullptr;
  if (nseg > 1 && !counters) return PASSL_ELAUNCH;
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(C / kFinCh, nseg), dim3(kFinThreads), 0, as_stream(stream),
                     partial, nblocks, M, C, seg_rows, scratch, counters, sums);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_bwd_finalize_sums(const double* sums_all, int world, int rank, int64_t M_total, int C,
                                              const float* gamma, const float* mean, const float* invstd,
                                              float* dgamma, float* dbeta, float* coef, passl_stream_t stream) {
  if (!sums_all || !gamma || !mean || !invstd || !dgamma || !dbeta || !coef || world <= 0 || rank < 0 ||
      rank >= world || M_total <= 0 || C <= 0)
    return PASSL_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums_all,
                     world, rank, (double)M_total, C, gamma, mean, invstd, dgamma, dbeta, coef);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_bn_bwd_apply(const void* dz, const void* z, const void* x,
                                      const float* coef, const float* scale, const float* shift,
                                      void* dx, void* dres, int64_t M, int C, int relu, int dtype,
                                      passl_stream_t stream) {
  if (!dz || !x || !coef || !dx || !relu_args_ok(relu, z, scale, shift) || M <= 0 || C <= 0 ||
      (C & 7) || !aligned16(dz) || !aligned16(x) || !aligned16(dx) || (dres && !aligned16(dres)))
    return PASSL_EINVAL;
  const int64_t nchunks = M * (C >> 3);
  const int U = (kThreads % (C >> 3)) == 0 ? stream_unroll() : 0;
#define PASSL_BN_BWD_APPLY_TILE(UU)                                                                      \
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_bwd_apply_tile_kernel<T, UU>),                           \
                                           dim3((unsigned)((nchunks + kThreads * UU - 1) / (kThreads * UU))), \
                                           dim3(kThreads), 0, as_stream(stream),                        \
                                           reinterpret_cast<const T*>(dz), z,                           \
                                           reinterpret_cast<const T*>(x), coef, scale, shift,           \
                                           reinterpret_cast<T*>(dx), reinterpret_cast<T*>(dres),        \
                                           nchunks, C >> 3, C, relu);)
  if (U == 2) { PASSL_BN_BWD_APPLY_TILE(2) }
  else if (U == 4) { PASSL_BN_BWD_APPLY_TILE(4) }
  else if (U == 8) { PASSL_BN_BWD_APPLY_TILE(8) }
  else {
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid_for(nchunks)),
                                             dim3(kThreads), 0, as_stream(stream),
                                             reinterpret_cast<const T*>(dz), z,
                                             reinterpret_cast<const T*>(x), coef, scale, shift,
                                             reinterpret_cast<T*>(dx), reinterpret_cast<T*>(dres),
                                             nchunks, C >> 3, C, relu);)
  }
#undef PASSL_BN_BWD_APPLY_TILE
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

