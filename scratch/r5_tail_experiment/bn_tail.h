// In-kernel BatchNorm finalize ("tail") of the convolution epilogue (igemm_epi.h), gfx950.
//
// A convolution launch with fused BatchNorm statistics leaves one slab row per 128-row output tile; until round 4 two
// more launches (bn_combine + bn_finalize, or their backward twins) turned the slab into the per-channel constants the
// streaming pass needs.  Under load those two launches cost 11-15 us each plus two dependent-launch gaps ON THE MAIN
// CHAIN, 53 + 53 times per MoCo step (profiles/r04_bench_bs256_bf16_kernel_stats.txt: 1.9 ms of 24.2).  Here the
// launch finishes the job itself: every wave that has written its share of a slab row ARRIVES at a counter, and the last
// arriver of a group of 16 rows combines them — a tree with fan-in 16 whose root writes what bn_finalize /
// bn_bwd_finalize would have written.  No spinning, no grid barrier: only "who came last" is decided at run time, the
// summation order is fixed by the tree (bit-reproducible run to run, independent of the arrival order).
//
//  * a wave owns BN / waves columns of the tile: it folds the workgroup's per-thread partial sums for them (the same
//    fixed order as before), PUBLISHES the slab entries — 8-byte relaxed agent-scope atomic stores = write-through
//    `global_store_dwordx2 sc1`, the last one a returning swap whose value (VMEM operations of a wave complete in
//    order) acknowledges them all — then lane 0 adds 1 to the counter of its level-1 node (16 consecutive slab rows)
//    with a returning agent-scope atomic.  The tile's own output stores are issued in two halves AFTER the publish and
//    after the arrival: both round trips run under store issue, nothing waits for the store queue to drain;
//  * the wave that reads 15 back (16 arrivals; fewer in the ragged last node) re-zeroes the counter, reads the 16 rows
//    with 8-byte relaxed agent-scope atomic loads (`global_load_dwordx2 sc1`: served past the CU's L1, coherent per
//    variable across the XCDs' L2s — MI355X_MICROARCH.md "8-B agent atomics both sides"), lanes = 4 (or 8) children
//    groups x 16 (or 8) column pairs, folds the groups with xor shuffles, publishes the node total (fp64) the same way
//    and arrives one level up; the root finalizes.
//  * forward sums are SHIFTED (igemm_epi.h): a node is centred on the shift of its FIRST tile (a value every level can
//    read from the slab: no extra state), children are re-centred in fp64 exactly as bn_finalize does.
//  * counters come from a library-owned pool that is all zero between launches (every counter is re-zeroed by the wave
//    that completes it), node totals live behind the slab (passl_hip_bn_partial_floats sizes the buffer).
//
// Memory ordering: payload stores and the flag are all `sc1` write-through operations of ONE wave, the flag issued
// after the payload's acknowledgement has arrived; the consumer's loads are control-dependent on the value its own returning atomic brought back.  No
// fence (a release fence would write back the XCD's whole dirty L2: the output tile the workgroup has just stored).
#pragma once
#include <string.h>
#include <type_traits>
#include "common.h"

namespace bn_tail {

constexpr int kFan = 16;
constexpr uint32_t kYLimit = 0xfffffff0u;      // the deferred stores of the tile are buffer stores: bytes of y addressable

struct Tail {
  int* counters;        // nullptr = no tail.  [tiles_n * waves][cstride], all zero between launches
  double* totals;       // node totals of the levels below the root: [(level offset + node) * NCOLS + col][2]
  int tiles;            // slab rows, over ALL launches that share the slab (bnb_tile_off)
  int cstride;          // nodes(tiles): counters per (column tile, wave)
  long long rows;       // rows the statistics are over (all those launches)
  const float* gamma;
  const float* beta;    // forward
  float* rmean;         // forward, may be null
  float* rvar;
  float* out;           // forward: [4][NCOLS] mean, invstd, scale, shift.  backward: coef [3][NCOLS]
  float* dgamma;        // backward (accumulated into)
  float* dbeta;
  float momentum, eps;
};

// nodes of the tree over `tiles` leaves (levels 1 .. root)
static inline int nodes(int tiles) {
  int n = tiles, s = 0;
  do { n = (n + kFan - 1) / kFan; s += n; } while (n > 1);
  return s;
}

// descriptor -> Tail (host).  this_tiles / this_rows: slab rows and output rows of THIS launch.
static inline int fill(const passl_conv_desc* d, int this_tiles, long long this_rows, Tail& t) {
  memset(&t, 0, sizeof(t));
  if (!d->tail_counters) return PASSL_OK;
  const bool fwd = d->stats != nullptr;
  if (!fwd && !d->bnb_partial) return PASSL_EINVAL;
  if (!d->tail_gamma || !d->tail_out) return PASSL_EINVAL;
  {
    const int64_t last = (int64_t)(d->N - 1) * d->y_sn + (int64_t)(d->OP - 1) * d->y_sh + (int64_t)(d->OQ - 1) * d->y_sw + d->NCOLS;
    if (last * 2 >= (int64_t)kYLimit) return PASSL_EUNSUPPORTED;   // ask without the tail (separate finalize launches)
  }
  if (fwd ? (!d->tail_beta || (d->tail_rmean && !d->tail_rvar)) : (!d->tail_dgamma || !d->tail_dbeta)) return PASSL_EINVAL;
  t.tiles = d->tail_tiles > 0 ? d->tail_tiles : this_tiles;
  t.rows = d->tail_rows > 0 ? d->tail_rows : this_rows;
  if (t.tiles < this_tiles + (fwd ? 0 : d->bnb_tile_off) || t.rows < this_rows) return PASSL_EINVAL;
  t.counters = d->tail_counters;
  t.cstride = nodes(t.tiles);
  float* slab = fwd ? d->stats : d->bnb_partial;
  t.totals = reinterpret_cast<double*>(slab + (int64_t)t.tiles * d->NCOLS * (fwd ? 3 : 2));
  t.gamma = d->tail_gamma; t.beta = d->tail_beta; t.rmean = d->tail_rmean; t.rvar = d->tail_rvar;
  t.out = d->tail_out; t.dgamma = d->tail_dgamma; t.dbeta = d->tail_dbeta;
  t.momentum = d->tail_momentum; t.eps = d->tail_eps;
  return PASSL_OK;
}

__device__ __forceinline__ void st8(float* p, float a, float b) {
  const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long xchg8(float* p, float a, float b) {
  const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
  return __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld8(const float* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}
__device__ __forceinline__ void st8d(double* p, double a) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(a),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld8d(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Called by EVERY thread of the workgroup after the barrier that completes red[J][BN * 2] (the per-thread partial sums
// of the tile: igemm_epi.h).  c0 = the shifts of the calling thread's column chunk (tid % (BN / 8)); fwd: statistics
// of the forward pass (shifted sums + shifts), else the backward sums.  row = this tile's slab row, nt = column tile.
//
// The Tail is NOT read from the kernel's by-value parameter copy: the compiler would fetch its 100 bytes into scalar
// registers at kernel entry and keep them live through the main loop (+24 SGPRs, spills in the 8-phase and stem
// kernels).  It is read from the kernarg segment through a pointer the optimizer cannot see through until the tail
// starts (kernarg()), so the scalar loads are issued there.
typedef const __attribute__((address_space(4))) Tail* TailPtr;
template <typename P>
__device__ __forceinline__ TailPtr kernarg() {
  typedef const __attribute__((address_space(4))) char* cptr;
  cptr base = (cptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(P, tail);
  asm volatile("" : "+s"(base));
  return reinterpret_cast<TailPtr>(base);
}

template <int BN, int NTHREADS, typename P, typename STORES>
__device__ __forceinline__ void run(const P& p, TailPtr tp, const float* red, const float (&c0)[8], bool fwd,
                                    int n0, int nt, int row, int tid, STORES&& stores) {
  const int t_tiles = tp->tiles;
  const long long t_rows = tp->rows;
  double* const t_totals = tp->totals;
  constexpr int NW = NTHREADS / 64;
  constexpr int CW = BN / NW;              // columns of a wave: 16 (BN 64) or 32 (128 / 4 waves, 256 / 8 waves)
  constexpr int PP = CW / 2;               // column pairs of a wave
  constexpr int SUB = 64 / PP;             // lanes per pair in the combine: 8 or 4
  constexpr int CPL = kFan / SUB;          // children per lane: 2 or 4
  constexpr int CPR = BN / 8;
  constexpr int J = NTHREADS / CPR;
  static_assert(CW == 16 || CW == 32, "wave owns 16 or 32 columns");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar registers
  const int pair = lane % PP, sub = lane / PP;
  const int colw = wave * CW;
  const int col = n0 + colw + pair * 2;
  const bool col_ok = col < p.NCOLS;
  const int NC = p.NCOLS;
  float* slab = fwd ? p.stats : p.bnb_partial;
  float* shifts = slab + (int64_t)t_tiles * NC * 2;

  // ---- this wave's share of the slab row, published write-through.  The LAST publish operation of the wave is a
  // returning swap: VMEM operations of a wave complete in order, so its return value means every publish store before
  // it has been acknowledged — a data dependency the compiler turns into a COUNTED wait (the tile's stores issued
  // after it stay in flight), where a bare s_waitcnt vmcnt(0) would drain the whole store queue of a write-bound
  // kernel (measured: +4 us per workgroup).
  if (fwd && lane < CPR && lane * 8 >= colw && lane * 8 < colw + CW && n0 + lane * 8 < NC) {
    float* sp = shifts + (int64_t)row * NC + n0 + lane * 8;
    st8(sp, c0[0], c0[1]); st8(sp + 2, c0[2], c0[3]); st8(sp + 4, c0[4], c0[5]); st8(sp + 6, c0[6], c0[7]);
  }
  unsigned long long ack = 0;
  if (sub == 0 && col_ok) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(red + j * (BN * 2) + (colw + pair * 2) * 2);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float* o = slab + ((int64_t)row * NC + col) * 2;
    st8(o, a.x, a.y);
    ack = xchg8(o + 2, a.z, a.w);
  }
  asm volatile("" ::: "memory");
  stores(0);                               // first half of the tile's stores: in flight under the publish
  int one;
  // = 1, once `ack` has arrived ("memory": not before the stores above have been issued)
  asm volatile("v_or_b32 %0, %1, %2\n\tv_and_b32 %0, 0, %0\n\tv_add_u32 %0, 1, %0"
               : "=v"(one) : "v"((int)(ack >> 32)), "v"((int)ack) : "memory");

  // ---- arrive; the last arriver of a node combines its children and arrives one level up.  The first level (children
  // = slab rows) is straight-line code, the levels above it a loop: no loop-header join between the tile's stores and
  // the waits that count them.
  int* cnt = tp->counters + (int64_t)(nt * NW + wave) * tp->cstride;
  const __amdgpu_buffer_rsrc_t rs_cnt = __builtin_amdgcn_make_buffer_rsrc(cnt, 0, (uint32_t)tp->cstride * 4u, 0x00020000);
  int idx = row, n = t_tiles, off = 0, off_prev = 0, span = 1;
  auto level = [&](auto leaves_c) __attribute__((always_inline)) -> bool {      // true: arrive one level up
    constexpr bool LEAVES = decltype(leaves_c)::value;
    const int node = idx / kFan, first = node * kFan;
    const int kids = (n - first) < kFan ? (n - first) : kFan;
    // lane 0 adds, the other lanes point beyond the buffer (dropped, return 0): no branch around the atomic, so the
    // wait for its return value is a counted one
    const int slot = off + node;
    int old = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(one, rs_cnt, lane == 0 ? slot * 4 : (int)0x7ffffff0, 0, 0);
    if constexpr (LEAVES) {                // second half of the tile's stores: in flight under the arrival
      asm volatile("" ::: "memory");
      stores(1);
      asm volatile("" ::: "memory");
    }
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != kids - 1) return false;
    if (lane == 0) __hip_atomic_store(cnt + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    float2 g = make_float2(0.f, 0.f);                      // the node's centre: the shift of its first tile
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;     // [column of the pair][sum, sum of squares]
    if (col_ok) {
      if (fwd) g = ld8(shifts + (int64_t)first * span * NC + col);
      double t00[CPL], t01[CPL], t10[CPL], t11[CPL];
      float2 sh[CPL];
      bool have[CPL];
      // all loads first, unconditionally (a child beyond the node re-reads the last one and is ignored below): one
      // round trip, and no branch between the loads
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const int child = first + sub * CPL + k;
        have[k] = child < first + kids;
        const int cc = have[k] ? child : first + kids - 1;
        if constexpr (LEAVES) {
          const float* s = slab + ((int64_t)cc * NC + col) * 2;
          const float2 v0 = ld8(s), v1 = ld8(s + 2);
          t00[k] = (double)v0.x; t01[k] = (double)v0.y; t10[k] = (double)v1.x; t11[k] = (double)v1.y;
        } else {
          const double* s = t_totals + ((int64_t)(off_prev + cc) * NC + col) * 2;
          t00[k] = ld8d(s); t01[k] = ld8d(s + 1); t10[k] = ld8d(s + 2); t11[k] = ld8d(s + 3);
        }
        sh[k] = make_float2(0.f, 0.f);
      }
      if (fwd) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
          const int child = first + sub * CPL + k;
          const int cc = child < first + kids ? child : first + kids - 1;
          sh[k] = ld8(shifts + (int64_t)cc * span * NC + col);
        }
      }
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        if (!have[k]) continue;
        if (fwd) {
          const int64_t ft = (int64_t)(first + sub * CPL + k) * span;
          int64_t rows = (int64_t)t_rows - ft * 128;
          if (rows > (int64_t)span * 128) rows = (int64_t)span * 128;
          const double nr = (double)rows;
          const double d0 = (double)sh[k].x - (double)g.x, d1 = (double)sh[k].y - (double)g.y;
          a00 += t00[k] + nr * d0; a01 += t01[k] + 2.0 * d0 * t00[k] + nr * d0 * d0;
          a10 += t10[k] + nr * d1; a11 += t11[k] + 2.0 * d1 * t10[k] + nr * d1 * d1;
        } else {
          a00 += t00[k]; a01 += t01[k]; a10 += t10[k]; a11 += t11[k];
        }
      }
    }
#pragma unroll
    for (int o = PP; o < 64; o <<= 1) {
      a00 += __shfl_xor(a00, o, 64); a01 += __shfl_xor(a01, o, 64);
      a10 += __shfl_xor(a10, o, 64); a11 += __shfl_xor(a11, o, 64);
    }
    const int nn = (n + kFan - 1) / kFan;
    if (nn == 1) {
      if (sub != 0 || !col_ok) return false;
      Tail t;
      t.gamma = tp->gamma; t.beta = tp->beta; t.rmean = tp->rmean; t.rvar = tp->rvar; t.out = tp->out;
      t.dgamma = tp->dgamma; t.dbeta = tp->dbeta; t.momentum = tp->momentum; t.eps = tp->eps;
      const double inv_m = 1.0 / (double)t_rows;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = col + e;
        const double t1 = e ? a10 : a00, t2 = e ? a11 : a01;
        if (fwd) {                                          // = bn_finalize_kernel (bn.hip)
          const double dm = t1 * inv_m;
          const double mu = (double)(e ? g.y : g.x) + dm;
          double var = t2 * inv_m - dm * dm;
          if (var < 0.0) var = 0.0;
          const float is = (float)(1.0 / sqrt(var + (double)t.eps));
          t.out[c] = (float)mu;
          t.out[NC + c] = is;
          const float sc = t.gamma[c] * is;
          t.out[2 * NC + c] = sc;
          t.out[3 * NC + c] = t.beta[c] - (float)mu * sc;
          if (t.rmean) {
            t.rmean[c] = t.momentum * t.rmean[c] + (1.0f - t.momentum) * (float)mu;
            t.rvar[c] = t.momentum * t.rvar[c] + (1.0f - t.momentum) * (float)var;
          }
        } else {                                            // = bn_bwd_finalize_kernel (bn.hip)
          t.dbeta[c] += (float)t1;
          t.dgamma[c] += (float)t2;
          const double is = (double)p.bnb_invstd[c];
          const double gi = (double)t.gamma[c] * is;
          const double B = -gi * is * t2 * inv_m;
          const double Cc = -gi * t1 * inv_m - B * (double)p.bnb_mean[c];
          t.out[c] = (float)gi;
          t.out[NC + c] = (float)B;
          t.out[2 * NC + c] = (float)Cc;
        }
      }
      return false;
    }
    unsigned long long ack2 = 0;
    if (sub == 0 && col_ok) {
      double* s = t_totals + ((int64_t)(off + node) * NC + col) * 2;
      st8d(s, a00); st8d(s + 1, a01); st8d(s + 2, a10);
      ack2 = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(s + 3), (unsigned long long)__double_as_longlong(a11),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // = 1, once the node total is acknowledged
    asm volatile("v_or_b32 %0, %1, %2\n\tv_and_b32 %0, 0, %0\n\tv_add_u32 %0, 1, %0"
                 : "=v"(one) : "v"((int)(ack2 >> 32)), "v"((int)ack2) : "memory");
    idx = node; n = nn; off_prev = off; off += nn; span *= kFan;
    return true;
  };
  if (!level(std::true_type{})) return;
  while (level(std::false_type{})) {}
}

}  // namespace bn_tail
