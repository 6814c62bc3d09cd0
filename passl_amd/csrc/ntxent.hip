// SimCLR head on gfx950: fused NT-Xent + CO2 consistency loss (forward and backward) over the
// all-pairs similarities of two L2-normalised embedding sets, fp32 throughout
// (v_mfma_f32_16x16x4_f32, exact fp32).
//
// Reference semantics: passl_v110/modeling/heads/simclr_contrastive_head.py:42-102.  For local
// rows i (a_i = hidden1, b_i = hidden2) and columns j of the "large" sets A, B (A = a, B = b on
// one rank; the all-gathered embeddings in the multi-rank extension, where row i's positive
// column is roff + i):
//     aa = a.A^T/T, ab = a.B^T/T, ba = b.A^T/T, bb = b.B^T/T        (aa, bb self-masked)
//     loss_i = LSE([ab_i | aa_i]) - ab_i,pos + LSE([ba_i | bb_i]) - ba_i,pos
//     CO2    = KL(Pb || Pa) + KL(Pa || Pb),  Pa = softmax([aa_i | ab_i]), Pb = softmax([ba_i | bb_i])
//              (positive column masked in all four), weight co2_weight (= 3 in the reference)
//     loss   = mean_i loss_i + co2_weight * sum_i CO2_i / B ;   acc1 = mean_i [argmax_j ab_ij == pos]
// The B x BL logit matrices never reach HBM.  Data flow (forward): one workgroup owns 16 rows;
// its 4 waves sweep the column tiles (16 columns each), the MFMA is issued "swapped"
// (A-operand = column tile, B-operand = row tile) so that every lane holds 4 logits of ONE row
// for each of the four products: row max / sum-exp / KL accumulators are online, in-lane, and
// merged with two wave shuffles (lanes l, l^16, l^32 share a row) and one LDS exchange between
// the waves.  Backward: two sweeps of the same tile code.  MODE 0: own = rows, swept = columns ->
// d(a_i), d(b_i); MODE 1: own = columns, swept = rows -> d(A_j), d(B_j).  The coefficient tile is
// already in the A-operand layout of the second MFMA (coefficient x swept vectors, staged in
// LDS).  The kl_div target carries no gradient in Paddle, so
//     g_aa = p_ce_a(aa) + w (Pa_aa - Pb_ba)            g_ab = p_ce_a(ab) - [pos] + w (Pa_ab - Pb_bb)
//     g_ba = p_ce_b(ba) - [pos] + w (Pb_ba - Pa_aa)    g_bb = p_ce_b(bb) + w (Pb_bb - Pa_ab)
// (tests/test_oracle_simclr.py::test_head_gradient_closed_form checks this against autograd).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int D = 128;
constexpr float kNeg = -1e30f;          // "masked" logit: exp(kNeg - m) == 0 for every real m
constexpr int PITCH = D + 4;            // LDS row pitch (floats) of a staged 16 x 128 tile
constexpr int kStatsStride = 8;         // floats per row in rowstats

__device__ __forceinline__ float shx(float v, int m) { return __shfl_xor(v, m, 64); }

// running (max, sum exp) [, sum exp * weight] merge:  (m, z, w) <- (m, z, w) (+) (m2, z2, w2)
__device__ __forceinline__ void merge3(float& m, float& z, float& w, float m2, float z2, float w2) {
  const float M = fmaxf(m, m2);
  const float s1 = __expf(m - M), s2 = __expf(m2 - M);
  z = z * s1 + z2 * s2;
  w = w * s1 + w2 * s2;
  m = M;
}

// lane (l15, l4) loads elements [32*l4, 32*l4+32) of row `r` (zeros when r is out of range)
__device__ __forceinline__ void load_row32(const float* __restrict__ x, int r, int nrows, int l4,
                                           float (&reg)[32]) {
  if (r < nrows) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const float4 t = *reinterpret_cast<const float4*>(x + (int64_t)r * D + l4 * 32 + v * 4);
      reg[v * 4] = t.x; reg[v * 4 + 1] = t.y; reg[v * 4 + 2] = t.z; reg[v * 4 + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int v = 0; v < 32; ++v) reg[v] = 0.f;
  }
}

// S^T fragment: acc[r] = sum_d own[row = l15][d] * swept[idx = 4*l4 + r][d]   (swapped MFMA)
__device__ __forceinline__ f32x4 dot_tile(const float (&swept)[32], const float (&own)[32]) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 32; ++ks)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(swept[ks], own[ks], acc, 0, 0, 0);
  return acc;
}

struct Online {           // per-lane online statistics of one row
  float m1, z1;           // CE row a: {ab (all), aa (masked)}
  float m2, z2;           // CE row b: {ba (all), bb (masked)}
  float mx, zx, wx;       // Pa over {aa, ab} (positive masked), wx = sum exp * (y - x)
  float my, zy, wy;       // Pb over {ba, bb}
  float cnt;              // #{j != pos : ab_ij > ab_i,pos}
};

__global__ void __launch_bounds__(kThreads) ntxent_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ A,
    const float* __restrict__ Bl, int B, int BL, int roff, float invT, float co2w,
    float* __restrict__ rowstats, float* __restrict__ out) {
  __shared__ float xch[4][16][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int row = blockIdx.x * 16 + l15;
  float areg[32], breg[32];
  load_row32(a, row, B, l4, areg);
  load_row32(b, row, B, l4, breg);
  float pos = 0.f;
#pragma unroll
  for (int v = 0; v < 32; ++v) pos += areg[v] * breg[v];
  pos = shx(pos, 16) + pos;
  pos = shx(pos, 32) + pos;
  pos *= invT;
  const int pcol = roff + row;

  Online s;
  s.m1 = s.m2 = s.mx = s.my = kNeg;
  s.z1 = s.z2 = s.zx = s.zy = s.wx = s.wy = s.cnt = 0.f;
  const int ntile = (BL + 15) >> 4;
  for (int t = wave; t < ntile; t += 4) {
    float Areg[32], Breg[32];
    load_row32(A, t * 16 + l15, BL, l4, Areg);
    load_row32(Bl, t * 16 + l15, BL, l4, Breg);
    const f32x4 aa = dot_tile(Areg, areg), ab = dot_tile(Breg, areg);
    const f32x4 ba = dot_tile(Areg, breg), bb = dot_tile(Breg, breg);
    float vab[4], vaa[4], vba[4], vbb[4], xab[4], xba[4], d1[4], d2[4];
    float t1 = kNeg, t2 = kNeg, tx = kNeg, ty = kNeg;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = t * 16 + l4 * 4 + r;
      const bool valid = j < BL, self = j == pcol;
      vab[r] = valid ? ab[r] * invT : kNeg;
      vba[r] = valid ? ba[r] * invT : kNeg;
      vaa[r] = (valid && !self) ? aa[r] * invT : kNeg;
      vbb[r] = (valid && !self) ? bb[r] * invT : kNeg;
      xab[r] = self ? kNeg : vab[r];
      xba[r] = self ? kNeg : vba[r];
      d1[r] = (valid && !self) ? vba[r] - vaa[r] : 0.f;
      d2[r] = (valid && !self) ? vbb[r] - vab[r] : 0.f;
      t1 = fmaxf(t1, fmaxf(vab[r], vaa[r]));
      t2 = fmaxf(t2, fmaxf(vba[r], vbb[r]));
      tx = fmaxf(tx, fmaxf(vaa[r], xab[r]));
      ty = fmaxf(ty, fmaxf(xba[r], vbb[r]));
      s.cnt += (valid && !self && vab[r] > pos) ? 1.f : 0.f;
    }
    const float n1 = fmaxf(s.m1, t1), n2 = fmaxf(s.m2, t2), nx = fmaxf(s.mx, tx), ny = fmaxf(s.my, ty);
    const float c1 = __expf(s.m1 - n1), c2 = __expf(s.m2 - n2), cx = __expf(s.mx - nx), cy = __expf(s.my - ny);
    s.z1 *= c1; s.z2 *= c2; s.zx *= cx; s.wx *= cx; s.zy *= cy; s.wy *= cy;
    s.m1 = n1; s.m2 = n2; s.mx = nx; s.my = ny;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s.z1 += __expf(vab[r] - n1) + __expf(vaa[r] - n1);
      s.z2 += __expf(vba[r] - n2) + __expf(vbb[r] - n2);
      const float ea = __expf(vaa[r] - nx), eb = __expf(xab[r] - nx);
      s.zx += ea + eb;
      s.wx += ea * d1[r] + eb * d2[r];
      const float fa = __expf(xba[r] - ny), fb = __expf(vbb[r] - ny);
      s.zy += fa + fb;
      s.wy += fa * d1[r] + fb * d2[r];
    }
  }
  // lanes l, l^16, l^32 hold partial statistics of the same row
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    float dummy = 0.f, dummy2 = 0.f;
    merge3(s.m1, s.z1, dummy, shx(s.m1, o), shx(s.z1, o), 0.f);
    merge3(s.m2, s.z2, dummy2, shx(s.m2, o), shx(s.z2, o), 0.f);
    merge3(s.mx, s.zx, s.wx, shx(s.mx, o), shx(s.zx, o), shx(s.wx, o));
    merge3(s.my, s.zy, s.wy, shx(s.my, o), shx(s.zy, o), shx(s.wy, o));
    s.cnt += shx(s.cnt, o);
  }
  if (l4 == 0) {
    float* e = xch[wave][l15];
    e[0] = s.m1; e[1] = s.z1; e[2] = s.m2; e[3] = s.z2; e[4] = s.mx; e[5] = s.zx; e[6] = s.wx;
    e[7] = s.my; e[8] = s.zy; e[9] = s.wy; e[10] = s.cnt;
  }
  __syncthreads();
  if (wave == 0 && l4 == 0 && row < B) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float* e = xch[w][l15];
      float dummy = 0.f, dummy2 = 0.f;
      merge3(s.m1, s.z1, dummy, e[0], e[1], 0.f);
      merge3(s.m2, s.z2, dummy2, e[2], e[3], 0.f);
      merge3(s.mx, s.zx, s.wx, e[4], e[5], e[6]);
      merge3(s.my, s.zy, s.wy, e[7], e[8], e[9]);
      s.cnt += e[10];
    }
    const float lce_a = s.m1 + __logf(s.z1), lce_b = s.m2 + __logf(s.z2);
    const float lx = s.mx + __logf(s.zx), ly = s.my + __logf(s.zy);
    const float kl = s.wy / s.zy - s.wx / s.zx;          // sum (Pb - Pa)(y - x)
    float* o = rowstats + (int64_t)row * kStatsStride;
    // o[7] = this row's loss term: ntxent_mean_kernel folds the rows in a fixed order (no atomics)
    o[0] = lce_a; o[1] = lce_b; o[2] = lx; o[3] = ly; o[4] = kl; o[5] = pos; o[6] = s.cnt;
    o[7] = (lce_a - pos) + (lce_b - pos) + co2w * kl;
  }
}

// out[0] = mean_i rowstats[i][7] (loss), out[1] = share of rows whose positive ranks first
__global__ void __launch_bounds__(kThreads) ntxent_mean_kernel(const float* __restrict__ rowstats, int B,
                                                               float* __restrict__ out) {
  __shared__ float red[kThreads][2];
  float a0 = 0.f, a1 = 0.f;
  for (int r = threadIdx.x; r < B; r += kThreads) {
    a0 += rowstats[(int64_t)r * kStatsStride + 7];
    a1 += rowstats[(int64_t)r * kStatsStride + 6] < 0.5f ? 1.f : 0.f;
  }
  red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[threadIdx.x][0] += red[threadIdx.x + s][0];
      red[threadIdx.x][1] += red[threadIdx.x + s][1];
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) out[threadIdx.x] = red[0][threadIdx.x] / (float)B;
}

// stage 16 rows x 128 floats of x (rows r0.., zero beyond nrows) into a wave-private LDS tile
__device__ __forceinline__ void stage_tile(const float* __restrict__ x, int r0, int nrows, int lane,
                                           float* tile) {
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int ch = it * 64 + lane;          // 512 float4 chunks, 32 per row
    const int r = ch >> 5, c = ch & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < nrows) v = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + r) * D + c * 4);
    *reinterpret_cast<float4*>(tile + r * PITCH + c * 4) = v;
  }
}

__device__ __forceinline__ void tile_row32(const float* tile, int l15, int l4, float (&reg)[32]) {
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const float4 t = *reinterpret_cast<const float4*>(tile + l15 * PITCH + l4 * 32 + v * 4);
    reg[v * 4] = t.x; reg[v * 4 + 1] = t.y; reg[v * 4 + 2] = t.z; reg[v * 4 + 3] = t.w;
  }
}

// MODE 0: own = local rows (a, b), swept = columns (A, Bl): out1 = d a, out2 = d b
// MODE 1: own = columns (A, Bl), swept = local rows (a, b): out1 = d A, out2 = d Bl
template <int MODE>
__global__ void __launch_bounds__(kThreads) ntxent_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ A,
    const float* __restrict__ Bl, const float* __restrict__ rowstats,
    const float* __restrict__ gscale, int B, int BL, int roff, float invT, float co2w,
    float* __restrict__ out1, float* __restrict__ out2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  float* tileP = lds + wave * (2 * 16 * PITCH);
  float* tileQ = tileP + 16 * PITCH;
  const float* ownPsrc = MODE == 0 ? a : A;
  const float* ownQsrc = MODE == 0 ? b : Bl;
  const float* swPsrc = MODE == 0 ? A : a;
  const float* swQsrc = MODE == 0 ? Bl : b;
  const int n_own = MODE == 0 ? B : BL, n_sw = MODE == 0 ? BL : B;
  const int own = blockIdx.x * 16 + l15;
  float ownP[32], ownQ[32];
  load_row32(ownPsrc, own, n_own, l4, ownP);
  load_row32(ownQsrc, own, n_own, l4, ownQ);
  const float coef = (gscale ? *gscale : 1.0f) * invT / (float)B;
  float st_own[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0 && own < B) {
    const float4 t = *reinterpret_cast<const float4*>(rowstats + (int64_t)own * kStatsStride);
    st_own[0] = t.x; st_own[1] = t.y; st_own[2] = t.z; st_own[3] = t.w;
  }
  f32x4 g1[8], g2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { g1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; g2[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int ntile = (n_sw + 15) >> 4;
  for (int t = wave; t < ntile; t += 4) {
    stage_tile(swPsrc, t * 16, n_sw, lane, tileP);
    stage_tile(swQsrc, t * 16, n_sw, lane, tileQ);
    float swP[32], swQ[32];
    tile_row32(tileP, l15, l4, swP);
    tile_row32(tileQ, l15, l4, swQ);
    const f32x4 pp = dot_tile(swP, ownP), pq = dot_tile(swQ, ownP);   // ownP . swP, ownP . swQ
    const f32x4 qp = dot_tile(swP, ownQ), qq = dot_tile(swQ, ownQ);
    float c11[4], c12[4], c21[4], c22[4];     // out1 += c11*swP + c12*swQ ; out2 += c21*swP + c22*swQ
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int sw = t * 16 + l4 * 4 + r;
      const int rowi = MODE == 0 ? own : sw;
      const int colj = MODE == 0 ? sw : own;
      const bool valid = own < n_own && sw < n_sw;
      float lce_a = st_own[0], lce_b = st_own[1], lx = st_own[2], ly = st_own[3];
      if (MODE == 1 && valid) {
        const float4 s4 = *reinterpret_cast<const float4*>(rowstats + (int64_t)rowi * kStatsStride);
        lce_a = s4.x; lce_b = s4.y; lx = s4.z; ly = s4.w;
      }
      const float vaa = pp[r] * invT, vbb = qq[r] * invT;
      const float vab = (MODE == 0 ? pq[r] : qp[r]) * invT;
      const float vba = (MODE == 0 ? qp[r] : pq[r]) * invT;
      const bool self = colj == roff + rowi;
      const float m = self ? 0.f : 1.f;
      const float e_ab = __expf(vab - lce_a), e_aa = m * __expf(vaa - lce_a);
      const float e_ba = __expf(vba - lce_b), e_bb = m * __expf(vbb - lce_b);
      const float pa_aa = m * __expf(vaa - lx), pa_ab = m * __expf(vab - lx);
      const float pb_ba = m * __expf(vba - ly), pb_bb = m * __expf(vbb - ly);
      const float k = valid ? coef : 0.f;
      const float g_aa = k * (e_aa + co2w * (pa_aa - pb_ba));
      const float g_ab = k * (e_ab - (1.f - m) + co2w * (pa_ab - pb_bb));
      const float g_ba = k * (e_ba - (1.f - m) + co2w * (pb_ba - pa_aa));
      const float g_bb = k * (e_bb + co2w * (pb_bb - pa_ab));
      if (MODE == 0) { c11[r] = g_aa; c12[r] = g_ab; c21[r] = g_ba; c22[r] = g_bb; }
      else           { c11[r] = g_aa; c12[r] = g_ba; c21[r] = g_ab; c22[r] = g_bb; }
    }
    // out[own = 4*l4' + r'][d = jd*16 + l15] += sum_sw c[own][sw] * swept[sw][d]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* rp = tileP + (l4 * 4 + r) * PITCH + l15;
      const float* rq = tileQ + (l4 * 4 + r) * PITCH + l15;
#pragma unroll
      for (int jd = 0; jd < 8; ++jd) {
        const float p = rp[jd * 16], q = rq[jd * 16];
        g1[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(c11[r], p, g1[jd], 0, 0, 0);
        g1[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(c12[r], q, g1[jd], 0, 0, 0);
        g2[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(c21[r], p, g2[jd], 0, 0, 0);
        g2[jd] = __builtin_amdgcn_mfma_f32_16x16x4f32(c22[r], q, g2[jd], 0, 0, 0);
      }
    }
  }
  // the 4 waves swept disjoint column tiles of the SAME 16 own rows: fold their partials through LDS in
  // wave order and store (no atomics: out1 / out2 are fully written, bit-reproducible)
  __syncthreads();                         // every wave is done with its staging tiles
#pragma unroll
  for (int jd = 0; jd < 8; ++jd)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tileP[(l4 * 4 + r) * PITCH + jd * 16 + l15] = g1[jd][r];
      tileQ[(l4 * 4 + r) * PITCH + jd * 16 + l15] = g2[jd][r];
    }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * 16 * (D / 4); e += kThreads) {
    const int which = e / (16 * (D / 4)), rem = e % (16 * (D / 4));
    const int r = rem / (D / 4), c4 = rem % (D / 4);
    const int orow = blockIdx.x * 16 + r;
    if (orow >= n_own) continue;
    float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(lds + w * (2 * 16 * PITCH) + which * (16 * PITCH) +
                                                         r * PITCH + c4 * 4);
      acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
    }
    *reinterpret_cast<float4*>((which ? out2 : out1) + (int64_t)orow * D + c4 * 4) = acc4;
  }
}

constexpr int kBwdLds = 4 * 2 * 16 * PITCH * (int)sizeof(float);

}  // namespace

extern "C" int passl_hip_ntxent_fwd(const float* a, const float* b, const float* a_all,
                                    const float* b_all, int B, int BL, int row_offset, int Dd,
                                    float T, float co2_weight, float* out, float* rowstats,
                                    passl_stream_t stream) {
  if (!a || !b || !a_all || !b_all || !out || !rowstats || B <= 0 || BL < B || Dd != D ||
      row_offset < 0 || row_offset + B > BL || !(T > 0.f) || !aligned16(a) || !aligned16(b) ||
      !aligned16(a_all) || !aligned16(b_all) || !aligned16(rowstats))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(ntxent_fwd_kernel, dim3((B + 15) / 16), dim3(kThreads), 0, st, a, b, a_all,
                     b_all, B, BL, row_offset, 1.0f / T, co2_weight, rowstats, out);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(ntxent_mean_kernel, dim3(1), dim3(kThreads), 0, st, rowstats, B, out);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_ntxent_bwd(const float* a, const float* b, const float* a_all,
                                    const float* b_all, const float* rowstats, const float* gscale,
                                    int B, int BL, int row_offset, int Dd, float T,
                                    float co2_weight, float* da, float* db, float* da_all,
                                    float* db_all, passl_stream_t stream) {
  if (!a || !b || !a_all || !b_all || !rowstats || !da || !db || !da_all || !db_all || B <= 0 ||
      BL < B || Dd != D || row_offset < 0 || row_offset + B > BL || !(T > 0.f) || !aligned16(a) ||
      !aligned16(b) || !aligned16(a_all) || !aligned16(b_all) || !aligned16(rowstats))
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ntxent_bwd_kernel<0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ntxent_bwd_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds);
    attr = true;
  }
  hipLaunchKernelGGL(ntxent_bwd_kernel<0>, dim3((B + 15) / 16), dim3(kThreads), kBwdLds, st, a, b,
                     a_all, b_all, rowstats, gscale, B, BL, row_offset, 1.0f / T, co2_weight, da, db);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(ntxent_bwd_kernel<1>, dim3((BL + 15) / 16), dim3(kThreads), kBwdLds, st, a, b,
                     a_all, b_all, rowstats, gscale, B, BL, row_offset, 1.0f / T, co2_weight,
                     da_all, db_all);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
