// Linear-probe head on gfx950: softmax cross-entropy with integer labels + top-1 / top-5 accuracy
// (forward and backward) over fp32 class scores [N][C].
//
// Reference: ClasHead.loss + accuracy, passl_v110/modeling/heads/clas_head.py:47-72
//   loss = mean_i (logsumexp(s_i) - s_i[label_i]);  acc_k = 100/N * #{i : label_i in top-k(s_i)}
// One wave per row (C = 1000: 16 scores per lane): row max / sum-exp by wave reductions; the
// label's rank = #{j : s_ij > s_i,label or (s_ij == s_i,label and j < label)} (ties resolve to the
// lower index, as a sorted top-k does), so no sort / top-k selection is materialised.
// A label outside [0, C) makes the loss NaN (the reference raises inside Paddle's kernel).
#include <math.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads) softmax_ce_fwd_kernel(const float* __restrict__ s,
                                                                  const int64_t* __restrict__ labels,
                                                                  int N, int C, float* __restrict__ lse,
                                                                  float* __restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* r = s + (int64_t)row * C;
  const int64_t lab = labels[row];
  const bool ok = lab >= 0 && lab < C;
  const float sl = ok ? r[lab] : 0.f;
  float m = -INFINITY;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, r[j]);
  m = wave_max(m);
  float z = 0.f, cnt = 0.f;
  for (int j = lane; j < C; j += 64) {
    const float v = r[j];
    z += __expf(v - m);
    cnt += (v > sl || (v == sl && j < lab)) ? 1.f : 0.f;
  }
  z = wave_sum(z);
  cnt = wave_sum(cnt);
  if (lane == 0) {
    const float l = m + __logf(z);
    lse[row] = l;
    const float invN = 1.0f / (float)N;
    terms[row] = ok ? (l - sl) * invN : NAN;               // [3][N]: loss term, top-1 hit, top-5 hit
    terms[N + row] = (ok && cnt < 0.5f) ? 100.0f * invN : 0.f;
    terms[2 * N + row] = (ok && cnt < 4.5f) ? 100.0f * invN : 0.f;
  }
}

// out[k] = sum_i terms[k][i] in one fixed order (wave k)
__global__ void __launch_bounds__(192) softmax_ce_finish_kernel(const float* __restrict__ terms, int N,
                                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float a = 0.f;
  for (int i = lane; i < N; i += 64) a += terms[(int64_t)w * N + i];
  a = wave_sum(a);
  if (lane == 0) out[w] = a;
}

// ds[i][j] = g/N (exp(s_ij - lse_i) - [j == label_i])
__global__ void __launch_bounds__(kThreads) softmax_ce_bwd_kernel(const float* __restrict__ s,
                                                                  const float* __restrict__ lse,
                                                                  const int64_t* __restrict__ labels,
                                                                  const float* __restrict__ gloss, int N,
                                                                  int C, float* __restrict__ ds) {
  const int64_t total = (int64_t)N * C;
  const float k = *gloss / (float)N;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kThreads) {
    const int i = (int)(e / C), j = (int)(e % C);
    ds[e] = k * (__expf(s[e] - lse[i]) - (labels[i] == j ? 1.0f : 0.0f));
  }
}

}  // namespace

// ws: 3 * N floats (per-row loss term and top-1 / top-5 hits, summed in a fixed order)
extern "C" int passl_hip_softmax_ce_fwd(const float* scores, const int64_t* labels, int N, int C,
                                        float* lse, float* out, float* ws, int64_t ws_floats,
                                        passl_stream_t stream) {
  if (!scores || !labels || !lse || !out || !ws || N <= 0 || C <= 0 || ws_floats < 3 * (int64_t)N)
    return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, st, scores, labels, N,
                     C, lse, ws);
  hipLaunchKernelGGL(softmax_ce_finish_kernel, dim3(1), dim3(192), 0, st, ws, N, out);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_softmax_ce_bwd(const float* scores, const float* lse, const int64_t* labels,
                                        const float* gloss, int N, int C, float* dscores,
                                        passl_stream_t stream) {
  if (!scores || !lse || !labels || !gloss || !dscores || N <= 0 || C <= 0) return PASSL_EINVAL;
  int64_t g = ((int64_t)N * C + kThreads - 1) / kThreads;
  if (g > 65535 * 4) g = 65535 * 4;
  hipLaunchKernelGGL(softmax_ce_bwd_kernel, dim3((unsigned)g), dim3(kThreads), 0, as_stream(stream),
                     scores, lse, labels, gloss, N, C, dscores);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
