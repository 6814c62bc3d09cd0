// kbench: convolution check / timing through the C ABI (include/passl_hip.h) without Python or torch.
//
// Why: a fresh GPU box spends 1-2 minutes on its first `import torch`; this binary starts in a second, so a
// kernel experiment can be checked and timed in a GPU call of well under a minute.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -Iinclude -Lpassl_amd/lib -lpassl_hip \
//         -Wl,-rpath,'$ORIGIN/../passl_amd/lib' -o tools/kbench          (tools/build_kbench.sh)
//
//   tools/kbench check [name=value ...]     bit-exact check of passl_hip_conv_igemm on small cases
//   tools/kbench time  [name=value ...]     per-layer table of the ResNet-50 3x3 / 1x1 shapes at N = 256
//   tools/kbench ab name=v0,v1 [...]        the same table for two values of ONE option, side by side
//   tools/kbench wcheck | wtime [name=value ...]   the same for passl_hip_conv_wgrad (== on fp32 sums; per-layer table)
//   tools/kbench fincheck                   BatchNorm finalize launches (forward / backward) against the host's fp64 arithmetic on the same slab
//   tools/kbench fintime                    ... and their times, alone and next to a stream that loads the memory system
//   tools/kbench finstress [iters=N]        ... and the multi-segment hand-off under load: every result bit-identical run to run
//   tools/kbench poolcheck | pooltime       the stem's fused BatchNorm + ReLU + max-pool backward: second form == first form; times
//   tools/kbench vtime                      the four epilogue variants of the K = 64 / 128 1x1 layers (PASSL_IGEMM_LEAN=1|2 to compare)
//   tools/kbench ablate                     the register-staged kernel's debug switches on the 1x1 shapes
//   tools/kbench sweep cfg [cfg ...]        cfg = "name=value,name=value": check + time the 3x3 shapes under each
//   name=value pairs are passl_hip_set_option() calls made before anything runs.
//
// Exactness: operands are multiples of 1/16 with magnitude < 1, so every product is a multiple of 1/256 and
// every partial sum of up to 4608 of them is exactly representable in fp32 — the fp32 accumulation is exact
// in ANY order, and the expected output is simply bf16(round-to-nearest-even of the integer sum / 256).
// A kernel that adds the right products gets every output bit right; the comparison is ==.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include "passl_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// ------------------------------------------------------------------------------------------------ data
__host__ __device__ static inline uint32_t mix(uint64_t i, uint32_t seed) {
  uint64_t z = i * 0x9E3779B97F4A7C15ull + ((uint64_t)seed << 32 | 0x7F4A7C15u);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)(z >> 33);
}
// integer in [-15, 15]; the element's value is that / 16
__host__ __device__ static inline int ival(uint64_t i, uint32_t seed) { return (int)(mix(i, seed) % 31u) - 15; }
__host__ __device__ static inline uint16_t bf16_of_small(int v) {        // v / 16 exactly, as bf16 bits
  float f = (float)v * 0.0625f;
  union { float f; uint32_t u; } c; c.f = f;
  return (uint16_t)(c.u >> 16);                                           // 5 significant bits: exact
}
__global__ void fill_kernel(uint16_t* p, int64_t n, uint32_t seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = bf16_of_small(ival((uint64_t)i, seed));
}
static void fill(void* p, int64_t n, uint32_t seed) {
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (uint16_t*)p, n, seed);
  CK(hipGetLastError());
}
static inline uint16_t bf16_rne(float f) {
  union { float f; uint32_t u; } c; c.f = f;
  const uint32_t lsb = (c.u >> 16) & 1u;
  return (uint16_t)((c.u + 0x7fffu + lsb) >> 16);
}
static inline float bf16_to_f(uint16_t b) { union { float f; uint32_t u; } c; c.u = (uint32_t)b << 16; return c.f; }

// ------------------------------------------------------------------------------------------------ shapes
struct Shape { int N, C, K, R, stride, H; const char* note; int mult; };
static passl_conv_desc make_desc(const Shape& s, const void* a, const void* b, void* y) {
  passl_conv_desc d;
  memset(&d, 0, sizeof(d));
  const int pad = s.R / 2;
  const int O = (s.H + 2 * pad - s.R) / s.stride + 1;
  d.a = a; d.b = b; d.y = y;
  d.N = s.N; d.OP = O; d.OQ = O; d.NCOLS = s.K; d.R = s.R; d.S = s.R; d.C = s.C;
  d.IH = s.H; d.IW = s.H; d.sh = d.sw = s.stride; d.ph = d.pw = pad;
  d.a_sw = s.C; d.a_sh = (int64_t)s.H * s.C; d.a_sn = (int64_t)s.H * s.H * s.C;
  d.y_sw = s.K; d.y_sh = (int64_t)O * s.K; d.y_sn = (int64_t)O * O * s.K;
  d.dtype = PASSL_BF16;
  return d;
}

struct Buffers {
  void *a = nullptr, *b = nullptr, *y = nullptr, *aux = nullptr; float* stats = nullptr; float* cols = nullptr;
  int64_t na = 0, nb = 0, ny = 0, nstats = 0, naux = 0;
  void ensure(int64_t a_, int64_t b_, int64_t y_, int64_t s_) {
    if (a_ > na) { if (a) CK(hipFree(a)); CK(hipMalloc(&a, a_ * 2)); na = a_; }
    if (b_ > nb) { if (b) CK(hipFree(b)); CK(hipMalloc(&b, b_ * 2)); nb = b_; }
    if (y_ > ny) { if (y) CK(hipFree(y)); CK(hipMalloc(&y, y_ * 2)); ny = y_; }
    if (s_ > nstats) { if (stats) CK(hipFree(stats)); CK(hipMalloc((void**)&stats, s_ * 4)); nstats = s_; }
    if (!cols) CK(hipMalloc((void**)&cols, 4 * 4096 * 4));          // four per-column fp32 vectors (<= 4096 columns)
  }
  void ensure_aux(int64_t n) { if (n > naux) { if (aux) CK(hipFree(aux)); CK(hipMalloc(&aux, n * 2)); naux = n; } }
};
enum Variant { V_RELU = 0, V_STATS = 1, V_RESIDUAL = 2, V_BNB = 3 };
static const char* vname(int v) { static const char* n[] = {"relu ", "stats", "resid", "bnb  "}; return n[v]; }

// ------------------------------------------------------------------------------------------------ check
// The exact sums (x 256) of one case, computed once per shape on the host.
static void reference(const Shape& s, std::vector<int32_t>& out) {
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
  const int64_t na = (int64_t)s.N * s.H * s.H * s.C, nb = (int64_t)s.K * KD;
  std::vector<int8_t> ha(na), hb(nb);
  for (int64_t i = 0; i < na; ++i) ha[i] = (int8_t)ival((uint64_t)i, 11u);
  for (int64_t i = 0; i < nb; ++i) hb[i] = (int8_t)ival((uint64_t)i, 23u);
  out.assign(M * s.K, 0);
  for (int64_t m = 0; m < M; ++m) {
    const int n = (int)(m / ((int64_t)O * O)), rem = (int)(m % ((int64_t)O * O)), op = rem / O, oq = rem % O;
    int32_t* accum = &out[m * s.K];
    for (int r = 0; r < s.R; ++r) {
      const int ih = op * s.stride + r - pad;
      if (ih < 0 || ih >= s.H) continue;
      for (int q = 0; q < s.R; ++q) {
        const int iw = oq * s.stride + q - pad;
        if (iw < 0 || iw >= s.H) continue;
        const int8_t* ap = &ha[(((int64_t)n * s.H + ih) * s.H + iw) * s.C];
        for (int k = 0; k < s.K; ++k) {
          const int8_t* bp = &hb[(int64_t)k * KD + ((int64_t)r * s.R + q) * s.C];
          int32_t t = 0;
          for (int c = 0; c < s.C; ++c) t += (int32_t)ap[c] * (int32_t)bp[c];
          accum[k] += t;
        }
      }
    }
  }
}

// Full-output comparison of one epilogue variant; returns the number of wrong outputs.
//   V_RELU      y = bf16(relu(sum))
//   V_STATS     y = bf16(sum) + the fused BatchNorm statistics slab
//   V_RESIDUAL  y = bf16(relu(bf16(sum) + residual))                (the block-end convolutions)
//   V_BNB       y = g = bf16(sum) masked by relu(bnb_y * scale + shift) > 0, and the BatchNorm-backward partial sums
//               sum g, sum g (bnb_y - mean) invstd                    (the data-gradient launches)
static int64_t check_case(const Shape& s, int variant, Buffers& B, int* kernel_used, const std::vector<int32_t>& ref) {
  const bool relu = variant == V_RELU || variant == V_RESIDUAL, with_stats = variant == V_STATS;
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
  const int64_t na = (int64_t)s.N * s.H * s.H * s.C, nb = (int64_t)s.K * KD, ny = M * s.K;
  const int tiles = (int)((M + 127) / 128);
  B.ensure(na, nb, ny, (int64_t)tiles * s.K * 3);
  fill(B.a, na, 11u); fill(B.b, nb, 23u);
  CK(hipMemset(B.y, 0xff, ny * 2));
  passl_conv_desc d = make_desc(s, B.a, B.b, B.y);
  d.relu = relu ? 1 : 0;
  if (with_stats) { d.stats = B.stats; d.stats_tiles = tiles; CK(hipMemset(B.stats, 0xff, (int64_t)tiles * s.K * 12)); }
  std::vector<float> hcols(4 * 4096, 0.f);          // mean, invstd, scale, shift per column
  if (variant == V_RESIDUAL || variant == V_BNB) { B.ensure_aux(ny); fill(B.aux, ny, 37u); }
  if (variant == V_RESIDUAL) d.residual = B.aux;
  if (variant == V_BNB) {
    for (int k = 0; k < s.K; ++k) {
      hcols[k] = (float)ival((uint64_t)k, 41u) * 0.0625f;                    // mean
      hcols[4096 + k] = 0.5f + (float)(mix((uint64_t)k, 43u) % 16u) * 0.125f;   // invstd
      hcols[8192 + k] = (mix((uint64_t)k, 47u) & 1u) ? 1.0f : -0.5f;            // scale
      hcols[12288 + k] = (float)ival((uint64_t)k, 53u) * 0.0625f;              // shift
    }
    CK(hipMemcpy(B.cols, hcols.data(), hcols.size() * 4, hipMemcpyHostToDevice));
    d.bnb_y = B.aux; d.bnb_mean = B.cols; d.bnb_invstd = B.cols + 4096; d.bnb_scale = B.cols + 8192;
    d.bnb_shift = B.cols + 12288; d.bnb_partial = B.stats; d.bnb_relu = 2; d.bnb_tile_off = 0;
    CK(hipMemset(B.stats, 0xff, (int64_t)tiles * s.K * 8));
  }
  int rc = passl_hip_conv_igemm(&d, nullptr);
  if (rc != PASSL_OK) { printf("    conv_igemm -> %d (%s)\n", rc, passl_hip_strerror(rc)); return -1; }
  CK(hipDeviceSynchronize());
  *kernel_used = passl_hip_last_igemm_kernel();
  std::vector<uint16_t> y(ny);
  CK(hipMemcpy(y.data(), B.y, ny * 2, hipMemcpyDeviceToHost));
  int64_t bad = 0, shown = 0;
  std::vector<double> bsum0, bsum1;
  if (variant == V_BNB) { bsum0.assign(s.K, 0.0); bsum1.assign(s.K, 0.0); }
  for (int64_t m = 0; m < M; ++m) {
    const int n = (int)(m / ((int64_t)O * O)), rem = (int)(m % ((int64_t)O * O)), op = rem / O, oq = rem % O;
    for (int k = 0; k < s.K; ++k) {
      float v = (float)ref[m * s.K + k] * (1.0f / 256.0f);
      uint16_t want;
      if (variant == V_RESIDUAL) {
        const float r = (float)ival((uint64_t)(m * s.K + k), 37u) * 0.0625f;
        float t = bf16_to_f(bf16_rne(v)) + r;
        want = bf16_rne(t < 0.f ? 0.f : t);
      } else if (variant == V_BNB) {
        const float yv = (float)ival((uint64_t)(m * s.K + k), 37u) * 0.0625f;
        const float g = (yv * hcols[8192 + k] + hcols[12288 + k]) > 0.f ? bf16_to_f(bf16_rne(v)) : 0.f;
        want = bf16_rne(g);
        bsum0[k] += g;
        bsum1[k] += (double)(g * (yv - hcols[k]) * hcols[4096 + k]);
      } else {
        if (relu && v < 0.f) v = 0.f;
        want = bf16_rne(v);
      }
      const uint16_t got = y[m * s.K + k];
      if (want != got && !((want & 0x7fff) == 0 && (got & 0x7fff) == 0)) {
        ++bad;
        if (shown < 6) { printf("    m=%lld (n=%d op=%d oq=%d) col=%d  want %g got %g\n", (long long)m, n, op, oq, k, bf16_to_f(want), bf16_to_f(got)); ++shown; }
      }
    }
  }
  if (variant == V_BNB && bad == 0) {
    // bnb_partial[t][col][0..1], summed over the tiles (the way passl_hip_bn_bwd_finalize consumes it)
    std::vector<float> st((int64_t)tiles * s.K * 2);
    CK(hipMemcpy(st.data(), B.stats, st.size() * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < s.K && bad == 0; ++k) {
      double g0 = 0, g1 = 0;
      for (int t = 0; t < tiles; ++t) { g0 += st[((int64_t)t * s.K + k) * 2]; g1 += st[((int64_t)t * s.K + k) * 2 + 1]; }
      if (!(fabs(g0 - bsum0[k]) <= 1e-3 * (1 + fabs(bsum0[k]))) || !(fabs(g1 - bsum1[k]) <= 2e-3 * (1 + fabs(bsum1[k])) + 1e-2)) {
        ++bad;
        printf("    bnb col %d: sum g %g want %g, sum g xhat %g want %g\n", k, g0, bsum0[k], g1, bsum1[k]);
      }
    }
  }
  if (with_stats && bad == 0) {
    // slab: [tiles][K][2] sums of (v - s), (v - s)^2, then [tiles][K] shifts s; v = the stored outputs.  Checked the
    // way passl_hip_bn_finalize consumes it — per column, sum_t (S0_t + n_t s_t) and sum_t (S1_t + 2 s_t S0_t + n_t s_t^2)
    // with n_t = 128 rows per tile (the rest in the last one) — which does not depend on WHICH rows a tile holds
    // (the 2-D tiles of the spatially tiled kernel are not runs of consecutive rows).
    std::vector<float> st((int64_t)tiles * s.K * 3);
    CK(hipMemcpy(st.data(), B.stats, st.size() * 4, hipMemcpyDeviceToHost));
    for (int k = 0; k < s.K && bad == 0; ++k) {
      double want0 = 0, want1 = 0, wabs = 0, got0 = 0, got1 = 0;
      for (int64_t m = 0; m < M; ++m) {
        const double v = (double)bf16_to_f(y[m * s.K + k]);
        want0 += v; want1 += v * v; wabs += fabs(v);
      }
      for (int t = 0; t < tiles; ++t) {
        const double n_t = (double)((t + 1 < tiles) ? 128 : M - (int64_t)128 * t);
        const double sh = st[(int64_t)tiles * s.K * 2 + (int64_t)t * s.K + k];
        const double g0 = st[((int64_t)t * s.K + k) * 2], g1 = st[((int64_t)t * s.K + k) * 2 + 1];
        got0 += g0 + n_t * sh;
        got1 += g1 + 2.0 * sh * g0 + n_t * sh * sh;
      }
      if (!(fabs(got0 - want0) <= 2e-4 * (1 + wabs)) || !(fabs(got1 - want1) <= 2e-4 * (1 + want1))) {
        ++bad;
        printf("    stats col %d: sum %g want %g, sum of squares %g want %g\n", k, got0, want0, got1, want1);
      }
    }
  }
  return bad;
}

static const char* kname(int k) {
  switch (k) { case 0: return "igemm"; case 1: return "ring"; case 2: return "stem"; case 3: return "8p"; case 4: return "wave"; default: return "?"; }
}

static int run_check() {
  // small batches on the real spatial sizes: tiles that cross image rows and images, a ragged last tile
  // (3 * 56 * 56 = 9408 = 73.5 tiles), every stage's 3x3 geometry, plus 1x1 and strided cases for the
  // kernels the options may re-route
  const Shape cases[] = {
      {3, 64, 64, 3, 1, 56, "stage-1 3x3", 0},   {2, 128, 128, 3, 1, 28, "stage-2 3x3", 0},
      {5, 256, 256, 3, 1, 14, "stage-3 3x3", 0}, {7, 512, 512, 3, 1, 7, "stage-4 3x3", 0},
      {3, 64, 128, 3, 1, 20, "3x3, 64 -> 128, odd width 20", 0}, {1, 128, 64, 3, 1, 9, "3x3, one image 9x9 (81 rows)", 0},
      {3, 128, 128, 3, 1, 16, "3x3, 16 x 16 (2-D tiles, 12 patches)", 0}, {5, 64, 64, 3, 1, 8, "3x3, 8 x 8 (5 patches: ragged)", 0},
      {2, 128, 128, 3, 2, 56, "3x3 stride 2", 0}, {2, 64, 256, 1, 1, 56, "1x1 64 -> 256", 0},
      {2, 512, 128, 1, 1, 28, "1x1 512 -> 128", 0},
      // dense 1x1 launches of the register-staged kernel (persistent form under igemm_persist=1; with
      // igemm_persist_grid=8 every workgroup walks many tiles): 1 / 2 / 4 K-tiles, 64- and 128-column tiles, ragged M
      {2, 64, 64, 1, 1, 56, "1x1 64 -> 64", 0},   {2, 256, 64, 1, 1, 56, "1x1 256 -> 64", 0},
      {3, 128, 512, 1, 1, 28, "1x1 128 -> 512", 0}, {3, 64, 256, 1, 1, 20, "1x1 64 -> 256, 1200 rows (ragged)", 0},
      {5, 256, 128, 1, 1, 14, "1x1 256 -> 128", 0},
  };
  Buffers B;
  int failures = 0;
  std::vector<int32_t> ref;
  for (const Shape& s : cases) {
    reference(s, ref);
    for (int variant = 0; variant < 4; ++variant) {
      int used = -1;
      const int64_t bad = check_case(s, variant, B, &used, ref);
      printf("%-34s N=%d %s: kernel %-5s %s\n", s.note, s.N, vname(variant), kname(used),
             bad == 0 ? "exact" : (bad < 0 ? "NOT RUN" : "WRONG"));
      if (bad != 0) { ++failures; if (bad > 0) printf("    %lld wrong outputs\n", (long long)bad); }
    }
  }
  printf(failures ? "CHECK FAILED (%d)\n" : "CHECK OK\n", failures);
  return failures ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dw[col][r][s][c] (fp32) = sum_m dy[m][col] * x[m @ (r, s)][c].  With at most ~16 K output rows the sum of products
// (multiples of 1/256, magnitude < 1) is exact in fp32 in any order and over any number of slices: ==.
static passl_wgrad_desc make_wdesc(const Shape& s, const void* a, const void* dy, float* dw, float* ws, int64_t ws_floats,
                                   int splits) {
  passl_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  d.a = a; d.dy = dy; d.dw = dw;
  d.N = s.N; d.OP = O; d.OQ = O; d.NCOLS = s.K; d.R = s.R; d.S = s.R; d.C = s.C;
  d.IH = s.H; d.IW = s.H; d.sh = d.sw = s.stride; d.ph = d.pw = pad;
  d.a_sw = s.C; d.a_sh = (int64_t)s.H * s.C; d.a_sn = (int64_t)s.H * s.H * s.C;
  d.dy_ld = s.K; d.ws = ws; d.ws_floats = ws_floats; d.dtype = PASSL_BF16; d.splits = splits;
  return d;
}
static int wgrad_splits(int64_t M, int ncols, int64_t kdim) {        // passl_amd/hip/plan.py: wgrad_splits
  const int64_t tiles = ((ncols + 127) / 128) * ((kdim + 127) / 128), nk = (M + 63) / 64;
  const int64_t target = 256;            // plan.py: wgrad_target_blocks(spatial = true): every shape here is a convolution
  int64_t sp = target / (tiles > 0 ? tiles : 1);
  if (sp > nk) sp = nk;
  return (int)(sp < 1 ? 1 : sp);
}
struct WBuffers {
  void *a = nullptr, *dy = nullptr; float *dw = nullptr, *ws = nullptr;
  int64_t na = 0, ndy = 0, ndw = 0, nws = 0;
  void ensure(int64_t a_, int64_t dy_, int64_t dw_, int64_t ws_) {
    if (a_ > na) { if (a) CK(hipFree(a)); CK(hipMalloc(&a, a_ * 2)); na = a_; }
    if (dy_ > ndy) { if (dy) CK(hipFree(dy)); CK(hipMalloc(&dy, dy_ * 2)); ndy = dy_; }
    if (dw_ > ndw) { if (dw) CK(hipFree(dw)); CK(hipMalloc((void**)&dw, dw_ * 4)); ndw = dw_; }
    if (ws_ > nws) { if (ws) CK(hipFree(ws)); CK(hipMalloc((void**)&ws, ws_ * 4)); nws = ws_; }
  }
};
static void wgrad_reference(const Shape& s, std::vector<int32_t>& want) {
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
  const int64_t na = (int64_t)s.N * s.H * s.H * s.C, ndy = M * s.K, ndw = (int64_t)s.K * KD;
  std::vector<int8_t> ha(na), hdy(ndy);
  for (int64_t i = 0; i < na; ++i) ha[i] = (int8_t)ival((uint64_t)i, 11u);
  for (int64_t i = 0; i < ndy; ++i) hdy[i] = (int8_t)ival((uint64_t)i, 29u);
  want.assign(ndw, 0);
  for (int64_t m = 0; m < M; ++m) {
    const int n = (int)(m / ((int64_t)O * O)), rem = (int)(m % ((int64_t)O * O)), op = rem / O, oq = rem % O;
    for (int r = 0; r < s.R; ++r) {
      const int ih = op * s.stride + r - pad;
      if (ih < 0 || ih >= s.H) continue;
      for (int q = 0; q < s.R; ++q) {
        const int iw = oq * s.stride + q - pad;
        if (iw < 0 || iw >= s.H) continue;
        const int8_t* ap = &ha[(((int64_t)n * s.H + ih) * s.H + iw) * s.C];
        const int8_t* gp = &hdy[m * s.K];
        for (int k = 0; k < s.K; ++k) {
          const int32_t gk = gp[k];
          if (!gk) continue;
          int32_t* wp = &want[(int64_t)k * KD + ((int64_t)r * s.R + q) * s.C];
          for (int c = 0; c < s.C; ++c) wp[c] += gk * (int32_t)ap[c];
        }
      }
    }
  }
}
static int64_t wgrad_check_case(const Shape& s, int splits, WBuffers& B, const std::vector<int32_t>& want) {
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
  const int64_t na = (int64_t)s.N * s.H * s.H * s.C, ndy = M * s.K, ndw = (int64_t)s.K * KD;
  if (splits <= 0) splits = wgrad_splits(M, s.K, KD);
  B.ensure(na, ndy, ndw, (int64_t)splits * ndw);
  fill(B.a, na, 11u); fill(B.dy, ndy, 29u);
  CK(hipMemset(B.dw, 0, ndw * 4));
  passl_wgrad_desc d = make_wdesc(s, B.a, B.dy, B.dw, splits > 1 ? B.ws : nullptr, splits > 1 ? (int64_t)splits * ndw : 0, splits);
  const int rc = passl_hip_conv_wgrad(&d, nullptr);
  if (rc != PASSL_OK) { printf("    conv_wgrad -> %d (%s)\n", rc, passl_hip_strerror(rc)); return -1; }
  CK(hipDeviceSynchronize());
  std::vector<float> dw(ndw);
  CK(hipMemcpy(dw.data(), B.dw, ndw * 4, hipMemcpyDeviceToHost));
  int64_t bad = 0;
  for (int64_t i = 0; i < ndw; ++i) {
    const float w = (float)want[i] * (1.0f / 256.0f);
    if (dw[i] != w) {
      if (bad < 5) printf("    dw[%lld] (col %lld, k %lld) want %g got %g\n", (long long)i, (long long)(i / KD), (long long)(i % KD), w, dw[i]);
      ++bad;
    }
  }
  return bad;
}
static int run_wcheck() {
  const Shape cases[] = {
      {3, 64, 64, 3, 1, 56, "3x3 64->64 @56", 0},    {2, 128, 128, 3, 1, 28, "3x3 128->128 @28", 0},
      {5, 256, 256, 3, 1, 14, "3x3 256->256 @14", 0}, {7, 512, 512, 3, 1, 7, "3x3 512->512 @7", 0},
      {2, 128, 128, 3, 2, 56, "3x3 stride 2 @56", 0}, {3, 64, 256, 1, 1, 56, "1x1 64->256 @56", 0},
      {3, 256, 64, 1, 1, 56, "1x1 256->64 @56", 0},   {4, 512, 128, 1, 1, 28, "1x1 512->128 @28", 0},
      {3, 64, 128, 3, 1, 20, "3x3 64->128, width 20", 0}, {2, 256, 512, 1, 2, 56, "1x1 stride 2 256->512", 0},
      {2, 128, 128, 3, 1, 16, "3x3 128->128 @16", 0}, {3, 96, 72, 3, 1, 16, "3x3 96->72 @16 (ragged blocks)", 0},
      {5, 64, 64, 3, 1, 8, "3x3 64->64 @8", 0},
  };
  WBuffers B;
  int failures = 0;
  std::vector<int32_t> want;
  for (const Shape& s : cases) {
    wgrad_reference(s, want);
    for (int splits : {0, 1, 7}) {
      const int64_t bad = wgrad_check_case(s, splits, B, want);
      printf("%-28s N=%d splits %-4s: %s\n", s.note, s.N, splits ? std::to_string(splits).c_str() : "auto",
             bad == 0 ? "exact" : (bad < 0 ? "NOT RUN" : "WRONG"));
      if (bad != 0) { ++failures; if (bad > 0) printf("    %lld wrong elements\n", (long long)bad); }
    }
  }
  printf(failures ? "WGRAD CHECK FAILED (%d)\n" : "WGRAD CHECK OK\n", failures);
  return failures ? 1 : 0;
}
static const Shape kR50[] = {
    {256, 64, 64, 3, 1, 56, "64->64 k3 @56", 3},     {256, 128, 128, 3, 1, 28, "128->128 k3 @28", 3},
    {256, 256, 256, 3, 1, 14, "256->256 k3 @14", 5}, {256, 512, 512, 3, 1, 7, "512->512 k3 @7", 2},
    {256, 128, 128, 3, 2, 56, "128->128 k3 s2 @56", 1}, {256, 256, 256, 3, 2, 28, "256->256 k3 s2 @28", 1},
    {256, 64, 256, 1, 1, 56, "64->256 k1 @56", 4},   {256, 256, 64, 1, 1, 56, "256->64 k1 @56", 2},
    {256, 128, 512, 1, 1, 28, "128->512 k1 @28", 4}, {256, 512, 128, 1, 1, 28, "512->128 k1 @28", 3},
    {256, 256, 1024, 1, 1, 14, "256->1024 k1 @14", 6}, {256, 1024, 256, 1, 1, 14, "1024->256 k1 @14", 5},
};
static int g_splits_override = 0, g_rows = 1000, g_iters = 60;      // pseudo-options "splits=N", "rows=N" of wtime
static int run_wtime() {
  WBuffers B;
  printf("%-22s %9s | %8s %7s  (weight gradient + slab reduction, N = 256, splits as the product chooses them)\n", "shape", "GFLOP", "us", "TF");
  double tot = 0;
  int row = 0;
  for (const Shape& s : kR50) {
    if (row++ >= g_rows) break;
    const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
    const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
    const int64_t na = (int64_t)s.N * s.H * s.H * s.C, ndy = M * s.K, ndw = (int64_t)s.K * KD;
    const int splits = g_splits_override ? g_splits_override : wgrad_splits(M, s.K, KD);
    B.ensure(na, ndy, ndw, (int64_t)splits * ndw);
    fill(B.a, na, 11u); fill(B.dy, ndy, 29u);
    passl_wgrad_desc d = make_wdesc(s, B.a, B.dy, B.dw, splits > 1 ? B.ws : nullptr, splits > 1 ? (int64_t)splits * ndw : 0, splits);
    for (int i = 0; i < 3; ++i) passl_hip_conv_wgrad(&d, nullptr);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) passl_hip_conv_wgrad(&d, nullptr);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / 20, gf = 2.0 * M * (double)s.K * KD * 1e-9;
    printf("%-22s %9.2f | %8.1f %7.1f  (%d slices)\n", s.note, gf, us, gf / us * 1e3, splits);
    tot += us * s.mult;
  }
  printf("sum over one backward pass of these layers (x multiplicity): %.1f us\n", tot);
  return 0;
}

// ------------------------------------------------------------------------------------------------ timing

static float time_shape(const Shape& s, bool with_stats, Buffers& B, int iters, int* used) {
  const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
  const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
  const int64_t na = (int64_t)s.N * s.H * s.H * s.C, nb = (int64_t)s.K * KD, ny = M * s.K;
  const int tiles = (int)((M + 127) / 128);
  B.ensure(na, nb, ny, (int64_t)tiles * s.K * 3);
  fill(B.a, na, 11u); fill(B.b, nb, 23u);
  passl_conv_desc d = make_desc(s, B.a, B.b, B.y);
  if (with_stats) { d.stats = B.stats; d.stats_tiles = tiles; }
  for (int i = 0; i < 3; ++i) {
    const int rc = passl_hip_conv_igemm(&d, nullptr);
    if (rc != PASSL_OK) { *used = -1; return -1.f; }
  }
  *used = passl_hip_last_igemm_kernel();
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) passl_hip_conv_igemm(&d, nullptr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1000.f / iters;
}

static int run_time(const char* ab_name, int v0, int v1) {
  Buffers B;
  const int iters = 20;
  printf("%-22s %9s | %-5s %8s %7s", "shape (N=256, +stats)", "GFLOP", "kern", "us", "TF");
  if (ab_name) printf(" | %-5s %8s %7s | %s=%d vs %d", "kern", "us", "TF", ab_name, v0, v1);
  printf("\n");
  double tot0 = 0, tot1 = 0;
  for (const Shape& s : kR50) {
    const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
    const double gf = 2.0 * s.N * O * O * (double)s.K * s.R * s.R * s.C * 1e-9;
    int u0 = -1, u1 = -1;
    if (ab_name) passl_hip_set_option(ab_name, v0);
    const float t0 = time_shape(s, true, B, iters, &u0);
    printf("%-22s %9.2f | %-5s %8.1f %7.1f", s.note, gf, kname(u0), t0, gf / t0 * 1e3);
    tot0 += t0 * s.mult;
    if (ab_name) {
      passl_hip_set_option(ab_name, v1);
      const float t1 = time_shape(s, true, B, iters, &u1);
      printf(" | %-5s %8.1f %7.1f | %+.1f %%", kname(u1), t1, gf / t1 * 1e3, (t0 / t1 - 1.0) * 100.0);
      tot1 += t1 * s.mult;
    }
    printf("\n");
  }
  printf("sum over one forward pass of these layers (x multiplicity): %.1f us", tot0);
  if (ab_name) printf(" vs %.1f us", tot1);
  printf("\n");
  return 0;
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// fincheck: passl_hip_bn_finalize / passl_hip_bn_bwd_finalize (one launch per slab since round 5) against the same
// arithmetic done on the host in fp64 from the slab the device produced; fintime: their times, alone and next to a
// stream that keeps the memory system busy (inside the training step these launches always run under load).
static void fill_f32(float* d, int n, uint32_t seed, float lo, float hi) {
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = lo + (hi - lo) * (float)(mix((uint64_t)i, seed) % 1024u) / 1024.0f;
  CK(hipMemcpy(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
}
static int64_t close_f32(const float* a, const double* want, int n, const char* what, double tol) {
  int64_t bad = 0;
  for (int i = 0; i < n; ++i) {
    const double d = fabs((double)a[i] - want[i]);
    if (!(d <= tol * fabs(want[i]) + 1e-7)) { if (bad < 4) printf("    %s[%d]: %.9g, host fp64 %.9g\n", what, i, a[i], want[i]); ++bad; }
  }
  return bad;
}
static int fin_case(int64_t M, int C, int nb) {
  const int64_t n = M * C;
  void *x, *dz;
  float *par, *parb, *cols;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&dz, n * 2));
  const int64_t pf = passl_hip_bn_partial_floats(nb, C, 1), pb = passl_hip_bn_partial_floats(nb, C, 0);
  CK(hipMalloc((void**)&par, pf * 4)); CK(hipMalloc((void**)&parb, pb * 4));
  CK(hipMalloc((void**)&cols, (size_t)16 * C * 4));
  fill(x, n, 61u); fill(dz, n, 71u);
  float *gamma = cols, *beta = cols + C, *rm = cols + 2 * C, *rv = cols + 3 * C, *st = cols + 4 * C;   // st: mean, invstd, scale, shift
  float *dg = cols + 8 * C, *db = cols + 9 * C, *cf = cols + 10 * C;
  fill_f32(gamma, C, 73u, 0.5f, 1.5f); fill_f32(beta, C, 79u, -0.5f, 0.5f);
  fill_f32(rm, C, 83u, -0.1f, 0.1f); fill_f32(rv, C, 89u, 0.5f, 1.5f);
  fill_f32(dg, C, 97u, -1.f, 1.f); fill_f32(db, C, 101u, -1.f, 1.f);
  std::vector<float> h0(16 * C);
  CK(hipMemcpy(h0.data(), cols, h0.size() * 4, hipMemcpyDeviceToHost));
  const int rpb = (int)((M + nb - 1) / nb);
  bool ok = passl_hip_bn_stats(x, par, M, C, nb, PASSL_BF16, nullptr) == PASSL_OK;
  ok = ok && passl_hip_bn_finalize(par, nb, M, C, rpb, gamma, beta, rm, rv, 0.9f, 1e-5f, st, st + C, st + 2 * C, st + 3 * C, nullptr) == PASSL_OK;
  ok = ok && passl_hip_bn_bwd_reduce(dz, nullptr, x, st, st + C, st + 2 * C, st + 3 * C, parb, M, C, nb, 2, PASSL_BF16, nullptr) == PASSL_OK;
  ok = ok && passl_hip_bn_bwd_finalize(parb, nb, M, C, gamma, st, st + C, dg, db, cf, nullptr) == PASSL_OK;
  CK(hipDeviceSynchronize());
  std::vector<float> h1(16 * C), hp(pf), hb(pb);
  CK(hipMemcpy(h1.data(), cols, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hp.data(), par, pf * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), parb, pb * 4, hipMemcpyDeviceToHost));
  int64_t bad = ok ? 0 : 1;
  std::vector<double> w(7 * C), wb(5 * C);
  for (int c = 0; c < C; ++c) {
    const double g0 = hp[(size_t)nb * C * 2 + c];
    double t1 = 0, t2 = 0;
    for (int b = 0; b < nb; ++b) {
      int64_t nr = M - (int64_t)b * rpb; if (nr > rpb) nr = rpb; if (nr <= 0) continue;
      const double p0 = hp[((size_t)b * C + c) * 2], p1 = hp[((size_t)b * C + c) * 2 + 1];
      const double d = (double)hp[(size_t)nb * C * 2 + (size_t)b * C + c] - g0;
      t1 += p0 + nr * d; t2 += p1 + 2.0 * d * p0 + nr * d * d;
    }
    const double dm = t1 / M, mu = g0 + dm;
    double var = t2 / M - dm * dm; if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = h0[c] * is;
    w[c] = (float)mu; w[C + c] = is; w[2 * C + c] = sc; w[3 * C + c] = h0[C + c] - (float)mu * sc;
    w[4 * C + c] = 0.9f * h0[2 * C + c] + 0.1f * (float)mu; w[5 * C + c] = 0.9f * h0[3 * C + c] + 0.1f * (float)var;
    double sg = 0, sgx = 0;
    for (int b = 0; b < nb; ++b) { sg += hb[((size_t)b * C + c) * 2]; sgx += hb[((size_t)b * C + c) * 2 + 1]; }
    const double isd = h1[5 * C + c], gi = (double)h0[c] * isd, B = -gi * isd * sgx / M, Cc = -gi * sg / M - B * (double)h1[4 * C + c];
    wb[c] = h0[8 * C + c] + (float)sgx; wb[C + c] = h0[9 * C + c] + (float)sg; wb[2 * C + c] = (float)gi; wb[3 * C + c] = (float)B; wb[4 * C + c] = (float)Cc;
  }
  bad += close_f32(&h1[4 * C], &w[0], 4 * C, "mean / invstd / scale / shift", 4e-7);
  bad += close_f32(&h1[2 * C], &w[4 * C], 2 * C, "running statistics", 4e-7);
  bad += close_f32(&h1[8 * C], &wb[0], 2 * C, "d-gamma / d-beta", 4e-7);
  bad += close_f32(&h1[10 * C], &wb[2 * C], 3 * C, "coefficients", 2e-6);
  // run to run: bit for bit
  std::vector<float> h2(16 * C);
  CK(hipMemcpy(cols, h0.data(), h0.size() * 4, hipMemcpyHostToDevice));
  passl_hip_bn_finalize(par, nb, M, C, rpb, gamma, beta, rm, rv, 0.9f, 1e-5f, st, st + C, st + 2 * C, st + 3 * C, nullptr);
  passl_hip_bn_bwd_finalize(parb, nb, M, C, gamma, st, st + C, dg, db, cf, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h2.data(), cols, h2.size() * 4, hipMemcpyDeviceToHost));
  if (memcmp(h1.data(), h2.data(), h1.size() * 4) != 0) { printf("    second run differs\n"); ++bad; }
  printf("BatchNorm finalize M=%lld C=%d, %d slab rows: %s\n", (long long)M, C, nb, bad ? "WRONG" : "= host fp64, bit-reproducible");
  for (void* q : {x, dz, (void*)par, (void*)parb, (void*)cols}) CK(hipFree(q));
  return bad ? 1 : 0;
}
static int run_fincheck() {
  int bad = 0;
  bad += fin_case(64LL * 56 * 56, 64, 1568);          // 128-row slabs, as the conv epilogue writes them
  bad += fin_case(32LL * 56 * 56, 256, 784);
  bad += fin_case(50000, 128, 391);                   // ragged last slab
  bad += fin_case(8LL * 14 * 14 + 7, 1024, 13);
  bad += fin_case(64LL * 112 * 112, 64, 6272);        // more rows than one batch of loads (4096)
  bad += fin_case(300, 2048, 3);
  bad += fin_case(6272LL * 128, 256, 6272);          // 10 / 7 row segments combined by the last block to arrive
  printf(bad ? "FINALIZE CHECK FAILED (%d)\n" : "FINALIZE CHECK OK\n", bad);
  return bad ? 1 : 0;
}
__global__ void hog_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
// finstress: the multi-segment finalize launches (last block of a channel group to arrive combines the segment totals)
// under load.  A second stream streams 1 GB copies the whole time; every iteration switches the input (three data
// sets in turn: a stale segment total of the previous launch gives a wrong answer), recomputes the slab and finalizes:
// every result must equal the first result of its data set BIT FOR BIT (the first one is checked by fincheck's
// arithmetic at the same sizes), forward and backward.
static int run_finstress(int iters) {
  struct S { int64_t M; int C; int nb; };
  const S shapes[] = {{6272LL * 128, 64, 6272}, {6272LL * 128, 256, 6272}, {25088LL * 128, 64, 25088}, {1568LL * 128, 512, 1568}};
  hipStream_t hog;
  CK(hipStreamCreate(&hog));
  void *hs, *hd;
  const int64_t hog_bytes = 1ll << 30;
  CK(hipMalloc(&hs, hog_bytes)); CK(hipMalloc(&hd, hog_bytes));
  int failures = 0;
  for (const S& sh : shapes) {
    const int64_t n = sh.M * sh.C;
    void *x[3], *dz;
    float *par, *parb, *cols;
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&x[k], n * 2)); fill(x[k], n, 61u + 100u * k); }
    CK(hipMalloc(&dz, n * 2)); fill(dz, n, 71u);
    const int64_t pf = passl_hip_bn_partial_floats(sh.nb, sh.C, 1), pb = passl_hip_bn_partial_floats(sh.nb, sh.C, 0);
    CK(hipMalloc((void**)&par, pf * 4)); CK(hipMalloc((void**)&parb, pb * 4));
    CK(hipMalloc((void**)&cols, (size_t)16 * sh.C * 4));
    std::vector<float> h0(16 * sh.C), h(16 * sh.C), first[3];
    for (int i = 0; i < 16 * sh.C; ++i) h0[i] = 0.5f + (float)(mix((uint64_t)i, 73u) % 1024u) / 1024.0f;
    float *gamma = cols, *beta = cols + sh.C, *rm = cols + 2 * sh.C, *rv = cols + 3 * sh.C, *st = cols + 4 * sh.C;
    float *dg = cols + 8 * sh.C, *db = cols + 9 * sh.C, *cf = cols + 10 * sh.C;
    int64_t flips = 0;
    for (int it = 0; it < iters; ++it) {
      const int set = it % 3;
      for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(hog_kernel, dim3(2048), dim3(256), 0, hog, (const uint4*)hs, (uint4*)hd, hog_bytes / 16);
      CK(hipMemcpyAsync(cols, h0.data(), h0.size() * 4, hipMemcpyHostToDevice, 0));
      bool ok = passl_hip_bn_stats(x[set], par, sh.M, sh.C, sh.nb, PASSL_BF16, nullptr) == PASSL_OK;
      ok = ok && passl_hip_bn_finalize(par, sh.nb, sh.M, sh.C, 128, gamma, beta, rm, rv, 0.9f, 1e-5f, st, st + sh.C, st + 2 * sh.C, st + 3 * sh.C, nullptr) == PASSL_OK;
      ok = ok && passl_hip_bn_bwd_reduce(dz, nullptr, x[set], st, st + sh.C, st + 2 * sh.C, st + 3 * sh.C, parb, sh.M, sh.C, sh.nb, 2, PASSL_BF16, nullptr) == PASSL_OK;
      ok = ok && passl_hip_bn_bwd_finalize(parb, sh.nb, sh.M, sh.C, gamma, st, st + sh.C, dg, db, cf, nullptr) == PASSL_OK;
      if (!ok) { printf("launch failed\n"); return 1; }
      CK(hipMemcpyAsync(h.data(), cols, h.size() * 4, hipMemcpyDeviceToHost, 0));
      CK(hipStreamSynchronize(0));
      if (first[set].empty()) first[set] = h;
      else if (memcmp(first[set].data(), h.data(), h.size() * 4) != 0) ++flips;
    }
    CK(hipStreamSynchronize(hog));
    const bool distinct = memcmp(first[0].data(), first[1].data(), first[0].size() * 4) != 0;
    printf("%lld x %d (%d slab rows): %d launches under load, %lld not bit-identical to the first of their data set%s\n", (long long)sh.M, sh.C,
           sh.nb, iters, (long long)flips, distinct ? "" : "; DATA SETS GIVE THE SAME RESULT");
    if (flips || !distinct) ++failures;
    for (int k = 0; k < 3; ++k) CK(hipFree(x[k]));
    for (void* q : {dz, (void*)par, (void*)parb, (void*)cols}) CK(hipFree(q));
  }
  printf(failures ? "FINALIZE STRESS FAILED (%d)\n" : "FINALIZE STRESS OK\n", failures);
  return failures ? 1 : 0;
}

static int run_fintime() {
  struct S { int nb; int C; };
  const S shapes[] = {{25088, 64}, {6272, 64}, {6272, 256}, {1568, 128}, {1568, 512}, {392, 256}, {392, 1024}, {98, 512}, {98, 2048}};
  hipStream_t hog;
  CK(hipStreamCreate(&hog));
  void *hs, *hd;
  const int64_t hog_bytes = 1ll << 30;
  CK(hipMalloc(&hs, hog_bytes)); CK(hipMalloc(&hd, hog_bytes));
  printf("%-14s %10s %10s %10s %10s   (us per launch: forward finalize alone | under load, backward finalize alone | under load)\n", "slab rows x C", "fwd", "fwd load", "bwd", "bwd load");
  for (const S& sh : shapes) {
    const int64_t M = (int64_t)sh.nb * 128;
    float *par, *cols;
    const int64_t pf = passl_hip_bn_partial_floats(sh.nb, sh.C, 1);
    CK(hipMalloc((void**)&par, pf * 4)); CK(hipMalloc((void**)&cols, (size_t)16 * sh.C * 4));
    CK(hipMemset(par, 0, pf * 4));
    fill_f32(cols, 16 * sh.C, 73u, 0.5f, 1.5f);
    float* st = cols + 4 * sh.C;
    float t[4];
    for (int form = 0; form < 4; ++form) {
      const bool load = form & 1, bwd = form >= 2;
      if (load) for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(hog_kernel, dim3(2048), dim3(256), 0, hog, (const uint4*)hs, (uint4*)hd, hog_bytes / 16);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int it = -3; it < 50; ++it) {
        if (it == 0) CK(hipEventRecord(e0, 0));
        if (!bwd) passl_hip_bn_finalize(par, sh.nb, M, sh.C, 128, cols, cols + sh.C, cols + 2 * sh.C, cols + 3 * sh.C, 0.9f, 1e-5f, st, st + sh.C, st + 2 * sh.C, st + 3 * sh.C, nullptr);
        else passl_hip_bn_bwd_finalize(par, sh.nb, M, sh.C, cols, st, st + sh.C, cols + 8 * sh.C, cols + 9 * sh.C, cols + 10 * sh.C, nullptr);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t[form], e0, e1));
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
      CK(hipDeviceSynchronize());
    }
    char name[32]; snprintf(name, sizeof(name), "%d x %d", sh.nb, sh.C);
    printf("%-14s %10.1f %10.1f %10.1f %10.1f\n", name, t[0] * 20.f, t[1] * 20.f, t[2] * 20.f, t[3] * 20.f);
    CK(hipFree(par)); CK(hipFree(cols));
  }
  return 0;
}

// sweep: every argument is one configuration "name=value,name=value,...": the 3x3 / stride-1 cases are checked
// (bit-exact) and the four ResNet-50 3x3 shapes timed under each, one row per configuration.
// ------------------------------------------------------------------------------------------------ stem pool
// The fused BatchNorm + ReLU + max-pool backward (csrc/stem_pool.hip): the second form of its reduce pass against the
// first on the same inputs — the slab's column sums to fp32 rounding (and dx, which both take from the same apply
// pass: a launch-to-launch identity check) — and the times of both.
struct PoolBufs {
  void *x = nullptr, *out = nullptr, *dy = nullptr, *dx0 = nullptr, *dx1 = nullptr;
  uint8_t* idx = nullptr;
  float *par = nullptr, *slab0 = nullptr, *slab1 = nullptr;     // par: scale, shift, mean, invstd, coef[3]
  void release() {
    for (void* q : {x, out, dy, dx0, dx1, (void*)idx, (void*)par, (void*)slab0, (void*)slab1}) if (q) CK(hipFree(q));
  }
};
static void pool_upload(void* dst, int64_t n, uint32_t seed, int es) {
  if (es == 2) { fill(dst, n, seed); return; }
  std::vector<float> h(n);
  for (int64_t i = 0; i < n; ++i) h[i] = (float)ival((uint64_t)i, seed) * 0.0625f;
  CK(hipMemcpy(dst, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
}
static int pool_setup(PoolBufs& B, int N, int H, int W, int C, int dtype, int max_rows) {
  const int es = dtype == PASSL_BF16 ? 2 : 4;
  const int P = (H - 1) / 2 + 1, Q = (W - 1) / 2 + 1;
  const int64_t nx = (int64_t)N * H * W * C, no = (int64_t)N * P * Q * C;
  CK(hipMalloc(&B.x, nx * es)); CK(hipMalloc(&B.dx0, nx * es)); CK(hipMalloc(&B.dx1, nx * es));
  CK(hipMalloc(&B.out, no * es)); CK(hipMalloc(&B.dy, no * es)); CK(hipMalloc((void**)&B.idx, no));
  CK(hipMalloc((void**)&B.par, (size_t)7 * C * 4));
  const size_t slab = ((size_t)max_rows * C * 2 + 16 * C * 4) * 4;
  CK(hipMalloc((void**)&B.slab0, slab)); CK(hipMalloc((void**)&B.slab1, slab));
  pool_upload(B.x, nx, 5u, es); pool_upload(B.dy, no, 17u, es);
  fill_f32(B.par, C, 3u, -1.5f, 1.5f);            // scale (both signs)
  fill_f32(B.par + C, C, 7u, -0.5f, 0.5f);        // shift
  fill_f32(B.par + 2 * C, C, 9u, -0.5f, 0.5f);    // mean
  fill_f32(B.par + 3 * C, C, 13u, 0.5f, 2.0f);    // invstd
  fill_f32(B.par + 4 * C, 3 * C, 19u, -1.0f, 1.0f);   // coef A, B, C
  const int rc = passl_hip_bn_relu_maxpool_fwd(B.x, B.par, B.par + C, B.out, B.idx, N, H, W, C, dtype, nullptr);
  if (rc != PASSL_OK) { printf("    bn_relu_maxpool_fwd -> %d\n", rc); return rc; }
  CK(hipDeviceSynchronize());
  return 0;
}
static int pool_backward(PoolBufs& B, int N, int H, int W, int C, int dtype, float* slab, void* dx, int* rows) {
  const int nb = passl_hip_bn_relu_maxpool_blocks(N, H, W, C);
  *rows = nb;
  int rc = passl_hip_bn_relu_maxpool_bwd_reduce(B.dy, B.idx, B.x, B.par + 2 * C, B.par + 3 * C, B.par, B.par + C, slab, nb, N, H,
                                                W, C, dtype, nullptr);
  if (rc == PASSL_OK)
    rc = passl_hip_bn_relu_maxpool_bwd_apply(B.dy, B.idx, B.x, B.par + 4 * C, B.par, B.par + C, dx, N, H, W, C, dtype, nullptr);
  if (rc != PASSL_OK) printf("    bn_relu_maxpool_bwd -> %d (%s)\n", rc, passl_hip_strerror(rc));
  CK(hipDeviceSynchronize());
  return rc;
}
static void pool_colsums(const float* slab_dev, int rows, int C, std::vector<double>& s) {
  std::vector<float> h((size_t)rows * C * 2);
  CK(hipMemcpy(h.data(), slab_dev, h.size() * 4, hipMemcpyDeviceToHost));
  s.assign((size_t)C * 2, 0.0);
  for (int b = 0; b < rows; ++b)
    for (int c = 0; c < C * 2; ++c) s[c] += (double)h[(size_t)b * C * 2 + c];
}
static int run_poolcheck() {
  struct Case { int N, H, W, C, dtype; };
  const Case cases[] = {{3, 112, 112, 64, PASSL_BF16}, {2, 15, 15, 64, PASSL_BF16}, {2, 30, 18, 128, PASSL_BF16},
                        {3, 16, 16, 8, PASSL_BF16},    {1, 7, 9, 256, PASSL_BF16},  {2, 15, 17, 64, PASSL_F32},
                        {1, 112, 112, 64, PASSL_F32}};
  int failures = 0;
  for (const Case& c : cases) {
    const int es = c.dtype == PASSL_BF16 ? 2 : 4;
    const int64_t nx = (int64_t)c.N * c.H * c.W * c.C;
    PoolBufs B;
    passl_hip_set_option("stem_pool_form", 0);
    const int rows_max = passl_hip_bn_relu_maxpool_blocks(c.N, c.H, c.W, c.C) + 4096;
    if (pool_setup(B, c.N, c.H, c.W, c.C, c.dtype, rows_max)) { ++failures; B.release(); continue; }
    int rows0 = 0;
    CK(hipMemset(B.dx0, 0xff, nx * es));
    if (pool_backward(B, c.N, c.H, c.W, c.C, c.dtype, B.slab0, B.dx0, &rows0)) { ++failures; B.release(); continue; }
    std::vector<double> want, got;
    pool_colsums(B.slab0, rows0, c.C, want);
    std::vector<uint8_t> h0((size_t)nx * es), h1((size_t)nx * es);
    CK(hipMemcpy(h0.data(), B.dx0, h0.size(), hipMemcpyDeviceToHost));
    passl_hip_set_option("stem_pool_form", 1);
      for (int wgs : {1024, 5, 4096}) {
        passl_hip_set_option("stem_pool_wgs", wgs);
        int rows1 = 0;
        CK(hipMemset(B.dx1, 0xff, nx * es));
        if (pool_backward(B, c.N, c.H, c.W, c.C, c.dtype, B.slab1, B.dx1, &rows1)) { ++failures; continue; }
        CK(hipMemcpy(h1.data(), B.dx1, h1.size(), hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { if (bad < 3) printf("    dx byte %zu: %02x vs %02x\n", i, h0[i], h1[i]); ++bad; }
        pool_colsums(B.slab1, rows1, c.C, got);
        int64_t sbad = 0;
        double scale_ref = 0;
        for (double v : want) scale_ref = fmax(scale_ref, fabs(v));
        for (size_t i = 0; i < want.size(); ++i)
          if (!(fabs(got[i] - want[i]) <= 2e-5 * fabs(want[i]) + 1e-6 * scale_ref)) { if (sbad < 3) printf("    column sum %zu: %.9g vs %.9g\n", i, got[i], want[i]); ++sbad; }
        printf("%dx%dx%dx%d %s, %4d workgroups asked (%4d slab rows; first form %4d): dx %s, sums %s\n", c.N, c.H, c.W, c.C,
               es == 2 ? "bf16" : "fp32", wgs, rows1, rows0, bad ? "DIFFERS" : "identical", sbad ? "DIFFER" : "agree");
        if (bad || sbad) ++failures;
      }
    passl_hip_set_option("stem_pool_wgs", 1024);
    B.release();
  }
  printf(failures ? "POOL CHECK FAILED (%d)\n" : "POOL CHECK OK\n", failures);
  return failures ? 1 : 0;
}
static int run_pooltime() {
  const int N = 256, H = 112, W = 112, C = 64, dtype = PASSL_BF16;
  PoolBufs B;
  passl_hip_set_option("stem_pool_form", 0);
  const int rows_max = passl_hip_bn_relu_maxpool_blocks(N, H, W, C) + 8192;
  if (pool_setup(B, N, H, W, C, dtype, rows_max)) return 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto&& fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / 20;
  };
  const double fwd = time_it([&] { passl_hip_bn_relu_maxpool_fwd(B.x, B.par, B.par + C, B.out, B.idx, N, H, W, C, dtype, nullptr); });
  printf("N = 256, 112 x 112 x 64 bf16.  forward %.1f us (514 + 51 MB)\n", fwd);
  printf("%-34s %10s %10s\n", "backward", "reduce us", "apply us");
  auto row = [&](const char* name) {
    const int nb = passl_hip_bn_relu_maxpool_blocks(N, H, W, C);
    const double r = time_it([&] { passl_hip_bn_relu_maxpool_bwd_reduce(B.dy, B.idx, B.x, B.par + 2 * C, B.par + 3 * C, B.par, B.par + C, B.slab0, nb, N, H, W, C, dtype, nullptr); });
    const double a = time_it([&] { passl_hip_bn_relu_maxpool_bwd_apply(B.dy, B.idx, B.x, B.par + 4 * C, B.par, B.par + C, B.dx0, N, H, W, C, dtype, nullptr); });
    printf("%-34s %10.1f %10.1f   (%d slab rows)\n", name, r, a, nb);
  };
  row("first form");
  passl_hip_set_option("stem_pool_form", 1);
  for (int wgs : {512, 768, 1024, 1536, 2048, 3136}) {
    passl_hip_set_option("stem_pool_wgs", wgs);
    char name[64];
    snprintf(name, sizeof(name), "second form of the reduce, %d wgs", wgs);
    row(name);
  }
  passl_hip_set_option("stem_pool_wgs", 1024);
  B.release();
  return 0;
}

static int apply_config(const char* cfg) {
  std::string c(cfg);
  size_t pos = 0;
  while (pos < c.size()) {
    size_t end = c.find(',', pos);
    if (end == std::string::npos) end = c.size();
    const std::string kv = c.substr(pos, end - pos);
    const size_t eq = kv.find('=');
    if (eq == std::string::npos) return -1;
    if (passl_hip_set_option(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1)) != PASSL_OK) return -1;
    pos = end + 1;
  }
  return 0;
}
static int run_sweep(int n, char** cfgs) {
  const Shape cases[] = {
      {3, 64, 64, 3, 1, 56, "stage-1", 0},  {2, 128, 128, 3, 1, 28, "stage-2", 0}, {5, 256, 256, 3, 1, 14, "stage-3", 0},
      {7, 512, 512, 3, 1, 7, "stage-4", 0}, {3, 64, 128, 3, 1, 20, "w20", 0},      {1, 128, 64, 3, 1, 9, "9x9", 0},
      {3, 128, 128, 3, 1, 16, "16x16", 0},  {5, 64, 64, 3, 1, 8, "8x8", 0},
  };
  const int ncases = sizeof(cases) / sizeof(cases[0]);
  std::vector<std::vector<int32_t>> refs(ncases);
  for (int i = 0; i < ncases; ++i) reference(cases[i], refs[i]);
  Buffers B;
  printf("%-58s %-6s |", "configuration", "check");
  for (int i = 0; i < 4; ++i) printf(" %-16s", kR50[i].note);
  printf("   (us, kernel)\n");
  for (int c = 0; c < n; ++c) {
    if (apply_config(cfgs[c]) != 0) { printf("%-58s bad option\n", cfgs[c]); continue; }
    int wrong = 0;
    for (int i = 0; i < ncases; ++i)
      for (int v = 0; v < 4; ++v) {
        int used = -1;
        if (check_case(cases[i], v, B, &used, refs[i]) != 0) ++wrong;
      }
    printf("%-58s %-6s |", cfgs[c], wrong ? "WRONG" : "exact");
    for (int i = 0; i < 4; ++i) {
      int used = -1;
      const float t = time_shape(kR50[i], true, B, 20, &used);
      printf(" %8.1f %-7s", t, kname(used));
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}

// vtime: the four epilogue variants (ReLU, fused statistics, residual + ReLU, BatchNorm-backward) of the K = 64 1x1
// layers at N = 256 — the launches the 4-workgroup form of the register-staged kernel takes (PASSL_IGEMM_LEAN).
static int run_vtime() {
  const Shape shapes[] = {kR50[6], {256, 64, 64, 1, 1, 56, "64->64 k1 @56", 1}, kR50[8]};
  Buffers B;
  printf("%-22s %9s %9s %9s %9s   (us: relu | statistics | residual + relu | BatchNorm-backward epilogue)\n", "shape (N=256)", "relu", "stats", "resid", "bnb");
  for (const Shape& s : shapes) {
    const int pad = s.R / 2, O = (s.H + 2 * pad - s.R) / s.stride + 1;
    const int64_t M = (int64_t)s.N * O * O, KD = (int64_t)s.R * s.R * s.C;
    const int64_t na = (int64_t)s.N * s.H * s.H * s.C, nb = (int64_t)s.K * KD, ny = M * s.K;
    const int tiles = (int)((M + 127) / 128);
    B.ensure(na, nb, ny, passl_hip_bn_partial_floats(tiles, s.K, 1));
    B.ensure_aux(ny);
    fill(B.a, na, 11u); fill(B.b, nb, 23u); fill(B.aux, ny, 37u);
    std::vector<float> hcols(4 * 4096, 0.5f);
    CK(hipMemcpy(B.cols, hcols.data(), hcols.size() * 4, hipMemcpyHostToDevice));
    float t[4];
    for (int v = 0; v < 4; ++v) {
      passl_conv_desc d = make_desc(s, B.a, B.b, B.y);
      d.relu = (v == V_RELU || v == V_RESIDUAL) ? 1 : 0;
      if (v == V_STATS) { d.stats = B.stats; d.stats_tiles = tiles; }
      if (v == V_RESIDUAL) d.residual = B.aux;
      if (v == V_BNB) {
        d.bnb_y = B.aux; d.bnb_mean = B.cols; d.bnb_invstd = B.cols + 4096; d.bnb_scale = B.cols + 8192; d.bnb_shift = B.cols + 12288;
        d.bnb_partial = B.stats; d.bnb_relu = 2;
      }
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int it = -3; it < 20; ++it) {
        if (it == 0) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
        if (passl_hip_conv_igemm(&d, nullptr) != PASSL_OK) { printf("conv failed\n"); return 1; }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&t[v], e0, e1));
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("%-22s %9.1f %9.1f %9.1f %9.1f\n", s.note, t[0] * 50.f, t[1] * 50.f, t[2] * 50.f, t[3] * 50.f);
  }
  return 0;
}

// ablate: the register-staged kernel's debug switches (PASSL_IGEMM_DBG: 2 = return before the epilogue, 4 = no A
// loads, 8 = no MFMAs) on the 1x1 shapes it serves: what a tile's time is made of.
static int run_ablate() {
  setenv("PASSL_IGEMM_DBG_DYNAMIC", "1", 1);
  const int sets[] = {0, 16, 16 | 32, 2, 8, 4, 2 | 8, 2 | 4 | 8};
  const Shape shapes[] = {kR50[6], kR50[7], kR50[8], kR50[10], {256, 64, 64, 1, 1, 56, "64->64 k1 @56", 1}};
  Buffers B;
  printf("%-22s", "PASSL_IGEMM_DBG =");
  for (int v : sets) printf(" %8d", v);
  printf("   (us, with fused statistics; 16 no statistics reduction / slab, 32 no statistics arithmetic, 2 no epilogue, 4 no A loads, 8 no MFMA)\n");
  for (const Shape& sh : shapes) {
    printf("%-22s", sh.note);
    for (int v : sets) {
      char buf[16]; snprintf(buf, sizeof(buf), "%d", v);
      setenv("PASSL_IGEMM_DBG", buf, 1);
      int used = -1;
      const float t = time_shape(sh, true, B, 20, &used);
      printf(" %8.1f", t);
    }
    printf("\n");
  }
  setenv("PASSL_IGEMM_DBG", "0", 1);
  return 0;
}

int main(int argc, char** argv) {
  if (passl_hip_abi_version() != PASSL_HIP_ABI_VERSION) {
    fprintf(stderr, "kbench was built against ABI %d of include/passl_hip.h, libpassl_hip.so is ABI %d: run tools/build_kbench.sh\n",
            PASSL_HIP_ABI_VERSION, passl_hip_abi_version());
    return 3;
  }
  if (argc < 2) { fprintf(stderr, "usage: kbench check|time|ab [name=value ...]\n"); return 2; }
  const std::string mode = argv[1];
  if (mode == "sweep") { printf("libpassl_hip ABI %d\n", passl_hip_abi_version()); return run_sweep(argc - 2, argv + 2); }
  const char* ab_name = nullptr; int v0 = 0, v1 = 0;
  static char namebuf[128];
  for (int i = 2; i < argc; ++i) {
    const char* eq = strchr(argv[i], '=');
    if (!eq) { fprintf(stderr, "bad option %s\n", argv[i]); return 2; }
    std::string name(argv[i], eq - argv[i]);
    const char* comma = strchr(eq + 1, ',');
    if (mode == "ab" && comma && !ab_name) {
      snprintf(namebuf, sizeof(namebuf), "%s", name.c_str());
      ab_name = namebuf; v0 = atoi(eq + 1); v1 = atoi(comma + 1);
      continue;
    }
    if (name == "splits") { g_splits_override = atoi(eq + 1); continue; }
    if (name == "rows") { g_rows = atoi(eq + 1); continue; }
    if (name == "iters") { g_iters = atoi(eq + 1); continue; }
    const int rc = passl_hip_set_option(name.c_str(), atoi(eq + 1));
    if (rc != PASSL_OK) { fprintf(stderr, "set_option(%s) -> %d\n", name.c_str(), rc); return 2; }
  }
  printf("libpassl_hip ABI %d\n", passl_hip_abi_version());
  if (mode == "check") return run_check();
  if (mode == "ablate") return run_ablate();
  if (mode == "vtime") return run_vtime();
  if (mode == "wcheck") return run_wcheck();
  if (mode == "fincheck") return run_fincheck();
  if (mode == "fintime") return run_fintime();
  if (mode == "poolcheck") return run_poolcheck();
  if (mode == "pooltime") return run_pooltime();
  if (mode == "finstress") return run_finstress(g_iters);
  if (mode == "wtime") return run_wtime();
  if (mode == "time") return run_time(nullptr, 0, 0);
  if (mode == "ab") { if (!ab_name) { fprintf(stderr, "ab needs name=v0,v1\n"); return 2; } return run_time(ab_name, v0, v1); }
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
