// store_probe: what bounds the store phase of a GEMM epilogue on MI355X?  (round 6; standalone, no library)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/probe/store_probe.cpp -o scratch/probe/store_probe
// Writes (and optionally reads) a [M][NCOLS] bf16 matrix the way a 128 x BN output tile does: thread -> (row, 16-byte
// chunk), rows NCOLS*2 bytes apart.  Variants: waves per workgroup, workgroups per CU (LDS ballast), one tile per
// workgroup vs persistent, a dependent 16-KB-per-tile read in front (the A operand), 16- vs 8-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct P {
  uint4* y; const uint4* a; int ntiles, tiles_n, row_chunks /* 16-byte chunks per full row */, bn_chunks /* per tile row */;
  int rows;          // rows per tile (128)
  int read_chunks;   // 16-byte chunks of A read per tile (0 = none); read before the stores, waited for
  int persist;       // 1: v = bid, bid + grid, ...
  int nt;            // nontemporal stores
};

template <int THREADS>
__global__ void __launch_bounds__(THREADS) tile_store(const P p) {
  extern __shared__ char ballast[];
  const int tid = threadIdx.x;
  uint4 v = make_uint4(tid, blockIdx.x, 3, 4);
  for (int t = blockIdx.x; t < p.ntiles; t += p.persist ? gridDim.x : p.ntiles) {
    const int mt = t / p.tiles_n, nt = t - mt * p.tiles_n;
    if (p.read_chunks) {
      // the A operand of this tile: read_chunks x 16 B, coalesced, all of it waited for (as the MFMAs need it)
      uint4 acc = make_uint4(0, 0, 0, 0);
      const uint4* src = p.a + (int64_t)mt * p.read_chunks;
      for (int i = tid; i < p.read_chunks; i += THREADS) { const uint4 x = src[i]; acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w; }
      v.x ^= acc.x; v.y ^= acc.y; v.z ^= acc.z; v.w ^= acc.w;
      if (tid == 0 && v.x == 0x12345678u) ballast[0] = 1;
      __syncthreads();
    }
    const int per_tile = p.rows * p.bn_chunks;
    for (int c = tid; c < per_tile; c += THREADS) {
      const int row = c / p.bn_chunks, cc = c - row * p.bn_chunks;
      uint4* dst = p.y + ((int64_t)mt * p.rows + row) * p.row_chunks + (int64_t)nt * p.bn_chunks + cc;
      if (p.nt) {
        typedef __attribute__((ext_vector_type(4))) unsigned int u4;
        u4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u4*>(dst));
      } else {
        *dst = v;
      }
    }
  }
}

static float run(const P& p, int threads, int grid, int lds, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (threads == 256) hipLaunchKernelGGL(tile_store<256>, dim3(grid), dim3(256), lds, 0, p);
    else if (threads == 512) hipLaunchKernelGGL(tile_store<512>, dim3(grid), dim3(512), lds, 0, p);
    else hipLaunchKernelGGL(tile_store<1024>, dim3(grid), dim3(1024), lds, 0, p);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1000.f / iters;
}

int main() {
  const int64_t M = 256ll * 56 * 56;              // 802816 rows
  uint4 *y, *a;
  CK(hipMalloc((void**)&y, M * 256 * 2));          // up to 256 columns of bf16
  CK(hipMalloc((void**)&a, M * 256 * 2));
  CK(hipMemset(a, 1, M * 256 * 2));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_store<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_store<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_store<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  printf("%-64s %8s %8s\n", "variant (802816 x NCOLS bf16 out)", "us", "TB/s");
  struct V { const char* name; int ncols, bn, threads, wg_per_cu, persist, read_cols, nt; };
  const V vs[] = {
      {"fill-like: NCOLS 256, BN 256 (full rows), 256 thr, 8 WG/CU", 256, 256, 256, 8, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 256 thr, 8 WG/CU", 256, 128, 256, 8, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 256 thr, 4 WG/CU", 256, 128, 256, 4, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 256 thr, 3 WG/CU", 256, 128, 256, 3, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 256 thr, 2 WG/CU", 256, 128, 256, 2, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 256 thr, 1 WG/CU", 256, 128, 256, 1, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 512 thr, 2 WG/CU", 256, 128, 512, 2, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 512 thr, 4 WG/CU", 256, 128, 512, 4, 0, 0, 0},
      {"tile 128x128 of NCOLS 256, 1024 thr, 2 WG/CU", 256, 128, 1024, 2, 0, 0, 0},
      {"tile 128x128, 256 thr, 4 WG/CU, nontemporal stores", 256, 128, 256, 4, 0, 0, 1},
      {"tile 128x128, 256 thr, persistent 4 WG/CU", 256, 128, 256, 4, 1, 0, 0},
      {"tile 128x128, 256 thr, persistent 2 WG/CU", 256, 128, 256, 2, 1, 0, 0},
      {"tile 128x128, 256 thr, persistent 8 WG/CU", 256, 128, 256, 8, 1, 0, 0},
      {"tile 128x128 + read 128x64 A first, 256 thr, 4 WG/CU", 256, 128, 256, 4, 0, 64, 0},
      {"tile 128x128 + read 128x64 A first, 256 thr, 3 WG/CU", 256, 128, 256, 3, 0, 64, 0},
      {"tile 128x128 + read 128x64 A first, 256 thr, 8 WG/CU", 256, 128, 256, 8, 0, 64, 0},
      {"tile 128x128 + read 128x64 A first, persistent 4 WG/CU", 256, 128, 256, 4, 1, 64, 0},
      {"tile 128x128 + read 128x64 A first, 512 thr, 4 WG/CU", 256, 128, 512, 4, 0, 64, 0},
      {"tile 128x64 of NCOLS 64 + read 128x256 A (256->64), 4 WG/CU", 64, 64, 256, 4, 0, 256, 0},
      {"tile 128x64 of NCOLS 64 + read 128x256 A (256->64), 8 WG/CU", 64, 64, 256, 8, 0, 256, 0},
      {"tile 128x64 of NCOLS 64 + read 128x256 A, persistent 8 WG/CU", 64, 64, 256, 8, 1, 256, 0},
  };
  for (const V& v : vs) {
    P p;
    p.y = y; p.a = a;
    p.rows = 128;
    p.tiles_n = v.ncols / v.bn;
    p.ntiles = (int)(M / 128) * p.tiles_n;
    p.row_chunks = v.ncols / 8; p.bn_chunks = v.bn / 8;
    p.read_chunks = v.read_cols ? 128 * v.read_cols / 8 : 0;
    p.persist = v.persist; p.nt = v.nt;
    // LDS ballast: floor(160 KB / wg_per_cu) - 1 KB so that exactly wg_per_cu workgroups fit
    int lds = (160 * 1024) / v.wg_per_cu - 1024;
    if (v.wg_per_cu >= 8) lds = 16 * 1024;
    const int grid = v.persist ? 256 * v.wg_per_cu : p.ntiles;
    const float us = run(p, v.threads, grid, lds, 10);
    const double bytes = (double)M * v.ncols * 2 + (v.read_cols ? (double)M * v.read_cols * 2 : 0.0);
    printf("%-64s %8.1f %8.2f\n", v.name, us, bytes / us * 1e-6);
  }
  return 0;
}
