// Probe: semantics and bank behaviour of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
#define LDS3(p) ((__attribute__((address_space(3))) bf16x4_t*)(p))

__global__ void sem(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  // lane l points at 8 bytes: elements 4l .. 4l+3
  bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS3(lds + threadIdx.x * 4));
  uint2 u = __builtin_bit_cast(uint2, v);
  out[threadIdx.x * 4 + 0] = u.x & 0xffff; out[threadIdx.x * 4 + 1] = u.x >> 16;
  out[threadIdx.x * 4 + 2] = u.y & 0xffff; out[threadIdx.x * 4 + 3] = u.y >> 16;
}

// timing: each of 4 waves does ITER x 16 tr reads of a [64 m][128 ch] tile, pitch 256 B
template <int MODE>
__global__ void timing(uint64_t* cyc, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16384; i += 256) ((uint32_t*)lds)[i] = i;
  __syncthreads();
  const int g = l >> 4, r4 = (l >> 2) & 3, c4 = l & 3;
  uint32_t acc = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {        // 8 fragments (16 channels each) x 2 halves
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = (it & 1) * 32 + 8 * g + 4 * h + r4;
        const int colb = (w & 1) * 128 + f * 32 * 0 + ((f * 32) & 127) + c4 * 8;   // byte col within 256
        int off;
        if (MODE == 0) off = row * 256 + colb;                                  // linear
        else if (MODE == 1) {                                                   // pair swizzle
          const int hsw = (row & 3) | (((row >> 3) & 1) << 2);
          off = row * 256 + (colb ^ (hsw << 5));
        } else if (MODE == 2) {                                                 // 16B-chunk swizzle by row&15
          off = row * 256 + (colb ^ ((row & 15) << 4));
        } else {                                                                // [4-row][16-col] blocks contiguous (128 B per 16-lane group)
          off = ((row >> 2) * 8 + (colb >> 5)) * 128 + r4 * 32 + c4 * 8;
        }
        bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS3(lds + off));
        uint2 u = __builtin_bit_cast(uint2, v);
        acc += u.x ^ u.y;
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}

// reference: ds_read_b128 conflict-free
__global__ void timing_b128(uint64_t* cyc, uint32_t* sink) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16384; i += 256) ((uint32_t*)lds)[i] = i;
  __syncthreads();
  uint32_t acc = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int row = (it & 1) * 64 + (f & 3) * 16 + (l & 15);
      const int slot = ((f >> 2) * 4 + (l >> 4)) ^ ((row >> 1) & 7);
      const uint4 v = *reinterpret_cast<const uint4*>(lds + w * 0 + row * 128 + slot * 16);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}

int main() {
  uint16_t* dout; hipMalloc(&dout, 512);
  sem<<<1, 64>>>(dout);
  uint16_t h[256]; hipMemcpy(h, dout, 512, hipMemcpyDeviceToHost);
  printf("semantics (lane: 4 element indices), lane l addressed elements 4l..4l+3\n");
  for (int l = 0; l < 64; ++l) printf("%2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : "   ");
  uint64_t* dc; uint32_t* ds; hipMalloc(&dc, 8 * 1024); hipMalloc(&ds, 4 * 256 * 256);
  uint64_t hc[4];
  const char* names[4] = {"linear", "pair-swizzle h(row)", "chunk-swizzle row&15", "4x16 blocks contiguous"};
#define RUN(K, name) for (int rep = 0; rep < 2; ++rep) { K<<<1, 256>>>(dc, ds); hipDeviceSynchronize(); } \
  hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost); printf("%-28s cycles/wave-instr: %.2f %.2f %.2f %.2f\n", name, hc[0]/4096.0, hc[1]/4096.0, hc[2]/4096.0, hc[3]/4096.0);
  RUN(timing<0>, names[0]); RUN(timing<1>, names[1]); RUN(timing<2>, names[2]); RUN(timing<3>, names[3]);
  for (int rep = 0; rep < 2; ++rep) { timing_b128<<<1, 256>>>(dc, ds); hipDeviceSynchronize(); }
  hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost); printf("%-28s cycles/wave-instr: %.2f %.2f %.2f %.2f\n", "ds_read_b128 swizzled", hc[0]/2048.0, hc[1]/2048.0, hc[2]/2048.0, hc[3]/2048.0);
  return 0;
}
