// Probe: buffer_load_dwordx4 ... lds — out-of-range lanes must land as zeros in LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const uint32_t* g, uint32_t* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  // lanes 0..31 in range (reversed order), lanes 32..47 far out of range, 48..63 just past the end
  uint32_t off = tid < 32 ? (31 - tid) * 16 : (tid < 48 ? 0x7ffffff0u : nbytes + (tid - 48) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  uint32_t *dg, *dout; (void)hipMalloc(&dg, 4096); (void)hipMalloc(&dout, 4096);
  (void)hipMemcpy(dg, h, 4096, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dg, dout, 512);   // buffer = 512 bytes = 32 chunks
  uint32_t o[1024]; (void)hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" %08x", o[256 + l * 4 + e]);
    if (l < 32) { for (int e = 0; e < 4; ++e) bad += o[256 + l*4 + e] != 1000u + (31 - l) * 4 + e; }
    else { for (int e = 0; e < 4; ++e) bad += o[256 + l*4 + e] != 0; }
    printf("\n");
  }
  printf("untouched before/after: %08x %08x ; mismatches: %d\n", o[255], o[512], bad);
  return 0;
}
