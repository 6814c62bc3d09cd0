// Optional per-launch HIP-event timing of the MFMA kernels (used by bench.py's roofline leg).
// Kernel classes: 0 = igemm_ring_kernel (LDS-DMA ring implicit GEMM), 1 = weight-gradient kernels,
// 2 = igemm_kernel (register-staged implicit GEMM: short reductions, stem, fp32), 3 = igemm_8p_kernel
// (256 x 256 tiles, 8-phase LDS-DMA schedule).
#pragma once
#include <hip/hip_runtime.h>
constexpr int kProfClasses = 4;
void passl_prof_begin(int kernel_class, hipStream_t st);
void passl_prof_end(int kernel_class, hipStream_t st);
// the launch opened as class `from` turned out to run a kernel of class `to`
void passl_prof_retag(int from, int to);
// algorithmic work of the launch just timed (summed per class while profiling is on)
void passl_prof_work(int kernel_class, double flops, double bytes);
