// Optional per-launch HIP-event timing of the MFMA kernels (used by bench.py's roofline leg).
#pragma once
#include <hip/hip_runtime.h>
void passl_prof_begin(int kernel_class, hipStream_t st);
void passl_prof_end(int kernel_class, hipStream_t st);
