// Helpers shared by the spatially tiled weight-gradient kernel's index arithmetic (wgrad_halo_geom.h): free of device
// intrinsics, so that the SAME functions are compiled into the kernel and into the host emulator
// (tests/emu/wgrad_halo_emu.cpp, run by tests/test_halo_geometry.py on the CPU).
// (The geometry of the spatially tiled 3x3 FORWARD kernels that lived here through round 4 went with those kernels:
// profiles/r05_negative_results.txt #1.)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HALO_HD __host__ __device__ __forceinline__
#else
#define HALO_HD static inline
#endif

namespace halo {

constexpr uint32_t kNoSrc = 0x7ffffff0u;        // buffer offset that the bounds check turns into zeros

// division by a run-time constant as multiply-high + shifts (the setup code divides per lane and per DMA piece)
struct FDiv { uint32_t mul, sh1, sh2; };
static inline FDiv make_fdiv(uint32_t d) {
  FDiv f = {0, 0, 0};
  if (d > 1) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = 1;
    f.sh2 = l - 1;
  }
  return f;
}
HALO_HD int fdiv(int n, const FDiv f) {
  const uint32_t t = (uint32_t)(((uint64_t)f.mul * (uint64_t)(uint32_t)n) >> 32);
  return (int)((t + (((uint32_t)n - t) >> f.sh1)) >> f.sh2);
}

}  // namespace halo
