// Shared device/host helpers for libpassl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/passl_hip.h"
#include "plan.h"     // re-defines hipLaunchKernelGGL: launches are visible to a recording step plan

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef unsigned short bf16_t;   // raw bf16 bits

#define PASSL_RETURN_IF_LAUNCH_FAILED()                 \
  do {                                                  \
    if (hipGetLastError() != hipSuccess) return PASSL_ELAUNCH; \
  } while (0)

static inline hipStream_t as_stream(passl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; NaN stays NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f));
}
// two floats -> packed bf16 pair, RNE, one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  const bf2_t v = __builtin_convertvector(f2_t{lo, hi}, bf2_t);
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int VEC = 4;           // elements per 16 bytes
  static constexpr int DT = PASSL_F32;
  __device__ static __forceinline__ void load8(const float* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const uint4 b = *reinterpret_cast<const uint4*>(p + 4);
    v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y);
    v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
    v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y);
    v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
  }
  __device__ static __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]),
                                              __float_as_uint(v[2]), __float_as_uint(v[3]));
    *reinterpret_cast<uint4*>(p + 4) = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]),
                                                  __float_as_uint(v[6]), __float_as_uint(v[7]));
  }
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int DT = PASSL_BF16;
  __device__ static __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
  }
  __device__ static __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                              pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
