// Launch interposer of the native step plan (csrc/plan.hip, include/passl_hip.h "step plans").
//
// Every kernel of this library is launched through hipLaunchKernelGGL.  The macro is re-defined here so
// that, WHILE A PLAN IS RECORDING, a launch is also appended to the plan — the kernel's host handle, its
// launch geometry, its stream and a byte copy of its arguments converted to the kernel's own parameter
// types — before it executes exactly as it always did.  Outside a recording the cost is one relaxed load
// of an int.  Replay (passl_hip_plan_replay) walks the list with hipLaunchKernel: no Python, no ctypes, no
// descriptor filling, no dispatch logic — the launch decisions the library took when the step was recorded
// are part of the plan.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstddef>
#include <tuple>
#include <utility>

namespace passl_rec {

extern std::atomic<int> g_recording;      // != 0 while some plan records (launches of ANY thread are taken)

void record_kernel(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t st, int nargs,
                   const void* const* argv, const size_t* sizes, const size_t* aligns);
void record_memset(void* dst, int value, size_t bytes, hipStream_t st);

template <typename... P, typename... A, size_t... I>
inline void capture_impl(void (*k)(P...), dim3 g, dim3 b, size_t shmem, hipStream_t st,
                         std::index_sequence<I...>, A&&... a) {
  // the values as the kernel receives them (implicit conversions of the call applied)
  std::tuple<P...> vals(static_cast<P>(a)...);
  const void* argv[sizeof...(P) + 1] = {static_cast<const void*>(&std::get<I>(vals))..., nullptr};
  const size_t sizes[sizeof...(P) + 1] = {sizeof(P)..., 0};
  const size_t aligns[sizeof...(P) + 1] = {alignof(P)..., 0};
  record_kernel(reinterpret_cast<const void*>(k), g, b, shmem, st, (int)sizeof...(P), argv, sizes, aligns);
}

template <typename... P, typename... A>
inline void capture(void (*k)(P...), dim3 g, dim3 b, size_t shmem, hipStream_t st, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel launched with a wrong number of arguments");
  capture_impl(k, g, b, shmem, st, std::index_sequence_for<P...>{}, std::forward<A>(a)...);
}

inline bool recording() { return g_recording.load(std::memory_order_relaxed) != 0; }

// hipMemsetAsync that a recording plan sees
inline hipError_t memset_async(void* dst, int value, size_t bytes, hipStream_t st) {
  if (__builtin_expect(recording(), 0)) record_memset(dst, value, bytes, st);
  return hipMemsetAsync(dst, value, bytes, st);
}

}  // namespace passl_rec

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                         \
  do {                                                                                                      \
    if (__builtin_expect(passl_rec::recording(), 0))                                                       \
      passl_rec::capture(kernel, dim3(grid), dim3(block), (size_t)(shmem), (stream), ##__VA_ARGS__);       \
    kernel<<<dim3(grid), dim3(block), (shmem), (stream)>>>(__VA_ARGS__);                                    \
  } while (0)
