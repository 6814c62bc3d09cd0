// Native step plan: the launch list of ONE training step, recorded once and replayed from one C call.
//
// Why not a HIP graph: measured in round 3 (profiles/r03_graph_vs_eager.txt) — hipGraphLaunch costs the host
// ~11 us per kernel node on this ROCm (15.6 ms for a 1400-launch MoCo step, no better than the eager Python
// launches), and a capture cannot hold the forked downsample branch whose backward autograd runs on another
// stream.  A plan is simpler: the library already makes every launch itself, so it writes down
// {kernel handle, grid, block, LDS bytes, stream, argument bytes} while a step executes normally
// (csrc/plan.h), plus the cross-stream edges the host code reports (event record / stream wait), and replays
// the list with hipLaunchKernel / hipEventRecord / hipStreamWaitEvent on the SAME streams.  Everything a
// launch refers to must stay where it was: the host side allocates the recorded step from a private memory
// pool and keeps step-varying scalars (learning rate, queue pointer, ...) in device memory
// (passl_amd/hip/replay.py).  Segments: the host closes a segment wherever something the library cannot replay
// has to happen between two launches (a collective of torch.distributed) and replays segment by segment.
//
// Also here: the four small kernels that replace the last ATen launches inside a step (zero fill, byte copy,
// bf16 -> fp32 cast, padded -> dense accumulate), so that a recorded step contains library launches only.
#include <mutex>
#include <vector>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace passl_rec {
std::atomic<int> g_recording{0};
}

namespace {

enum Kind : int { K_KERNEL = 0, K_RECORD = 1, K_WAIT = 2, K_MEMSET = 3 };

struct Entry {
  int kind;
  hipStream_t st;
  const void* fn;        // K_KERNEL
  dim3 grid, block;
  size_t shmem;
  size_t argv_index;     // first slot in Plan::argv
  int nargs;
  int event;             // K_RECORD / K_WAIT
  void* dst;             // K_MEMSET
  int value;
  size_t bytes;
};

struct ArgRef { size_t off; };

}  // namespace

struct passl_plan {
  std::vector<Entry> entries;
  std::vector<size_t> seg_end;          // entries.size() at every cut; segment s = [seg_end[s-1], seg_end[s])
  std::vector<char> blob;               // argument bytes while recording
  std::vector<ArgRef> arg_refs;         // offset of every argument, in launch order
  std::vector<void*> argv;              // resolved at record_end: pointers into `args`
  char* args = nullptr;                 // 64-byte aligned copy of blob
  std::vector<hipEvent_t> events;
  int n_events = 0;
  bool recording = false;
  bool finished = false;
  int64_t n_kernels = 0, n_records = 0, n_waits = 0, n_memsets = 0, replays = 0;
};

namespace {
std::mutex g_mu;
passl_plan* g_plan = nullptr;           // the plan that records (at most one per process)
}

namespace passl_rec {

void record_kernel(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t st, int nargs,
                   const void* const* argv, const size_t* sizes, const size_t* aligns) {
  std::lock_guard<std::mutex> lk(g_mu);
  passl_plan* p = g_plan;
  if (!p || !p->recording) return;
  Entry e{};
  e.kind = K_KERNEL;
  e.st = st;
  e.fn = fn;
  e.grid = grid;
  e.block = block;
  e.shmem = shmem;
  e.argv_index = p->arg_refs.size();
  e.nargs = nargs;
  for (int i = 0; i < nargs; ++i) {
    const size_t al = aligns[i] < 8 ? 8 : aligns[i];
    size_t off = (p->blob.size() + al - 1) / al * al;
    p->blob.resize(off + sizes[i]);
    memcpy(p->blob.data() + off, argv[i], sizes[i]);
    p->arg_refs.push_back(ArgRef{off});
  }
  p->entries.push_back(e);
  p->n_kernels += 1;
}

void record_memset(void* dst, int value, size_t bytes, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_mu);
  passl_plan* p = g_plan;
  if (!p || !p->recording) return;
  Entry e{};
  e.kind = K_MEMSET;
  e.st = st;
  e.dst = dst;
  e.value = value;
  e.bytes = bytes;
  p->entries.push_back(e);
  p->n_memsets += 1;
}

}  // namespace passl_rec

extern "C" int passl_hip_plan_create(passl_plan_t** out) {
  if (!out) return PASSL_EINVAL;
  *out = new passl_plan();
  return PASSL_OK;
}

extern "C" int passl_hip_plan_destroy(passl_plan_t* p) {
  if (!p) return PASSL_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_plan == p) {
      g_plan = nullptr;
      passl_rec::g_recording.store(0);
    }
  }
  for (hipEvent_t ev : p->events) (void)hipEventDestroy(ev);
  free(p->args);
  delete p;
  return PASSL_OK;
}

extern "C" int passl_hip_plan_record_begin(passl_plan_t* p) {
  if (!p) return PASSL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_plan != nullptr || p->finished || p->recording) return PASSL_EINVAL;   // one recording at a time; a plan records once
  p->recording = true;
  g_plan = p;
  passl_rec::g_recording.store(1);
  return PASSL_OK;
}

extern "C" int passl_hip_plan_cut(passl_plan_t* p) {
  if (!p) return PASSL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!p->recording) return PASSL_EINVAL;
  p->seg_end.push_back(p->entries.size());
  return (int)p->seg_end.size();          // index of the segment that starts here
}

extern "C" int passl_hip_plan_record_end(passl_plan_t* p) {
  if (!p) return PASSL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!p->recording || g_plan != p) return PASSL_EINVAL;
  passl_rec::g_recording.store(0);
  g_plan = nullptr;
  p->recording = false;
  p->seg_end.push_back(p->entries.size());
  // freeze the argument bytes and resolve every argument pointer
  const size_t nbytes = (p->blob.size() + 63) / 64 * 64 + 64;
  p->args = static_cast<char*>(aligned_alloc(64, nbytes));
  if (!p->args) return PASSL_EINVAL;
  if (!p->blob.empty()) memcpy(p->args, p->blob.data(), p->blob.size());
  p->argv.resize(p->arg_refs.size() + 1);
  for (size_t i = 0; i < p->arg_refs.size(); ++i) p->argv[i] = p->args + p->arg_refs[i].off;
  p->argv[p->arg_refs.size()] = nullptr;
  p->blob.clear();
  p->blob.shrink_to_fit();
  p->events.resize(p->n_events);
  for (int i = 0; i < p->n_events; ++i)
    if (hipEventCreateWithFlags(&p->events[i], hipEventDisableTiming) != hipSuccess) return PASSL_ELAUNCH;
  p->finished = true;
  return PASSL_OK;
}

// "everything enqueued on `stream` so far" as a plan-owned event; returns its id (>= 0).  While recording the
// real ordering of the executing step is the caller's business (its own events): nothing is enqueued here.
extern "C" int passl_hip_plan_event_record(passl_plan_t* p, passl_stream_t stream) {
  if (!p) return PASSL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!p->recording) return PASSL_EINVAL;
  Entry e{};
  e.kind = K_RECORD;
  e.st = as_stream(stream);
  e.event = p->n_events++;
  p->entries.push_back(e);
  p->n_records += 1;
  return e.event;
}

extern "C" int passl_hip_plan_stream_wait(passl_plan_t* p, passl_stream_t stream, int event_id) {
  if (!p) return PASSL_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!p->recording || event_id < 0 || event_id >= p->n_events) return PASSL_EINVAL;
  Entry e{};
  e.kind = K_WAIT;
  e.st = as_stream(stream);
  e.event = event_id;
  p->entries.push_back(e);
  p->n_waits += 1;
  return PASSL_OK;
}

extern "C" int passl_hip_plan_replay(passl_plan_t* p, int segment) {
  if (!p || !p->finished || segment < 0 || segment >= (int)p->seg_end.size()) return PASSL_EINVAL;
  const size_t b = segment == 0 ? 0 : p->seg_end[segment - 1];
  const size_t e = p->seg_end[segment];
  const Entry* en = p->entries.data();
  void** argv = p->argv.data();
  for (size_t i = b; i < e; ++i) {
    const Entry& x = en[i];
    hipError_t rc;
    switch (x.kind) {
      case K_KERNEL:
        rc = hipLaunchKernel(x.fn, x.grid, x.block, argv + x.argv_index, x.shmem, x.st);
        break;
      case K_RECORD:
        rc = hipEventRecord(p->events[x.event], x.st);
        break;
      case K_WAIT:
        rc = hipStreamWaitEvent(x.st, p->events[x.event], 0);
        break;
      default:
        rc = hipMemsetAsync(x.dst, x.value, x.bytes, x.st);
        break;
    }
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      return PASSL_ELAUNCH;
    }
  }
  if (segment == 0) p->replays += 1;
  return PASSL_OK;
}

// what: 0 segments, 1 kernel launches, 2 event records, 3 stream waits, 4 memsets, 5 argument bytes, 6 replays,
// 7 distinct streams
extern "C" int64_t passl_hip_plan_info(passl_plan_t* p, int what) {
  if (!p) return PASSL_EINVAL;
  switch (what) {
    case 0: return (int64_t)p->seg_end.size();
    case 1: return p->n_kernels;
    case 2: return p->n_records;
    case 3: return p->n_waits;
    case 4: return p->n_memsets;
    case 5: return p->arg_refs.empty() ? 0 : (int64_t)(p->arg_refs.back().off + 8);
    case 6: return p->replays;
    case 7: {
      std::vector<hipStream_t> seen;
      for (const Entry& x : p->entries) {
        bool hit = false;
        for (hipStream_t s : seen) hit = hit || s == x.st;
        if (!hit) seen.push_back(x.st);
      }
      return (int64_t)seen.size();
    }
    default: return PASSL_EINVAL;
  }
}

// ------------------------------------------------------------------------------------------------------------
// the step's last ATen launches as library kernels (a recorded step must not contain foreign launches)
namespace {
constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads) fill_zero_kernel(uint4* __restrict__ p, int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * kThreads * 4;
  for (int64_t i = (int64_t)blockIdx.x * kThreads * 4 + threadIdx.x; i < n16; i += stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + (int64_t)u * kThreads;
      if (j < n16) p[j] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

__global__ void __launch_bounds__(kThreads) copy16_kernel(const uint4* __restrict__ s, uint4* __restrict__ d,
                                                          int64_t n16) {
  const int64_t stride = (int64_t)gridDim.x * kThreads * 4;
  for (int64_t i = (int64_t)blockIdx.x * kThreads * 4 + threadIdx.x; i < n16; i += stride) {
    uint4 v[4];
    // all loads of a trip back to back (a load behind `if` waits for the previous one: DESIGN.md 3.4); lanes past
    // the end re-read the last chunk
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t j = i + (int64_t)u * kThreads;
      j = j < n16 ? j : n16 - 1;
      v[u] = s[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t j = i + (int64_t)u * kThreads;
      if (j < n16) d[j] = v[u];
    }
  }
}

__global__ void __launch_bounds__(kThreads) small_bytes_kernel(const unsigned char* __restrict__ s,
                                                               unsigned char* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) d[i] = s ? s[i] : (unsigned char)0;
}

__global__ void __launch_bounds__(kThreads) cast_bf16_f32_kernel(const bf16_t* __restrict__ s, float* __restrict__ d,
                                                                 int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n8; i += stride) {
    float v[8];
    ElemTraits<bf16_t>::load8(s + i * 8, v);
    ElemTraits<float>::store8(d + i * 8, v);
  }
}

__global__ void __launch_bounds__(kThreads) cast_bf16_f32_tail_kernel(const bf16_t* __restrict__ s,
                                                                      float* __restrict__ d, int64_t from, int64_t n) {
  const int64_t i = from + (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) d[i] = bf2f(s[i]);
}

// dst[row][r][s][c] += src[row][r][s][c] over the dense extents (dst_S x dst_C) of a padded (src_S x src_C) block
__global__ void __launch_bounds__(kThreads) unpad_add_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int64_t n, int R, int dS, int dC, int sS, int sC) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % dC);
  int64_t t = i / dC;
  const int s = (int)(t % dS);
  t /= dS;
  const int r = (int)(t % R);
  const int64_t row = t / R;
  dst[i] += src[((row * R + r) * sS + s) * sC + c];
}

inline unsigned grid_for16(int64_t n16) {
  int64_t g = (n16 + kThreads * 4 - 1) / (kThreads * 4);
  return (unsigned)(g < 1 ? 1 : (g > 256 * 8 ? 256 * 8 : g));
}
}  // namespace

extern "C" int passl_hip_fill_zero(void* p, int64_t bytes, passl_stream_t stream) {
  if (bytes < 0 || !p) return PASSL_EINVAL;
  if (bytes == 0) return PASSL_OK;
  hipStream_t st = as_stream(stream);
  char* c = static_cast<char*>(p);
  const int64_t head = aligned16(c) ? 0 : 16 - (int64_t)(reinterpret_cast<uintptr_t>(c) & 15u);
  const int64_t h = head < bytes ? head : bytes;
  if (h > 0)
    hipLaunchKernelGGL(small_bytes_kernel, dim3(1), dim3(kThreads), 0, st, (const unsigned char*)nullptr,
                       reinterpret_cast<unsigned char*>(c), h);
  const int64_t n16 = (bytes - h) / 16;
  if (n16 > 0)
    hipLaunchKernelGGL(fill_zero_kernel, dim3(grid_for16(n16)), dim3(kThreads), 0, st,
                       reinterpret_cast<uint4*>(c + h), n16);
  const int64_t tail = bytes - h - n16 * 16;
  if (tail > 0)
    hipLaunchKernelGGL(small_bytes_kernel, dim3(1), dim3(kThreads), 0, st, (const unsigned char*)nullptr,
                       reinterpret_cast<unsigned char*>(c + h + n16 * 16), tail);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_copy_bytes(void* dst, const void* src, int64_t bytes, passl_stream_t stream) {
  if (bytes < 0 || !dst || !src) return PASSL_EINVAL;
  if (bytes == 0) return PASSL_OK;
  hipStream_t st = as_stream(stream);
  if (aligned16(dst) && aligned16(src)) {
    const int64_t n16 = bytes / 16;
    if (n16 > 0)
      hipLaunchKernelGGL(copy16_kernel, dim3(grid_for16(n16)), dim3(kThreads), 0, st,
                         static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    const int64_t tail = bytes - n16 * 16;
    if (tail > 0)
      hipLaunchKernelGGL(small_bytes_kernel, dim3(1), dim3(kThreads), 0, st,
                         static_cast<const unsigned char*>(src) + n16 * 16, static_cast<unsigned char*>(dst) + n16 * 16,
                         tail);
  } else {
    const int64_t g = (bytes + kThreads - 1) / kThreads;
    if (g > 0x7fffffffLL) return PASSL_EINVAL;
    hipLaunchKernelGGL(small_bytes_kernel, dim3((unsigned)g), dim3(kThreads), 0, st,
                       static_cast<const unsigned char*>(src), static_cast<unsigned char*>(dst), bytes);
  }
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_cast_bf16_to_f32(const void* src, float* dst, int64_t n, passl_stream_t stream) {
  if (n < 0 || !src || !dst) return PASSL_EINVAL;
  if (n == 0) return PASSL_OK;
  if (!aligned16(src) || !aligned16(dst)) return PASSL_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t n8 = n / 8;
  if (n8 > 0) {
    int64_t g = (n8 + kThreads - 1) / kThreads;
    g = g > 256 * 16 ? 256 * 16 : g;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)g), dim3(kThreads), 0, st,
                       static_cast<const bf16_t*>(src), dst, n8);
  }
  if (n8 * 8 < n)
    hipLaunchKernelGGL(cast_bf16_f32_tail_kernel, dim3(1), dim3(kThreads), 0, st, static_cast<const bf16_t*>(src), dst,
                       n8 * 8, n);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}

extern "C" int passl_hip_unpad_add(const float* src, float* dst, int64_t rows, int R, int dst_S, int dst_C,
                                   int src_S, int src_C, passl_stream_t stream) {
  if (!src || !dst || rows <= 0 || R <= 0 || dst_S <= 0 || dst_C <= 0 || src_S < dst_S || src_C < dst_C)
    return PASSL_EINVAL;
  const int64_t n = rows * R * dst_S * dst_C;
  const int64_t g = (n + kThreads - 1) / kThreads;
  if (g > 0x7fffffffLL) return PASSL_EINVAL;
  hipLaunchKernelGGL(unpad_add_kernel, dim3((unsigned)g), dim3(kThreads), 0, as_stream(stream), src, dst, n, R, dst_S,
                     dst_C, src_S, src_C);
  PASSL_RETURN_IF_LAUNCH_FAILED();
  return PASSL_OK;
}
