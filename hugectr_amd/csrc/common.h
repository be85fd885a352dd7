// common.h -- shared host/device helpers for libhugectr_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/hugectr_amd.h"

namespace hctr {

void set_error(const std::string& msg);

constexpr int kWave = 64;  // CDNA wavefront
constexpr uint64_t kInvalidIndex = ~0ull;  // std::numeric_limits<size_t>::max()

#define HCTR_HIP(call)                                                                   \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      ::hctr::set_error(std::string(#call) + ": " + hipGetErrorString(e__) + " at " +    \
                        __FILE__ + ":" + std::to_string(__LINE__));                      \
      (void)hipGetLastError(); /* reported: a later launch check must not see it again */ \
      return HCTR_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

#define HCTR_LAUNCH_CHECK() HCTR_HIP(hipGetLastError())

#define HCTR_REQUIRE(cond, msg)                                          \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ::hctr::set_error(std::string("invalid argument: ") + (msg));      \
      return HCTR_ERR_INVALID_ARG;                                       \
    }                                                                    \
  } while (0)

#define HCTR_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != HCTR_OK) return rc__; \
  } while (0)

// dynamically sized LDS of a kernel (the launch's shared-memory bytes).  Spelled through a macro so
// that the host interpreter of tests/emu (test infrastructure; it cannot express an unsized extern
// array) can supply its own definition -- for hipcc this IS the plain declaration.
#ifndef HCTR_DYN_LDS
#define HCTR_DYN_LDS(T, name) extern __shared__ T name[]
#define HCTR_DYN_LDS16(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#endif

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no VGPR round
// trip); lane l's bytes land at lds_wave_base + 16 l, lds_wave_base must be wave-uniform.  A macro
// for the same reason as HCTR_DYN_LDS: the host interpreter of tests/emu supplies its own.
#ifndef HCTR_GLOBAL_LOAD_LDS16
#define HCTR_GLOBAL_LOAD_LDS16(gptr, lds_wave_base)                                       \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                   (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#endif

// counted wait on this wave's vector-memory queue + a barrier that does not drain it (a plain
// __syncthreads() waits for every outstanding LDS DMA: a ring of more than two staging buffers
// would never have a tile in flight across it).  Macros: the host interpreter has no queues.
#ifndef HCTR_WAIT_VMCNT
#define HCTR_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define HCTR_RAW_BARRIER()                                  \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
  } while (0)
#endif

template <typename T>
static inline T ceil_div(T a, T b) {
  return (a + b - 1) / b;
}

static inline hipStream_t as_stream(hctr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// grid cap for grid-stride memory-bound kernels: 256 CUs x 8 blocks
constexpr int kMaxGrid = 2048;

static inline int grid_for(size_t work_items, int block, int cap = kMaxGrid) {
  size_t g = ceil_div<size_t>(work_items, (size_t)block);
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

// key-type traits: the reference instantiates <unsigned int> and <long long>
template <typename K>
struct KeyTraits;
template <>
struct KeyTraits<uint32_t> {
  static constexpr int64_t empty = 0xFFFFFFFFll;
  static constexpr int bytes = 4;
};
template <>
struct KeyTraits<long long> {
  static constexpr int64_t empty = INT64_MAX;
  static constexpr int bytes = 8;
};

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
  return (x << r) | (x >> (32 - r));
}
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
// MurmurHash3_x86_32, seed 0, over the 4 or 8 key bytes
// (R/HugeCTR/include/hashtable/cudf/hash_functions.cuh:66-107)
__host__ __device__ __forceinline__ uint32_t murmur3_block(uint32_t h1, uint32_t k1) {
  k1 *= 0xcc9e2d51u;
  k1 = rotl32(k1, 15);
  k1 *= 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13);
  return h1 * 5u + 0xe6546b64u;
}
__host__ __device__ __forceinline__ uint32_t murmur3_key(uint32_t key) {
  uint32_t h1 = murmur3_block(0u, key);
  h1 ^= 4u;
  return fmix32(h1);
}
__host__ __device__ __forceinline__ uint32_t murmur3_key(long long key) {
  const uint64_t u = (uint64_t)key;
  uint32_t h1 = murmur3_block(0u, (uint32_t)(u & 0xFFFFFFFFull));
  h1 = murmur3_block(h1, (uint32_t)(u >> 32));
  h1 ^= 8u;
  return fmix32(h1);
}


// Optional per-kernel timing with hipEvents recorded on the launch stream (bench.py's roofline
// leg).  Off by default: no events are created or recorded.
struct Profiler {
  static constexpr int kCats = 4;  // 0 gather/pool, 1 hash/index, 2 sort, 3 segmented update
  static constexpr size_t kMaxPairs = 16384;
  bool enabled = false;
  std::vector<hipEvent_t> start[kCats], stop[kCats];
  size_t used[kCats] = {0, 0, 0, 0};

  void begin(int cat, hipStream_t s) {
    if (!enabled || used[cat] >= kMaxPairs) return;
    if (used[cat] >= start[cat].size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      start[cat].push_back(a);
      stop[cat].push_back(b);
    }
    (void)hipEventRecord(start[cat][used[cat]], s);
  }
  void end(int cat, hipStream_t s) {
    if (!enabled || used[cat] >= kMaxPairs || used[cat] >= start[cat].size()) return;
    (void)hipEventRecord(stop[cat][used[cat]], s);
    used[cat]++;
  }
  void reset() {
    for (int c = 0; c < kCats; c++) used[c] = 0;
  }
  int get(int cat, double* total_ms, uint64_t* launches) {
    double tot = 0.0;
    for (size_t i = 0; i < used[cat]; i++) {
      if (hipEventSynchronize(stop[cat][i]) != hipSuccess) return -1;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, start[cat][i], stop[cat][i]) != hipSuccess) return -1;
      tot += ms;
    }
    *total_ms = tot;
    *launches = used[cat];
    return 0;
  }
  void destroy() {
    for (int c = 0; c < kCats; c++) {
      for (auto e : start[c]) (void)hipEventDestroy(e);
      for (auto e : stop[c]) (void)hipEventDestroy(e);
      start[c].clear();
      stop[c].clear();
      used[c] = 0;
    }
  }
};

}  // namespace hctr
