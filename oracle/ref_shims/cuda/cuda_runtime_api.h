// <cuda_runtime_api.h> stand-in -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libref_cache.so).
// The reference's embedding cache (R/gpu_cache/src/nv_gpu_cache.cu) is CUDA source; to pin
// oracle/cache_oracle.py against the reference's OWN code it is compiled as plain C++ from where it
// lies and stepped through by the host interpreter of tests/emu (hipemu.h: fibers = threads,
// 32-lane tiles = wavefronts of width 32).  This header restates the part of the CUDA language and
// runtime that file uses: declarations and trivial host-memory forwards, no reference code.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../../tests/emu/hipemu.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define threadIdx (hipemu::ids()->tid)
#define blockIdx (hipemu::ids()->bid)
#define blockDim (hipemu::ids()->bdim)
#define gridDim (hipemu::ids()->gdim)

typedef struct refemu_stream* cudaStream_t;
typedef struct refemu_event* cudaEvent_t;
enum cudaError_t { cudaSuccess = 0, cudaErrorMemoryAllocation = 2 };
enum cudaMemoryType {
  cudaMemoryTypeUnregistered = 0,
  cudaMemoryTypeHost = 1,
  cudaMemoryTypeDevice = 2,
  cudaMemoryTypeManaged = 3
};
struct cudaPointerAttributes {
  cudaMemoryType type;
  int device;
};
#define CUDART_VERSION 12000

static inline const char* cudaGetErrorString(cudaError_t e) {
  return e == cudaSuccess ? "no error" : "error (host interpreter)";
}
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaGetDevice(int* d) {
  *d = 0;
  return cudaSuccess;
}
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
static inline cudaError_t cudaPointerGetAttributes(cudaPointerAttributes* a, const void*) {
  a->type = cudaMemoryTypeDevice;
  a->device = 0;
  return cudaSuccess;
}
static inline cudaError_t cudaMalloc(void** p, size_t bytes) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? (bytes + 255) / 256 * 256 : 256) != 0)
    return cudaErrorMemoryAllocation;
  memset(q, 0xA5, bytes);  // fresh device memory holds garbage
  *p = q;
  return cudaSuccess;
}
template <typename T>
static inline cudaError_t cudaMallocManaged(T** p, size_t bytes) {
  void* q = nullptr;
  const cudaError_t rc = cudaMalloc(&q, bytes);
  *p = (T*)q;
  return rc;
}
static inline cudaError_t cudaMemPrefetchAsync(const void*, size_t, int, cudaStream_t = nullptr) {
  return cudaSuccess;
}
static inline cudaError_t cudaFree(void* p) {
  free(p);
  return cudaSuccess;
}
static inline cudaError_t cudaMemsetAsync(void* p, int v, size_t n, cudaStream_t = nullptr) {
  memset(p, v, n);
  return cudaSuccess;
}
static inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
enum cudaMemcpyKind {
  cudaMemcpyHostToHost,
  cudaMemcpyHostToDevice,
  cudaMemcpyDeviceToHost,
  cudaMemcpyDeviceToDevice,
  cudaMemcpyDefault
};
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) {
  memmove(d, s, n);
  return cudaSuccess;
}
static inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind,
                                          cudaStream_t = nullptr) {
  memmove(d, s, n);
  return cudaSuccess;
}

// ---- the CUDA builtins the file uses (its own overloads for long / long long / unsigned long
// forward to the unsigned long long one, as on the device) ------------------------------------------
static inline void __syncthreads() { hipemu::syncthreads(); }
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) {
  return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline float atomicAdd(float* p, float v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
  unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    unsigned nu;
    memcpy(&nu, &f, 4);
    if (__atomic_compare_exchange_n(u, &old, nu, true, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) {
      memcpy(&f, &old, 4);
      return f;
    }
  }
}
struct alignas(8) uint2 {
  unsigned x, y;
};
struct alignas(16) uint4 {
  unsigned x, y, z, w;
};
static inline double __longlong_as_double(long long v) {
  double d;
  memcpy(&d, &v, 8);
  return d;
}
static inline long long __double_as_longlong(double v) {
  long long d;
  memcpy(&d, &v, 8);
  return d;
}
static inline float __int_as_float(int v) {
  float d;
  memcpy(&d, &v, 4);
  return d;
}
static inline int __float_as_int(float v) {
  int d;
  memcpy(&d, &v, 4);
  return d;
}
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp,
                                           unsigned long long v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
// a CAS that did not swap is the body of a spin loop (warp_lock_mutex): threads are fibers here,
// so the spinning one must let the lock's holder run
static inline int atomicCAS(int* p, int cmp, int v) {
  const int want = cmp;
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  if (cmp != want) hipemu::spin_pause();
  return cmp;
}

// kernel<<<grid, block, shmem, stream>>>(args) is rewritten by oracle/ref_launch_rewrite.py (the
// only edit made to the reference's text, into oracle/_ref/gen/) to this macro
namespace refemu {
static inline dim3 to_dim3(dim3 d) { return d; }
template <typename T>
static inline dim3 to_dim3(T v) {
  return dim3((unsigned)v);
}
struct Cfg {
  dim3 g, b;
  size_t shmem;
  template <typename G, typename B>
  Cfg(G g_, B b_, size_t s_ = 0, cudaStream_t = nullptr)
      : g(to_dim3(g_)), b(to_dim3(b_)), shmem(s_) {}
};
}  // namespace refemu
#define REFEMU_LAUNCH(kernel, cfg, ...)                       \
  do {                                                         \
    ::refemu::Cfg refemu_cfg cfg;                              \
    hipemu::launch(refemu_cfg.g, refemu_cfg.b, refemu_cfg.shmem, \
                   [&]() { kernel(__VA_ARGS__); });            \
  } while (0)
