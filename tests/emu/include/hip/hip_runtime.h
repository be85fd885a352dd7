// <hip/hip_runtime.h> of the host interpreter (tests/emu/hipemu.h) -- TEST INFRASTRUCTURE ONLY.
// The subset of the HIP language and runtime API that hugectr_amd/csrc/*.hip uses, mapped onto
// fibers (threads of a workgroup), OS threads (workgroups) and host memory ("device" pointers are
// host pointers; every call is synchronous, streams and events are ignored).
#pragma once
#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../../hipemu.h"

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(n) alignas(n)
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5

#define threadIdx (hipemu::ids()->tid)
#define blockIdx (hipemu::ids()->bid)
#define blockDim (hipemu::ids()->bdim)
#define gridDim (hipemu::ids()->gdim)

// ---- vector types ---------------------------------------------------------------------------------
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
struct alignas(16) longlong2 { long long x, y; };
struct alignas(4) ushort2 { unsigned short x, y; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) {
  return uint4{x, y, z, w};
}
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) {
  return ulonglong2{x, y};
}

// ---- runtime API ------------------------------------------------------------------------------------
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind {
  hipMemcpyHostToHost,
  hipMemcpyHostToDevice,
  hipMemcpyDeviceToHost,
  hipMemcpyDeviceToDevice,
  hipMemcpyDefault
};
constexpr unsigned hipHostMallocMapped = 2, hipHostMallocPortable = 1;
constexpr unsigned hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventDefault = 0;

static inline const char* hipGetErrorString(hipError_t e) {
  return e == hipSuccess ? "no error"
                         : "error (host interpreter; after a launch: invalid configuration, e.g. a "
                           "zero-sized grid)";
}
static inline hipError_t hipGetLastError() {
  return hipemu::take_launch_error() ? hipErrorInvalidValue : hipSuccess;
}
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
template <typename T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? (bytes + 255) / 256 * 256 : 256) != 0)
    return hipErrorOutOfMemory;
  memset(q, 0xA5, bytes);  // fresh device memory holds garbage
  *p = (T*)q;
  return hipSuccess;
}
template <typename T>
static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) {
  return hipMalloc(p, bytes);
}
template <typename T>
static inline hipError_t hipHostGetDevicePointer(T** d, void* h, unsigned) {
  *d = (T*)h;
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipHostFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* p, int v, size_t n) {
  memset(p, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) {
  memset(p, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind,
                                        hipStream_t = nullptr) {
  memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) {
  *s = nullptr;
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
  *s = nullptr;
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
constexpr unsigned hipStreamNonBlocking = 1, hipStreamDefault = 0;
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) {
  *lo = 0;
  *hi = -1;
  return hipSuccess;
}
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) {
  *s = nullptr;
  return hipSuccess;
}
// (the interpreter runs every workgroup of a small grid on its own OS thread: "all resident")
template <typename F>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) {
  *n = 2;
  return hipSuccess;
}
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F>
static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) {
  return hipSuccess;
}
enum hipDeviceAttribute_t {
  hipDeviceAttributeMultiprocessorCount = 16,
  hipDeviceAttributeVirtualMemoryManagementSupported = 10000
};
static inline hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeVirtualMemoryManagementSupported ? 1 : 256;
  return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) {
  return hipSuccess;
}

// ---- virtual memory management: a reserved range is an inaccessible anonymous mapping, a mapped
//      piece is the same addresses made readable / writable (the host pages in what is touched) ----
typedef void* hipDeviceptr_t;
typedef size_t hipMemGenericAllocationHandle_t;  // (the piece's size)
enum hipMemAllocationType { hipMemAllocationTypePinned = 1 };
enum hipMemLocationType { hipMemLocationTypeDevice = 1 };
enum hipMemAccessFlags { hipMemAccessFlagsProtReadWrite = 3 };
struct hipMemLocation {
  hipMemLocationType type;
  int id;
};
struct hipMemAllocationProp {
  hipMemAllocationType type;
  hipMemLocation location;
};
struct hipMemAccessDesc {
  hipMemLocation location;
  hipMemAccessFlags flags;
};
static inline hipError_t hipMemAddressReserve(hipDeviceptr_t* p, size_t bytes, size_t, void*,
                                              unsigned long long) {
  void* q = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (q == MAP_FAILED) return hipErrorOutOfMemory;
  *p = q;
  return hipSuccess;
}
static inline hipError_t hipMemAddressFree(hipDeviceptr_t p, size_t bytes) {
  return munmap(p, bytes) == 0 ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t bytes,
                                      const hipMemAllocationProp*, unsigned long long) {
  *h = bytes;
  return hipSuccess;
}
static inline hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipSuccess; }
static inline hipError_t hipMemMap(void* p, size_t bytes, size_t, hipMemGenericAllocationHandle_t h,
                                   unsigned long long) {
  return h == bytes && (reinterpret_cast<uintptr_t>(p) & 4095) == 0 ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipMemSetAccess(void* p, size_t bytes, const hipMemAccessDesc*, size_t) {
  if (mprotect(p, bytes, PROT_READ | PROT_WRITE) != 0) return hipErrorInvalidValue;
  memset(p, 0xA5, bytes < (1u << 20) ? bytes : (1u << 20));  // fresh device memory holds garbage
  return hipSuccess;
}
static inline hipError_t hipMemUnmap(void* p, size_t bytes) {
  (void)madvise(p, bytes, MADV_DONTNEED);
  return mprotect(p, bytes, PROT_NONE) == 0 ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = nullptr;
  return hipSuccess;
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  *e = nullptr;
  return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 0.f;
  return hipSuccess;
}

// (kernel names that contain a comma arrive in parentheses, as HIP's own macro needs them)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), (size_t)(shmem), [&]() { (kernel)(__VA_ARGS__); })

// ---- workgroup / wavefront --------------------------------------------------------------------------
static inline void __syncthreads() { hipemu::syncthreads(); }

namespace hipemu {
template <typename T>
static inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffles move at most 64 bits");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T>
static inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
}  // namespace hipemu

// every collective call site gets its own number, in source order (__COUNTER__): lanes that wait
// at different sites are released lowest site first
#define __ballot(p) hipemu::collective(hipemu::OP_BALLOT, (p) ? 1 : 0, 0, 64, __COUNTER__)
#define __all(p) ((int)hipemu::collective(hipemu::OP_ALL, (p) ? 1 : 0, 0, 64, __COUNTER__))
#define __any(p) ((int)hipemu::collective(hipemu::OP_ANY, (p) ? 1 : 0, 0, 64, __COUNTER__))
#define __builtin_amdgcn_wave_barrier() \
  ((void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 64, __COUNTER__))
#define __builtin_amdgcn_readfirstlane(v) \
  ((int)hipemu::collective(hipemu::OP_FIRST, (uint64_t)(uint32_t)(v), 0, 64, __COUNTER__))
#define HIPEMU_SHFL_(op, v, a, w) \
  hipemu::shfl_((op), (v), (int)(a), (int)(w), __COUNTER__)
#define HIPEMU_SHFL_PICK_(_1, _2, _3, name, ...) name
#define HIPEMU_SHFL3_(op, v, a, w) HIPEMU_SHFL_(op, v, a, w)
#define HIPEMU_SHFL2_(op, v, a) HIPEMU_SHFL_(op, v, a, 64)
#define __shfl(...) \
  HIPEMU_SHFL_PICK_(__VA_ARGS__, HIPEMU_SHFL3_, HIPEMU_SHFL2_)(hipemu::OP_SHFL, __VA_ARGS__)
#define __shfl_up(...) \
  HIPEMU_SHFL_PICK_(__VA_ARGS__, HIPEMU_SHFL3_, HIPEMU_SHFL2_)(hipemu::OP_SHFL_UP, __VA_ARGS__)
#define __shfl_down(...) \
  HIPEMU_SHFL_PICK_(__VA_ARGS__, HIPEMU_SHFL3_, HIPEMU_SHFL2_)(hipemu::OP_SHFL_DOWN, __VA_ARGS__)
#define __shfl_xor(...) \
  HIPEMU_SHFL_PICK_(__VA_ARGS__, HIPEMU_SHFL3_, HIPEMU_SHFL2_)(hipemu::OP_SHFL_XOR, __VA_ARGS__)

namespace hipemu {
template <typename T>
static inline T shfl_(Op op, T v, int arg, int width, int site) {
  return from_bits<T>(collective(op, to_bits(v), arg, width, site));
}
}  // namespace hipemu

#define __builtin_amdgcn_s_sleep(n) hipemu::spin_pause()
#define __builtin_amdgcn_s_waitcnt(n) ((void)0)
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_block() { std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline void __threadfence_system() { std::atomic_thread_fence(std::memory_order_seq_cst); }

// ---- bit / math intrinsics ----------------------------------------------------------------------
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static inline int __clzll(long long v) {
  return v == 0 ? 64 : __builtin_clzll((unsigned long long)v);
}
static inline float __uint_as_float(unsigned v) { return hipemu::from_bits<float>(v); }
static inline unsigned __float_as_uint(float v) { return (unsigned)hipemu::to_bits(v); }
static inline float __int_as_float(int v) { return hipemu::from_bits<float>((unsigned)v); }
static inline int __float_as_int(float v) { return (int)hipemu::to_bits(v); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
#define __expf(a) expf(a)
#define __logf(a) logf(a)
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
using std::signbit;

// ---- atomics (workgroups run on OS threads: these must be real) --------------------------------------
template <typename T>
static inline T hipemu_fetch_min(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED,
                                                 __ATOMIC_RELAXED)) {
  }
  return old;
}
template <typename T>
static inline T hipemu_fetch_max(T* p, T v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED,
                                                 __ATOMIC_RELAXED)) {
  }
  return old;
}
#define HIPEMU_INT_ATOMICS_(T)                                                                  \
  static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }   \
  static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }   \
  static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }     \
  static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }   \
  static inline T atomicXor(T* p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }   \
  static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); } \
  static inline T atomicMin(T* p, T v) { return hipemu_fetch_min(p, v); }                       \
  static inline T atomicMax(T* p, T v) { return hipemu_fetch_max(p, v); }                       \
  static inline T atomicCAS(T* p, T cmp, T v) {                                                 \
    __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);         \
    return cmp;                                                                                 \
  }
HIPEMU_INT_ATOMICS_(int)
HIPEMU_INT_ATOMICS_(unsigned)
HIPEMU_INT_ATOMICS_(unsigned long long)
HIPEMU_INT_ATOMICS_(unsigned long)
HIPEMU_INT_ATOMICS_(long long)
#undef HIPEMU_INT_ATOMICS_
static inline float atomicAdd(float* p, float v) {
  unsigned* u = reinterpret_cast<unsigned*>(p);
  unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED);
  for (;;) {
    const float nf = hipemu::from_bits<float>(old) + v;
    if (__atomic_compare_exchange_n(u, &old, (unsigned)hipemu::to_bits(nf), true, __ATOMIC_RELAXED,
                                    __ATOMIC_RELAXED))
      return hipemu::from_bits<float>(old);
  }
}
static inline void atomicAddNoRet(float* p, float v) { (void)atomicAdd(p, v); }
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }

template <typename T>
static inline T hipemu_atomic_load(const T* p) {
  T v;
  __atomic_load(p, &v, __ATOMIC_SEQ_CST);
  return v;
}
template <typename T, typename V>
static inline void hipemu_atomic_store(T* p, V v) {
  T t = (T)v;
  __atomic_store(p, &t, __ATOMIC_SEQ_CST);
}
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load((p))
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store((p), (v))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_and(p, v, order, scope) __atomic_fetch_and((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_exchange(p, v, order, scope) __atomic_exchange_n((p), (v), __ATOMIC_SEQ_CST)
#define __hip_atomic_compare_exchange_strong(p, expected, desired, so, fo, scope) \
  __atomic_compare_exchange_n((p), (expected), (desired), false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_min(p, v, order, scope) hipemu_fetch_min((p), (v))
#define __hip_atomic_fetch_max(p, v, order, scope) hipemu_fetch_max((p), (v))

// ---- the rest of the device library used by the kernels ---------------------------------------------
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// global_load_lds_dwordx4 (common.h: HCTR_GLOBAL_LOAD_LDS16): lane l's 16 bytes land at base + 16 l
// (synchronously here; on the device the bytes are only there after the wave's vmcnt has drained
// and a barrier was passed -- the interpreter cannot see a missing one)
#define HCTR_GLOBAL_LOAD_LDS16(gptr, lds_wave_base)                                             \
  memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * (threadIdx.x & 63), (gptr), 16)

#define HCTR_WAIT_VMCNT(n) ((void)0)
#define HCTR_RAW_BARRIER() __syncthreads()

// dynamically sized LDS (hugectr_amd/csrc/common.h spells it through these macros)
#define HCTR_DYN_LDS(T, name) T* name = reinterpret_cast<T*>(hipemu::dyn_shared())
#define HCTR_DYN_LDS16(T, name) T* name = reinterpret_cast<T*>(hipemu::dyn_shared())

// ---- MFMA: D = A x B + C on one wavefront, operands spread over the lanes as the CDNA3/4 ISA
// lays them out.  fp32 products summed in k order (the matrix core's own order of accumulation is
// not specified: compare with a tolerance).
namespace hipemu {
struct MfmaScratch {
  float a[64][8], b[64][8];
};
static inline int linear_tid() {
  const Ids* i = ids();
  return (int)(i->tid.x + i->bdim.x * (i->tid.y + i->bdim.y * i->tid.z));
}
static inline MfmaScratch* mfma_scratch() {
  static thread_local MfmaScratch s[16];  // one per wavefront of the workgroup
  return &s[linear_tid() >> 6];
}
// 32x32x16, 16-bit inputs: A[i][k]: lane i + 32 * (k / 8), element k % 8; B[k][j]: lane j + 32 *
// (k / 8), element k % 8; D[i][j]: lane j + 32 * ((i / 4) % 2), register 4 * (i / 8) + i % 4
template <typename V8, typename Acc16>
static inline Acc16 mfma_32x32x16(V8 a, V8 b, Acc16 c, int s1, int s2) {
  MfmaScratch* sc = mfma_scratch();
  const int lane = linear_tid() & 63;
  for (int k = 0; k < 8; k++) {
    sc->a[lane][k] = (float)a[k];
    sc->b[lane][k] = (float)b[k];
  }
  (void)collective(OP_WAVE_BARRIER, 0, 0, 64, s1);
  const int j = lane & 31, hb = lane >> 5;
  for (int r = 0; r < 16; r++) {
    const int i = (r / 4) * 8 + hb * 4 + (r % 4);
    float acc = c[r];
    for (int k = 0; k < 16; k++) acc += sc->a[i + 32 * (k / 8)][k % 8] * sc->b[j + 32 * (k / 8)][k % 8];
    c[r] = acc;
  }
  (void)collective(OP_WAVE_BARRIER, 0, 0, 64, s2);
  return c;
}
// 16x16x4 fp32: A[i][k]: lane i + 16 * k; B[k][j]: lane j + 16 * k; D[i][j]: lane j + 16 * (i / 4),
// register i % 4
template <typename Acc4>
static inline Acc4 mfma_16x16x4(float a, float b, Acc4 c, int s1, int s2) {
  MfmaScratch* sc = mfma_scratch();
  const int lane = linear_tid() & 63;
  sc->a[lane][0] = a;
  sc->b[lane][0] = b;
  (void)collective(OP_WAVE_BARRIER, 0, 0, 64, s1);
  const int j = lane & 15, q = lane >> 4;
  for (int r = 0; r < 4; r++) {
    const int i = 4 * q + r;
    float acc = c[r];
    for (int k = 0; k < 4; k++) acc += sc->a[i + 16 * k][0] * sc->b[j + 16 * k][0];
    c[r] = acc;
  }
  (void)collective(OP_WAVE_BARRIER, 0, 0, 64, s2);
  return c;
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) \
  hipemu::mfma_32x32x16((a), (b), (c), __COUNTER__, __COUNTER__)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) \
  hipemu::mfma_32x32x16((a), (b), (c), __COUNTER__, __COUNTER__)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) \
  hipemu::mfma_16x16x4((a), (b), (c), __COUNTER__, __COUNTER__)
