// TEST INFRASTRUCTURE ONLY (see cuda_runtime_api.h): the rest of the CUDA device vocabulary that the
// reference's legacy-embedding kernels use (R/HugeCTR/src/embeddings/*_functor.cu,
// R/HugeCTR/src/optimizers/sparse_optimizer.cu) -- float2, the paired-half type and the four
// paired-half intrinsics, mixed-type min / max, and the two cub entry points the optimizer's host
// code calls -- on top of the IEEE binary16 `__half` of ref_shims/common.hpp.  Declarations and
// arithmetic defined by the CUDA documentation, no reference code.
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>
#include <type_traits>
#include <vector>

#include "cuda_runtime_api.h"

struct alignas(8) float2 {
  float x, y;
};
struct alignas(4) __half2 {
  __half x, y;
};
static inline __half2 __float2half2_rn(float v) {
  __half2 r;
  r.x = __float2half(v);
  r.y = r.x;
  return r;
}
static inline __half2 __float22half2_rn(float2 v) {
  __half2 r;
  r.x = __float2half(v.x);
  r.y = __float2half(v.y);
  return r;
}
static inline float2 __half22float2(__half2 v) { return float2{__half2float(v.x), __half2float(v.y)}; }
// (the product of two binary16 values is exact in binary32: one rounding, as the instruction does)
static inline __half2 __hmul2(__half2 a, __half2 b) {
  __half2 r;
  r.x = __float2half(__half2float(a.x) * __half2float(b.x));
  r.y = __float2half(__half2float(a.y) * __half2float(b.y));
  return r;
}
// (a sum of two binary16 values is exact in binary32 as well)
static inline __half2 __hadd2(__half2 a, __half2 b) {
  __half2 r;
  r.x = __float2half(__half2float(a.x) + __half2float(b.x));
  r.y = __float2half(__half2float(a.y) + __half2float(b.y));
  return r;
}

#ifdef REFSHIM_HALF_ARITH
// binary16 arithmetic of device code (cuda_fp16.hpp: operators on two __half values round once to
// binary16; a sum / product of two binary16 values is exact in binary32)
static inline __half operator+(const __half& a, const __half& b) {
  return __float2half(__half2float(a) + __half2float(b));
}
static inline __half operator-(const __half& a, const __half& b) {
  return __float2half(__half2float(a) - __half2float(b));
}
static inline __half operator*(const __half& a, const __half& b) {
  return __float2half(__half2float(a) * __half2float(b));
}
#endif
static inline unsigned short __half_as_ushort(__half h) { return h.bits; }
static inline __half __ushort_as_half(unsigned short u) {
  __half h;
  h.bits = u;
  return h;
}
struct alignas(16) float4 {
  float x, y, z, w;
};
// __shfl_sync on a 32-lane warp (the interpreter runs such code with set_wave_width(32)); the call
// site is the caller's source line
template <typename T>
static inline T __shfl_sync(unsigned, T v, int src, int width = 32, int site = __builtin_LINE()) {
  uint64_t b = 0;
  static_assert(sizeof(T) <= 8, "shuffles move at most 64 bits");
  memcpy(&b, &v, sizeof(T));
  b = hipemu::collective(hipemu::OP_SHFL, b, src, width, site);
  T r;
  memcpy(&r, &b, sizeof(T));
  return r;
}
#define warpSize 32

// CUDA's global min / max take mixed integer types (max(1, uint32_t), min(size_t, size_t) ...)
template <typename A, typename B>
static inline typename std::common_type<A, B>::type max(A a, B b) {
  using T = typename std::common_type<A, B>::type;
  return (T)a > (T)b ? (T)a : (T)b;
}
template <typename A, typename B>
static inline typename std::common_type<A, B>::type min(A a, B b) {
  using T = typename std::common_type<A, B>::type;
  return (T)a < (T)b ? (T)a : (T)b;
}

// cub::DeviceRadixSort::SortPairs / cub::DeviceScan::InclusiveSum by their documented contracts:
// a STABLE sort of (key, value) pairs on key bits [begin_bit, end_bit); an inclusive prefix sum.
// A null temp-storage pointer asks for the size only.
namespace cub {
struct DeviceRadixSort {
  template <typename K, typename V, typename N>
  static cudaError_t SortPairs(void* temp, size_t& temp_bytes, const K* kin, K* kout, const V* vin,
                               V* vout, N n, int begin_bit = 0, int end_bit = sizeof(K) * 8,
                               cudaStream_t = nullptr) {
    if (temp == nullptr) {
      temp_bytes = 16;
      return cudaSuccess;
    }
    using U = typename std::make_unsigned<K>::type;
    const int width = end_bit - begin_bit;
    const U mask = width >= (int)sizeof(K) * 8 ? ~(U)0 : (U)((((U)1) << width) - 1);
    std::vector<size_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      return (((U)kin[a] >> begin_bit) & mask) < (((U)kin[b] >> begin_bit) & mask);
    });
    for (size_t i = 0; i < (size_t)n; i++) {
      kout[i] = kin[order[i]];
      vout[i] = vin[order[i]];
    }
    return cudaSuccess;
  }
};
// cub::DeviceSegmentedRadixSort::SortPairs by its documented contract: every segment
// [begin_offsets[s], end_offsets[s]) sorted on its own, stably, on key bits [begin_bit, end_bit);
// items outside every segment are not touched
struct DeviceSegmentedRadixSort {
  template <typename K, typename V, typename N, typename O>
  static cudaError_t SortPairs(void* temp, size_t& temp_bytes, const K* kin, K* kout, const V* vin,
                               V* vout, N n, int num_segments, const O* begin_offsets,
                               const O* end_offsets, int begin_bit = 0,
                               int end_bit = sizeof(K) * 8, cudaStream_t = nullptr) {
    if (temp == nullptr) {
      temp_bytes = 16;
      return cudaSuccess;
    }
    (void)n;
    for (int s = 0; s < num_segments; s++) {
      const size_t b = (size_t)begin_offsets[s], e = (size_t)end_offsets[s];
      if (e > b) {
        size_t tb = 16;
        char tmp[16];
        DeviceRadixSort::SortPairs(tmp, tb, kin + b, kout + b, vin + b, vout + b, e - b, begin_bit,
                                   end_bit);
      }
    }
    return cudaSuccess;
  }
};
// cub::DeviceSelect::Flagged / If: the flagged (or accepted) items in their input order, their count
struct DeviceSelect {
  template <typename T, typename F, typename C, typename N>
  static cudaError_t Flagged(void* temp, size_t& temp_bytes, const T* in, const F* flags, T* out,
                             C* num_selected, N n, cudaStream_t = nullptr) {
    if (temp == nullptr) {
      temp_bytes = 16;
      return cudaSuccess;
    }
    size_t m = 0;
    for (size_t i = 0; i < (size_t)n; i++)
      if (flags[i]) out[m++] = in[i];
    *num_selected = (C)m;
    return cudaSuccess;
  }
  template <typename T, typename C, typename N, typename Op>
  static cudaError_t If(void* temp, size_t& temp_bytes, const T* in, T* out, C* num_selected, N n,
                        Op op, cudaStream_t = nullptr) {
    if (temp == nullptr) {
      temp_bytes = 16;
      return cudaSuccess;
    }
    size_t m = 0;
    for (size_t i = 0; i < (size_t)n; i++)
      if (op(in[i])) out[m++] = in[i];
    *num_selected = (C)m;
    return cudaSuccess;
  }
};
struct DeviceScan {
  template <typename T, typename N>
  static cudaError_t InclusiveSum(void* temp, size_t& temp_bytes, const T* in, T* out, N n,
                                  cudaStream_t = nullptr) {
    if (temp == nullptr) {
      temp_bytes = 16;
      return cudaSuccess;
    }
    T acc = 0;
    for (size_t i = 0; i < (size_t)n; i++) {
      acc += in[i];
      out[i] = acc;
    }
    return cudaSuccess;
  }
};
}  // namespace cub
