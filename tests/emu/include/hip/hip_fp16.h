// <hip/hip_fp16.h> of the host interpreter -- TEST INFRASTRUCTURE ONLY (tests/emu/hipemu.h).
// binary16 through the compiler's _Float16 (IEEE round-to-nearest-even conversions).
#pragma once
#include "hip_runtime.h"

struct alignas(2) __half {
  _Float16 v;
  __half() = default;
  __half(float f) : v((_Float16)f) {}
  operator float() const { return (float)v; }
};
struct alignas(4) __half2 {
  __half x, y;
};
static inline float __half2float(__half h) { return (float)h.v; }
static inline __half __float2half_rn(float f) {
  __half h;
  h.v = (_Float16)f;
  return h;
}
static inline __half __float2half(float f) { return __float2half_rn(f); }
static inline float2 __half22float2(__half2 h) { return float2{(float)h.x.v, (float)h.y.v}; }
static inline __half2 __floats2half2_rn(float a, float b) {
  __half2 h;
  h.x = __float2half_rn(a);
  h.y = __float2half_rn(b);
  return h;
}
static inline __half2 __float22half2_rn(float2 f) { return __floats2half2_rn(f.x, f.y); }
