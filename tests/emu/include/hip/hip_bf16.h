// <hip/hip_bf16.h> of the host interpreter -- TEST INFRASTRUCTURE ONLY (tests/emu/hipemu.h).
#pragma once
#include "hip_runtime.h"

struct alignas(2) __hip_bfloat16 {
  unsigned short bits;
};
static inline float __bfloat162float(__hip_bfloat16 b) {
  return hipemu::from_bits<float>((uint64_t)b.bits << 16);
}
static inline __hip_bfloat16 __float2bfloat16(float f) {  // round to nearest even, NaN kept quiet
  unsigned u = (unsigned)hipemu::to_bits(f);
  __hip_bfloat16 b;
  if ((u & 0x7fffffffu) > 0x7f800000u) {
    b.bits = (unsigned short)((u >> 16) | 0x40u);
    return b;
  }
  u += 0x7fffu + ((u >> 16) & 1u);
  b.bits = (unsigned short)(u >> 16);
  return b;
}
