// block_prims.h -- wave64 / workgroup scan + reduce primitives (gfx950, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hctr {

// inclusive scan across the 64 lanes of a wavefront
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T up = __shfl_up(v, d, 64);
    if (lane >= d) v += up;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// exclusive scan over a workgroup of BLOCK threads (BLOCK multiple of 64, <= 1024).
// Returns the exclusive prefix of `v` for this thread; *total receives the block sum.
// smem must hold BLOCK/64 + 1 elements of T.
template <typename T, int BLOCK>
__device__ __forceinline__ T block_exclusive_scan(T v, T* smem, T* total) {
  constexpr int NW = BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  T inc = wave_inclusive_scan(v);
  if (lane == 63) smem[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      T t = smem[w];
      smem[w] = run;
      run += t;
    }
    smem[NW] = run;
  }
  __syncthreads();
  T excl = inc - v + smem[wave];
  *total = smem[NW];
  __syncthreads();
  return excl;
}

template <typename T, int BLOCK>
__device__ __forceinline__ T block_reduce_sum(T v, T* smem) {
  constexpr int NW = BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  v = wave_reduce_sum(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  T r = 0;
#pragma unroll
  for (int w = 0; w < NW; w++) r += smem[w];
  __syncthreads();
  return r;
}

}  // namespace hctr
