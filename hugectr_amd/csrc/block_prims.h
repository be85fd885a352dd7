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

// Walks a CSR key-parallel: f(bucket, key position) for every key of every bucket, one wavefront
// per 64 consecutive buckets with its lanes striding THE KEYS of those buckets -- coalesced and
// balanced whatever the bucket lengths (a thread-per-bucket loop over a 100-hot table reads its
// keys 800 bytes apart).  Lane b holds row_offset[base + b]; the bucket of a key position is the
// last of those <= it, found by a 6-step search over the lanes' registers.
// gridDim.y wavefronts share a 64-bucket chunk, taking its trips round robin: with mixed hotness
// (one 100-hot table among one-hot ones) the heavy chunks would otherwise be the kernel's tail.
// Must be called by whole wavefronts (all 64 lanes), blockDim a multiple of 64.
template <typename OffT, typename F>
__device__ __forceinline__ void for_each_key_wave(size_t buckets, const OffT* __restrict__ row_offset,
                                                  F f) {
  constexpr int kKW = 4;  // key positions per lane per trip: their searches and f's loads overlap
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const unsigned part = blockIdx.y, parts = gridDim.y;
  for (size_t base = wave * 64; base < buckets; base += nwaves * 64) {
    const int nb = (int)(buckets - base < (size_t)64 ? buckets - base : (size_t)64);
    const unsigned long long first = (unsigned long long)row_offset[base];
    const unsigned long long end = (unsigned long long)row_offset[base + (size_t)nb];
    if (first + (unsigned long long)part * (64 * kKW) >= end) continue;  // no trip for this part
    const unsigned long long mine =
        (unsigned long long)row_offset[base + (size_t)(lane < nb ? lane : nb)];
    // one key in every bucket of the chunk (one-hot input): lane = bucket = key position
    if (end - first == (unsigned long long)nb &&
        __all(lane >= nb || mine == first + (unsigned)lane)) {
      if (lane < nb) f(base + (size_t)lane, (size_t)(first + (unsigned)lane));
      continue;
    }
    for (unsigned long long jb = first + (unsigned long long)part * (64 * kKW); jb < end;
         jb += (unsigned long long)parts * (64 * kKW)) {
      int lo[kKW];
#pragma unroll
      for (int k = 0; k < kKW; k++) {
        const unsigned long long j = jb + (unsigned)(k * 64 + lane);
        lo[k] = 0;  // largest b in [0, nb) with offset[b] <= j (empty buckets in between skipped)
        if (jb + (unsigned)(k * 64) >= end) continue;  // (wave-uniform: nothing left for this k)
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
          const int probe = lo[k] + step;
          const unsigned long long v = __shfl(mine, probe < nb ? probe : 0, 64);
          if (probe < nb && v <= j) lo[k] = probe;
        }
      }
#pragma unroll
      for (int k = 0; k < kKW; k++) {
        const unsigned long long j = jb + (unsigned)(k * 64 + lane);
        if (j < end) f(base + (size_t)lo[k], (size_t)j);
      }
    }
  }
}

}  // namespace hctr
