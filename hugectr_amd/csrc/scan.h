// scan.h -- tile-based device-wide exclusive scan (3 launches, no host sync, no temp sizing).
#pragma once
#include "block_prims.h"
#include "common.h"

namespace hctr {
namespace scan_detail {
constexpr int kBlock = 256;
constexpr int kTile = 1024;

// ---- tile-based exclusive scan of per-bucket lengths -> CSR row offsets ------------------------
template <typename OffT>
__global__ void __launch_bounds__(kBlock)
    tile_sum_kernel(const OffT* __restrict__ lens, size_t n, size_t n_tiles,
                    unsigned long long* __restrict__ tile_sums) {
  __shared__ unsigned long long smem[kBlock / 64 + 1];
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    unsigned long long c = 0;
#pragma unroll
    for (int r = 0; r < kTile / kBlock; r++) {
      size_t i = tile * kTile + r * kBlock + threadIdx.x;
      if (i < n) c += (unsigned long long)lens[i];
    }
    unsigned long long tot = block_reduce_sum<unsigned long long, kBlock>(c, smem);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}

static __global__ void __launch_bounds__(1024)
    scan_tiles_u64_kernel(unsigned long long* sums, size_t m, unsigned long long* d_total) {
  __shared__ unsigned long long smem[1024 / 64 + 1];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < m; base += 1024) {
    size_t i = base + threadIdx.x;
    unsigned long long v = (i < m) ? sums[i] : 0ull;
    unsigned long long tot;
    unsigned long long ex = block_exclusive_scan<unsigned long long, 1024>(v, smem, &tot);
    unsigned long long c = carry;
    if (i < m) sums[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *d_total = carry;
}

// offsets[i] = exclusive prefix of lens; offsets[n] = total.  lens and offsets may not alias.
template <typename OffT>
__global__ void __launch_bounds__(kBlock)
    tile_downsweep_kernel(const OffT* __restrict__ lens, size_t n, size_t n_tiles,
                          const unsigned long long* __restrict__ tile_sums,
                          const unsigned long long* __restrict__ d_total,
                          OffT* __restrict__ offsets) {
  __shared__ unsigned long long smem[kBlock / 64 + 1];
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[n] = (OffT)*d_total;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    unsigned long long run = tile_sums[tile];
#pragma unroll
    for (int r = 0; r < kTile / kBlock; r++) {
      size_t i = tile * kTile + r * kBlock + threadIdx.x;
      unsigned long long v = (i < n) ? (unsigned long long)lens[i] : 0ull;
      unsigned long long tot;
      unsigned long long ex = block_exclusive_scan<unsigned long long, kBlock>(v, smem, &tot);
      if (i < n) offsets[i] = (OffT)(run + ex);
      run += tot;
    }
  }
}


}  // namespace scan_detail

// offsets[i] = exclusive prefix of lens[0..n), offsets[n] = total (also in *d_total, u64).
// tile_sums: scratch of ceil(n / 1024) + 1 u64.  lens and offsets must not alias.
template <typename OffT>
int exclusive_scan_to_offsets(const OffT* lens, size_t n, unsigned long long* tile_sums,
                              unsigned long long* d_total, OffT* offsets, hipStream_t s) {
  using namespace scan_detail;
  const size_t n_tiles = ceil_div<size_t>(n, kTile);
  const int tgrid = (int)(n_tiles < (size_t)kMaxGrid ? (n_tiles ? n_tiles : 1) : (size_t)kMaxGrid);
  hipLaunchKernelGGL(tile_sum_kernel<OffT>, dim3(tgrid), dim3(kBlock), 0, s, lens, n, n_tiles,
                     tile_sums);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_tiles_u64_kernel, dim3(1), dim3(1024), 0, s, tile_sums, n_tiles, d_total);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(tile_downsweep_kernel<OffT>, dim3(tgrid), dim3(kBlock), 0, s, lens, n, n_tiles,
                     tile_sums, d_total, offsets);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // namespace hctr
