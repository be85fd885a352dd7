// embedding_kernels.hip -- gather + intra-slot pooling forward, and the all-to-all reorder maps.
//
// forward_sum / forward_mean semantics: R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu:28-241
// (fp32 accumulate in key order j = 0..n-1; SIZE_MAX row index contributes 0; mean multiplies by
// 1.0f/n when n > 1).  The reference launches one block per SAMPLE with D threads that walk the
// slots serially; here a "group" of D/4 lanes owns BU buckets at a time, each lane moving 16 B,
// so a wavefront keeps 64/(D/4) * BU independent row reads in flight and every row read/write is
// a fully coalesced D*4-byte segment.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "common.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;

template <typename OutT>
struct Store4;
template <>
struct Store4<float> {
  __device__ __forceinline__ static void st(float* p, float4 v) {
    *reinterpret_cast<float4*>(p) = v;
  }
  __device__ __forceinline__ static void st1(float* p, float v) { *p = v; }
};
template <>
struct Store4<__half> {
  __device__ __forceinline__ static void st(__half* p, float4 v) {
    __half2 a = __float22half2_rn(make_float2(v.x, v.y));
    __half2 b = __float22half2_rn(make_float2(v.z, v.w));
    uint2 u;
    u.x = *reinterpret_cast<uint32_t*>(&a);
    u.y = *reinterpret_cast<uint32_t*>(&b);
    *reinterpret_cast<uint2*>(p) = u;
  }
  __device__ __forceinline__ static void st1(__half* p, float v) { *p = __float2half_rn(v); }
};
template <>
struct Store4<__hip_bfloat16> {
  __device__ __forceinline__ static void st(__hip_bfloat16* p, float4 v) {
    __hip_bfloat16 a = __float2bfloat16(v.x), b = __float2bfloat16(v.y);
    __hip_bfloat16 c = __float2bfloat16(v.z), d = __float2bfloat16(v.w);
    uint2 u;
    u.x = (uint32_t) * reinterpret_cast<uint16_t*>(&a) |
          ((uint32_t) * reinterpret_cast<uint16_t*>(&b) << 16);
    u.y = (uint32_t) * reinterpret_cast<uint16_t*>(&c) |
          ((uint32_t) * reinterpret_cast<uint16_t*>(&d) << 16);
    *reinterpret_cast<uint2*>(p) = u;
  }
  __device__ __forceinline__ static void st1(__hip_bfloat16* p, float v) {
    *p = __float2bfloat16(v);
  }
};

// Mean of a 16-bit output, even embedding_vec_size (forward_mean_align2_kernel,
// forward_per_gpu_functor.cu:136-176): the fp32 sum and the scaler are BOTH rounded to the output
// type and multiplied there (__hmul2).  The product of two 11-bit (8-bit) significands is exact in
// fp32, so rounding it once on the store reproduces the half multiply.  fp32 output and the generic
// (odd size) kernel multiply in fp32 (forward_mean_kernel :98-131).  bf16 follows the fp16 rule.
template <typename OutT>
__device__ __forceinline__ float mean_product(float sum, float sc) {
  return sum * sc;
}
template <>
__device__ __forceinline__ float mean_product<__half>(float sum, float sc) {
  return __half2float(__float2half_rn(sum)) * __half2float(__float2half_rn(sc));
}
template <>
__device__ __forceinline__ float mean_product<__hip_bfloat16>(float sum, float sc) {
  return __bfloat162float(__float2bfloat16(sum)) * __bfloat162float(__float2bfloat16(sc));
}

// Output row of bucket u.  Identity, or -- embedding_collection on one GPU with a batch-major
// output -- the transpose of the [lookup][sample] bucket order the owner pools in into
// [sample][lookup]: the reorder the reference runs as a separate pass after its all-to-all
// (NetworkForward, R/HugeCTR/embedding/operators/network_forward.cu) folded into the store address.
struct OutMap {
  uint32_t inner;  // samples per lookup (0 = identity)
  uint32_t outer;  // lookups
};
__device__ __forceinline__ size_t out_row(size_t u, OutMap m) {
  if (m.inner == 0u) return u;
  const uint32_t v = (uint32_t)u;
  return (size_t)(v % m.inner) * m.outer + v / m.inner;
}

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

// LPR lanes per row (D = 4*LPR), BU buckets per group per iteration.
//
// one_hot (device flag, may be NULL): set by the caller's index stage when EVERY bucket of the batch
// holds exactly one key (the Criteo / DLRM case).  Then bucket u's only key sits at position u, the
// row offsets need not be read, and a bucket costs two dependent reads (row index, row) instead of
// three.  A lane group walks ~26 iterations of its grid-stride loop at the bench shape, and a loop
// that finishes one iteration before starting the next is bound by those round trips, not by bytes
// (192 us for 1.34 GB algorithmic, 0.79 GB of it real HBM traffic); so the one-hot loop loads the
// row indices of iteration i + 1 behind the rows of iteration i.  Every load of it is
// unconditional on a clamped address: a load under a branch ends its basic block with
// `s_waitcnt vmcnt(0)` and serialises exactly what the loop is meant to overlap.
template <int LPR, int BU, typename OffT, typename OutT>
__global__ void __launch_bounds__(kBlock)
    pool_vec4_kernel(size_t buckets, int combiner, const OffT* __restrict__ row_offset,
                     const uint64_t* __restrict__ value_index, const float* __restrict__ table,
                     OutT* __restrict__ out, const uint32_t* __restrict__ one_hot, OutMap om) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;  // groups per block
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const size_t stride = (size_t)gridDim.x * GPB * BU;
  if (one_hot != nullptr && *one_hot != 0u) {
    const size_t first = ((size_t)blockIdx.x * GPB + g) * BU;
    if (first >= buckets) return;
    const size_t last = buckets - 1;
    uint64_t idx[BU], nxt[BU];
#pragma unroll
    for (int k = 0; k < BU; k++) nxt[k] = value_index[first + k < buckets ? first + k : last];
    for (size_t u0 = first; u0 < buckets; u0 += stride) {
      float4 r[BU];
#pragma unroll
      for (int k = 0; k < BU; k++) {
        idx[k] = nxt[k];
        const uint64_t row = idx[k] != kInvalidIndex ? idx[k] : 0ull;  // always a legal read
        r[k] = ld4(table + row * (uint64_t)D + l * 4);
      }
      const size_t un = u0 + stride;
#pragma unroll
      for (int k = 0; k < BU; k++) nxt[k] = value_index[un + k < buckets ? un + k : last];
#pragma unroll
      for (int k = 0; k < BU; k++) {
        if (u0 + k < buckets) {
          const bool live = idx[k] != kInvalidIndex;
          // sum = 0.0f + row, one key: mean == sum (n = 1)
          const float4 v = make_float4(0.f + (live ? r[k].x : 0.f), 0.f + (live ? r[k].y : 0.f),
                                       0.f + (live ? r[k].z : 0.f), 0.f + (live ? r[k].w : 0.f));
          Store4<OutT>::st(out + out_row(u0 + k, om) * (size_t)D + l * 4, v);
        }
      }
    }
    return;
  }
  for (size_t u0 = ((size_t)blockIdx.x * GPB + g) * BU; u0 < buckets; u0 += stride) {
    long long off[BU];
    int n[BU];
    uint64_t idx0[BU];
    float4 acc[BU];
#pragma unroll
    for (int k = 0; k < BU; k++) {
      const size_t u = u0 + k;
      if (u < buckets) {
        off[k] = (long long)row_offset[u];
        n[k] = (int)((long long)row_offset[u + 1] - off[k]);
      } else {
        off[k] = 0;
        n[k] = 0;
      }
    }
#pragma unroll
    for (int k = 0; k < BU; k++) idx0[k] = (n[k] > 0) ? value_index[off[k]] : kInvalidIndex;
#pragma unroll
    for (int k = 0; k < BU; k++) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx0[k] != kInvalidIndex) r = ld4(table + idx0[k] * (uint64_t)D + l * 4);
      // sum starts at 0.0f and adds in key order, as the reference does
      acc[k] = make_float4(0.f + r.x, 0.f + r.y, 0.f + r.z, 0.f + r.w);
    }
#pragma unroll
    for (int k = 0; k < BU; k++) {
      for (int j = 1; j < n[k]; j++) {
        const uint64_t idx = value_index[off[k] + j];
        if (idx != kInvalidIndex) {
          float4 r = ld4(table + idx * (uint64_t)D + l * 4);
          acc[k].x += r.x;
          acc[k].y += r.y;
          acc[k].z += r.z;
          acc[k].w += r.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < BU; k++) {
      const size_t u = u0 + k;
      if (u < buckets) {
        float4 v = acc[k];
        if (combiner == 1 && n[k] > 1) {
          const float sc = 1.0f / (float)n[k];
          v.x = mean_product<OutT>(v.x, sc);
          v.y = mean_product<OutT>(v.y, sc);
          v.z = mean_product<OutT>(v.z, sc);
          v.w = mean_product<OutT>(v.w, sc);
        }
        Store4<OutT>::st(out + out_row(u, om) * (size_t)D + l * 4, v);
      }
    }
  }
}

// Multi-hot variant.  A lane group owns NB consecutive buckets = one contiguous range of keys and
// walks that flat range JU keys at a time: the JU row reads of a chunk are independent of bucket
// boundaries and are issued back to back (the next chunk's indices are already in registers), so
// a group keeps JU rows in flight whether its buckets hold 1 key or 1000.  Sums still start at
// 0.0f and add in key order inside each bucket -- same bits as the one-hot kernel and the oracle.
template <int LPR, int NB, int JU, typename OffT, typename OutT>
__global__ void __launch_bounds__(kBlock)
    pool_flat_kernel(size_t buckets, int combiner, const OffT* __restrict__ row_offset,
                     const uint64_t* __restrict__ value_index, const float* __restrict__ table,
                     OutT* __restrict__ out, OutMap om) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const size_t stride = (size_t)gridDim.x * GPB * NB;
  for (size_t u0 = ((size_t)blockIdx.x * GPB + g) * NB; u0 < buckets; u0 += stride) {
    const int nb = (int)((buckets - u0) < (size_t)NB ? (buckets - u0) : (size_t)NB);
    const long long kbeg = (long long)row_offset[u0];
    int e[NB];  // bucket ends relative to kbeg (group-uniform)
#pragma unroll
    for (int i = 0; i < NB; i++)
      e[i] = (int)((long long)row_offset[u0 + (size_t)(i < nb ? i + 1 : nb)] - kbeg);
    const int total = e[NB - 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur = -1, cnt = 0;  // bucket being summed, keys seen in it
    uint64_t idx[JU], nxt[JU];
#pragma unroll
    for (int k = 0; k < JU; k++)
      nxt[k] = total > 0 ? value_index[kbeg + (k < total ? k : total - 1)] : kInvalidIndex;
    for (int j0 = 0; j0 < total; j0 += JU) {
      float4 r[JU];
#pragma unroll
      for (int k = 0; k < JU; k++) {
        idx[k] = nxt[k];
        const uint64_t row = idx[k] != kInvalidIndex ? idx[k] : 0ull;  // always a legal read
        r[k] = ld4(table + row * (uint64_t)D + l * 4);
      }
      if (j0 + JU < total) {
#pragma unroll
        for (int k = 0; k < JU; k++) {
          const int p = j0 + JU + k;
          nxt[k] = value_index[kbeg + (p < total ? p : total - 1)];
        }
      }
#pragma unroll
      for (int k = 0; k < JU; k++) {
        const int p = j0 + k;
        if (p < total) {
          int b = 0;  // bucket of position p = number of bucket ends <= p
#pragma unroll
          for (int i = 0; i < NB - 1; i++) b += (p >= e[i]) ? 1 : 0;
          if (b != cur) {
            if (cur >= 0) {
              float4 v = acc;
              if (combiner == 1 && cnt > 1) {
                const float sc = 1.0f / (float)cnt;
                v.x = mean_product<OutT>(v.x, sc);
                v.y = mean_product<OutT>(v.y, sc);
                v.z = mean_product<OutT>(v.z, sc);
                v.w = mean_product<OutT>(v.w, sc);
              }
              Store4<OutT>::st(out + out_row(u0 + (size_t)cur, om) * (size_t)D + l * 4, v);
            }
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            cur = b;
            cnt = 0;
          }
          cnt++;
          if (idx[k] != kInvalidIndex) {
            acc.x += r[k].x;
            acc.y += r[k].y;
            acc.z += r[k].z;
            acc.w += r[k].w;
          }
        }
      }
    }
    if (cur >= 0) {
      float4 v = acc;
      if (combiner == 1 && cnt > 1) {
        const float sc = 1.0f / (float)cnt;
        v.x = mean_product<OutT>(v.x, sc);
        v.y = mean_product<OutT>(v.y, sc);
        v.z = mean_product<OutT>(v.z, sc);
        v.w = mean_product<OutT>(v.w, sc);
      }
      Store4<OutT>::st(out + out_row(u0 + (size_t)cur, om) * (size_t)D + l * 4, v);
    }
    // empty buckets pool to zeros
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int len = e[i] - (i > 0 ? e[i - 1] : 0);
      if (i < nb && len == 0)
        Store4<OutT>::st(out + out_row(u0 + (size_t)i, om) * (size_t)D + l * 4,
                         make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
}

// any embedding_vec_size: one wavefront per bucket, lanes stride over the vector
template <typename OffT, typename OutT>
__global__ void __launch_bounds__(kBlock)
    pool_generic_kernel(size_t buckets, int D, int combiner, const OffT* __restrict__ row_offset,
                        const uint64_t* __restrict__ value_index, const float* __restrict__ table,
                        OutT* __restrict__ out, OutMap om) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t u = wave; u < buckets; u += nwaves) {
    const long long off = (long long)row_offset[u];
    const int n = (int)((long long)row_offset[u + 1] - off);
    const float sc = (combiner == 1 && n > 1) ? 1.0f / (float)n : 1.0f;
    for (int v = lane; v < D; v += 64) {
      float sum = 0.0f;
      for (int j = 0; j < n; j++) {
        const uint64_t idx = value_index[off + j];
        sum += (idx != kInvalidIndex) ? table[idx * (uint64_t)D + v] : 0.0f;
      }
      // even sizes take the reference's align2 rule also here (e.g. D = 6, 10)
      const float m = (D % 2 == 0) ? mean_product<OutT>(sum, sc) : sum * sc;
      Store4<OutT>::st1(out + out_row(u, om) * (size_t)D + v, (combiner == 1) ? m : sum);
    }
  }
}

// ---- pooling through per-key row pointers (the ILookup::lookup(..., float** embedding_vec) seam,
//      R/HugeCTR/embedding/embedding_table.hpp:22-33, consumed by generic_lookup.cuh:318-416): the
//      rows of a dynamic table live in per-class stores, so a key carries its row's address.
//      nullptr = key not in the table (contributes 0, still counts for the mean).
template <int LPR, int BU, typename OutT>
__global__ void __launch_bounds__(kBlock)
    pool_ptrs_vec4_kernel(size_t buckets, int combiner, const long long* __restrict__ row_offset,
                          const float* const* __restrict__ rows, OutT* __restrict__ out,
                          OutMap om) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const size_t stride = (size_t)gridDim.x * GPB * BU;
  for (size_t u0 = ((size_t)blockIdx.x * GPB + g) * BU; u0 < buckets; u0 += stride) {
    long long off[BU];
    int n[BU];
    float4 acc[BU];
#pragma unroll
    for (int k = 0; k < BU; k++) {
      const size_t u = u0 + k;
      off[k] = (u < buckets) ? row_offset[u] : 0;
      n[k] = (u < buckets) ? (int)(row_offset[u + 1] - off[k]) : 0;
      acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < BU; k++) {
      for (int j = 0; j < n[k]; j++) {
        const float* r = rows[off[k] + j];
        if (r != nullptr) {
          const float4 v = ld4(r + l * 4);
          acc[k].x += v.x;
          acc[k].y += v.y;
          acc[k].z += v.z;
          acc[k].w += v.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < BU; k++) {
      const size_t u = u0 + k;
      if (u < buckets) {
        float4 v = acc[k];
        if (combiner == 1 && n[k] > 1) {
          const float sc = 1.0f / (float)n[k];
          v.x = mean_product<OutT>(v.x, sc);
          v.y = mean_product<OutT>(v.y, sc);
          v.z = mean_product<OutT>(v.z, sc);
          v.w = mean_product<OutT>(v.w, sc);
        }
        Store4<OutT>::st(out + out_row(u, om) * (size_t)D + l * 4, v);
      }
    }
  }
}

template <typename OutT>
__global__ void __launch_bounds__(kBlock)
    pool_ptrs_generic_kernel(size_t buckets, int D, int combiner,
                             const long long* __restrict__ row_offset,
                             const float* const* __restrict__ rows, OutT* __restrict__ out,
                             OutMap om) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t u = wave; u < buckets; u += nwaves) {
    const long long off = row_offset[u];
    const int n = (int)(row_offset[u + 1] - off);
    const float sc = (combiner == 1 && n > 1) ? 1.0f / (float)n : 1.0f;
    for (int v = lane; v < D; v += 64) {
      float sum = 0.0f;
      for (int j = 0; j < n; j++) {
        const float* r = rows[off + j];
        sum += (r != nullptr) ? r[v] : 0.0f;
      }
      const float m = (D % 2 == 0) ? mean_product<OutT>(sum, sc) : sum * sc;
      Store4<OutT>::st1(out + out_row(u, om) * (size_t)D + v, (combiner == 1) ? m : sum);
    }
  }
}

template <typename OutT>
int launch_pool_ptrs(size_t buckets, int D, int combiner, const long long* ro,
                     const float* const* rows, OutT* out, hipStream_t s, OutMap om) {
  // row stores are hipMalloc'ed ([capacity][D] fp32): rows are 16-byte aligned iff D % 4 == 0
  const bool aligned = reinterpret_cast<uintptr_t>(out) % 16 == 0;
#define HCTR_PP(LPR_)                                                                          \
  case LPR_:                                                                                    \
    hipLaunchKernelGGL((pool_ptrs_vec4_kernel<LPR_, 4, OutT>),                                  \
                       dim3(grid_for(ceil_div<size_t>(buckets, 4), kBlock / LPR_, 256 * 8)),    \
                       dim3(kBlock), 0, s, buckets, combiner, ro, rows, out, om);               \
    break;
  bool done = aligned && D % 4 == 0;
  if (done) {
    switch (D / 4) {
      HCTR_PP(1) HCTR_PP(2) HCTR_PP(4) HCTR_PP(8) HCTR_PP(16) HCTR_PP(32) HCTR_PP(64)
      default: done = false;
    }
  }
#undef HCTR_PP
  if (!done)
    hipLaunchKernelGGL((pool_ptrs_generic_kernel<OutT>), dim3(grid_for(buckets * 64, kBlock)),
                       dim3(kBlock), 0, s, buckets, D, combiner, ro, rows, out, om);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// ---- weighted pooling (SOK lookup_sparse with sp_weights, R/sparse_operation_kit/.../lookup.py
//      :425-541): out[b] = sum_j w_j * row_j, mean divides by sum_j w_j; without weights w = 1 and
//      mean divides by the key count.  One wavefront per bucket, lanes stride over the vector.
__global__ void __launch_bounds__(kBlock)
    pool_weighted_kernel(size_t buckets, int D, int combiner, const long long* __restrict__ ro,
                         const uint64_t* __restrict__ value_index,
                         const float* __restrict__ weights, const float* __restrict__ table,
                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t u = wave; u < buckets; u += nwaves) {
    const long long off = ro[u];
    const int n = (int)(ro[u + 1] - off);
    float denom = 0.f;
    for (int j = 0; j < n; j++) denom += weights ? weights[off + j] : 1.0f;
    for (int v = lane; v < D; v += 64) {
      float sum = 0.0f;
      for (int j = 0; j < n; j++) {
        const uint64_t idx = value_index[off + j];
        const float w = weights ? weights[off + j] : 1.0f;
        if (idx != kInvalidIndex) sum += w * table[idx * (uint64_t)D + v];
      }
      out[u * (size_t)D + v] = (combiner == 1 && n > 0) ? sum / denom : sum;
    }
  }
}

// vec4 form of the two SOK kernels: LPR = D/4 lanes per bucket, float4 per lane, 4 keys in flight
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    pool_weighted_vec4_kernel(size_t buckets, int combiner, const long long* __restrict__ ro,
                              const uint64_t* __restrict__ value_index,
                              const float* __restrict__ weights, const float* __restrict__ table,
                              float* __restrict__ out) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (size_t u = (size_t)blockIdx.x * GPB + g; u < buckets; u += (size_t)gridDim.x * GPB) {
    const long long off = ro[u];
    const int n = (int)(ro[u + 1] - off);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float denom = 0.f;
    for (int j0 = 0; j0 < n; j0 += 4) {
      float4 r[4];
      float w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int j = j0 + k < n ? j0 + k : n - 1;
        const uint64_t idx = value_index[off + j];
        w[k] = (j0 + k < n) ? (weights ? weights[off + j] : 1.0f) : 0.0f;
        const bool ok = idx != kInvalidIndex;
        r[k] = ld4(table + (ok ? idx : 0ull) * (uint64_t)D + l * 4);
        if (!ok) r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (j0 + k < n) {
          denom += w[k];
          acc.x += w[k] * r[k].x;
          acc.y += w[k] * r[k].y;
          acc.z += w[k] * r[k].z;
          acc.w += w[k] * r[k].w;
        }
      }
    }
    if (combiner == 1 && n > 0) {
      acc.x /= denom;
      acc.y /= denom;
      acc.z /= denom;
      acc.w /= denom;
    }
    *reinterpret_cast<float4*>(out + u * (size_t)D + l * 4) = acc;
  }
}

template <int LPR>
__global__ void __launch_bounds__(kBlock)
    expand_key_grads_vec4_kernel(size_t buckets, int combiner, const long long* __restrict__ ro,
                                 const float* __restrict__ weights, const float* __restrict__ top,
                                 float* __restrict__ out) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (size_t u = (size_t)blockIdx.x * GPB + g; u < buckets; u += (size_t)gridDim.x * GPB) {
    const long long off = ro[u];
    const int n = (int)(ro[u + 1] - off);
    if (n == 0) continue;
    float denom = 1.0f;
    if (combiner == 1) {
      denom = 0.f;
      for (int j = 0; j < n; j++) denom += weights ? weights[off + j] : 1.0f;
    }
    const float4 t = ld4(top + u * (size_t)D + l * 4);
    for (int j = 0; j < n; j++) {
      const float sc = (weights ? weights[off + j] : 1.0f) / denom;
      *reinterpret_cast<float4*>(out + (size_t)(off + j) * D + l * 4) =
          make_float4(t.x * sc, t.y * sc, t.z * sc, t.w * sc);
    }
  }
}

// gradient of the pooled vector w.r.t. every looked-up row: g_j = top[b] * w_j (/ denom for mean)
__global__ void __launch_bounds__(kBlock)
    expand_key_grads_kernel(size_t buckets, int D, int combiner, const long long* __restrict__ ro,
                            const float* __restrict__ weights, const float* __restrict__ top,
                            float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t u = wave; u < buckets; u += nwaves) {
    const long long off = ro[u];
    const int n = (int)(ro[u + 1] - off);
    float denom = 1.0f;
    if (combiner == 1) {
      denom = 0.f;
      for (int j = 0; j < n; j++) denom += weights ? weights[off + j] : 1.0f;
    }
    for (int j = 0; j < n; j++) {
      const float sc = (weights ? weights[off + j] : 1.0f) / denom;
      for (int v = lane; v < D; v += 64)
        out[(size_t)(off + j) * D + v] = top[u * (size_t)D + v] * sc;
    }
  }
}

template <typename OffT, typename OutT>
int launch_pool(size_t buckets, int D, int combiner, const OffT* ro, const uint64_t* vi,
                const float* table, OutT* out, bool multi_hot, hipStream_t s,
                const uint32_t* one_hot, OutMap om) {
#define HCTR_POOL_CASE(LPR_, BU_)                                                              \
  {                                                                                            \
    constexpr int GPB = kBlock / LPR_;                                                         \
    if (multi_hot) {                                                                           \
      const int grid = grid_for(ceil_div<size_t>(buckets, (size_t)8), GPB, 256 * 8);          \
      hipLaunchKernelGGL((pool_flat_kernel<LPR_, 8, 8, OffT, OutT>), dim3(grid), dim3(kBlock), \
                         0, s, buckets, combiner, ro, vi, table, out, om);                     \
    } else {                                                                                   \
      const int grid = grid_for(ceil_div<size_t>(buckets, (size_t)BU_), GPB, 256 * 8);        \
      hipLaunchKernelGGL((pool_vec4_kernel<LPR_, BU_, OffT, OutT>), dim3(grid), dim3(kBlock), \
                         0, s, buckets, combiner, ro, vi, table, out, one_hot, om);            \
    }                                                                                          \
  }
  const bool aligned = (reinterpret_cast<uintptr_t>(table) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  if (aligned && D % 4 == 0) {
    switch (D / 4) {
      case 1: HCTR_POOL_CASE(1, 4) break;
      case 2: HCTR_POOL_CASE(2, 4) break;
      case 4: HCTR_POOL_CASE(4, 4) break;
      case 8: HCTR_POOL_CASE(8, 4) break;
      case 16: HCTR_POOL_CASE(16, 4) break;
      case 32: HCTR_POOL_CASE(32, 4) break;
      case 64: HCTR_POOL_CASE(64, 4) break;
      default: {
        const int grid = grid_for(buckets * 64, kBlock);
        hipLaunchKernelGGL((pool_generic_kernel<OffT, OutT>), dim3(grid), dim3(kBlock), 0, s,
                           buckets, D, combiner, ro, vi, table, out, om);
      }
    }
  } else {
    const int grid = grid_for(buckets * 64, kBlock);
    hipLaunchKernelGGL((pool_generic_kernel<OffT, OutT>), dim3(grid), dim3(kBlock), 0, s, buckets,
                       D, combiner, ro, vi, table, out, om);
  }
#undef HCTR_POOL_CASE
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// ---- reorder ---------------------------------------------------------------------------------
// FWD: out[b][s][:] = in[(offset_pre(g) + b*spg(g) + s/N)][:], g = s % N
// (forward_reorder_functor.cu:43-57); BWD is the inverse map.
template <typename T, bool FWD>
__global__ void __launch_bounds__(kBlock)
    reorder_kernel(size_t bpg, int S, int D, int N, const T* __restrict__ in,
                   T* __restrict__ out) {
  const size_t total = bpg * (size_t)S * (size_t)D;
  const int q = S / N, rem = S % N;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const int d = (int)(i % D);
    const size_t bs = i / D;
    const int s = (int)(bs % S);
    const size_t b = bs / S;
    const int g = s % N;
    const int spg = q + (g < rem ? 1 : 0);
    const size_t offset_pre = bpg * ((size_t)g * q + (size_t)(g < rem ? g : rem));
    const size_t a2a = (offset_pre + b * spg + (size_t)(s / N)) * D + d;
    if (FWD) out[i] = in[a2a];
    else out[a2a] = in[i];
  }
}

template <bool FWD>
int launch_reorder(size_t bpg, int S, int D, int N, const void* in, void* out, int dtype,
                   hipStream_t s) {
  if (bpg == 0 || S == 0) return HCTR_OK;
  const int esz = (dtype == HCTR_EMB_F32) ? 4 : 2;
  const size_t row_bytes = (size_t)D * esz;
  const bool a16 = row_bytes % 16 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (a16) {
    const int D16 = (int)(row_bytes / 16);
    const size_t total = bpg * (size_t)S * D16;
    hipLaunchKernelGGL((reorder_kernel<float4, FWD>), dim3(grid_for(total, kBlock)), dim3(kBlock),
                       0, s, bpg, S, D16, N, (const float4*)in, (float4*)out);
  } else if (esz == 4) {
    const size_t total = bpg * (size_t)S * D;
    hipLaunchKernelGGL((reorder_kernel<float, FWD>), dim3(grid_for(total, kBlock)), dim3(kBlock),
                       0, s, bpg, S, D, N, (const float*)in, (float*)out);
  } else {
    const size_t total = bpg * (size_t)S * D;
    hipLaunchKernelGGL((reorder_kernel<uint16_t, FWD>), dim3(grid_for(total, kBlock)),
                       dim3(kBlock), 0, s, bpg, S, D, N, (const uint16_t*)in, (uint16_t*)out);
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // namespace

int forward_pool_dispatch(size_t buckets, int D, int combiner, const void* ro, int key_type,
                          const uint64_t* vi, const float* table, void* out, int out_dtype,
                          bool multi_hot, hipStream_t s, const uint32_t* one_hot,
                          uint32_t map_inner = 0, uint32_t map_outer = 0) {
  if (buckets == 0) return HCTR_OK;
  const OutMap om{map_inner, map_outer};
  if (map_inner != 0u &&
      ((size_t)map_inner * map_outer != buckets || buckets > (size_t)0xFFFFFFFFu)) {
    set_error("output map: inner * outer must equal the bucket count (< 2^32)");
    return HCTR_ERR_INVALID_ARG;
  }
  // HCTR_POOL_KERNEL=bucket|flat pins the kernel (measurements); default: the caller's hint
  static const int forced = [] {
    const char* v = getenv("HCTR_POOL_KERNEL");
    if (!v) return -1;
    return v[0] == 'f' ? 1 : (v[0] == 'b' ? 0 : -1);
  }();
  if (forced >= 0) multi_hot = forced == 1;
#define HCTR_POOL_OUT(OffT)                                                                       \
  switch (out_dtype) {                                                                            \
    case HCTR_EMB_F32:                                                                            \
      return launch_pool<OffT, float>(buckets, D, combiner, (const OffT*)ro, vi, table,           \
                                      (float*)out, multi_hot, s, one_hot, om);                      \
    case HCTR_EMB_F16:                                                                            \
      return launch_pool<OffT, __half>(buckets, D, combiner, (const OffT*)ro, vi, table,          \
                                       (__half*)out, multi_hot, s, one_hot, om);                    \
    case HCTR_EMB_BF16:                                                                           \
      return launch_pool<OffT, __hip_bfloat16>(buckets, D, combiner, (const OffT*)ro, vi, table,  \
                                               (__hip_bfloat16*)out, multi_hot, s, one_hot, om);   \
    default:                                                                                      \
      set_error("out_dtype");                                                                     \
      return HCTR_ERR_INVALID_ARG;                                                                \
  }
  if (key_type == HCTR_KEY_U32) {
    HCTR_POOL_OUT(uint32_t)
  } else if (key_type == HCTR_KEY_I64) {
    HCTR_POOL_OUT(long long)
  }
#undef HCTR_POOL_OUT
  set_error("key_type");
  return HCTR_ERR_INVALID_ARG;
}

}  // namespace hctr

using namespace hctr;

extern "C" {

int hctr_forward_pool(size_t buckets, int vec_size, int combiner, const void* row_offset,
                      int key_type, const uint64_t* value_index, const float* table, void* out,
                      int out_dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  HCTR_REQUIRE(buckets == 0 || (row_offset && value_index && table && out), "null pointer");
  return forward_pool_dispatch(buckets, vec_size, combiner, row_offset, key_type, value_index,
                               table, out, out_dtype, false, as_stream(stream), nullptr);
}

static int forward_pool_ptrs_impl(size_t buckets, int vec_size, int combiner,
                                  const int64_t* row_offset, const float* const* rows, void* out,
                                  int out_dtype, OutMap om, hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  if (buckets == 0) return HCTR_OK;
  HCTR_REQUIRE(row_offset && rows && out, "null pointer");
  HCTR_REQUIRE(om.inner == 0u || ((size_t)om.inner * om.outer == buckets &&
                                  buckets <= (size_t)0xFFFFFFFFu),
               "output map: samples * lookups must equal the bucket count (< 2^32)");
  hipStream_t s = as_stream(stream);
  const long long* ro = (const long long*)row_offset;
  switch (out_dtype) {
    case HCTR_EMB_F32:
      return launch_pool_ptrs<float>(buckets, vec_size, combiner, ro, rows, (float*)out, s, om);
    case HCTR_EMB_F16:
      return launch_pool_ptrs<__half>(buckets, vec_size, combiner, ro, rows, (__half*)out, s, om);
    case HCTR_EMB_BF16:
      return launch_pool_ptrs<__hip_bfloat16>(buckets, vec_size, combiner, ro, rows,
                                              (__hip_bfloat16*)out, s, om);
    default:
      HCTR_REQUIRE(false, "out_dtype");
  }
  return HCTR_OK;
}

int hctr_forward_pool_ptrs(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                           const float* const* rows, void* out, int out_dtype,
                           hctr_stream_t stream) {
  return forward_pool_ptrs_impl(buckets, vec_size, combiner, row_offset, rows, out, out_dtype,
                                OutMap{0u, 0u}, stream);
}

int hctr_forward_pool_ptrs_mapped(size_t buckets, int vec_size, int combiner,
                                  const int64_t* row_offset, const float* const* rows, void* out,
                                  int out_dtype, size_t samples, size_t lookups,
                                  hctr_stream_t stream) {
  HCTR_REQUIRE(samples <= 0xFFFFFFFFull && lookups <= 0xFFFFFFFFull &&
                   ((samples == 0) == (lookups == 0)),
               "samples / lookups");
  return forward_pool_ptrs_impl(buckets, vec_size, combiner, row_offset, rows, out, out_dtype,
                                OutMap{(uint32_t)samples, (uint32_t)lookups}, stream);
}

int hctr_forward_pool_weighted(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                               const uint64_t* value_index, const float* weights,
                               const float* table, float* out, hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  if (buckets == 0) return HCTR_OK;
  HCTR_REQUIRE(row_offset && value_index && table && out, "null pointer");
  hipStream_t s = as_stream(stream);
  const bool v4 = vec_size % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0;
  bool done = false;
#define HCTR_PW(LPR_)                                                                            \
  case LPR_:                                                                                      \
    hipLaunchKernelGGL(pool_weighted_vec4_kernel<LPR_>,                                           \
                       dim3(grid_for(buckets, kBlock / LPR_, 8192)), dim3(kBlock), 0, s, buckets, \
                       combiner, (const long long*)row_offset, value_index, weights, table, out); \
    done = true;                                                                                  \
    break;
  if (v4) {
    switch (vec_size / 4) {
      HCTR_PW(1) HCTR_PW(2) HCTR_PW(4) HCTR_PW(8) HCTR_PW(16) HCTR_PW(32) HCTR_PW(64)
      default: break;
    }
  }
#undef HCTR_PW
  if (!done)
    hipLaunchKernelGGL(pool_weighted_kernel, dim3(grid_for(buckets * 64, kBlock, 8192)),
                       dim3(kBlock), 0, s, buckets, vec_size, combiner,
                       (const long long*)row_offset, value_index, weights, table, out);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_expand_key_grads(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                          const float* weights, const float* top_grad, float* key_grads,
                          hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  if (buckets == 0) return HCTR_OK;
  HCTR_REQUIRE(row_offset && top_grad && key_grads, "null pointer");
  hipStream_t s = as_stream(stream);
  const bool v4 = vec_size % 4 == 0 && reinterpret_cast<uintptr_t>(top_grad) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(key_grads) % 16 == 0;
  bool done = false;
#define HCTR_EG(LPR_)                                                                            \
  case LPR_:                                                                                      \
    hipLaunchKernelGGL(expand_key_grads_vec4_kernel<LPR_>,                                        \
                       dim3(grid_for(buckets, kBlock / LPR_, 8192)), dim3(kBlock), 0, s, buckets, \
                       combiner, (const long long*)row_offset, weights, top_grad, key_grads);     \
    done = true;                                                                                  \
    break;
  if (v4) {
    switch (vec_size / 4) {
      HCTR_EG(1) HCTR_EG(2) HCTR_EG(4) HCTR_EG(8) HCTR_EG(16) HCTR_EG(32) HCTR_EG(64)
      default: break;
    }
  }
#undef HCTR_EG
  if (!done)
    hipLaunchKernelGGL(expand_key_grads_kernel, dim3(grid_for(buckets * 64, kBlock, 8192)),
                       dim3(kBlock), 0, s, buckets, vec_size, combiner,
                       (const long long*)row_offset, weights, top_grad, key_grads);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_forward_pool_multihot(size_t buckets, int vec_size, int combiner, const void* row_offset,
                               int key_type, const uint64_t* value_index, const float* table,
                               void* out, int out_dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  HCTR_REQUIRE(buckets == 0 || (row_offset && value_index && table && out), "null pointer");
  return forward_pool_dispatch(buckets, vec_size, combiner, row_offset, key_type, value_index,
                               table, out, out_dtype, true, as_stream(stream), nullptr);
}

int hctr_forward_pool_mapped(size_t buckets, int vec_size, int combiner, const void* row_offset,
                             int key_type, const uint64_t* value_index, const float* table,
                             void* out, int out_dtype, int multi_hot, size_t samples,
                             size_t lookups, const uint32_t* one_hot, hctr_stream_t stream) {
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  HCTR_REQUIRE(buckets == 0 || (row_offset && value_index && table && out), "null pointer");
  HCTR_REQUIRE((samples == 0 && lookups == 0) || (samples > 0 && lookups > 0 &&
                                                  samples <= 0xFFFFFFFFull &&
                                                  lookups <= 0xFFFFFFFFull),
               "samples / lookups");
  return forward_pool_dispatch(buckets, vec_size, combiner, row_offset, key_type, value_index,
                               table, out, out_dtype, multi_hot != 0, as_stream(stream), one_hot,
                               (uint32_t)samples, (uint32_t)lookups);
}

int hctr_forward_reorder(size_t batch_per_gpu, int slot_num, int vec_size, int gpu_num,
                         const void* in, void* out, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(gpu_num > 0 && slot_num >= 0 && vec_size > 0, "shape");
  return launch_reorder<true>(batch_per_gpu, slot_num, vec_size, gpu_num, in, out, dtype,
                              as_stream(stream));
}

int hctr_backward_reorder(size_t batch_per_gpu, int slot_num, int vec_size, int gpu_num,
                          const void* in, void* out, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(gpu_num > 0 && slot_num >= 0 && vec_size > 0, "shape");
  return launch_reorder<false>(batch_per_gpu, slot_num, vec_size, gpu_num, in, out, dtype,
                               as_stream(stream));
}

}  // extern "C"
