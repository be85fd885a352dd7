// unique_exchange.hip -- "unique-row" form of the localized embedding exchange (multi-GPU, one-hot).
//
// The reference returns one pooled vector per (sample, slot) through the all-to-all
// (R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264) and sends one gradient per
// (sample, slot) back.  With one key per bucket the pooled vector IS the table row, and power-law
// keys repeat rows heavily (Criteo-1TB shape, alpha = 1.1: 227 k distinct rows for 1.7 M keys), so
// on xGMI -- 7 point-to-point links, the all-to-all is the slowest stage of a weak-scaled step --
// it pays to ship every distinct row ONCE per destination GPU plus a 8-byte (row id, bucket) pair
// per position, and to return the per-row SUM of the destination's gradients instead of one
// gradient per sample (HugeCTR's hybrid embedding exploits the same skew with a frequent/
// infrequent split, R/HugeCTR/src/embeddings/hybrid_sparse_embedding.cu).
//
// Owner rank (holds the slots):   positions p = (b_global, s_local) of its pooled layout
//   key(p) = peer(p) << rowbits | row(p)           peer = b_global / batch_per_gpu
//   stable radix sort -> per peer a run per distinct row; u = run index inside the peer segment
//   meta[q] = (u, bucket on the receiver = b_local * S + s_global), rows[peer_off[j] + u]
// Receiver: E[bucket] = rows[u] (expand), backward: sum of dE over each run (the sorted list is
// exactly what the segmented-reduce kernels of sparse_update.hip consume), sums travel back, the
// owner runs its normal sparse update on (row, summed gradient) entries.
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"
#include "radix_sort.h"
#include "scan.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock)
    uniq_keys_kernel(size_t P, size_t ppp, const uint64_t* __restrict__ vi, int rowbits,
                     uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < P;
       p += (size_t)gridDim.x * kBlock) {
    keys[p] = ((uint32_t)(p / ppp) << rowbits) | (uint32_t)vi[p];
    vals[p] = (uint32_t)p;
  }
}

__global__ void __launch_bounds__(kBlock)
    uniq_flags_kernel(size_t P, const uint32_t* __restrict__ keys, uint32_t* __restrict__ flags) {
  for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < P;
       q += (size_t)gridDim.x * kBlock)
    flags[q] = (q == 0 || keys[q] != keys[q - 1]) ? 1u : 0u;
}

// gid_ex = exclusive scan of flags: run id of q = gid_ex[q] + flags[q] - 1
__global__ void __launch_bounds__(kBlock)
    uniq_emit_kernel(size_t P, size_t ppp, int bl, int s_local, int s_total, int rank, int world,
                     int rowbits, const uint32_t* __restrict__ keys,
                     const uint32_t* __restrict__ vals, const uint32_t* __restrict__ flags,
                     const uint32_t* __restrict__ gid_ex, uint32_t* __restrict__ meta,
                     uint64_t* __restrict__ urow, long long* __restrict__ peer_off) {
  const uint32_t rowmask = rowbits >= 32 ? 0xFFFFFFFFu : ((1u << rowbits) - 1u);
  for (size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x; q < P;
       q += (size_t)gridDim.x * kBlock) {
    const uint32_t gid = gid_ex[q] + flags[q] - 1u;
    const size_t peer = q / ppp;  // a peer's segment is contiguous after the sort
    const uint32_t first = gid_ex[peer * ppp];  // a new peer always starts a new run
    const uint32_t p = vals[q];
    const uint32_t b_global = p / (uint32_t)s_local, sl = p % (uint32_t)s_local;
    const uint32_t b_local = b_global - (uint32_t)peer * (uint32_t)bl;
    const uint32_t s_global = sl * (uint32_t)world + (uint32_t)rank;
    meta[2 * q] = gid - first;
    meta[2 * q + 1] = b_local * (uint32_t)s_total + s_global;
    if (flags[q]) urow[gid] = (uint64_t)(keys[q] & rowmask);
    if (q % ppp == 0) peer_off[peer] = (long long)first;
    if (q == P - 1) peer_off[world] = (long long)gid + 1;
  }
}

// receiver: a group of `row16` lanes (16 B each) copies one row; the owner of position q is found
// against the (at most 64) segment starts held in registers; also emits the globally numbered
// sorted (row, bucket) list for the backward reduce
template <int ROW16>
__global__ void __launch_bounds__(kBlock)
    uniq_expand_kernel(size_t Q, int n_owners, const long long* __restrict__ q_off,
                       const long long* __restrict__ r_off, const uint32_t* __restrict__ meta,
                       const uint4* __restrict__ rows, uint4* __restrict__ out,
                       uint32_t* __restrict__ sorted_rows, uint32_t* __restrict__ sorted_buckets,
                       uint32_t* __restrict__ row_of) {
  constexpr int GPB = kBlock / ROW16;
  __shared__ long long s_q[65], s_r[65];
  for (int i = threadIdx.x; i <= n_owners; i += kBlock) {
    s_q[i] = q_off[i];
    s_r[i] = r_off[i];
  }
  __syncthreads();
  const int g = threadIdx.x / ROW16, c = threadIdx.x % ROW16;
  constexpr int U = 4;  // positions in flight per group
  for (size_t q0 = ((size_t)blockIdx.x * GPB + g) * U; q0 < Q;
       q0 += (size_t)gridDim.x * GPB * U) {
    uint32_t u[U], bkt[U];
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const size_t q = q0 + k < Q ? q0 + k : Q - 1;
      const uint2 m = *reinterpret_cast<const uint2*>(meta + 2 * q);
      int j = 0;
      while (j + 1 < n_owners && (long long)q >= s_q[j + 1]) j++;
      u[k] = m.x + (uint32_t)s_r[j];
      bkt[k] = m.y;
    }
    if (out != nullptr) {
#pragma unroll
      for (int k = 0; k < U; k++) v[k] = rows[(size_t)u[k] * ROW16 + c];
    }
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (q0 + k < Q) {
        if (out != nullptr) out[(size_t)bkt[k] * ROW16 + c] = v[k];
        if (c == 0) {
          sorted_rows[q0 + k] = u[k];
          sorted_buckets[q0 + k] = bkt[k];
          if (row_of != nullptr) row_of[bkt[k]] = u[k];  // bucket -> row of the received table
        }
      }
    }
  }
}

// urow-indexed gather of table rows into the send buffer (fp32 table -> out dtype)
template <typename OutT>
__global__ void __launch_bounds__(kBlock)
    uniq_gather_kernel(size_t U, int D, const uint64_t* __restrict__ urow,
                       const float* __restrict__ table, OutT* __restrict__ out) {
  const int d4 = D / 4;
  const size_t total = U * (size_t)d4;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t u = i / d4;
    const int c = (int)(i % d4);
    const float4 v = *reinterpret_cast<const float4*>(table + urow[u] * (uint64_t)D + c * 4);
    OutT* o = out + u * (size_t)D + c * 4;
    if constexpr (sizeof(OutT) == 4) {
      *reinterpret_cast<float4*>(o) = v;
    } else {
      OutT t[4] = {(OutT)v.x, (OutT)v.y, (OutT)v.z, (OutT)v.w};
      *reinterpret_cast<uint2*>(o) = *reinterpret_cast<uint2*>(t);
    }
  }
}

}  // namespace
}  // namespace hctr

using namespace hctr;

struct hctr_uniq {
  size_t cap = 0;
  uint32_t *keys_in = nullptr, *keys_out = nullptr, *vals_in = nullptr, *vals_out = nullptr;
  uint32_t *flags = nullptr, *gid = nullptr;
  void* sort_temp = nullptr;
  size_t temp_bytes = 0;
  unsigned long long* tile_sums = nullptr;
  unsigned long long* d_total = nullptr;
};

extern "C" {

int hctr_uniq_create(size_t max_positions, hctr_uniq** out) {
  HCTR_REQUIRE(out && max_positions > 0 && max_positions < 0xFFFFFFF0ull, "max_positions");
  hctr_uniq* u = new hctr_uniq();
  u->cap = max_positions;
  const size_t n = max_positions;
  bool ok = hipMalloc(&u->keys_in, n * 4) == hipSuccess &&
            hipMalloc(&u->keys_out, n * 4) == hipSuccess &&
            hipMalloc(&u->vals_in, n * 4) == hipSuccess &&
            hipMalloc(&u->vals_out, n * 4) == hipSuccess &&
            hipMalloc(&u->flags, n * 4) == hipSuccess &&
            hipMalloc(&u->gid, (n + 1) * 4) == hipSuccess &&
            hipMalloc(&u->tile_sums, (n / 1024 + 2) * 8) == hipSuccess &&
            hipMalloc(&u->d_total, 8) == hipSuccess;
  u->temp_bytes = radix_sort_temp_bytes(n);
  if (ok) ok = hipMalloc(&u->sort_temp, u->temp_bytes) == hipSuccess;
  if (!ok) {
    set_error("hctr_uniq_create: allocation failed");
    void* ptrs[] = {u->keys_in, u->keys_out, u->vals_in, u->vals_out, u->flags,
                    u->gid,     u->tile_sums, u->d_total, u->sort_temp};
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    delete u;
    return HCTR_ERR_HIP;
  }
  *out = u;
  return HCTR_OK;
}

int hctr_uniq_destroy(hctr_uniq* u) {
  if (!u) return HCTR_OK;
  (void)hipDeviceSynchronize();
  void* ptrs[] = {u->keys_in, u->keys_out, u->vals_in, u->vals_out, u->flags,
                  u->gid,     u->tile_sums, u->d_total, u->sort_temp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete u;
  return HCTR_OK;
}

int hctr_uniq_plan(hctr_uniq* u, size_t positions, size_t positions_per_peer, int batch_per_gpu,
                   int slots_local, int slots_total, int rank, int world,
                   const uint64_t* value_index, uint64_t max_rows, uint32_t* meta, uint64_t* urow,
                   int64_t* peer_off, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  HCTR_REQUIRE(positions > 0 && positions <= u->cap, "positions exceed the plan's capacity");
  HCTR_REQUIRE(world > 0 && positions == positions_per_peer * (size_t)world &&
                   positions_per_peer == (size_t)batch_per_gpu * (size_t)slots_local,
               "positions must be world * batch_per_gpu * slots_local");
  HCTR_REQUIRE(value_index && meta && urow && peer_off, "null pointer");
  int rowbits = 1;
  while (rowbits < 32 && ((uint64_t)1 << rowbits) < max_rows) rowbits++;
  int peerbits = 0;
  while ((1 << peerbits) < world) peerbits++;
  HCTR_REQUIRE(rowbits + peerbits <= 32, "rows x peers do not fit the 32-bit sort key");
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(positions, kBlock, 4096);
  hipLaunchKernelGGL(uniq_keys_kernel, dim3(grid), dim3(kBlock), 0, s, positions,
                     positions_per_peer, value_index, rowbits, u->keys_in, u->vals_in);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(radix_sort_pairs_u32(u->sort_temp, u->temp_bytes, u->keys_in, u->keys_out, u->vals_in,
                                u->vals_out, positions, rowbits + peerbits, s));
  hipLaunchKernelGGL(uniq_flags_kernel, dim3(grid), dim3(kBlock), 0, s, positions, u->keys_out,
                     u->flags);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(exclusive_scan_to_offsets<uint32_t>(u->flags, positions, u->tile_sums, u->d_total,
                                               u->gid, s));
  hipLaunchKernelGGL(uniq_emit_kernel, dim3(grid), dim3(kBlock), 0, s, positions,
                     positions_per_peer, batch_per_gpu, slots_local, slots_total, rank, world,
                     rowbits, u->keys_out, u->vals_out, u->flags, u->gid, meta, urow,
                     (long long*)peer_off);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_uniq_gather_rows(size_t n_rows, int vec_size, const uint64_t* urow, const float* table,
                          void* out, int out_dtype, hctr_stream_t stream) {
  if (n_rows == 0) return HCTR_OK;
  HCTR_REQUIRE(urow && table && out && vec_size > 0 && vec_size % 4 == 0, "arguments");
  hipStream_t s = as_stream(stream);
  const dim3 grid(grid_for(n_rows * (size_t)(vec_size / 4), kBlock, 8192));
  switch (out_dtype) {
    case HCTR_EMB_F32:
      hipLaunchKernelGGL(uniq_gather_kernel<float>, grid, dim3(kBlock), 0, s, n_rows, vec_size,
                         urow, table, (float*)out);
      break;
    case HCTR_EMB_F16:
      hipLaunchKernelGGL(uniq_gather_kernel<_Float16>, grid, dim3(kBlock), 0, s, n_rows, vec_size,
                         urow, table, (_Float16*)out);
      break;
    case HCTR_EMB_BF16:
      hipLaunchKernelGGL(uniq_gather_kernel<__bf16>, grid, dim3(kBlock), 0, s, n_rows, vec_size,
                         urow, table, (__bf16*)out);
      break;
    default:
      HCTR_REQUIRE(false, "out_dtype");
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_uniq_expand(size_t positions, int n_owners, const int64_t* q_off, const int64_t* r_off,
                     const uint32_t* meta, const void* rows, int vec_size, int dtype, void* out,
                     uint32_t* sorted_rows, uint32_t* sorted_buckets, uint32_t* row_of,
                     hctr_stream_t stream) {
  if (positions == 0) return HCTR_OK;
  HCTR_REQUIRE(q_off && r_off && meta && rows && sorted_rows && sorted_buckets, "null pointer");
  HCTR_REQUIRE(out || row_of, "expand needs an output: the expanded tensor and / or row_of");
  const size_t row_bytes = (size_t)vec_size * (dtype == HCTR_EMB_F32 ? 4 : 2);
  HCTR_REQUIRE(row_bytes % 16 == 0, "row bytes must be a multiple of 16");
  HCTR_REQUIRE(reinterpret_cast<uintptr_t>(rows) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0,
               "rows / out must be 16-byte aligned");
  const int row16 = (int)(row_bytes / 16);
  HCTR_REQUIRE(n_owners >= 1 && n_owners <= 64, "1..64 owners");
  hipStream_t s = as_stream(stream);
#define HCTR_EXPAND_CASE(R16)                                                                     \
  case R16:                                                                                       \
    hipLaunchKernelGGL(uniq_expand_kernel<R16>,                                                   \
                       dim3(grid_for(ceil_div<size_t>(positions, 4), kBlock / R16, 8192)),        \
                       dim3(kBlock), 0, s, positions, n_owners, (const long long*)q_off,          \
                       (const long long*)r_off, meta, (const uint4*)rows, (uint4*)out,            \
                       sorted_rows, sorted_buckets, row_of);                                      \
    break;
  switch (row16) {
    HCTR_EXPAND_CASE(1)
    HCTR_EXPAND_CASE(2)
    HCTR_EXPAND_CASE(4)
    HCTR_EXPAND_CASE(8)
    HCTR_EXPAND_CASE(16)
    HCTR_EXPAND_CASE(32)
    HCTR_EXPAND_CASE(64)
    default:
      HCTR_REQUIRE(false, "row size: 16 B x a power of two up to 1 KiB");
  }
#undef HCTR_EXPAND_CASE
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // extern "C"
