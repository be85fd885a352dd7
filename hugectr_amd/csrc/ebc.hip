// ebc.hip -- embedding_collection (EBC / SparseOperationKit) path on static tables:
// key routing -> keys_to_indices -> pooled lookup -> (all-to-all) -> network forward/backward.
//
// Reference pieces restated MI355X-first:
//   key routing      R/HugeCTR/embedding/data_distributor/key_filtering_operators.cu:37-300
//                    (owner of a key: shard `key % num_shards`, :86-88)
//   keys_to_indices  R/HugeCTR/embedding/operators/keys_to_indices.cu:24-43
//                    (idx = table_start + key / num_shards)
//   pooled lookup    R/HugeCTR/embedding/operators/generic_lookup.cuh:318-416 -> hctr_forward_pool
//   network forward  R/HugeCTR/embedding/operators/network_forward.cu (sum row-shard partials,
//                    Average divides by the bucket's TOTAL key count on the receiver)
// The reference routes keys with two all-to-alls after a host-synchronised count exchange
// (sparse_data_distribution_op_impl.cu:303-333).  SparseOperationKit instead all-gathers the keys
// and lets every rank pick its own (sparse_operation_kit/lookup.py:490-495); that is the flow
// implemented here: one fused filter+index pass over the gathered CSR, no host sync.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "block_prims.h"
#include "common.h"
#include "radix_sort.h"
#include "scan.h"
#include "sparse_update.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;

struct LookupDesc {  // one per lookup resolved on this rank
  int global_lookup;    // index into the global feature-major bucket_range
  int num_shards;       // row shards of the lookup's table
  int shard_id;         // this rank's shard of that table
  long long row_start;  // first row of the table's shard in the rank's flat table
};

__device__ __forceinline__ LookupDesc load_desc(const int* __restrict__ d3,
                                                const long long* __restrict__ rs, int ll) {
  LookupDesc d;
  d.global_lookup = d3[3 * ll];
  d.num_shards = d3[3 * ll + 1];
  d.shard_id = d3[3 * ll + 2];
  d.row_start = rs[ll];
  return d;
}

// output bucket order = [peer][local lookup][b_local] so that the pooled vectors are already the
// all-to-all send buffer
__device__ __forceinline__ void decode_bucket(size_t ob, int n_local, size_t bpg, int& peer,
                                              int& ll, size_t& bl) {
  bl = ob % bpg;
  const size_t t = ob / bpg;
  ll = (int)(t % n_local);
  peer = (int)(t / n_local);
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ebc_count_kernel(size_t batch, size_t bpg, int n_local, const int* __restrict__ d3,
                     const long long* __restrict__ rs, const K* __restrict__ keys, const K* __restrict__ bucket_range,
                     long long* __restrict__ lens) {
  const size_t total = (size_t)n_local * batch;
  for (size_t ob = (size_t)blockIdx.x * kBlock + threadIdx.x; ob < total;
       ob += (size_t)gridDim.x * kBlock) {
    int peer, ll;
    size_t bl;
    decode_bucket(ob, n_local, bpg, peer, ll, bl);
    const LookupDesc d = load_desc(d3, rs, ll);
    const size_t sb = (size_t)d.global_lookup * batch + (size_t)peer * bpg + bl;
    long long c = 0;
    for (size_t q = (size_t)bucket_range[sb]; q < (size_t)bucket_range[sb + 1]; q++)
      c += ((long long)keys[q] % d.num_shards == d.shard_id) ? 1 : 0;
    lens[ob] = c;
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ebc_index_kernel(size_t batch, size_t bpg, int n_local, const int* __restrict__ d3,
                     const long long* __restrict__ rs, const K* __restrict__ keys, const K* __restrict__ bucket_range,
                     const long long* __restrict__ out_range, uint64_t* __restrict__ out_idx) {
  const size_t total = (size_t)n_local * batch;
  for (size_t ob = (size_t)blockIdx.x * kBlock + threadIdx.x; ob < total;
       ob += (size_t)gridDim.x * kBlock) {
    int peer, ll;
    size_t bl;
    decode_bucket(ob, n_local, bpg, peer, ll, bl);
    const LookupDesc d = load_desc(d3, rs, ll);
    const size_t sb = (size_t)d.global_lookup * batch + (size_t)peer * bpg + bl;
    size_t dst = (size_t)out_range[ob];
    for (size_t q = (size_t)bucket_range[sb]; q < (size_t)bucket_range[sb + 1]; q++) {
      const long long k = (long long)keys[q];
      if (k % d.num_shards == d.shard_id)
        // keys_to_indices.cu:31-42; row_start < 0: dynamic table, the key itself is kept
        out_idx[dst++] = d.row_start < 0 ? (uint64_t)k : (uint64_t)(d.row_start + k / d.num_shards);
    }
  }
}

// One GPU owning every lookup whole (num_shards == 1, local lookup order == global order): the
// owner's buckets ARE the input buckets, so count + scan + index collapse into one pass --
// out_range = bucket_range (as int64), row = row_start(lookup) + key.  Also raises the one-hot
// flag the pooling kernel's offset-free loop keys on (every bucket exactly its own key).
template <typename K>
__global__ void __launch_bounds__(kBlock)
    ebc_route_whole_kernel(size_t total, size_t batch, const long long* __restrict__ rs,
                           const K* __restrict__ keys, const K* __restrict__ bucket_range,
                           long long* __restrict__ out_range, uint64_t* __restrict__ out_idx,
                           unsigned long long* __restrict__ d_nnz,
                           uint32_t* __restrict__ one_hot) {
  bool ragged = false;
  for (size_t ob = (size_t)blockIdx.x * kBlock + threadIdx.x; ob < total && blockIdx.y == 0;
       ob += (size_t)gridDim.x * kBlock) {
    const size_t b0 = (size_t)bucket_range[ob], b1 = (size_t)bucket_range[ob + 1];
    out_range[ob] = (long long)b0;
    if (ob == total - 1) {
      out_range[total] = (long long)b1;
      if (d_nnz) *d_nnz = (unsigned long long)b1;
    }
    ragged |= (b0 != ob) | (b1 != ob + 1);
  }
  // keys: lanes stride the keys of 64 buckets (block_prims.h), coalesced for any hotness
  for_each_key_wave(total, bucket_range, [&](size_t ob, size_t q) {
    const long long r0 = rs[ob / batch];
    const long long k = (long long)keys[q];
    out_idx[q] = r0 < 0 ? (uint64_t)k : (uint64_t)(r0 + k);
  });
  if (one_hot && ragged) *one_hot = 0u;
}

// ---- LocalReduceIndexCalculation + LocalReduce (R/HugeCTR/embedding/operators/
//      index_calculation.cu, model_backward.cu:113-...): sort the keys of the local lookups by
//      row id, find the unique ones, sum the gradients of every unique row (ascending position).
__global__ void __launch_bounds__(kBlock)
    lr_expand_kernel(size_t buckets, const long long* __restrict__ bucket_range,
                     uint32_t* __restrict__ pos_bucket, uint32_t* __restrict__ pos_iota,
                     uint32_t map_inner, uint32_t map_outer, const uint64_t* __restrict__ row_ids,
                     uint32_t* __restrict__ rows32) {
  // key-parallel (block_prims.h); the bucket is recorded as its gradient row
  // (SparseUpdater::map_inner: batch-major output of a one-GPU collection)
  for_each_key_wave(buckets, bucket_range, [&](size_t b, size_t q) {
    pos_bucket[q] = map_inner ? ((uint32_t)b % map_inner) * map_outer + (uint32_t)b / map_inner
                              : (uint32_t)b;
    pos_iota[q] = (uint32_t)q;
    if (rows32) rows32[q] = (uint32_t)row_ids[q];  // 32-bit sort keys (row ids < 2^32)
  });
}

template <typename RowT>
__global__ void __launch_bounds__(kBlock)
    lr_flags_kernel(size_t n, const RowT* __restrict__ sorted, uint32_t* __restrict__ flags) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock)
    flags[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}

// heads_before[i] = number of run heads in [0, i): compact id of position i = heads_before[i] +
// flag[i] - 1
template <typename RowT>
__global__ void __launch_bounds__(kBlock)
    lr_emit_kernel(size_t n, const RowT* __restrict__ sorted, const uint32_t* __restrict__ pos,
                   const uint32_t* __restrict__ flags, const uint32_t* __restrict__ heads_before,
                   const uint32_t* __restrict__ pos_bucket, const uint64_t* __restrict__ src_keys,
                   uint32_t* __restrict__ sorted_cid, uint32_t* __restrict__ sorted_bucket,
                   uint64_t* __restrict__ unique_rows, uint64_t* __restrict__ unique_keys,
                   uint64_t* __restrict__ d_num_unique) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const uint32_t f = flags[i];
    const uint32_t cid = heads_before[i] + f - 1u;
    const uint32_t p = pos[i];
    sorted_cid[i] = cid;
    sorted_bucket[i] = pos_bucket[p];
    if (f) {
      unique_rows[cid] = (uint64_t)sorted[i];
      if (unique_keys) unique_keys[cid] = src_keys[p];
    }
    if (i == n - 1) *d_num_unique = (uint64_t)cid + 1;
  }
}

// keys that arrived through the key-route all-to-all, already in the owner's bucket order
// [source peer][local lookup][b_local]: key -> row of the rank's flat table, in place
__global__ void __launch_bounds__(kBlock)
    ebc_routed_keys_to_indices_kernel(size_t nb, size_t bpg, int n_local,
                                      const int* __restrict__ d3, const long long* __restrict__ rs,
                                      const long long* __restrict__ out_range,
                                      uint64_t* __restrict__ keys) {
  for (size_t ob = (size_t)blockIdx.x * kBlock + threadIdx.x; ob < nb;
       ob += (size_t)gridDim.x * kBlock) {
    const int ll = (int)((ob / bpg) % (size_t)n_local);
    const LookupDesc d = load_desc(d3, rs, ll);
    if (d.row_start < 0) continue;  // dynamic table: the key itself is the handle
    for (long long q = out_range[ob]; q < out_range[ob + 1]; q++)
      keys[q] = (uint64_t)(d.row_start + (long long)keys[q] / d.num_shards);
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    keys_to_indices_kernel(size_t n, const K* __restrict__ keys, long long table_start,
                           int num_shards, uint64_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock)
    out[i] = (uint64_t)(table_start + (long long)keys[i] / num_shards);
}

// recv / send blocks: block t holds [bpg][ev] vectors of (source rank, its local lookup).
// src_blocks[l * max_shards + s] = block index of shard s of global lookup l, or -1.
template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_as_f32<__half>(const __half* p) { return __half2float(*p); }
template <>
__device__ __forceinline__ float ld_as_f32<__hip_bfloat16>(const __hip_bfloat16* p) {
  return __bfloat162float(*p);
}
template <typename T>
__device__ __forceinline__ void st_from_f32(T* p, float v);
template <>
__device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_from_f32<__half>(__half* p, float v) { *p = __float2half_rn(v); }
template <>
__device__ __forceinline__ void st_from_f32<__hip_bfloat16>(__hip_bfloat16* p, float v) {
  *p = __float2bfloat16(v);
}

template <typename T, bool FWD>
__global__ void __launch_bounds__(kBlock)
    ebc_network_kernel(size_t bpg, int num_lookup, int ev, int max_shards,
                       const int* __restrict__ src_blocks, const int* __restrict__ combiner,
                       const long long* __restrict__ bucket_counts, int batch_major,
                       const T* __restrict__ in, T* __restrict__ out) {
  // FWD: out[l][b][:] = (sum_s recv[block(l,s)][b][:]) / count   (count only for Average)
  // BWD: send[block(l,s)][b][:] = grad[l][b][:] / count            for every shard s of l
  // A DIVISION by the count, as the reference's kernels (accum /= average_pooling_factor,
  // generic_lookup.cuh:336-341, 733-738) and its CPU code (v /= (end - start)) do it: a product
  // with the reciprocal differs in the last bit for most counts
  const size_t total = (size_t)num_lookup * bpg * ev;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const int e = (int)(i % ev);
    const size_t lb = i / ev;
    const size_t b = lb % bpg;
    const int l = (int)(lb / bpg);
    float div = 1.0f;
    if (combiner[l] == 1) {
      const long long c = bucket_counts[(size_t)l * bpg + b];
      if (c > 0) div = (float)c;
    }
    const size_t dense_idx = batch_major ? (b * (size_t)num_lookup + l) * ev + e
                                         : ((size_t)l * bpg + b) * ev + e;
    if (FWD) {
      float acc = 0.f;
      for (int s = 0; s < max_shards; s++) {
        const int blk = src_blocks[l * max_shards + s];
        if (blk >= 0) acc += ld_as_f32<T>(in + ((size_t)blk * bpg + b) * ev + e);
      }
      st_from_f32<T>(out + dense_idx, acc / div);
    } else {
      const float g = ld_as_f32<T>(in + dense_idx) / div;
      for (int s = 0; s < max_shards; s++) {
        const int blk = src_blocks[l * max_shards + s];
        if (blk >= 0) st_from_f32<T>(out + ((size_t)blk * bpg + b) * ev + e, g);
      }
    }
  }
}

// One GPU, Average lookups: network_forward's receiver arithmetic applied IN PLACE to the pooled
// sums the gather stored straight into the output (and network_backward's to the output's gradient):
// the sum already rounded to the vector type, divided by the keys of the bucket, rounded again
// (R/HugeCTR/embedding/operators/network_forward.cu:272-292, network_backward.cu).  Only the
// vectors of Average lookups with more than one key are touched; the same bits as
// ebc_network_vec_kernel leaves (which moves a vector of a single shard with scale 1 untouched).
template <typename T, bool FWD>
__global__ void __launch_bounds__(kBlock)
    ebc_scale_average_kernel(size_t bpg, int num_lookup, int ev, const int* __restrict__ combiner,
                             const long long* __restrict__ bucket_counts, int batch_major,
                             T* __restrict__ data) {
  const size_t total = (size_t)num_lookup * bpg * ev;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const int e = (int)(i % ev);
    const size_t lb = i / ev;
    const size_t b = lb % bpg;
    const int l = (int)(lb / bpg);
    if (combiner[l] != 1) continue;
    const long long c = bucket_counts[(size_t)l * bpg + b];
    if (c <= 1) continue;  // scale 1: the vector moves as it is
    const float div = (float)c;
    T* p = data + (batch_major ? (b * (size_t)num_lookup + l) * ev + e
                               : ((size_t)l * bpg + b) * ev + e);
    const float v = ld_as_f32<T>(p);
    st_from_f32<T>(p, FWD ? (0.f + v) / div : v / div);
  }
}

// 16-byte vector form (ev * sizeof(T) % 16 == 0, 16-byte aligned buffers): one thread moves
// 16 B of one (lookup, sample) vector; arithmetic only where a scale or a shard sum is needed
template <typename T>
struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
};

template <typename T, bool FWD>
__global__ void __launch_bounds__(kBlock)
    ebc_network_vec_kernel(size_t bpg, int num_lookup, int row16, int max_shards,
                           const int* __restrict__ src_blocks, const int* __restrict__ combiner,
                           const long long* __restrict__ bucket_counts, int batch_major,
                           const uint4* __restrict__ in, uint4* __restrict__ out) {
  constexpr int N = Vec16<T>::N;
  const size_t total = (size_t)num_lookup * bpg * row16;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t lb = i / (uint32_t)row16;
    const uint32_t c = (uint32_t)(i - lb * (uint32_t)row16);
    const uint32_t l = (uint32_t)(lb / bpg);
    const size_t b = lb - (size_t)l * bpg;
    float div = 1.0f;
    if (combiner[l] == 1) {
      const long long cnt = bucket_counts[(size_t)l * bpg + b];
      if (cnt > 0) div = (float)cnt;
    }
    const size_t dense_idx =
        (batch_major ? (b * (size_t)num_lookup + l) : ((size_t)l * bpg + b)) * row16 + c;
    const int* blocks = src_blocks + (size_t)l * max_shards;
    if (FWD) {
      int first = -1, n = 0;
      for (int s = 0; s < max_shards; s++)
        if (blocks[s] >= 0) {
          if (first < 0) first = blocks[s];
          n++;
        }
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (n == 1 && div == 1.0f) {
        v = in[((size_t)first * bpg + b) * row16 + c];  // plain move
      } else if (n >= 1) {
        float acc[N];
#pragma unroll
        for (int k = 0; k < N; k++) acc[k] = 0.f;
        for (int s = 0; s < max_shards; s++) {
          if (blocks[s] < 0) continue;
          const uint4 r = in[((size_t)blocks[s] * bpg + b) * row16 + c];
          const T* rt = reinterpret_cast<const T*>(&r);
#pragma unroll
          for (int k = 0; k < N; k++) acc[k] += ld_as_f32<T>(rt + k);
        }
        T* vt = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int k = 0; k < N; k++) st_from_f32<T>(vt + k, acc[k] / div);
      }
      out[dense_idx] = v;
    } else {
      uint4 v = in[dense_idx];
      if (div != 1.0f) {
        T* vt = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int k = 0; k < N; k++) st_from_f32<T>(vt + k, ld_as_f32<T>(vt + k) / div);
      }
      for (int s = 0; s < max_shards; s++)
        if (blocks[s] >= 0) out[((size_t)blocks[s] * bpg + b) * row16 + c] = v;
    }
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ebc_bucket_count_kernel(size_t batch, size_t bpg, int num_lookup, int rank,
                            const K* __restrict__ bucket_range, long long* __restrict__ counts) {
  // total key count of every (lookup, local sample) bucket of THIS rank's sample slice
  const size_t total = (size_t)num_lookup * bpg;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t l = i / bpg, b = i % bpg;
    const size_t sb = l * batch + (size_t)rank * bpg + b;
    counts[i] = (long long)bucket_range[sb + 1] - (long long)bucket_range[sb];
  }
}

}  // namespace
}  // namespace hctr

using namespace hctr;

struct hctr_updater {
  SparseUpdater impl;
  float ftrl_lambda1 = 0.f, ftrl_lambda2 = 0.f, ftrl_beta = 0.f;
  // hctr_ebc_local_reduce scratch (allocated on first use, sized by impl.max_nnz)
  uint64_t* lr_sorted = nullptr;
  uint32_t *lr_pos_in = nullptr, *lr_pos_out = nullptr, *lr_pos_bucket = nullptr;
  uint32_t *lr_flags = nullptr, *lr_heads = nullptr, *lr_cid = nullptr, *lr_sbucket = nullptr;
  uint32_t* lr_rows32 = nullptr;  // row ids as 32-bit sort keys (own radix sort)
  void* lr_temp = nullptr;
  size_t lr_temp_bytes = 0;
  unsigned long long *lr_tile_sums = nullptr, *lr_total = nullptr;
  uint64_t* lr_num_unique = nullptr;

  int lr_alloc() {
    if (lr_sorted) return HCTR_OK;
    const size_t n = impl.max_nnz ? impl.max_nnz : 1;
    HCTR_HIP(hipMalloc(&lr_sorted, n * 8));
    HCTR_HIP(hipMalloc(&lr_pos_in, n * 4));
    HCTR_HIP(hipMalloc(&lr_pos_out, n * 4));
    HCTR_HIP(hipMalloc(&lr_pos_bucket, n * 4));
    HCTR_HIP(hipMalloc(&lr_flags, n * 4));
    HCTR_HIP(hipMalloc(&lr_heads, (n + 1) * 4));
    HCTR_HIP(hipMalloc(&lr_cid, n * 4));
    HCTR_HIP(hipMalloc(&lr_sbucket, n * 4));
    HCTR_HIP(hipMalloc(&lr_rows32, n * 4));
    HCTR_HIP(hipMalloc(&lr_tile_sums, (n / 1024 + 2) * 8));
    HCTR_HIP(hipMalloc(&lr_total, 8));
    HCTR_HIP(hipMalloc(&lr_num_unique, 8));
    size_t tb = 0;
    if (rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                  (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64, nullptr,
                                  false) != hipSuccess) {
      set_error("rocprim::radix_sort_pairs (size query) failed");
      return HCTR_ERR_HIP;
    }
    const size_t own = radix_sort_temp_bytes(n);
    lr_temp_bytes = tb > own ? tb : own;
    if (lr_temp_bytes == 0) lr_temp_bytes = 16;
    HCTR_HIP(hipMalloc(&lr_temp, lr_temp_bytes));
    return HCTR_OK;
  }
  void lr_free() {
    void* ptrs[] = {lr_sorted, lr_pos_in, lr_pos_out, lr_pos_bucket, lr_flags, lr_heads, lr_cid,
                    lr_sbucket, lr_temp, lr_tile_sums, lr_total, lr_num_unique, lr_rows32};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    lr_sorted = nullptr;
  }
};


// ILookup::lookup of the static (ragged) table: ragged_static_embedding_table_lookup_kernel,
// R/HugeCTR/embedding_storage/ragged_static_embedding.cu:33-51.  `keys` are the indices
// keys_to_indices produced (globally numbered over the group's tables); position tid belongs to
// the id space whose offset range holds it; the id space's slot in the table's own (ascending)
// id-space list gives its first index, its element offset in the flat fp32 array and its vector
// size.  One thread per key, two binary searches over a handful of entries, one pointer store.
template <typename K>
__global__ void __launch_bounds__(kBlock)
    static_lookup_kernel(const K* __restrict__ keys, size_t num_keys,
                         const uint32_t* __restrict__ id_space_offset, size_t num_id_space_offset,
                         const int* __restrict__ id_space_list,
                         const int* __restrict__ local_id_space_list, size_t num_local,
                         const uint64_t* __restrict__ table_index_start, float* emb_table,
                         const uint64_t* __restrict__ table_ev_offset,
                         const int* __restrict__ local_ev_size, float** __restrict__ emb_vec,
                         uint32_t* d_error) {
  for (size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x; tid < num_keys;
       tid += (size_t)gridDim.x * kBlock) {
    // last i with id_space_offset[i] <= tid  (bs_upper_bound_sub_one)
    size_t lo = 0, hi = num_id_space_offset;
    while (lo < hi) {
      const size_t mid = (lo + hi) >> 1;
      if ((size_t)id_space_offset[mid] <= tid) lo = mid + 1;
      else hi = mid;
    }
    const int id_space = id_space_list[lo - 1];
    size_t a = 0, b = num_local;
    while (a < b) {
      const size_t mid = (a + b) >> 1;
      if (local_id_space_list[mid] <= id_space) a = mid + 1;
      else b = mid;
    }
    if (a == 0 || local_id_space_list[a - 1] != id_space) {
      atomicOr(d_error, 1u);  // the table does not hold this id space
      emb_vec[tid] = nullptr;
      continue;
    }
    const size_t t = a - 1;
    const uint64_t idx = (uint64_t)keys[tid];
    const uint64_t rows = (table_index_start[t + 1] - table_index_start[t]);
    if (idx < table_index_start[t] || idx - table_index_start[t] >= rows) {
      atomicOr(d_error, 2u);  // index outside the table's shard
      emb_vec[tid] = nullptr;
      continue;
    }
    emb_vec[tid] = emb_table + table_ev_offset[t] + (idx - table_index_start[t]) * (uint64_t)local_ev_size[t];
  }
}

extern "C" {

int hctr_ebc_keys_to_indices(const void* keys, int key_type, size_t n, int64_t table_start,
                             int num_shards, uint64_t* out, hctr_stream_t stream) {
  HCTR_REQUIRE(num_shards >= 1, "num_shards");
  if (n == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && out, "null pointer");
  hipStream_t s = as_stream(stream);
  if (key_type == HCTR_KEY_U32)
    hipLaunchKernelGGL(keys_to_indices_kernel<uint32_t>, dim3(grid_for(n, kBlock)), dim3(kBlock), 0,
                       s, n, (const uint32_t*)keys, (long long)table_start, num_shards, out);
  else if (key_type == HCTR_KEY_I64)
    hipLaunchKernelGGL(keys_to_indices_kernel<long long>, dim3(grid_for(n, kBlock)), dim3(kBlock),
                       0, s, n, (const long long*)keys, (long long)table_start, num_shards, out);
  else
    HCTR_REQUIRE(false, "key_type");
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_static_lookup(const void* keys, int key_type, size_t num_keys,
                       const uint32_t* num_keys_per_table_offset, size_t num_table_offset,
                       const int32_t* table_id_list, const int32_t* local_table_ids,
                       size_t num_local_tables, const uint64_t* table_index_start,
                       float* emb_table, const uint64_t* table_ev_offset,
                       const int32_t* local_ev_sizes, float** embedding_vec, uint32_t* d_error,
                       hctr_stream_t stream) {
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && num_keys_per_table_offset && table_id_list && local_table_ids &&
                   table_index_start && emb_table && table_ev_offset && local_ev_sizes &&
                   embedding_vec && d_error,
               "null pointer");
  HCTR_REQUIRE(num_table_offset >= 2 && num_local_tables >= 1, "empty table lists");
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(num_keys, kBlock);
#define HCTR_SL(K)                                                                             \
  hipLaunchKernelGGL(static_lookup_kernel<K>, dim3(grid), dim3(kBlock), 0, s, (const K*)keys,   \
                     num_keys, num_keys_per_table_offset, num_table_offset, table_id_list,      \
                     local_table_ids, num_local_tables, table_index_start, emb_table,           \
                     table_ev_offset, local_ev_sizes, embedding_vec, d_error)
  if (key_type == HCTR_KEY_U32) HCTR_SL(uint32_t);
  else if (key_type == HCTR_KEY_I64) HCTR_SL(long long);
  else if (key_type == 2) HCTR_SL(uint64_t);  // the uint64 indices of hctr_ebc_keys_to_indices
  else HCTR_REQUIRE(false, "key_type");
#undef HCTR_SL
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_ebc_route_workspace_bytes(size_t batch, int num_local_lookups) {
  const size_t nb = (size_t)num_local_lookups * batch;
  return (nb + 1) * sizeof(long long) + (ceil_div<size_t>(nb + 1, 1024) + 2) * 8 + 64;
}

int hctr_ebc_route_keys(size_t batch, int world, int num_local_lookups, const int32_t* lookup_desc,
                        const int64_t* row_start, const void* keys, const void* bucket_range,
                        int key_type, int64_t* out_bucket_range, uint64_t* out_indices,
                        uint64_t* d_nnz, void* workspace, hctr_stream_t stream) {
  HCTR_REQUIRE(world >= 1 && batch % world == 0, "batch must be divisible by world");
  HCTR_REQUIRE(num_local_lookups >= 0, "num_local_lookups");
  hipStream_t s = as_stream(stream);
  const size_t nb = (size_t)num_local_lookups * batch;
  if (nb == 0) {
    HCTR_HIP(hipMemsetAsync(out_bucket_range, 0, sizeof(int64_t), s));
    if (d_nnz) HCTR_HIP(hipMemsetAsync(d_nnz, 0, sizeof(uint64_t), s));
    return HCTR_OK;
  }
  HCTR_REQUIRE(lookup_desc && row_start && bucket_range && out_bucket_range && out_indices &&
                   workspace,
               "null pointer");
  // workspace: [lens (nb+1) i64][tile_sums][d_total]
  long long* lens = (long long*)workspace;
  unsigned long long* tile_sums = (unsigned long long*)(lens + nb + 1);
  unsigned long long* d_total = tile_sums + ceil_div<size_t>(nb + 1, 1024) + 1;
  const int* d3 = lookup_desc;
  const long long* rs = (const long long*)row_start;
  const size_t bpg = batch / world;
  const int grid = grid_for(nb, kBlock);
  int rc = HCTR_OK;
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ebc_count_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, batch, bpg,
                       num_local_lookups, d3, rs, (const uint32_t*)keys,
                       (const uint32_t*)bucket_range, lens);
    rc = exclusive_scan_to_offsets<long long>(lens, nb, tile_sums, d_total,
                                              (long long*)out_bucket_range, s);
    if (rc == HCTR_OK)
      hipLaunchKernelGGL(ebc_index_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, batch, bpg,
                         num_local_lookups, d3, rs, (const uint32_t*)keys,
                         (const uint32_t*)bucket_range, (const long long*)out_bucket_range,
                         out_indices);
  } else if (key_type == HCTR_KEY_I64) {
    hipLaunchKernelGGL(ebc_count_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, batch, bpg,
                       num_local_lookups, d3, rs, (const long long*)keys,
                       (const long long*)bucket_range, lens);
    rc = exclusive_scan_to_offsets<long long>(lens, nb, tile_sums, d_total,
                                              (long long*)out_bucket_range, s);
    if (rc == HCTR_OK)
      hipLaunchKernelGGL(ebc_index_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, batch, bpg,
                         num_local_lookups, d3, rs, (const long long*)keys,
                         (const long long*)bucket_range, (const long long*)out_bucket_range,
                         out_indices);
  } else {
    set_error("key_type");
    rc = HCTR_ERR_INVALID_ARG;
  }
  if (rc == HCTR_OK && hipGetLastError() != hipSuccess) {
    set_error("ebc route launch failed");
    rc = HCTR_ERR_HIP;
  }
  if (rc == HCTR_OK && d_nnz)
    HCTR_HIP(hipMemcpyAsync(d_nnz, d_total, sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
  return rc;
}

int hctr_ebc_route_whole(size_t batch, int num_lookups, const int64_t* row_start, const void* keys,
                         const void* bucket_range, int key_type, int64_t* out_bucket_range,
                         uint64_t* out_indices, uint64_t* d_nnz, uint32_t* one_hot,
                         size_t nnz_hint, hctr_stream_t stream) {
  HCTR_REQUIRE(num_lookups >= 0, "num_lookups");
  hipStream_t s = as_stream(stream);
  const size_t nb = (size_t)num_lookups * batch;
  if (one_hot) HCTR_HIP(hipMemsetAsync(one_hot, nb > 0 ? 1 : 0, sizeof(uint32_t), s));
  if (nb == 0) {
    HCTR_HIP(hipMemsetAsync(out_bucket_range, 0, sizeof(int64_t), s));
    if (d_nnz) HCTR_HIP(hipMemsetAsync(d_nnz, 0, sizeof(uint64_t), s));
    return HCTR_OK;
  }
  HCTR_REQUIRE(row_start && bucket_range && out_bucket_range && out_indices, "null pointer");
  const size_t avg = nnz_hint > 0 ? (nnz_hint + nb - 1) / nb : 1;  // wavefronts per bucket chunk
  const dim3 grid(grid_for(nb, kBlock), (unsigned)(avg > 16 ? 16 : avg));
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ebc_route_whole_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, nb, batch,
                       (const long long*)row_start, (const uint32_t*)keys,
                       (const uint32_t*)bucket_range, (long long*)out_bucket_range, out_indices,
                       (unsigned long long*)d_nnz, one_hot);
  } else if (key_type == HCTR_KEY_I64) {
    hipLaunchKernelGGL(ebc_route_whole_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, nb, batch,
                       (const long long*)row_start, (const long long*)keys,
                       (const long long*)bucket_range, (long long*)out_bucket_range, out_indices,
                       (unsigned long long*)d_nnz, one_hot);
  } else {
    set_error("key_type");
    return HCTR_ERR_INVALID_ARG;
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_ebc_routed_keys_to_indices(size_t batch_per_gpu, int world, int num_local_lookups,
                                    const int32_t* lookup_desc, const int64_t* row_start,
                                    const int64_t* out_bucket_range, uint64_t* keys_inout,
                                    hctr_stream_t stream) {
  HCTR_REQUIRE(world >= 1 && num_local_lookups >= 0, "arguments");
  const size_t nb = (size_t)world * (size_t)num_local_lookups * batch_per_gpu;
  if (nb == 0) return HCTR_OK;
  HCTR_REQUIRE(lookup_desc && row_start && out_bucket_range && keys_inout, "null pointer");
  hipLaunchKernelGGL(ebc_routed_keys_to_indices_kernel, dim3(grid_for(nb, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), nb, batch_per_gpu, num_local_lookups, lookup_desc,
                     (const long long*)row_start, (const long long*)out_bucket_range, keys_inout);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_ebc_bucket_counts(size_t batch, int world, int rank, int num_lookup,
                           const void* bucket_range, int key_type, int64_t* counts,
                           hctr_stream_t stream) {
  HCTR_REQUIRE(world >= 1 && batch % world == 0 && rank >= 0 && rank < world, "rank/world/batch");
  const size_t bpg = batch / world;
  const size_t total = (size_t)num_lookup * bpg;
  if (total == 0) return HCTR_OK;
  HCTR_REQUIRE(bucket_range && counts, "null pointer");
  hipStream_t s = as_stream(stream);
  if (key_type == HCTR_KEY_U32)
    hipLaunchKernelGGL(ebc_bucket_count_kernel<uint32_t>, dim3(grid_for(total, kBlock)),
                       dim3(kBlock), 0, s, batch, bpg, num_lookup, rank,
                       (const uint32_t*)bucket_range, (long long*)counts);
  else
    hipLaunchKernelGGL(ebc_bucket_count_kernel<long long>, dim3(grid_for(total, kBlock)),
                       dim3(kBlock), 0, s, batch, bpg, num_lookup, rank,
                       (const long long*)bucket_range, (long long*)counts);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

static int ebc_network(bool fwd, size_t bpg, int num_lookup, int ev, int max_shards,
                       const int32_t* src_blocks, const int32_t* combiner,
                       const int64_t* bucket_counts, int batch_major, const void* in, void* out,
                       int dtype, hipStream_t s) {
  const size_t total = (size_t)num_lookup * bpg * ev;
  if (total == 0) return HCTR_OK;
  HCTR_REQUIRE(src_blocks && combiner && in && out, "null pointer");
  const int grid = grid_for(total, kBlock);
  const size_t esz = dtype == HCTR_EMB_F32 ? 4 : 2;
  const bool vec = ((size_t)ev * esz) % 16 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0;
#define HCTR_NET(T)                                                                             \
  if (vec) {                                                                                    \
    const int row16 = (int)((size_t)ev * sizeof(T) / 16);                                       \
    const int vgrid = grid_for((size_t)num_lookup * bpg * row16, kBlock, 8192);                 \
    if (fwd)                                                                                    \
      hipLaunchKernelGGL((ebc_network_vec_kernel<T, true>), dim3(vgrid), dim3(kBlock), 0, s,    \
                         bpg, num_lookup, row16, max_shards, src_blocks, combiner,              \
                         (const long long*)bucket_counts, batch_major, (const uint4*)in,        \
                         (uint4*)out);                                                          \
    else                                                                                        \
      hipLaunchKernelGGL((ebc_network_vec_kernel<T, false>), dim3(vgrid), dim3(kBlock), 0, s,   \
                         bpg, num_lookup, row16, max_shards, src_blocks, combiner,              \
                         (const long long*)bucket_counts, batch_major, (const uint4*)in,        \
                         (uint4*)out);                                                          \
  } else if (fwd)                                                                               \
    hipLaunchKernelGGL((ebc_network_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, bpg,      \
                       num_lookup, ev, max_shards, src_blocks, combiner,                        \
                       (const long long*)bucket_counts, batch_major, (const T*)in, (T*)out);    \
  else                                                                                          \
    hipLaunchKernelGGL((ebc_network_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, bpg,     \
                       num_lookup, ev, max_shards, src_blocks, combiner,                        \
                       (const long long*)bucket_counts, batch_major, (const T*)in, (T*)out);
  if (dtype == HCTR_EMB_F32) {
    HCTR_NET(float)
  } else if (dtype == HCTR_EMB_F16) {
    HCTR_NET(__half)
  } else if (dtype == HCTR_EMB_BF16) {
    HCTR_NET(__hip_bfloat16)
  } else {
    HCTR_REQUIRE(false, "dtype");
  }
#undef HCTR_NET
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_ebc_scale_average(size_t batch_per_gpu, int num_lookup, int ev_size,
                           const int32_t* d_combiner, const int64_t* d_bucket_counts,
                           int batch_major, void* data, int dtype, int forward,
                           hctr_stream_t stream) {
  const size_t total = (size_t)num_lookup * batch_per_gpu * ev_size;
  if (total == 0) return HCTR_OK;
  HCTR_REQUIRE(d_combiner && d_bucket_counts && data, "null pointer");
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(total, kBlock, 8192);
#define HCTR_SCALE(T)                                                                          \
  if (forward)                                                                                 \
    hipLaunchKernelGGL((ebc_scale_average_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s,    \
                       batch_per_gpu, num_lookup, ev_size, d_combiner,                         \
                       (const long long*)d_bucket_counts, batch_major, (T*)data);              \
  else                                                                                         \
    hipLaunchKernelGGL((ebc_scale_average_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s,   \
                       batch_per_gpu, num_lookup, ev_size, d_combiner,                         \
                       (const long long*)d_bucket_counts, batch_major, (T*)data);
  if (dtype == HCTR_EMB_F32) {
    HCTR_SCALE(float)
  } else if (dtype == HCTR_EMB_F16) {
    HCTR_SCALE(__half)
  } else if (dtype == HCTR_EMB_BF16) {
    HCTR_SCALE(__hip_bfloat16)
  } else {
    HCTR_REQUIRE(false, "dtype");
  }
#undef HCTR_SCALE
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_ebc_network_forward(size_t batch_per_gpu, int num_lookup, int ev_size, int max_shards,
                             const int32_t* d_src_blocks, const int32_t* d_combiner,
                             const int64_t* d_bucket_counts, int batch_major, const void* recv,
                             void* out, int dtype, hctr_stream_t stream) {
  return ebc_network(true, batch_per_gpu, num_lookup, ev_size, max_shards, d_src_blocks,
                     d_combiner, d_bucket_counts, batch_major, recv, out, dtype, as_stream(stream));
}

int hctr_ebc_network_backward(size_t batch_per_gpu, int num_lookup, int ev_size, int max_shards,
                              const int32_t* d_src_blocks, const int32_t* d_combiner,
                              const int64_t* d_bucket_counts, int batch_major, const void* grad,
                              void* send, int dtype, hctr_stream_t stream) {
  return ebc_network(false, batch_per_gpu, num_lookup, ev_size, max_shards, d_src_blocks,
                     d_combiner, d_bucket_counts, batch_major, grad, send, dtype,
                     as_stream(stream));
}

// ---- stateless sparse optimizer on a flat table (IGroupedEmbeddingTable::update) ----------------
int hctr_updater_create(size_t max_nnz, size_t max_rows, int vec_size, hctr_updater** out) {
  HCTR_REQUIRE(out && vec_size > 0 && max_rows > 0, "arguments");
  hctr_updater* u = new hctr_updater();
  int rc = u->impl.create(max_nnz, max_rows, vec_size);
  if (rc != HCTR_OK) {
    u->impl.destroy();
    delete u;
    return rc;
  }
  *out = u;
  return HCTR_OK;
}

int hctr_updater_set_ftrl(hctr_updater* u, float lambda1, float lambda2, float beta) {
  HCTR_REQUIRE(u, "null handle");
  u->ftrl_lambda1 = lambda1;
  u->ftrl_lambda2 = lambda2;
  u->ftrl_beta = beta;
  u->impl.allow_ftrl = true;
  return HCTR_OK;
}

int hctr_updater_set_grad_map(hctr_updater* u, size_t samples, size_t lookups) {
  HCTR_REQUIRE(u, "null handle");
  HCTR_REQUIRE((samples == 0 && lookups == 0) ||
                   (samples > 0 && lookups > 0 && samples * lookups <= 0xFFFFFFFFull),
               "samples / lookups");
  u->impl.map_inner = (uint32_t)samples;
  u->impl.map_outer = (uint32_t)lookups;
  return HCTR_OK;
}

int hctr_updater_set_row_bound(hctr_updater* u, uint64_t rows) {
  HCTR_REQUIRE(u, "null handle");
  u->impl.row_bound = (size_t)rows;
  return HCTR_OK;
}

int hctr_updater_destroy(hctr_updater* u) {
  if (!u) return HCTR_OK;
  (void)hipDeviceSynchronize();
  u->impl.destroy();
  u->lr_free();
  delete u;
  return HCTR_OK;
}

int hctr_updater_update(hctr_updater* u, size_t buckets, size_t nnz, const int64_t* bucket_range,
                        const uint64_t* indices, const void* grad, int grad_dtype, int optimizer,
                        int update_type, float lr, float beta1, float beta2, float epsilon,
                        float momentum_factor, float scaler, uint64_t times, float* table,
                        float* state0, float* state1, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  if (buckets == 0) return HCTR_OK;
  HCTR_REQUIRE(bucket_range && indices && grad && table, "null pointer");
  OptState o;
  o.optimizer = optimizer;
  o.update_type = update_type;
  o.lr = lr;
  o.beta1 = beta1;
  o.beta2 = beta2;
  o.epsilon = epsilon;
  o.momentum_factor = momentum_factor;
  o.scaler = scaler;
  o.atomic_update = 0;
  o.times = times;
  o.ftrl_lambda1 = u->ftrl_lambda1;
  o.ftrl_lambda2 = u->ftrl_lambda2;
  o.ftrl_beta = u->ftrl_beta;
  HCTR_REQUIRE(optimizer != HCTR_OPT_FTRL || (state0 && state1), "Ftrl needs state0 (n), state1 (z)");
  return u->impl.update(buckets, nnz, 0, bucket_range, HCTR_KEY_I64, indices, grad, grad_dtype, o,
                        table, state0, state1, nullptr, as_stream(stream));
}

int hctr_updater_reduce_presorted(hctr_updater* u, size_t positions, size_t buckets,
                                  const int64_t* row_offset, const uint32_t* sorted_rows,
                                  const uint32_t* sorted_buckets, const void* grad, int grad_dtype,
                                  size_t n_rows, float* out_sum, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  if (n_rows == 0) return HCTR_OK;
  HCTR_REQUIRE(out_sum, "null pointer");
  hipStream_t s = as_stream(stream);
  HCTR_HIP(hipMemsetAsync(out_sum, 0, n_rows * (size_t)u->impl.D * sizeof(float), s));
  if (positions == 0) return HCTR_OK;
  HCTR_REQUIRE(row_offset && sorted_rows && sorted_buckets && grad, "null pointer");
  HCTR_REQUIRE(positions <= u->impl.max_nnz, "positions exceed the updater's capacity");
  // out_sum[row] = sum of grad[bucket] over the row's run: the segmented reduce of the sparse
  // update with a store-only "optimizer" (rows that own no position keep the zero of the memset)
  OptState o;
  o.optimizer = kOptStoreSumId;
  o.update_type = HCTR_UPDATE_LOCAL;
  o.lr = 0.0f;
  o.scaler = 1.0f;
  o.atomic_update = 0;
  o.times = 1;
  u->impl.ext_rows = sorted_rows;
  u->impl.ext_buckets = sorted_buckets;
  const int rc = u->impl.update(buckets, positions, 0, row_offset, HCTR_KEY_I64, nullptr, grad,
                                grad_dtype, o, out_sum, nullptr, nullptr, nullptr, s);
  u->impl.ext_rows = nullptr;
  u->impl.ext_buckets = nullptr;
  return rc;
}

int hctr_ebc_local_reduce(hctr_updater* u, size_t buckets, size_t nnz, const int64_t* bucket_range,
                          const uint64_t* row_ids, uint64_t max_row_id, const uint64_t* keys,
                          const void* grad, int grad_dtype, size_t* num_unique,
                          uint64_t* unique_row_ids, uint64_t* unique_keys, float* wgrad,
                          hctr_stream_t stream) {
  HCTR_REQUIRE(u && num_unique, "null pointer");
  *num_unique = 0;
  if (buckets == 0 || nnz == 0) return HCTR_OK;
  HCTR_REQUIRE(bucket_range && row_ids && grad && unique_row_ids && wgrad, "null pointer");
  HCTR_REQUIRE(!unique_keys || keys, "unique_keys requested without keys");
  HCTR_REQUIRE(nnz <= u->impl.max_nnz && nnz < 0xFFFFFFF0ull, "nnz exceeds the updater's capacity");
  HCTR_REQUIRE(buckets < 0xFFFFFFF0ull, "buckets");
  HCTR_TRY(u->lr_alloc());
  hipStream_t s = as_stream(stream);
  unsigned bits = 1;
  while (bits < 64 && (max_row_id >> bits) != 0) bits++;
  HCTR_REQUIRE(u->impl.map_inner == 0u || (size_t)u->impl.map_inner * u->impl.map_outer == buckets,
               "gradient map: samples * lookups must equal the bucket count");
  // row ids below 2^32 (every table this side of 4 G rows): 32-bit keys through the path's own
  // radix sort; wider ids keep the library's 64-bit sort
  const bool k32 = max_row_id < 0xFFFFFFF0ull;
  hipLaunchKernelGGL(lr_expand_kernel, dim3(grid_for(buckets, kBlock)), dim3(kBlock), 0, s,
                     buckets, (const long long*)bucket_range, u->lr_pos_bucket, u->lr_pos_in,
                     u->impl.map_inner, u->impl.map_outer, row_ids, k32 ? u->lr_rows32 : nullptr);
  HCTR_LAUNCH_CHECK();
  // stable: equal rows keep ascending positions, so a row's gradients are summed in ascending
  // bucket order
  uint32_t* sorted32 = reinterpret_cast<uint32_t*>(u->lr_sorted);
  if (k32) {
    HCTR_TRY(radix_sort_pairs_u32(u->lr_temp, u->lr_temp_bytes, u->lr_rows32, sorted32, u->lr_pos_in,
                                  u->lr_pos_out, nnz, (int)bits, s));
  } else {
    size_t tb = u->lr_temp_bytes;
    if (rocprim::radix_sort_pairs(u->lr_temp, tb, row_ids, u->lr_sorted, u->lr_pos_in,
                                  u->lr_pos_out, nnz, 0, bits, s, false) != hipSuccess) {
      set_error("rocprim::radix_sort_pairs failed");
      return HCTR_ERR_HIP;
    }
  }
  const int grid = grid_for(nnz, kBlock, 4096);
  if (k32)
    hipLaunchKernelGGL(lr_flags_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, nnz, sorted32,
                       u->lr_flags);
  else
    hipLaunchKernelGGL(lr_flags_kernel<uint64_t>, dim3(grid), dim3(kBlock), 0, s, nnz, u->lr_sorted,
                       u->lr_flags);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(exclusive_scan_to_offsets<uint32_t>(u->lr_flags, nnz, u->lr_tile_sums, u->lr_total,
                                               u->lr_heads, s));
  if (k32)
    hipLaunchKernelGGL(lr_emit_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, nnz, sorted32,
                       u->lr_pos_out, u->lr_flags, u->lr_heads, u->lr_pos_bucket, keys, u->lr_cid,
                       u->lr_sbucket, unique_row_ids, unique_keys, u->lr_num_unique);
  else
    hipLaunchKernelGGL(lr_emit_kernel<uint64_t>, dim3(grid), dim3(kBlock), 0, s, nnz, u->lr_sorted,
                       u->lr_pos_out, u->lr_flags, u->lr_heads, u->lr_pos_bucket, keys, u->lr_cid,
                       u->lr_sbucket, unique_row_ids, unique_keys, u->lr_num_unique);
  HCTR_LAUNCH_CHECK();
  // the caller sizes the optimizer step from the count (the reference reads num_unique_keys on the
  // host at the same point, dynamic_embedding.cu:186-190)
  uint64_t nu = 0;
  HCTR_HIP(hipMemcpyAsync(&nu, u->lr_num_unique, sizeof(nu), hipMemcpyDeviceToHost, s));
  HCTR_HIP(hipStreamSynchronize(s));
  *num_unique = (size_t)nu;
  // the sorted bucket list already names gradient rows: the presorted reduce runs unmapped
  const uint32_t mi = u->impl.map_inner, mo = u->impl.map_outer;
  u->impl.map_inner = u->impl.map_outer = 0u;
  const int rc = hctr_updater_reduce_presorted(u, nnz, buckets, bucket_range, u->lr_cid,
                                               u->lr_sbucket, grad, grad_dtype, (size_t)nu, wgrad,
                                               stream);
  u->impl.map_inner = mi;
  u->impl.map_outer = mo;
  return rc;
}

}  // extern "C"
