// embedding_handle.hip -- IEmbedding-shaped handle: LocalizedSlotSparseEmbeddingHash /
// DistributedSlotSparseEmbeddingHash for ONE GPU (one process per GPU; rank/world placement).
//
// Reference: R/HugeCTR/include/embeddings/localized_slot_sparse_embedding_hash.hpp:57-611,
// R/HugeCTR/src/embeddings/localized_slot_sparse_embedding_hash.cu,
// R/HugeCTR/src/embeddings/distributed_slot_sparse_embedding_hash.cu.
// The all-to-all / reduce-scatter between ranks is issued by the caller (RCCL through
// torch.distributed); this file produces and consumes the buffers in the reference's layout.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "block_prims.h"
#include <cstdlib>

#include "common.h"
#include "hashtable.h"
#include "scan.h"
#include "sparse_update.h"

namespace hctr {

int forward_pool_dispatch(size_t buckets, int D, int combiner, const void* ro, int key_type,
                          const uint64_t* vi, const float* table, void* out, int out_dtype,
                          bool multi_hot, hipStream_t s, const uint32_t* one_hot,
                          uint32_t map_inner = 0, uint32_t map_outer = 0);

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;

// world == 1: private copy of the row offsets (what hipMemcpyAsync did) that also clears *one_hot
// when a bucket does not hold exactly one key -- the gather then takes its one-hot loop, which
// needs no row offsets (embedding_kernels.hip).  *one_hot is preset non-zero by the caller.
template <typename K>
__global__ void __launch_bounds__(kBlock)
    copy_offsets_check_kernel(const K* __restrict__ ro, size_t n_offsets, K* __restrict__ dst,
                              uint32_t* __restrict__ one_hot) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_offsets;
       i += (size_t)gridDim.x * kBlock) {
    const K v = ro[i];
    dst[i] = v;
    bad |= v != (K)i;  // lengths all 1 and ro[0] == 0  <=>  ro[i] == i for every i
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *one_hot = 0u;
}

// ---- localized: keep buckets whose slot % world == rank ----------------------------------------
// (select_value_and_rowoffset_by_slot_id_kernel, localized_slot_sparse_embedding_hash.cu:35-54)
template <typename K>
__global__ void __launch_bounds__(kBlock)
    localized_lens_kernel(const K* __restrict__ ro, size_t batch, int S, int spg, int rank,
                          int world, K* __restrict__ lens, uint32_t* __restrict__ one_hot) {
  const size_t total = batch * (size_t)spg;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t b = i / spg;
    const int j = (int)(i % spg);
    const size_t t = b * S + rank + (size_t)world * j;
    const K len = ro[t + 1] - ro[t];
    lens[i] = len;
    bad |= len != (K)1;
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *one_hot = 0u;
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    localized_copy_keys_kernel(const K* __restrict__ ro, const K* __restrict__ keys, size_t batch,
                               int S, int spg, int rank, int world, const K* __restrict__ out_ro,
                               K* __restrict__ out_keys) {
  const size_t total = batch * (size_t)spg;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t b = i / spg;
    const int j = (int)(i % spg);
    const size_t t = b * S + rank + (size_t)world * j;
    const size_t src = (size_t)ro[t], n = (size_t)(ro[t + 1] - ro[t]);
    const size_t dst = (size_t)out_ro[i];
    for (size_t q = 0; q < n; q++) out_keys[dst + q] = keys[src + q];
  }
}

// ---- distributed: keep keys with key % world == rank; all buckets kept --------------------------
// (distributed_slot_sparse_embedding_hash.cu:35-52,94-152)
template <typename K>
__global__ void __launch_bounds__(kBlock)
    distributed_lens_kernel(const K* __restrict__ ro, const K* __restrict__ keys, size_t buckets,
                            int rank, int world, K* __restrict__ lens,
                            uint32_t* __restrict__ one_hot) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < buckets;
       i += (size_t)gridDim.x * kBlock) {
    K c = 0;
    for (size_t q = (size_t)ro[i]; q < (size_t)ro[i + 1]; q++)
      c += ((keys[q] % (K)world) == (K)rank) ? 1 : 0;
    lens[i] = c;
    bad |= c != (K)1;
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *one_hot = 0u;
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    distributed_copy_keys_kernel(const K* __restrict__ ro, const K* __restrict__ keys,
                                 size_t buckets, int rank, int world, const K* __restrict__ out_ro,
                                 K* __restrict__ out_keys) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < buckets;
       i += (size_t)gridDim.x * kBlock) {
    size_t dst = (size_t)out_ro[i];
    for (size_t q = (size_t)ro[i]; q < (size_t)ro[i + 1]; q++) {
      const K k = keys[q];
      if ((k % (K)world) == (K)rank) out_keys[dst++] = k;
    }
  }
}

// ---- table initialisation ----------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// rows [row_begin, row_end) ~ U(-bound, bound); counter-based so the result is launch-shape
// independent (UniformGenerator::fill semantics, init_embedding_functor.cu:45-48)
__global__ void __launch_bounds__(kBlock)
    uniform_fill_kernel(float* __restrict__ table, size_t elem_begin, size_t elem_end, float bound,
                        uint64_t seed) {
  for (size_t i = elem_begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < elem_end;
       i += (size_t)gridDim.x * kBlock) {
    const uint64_t r = splitmix64(seed ^ (i * 0xD1342543DE82EF95ull));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // [0,1)
    table[i] = (2.0f * u - 1.0f) * bound;
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) fill_kernel(T* p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock)
    p[i] = v;
}


__global__ void gather_rows_kernel(const uint64_t* __restrict__ rows, size_t n, int D,
                                   const float* __restrict__ table, float* __restrict__ out,
                                   const uint64_t* __restrict__ slot_id_in,
                                   uint64_t* __restrict__ slot_id_out) {
  const size_t total = n * (size_t)D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / D, d = i % D;
    out[i] = table[rows[r] * (uint64_t)D + d];
    if (d == 0 && slot_id_out) slot_id_out[r] = slot_id_in[rows[r]];
  }
}

__global__ void scatter_rows_kernel(const uint64_t* __restrict__ rows, size_t n, int D,
                                    const float* __restrict__ in, float* __restrict__ table,
                                    const uint64_t* __restrict__ slot_id_in,
                                    uint64_t* __restrict__ slot_id) {
  const size_t total = n * (size_t)D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / D, d = i % D;
    table[rows[r] * (uint64_t)D + d] = in[i];
    if (d == 0 && slot_id_in && slot_id) slot_id[rows[r]] = slot_id_in[r];
  }
}

// forward_scale_kernel / forward_scale_align2_kernel (forward_scale_functor.cu:28-77): the
// distributed embedding pools with SUM, reduce-scatters, and only then divides a mean bucket by
// its key count over all GPUs.  T = float: x * (1/n).  16-bit T with an even vector size (the
// reference's align2 kernel): the scaler is rounded to T and the product is formed in T
// (__hmul2), otherwise float multiply + one rounding.
template <typename T, typename K>
__global__ void __launch_bounds__(kBlock)
    forward_scale_kernel(size_t buckets, int D, const K* __restrict__ ro, T* __restrict__ x) {
  const size_t total = buckets * (size_t)D;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t u = i / D;
    const long long n = (long long)ro[u + 1] - (long long)ro[u];
    if (n <= 1) continue;
    const float sc = 1.0f / (float)n;
    if constexpr (std::is_same<T, float>::value) {
      x[i] = x[i] * sc;
    } else if constexpr (std::is_same<T, __half>::value) {
      const float v = __half2float(x[i]);
      x[i] = (D % 2 == 0) ? __float2half_rn(v * __half2float(__float2half_rn(sc)))
                          : __float2half_rn(v * sc);
    } else {
      const float v = __bfloat162float(x[i]);
      x[i] = (D % 2 == 0) ? __float2bfloat16(v * __bfloat162float(__float2bfloat16(sc)))
                          : __float2bfloat16(v * sc);
    }
  }
}

__global__ void iota_kernel(uint64_t* p, size_t n, uint64_t base) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = base + i;
}

}  // namespace
}  // namespace hctr

using namespace hctr;

struct hctr_embedding {
  hctr_embedding_params p;
  std::vector<size_t> slot_sizes;
  int spg = 0;         // slots resolved on this rank
  size_t buckets_max = 0;  // max(batch) * buckets-per-sample
  size_t nnz_max = 0;
  int key_bytes = 8;
  HashTable ht;
  SparseUpdater upd;
  Profiler prof;
  OptState opt;
  float* table = nullptr;
  float* state0 = nullptr;
  float* state1 = nullptr;
  uint64_t* prev_time = nullptr;
  uint64_t* slot_id = nullptr;
  // per-batch state: CSR after filtering + resolved row indices.  Train and eval batches have
  // their own buffers (as the reference's row_offsets/value tensors do) so an eval forward
  // between forward(train) and update_params cannot clobber the training batch.
  struct BatchBufs {
    void* ro = nullptr;               // key-typed [buckets + 1]
    void* keys = nullptr;             // key-typed [nnz]
    uint64_t* value_index = nullptr;  // [nnz]
    // distributed + mean on N > 1 GPUs: the unfiltered full-batch row offsets.  Their bucket
    // lengths are what the reference gets from all_reduce(row_offsets) (the per-GPU filtered
    // counts add up to the full counts): the divisor of forward_scale and of backward_mean
    // (distributed_slot_sparse_embedding_hash.hpp:181-197,216-221).
    void* ro_full = nullptr;  // key-typed [batch * slot_num + 1]
    uint32_t* one_hot = nullptr;  // device flag: every bucket of the filtered CSR holds one key
  };
  BatchBufs tb, eb;
  // a second set of training buffers: hctr_emb_index_ahead resolves the NEXT batch into it while
  // the current batch's update still reads tb (hctr_emb_index_adopt swaps the two)
  BatchBufs tb_spare;
  bool ahead_valid = false;
  size_t ahead_buckets = 0, ahead_nnz = 0;
  const uint32_t* ahead_one_hot = nullptr;
  void*& ro = tb.ro;
  uint64_t*& value_index = tb.value_index;
  void* lens = nullptr;  // key-typed [buckets_max] scratch
  unsigned long long* tile_sums = nullptr;
  unsigned long long* d_nnz = nullptr;
  uint64_t* h_nnz = nullptr;  // pinned
  hipEvent_t nnz_event = nullptr;
  bool nnz_pending = false;
  // host-side upper bound of the rows handed out: the exact counter of a past batch (async copy
  // + event, never waited on) plus the keys of the batches enqueued since
  // The index stage's finish kernel POSTS {rows handed out, batch sequence number} and the
  // hash table's error flags to these pinned words (no copy launch, no event): h_rows[0] = row
  // counter after batch h_rows[1].  The host reads the sequence number first, so the counter it
  // pairs with it is never older; cum_keys[seq % kSeqRing] = keys enqueued through batch seq.
  // NOT thread-safe across streams: one stream order per handle (the index stage running ahead
  // on a side stream is ordered against the main stream by the caller's events).
  uint64_t* h_rows = nullptr;  // pinned [2]
  static constexpr int kSeqRing = 64;
  uint64_t seq = 0, min_valid_seq = 1, cum_total = 0;
  uint64_t cum_keys[kSeqRing] = {0};
  uint32_t* h_err = nullptr;  // pinned copy of the hash table's error flags (poll_overflow)
  uint32_t flip = 0;          // the training batch's one-hot flag is word flip % 4 of tb.one_hot
                              // (two batches can be in flight: the word of batch i is preset
                              //  again by the finish kernel of batch i + 3 at the earliest)
  const uint32_t* cur_one_hot = nullptr;  // the flag word of the batch update_params will take
                                          // (world == 1 only: the sort reads the rows in place)
  size_t last_exact_nnz = 0;     // world > 1: exact live nnz of the previous train batch
  // side-stream sort right after the index stage: on by default when world > 1 (it then runs
  // inside the all-to-all wait); on one GPU it would only share the chip with the dense tower --
  // measured: the step is as long as with the sort in line, 100 us (HCTR_PRESORT=1 / 0 overrides)
  bool presort_enabled = true;
  // one GPU: the grouping work of the update's hot-row path (SparseUpdater::prework) right behind
  // the index stage, on the updater's side streams, under the gather and the dense tower
  // (HCTR_PREWORK=0: inside the update)
  bool prework_enabled = false;
  // Ahead only pays while most keys of a batch are known (measured, MI355X: Criteo-1TB shape with
  // 4 % new keys a batch: step - 16 us; uniform keys over 416 M rows, every key new: the count's
  // 1.7 M device atomics next to the gather cost the step + 78 us): decided per batch from the
  // row counter the index stage posts -- rows handed out between the last two posts seen
  uint64_t post_seq = 0, post_rows = 0, post_new = ~0ull;  // post_new: new rows of the last posted batch
  size_t cur_buckets = 0;
  size_t cur_nnz_bound = 0;
  size_t eval_nnz = 0;  // keys of the last evaluation batch (host count)
  const void* top_grad = nullptr;
  bool has_train_batch = false;

  size_t buckets_per_sample() const {
    return p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH ? (size_t)spg : p.slot_num;
  }
  // the mean's divisor comes from the full CSR, after the reduce-scatter
  bool scale_after_reduce() const {
    return p.embedding_type == HCTR_EMB_DISTRIBUTED_SLOT_HASH && p.world > 1 && p.combiner == 1;
  }
};

static int report_ht_flags(uint32_t f);

namespace {

int free_all(hctr_embedding* e) {
  e->ht.destroy();
  e->upd.destroy();
  e->prof.destroy();
  void* ptrs[] = {e->table,  e->state0,  e->state1,         e->prev_time, e->slot_id,
                  e->tb.ro,  e->tb.keys, e->tb.value_index, e->eb.ro,     e->eb.keys,
                  e->eb.value_index,     e->lens,           e->tile_sums, e->d_nnz,
                  e->tb.ro_full,         e->eb.ro_full,     e->tb.one_hot, e->eb.one_hot,
                  e->tb_spare.ro,        e->tb_spare.value_index};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  if (e->h_nnz) (void)hipHostFree(e->h_nnz);
  if (e->nnz_event) (void)hipEventDestroy(e->nnz_event);
  if (e->h_err) (void)hipHostFree(e->h_err);
  if (e->h_rows) (void)hipHostFree(e->h_rows);
  return HCTR_OK;
}

int num_states(int optimizer) {
  switch (optimizer) {
    case HCTR_OPT_ADAM: return 2;
    case HCTR_OPT_ADAGRAD:
    case HCTR_OPT_MOMENTUM_SGD:
    case HCTR_OPT_NESTEROV: return 1;
    default: return 0;
  }
}

int reset_opt_states(hctr_embedding* e, hipStream_t s) {
  const size_t elems = e->p.max_vocabulary_size_per_gpu * e->p.embedding_vec_size;
  // EmbeddingOptimizer::initialize, sparse_optimizer.cu:111-168.  AdaGrad: the reference memsets
  // BYTES with initial_accu_value (only right for 0); we fill the float value.
  // (fp16 embeddings: the state arrays are __half arrays, optimizer.hpp:284-296)
  const bool half = e->opt.state_half != 0;
  if (e->state0) {
    const float v0 = e->p.optimizer == HCTR_OPT_ADAGRAD ? e->p.initial_accu_value : 0.0f;
    if (half)
      hipLaunchKernelGGL(fill_kernel<__half>, dim3(grid_for(elems, 256, 8192)), dim3(256), 0, s,
                         reinterpret_cast<__half*>(e->state0), elems, __float2half_rn(v0));
    else
      hipLaunchKernelGGL(fill_kernel<float>, dim3(grid_for(elems, 256, 8192)), dim3(256), 0, s,
                         e->state0, elems, v0);
    HCTR_LAUNCH_CHECK();
  }
  if (e->state1)
    HCTR_HIP(hipMemsetAsync(e->state1, 0, elems * (half ? sizeof(__half) : sizeof(float)), s));
  if (e->prev_time) {
    hipLaunchKernelGGL(fill_kernel<uint64_t>, dim3(grid_for(elems, 256, 8192)), dim3(256), 0, s,
                       e->prev_time, elems, (uint64_t)1);
    HCTR_LAUNCH_CHECK();
  }
  e->opt.times = 0;
  return HCTR_OK;
}

template <typename K>
int exclusive_scan_lens(hctr_embedding* e, void* ro_dst, size_t n, hipStream_t s) {
  return exclusive_scan_to_offsets<K>((const K*)e->lens, n, e->tile_sums, e->d_nnz, (K*)ro_dst, s);
}

// row-count bound for the sort's key width (SparseUpdater::row_bound): the counter after a past
// batch q (posted by its index stage's finish kernel) + every key enqueued since.  Called when the
// index stage is enqueued (a side-stream presort needs a bound then) and again by update_params,
// when this batch's own post has usually landed and the bound is exact.
void refresh_row_bound(hctr_embedding* e) {
  const uint64_t q = *(volatile uint64_t*)(e->h_rows + 1);  // sequence number first ...
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  const uint64_t rows = *(volatile uint64_t*)e->h_rows;     // ... rows at least that new
  if (q >= e->min_valid_seq && q <= e->seq && e->seq - q < (uint64_t)hctr_embedding::kSeqRing)
    e->upd.row_bound = rows + (e->cum_total - e->cum_keys[q % hctr_embedding::kSeqRing]);
  else
    e->upd.row_bound = 0;  // unknown: the sort takes the full key width
  if (q > e->post_seq) {  // a newer post: new rows per batch since the last one seen
    if (e->post_seq >= e->min_valid_seq && rows >= e->post_rows)
      e->post_new = (rows - e->post_rows) / (q - e->post_seq);
    e->post_seq = q;
    e->post_rows = rows;
  }
}

// returns (via *ro_out / *keys_out) the CSR this rank resolves
template <typename K>
int filter_keys(hctr_embedding* e, hctr_embedding::BatchBufs& bb, size_t batch, const K* ro_in,
                const K* keys_in, size_t nnz, const K** ro_out, const K** keys_out,
                size_t* buckets_out, hipStream_t s, uint32_t* one_hot, bool fused_train) {
  const int world = e->p.world, rank = e->p.rank, S = (int)e->p.slot_num;
  const bool localized = e->p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH;
  const size_t buckets = batch * e->buckets_per_sample();
  *buckets_out = buckets;
  // training batches: the flag of THIS batch was preset by the previous index stage's finish
  // kernel (two words, alternating); evaluation keeps the memset
  if (!fused_train)
    HCTR_HIP(hipMemsetAsync(one_hot, 1, sizeof(uint32_t), s));  // non-zero = "one-hot so far"
  if (world == 1) {
    // nothing to filter: keep a private copy of the row offsets (update_params needs them after
    // the caller's buffers may have been recycled); keys are consumed by the hash stage now.
    // Training: the copy + one-hot check ride in the probe kernel (IndexExtras), no launch here.
    if (!fused_train) {
      hipLaunchKernelGGL(copy_offsets_check_kernel<K>, dim3(grid_for(buckets + 1, kBlock, 1024)),
                         dim3(kBlock), 0, s, ro_in, buckets + 1, (K*)bb.ro, one_hot);
      HCTR_LAUNCH_CHECK();
    }
    *ro_out = (const K*)bb.ro;
    *keys_out = keys_in;
    return HCTR_OK;
  }
  if (buckets == 0) {
    HCTR_HIP(hipMemsetAsync(bb.ro, 0, sizeof(K), s));
    *ro_out = (const K*)bb.ro;
    *keys_out = (const K*)bb.keys;
    return HCTR_OK;
  }
  (void)nnz;
  if (localized) {
    hipLaunchKernelGGL(localized_lens_kernel<K>, dim3(grid_for(buckets, kBlock)), dim3(kBlock), 0,
                       s, ro_in, batch, S, e->spg, rank, world, (K*)e->lens, one_hot);
    HCTR_LAUNCH_CHECK();
    HCTR_TRY(exclusive_scan_lens<K>(e, bb.ro, buckets, s));
    hipLaunchKernelGGL(localized_copy_keys_kernel<K>, dim3(grid_for(buckets, kBlock)),
                       dim3(kBlock), 0, s, ro_in, keys_in, batch, S, e->spg, rank, world,
                       (const K*)bb.ro, (K*)bb.keys);
    HCTR_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(distributed_lens_kernel<K>, dim3(grid_for(buckets, kBlock)), dim3(kBlock),
                       0, s, ro_in, keys_in, buckets, rank, world, (K*)e->lens, one_hot);
    HCTR_LAUNCH_CHECK();
    HCTR_TRY(exclusive_scan_lens<K>(e, bb.ro, buckets, s));
    hipLaunchKernelGGL(distributed_copy_keys_kernel<K>, dim3(grid_for(buckets, kBlock)),
                       dim3(kBlock), 0, s, ro_in, keys_in, buckets, rank, world, (const K*)bb.ro,
                       (K*)bb.keys);
    HCTR_LAUNCH_CHECK();
  }
  *ro_out = (const K*)bb.ro;
  *keys_out = (const K*)bb.keys;
  return HCTR_OK;
}

// ahead: index-only call for the batch AFTER the current one (hctr_emb_index_ahead): resolved into
// tb_spare, the handle's current training batch stays what it is
template <typename K>
int forward_typed(hctr_embedding* e, int is_train, const K* ro_in, const K* keys_in, size_t nnz,
                  void* out, hipStream_t s, bool ahead = false) {
  const size_t batch = is_train ? e->p.train_batch_size : e->p.evaluate_batch_size;
  const K *ro = nullptr, *keys = nullptr;
  size_t buckets = 0;
  hctr_embedding::BatchBufs& bb = ahead ? e->tb_spare : (is_train ? e->tb : e->eb);
  HCTR_REQUIRE(bb.ro != nullptr, "forward: batch size 0 configured for this mode");
  if (is_train && !ahead) e->ahead_valid = false;  // (a batch indexed ahead is stale now)
  // the training index stage is two launches: its probe kernel also copies / checks the row
  // offsets (world == 1), its finish kernel presets the next batch's one-hot flag and posts the
  // row counter + error flags to pinned host words
  const bool fused_train = is_train != 0 && nnz > 0;
  // A finish kernel that timed out at its grid barrier (error bit 4) left this table's
  // first-occurrence masks and region counts half written: a later inserting batch would rank its
  // keys against them, and the slots the batch claimed still hold PENDING | position.  Sticky: no
  // further training batch until the caller has cleared the table with hctr_emb_reset (the one
  // call that empties the hash index and zeroes this flag; init_params / load do neither).
  if (is_train && (*(volatile uint32_t*)e->h_err & 4u) != 0u) return report_ht_flags(4u);
  uint32_t* one_hot = bb.one_hot;
  uint32_t* one_hot_next = nullptr;
  if (fused_train) {
    one_hot = e->tb.one_hot + (e->flip & 3u);
    one_hot_next = e->tb.one_hot + ((e->flip + 1u) & 3u);
    e->flip++;
  }
  HCTR_TRY(filter_keys<K>(e, bb, batch, ro_in, keys_in, nnz, &ro, &keys, &buckets, s, one_hot,
                          fused_train));
  if (buckets == 0) {
    if (fused_train) e->flip--;  // (nothing ran that would preset the other flag)
    if (is_train) {
      // a rank that owns no slot (localized, slot_num < GPUs): an empty training batch, so that
      // backward() / update_params() of the common loop are no-ops instead of errors
      e->cur_buckets = 0;
      e->cur_nnz_bound = 0;
      e->has_train_batch = true;
      e->nnz_pending = false;
    }
    return HCTR_OK;
  }
  const uint32_t* batch_one_hot = (fused_train && e->p.world == 1) ? one_hot : nullptr;
  if (is_train && !ahead) e->cur_one_hot = batch_one_hot;
  if (bb.ro_full)
    HCTR_HIP(hipMemcpyAsync(bb.ro_full, ro_in, (batch * e->p.slot_num + 1) * sizeof(K),
                            hipMemcpyDeviceToDevice, s));
  // the live key count of this rank is ro[buckets] (device); nnz is its host upper bound
  const K* d_live = ro + buckets;
  // widen the live count to the uint64 the hash kernels read
  // (for world == 1 nnz is exact and d_n can be skipped)
  const uint64_t* d_n = nullptr;
  if (e->p.world > 1) {
    d_n = (const uint64_t*)e->d_nnz;  // written by exclusive_scan_lens
  }
  (void)d_live;
  if (nnz > 0) {
    e->prof.begin(1, s);
    if (is_train) {
      SlotSink sink;
      sink.slot_id = e->slot_id;
      sink.row_offset = ro;
      sink.buckets = buckets;
      sink.buckets_per_sample = (int)e->buckets_per_sample();
      sink.rank = e->p.rank;
      sink.world = e->p.world;
      sink.localized = e->p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH ? 1 : 0;
      IndexExtras x;
      if (e->p.world == 1) {
        x.ro_src = ro_in;
        x.ro_dst = bb.ro;
        x.n_offsets = buckets + 1;
        x.one_hot = one_hot;
      }
      x.one_hot_next = one_hot_next;
      e->seq++;
      e->cum_total += nnz;
      e->cum_keys[e->seq % hctr_embedding::kSeqRing] = e->cum_total;
      x.host_rows = e->h_rows;
      x.host_seq = e->h_rows + 1;
      x.seq = e->seq;
      x.host_error = e->h_err;
      x.two_launches = ahead;  // (beside the dense tower's GEMMs: no grid barrier)
      HCTR_TRY(e->ht.get_insert(keys, nnz, d_n, bb.value_index, s, &sink, &x));
      refresh_row_bound(e);
    } else {
      HCTR_TRY(e->ht.get_mark(keys, nnz, d_n, bb.value_index, s));
    }
    e->prof.end(1, s);
    // (the switch is read per call: bench.py times the update with and without the work ahead)
    const char* pw_env = getenv("HCTR_PREWORK");
    const bool pw_on = pw_env ? pw_env[0] != '0' : e->prework_enabled;
    // (a batch of mostly unseen keys keeps the work inside the update: its count is one device
    //  atomic per position, next to the gather that is the slower place -- "2": regardless)
    const bool pw_few_new = (pw_env && pw_env[0] == '2') ||
                            (e->post_new != ~0ull && e->post_new * 4 < nnz);
    if (is_train && !ahead && e->p.world == 1 && pw_on && pw_few_new &&
        !(e->opt.optimizer == HCTR_OPT_SGD && e->opt.atomic_update)) {
      e->upd.one_hot_flag = batch_one_hot;
      e->upd.scale_row_offset = nullptr;
      const int prc = e->upd.prework(buckets, nnz, e->p.combiner, ro, e->p.key_type,
                                     bb.value_index, s);
      e->upd.one_hot_flag = nullptr;
      HCTR_TRY(prc);
    }
  }
  if (!is_train) e->eval_nnz = nnz;
  if (ahead) {
    e->ahead_buckets = buckets;
    e->ahead_nnz = nnz;
    e->ahead_one_hot = batch_one_hot;
    e->ahead_valid = true;
    return HCTR_OK;
  }
  if (out == nullptr) {  // hctr_emb_index: resolve rows only (unique-row exchange)
    if (is_train) {
      e->cur_buckets = buckets;
      e->cur_nnz_bound = nnz;
      e->has_train_batch = true;
      e->nnz_pending = false;
    }
    return HCTR_OK;
  }
  e->prof.begin(0, s);
  // more keys than buckets in the full-batch CSR (host numbers) -> the flat multi-hot walk
  const size_t full_buckets = batch * (size_t)e->p.slot_num;
  const bool multi_hot = nnz > full_buckets + full_buckets / 2;
  // distributed on N > 1 GPUs pools partial SUMS (forward_per_gpu is called with combiner 0,
  // distributed_slot_sparse_embedding_hash.hpp:162-170); hctr_emb_forward_scale divides later
  const int pool_combiner = e->scale_after_reduce() ? 0 : e->p.combiner;
  HCTR_TRY(forward_pool_dispatch(buckets, (int)e->p.embedding_vec_size, pool_combiner, ro,
                                 e->p.key_type, bb.value_index, e->table, out, e->p.out_dtype,
                                 multi_hot, s, one_hot));
  e->prof.end(0, s);
  if (nnz > 0 && is_train && e->presort_enabled &&
      !(e->opt.optimizer == HCTR_OPT_SGD && e->opt.atomic_update)) {
    // The update's (row, bucket) sort depends on the index stage only: start it on the updater's
    // side stream, under everything the caller does between forward and update_params (after
    // the gather: both are memory-bound and would only share HBM).  Its size must be known on
    // the host: exact for one rank; for world > 1 the previous batch's exact count plus 1/8
    // head room (update_params re-sorts if that was short).
    size_t n_sort = nnz;
    if (e->p.world > 1) {
      n_sort = e->last_exact_nnz ? e->last_exact_nnz + e->last_exact_nnz / 8 + 1024 : 0;
      if (n_sort > nnz) n_sort = nnz;
    }
    if (n_sort > 0) {
      e->upd.one_hot_flag = e->cur_one_hot;
      const int prc = e->upd.presort(buckets, n_sort, ro, e->p.key_type, bb.value_index, s);
      e->upd.one_hot_flag = nullptr;
      HCTR_TRY(prc);
    }
  }
  if (is_train) {
    e->cur_buckets = buckets;
    e->cur_nnz_bound = nnz;
    e->has_train_batch = true;
    e->nnz_pending = false;
    if (e->p.world > 1) {
      // exact per-rank nnz for the sort in update_params, fetched without blocking the host:
      // by the time update_params runs (after the dense fwd/bwd) the copy has long completed.
      HCTR_HIP(hipMemcpyAsync(e->h_nnz, e->d_nnz, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      HCTR_HIP(hipEventRecord(e->nnz_event, s));
      e->nnz_pending = true;
    }
  }
  return HCTR_OK;
}

}  // namespace

extern "C" {

int hctr_emb_create(const hctr_embedding_params* params, hctr_embedding** out) {
  HCTR_REQUIRE(params && out, "null pointer");
  const hctr_embedding_params& p = *params;
  HCTR_REQUIRE(p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH ||
                   p.embedding_type == HCTR_EMB_DISTRIBUTED_SLOT_HASH,
               "embedding_type");
  HCTR_REQUIRE(p.key_type == HCTR_KEY_U32 || p.key_type == HCTR_KEY_I64, "key_type");
  HCTR_REQUIRE(p.out_dtype >= HCTR_EMB_F32 && p.out_dtype <= HCTR_EMB_BF16, "out_dtype");
  HCTR_REQUIRE(p.world >= 1 && p.rank >= 0 && p.rank < p.world, "rank/world");
  HCTR_REQUIRE(p.embedding_vec_size > 0 && p.slot_num > 0, "embedding_vec_size/slot_num");
  HCTR_REQUIRE(p.combiner == 0 || p.combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  HCTR_REQUIRE(p.train_batch_size % p.world == 0 && p.evaluate_batch_size % p.world == 0,
               "batch size must be divisible by the number of GPUs");
  HCTR_REQUIRE(p.scaler > 0.f, "scaler must be > 0");
  hctr_embedding* e = new hctr_embedding();
  e->p = p;
  if (p.slot_size_array) e->slot_sizes.assign(p.slot_size_array, p.slot_size_array + p.slot_num);
  e->p.slot_size_array = nullptr;
  const bool localized = p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH;
  e->spg = (int)(p.slot_num / p.world + ((size_t)p.rank < p.slot_num % p.world ? 1 : 0));
  e->key_bytes = p.key_type == HCTR_KEY_U32 ? 4 : 8;
  if (e->p.max_vocabulary_size_per_gpu == 0 && !e->slot_sizes.empty()) {
    // cal_max_voc_size_per_gpu, localized_slot_sparse_embedding_hash.hpp:100-122
    size_t mx = 0;
    for (int g = 0; g < p.world; g++) {
      size_t tot = 0;
      for (size_t i = 0; i < p.slot_num; i++) {
        if (localized ? ((int)(i % p.world) == g) : true)
          tot += localized ? e->slot_sizes[i] : ceil_div<size_t>(e->slot_sizes[i], p.world);
      }
      if (tot > mx) mx = tot;
    }
    e->p.max_vocabulary_size_per_gpu = mx;
  }
  if (e->p.max_vocabulary_size_per_gpu == 0) {
    delete e;
    HCTR_REQUIRE(false, "max_vocabulary_size_per_gpu is 0 and no slot_size_array given");
  }
  const size_t bmax =
      p.train_batch_size > p.evaluate_batch_size ? p.train_batch_size : p.evaluate_batch_size;
  e->buckets_max = bmax * e->buckets_per_sample();
  e->nnz_max = bmax * p.max_feature_num;
  if (e->nnz_max == 0) e->nnz_max = 1;
  const size_t V = e->p.max_vocabulary_size_per_gpu, D = p.embedding_vec_size;

  int rc = HCTR_OK;
  auto fail = [&](int code) {
    free_all(e);
    delete e;
    return code;
  };
#define HCTR_ALLOC(ptr, bytes)                                                         \
  do {                                                                                 \
    hipError_t er = hipMalloc((void**)&(ptr), (bytes));                                \
    if (er != hipSuccess) {                                                            \
      set_error(std::string("hipMalloc(" #ptr "): ") + hipGetErrorString(er));         \
      (void)hipGetLastError();                                                         \
      return fail(HCTR_ERR_HIP);                                                       \
    }                                                                                  \
  } while (0)
  HCTR_ALLOC(e->table, V * D * sizeof(float));
  const int ns = num_states(p.optimizer);
  // the state of fp16 embeddings is stored in fp16 (OptimizerTensor<__half>, optimizer.hpp:284-296)
  const size_t state_elem = p.out_dtype == HCTR_EMB_F16 ? sizeof(__half) : sizeof(float);
  if (ns >= 1) HCTR_ALLOC(e->state0, V * D * state_elem);
  if (ns >= 2) HCTR_ALLOC(e->state1, V * D * state_elem);
  if (p.optimizer == HCTR_OPT_ADAM && p.update_type == HCTR_UPDATE_LAZY_GLOBAL)
    HCTR_ALLOC(e->prev_time, V * D * sizeof(uint64_t));
  HCTR_ALLOC(e->slot_id, V * sizeof(uint64_t));
  for (int mode = 0; mode < 2; mode++) {
    const size_t bsz = mode == 0 ? p.train_batch_size : p.evaluate_batch_size;
    if (bsz == 0) continue;
    hctr_embedding::BatchBufs& bb = mode == 0 ? e->tb : e->eb;
    size_t nn = bsz * p.max_feature_num;
    if (nn == 0) nn = 1;
    HCTR_ALLOC(bb.ro, (bsz * e->buckets_per_sample() + 1) * e->key_bytes);
    HCTR_ALLOC(bb.keys, nn * e->key_bytes);
    HCTR_ALLOC(bb.value_index, nn * sizeof(uint64_t));
    HCTR_ALLOC(bb.one_hot, 64);
    (void)hipMemset(bb.one_hot, 1, 64);  // both flags of the training batches start "one-hot"
    if (e->scale_after_reduce()) HCTR_ALLOC(bb.ro_full, (bsz * p.slot_num + 1) * e->key_bytes);
    if (mode == 0 && p.world == 1) {  // (index-ahead is a one-GPU schedule)
      HCTR_ALLOC(e->tb_spare.ro, (bsz * e->buckets_per_sample() + 1) * e->key_bytes);
      HCTR_ALLOC(e->tb_spare.value_index, nn * sizeof(uint64_t));
    }
  }
  HCTR_ALLOC(e->lens, (e->buckets_max + 1) * e->key_bytes);
  HCTR_ALLOC(e->tile_sums, (ceil_div<size_t>(e->buckets_max + 1, kTile) + 1) * 8);
  HCTR_ALLOC(e->d_nnz, 8);
#undef HCTR_ALLOC
  if (hipHostMalloc((void**)&e->h_nnz, 8, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&e->h_err, 8, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&e->h_rows, 16, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&e->nnz_event, hipEventDisableTiming) != hipSuccess) {
    set_error("pinned host / event allocation failed");
    return fail(HCTR_ERR_HIP);
  }
  e->h_rows[0] = e->h_rows[1] = 0;
  *e->h_err = 0u;
  if ((rc = e->ht.create(V, p.key_type)) != HCTR_OK) return fail(rc);
  if ((rc = e->ht.reserve(e->nnz_max)) != HCTR_OK) return fail(rc);
  if ((rc = e->upd.create(e->nnz_max, V, (int)D, true)) != HCTR_OK) return fail(rc);
  // sample-major batches: bucket b * S + s -- the positions with the same p % S come from one table
  // (the sparse update's hot-row streams)
  e->upd.hot_streams = (uint32_t)e->buckets_per_sample();
  e->presort_enabled = p.world > 1;
  if (const char* ps = getenv("HCTR_PRESORT")) e->presort_enabled = ps[0] != '0';
  // measured (MI355X, Criteo-1TB shape, nine runs on one box): the step with the grouping ahead is
  // within the run-to-run noise of the step without (2.265 / 2.286 vs 2.233 ms) -- off unless asked
  // for (HCTR_PREWORK=1, then still decided per batch from the posted row counter)
  e->prework_enabled = false;
  e->upd.prof = &e->prof;
  e->opt.optimizer = p.optimizer;
  e->opt.update_type = p.update_type;
  e->opt.lr = p.lr;
  e->opt.beta1 = p.beta1;
  e->opt.beta2 = p.beta2;
  e->opt.epsilon = p.epsilon;
  e->opt.momentum_factor = p.momentum_factor;
  e->opt.scaler = p.scaler;
  e->opt.atomic_update = p.atomic_update;
  // OptimizerTensor<TypeEmbeddingComp>: fp16 embeddings keep their optimizer state in fp16
  e->opt.state_half = p.out_dtype == HCTR_EMB_F16 ? 1 : 0;
  if (hipMemset(e->slot_id, 0, V * sizeof(uint64_t)) != hipSuccess ||
      hipMemset(e->table, 0, V * D * sizeof(float)) != hipSuccess) {
    set_error("hipMemset failed");
    return fail(HCTR_ERR_HIP);
  }
  if ((rc = reset_opt_states(e, nullptr)) != HCTR_OK) return fail(rc);
  if (hipDeviceSynchronize() != hipSuccess) {
    set_error("hipDeviceSynchronize failed after create");
    return fail(HCTR_ERR_HIP);
  }
  *out = e;
  return HCTR_OK;
}

int hctr_emb_destroy(hctr_embedding* e) {
  if (!e) return HCTR_OK;
  (void)hipDeviceSynchronize();
  free_all(e);
  delete e;
  return HCTR_OK;
}

int hctr_emb_init_params(hctr_embedding* e, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  hipStream_t s = as_stream(stream);
  const size_t V = e->p.max_vocabulary_size_per_gpu, D = e->p.embedding_vec_size;
  const bool localized = e->p.embedding_type == HCTR_EMB_LOCALIZED_SLOT_HASH;
  const uint64_t seed = e->p.seed * 0x9E3779B97F4A7C15ull + (uint64_t)e->p.rank + 1;
  if (e->slot_sizes.empty() || !localized) {
    // no slot sizes (or distributed): U(-0.05, 0.05) over the whole table
    // (localized_slot_sparse_embedding_hash.cu:1261-1279)
    hipLaunchKernelGGL(uniform_fill_kernel, dim3(grid_for(V * D, kBlock, 8192)), dim3(kBlock), 0,
                       s, e->table, (size_t)0, V * D, 0.05f, seed);
    HCTR_LAUNCH_CHECK();
    return HCTR_OK;
  }
  // init_embedding_per_gpu (init_embedding_functor.cu:24-55): consecutive row ranges, one per
  // slot owned by this rank, bound sqrt(1/slot_size); slot ids preset per range
  size_t row = 0;
  for (size_t i = 0; i < e->slot_sizes.size(); i++) {
    if ((int)(i % e->p.world) != e->p.rank) continue;
    size_t n = e->slot_sizes[i];
    if (row + n > V) n = V - row;
    if (n == 0) continue;
    const float bound = sqrtf(1.f / (float)e->slot_sizes[i]);
    hipLaunchKernelGGL(uniform_fill_kernel, dim3(grid_for(n * D, kBlock, 8192)), dim3(kBlock), 0,
                       s, e->table, row * D, (row + n) * D, bound, seed);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(fill_kernel<uint64_t>, dim3(grid_for(n, kBlock, 8192)), dim3(kBlock), 0, s,
                       e->slot_id + row, n, (uint64_t)i);
    HCTR_LAUNCH_CHECK();
    row += n;
  }
  return HCTR_OK;
}

int hctr_emb_forward(hctr_embedding* e, int is_train, const void* row_offset, const void* keys,
                     size_t nnz, void* out, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  // (a rank that owns no slot has an empty output, whose pointer may be null)
  HCTR_REQUIRE(row_offset && (out || e->buckets_per_sample() == 0), "null pointer");
  HCTR_REQUIRE(nnz == 0 || keys, "keys is null");
  HCTR_REQUIRE(nnz <= (is_train ? e->p.train_batch_size : e->p.evaluate_batch_size) *
                          (e->p.max_feature_num ? e->p.max_feature_num : 1),
               "nnz exceeds batch_size * max_feature_num");
  hipStream_t s = as_stream(stream);
  if (e->p.key_type == HCTR_KEY_U32)
    return forward_typed<uint32_t>(e, is_train, (const uint32_t*)row_offset,
                                   (const uint32_t*)keys, nnz, out, s);
  return forward_typed<long long>(e, is_train, (const long long*)row_offset,
                                  (const long long*)keys, nnz, out, s);
}

int hctr_emb_forward_scale(hctr_embedding* e, int is_train, void* out_local,
                           hctr_stream_t stream) {
  HCTR_REQUIRE(e && out_local, "null pointer");
  if (!e->scale_after_reduce()) return HCTR_OK;  // pooled with the final divisor already
  const hctr_embedding::BatchBufs& bb = is_train ? e->tb : e->eb;
  HCTR_REQUIRE(bb.ro_full != nullptr, "forward_scale: batch size 0 configured for this mode");
  const size_t batch = is_train ? e->p.train_batch_size : e->p.evaluate_batch_size;
  const size_t bpg = batch / (size_t)e->p.world;
  const size_t buckets = bpg * e->p.slot_num, first = (size_t)e->p.rank * buckets;
  if (buckets == 0) return HCTR_OK;
  const int D = (int)e->p.embedding_vec_size;
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(buckets * (size_t)D, kBlock);
#define HCTR_FS(T, K)                                                                         \
  hipLaunchKernelGGL((forward_scale_kernel<T, K>), dim3(grid), dim3(kBlock), 0, s, buckets, D, \
                     (const K*)bb.ro_full + first, (T*)out_local)
  if (e->p.key_type == HCTR_KEY_U32) {
    if (e->p.out_dtype == HCTR_EMB_F32) HCTR_FS(float, uint32_t);
    else if (e->p.out_dtype == HCTR_EMB_F16) HCTR_FS(__half, uint32_t);
    else HCTR_FS(__hip_bfloat16, uint32_t);
  } else {
    if (e->p.out_dtype == HCTR_EMB_F32) HCTR_FS(float, long long);
    else if (e->p.out_dtype == HCTR_EMB_F16) HCTR_FS(__half, long long);
    else HCTR_FS(__hip_bfloat16, long long);
  }
#undef HCTR_FS
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_emb_index(hctr_embedding* e, int is_train, const void* row_offset, const void* keys,
                   size_t nnz, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  HCTR_REQUIRE(row_offset, "null pointer");
  HCTR_REQUIRE(nnz == 0 || keys, "keys is null");
  HCTR_REQUIRE(nnz <= (is_train ? e->p.train_batch_size : e->p.evaluate_batch_size) *
                          (e->p.max_feature_num ? e->p.max_feature_num : 1),
               "nnz exceeds batch_size * max_feature_num");
  hipStream_t s = as_stream(stream);
  if (e->p.key_type == HCTR_KEY_U32)
    return forward_typed<uint32_t>(e, is_train, (const uint32_t*)row_offset,
                                   (const uint32_t*)keys, nnz, nullptr, s);
  return forward_typed<long long>(e, is_train, (const long long*)row_offset,
                                  (const long long*)keys, nnz, nullptr, s);
}

int hctr_emb_index_ahead(hctr_embedding* e, const void* row_offset, const void* keys, size_t nnz,
                         hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  HCTR_REQUIRE(row_offset && keys && nnz > 0, "null pointer / empty batch");
  HCTR_REQUIRE(e->p.world == 1 && e->tb_spare.ro != nullptr, "index_ahead: one GPU, training mode");
  HCTR_REQUIRE(nnz <= e->p.train_batch_size * (e->p.max_feature_num ? e->p.max_feature_num : 1),
               "nnz exceeds batch_size * max_feature_num");
  HCTR_REQUIRE(!e->ahead_valid, "index_ahead: the batch indexed ahead has not been adopted yet");
  hipStream_t s = as_stream(stream);
  if (e->p.key_type == HCTR_KEY_U32)
    return forward_typed<uint32_t>(e, 1, (const uint32_t*)row_offset, (const uint32_t*)keys, nnz,
                                   nullptr, s, true);
  return forward_typed<long long>(e, 1, (const long long*)row_offset, (const long long*)keys, nnz,
                                  nullptr, s, true);
}

int hctr_emb_index_adopt(hctr_embedding* e) {
  HCTR_REQUIRE(e, "null handle");
  HCTR_REQUIRE(e->ahead_valid, "index_adopt: no batch has been indexed ahead");
  // (pointers only: kernels of the previous batch's update that are still queued hold the old ones)
  std::swap(e->tb.ro, e->tb_spare.ro);
  std::swap(e->tb.value_index, e->tb_spare.value_index);
  e->cur_buckets = e->ahead_buckets;
  e->cur_nnz_bound = e->ahead_nnz;
  e->cur_one_hot = e->ahead_one_hot;
  e->has_train_batch = true;
  e->nnz_pending = false;
  e->top_grad = nullptr;
  e->ahead_valid = false;
  return HCTR_OK;
}

int hctr_emb_update_rows(hctr_embedding* e, size_t n, const int64_t* row_offset,
                         const uint64_t* rows, const void* grad, int grad_dtype,
                         hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  if (n == 0) {
    e->opt.times++;
    return HCTR_OK;
  }
  HCTR_REQUIRE(row_offset && rows && grad, "null pointer");
  e->opt.times++;
  return e->upd.update(n, n, 0, row_offset, HCTR_KEY_I64, rows, grad, grad_dtype, e->opt, e->table,
                       e->state0, e->state1, e->prev_time, as_stream(stream));
}

int hctr_emb_backward(hctr_embedding* e, const void* top_grad, hctr_stream_t stream) {
  (void)stream;
  HCTR_REQUIRE(e, "null pointer");
  HCTR_REQUIRE(e->has_train_batch, "backward() before forward(is_train=1)");
  // (an empty batch has an empty gradient, whose pointer may be null: never dereferenced)
  HCTR_REQUIRE(top_grad || e->cur_buckets == 0, "null pointer");
  e->top_grad = top_grad ? top_grad : (const void*)e;
  return HCTR_OK;
}

int hctr_emb_forward_interaction(hctr_embedding* e, int is_train, const void* mlp, void* pooled,
                                 void* out, hctr_stream_t stream) {
  HCTR_REQUIRE(e && mlp && pooled && out, "null pointer");
  HCTR_REQUIRE(e->p.world == 1, "forward_interaction: one GPU (the pooled vectors are exchanged "
                                "before the interaction otherwise)");
  HCTR_REQUIRE(!is_train || e->has_train_batch, "forward_interaction before the index stage");
  const size_t batch = is_train ? e->p.train_batch_size : e->p.evaluate_batch_size;
  // the kernel reads value_index as [sample][slot]: the indexed batch must hold one key per bucket
  HCTR_REQUIRE((is_train ? e->cur_nnz_bound : e->eval_nnz) == batch * e->p.slot_num,
               "forward_interaction: the indexed batch does not hold one key per bucket");
  hctr_embedding::BatchBufs& bb = is_train ? e->tb : e->eb;
  HCTR_REQUIRE(bb.value_index != nullptr, "batch size 0 configured for this mode");
  hipStream_t s = as_stream(stream);
  e->prof.begin(0, s);
  const int rc = hctr_interaction_fwd_gather(batch, (int)e->p.slot_num,
                                             (int)e->p.embedding_vec_size, mlp, e->table,
                                             bb.value_index, pooled, out, e->p.out_dtype, stream);
  e->prof.end(0, s);
  return rc;
}

int hctr_emb_get_wgrad(hctr_embedding* e, void* wgrad, hctr_stream_t stream) {
  HCTR_REQUIRE(e && wgrad, "null pointer");
  HCTR_REQUIRE(e->top_grad, "get_wgrad() before backward()");
  return materialize_wgrad(e->cur_buckets, (int)e->p.embedding_vec_size, e->p.combiner,
                           e->tb.ro_full ? e->tb.ro_full : e->ro, e->p.key_type, e->top_grad,
                           wgrad, e->p.out_dtype, as_stream(stream));
}

int hctr_emb_update_params(hctr_embedding* e, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  HCTR_REQUIRE(e->has_train_batch && e->top_grad, "update_params() before forward()+backward()");
  hipStream_t s = as_stream(stream);
  size_t nnz = e->cur_nnz_bound;
  if (e->nnz_pending) {
    HCTR_HIP(hipEventSynchronize(e->nnz_event));
    nnz = (size_t)*e->h_nnz;
    e->last_exact_nnz = nnz;
    e->nnz_pending = false;
  }
  e->opt.times++;  // update_params(): adam.times++ before the update (…hash.hpp:346-347)
  if (e->seq > 0) refresh_row_bound(e);
  e->upd.scale_row_offset = e->tb.ro_full;  // NULL unless distributed + mean + N > 1
  e->upd.one_hot_flag = e->cur_one_hot;
  const int rc = e->upd.update(e->cur_buckets, nnz, e->p.combiner, e->ro, e->p.key_type,
                               e->value_index, e->top_grad, e->p.out_dtype, e->opt, e->table,
                               e->state0, e->state1, e->prev_time, s);
  // key-typed, batch-sized: never left behind for update_rows (int64 offsets of another length)
  e->upd.scale_row_offset = nullptr;
  e->upd.one_hot_flag = nullptr;
  return rc;
}

int hctr_emb_set_learning_rate(hctr_embedding* e, float lr) {
  HCTR_REQUIRE(e, "null handle");
  e->opt.lr = lr;
  e->p.lr = lr;
  return HCTR_OK;
}

int hctr_emb_get_vocabulary_size(hctr_embedding* e, hctr_stream_t stream, size_t* out) {
  HCTR_REQUIRE(e && out, "null pointer");
  return e->ht.count(as_stream(stream), out);
}

size_t hctr_emb_get_max_vocabulary_size(const hctr_embedding* e) {
  return e ? e->p.max_vocabulary_size_per_gpu : 0;
}

size_t hctr_emb_slots_on_rank(const hctr_embedding* e) { return e ? e->buckets_per_sample() : 0; }

// error flags of the hash table: bit 4 = the index stage's grid barrier never opened (its workgroups
// were not resident together); anything else = more distinct keys than rows
static int report_ht_flags(uint32_t f) {
  if ((f & 4u) != 0u) {
    set_error("index stage: the cooperative finish kernel timed out at its grid barrier (device "
              "partitioned or CU-masked below the kernel's grid?); none of the batch's unseen keys got a row, "
              "and no further training batch is taken until hctr_emb_reset has cleared the table");
    return HCTR_ERR_HIP;
  }
  if (f != 0u) {
    // check_overflow, localized_slot_sparse_embedding_hash.hpp:552-569
    set_error("embedding hash table overflow: more distinct keys than max_vocabulary_size_per_gpu");
    return HCTR_ERR_OVERFLOW;
  }
  return HCTR_OK;
}

int hctr_emb_check_overflow(hctr_embedding* e, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  uint32_t f = 0;
  HCTR_TRY(e->ht.error_flags(as_stream(stream), &f));
  return report_ht_flags(f);
}

int hctr_emb_poll_overflow(hctr_embedding* e, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  (void)stream;
  // the flags as the finish kernel of a completed index stage posted them (no copy, no sync)
  return report_ht_flags(*(volatile uint32_t*)e->h_err);
}

int hctr_emb_dump(hctr_embedding* e, int64_t* d_keys, uint64_t* d_slot_id, float* d_vectors,
                  size_t* count, hctr_stream_t stream) {
  HCTR_REQUIRE(e && d_keys && d_vectors && count, "null pointer");
  hipStream_t s = as_stream(stream);
  size_t n = 0;
  HCTR_TRY(e->ht.count(s, &n));
  if (n == 0) {
    *count = 0;
    return HCTR_OK;
  }
  uint64_t* rows = nullptr;
  HCTR_HIP(hipMalloc(&rows, n * sizeof(uint64_t)));
  int rc = e->ht.dump(d_keys, rows, &n, s);
  if (rc == HCTR_OK) {
    const int D = (int)e->p.embedding_vec_size;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * D, 256)), dim3(256), 0, s, rows, n, D,
                       e->table, d_vectors, e->slot_id, d_slot_id);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      set_error("dump gather failed");
      rc = HCTR_ERR_HIP;
    }
  }
  (void)hipFree(rows);
  *count = n;
  return rc;
}

int hctr_emb_load(hctr_embedding* e, const int64_t* d_keys, const uint64_t* d_slot_id,
                  const float* d_vectors, size_t count, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  if (count == 0) return HCTR_OK;
  HCTR_REQUIRE(d_keys && d_vectors, "null pointer");
  hipStream_t s = as_stream(stream);
  size_t head = 0;
  HCTR_TRY(e->ht.value_head(s, &head));
  HCTR_REQUIRE(head + count <= e->p.max_vocabulary_size_per_gpu,
               "load: more rows than max_vocabulary_size_per_gpu");
  // rows head .. head+count-1 in file order, then insert (key -> row) pairs and bump the head
  // (load_parameters, localized_slot_sparse_embedding_hash.cu:383-440: hash_table->insert +
  //  set_value_head)
  uint64_t* rows = nullptr;
  void* keys_typed = nullptr;
  HCTR_HIP(hipMalloc(&rows, count * sizeof(uint64_t)));
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, rows, count,
                     (uint64_t)head);
  int rc = HCTR_OK;
  const void* kptr = d_keys;
  if (e->p.key_type == HCTR_KEY_U32) {
    // narrow int64 file keys to u32
    if (hipMalloc(&keys_typed, count * sizeof(uint32_t)) != hipSuccess) {
      (void)hipFree(rows);
      set_error("hipMalloc failed in load");
      return HCTR_ERR_HIP;
    }
    std::vector<int64_t> h(count);
    std::vector<uint32_t> h32(count);
    (void)hipMemcpyAsync(h.data(), d_keys, count * 8, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    for (size_t i = 0; i < count; i++) h32[i] = (uint32_t)h[i];
    (void)hipMemcpyAsync(keys_typed, h32.data(), count * 4, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
    kptr = keys_typed;
  }
  rc = e->ht.insert(kptr, rows, count, s);
  if (rc == HCTR_OK) {
    const int D = (int)e->p.embedding_vec_size;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(count * D, 256)), dim3(256), 0, s, rows,
                       count, D, d_vectors, e->table, d_slot_id, e->slot_id);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      set_error("load scatter failed");
      rc = HCTR_ERR_HIP;
    }
  }
  if (rc == HCTR_OK) rc = e->ht.set_value_head(head + count, s);
  // a counter posted by a batch enqueued before this load says nothing about the rows now:
  // only posts of later batches re-establish the bound (hipStreamSynchronize above: nothing of
  // this stream is still in flight)
  e->min_valid_seq = e->seq + 1;
  e->upd.row_bound = 0;
  (void)hipFree(rows);
  if (keys_typed) (void)hipFree(keys_typed);
  return rc;
}

int hctr_emb_profiling(hctr_embedding* e, int enable) {
  HCTR_REQUIRE(e, "null handle");
  e->prof.enabled = enable != 0;
  e->prof.reset();
  return HCTR_OK;
}

int hctr_emb_profile_get(hctr_embedding* e, int which, double* total_ms, uint64_t* launches) {
  HCTR_REQUIRE(e && total_ms && launches, "null pointer");
  HCTR_REQUIRE(which >= 0 && which < Profiler::kCats, "which");
  if (e->prof.get(which, total_ms, launches) != 0) {
    set_error("hipEventElapsedTime failed");
    return HCTR_ERR_HIP;
  }
  return HCTR_OK;
}

float* hctr_emb_table_ptr(hctr_embedding* e) { return e ? e->table : nullptr; }
float* hctr_emb_opt_state_ptr(hctr_embedding* e, int k) {
  if (!e) return nullptr;
  return k == 0 ? e->state0 : (k == 1 ? e->state1 : nullptr);
}
const uint64_t* hctr_emb_value_index_ptr(hctr_embedding* e) {
  return e ? e->value_index : nullptr;
}

int hctr_emb_reset(hctr_embedding* e, hctr_stream_t stream) {
  HCTR_REQUIRE(e, "null handle");
  hipStream_t s = as_stream(stream);
  HCTR_TRY(e->ht.clear(s));
  HCTR_HIP(hipStreamSynchronize(s));  // (no post of an older batch can land after this)
  e->min_valid_seq = e->seq + 1;
  *e->h_err = 0u;
  e->upd.row_bound = 0;
  HCTR_TRY(reset_opt_states(e, s));
  e->has_train_batch = false;
  e->top_grad = nullptr;
  return hctr_emb_init_params(e, stream);
}

}  // extern "C"
