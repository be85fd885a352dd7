/*
 * hugectr_amd.h -- C ABI of the MI355X-native sparse-embedding hot path (libhugectr_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Every entry
 * point names the NVIDIA-Merlin/HugeCTR interface it replaces (R = the reference checkout).
 * All device pointers are HIP device pointers on the calling thread's current device; `stream`
 * is a hipStream_t passed as void* (NULL = the default stream).  Nothing here synchronises the
 * host unless the comment says so.  Every function returns 0 on success or a negative
 * hctr_status; hctr_last_error() gives the message for the calling thread.
 *
 * One process drives one GPU ("rank" of "world"), which is how HugeCTR's per-GPU OpenMP threads
 * (R/HugeCTR/include/embeddings/localized_slot_sparse_embedding_hash.hpp:217-283) map onto
 * torch.distributed / RCCL.  The all-to-all between ranks is the caller's (RCCL) -- this library
 * produces/consumes the send/receive buffers in the reference's wire layout.
 */
#ifndef HUGECTR_AMD_H
#define HUGECTR_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hctr_stream_t;

typedef enum {
  HCTR_OK = 0,
  HCTR_ERR_INVALID_ARG = -1,
  HCTR_ERR_HIP = -2,
  HCTR_ERR_OVERFLOW = -3, /* hash table fuller than max_vocabulary_size_per_gpu */
  HCTR_ERR_UNSUPPORTED = -4,
  HCTR_ERR_IO = -5
} hctr_status;

/* Values follow R/HugeCTR/include/common.hpp:82-94,145-149 */
typedef enum { HCTR_OPT_FTRL = 0, HCTR_OPT_ADAM = 1, HCTR_OPT_RMSPROP = 2, HCTR_OPT_ADAGRAD = 3,
               HCTR_OPT_NESTEROV = 4, HCTR_OPT_MOMENTUM_SGD = 5, HCTR_OPT_SGD = 6 } hctr_optimizer_t;
typedef enum { HCTR_UPDATE_LOCAL = 0, HCTR_UPDATE_GLOBAL = 1, HCTR_UPDATE_LAZY_GLOBAL = 2 } hctr_update_t;
typedef enum { HCTR_EMB_DISTRIBUTED_SLOT_HASH = 0, HCTR_EMB_LOCALIZED_SLOT_HASH = 1 } hctr_embedding_t;
typedef enum { HCTR_KEY_U32 = 0, HCTR_KEY_I64 = 1 } hctr_key_t;
typedef enum { HCTR_EMB_F32 = 0, HCTR_EMB_F16 = 1, HCTR_EMB_BF16 = 2 } hctr_emb_dtype_t;

const char* hctr_last_error(void);
int hctr_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Stateless kernels (usable on caller-owned buffers; the handle API below is built on them)   */
/* ------------------------------------------------------------------------------------------ */

/* MurmurHash3_32(key bytes, seed 0): R/HugeCTR/include/hashtable/cudf/hash_functions.cuh:66-107.
 * keys: device [n] of key_type; out: device uint32 [n]. */
int hctr_hash_keys(const void* keys, int key_type, size_t n, uint32_t* out, hctr_stream_t stream);

/* Open-addressing key -> row-index map; replaces HashTable<Key,size_t>
 * (R/HugeCTR/include/hashtable/nv_hashtable.hpp:31-189, src/hashtable/nv_hashtable.cu:169-303).
 * Physical slot count = (size_t)(capacity / 0.75f); slot = murmur(key) % slots; linear probing;
 * new keys receive consecutive indices in order of FIRST OCCURRENCE in `keys` (deterministic,
 * where the reference's atomicAdd order is racy -- see DESIGN.md section 4, q1).  One get_insert call
 * takes at most 2^31 - 1 keys.  A key whose entry holds "no row" (SIZE_MAX, as hctr_ht_insert can
 * write it and a dynamic table's remove() leaves it) is handed a row like an unseen key. */
typedef struct hctr_hashtable hctr_hashtable;
int hctr_ht_create(size_t capacity, int key_type, hctr_hashtable** out);
int hctr_ht_destroy(hctr_hashtable* ht);
int hctr_ht_clear(hctr_hashtable* ht, hctr_stream_t stream);
/* get_insert (train) / get_mark (eval: miss -> SIZE_MAX) / insert (key,val pairs).
 * d_n: optional device uint64 holding the live count (<= n); NULL means n. */
int hctr_ht_get_insert(hctr_hashtable* ht, const void* keys, size_t n, const uint64_t* d_n,
                       uint64_t* value_index, hctr_stream_t stream);
int hctr_ht_get_mark(hctr_hashtable* ht, const void* keys, size_t n, const uint64_t* d_n,
                     uint64_t* value_index, hctr_stream_t stream);
int hctr_ht_insert(hctr_hashtable* ht, const void* keys, const uint64_t* vals, size_t n,
                   hctr_stream_t stream);
/* host-synchronising queries (get_size / get_value_head / get_capacity, nv_hashtable.cu:239-301) */
int hctr_ht_size(hctr_hashtable* ht, hctr_stream_t stream, size_t* out);
int hctr_ht_value_head(hctr_hashtable* ht, hctr_stream_t stream, size_t* out);
int hctr_ht_set_value_head(hctr_hashtable* ht, size_t v, hctr_stream_t stream);
size_t hctr_ht_table_size(const hctr_hashtable* ht);
/* dump occupied (key,val) pairs; d_keys device int64 [>=size], d_vals device uint64; host-syncs */
int hctr_ht_dump(hctr_hashtable* ht, int64_t* d_keys, uint64_t* d_vals, size_t* count,
                 hctr_stream_t stream);
/* Error word of the map (host-syncs): bit 0 = a probe met a full table (the reference: `end()`,
 * only a debug assert, nv_hashtable.cu:61-72), bit 1 = more new keys than free rows (what
 * check_overflow() reports, localized_slot_sparse_embedding_hash.hpp:552-569), bit 2 (value 4) =
 * the cooperative second launch of a get_insert could not get all its workgroups onto the device
 * and gave up: then NO unseen key of that call received a row (all-or-nothing; they read SIZE_MAX)
 * and every later get_insert gives up the same way until hctr_ht_recover or hctr_ht_clear.
 * The reference has no counterpart (its insert is one kernel with a racing atomicAdd). */
int hctr_ht_error_flags(hctr_hashtable* ht, hctr_stream_t stream, uint32_t* out);
/* After error bit 2: `keys` = the keys of every get_insert since the one that gave up.  Puts the
 * map back as it was before those calls (their claimed slots read as erased keys) and clears the
 * bit; the calls can then be issued again and hand out the rows an undisturbed run would have. */
int hctr_ht_recover(hctr_hashtable* ht, const void* keys, size_t n, hctr_stream_t stream);

/* forward_sum / forward_mean: R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu:28-241.
 * out[u,:] = sum_j table[value_index[row_offset[u]+j],:]  (SIZE_MAX index adds 0; combiner 1
 * scales by 1/n when n > 1).  row_offset has key_type elements, [buckets+1]. */
int hctr_forward_pool(size_t buckets, int vec_size, int combiner, const void* row_offset,
                      int key_type, const uint64_t* value_index, const float* table, void* out,
                      int out_dtype, hctr_stream_t stream);
/* SOK lookup_sparse with sp_weights (R/sparse_operation_kit/sparse_operation_kit/lookup.py:425-541):
 * out[b] = sum_j w_j * table[value_index[j]] (weights NULL: w = 1); combiner 1 divides by sum_j w_j.
 * int64 row_offset, fp32 out.  hctr_expand_key_grads is its backward: key_grads[j] =
 * top_grad[bucket(j)] * w_j (/ sum w for mean), one row per key. */
int hctr_forward_pool_weighted(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                               const uint64_t* value_index, const float* weights,
                               const float* table, float* out, hctr_stream_t stream);
int hctr_expand_key_grads(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                          const float* weights, const float* top_grad, float* key_grads,
                          hctr_stream_t stream);
/* same contract; walks each lane group's keys as one flat range (8 row reads in flight whatever the
 * bucket lengths): the kernel of choice for multi-hot buckets. */
int hctr_forward_pool_multihot(size_t buckets, int vec_size, int combiner, const void* row_offset,
                      int key_type, const uint64_t* value_index, const float* table, void* out,
                      int out_dtype, hctr_stream_t stream);
/* same contract with the store address transposed: buckets are numbered lookup * samples + sample
 * (samples * lookups == buckets) and bucket u lands in output row sample * lookups + lookup --
 * the batch-major output of embedding_collection on one GPU, where the reference's reorder after
 * the all-to-all (R/HugeCTR/embedding/operators/network_forward.cu) has nothing to exchange and
 * reduces to this address map; (samples, lookups) = (0, 0) is the identity.  multi_hot != 0
 * picks the flat-range kernel.  one_hot (DEVICE u32, may be NULL): non-zero promises
 * row_offset[i] == i for all i, and the kernel then walks the keys without reading row_offset. */
int hctr_forward_pool_mapped(size_t buckets, int vec_size, int combiner, const void* row_offset,
                             int key_type, const uint64_t* value_index, const float* table,
                             void* out, int out_dtype, int multi_hot, size_t samples,
                             size_t lookups, const uint32_t* one_hot, hctr_stream_t stream);
/* pooling through per-key row pointers -- what embedding::ILookup::lookup(keys, ..., float**
 * embedding_vec) hands to the pooling kernel (R/HugeCTR/embedding/embedding_table.hpp:22-33,
 * generic_lookup.cuh:318-416): rows[j] = device address of key j's fp32 vector, NULL = key not in
 * the table (adds 0, still counts for the mean).  int64 row_offset. */
int hctr_forward_pool_ptrs(size_t buckets, int vec_size, int combiner, const int64_t* row_offset,
                           const float* const* rows, void* out, int out_dtype,
                           hctr_stream_t stream);
/* same with the transposed store of hctr_forward_pool_mapped (one-GPU embedding_collection on
 * dynamic tables, batch-major output); (samples, lookups) = (0, 0) is the identity */
int hctr_forward_pool_ptrs_mapped(size_t buckets, int vec_size, int combiner,
                                  const int64_t* row_offset, const float* const* rows, void* out,
                                  int out_dtype, size_t samples, size_t lookups,
                                  hctr_stream_t stream);

/* forward_reorder / backward_reorder: R/HugeCTR/src/embeddings/forward_reorder_functor.cu:26-98,
 * backward_reorder_functor.cu.  in [gpu][b][slot_in_gpu][D] <-> out [b][slot][D]. */
int hctr_forward_reorder(size_t batch_per_gpu, int slot_num, int vec_size, int gpu_num,
                         const void* in, void* out, int dtype, hctr_stream_t stream);
int hctr_backward_reorder(size_t batch_per_gpu, int slot_num, int vec_size, int gpu_num,
                          const void* in, void* out, int dtype, hctr_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* IEmbedding-shaped handle: replaces LocalizedSlotSparseEmbeddingHash /                       */
/* DistributedSlotSparseEmbeddingHash behind class IEmbedding (R/HugeCTR/include/embedding.hpp */
/* :26-67) with SparseEmbeddingHashParams (:69-93).                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int embedding_type; /* hctr_embedding_t */
  int key_type;       /* hctr_key_t: type of keys AND row offsets, as in the reference */
  int out_dtype;      /* hctr_emb_dtype_t of the pooled output / top gradients */
  size_t train_batch_size;
  size_t evaluate_batch_size;
  size_t max_vocabulary_size_per_gpu;
  size_t embedding_vec_size;
  size_t max_feature_num; /* max keys per sample over all slots */
  size_t slot_num;
  int combiner;                  /* 0 sum, 1 mean */
  const size_t* slot_size_array; /* [slot_num] or NULL */
  /* OptParams, R/HugeCTR/include/optimizer.hpp:149-155 */
  int optimizer;   /* hctr_optimizer_t */
  int update_type; /* hctr_update_t */
  float lr;
  float beta1, beta2, epsilon;     /* Adam (epsilon also AdaGrad) */
  float initial_accu_value;        /* AdaGrad */
  float momentum_factor;           /* MomentumSGD factor / Nesterov mu */
  int atomic_update;               /* SGD: 1 = fp32 atomicAdd path (optimizer_wrapper.hpp:40) */
  float scaler;                    /* loss scaler the gradients are divided by */
  /* placement: this process is GPU `rank` of `world` (global ids, resource_manager semantics) */
  int rank;
  int world;
  uint64_t seed; /* table initialisation */
} hctr_embedding_params;

typedef struct hctr_embedding hctr_embedding;

int hctr_emb_create(const hctr_embedding_params* params, hctr_embedding** out);
int hctr_emb_destroy(hctr_embedding* emb);

/* IEmbedding::init_params: uniform(+-sqrt(1/slot_size)) per slot, or +-0.05 without slot sizes
 * (R/HugeCTR/src/embeddings/init_embedding_functor.cu:24-55) */
int hctr_emb_init_params(hctr_embedding* emb, hctr_stream_t stream);

/* IEmbedding::forward(is_train) up to (not including) the all-to-all.
 * row_offset [batch*slot_num+1], keys [nnz]: the FULL-batch CSR every rank receives from the
 * reader (R/HugeCTR/src/data_readers/data_collector.cu:86-113).  Localized: filter slots with
 * slot % world == rank, resolve + pool -> out [batch][slots_on_rank][D] (== the all-to-all send
 * buffer [peer][batch/world][slots_on_rank][D]).  Distributed: filter keys with key % world ==
 * rank -> partial sums out [batch][slot_num][D] (reduce-scatter input). */
int hctr_emb_forward(hctr_embedding* emb, int is_train, const void* row_offset, const void* keys,
                     size_t nnz, void* out, hctr_stream_t stream);

/* SparseEmbeddingFunctors::forward_scale (R/HugeCTR/src/embeddings/forward_scale_functor.cu:28-77,
 * called at distributed_slot_sparse_embedding_hash.hpp:181-197): distributed embedding with
 * combiner mean on world > 1 GPUs.  hctr_emb_forward then writes partial SUMS; after the caller's
 * reduce-scatter this divides out_local [batch/world][slot_num][D] (this rank's samples) in place
 * by each bucket's key count over ALL GPUs (n > 1 only; 16-bit + even D: scaler rounded to the
 * type first, the reference's align2 kernel).  The count comes from the full-batch row offsets
 * hctr_emb_forward was given -- every rank holds them, so the reference's all_reduce(row_offsets)
 * (all_reduce_functor.cu:55) needs no collective here.  backward / update_params divide the
 * gradients by the same global counts (hpp:216-221).  No-op in every other configuration. */
int hctr_emb_forward_scale(hctr_embedding* emb, int is_train, void* out_local,
                           hctr_stream_t stream);

/* IEmbedding::backward after the all-to-all: top_grad has the layout of forward's `out`.
 * Zero-copy: only records the pointer; the gradient must stay valid until update_params. */
int hctr_emb_backward(hctr_embedding* emb, const void* top_grad, hctr_stream_t stream);
/* materialise wgrad (backward_sum/backward_mean, backward_functor.cu:26-104) for inspection */
int hctr_emb_get_wgrad(hctr_embedding* emb, void* wgrad, hctr_stream_t stream);

/* IEmbedding::update_params: EmbeddingOptimizer::update (R/HugeCTR/src/optimizers/
 * sparse_optimizer.cu:622-864) fused with backward: sort by row, per-row ordered reduce,
 * optimizer math on weights + state.  No host synchronisation. */
int hctr_emb_update_params(hctr_embedding* emb, hctr_stream_t stream);

int hctr_emb_set_learning_rate(hctr_embedding* emb, float lr);
/* host-synchronising queries */
int hctr_emb_get_vocabulary_size(hctr_embedding* emb, hctr_stream_t stream, size_t* out);
size_t hctr_emb_get_max_vocabulary_size(const hctr_embedding* emb);
size_t hctr_emb_slots_on_rank(const hctr_embedding* emb);
int hctr_emb_check_overflow(hctr_embedding* emb, hctr_stream_t stream);
/* IEmbedding::check_overflow as Model::train calls it on every iteration
 * (R/HugeCTR/src/pybind/model.cpp:1088) without its host synchronisation: reports
 * HCTR_ERR_OVERFLOW once an earlier, already completed device-to-host copy of the table's error
 * flags shows an overflow (at most two calls late), then queues the next copy on `stream`.  Safe
 * to be late: keys that found no free row resolve to "no row" (pooled as zeros, skipped by the
 * update) -- nothing is read or written outside the table. */
int hctr_emb_poll_overflow(hctr_embedding* emb, hctr_stream_t stream);

/* dump_parameters / load_parameters (buffer form, R/.../localized_slot_sparse_embedding_hash.cu
 * :383-440,1260-1340): keys int64, slot_id size_t (localized), emb_vector fp32, all DEVICE
 * buffers of capacity >= vocabulary size; *count returns rows written.  Host-synchronising. */
int hctr_emb_dump(hctr_embedding* emb, int64_t* d_keys, uint64_t* d_slot_id, float* d_vectors,
                  size_t* count, hctr_stream_t stream);
int hctr_emb_load(hctr_embedding* emb, const int64_t* d_keys, const uint64_t* d_slot_id,
                  const float* d_vectors, size_t count, hctr_stream_t stream);
/* raw device views (owned by the handle): table [max_vocab][D] fp32, optimizer state k
 * [max_vocab][D] -- fp32, or __half when out_dtype is HCTR_EMB_F16: the reference's
 * OptimizerTensor<TypeEmbeddingComp> (R/HugeCTR/include/optimizer.hpp:284-296) */
float* hctr_emb_table_ptr(hctr_embedding* emb);
float* hctr_emb_opt_state_ptr(hctr_embedding* emb, int k);
const uint64_t* hctr_emb_value_index_ptr(hctr_embedding* emb);
int hctr_emb_reset(hctr_embedding* emb, hctr_stream_t stream);

/* Measurement support (bench.py roofline leg; no reference counterpart): when enabled, hipEvents
 * are recorded on the launch stream around  which = 0 the gather+pool kernel, 1 the hash/index
 * stage, 2 the radix sort, 3 the segmented reduce + optimizer kernels.  profile_get host-syncs. */
int hctr_emb_profiling(hctr_embedding* emb, int enable);
int hctr_emb_profile_get(hctr_embedding* emb, int which, double* total_ms, uint64_t* launches);

/* ------------------------------------------------------------------------------------------ */
/* embedding_collection (EBC / SparseOperationKit) on static tables                             */
/* ------------------------------------------------------------------------------------------ */
/* KeysToIndicesConverter::convert (R/HugeCTR/embedding/operators/keys_to_indices.cu:24-43):
 * idx = table_start + key / num_shards */
int hctr_ebc_keys_to_indices(const void* keys, int key_type, size_t n, int64_t table_start,
                             int num_shards, uint64_t* out, hctr_stream_t stream);

/* embedding::ILookup::lookup(keys, num_keys, num_keys_per_table_offset, num_table_offset,
 * table_id_list, float** embedding_vec) (R/HugeCTR/embedding/embedding_table.hpp:22-33) for the
 * STATIC table, RaggedStaticEmbeddingTable::lookup (R/HugeCTR/embedding_storage/
 * ragged_static_embedding.cu:33-51,553-575): embedding_vec[i] = address of key i's fp32 vector
 * inside the caller-owned flat table.  keys = indices as hctr_ebc_keys_to_indices numbers them
 * (key_type HCTR_KEY_U32 / HCTR_KEY_I64, or 2 = the uint64 that function writes); position i
 * belongs to table_id_list[t] for num_keys_per_table_offset[t] <= i < [t + 1].  The table is
 * described as the reference's is: local_table_ids [n] ascending, table_index_start [n + 1] (first
 * index of each table and the end), table_ev_offset [n] (element offset of each table in
 * emb_table), local_ev_sizes [n].  All arrays DEVICE.  *d_error (device uint32, caller-zeroed): bit
 * 0 = a position names a table this shard does not hold, bit 1 = index outside the table; such
 * positions get NULL (hctr_forward_pool_ptrs adds 0 for NULL).  The dynamic table's counterpart is
 * hctr_det_lookup_rows. */
int hctr_static_lookup(const void* keys, int key_type, size_t num_keys,
                       const uint32_t* num_keys_per_table_offset, size_t num_table_offset,
                       const int32_t* table_id_list, const int32_t* local_table_ids,
                       size_t num_local_tables, const uint64_t* table_index_start,
                       float* emb_table, const uint64_t* table_ev_offset,
                       const int32_t* local_ev_sizes, float** embedding_vec, uint32_t* d_error,
                       hctr_stream_t stream);

/* Key routing on the gathered global CSR (DataDistributor semantics,
 * R/HugeCTR/embedding/data_distributor/key_filtering_operators.cu:37-300, with the SOK
 * all-gather flow): for every lookup resolved on this rank keep the keys with
 * key % num_shards == shard_id and convert them to row indices of the rank's flat table.
 *   keys / bucket_range: feature-major global batch, bucket = lookup * batch + b, [L*batch+1]
 *   lookup_desc (DEVICE int32): {global_lookup, num_shards, shard_id} x num_local_lookups
 *   row_start   (DEVICE int64): first row of each local lookup's table shard; a negative value
 *                 marks a dynamic table: the key itself is written to out_indices
 * Output buckets are ordered [peer][local lookup][b_local] (= the all-to-all send layout):
 *   out_bucket_range int64 [world * num_local_lookups * batch/world + 1], out_indices [<= nnz]. */
/* The reference's own key route (DataDistributor, sparse_data_distribution_op_impl.cu:215-395) for
 * data-parallel input is two all-to-alls: bucket lengths, then keys.  Sender: one
 * hctr_ebc_route_keys call per destination on the LOCAL batch (batch = batch/world, world = 1,
 * the destination's lookup_desc, row_start < 0) yields that destination's lengths and keys.
 * Receiver: the received keys are already in its bucket order [source][local lookup][b_local];
 * out_bucket_range = prefix sum of the received lengths, and this call turns the keys into rows of
 * its flat table in place (keys_to_indices.cu:31-42; lookups with row_start < 0 keep the key). */
int hctr_ebc_routed_keys_to_indices(size_t batch_per_gpu, int world, int num_local_lookups,
                                    const int32_t* lookup_desc, const int64_t* row_start,
                                    const int64_t* out_bucket_range, uint64_t* keys_inout,
                                    hctr_stream_t stream);
size_t hctr_ebc_route_workspace_bytes(size_t batch, int num_local_lookups);
int hctr_ebc_route_keys(size_t batch, int world, int num_local_lookups, const int32_t* lookup_desc,
                        const int64_t* row_start, const void* keys, const void* bucket_range,
                        int key_type, int64_t* out_bucket_range, uint64_t* out_indices,
                        uint64_t* d_nnz, void* workspace, hctr_stream_t stream);
/* hctr_ebc_route_keys for the one-GPU case in which this rank owns every lookup whole (world = 1,
 * num_shards = 1 everywhere, lookup_desc = identity): nothing is filtered, so the output CSR is the
 * input CSR and the three passes (count, scan, index) are one.  one_hot (DEVICE u32, may be NULL)
 * is left non-zero iff every bucket holds exactly its own key (bucket_range[i] == i) -- the flag
 * hctr_forward_pool_mapped takes.  nnz_hint: the host's key count (or upper bound; 0 = unknown),
 * a launch-shape hint only. */
int hctr_ebc_route_whole(size_t batch, int num_lookups, const int64_t* row_start, const void* keys,
                         const void* bucket_range, int key_type, int64_t* out_bucket_range,
                         uint64_t* out_indices, uint64_t* d_nnz, uint32_t* one_hot,
                         size_t nnz_hint, hctr_stream_t stream);
/* total key count of each (lookup, local sample) bucket of this rank: Average divides by it on
 * the receiving side (R/HugeCTR/embedding/operators/network_forward.cu:272-283) */
int hctr_ebc_bucket_counts(size_t batch, int world, int rank, int num_lookup,
                           const void* bucket_range, int key_type, int64_t* counts,
                           hctr_stream_t stream);
/* One GPU (nothing to exchange): the Average arithmetic of NetworkForward (forward != 0: the
 * pooled sum, already rounded to the vector type, divided by the bucket's key count, rounded again --
 * network_forward.cu:272-292) or of NetworkBackward (forward == 0) applied in place to the
 * [lookup][b][ev] / [b][lookup][ev] output (its gradient); vectors of Sum lookups and of buckets
 * with at most one key are not touched. */
int hctr_ebc_scale_average(size_t batch_per_gpu, int num_lookup, int ev_size,
                           const int32_t* d_combiner, const int64_t* d_bucket_counts,
                           int batch_major, void* data, int dtype, int forward,
                           hctr_stream_t stream);
/* NetworkForward / NetworkBackward: blocks of [batch_per_gpu][ev] vectors, one per (source rank,
 * its local lookup); d_src_blocks[l * max_shards + s] = block of shard s of lookup l or -1.
 * out / grad layout: feature-major [lookup][b][ev] or batch-major [b][lookup][ev]. */
int hctr_ebc_network_forward(size_t batch_per_gpu, int num_lookup, int ev_size, int max_shards,
                             const int32_t* d_src_blocks, const int32_t* d_combiner,
                             const int64_t* d_bucket_counts, int batch_major, const void* recv,
                             void* out, int dtype, hctr_stream_t stream);
int hctr_ebc_network_backward(size_t batch_per_gpu, int num_lookup, int ev_size, int max_shards,
                              const int32_t* d_src_blocks, const int32_t* d_combiner,
                              const int64_t* d_bucket_counts, int batch_major, const void* grad,
                              void* send, int dtype, hctr_stream_t stream);

/* IGroupedEmbeddingTable::update on a caller-owned flat fp32 table
 * (R/HugeCTR/embedding_storage/ragged_static_embedding.cu:275-353,593-700): sort + segmented
 * reduce + optimizer, same kernels as hctr_emb_update_params.  indices[nnz] row per key,
 * bucket_range int64 [buckets+1], grad [buckets][vec]. */
typedef struct hctr_updater hctr_updater;
int hctr_updater_create(size_t max_nnz, size_t max_rows, int vec_size, hctr_updater** out);
int hctr_updater_destroy(hctr_updater* u);
/* Ftrl hyper-parameters for optimizer = HCTR_OPT_FTRL (FtrlOptimizer,
 * R/HugeCTR/embedding_storage/ragged_static_embedding.cu:159-290): state0 = n, state1 = z */
int hctr_updater_set_ftrl(hctr_updater* u, float lambda1, float lambda2, float beta);
/* One-GPU embedding_collection with a batch-major ([sample][lookup][vec]) output: buckets are
 * numbered lookup * samples + sample, their gradient row is sample * lookups + lookup of `grad`.
 * The reference transposes in a pass of its own on both sides of the all-to-all
 * (R/HugeCTR/embedding/operators/network_backward.cu); with one GPU there is no exchange, so the
 * transpose is an address computation of the update (and of hctr_forward_pool_mapped).  Sum
 * combiner only; (0, 0) switches the map off. */
int hctr_updater_set_grad_map(hctr_updater* u, size_t samples, size_t lookups);
/* rows handed out so far are < rows (0 = unknown: max_rows of hctr_updater_create): the row sort
 * of the next updates covers log2(rows) bits only.  For a table that grows (hctr_det_row_store). */
int hctr_updater_set_row_bound(hctr_updater* u, uint64_t rows);
int hctr_updater_update(hctr_updater* u, size_t buckets, size_t nnz, const int64_t* bucket_range,
                        const uint64_t* indices, const void* grad, int grad_dtype, int optimizer,
                        int update_type, float lr, float beta1, float beta2, float epsilon,
                        float momentum_factor, float scaler, uint64_t times, float* table,
                        float* state0, float* state1, hctr_stream_t stream);

/* cub::DeviceRadixSort::SortPairs as the sparse optimizer calls it (R/HugeCTR/src/optimizers/
 * sparse_optimizer.cu:657-676: keys = row indices, values = bucket ids, bits [0, end_bit)): a stable
 * LSD radix sort of (uint32 key, uint32 value) pairs sized for <= 2^24 pairs -- three short
 * launches per 10-bit digit, no inter-workgroup waits (csrc/radix_sort.hip).  Sorts by key bits
 * [0, 10 * ceil(end_bit / 10)); inputs are left untouched.  temp: DEVICE workspace of
 * hctr_radix_sort_temp_bytes(n) bytes.  hctr_emb_update_params / hctr_updater_update use it
 * internally. */
size_t hctr_radix_sort_temp_bytes(size_t n);
int hctr_radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in,
                              uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                              size_t n, int end_bit, hctr_stream_t stream);

/* LocalReduceIndexCalculation + LocalReduce (R/HugeCTR/embedding/operators/index_calculation.cu,
 * model_backward.cu:113-...): the Wgrad{unique_keys, ev_start_indices, data} a grouped table's
 * update() consumes (R/HugeCTR/embedding/common.hpp:352-373).  row_ids[nnz] < = max_row_id identify
 * the row of every key (any numbering that is unique per (table, key), e.g. hctr_det_lookup_rows);
 * grad [buckets][vec] of grad_dtype.  Outputs, ordered by ascending row id: unique_row_ids
 * [<= nnz], unique_keys [<= nnz] (= keys[first position of the row]; both NULL to skip), wgrad
 * [*num_unique][vec] fp32 sums in ascending bucket order (ev_start_indices = i * vec).
 * *num_unique is returned on the HOST (one stream synchronisation, where the reference reads
 * num_unique_keys).  The updater must have been created with max_nnz >= nnz. */
int hctr_ebc_local_reduce(hctr_updater* u, size_t buckets, size_t nnz, const int64_t* bucket_range,
                          const uint64_t* row_ids, uint64_t max_row_id, const uint64_t* keys,
                          const void* grad, int grad_dtype, size_t* num_unique,
                          uint64_t* unique_row_ids, uint64_t* unique_keys, float* wgrad,
                          hctr_stream_t stream);

/* ---- unique-row exchange (multi-GPU, one key per bucket): ship every distinct row once per
 * destination GPU + an (index, bucket) pair per position, return per-row gradient sums instead of
 * per-sample gradients.  Replaces the payload of all2all_forward / all2all_backward
 * (R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264) when keys repeat; the collective
 * itself stays the caller's.  See hugectr_amd/parallel.py:UniqueExchange for the call order. */
typedef struct hctr_uniq hctr_uniq;
int hctr_uniq_create(size_t max_positions, hctr_uniq** out);
int hctr_uniq_destroy(hctr_uniq* u);
/* owner: value_index[positions] rows of the pooled layout [world][batch_per_gpu][slots_local] ->
 * meta[positions][2] = (unique index inside the peer segment, bucket b_local * slots_total +
 * s_global on the receiver), sorted by row inside each peer segment; urow = distinct rows,
 * peer-major; peer_off[world + 1] (device) = offsets of the peers' segments in urow */
int hctr_uniq_plan(hctr_uniq* u, size_t positions, size_t positions_per_peer, int batch_per_gpu,
                   int slots_local, int slots_total, int rank, int world,
                   const uint64_t* value_index, uint64_t max_rows, uint32_t* meta, uint64_t* urow,
                   int64_t* peer_off, hctr_stream_t stream);
int hctr_uniq_gather_rows(size_t n_rows, int vec_size, const uint64_t* urow, const float* table,
                          void* out, int out_dtype, hctr_stream_t stream);
/* receiver: out[bucket] = rows[r_off[owner] + index]; owner j holds positions [q_off[j], q_off[j+1])
 * (device int64 arrays); also emits the globally numbered sorted (row, bucket) list for
 * hctr_updater_reduce_presorted.  out may be NULL when row_of (uint32 [buckets], bucket -> row of
 * the received table) is requested instead: hctr_interaction_*_indexed read the rows through it and
 * the expanded tensor is never materialised. */
int hctr_uniq_expand(size_t positions, int n_owners, const int64_t* q_off, const int64_t* r_off,
                     const uint32_t* meta, const void* rows, int vec_size, int dtype, void* out,
                     uint32_t* sorted_rows, uint32_t* sorted_buckets, uint32_t* row_of,
                     hctr_stream_t stream);
/* receiver backward: out_sum[row] = sum over the row's run of grad[bucket] (ascending position,
 * fp32), row_offset = int64 [buckets + 1] with row_offset[buckets] == positions */
int hctr_updater_reduce_presorted(hctr_updater* u, size_t positions, size_t buckets,
                                  const int64_t* row_offset, const uint32_t* sorted_rows,
                                  const uint32_t* sorted_buckets, const void* grad, int grad_dtype,
                                  size_t n_rows, float* out_sum, hctr_stream_t stream);
/* owner: index stage only (hctr_emb_forward without the gather), and the sparse update driven by
 * (row, gradient) entries: entry i updates rows[i] with grad[i][:] (row_offset = arange(n + 1)) */
int hctr_emb_index(hctr_embedding* emb, int is_train, const void* row_offset, const void* keys,
                   size_t nnz, hctr_stream_t stream);
/* One GPU, inter-iteration overlap (solver.train_inter_iteration_overlap; the reference resolves the
 * next batch's keys while the current iteration computes, R/HugeCTR/src/pybind/model_pipeline.cpp:
 * 299-346): the index stage of the batch AFTER the current training batch, into a second set of
 * buffers -- the current batch (its rows, its pending backward / update_params) stays what it is.
 * `stream` must be ordered behind the previous index stage (the hash table is shared state).
 * hctr_emb_index_adopt makes that batch the current one (host-side pointer swap, no launch); the
 * caller orders its stream behind index_ahead's before reading the rows. */
int hctr_emb_index_ahead(hctr_embedding* emb, const void* row_offset, const void* keys, size_t nnz,
                         hctr_stream_t stream);
int hctr_emb_index_adopt(hctr_embedding* emb);
int hctr_emb_update_rows(hctr_embedding* emb, size_t n, const int64_t* row_offset,
                         const uint64_t* rows, const void* grad, int grad_dtype,
                         hctr_stream_t stream);

/* ------------------------------------------------------------------------------------------ */
/* Dense ops on the path                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* InteractionLayer<T>::fprop / bprop (R/HugeCTR/src/layers/interaction_layer.cu:1046-1237).
 * mlp [B][W], emb [B][n_emb][W] -> out [B][W + n_ins(n_ins-1)/2 + 1], n_ins = n_emb+1; pairs
 * row-major over the strict lower triangle, last column zero. dtype: f32/f16/bf16 */
int hctr_interaction_fwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         void* out, int dtype, hctr_stream_t stream);
int hctr_interaction_bwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         const void* top_grad, void* mlp_grad, void* emb_grad, int dtype,
                         hctr_stream_t stream);

/* Interaction on a table of distinct rows: embedding s of sample b is rows[row_of[b * n_emb + s]]
 * (unique-row exchange).  16-bit dtypes, width 32/64/128, n_emb <= 31; emb_grad is the dense
 * [batch][n_emb][width] gradient. */
int hctr_interaction_fwd_indexed(size_t batch, int n_emb, int width, const void* mlp,
                                 const void* rows, const uint32_t* row_of, void* out, int dtype,
                                 hctr_stream_t stream);
int hctr_interaction_bwd_indexed(size_t batch, int n_emb, int width, const void* mlp,
                                 const void* rows, const uint32_t* row_of, const void* top_grad,
                                 void* mlp_grad, void* emb_grad, int dtype, hctr_stream_t stream);
/* the same when row_of is a BIJECTION onto the rows (the reorder map of the localized embedding's
 * all-to-all receive buffer, forward_reorder_functor.cu:43-57): the embedding gradient of (b, s) is
 * written to row row_of[b * n_emb + s] of grad_rows [batch * n_emb][width] -- the all-to-all send
 * layout of backward_reorder -- so neither reorder pass exists as a kernel */
int hctr_interaction_bwd_indexed_scatter(size_t batch, int n_emb, int width, const void* mlp,
                                         const void* rows, const uint32_t* row_of,
                                         const void* top_grad, void* mlp_grad, void* grad_rows,
                                         int dtype, hctr_stream_t stream);

/* Gather fused into InteractionLayer::fprop (one GPU, ONE key per bucket, sum combiner): the
 * pooled vector of a one-hot bucket is its table row rounded to the 16-bit type
 * (forward_sum_kernel, R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu:28-75, followed by
 * InteractionLayer<__half>::fprop, R/HugeCTR/src/layers/interaction_layer.cu:1046-1127).  Rows are
 * read through value_index ([batch * n_emb] row numbers, ~0 = no row -> zeros) straight into the
 * interaction's LDS tile; `pooled` [batch][n_emb][width] is written once (the backward needs it);
 * out as hctr_interaction_fwd.  Bit-identical to hctr_emb_forward + hctr_interaction_fwd.
 * 16-bit dtypes, width 16/32/64/128, n_emb <= 31. */
int hctr_interaction_fwd_gather(size_t batch, int n_emb, int width, const void* mlp,
                                const float* table, const uint64_t* value_index, void* pooled,
                                void* out, int dtype, hctr_stream_t stream);
/* the same on an embedding handle whose index stage has run (hctr_emb_index): its table, its
 * value_index, its vector size and output type; world = 1, sum combiner (or mean: one key), the
 * batch of `is_train`.  Timed as the handle's gather stage (hctr_emb_profile_get stage 0). */
int hctr_emb_forward_interaction(hctr_embedding* e, int is_train, const void* mlp, void* pooled,
                                 void* out, hctr_stream_t stream);

/* MultiCrossLayer<T> v1 (projection_dim = 0): x_{l+1} = x0 * (x_l . w_l) + b_l + x_l
 * (R/HugeCTR/src/layers/multi_cross_layer.cu:582-601,1023-1060).  kernels/biases [layers][w];
 * outputs [layers][B][w] (outputs[layers-1] is the layer output), hiddens [layers][B]. */
int hctr_cross_v1_fwd(size_t batch, int width, int layers, const float* x0, const float* kernels,
                      const float* biases, float* outputs, float* hiddens, hctr_stream_t stream);
int hctr_cross_v1_bwd(size_t batch, int width, int layers, const float* x0, const float* kernels,
                      const float* outputs, const float* hiddens, const float* out_grad,
                      float* in_grad, float* kernel_grads, float* bias_grads, float* workspace,
                      hctr_stream_t stream);
size_t hctr_cross_v1_bwd_workspace_bytes(size_t batch, int width, int layers);

/* ---- dynamic embedding table (EBC "dynamic" tables / SOK DynamicVariable backend) -------------
 * det::DynamicEmbeddingTable<Key, float>
 * (R/third_party/dynamic_embedding_table/dynamic_embedding_table.hpp:25-66): num_classes maps key ->
 * fp32 vector of dimension_per_class[c]; maps grow on demand.  Keys of one call are grouped by id
 * space: id_spaces[i] (class index) owns keys[id_space_offsets[i] .. id_space_offsets[i+1]) -- both
 * arrays live on the HOST, as in the reference; keys / elements are device pointers.  elements of
 * consecutive keys are packed back to back with each key's own dimension.
 * initializer: "ones" | "zeros" | a float literal | anything else (also "") = uniform (0, 1]
 * (dynamic_embedding_table.cu:66-84; the reference seeds curand from std::random_device, here the
 * value is a pure function of (seed, class, row, element)). */
typedef struct hctr_det hctr_det;
int hctr_det_create(size_t num_classes, const size_t* dimension_per_class, const char* initializer,
                    size_t initial_capacity_per_class, int key_type, uint64_t seed,
                    hctr_det** out);
int hctr_det_destroy(hctr_det* h);
size_t hctr_det_num_classes(const hctr_det* h);
/* lookup: unseen keys are inserted and initialised, then all vectors are copied out */
int hctr_det_lookup(hctr_det* h, const void* keys, float* elements, size_t num_keys,
                    const size_t* id_spaces, const size_t* id_space_offsets, size_t num_id_spaces,
                    hctr_stream_t stream);
/* lookup_unsafe: device pointers to the stored vectors (valid until the next inserting call) */
int hctr_det_lookup_unsafe(hctr_det* h, const void* keys, float** elements, size_t num_keys,
                           const size_t* id_spaces, const size_t* id_space_offsets,
                           size_t num_id_spaces, hctr_stream_t stream);
/* scatter_add / scatter_update: keys that are not in the table are skipped
 * (cuco/detail/dynamic_map_kernels.cuh:143-183) */
int hctr_det_scatter_add(hctr_det* h, const void* keys, const float* elements, size_t num_keys,
                         const size_t* id_spaces, const size_t* id_space_offsets,
                         size_t num_id_spaces, hctr_stream_t stream);
int hctr_det_scatter_update(hctr_det* h, const void* keys, const float* elements, size_t num_keys,
                            const size_t* id_spaces, const size_t* id_space_offsets,
                            size_t num_id_spaces, hctr_stream_t stream);
int hctr_det_remove(hctr_det* h, const void* keys, size_t num_keys, const size_t* id_spaces,
                    const size_t* id_space_offsets, size_t num_id_spaces, hctr_stream_t stream);
/* eXport: up to num_keys (key, vector) pairs of one class; *exported = how many (host sync) */
int hctr_det_export(hctr_det* h, size_t class_index, void* keys, float* values, size_t num_keys,
                    size_t* exported, hctr_stream_t stream);
/* row indices of keys in one class's row store (insert != 0: unseen keys are inserted and
 * initialised; else unseen -> SIZE_MAX) and the store itself ([capacity][dim] fp32; the pointer is
 * fixed for the table's life -- growth maps memory behind it): lets the path's gather / update
 * kernels run on dynamic tables */
int hctr_det_lookup_index(hctr_det* h, size_t class_index, const void* keys, size_t num_keys,
                          int insert, uint64_t* row_index, hctr_stream_t stream);
int hctr_det_rows(hctr_det* h, size_t class_index, float** rows, size_t* capacity);
/* Storage: every class owns a fixed region of ONE reserved address range per table, stride_rows
 * (a power of two >= 2^24, classes * stride_rows < 2^32 - 16) rows long, and physical memory is
 * mapped behind the rows it holds as it grows (hipMemAddressReserve / hipMemMap) -- where the
 * reference's cuCollections map adds sub-maps
 * (R/third_party/dynamic_embedding_table/dynamic_embedding_table.cu), here a row never moves, a
 * growth step copies nothing and allocates its increment only.  Limits that follow: at most 255
 * classes per table, vectors of at most 16384 floats, stride_rows rows per class.
 * A table whose classes share ONE dimension (an embedding_collection group: one ev_size): class c
 * owns rows [class_row_base[c], class_row_base[c] + capacity_c), class_row_base[c] = c *
 * stride_rows, so the table-wide row numbers of
 * hctr_det_lookup_rows index one flat (sparsely backed) [total_rows][dim] fp32 table: the static tables' gather
 * (hctr_forward_pool*) and sparse update (hctr_updater_update) then run on a dynamic table as they
 * are -- embedding::DynamicEmbeddingTable::lookup + update
 * (R/HugeCTR/embedding_storage/dynamic_embedding.cu:130-330) without the pointer list, the unique
 * list and the wgrad buffer in between.  *rows = NULL when the classes differ in dimension.  The
 * pointer, every class's hctr_det_rows pointer, class_row_base and every row number handed out
 * stay valid for the table's life; only rows below a class's capacity have memory behind them. */
int hctr_det_row_store(hctr_det* h, float** rows, uint64_t* total_rows);
/* Optimizer state of the flat row store: num_state (1 or 2) arrays [total_rows][dim] fp32 that
 * share the row numbers (and the backing, piece for piece) of hctr_det_row_store, created
 * (zero-filled, on `stream`) by the first call that asks for them; memory mapped behind a class that
 * grows is zero-filled as well, the addresses never change.  A row nobody updated yet
 * holds zeros -- what the reference's state table (a second DynamicEmbeddingTable with the "zeros"
 * initializer, keyed like the weights: embedding::DynamicEmbeddingTable::update,
 * R/HugeCTR/embedding_storage/dynamic_embedding.cu:227-317) hands out for a key it meets first.
 * With them hctr_updater_update runs AdaGrad / Adam / MomentumSGD on a dynamic table as on a
 * static one: one probe per key (the forward's) instead of three.  state1 may be NULL when
 * num_state == 1.  hctr_det_clear zeroes the state; a key that is removed and met again starts
 * from zero state (its row number is a new one). */
int hctr_det_state_store(hctr_det* h, int num_state, float** state0, float** state1,
                         hctr_stream_t stream);
/* embedding::DynamicEmbeddingTable::lookup (R/HugeCTR/embedding_storage/dynamic_embedding.cu:
 * 130-160): keys grouped by id space (HOST id_spaces / id_space_offsets as in hctr_det_lookup) ->
 * per key the address of its vector (elements, may be NULL) and / or a row number that is unique
 * over all classes: class_row_base[class] + row inside the class (row_index, may be NULL; the
 * sort key of hctr_ebc_local_reduce).  class_row_base (HOST, num_classes + 1 entries, may be NULL)
 * receives the bases used by this call = running sum of the class capacities after any growth.
 * insert != 0: unseen keys are inserted and initialised first (the training lookup); else unseen
 * keys give NULL / SIZE_MAX.  Pointers and row numbers stay valid until the next inserting call.
 * The id spaces must tile keys[0, num_keys) without gaps.  All of them are probed by ONE launch;
 * only classes that met an unseen key take the inserting path (one stream synchronisation to learn
 * which, when insert != 0). */
int hctr_det_lookup_rows(hctr_det* h, const void* keys, size_t num_keys, const size_t* id_spaces,
                         const size_t* id_space_offsets, size_t num_id_spaces, int insert,
                         float** elements, uint64_t* row_index, uint64_t* class_row_base,
                         hctr_stream_t stream);
int hctr_det_clear(hctr_det* h, hctr_stream_t stream);
int hctr_det_size_per_class(hctr_det* h, size_t* sizes, hctr_stream_t stream); /* host sync */
int hctr_det_capacity_per_class(const hctr_det* h, size_t* capacities);
/* Inserting lookups whose hash index gave up (hctr_ht_error_flags bit 2) and were repaired and
 * issued again inside the call, since create.  Every inserting hctr_det_* call verifies its index
 * before it uses a row (one wait for the stream); an error it cannot repair is HCTR_ERR_HIP, never
 * a silently missing row. */
int hctr_det_repair_count(const hctr_det* h, uint64_t* out);

/* embedding::DynamicEmbeddingTable::update (R/HugeCTR/embedding_storage/dynamic_embedding.cu:176-330,
 * optimizers.cuh:29-233): optimizer step on the unique keys of a batch.  wgrad holds the summed
 * gradient of key k at [ev_start_indices[k], +dim) (device, uint32 offsets).  `states` is a second
 * table whose class dimensions are dim * {1: momentum/nesterov/adagrad/rmsprop, 2: adam (m|v),
 * ftrl (n|z)} created with initializer "zeros"; NULL for SGD.  The Adam step counter lives in
 * `weights` and is incremented by every Adam call, as the reference does. */
typedef struct {
  int optimizer; /* hctr_optimizer_t */
  float lr, beta1, beta2, epsilon; /* adam; adagrad / rmsprop use epsilon */
  float momentum_factor;           /* momentum: factor, nesterov: mu */
  float rmsprop_beta;
  float ftrl_lambda1, ftrl_lambda2, ftrl_beta;
  float scaler;
} hctr_det_opt_params;
int hctr_det_update(hctr_det* weights, hctr_det* states, const hctr_det_opt_params* p,
                    const void* unique_keys, size_t num_unique_keys, const size_t* id_spaces,
                    const size_t* id_space_offsets, size_t num_id_spaces,
                    const uint32_t* ev_start_indices, const float* wgrad, hctr_stream_t stream);

/* MLP helper around the path (MLPLayer bprop, R/HugeCTR/src/layers/mlp_layer.cu): fused
 * dz = dy * (y > 0) and db[n] = sum_rows dz (deterministic two-stage column sum); 16-bit tensors
 * [rows][n], n % 8 == 0; workspace >= hctr_relu_bwd_bias_workspace_bytes. */
size_t hctr_relu_bwd_bias_workspace_bytes(size_t rows, int n);
int hctr_relu_bwd_bias(size_t rows, int n, const void* dy, const void* y, void* dz, float* db,
                       float* workspace, int dtype, hctr_stream_t stream);

/* out[i] = sum_g in[g][i], i < n (n % 8 == 0): fixed-order reduction of the 16-bit partial products of
 * a split-K weight-gradient GEMM into fp32. */
int hctr_sum_groups(int groups, size_t n, const void* in, int dtype, float* out,
                    hctr_stream_t stream);

/* dense SGD on a flat fp32 parameter buffer fused with the refresh of its 16-bit compute copy
 * (SGDOptimizer + the mixed-precision weight conversion of the reference's dense layers):
 * w -= lr * grad_scale * g; w16 = (16-bit)w.  n % 4 == 0, 16-byte aligned buffers. */
int hctr_sgd_shadow(size_t n, float lr, float grad_scale, float* w, const float* g, void* w16,
                    int dtype, hctr_stream_t stream);

/* BinaryCrossEntropyLoss (R/HugeCTR/src/loss.cu:231-262): *loss = mean_i bce(logit_i, label_i);
 * dlogit_i = (sigmoid(logit_i) - label_i) * grad_scale (grad_scale = scaler / batch / total_gpu_count
 * in the reference); dlogit may be NULL (evaluation).  dtype of logit/dlogit: hctr_emb_dtype_t.
 * Deterministic (fixed-order block sums instead of the reference's atomicAdd). */
size_t hctr_bce_loss_workspace_bytes(void);
int hctr_bce_loss(size_t batch, const void* logit, const float* label, float grad_scale,
                  void* dlogit, float* loss, float* workspace, int dtype, hctr_stream_t stream);

/* Logit head: the network's last fully connected layer (K -> 1) + BinaryCrossEntropyLoss + both
 * backward passes in one sweep over the activations (MLPLayer's last layer,
 * R/HugeCTR/src/layers/mlp_layer.cu, and R/HugeCTR/src/loss.cu:231-262): z = x.w + bias,
 * *loss = mean_i bce(z_i, label_i), dz = (sigmoid(z) - label) * grad_scale, dx[i][:] = dz_i * w
 * (NULL to skip), dw[k] = sum_i dz_i x[i][k], *db = sum_i dz_i (fp32, fixed-order sums).
 * x / w / bias / dx 16-bit (hctr_emb_dtype_t F16 or BF16), K % 4 == 0, K <= 2048. */
size_t hctr_logit_head_workspace_bytes(int k);
int hctr_logit_head(size_t batch, int k, const void* x, const void* w, const void* bias,
                    const float* label, float grad_scale, void* dx, float* dw, float* db,
                    float* loss, float* workspace, int dtype, hctr_stream_t stream);

/* First layer of an MLP with a handful of input features (DLRM bottom MLP: 13 -> 512; MLPLayer,
 * R/HugeCTR/src/layers/mlp_layer.cu): y = relu(x W^T + b), x fp32 [batch][K] (rounded to the 16-bit
 * type like the GEMM path), W [N][K] / bias [N] / y 16-bit, 1 <= K <= 16, N % 4 == 0, N <= 512.
 * bwd: dz = dy * (y > 0) is folded into dw[N][K] = dz^T x and db[N] = sum dz (fp32, fixed-order
 * sums); dz itself is not produced -- the first layer has no data gradient.  Runs on the matrix
 * cores (v_mfma_f32_16x16x4_f32) when K < 16, N % 8 == 0 and dy / y are 16-byte aligned, on the
 * vector ALU otherwise or when the environment has HCTR_SKINNY_BWD=valu; both forms are
 * deterministic, they differ from each other in summation order only. */
int hctr_skinny_fc_fwd(size_t batch, int k, int n, const float* x, const void* w, const void* bias,
                       void* y, int dtype, hctr_stream_t stream);
size_t hctr_skinny_fc_bwd_workspace_bytes(int n);
int hctr_skinny_fc_bwd(size_t batch, int k, int n, const float* x, const void* dy, const void* y,
                       float* dw, float* db, float* workspace, int dtype, hctr_stream_t stream);

/* The GEMMs of MultiCrossLayer v2 with the elementwise work the reference fuses into its GEMM
 * epilogues (MultiCrossForwardFunctorv2 / MultiCrossBackwardFunctorv2,
 * R/HugeCTR/src/layers/multi_cross_layer.cu:582-700, 732-812; fused_mul_fma3 :391-424), as this
 * library's own matrix-core kernel (hugectr_amd/csrc/cross_gemm.hip):
 *   c[m][n] = epilogue( sum_k a[m][k] * bt[n][k] ),  a [m][lda], bt [n][ldb] (both K-contiguous),
 *   c [m][ldc]; 16-bit operands of `dtype` (hctr_emb_dtype_t F16 or BF16), fp32 accumulation.
 * epilogue 0: c = (T)acc;
 *          1: h_out = (T)(acc + bias[n]); c = (T)(xl + x0 * h_out)   -- the forward's second GEMM:
 *             bias + X_0 .* H + X_l in one pass, H kept for the backward (bias 16-bit [n]; x0, xl,
 *             h_out [m][ldc]);
 *          2: c = (T)(acc + xl)                                       -- the backward's residual
 *             (dY_{l-1} = S1 U^T + dY_l).
 * n % 128 == 0, k % 64 == 0, leading dimensions multiples of 8 elements, 16-byte aligned buffers;
 * any m. */
int hctr_gemm_nt16(size_t m, int n, int k, const void* a, int lda, const void* bt, int ldb, void* c,
                   int ldc, int epilogue, const void* bias, const void* x0, const void* xl,
                   void* h_out, int dtype, hctr_stream_t stream);
/* fp32 master weights [batch][rows][cols] -> their 16-bit copy dst [batch][rows][cols] and its
 * transpose dst_t [batch][cols][rows] in one pass (either may be NULL): the operands of
 * hctr_gemm_nt16 for MultiCrossLayer v2 (the reference converts its master weights per step too,
 * R/HugeCTR/src/layers/multi_cross_layer.cu:582-600; the transpose exists only because both GEMM
 * operands are read K-contiguous here). */
int hctr_convert_transpose16(size_t batch, int rows, int cols, const float* src, void* dst,
                             void* dst_t, int dtype, hctr_stream_t stream);
/* One layer's elementwise step of MultiCrossBackwardFunctorv2 in the activations' 16-bit type
 * (fused_mul_fma3, R/HugeCTR/src/layers/multi_cross_layer.cu:391-424 / 127-165, + the bias gradient
 * the reference takes in the dV GEMM's epilogue, :770-776): s0 = dy .* x0, acc = (first ? 0 : acc)
 * + dy .* h, each rounded once to the 16-bit type; db[c] = sum_b s0[b][c] in fp32 (two-stage, fixed
 * order).  All arrays [batch][width], width % 8 == 0; workspace: hctr_cross_v2_bwd_step_workspace_bytes
 * (batch, width).  first != 0: acc is written without being read (the last layer, visited first). */
size_t hctr_cross_v2_bwd_step_workspace_bytes(size_t batch, int width);
int hctr_cross_v2_bwd_step(size_t batch, int width, const void* dy, const void* x0, const void* h,
                           void* acc, void* s0, float* db, float* workspace, int first, int dtype,
                           hctr_stream_t stream);

/* ---- embedding cache in HBM + host<->HBM tiered table (BASELINE config 4) ----------------------
 * gpu_cache::gpu_cache<key, ref_counter, empty_key, SET_ASSOCIATIVITY 2, SLAB_SIZE 32>
 * (R/gpu_cache/include/nv_gpu_cache.hpp:46-124, R/gpu_cache/src/nv_gpu_cache.cu): a key lives in set
 * MurmurHash3_32(key) % capacity_in_set, 64 slots per set, least-recently-used replacement driven
 * by a global counter that every Query advances.  All pointers are device pointers; the calls of
 * one cache must be stream-ordered.  Results are deterministic: inside one call, keys that share a
 * set are applied in position order. */
typedef struct hctr_cache hctr_cache;
int hctr_cache_create(size_t capacity_in_set, int vec_size, int key_type, hctr_cache** out);
int hctr_cache_destroy(hctr_cache* c);
size_t hctr_cache_capacity_in_set(const hctr_cache* c);
/* Query (:53-56): hits copy their vector to values[i] and become most recent; misses leave
 * values[i] untouched and are listed in ascending position: missing_index / missing_keys
 * [*d_missing_len].  values / missing_index / missing_keys may be NULL. */
int hctr_cache_query(hctr_cache* c, const void* keys, size_t len, float* values,
                     uint64_t* missing_index, void* missing_keys, size_t* d_missing_len,
                     hctr_stream_t stream);
/* Replace (:58-60): a cached key is refreshed (its vector stays); a new key takes an empty slot
 * of its set, else evicts the set's least recently used key. */
int hctr_cache_replace(hctr_cache* c, const void* keys, size_t len, const float* values,
                       hctr_stream_t stream);
/* Update (:62-64): overwrite the vectors of the keys that are cached, ignore the others */
int hctr_cache_update(hctr_cache* c, const void* keys, size_t len, const float* values,
                      hctr_stream_t stream);
/* Dump (:66-68): the cached keys of sets [start_set_index, end_set_index), (set, slot) order */
int hctr_cache_dump(hctr_cache* c, void* keys, size_t* d_dump_counter, size_t start_set_index,
                    size_t end_set_index, hctr_stream_t stream);

/* The role of gpu_cache::UvmTable (R/gpu_cache/include/uvm_table.hpp:133-174) for the training
 * path: the full table [host_rows][vec] fp32 lives in pinned host memory (key = row), the cache
 * above holds the hot rows.  lookup = Query + the GPU reading the missing rows straight out of
 * host memory into `out` + Replace -- three launches, no host synchronisation (rows outside
 * [0, host_rows) read as zeros).  d_missing_len (device, may be NULL) receives the miss count.
 * scatter: update of UNIQUE rows, new = (add ? old : 0) + alpha * values[i] (alpha = -lr and the
 * per-row gradient sums of hctr_ebc_local_reduce make it the SGD step of a tiered embedding).  The
 * cache is WRITE-BACK: a cached row is updated in HBM only and goes home to the host table when
 * its slot is taken over by another row or at hctr_tiered_flush; a row that is not cached is
 * updated in the host table.  Every lookup reads current values either way; the HOST pointer of
 * hctr_tiered_host_rows shows them after hctr_tiered_flush (host sync). */
typedef struct hctr_tiered hctr_tiered;
int hctr_tiered_create(size_t host_rows, int vec_size, size_t cache_capacity_in_set,
                       hctr_tiered** out);
int hctr_tiered_destroy(hctr_tiered* t);
float* hctr_tiered_host_rows(hctr_tiered* t); /* HOST pointer: initialise / checkpoint the table */
hctr_cache* hctr_tiered_cache(hctr_tiered* t);
int hctr_tiered_lookup(hctr_tiered* t, const int64_t* keys, size_t len, float* out,
                       size_t* d_missing_len, hctr_stream_t stream);
int hctr_tiered_scatter(hctr_tiered* t, const int64_t* unique_keys, size_t len, const float* values,
                        int add, float alpha, hctr_stream_t stream);
int hctr_tiered_flush(hctr_tiered* t, hctr_stream_t stream);

/* gpu_cache::UvmTable<key_type, index_type, vec_type> as a whole
 * (R/gpu_cache/include/uvm_table.hpp:127-174; src/uvm_table.cu:283-510): a key -> vector map whose
 * vectors live in host memory with the hot ones in HBM -- BASELINE configs[3]'s 10 B-row key spaces
 * need an index in front of the store, not a store as large as the key space.  The constructor's
 * arguments are the reference's (device_table_capacity / host_table_capacity in vectors,
 * max_batch_size, vec_size, default_value) + the key type.  One device index (the path's own hash
 * map, 16 bytes of HBM per slot, capacity / 0.75 slots) maps a key to a row of the pinned host
 * store, handed out on first touch in order of first occurrence; the set-associative cache of
 * hctr_cache_* keeps the hot rows (device_table_capacity rounded up to whole sets of 64).
 *   add   (:137, uvm_table.cu:318-418): HOST keys / vectors, synchronous; a key met again gets the
 *         new vector (its last one when a call lists it twice); HCTR_ERR_OVERFLOW when the host
 *         store is full (the reference spills into an unordered_map on the host).
 *   query (:136, :421-497): DEVICE keys -> DEVICE vectors; a key that was never added reads
 *         default_value in every element.  Asynchronous on `stream`.
 *   clear (:138): forgets every key (index and cache).
 * For the training path (not in the reference's class, which serves inference):
 *   lookup: as query, but a key met for the first time takes the next row of the host store -- whose
 *         content is whatever hctr_tiered_host_rows(hctr_uvm_tier(u)) holds there (the caller's
 *         initialisation) -- and row_index (may be NULL; len <= max_batch_size) receives every
 *         key's row for the update; hctr_uvm_check_overflow reports a full store (host sync).
 *   scatter_rows: hctr_tiered_scatter on rows handed back by lookup (unique within the call). */
typedef struct hctr_uvm hctr_uvm;
int hctr_uvm_create(size_t device_table_capacity, size_t host_table_capacity, size_t max_batch_size,
                    int vec_size, float default_value, int key_type, hctr_uvm** out);
int hctr_uvm_destroy(hctr_uvm* u);
hctr_tiered* hctr_uvm_tier(hctr_uvm* u);
int hctr_uvm_add(hctr_uvm* u, const void* h_keys, const float* h_vectors, size_t len);
int hctr_uvm_query(hctr_uvm* u, const void* d_keys, size_t len, float* d_vectors,
                   hctr_stream_t stream);
int hctr_uvm_clear(hctr_uvm* u, hctr_stream_t stream);
int hctr_uvm_lookup(hctr_uvm* u, const void* d_keys, size_t len, float* d_vectors,
                    uint64_t* d_row_index, size_t* d_missing_len, hctr_stream_t stream);
int hctr_uvm_scatter_rows(hctr_uvm* u, const uint64_t* d_unique_rows, size_t len,
                          const float* values, int add, float alpha, hctr_stream_t stream);
int hctr_uvm_check_overflow(hctr_uvm* u, hctr_stream_t stream);
int hctr_uvm_size(hctr_uvm* u, hctr_stream_t stream, size_t* out); /* keys held; host sync */

#ifdef __cplusplus
}
#endif
#endif
