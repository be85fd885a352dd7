/*
 * hctr_oracle.h -- CPU restatement of the NVIDIA-Merlin/HugeCTR sparse-embedding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the reported CPU baseline.  The product path is hugectr_amd/csrc (HIP, gfx950).
 *
 * PARITY STATUS: the reference (R = /root/reference) stores no golden vectors for this path (all
 * its tests draw random data at run time and compare GPU vs CPU in-process) and its product code
 * cannot be compiled here (every TU pulls in cublas/curand/nvml/mpi/nccl through
 * HugeCTR/include/common.hpp:18-38).  Pinned against code compiled FROM THE REFERENCE CHECKOUT
 * (oracle/Makefile `ref` -> oracle/_ref/, built where the checkout is present):
 *   - hash / index stage: the reference's MurmurHash3 functor and gpu_cache hashes
 *     (tests/test_ref_hash_cpu.py), plus public MurmurHash3_x86_32 known-answer vectors;
 *   - forward sum / mean, backward, and EVERY optimizer x Local / Global / LazyGlobal, fp32 and
 *     fp16 outputs / state: the reference's own CPU test oracle SparseEmbeddingHashCpu
 *     (R/test/utest/embedding/sparse_embedding_hash_cpu.hpp), compiled with a declarations-only
 *     stand-in for common.hpp (oracle/ref_shims/) and driven over several training steps from
 *     Norm dataset + sparse model files (tests/test_ref_embedding_cpu.py): sum forward / wgrad
 *     bit-equal, tables within 5e-7.
 *   - the same functions against the reference's DEVICE code (the CUDA kernels of
 *     forward_per_gpu_functor.cu / backward_functor.cu / forward_scale_functor.cu and
 *     EmbeddingOptimizer::update with its kernels, cut out of the checkout and executed by the
 *     host interpreter of tests/emu): forward / backward bit-equal incl. the fp16 align2 rules,
 *     ten optimizer variants within 1e-6 (tests/test_ref_gpu_kernels_cpu.py).
 * The other oracles are pinned by their own reference builds (oracle/Makefile `ref`): Interaction /
 * Cross (the CPU references inline in the reference's gtest files, tests/test_ref_layers_cpu.py),
 * embedding_collection (tests/test_ref_ebc_cpu.py), the dynamic table (tests/test_ref_det_cpu.py),
 * the embedding cache (the reference's CUDA kernels executed by the host interpreter,
 * tests/test_ref_cache_cpu.py).  The reorder maps: the reference's reorder kernels, executed
 * (tests/test_ref_gpu_kernels_cpu.py).
 *
 * All functions are plain C, single-threaded unless `threads > 1` is passed where offered.
 */
#ifndef HCTR_ORACLE_H
#define HCTR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCO_INVALID_INDEX ((uint64_t)-1) /* std::numeric_limits<size_t>::max(), nv_hashtable.hpp:33 */

/* ---- hash / index stage ------------------------------------------------------------------- */
uint32_t hco_murmur3_32(const void* data, int len, uint32_t seed);
uint32_t hco_hash_key(int64_t key, int key_bytes);

typedef struct hco_hashtable hco_hashtable;
hco_hashtable* hco_ht_create(uint64_t capacity, int key_bytes);
void hco_ht_destroy(hco_hashtable* ht);
void hco_ht_clear(hco_hashtable* ht);
uint64_t hco_ht_table_size(const hco_hashtable* ht); /* physical slots = (size_t)(capacity/0.75f) */
uint64_t hco_ht_size(const hco_hashtable* ht);       /* number of occupied slots */
uint64_t hco_ht_value_head(const hco_hashtable* ht); /* counter */
void hco_ht_set_value_head(hco_hashtable* ht, uint64_t v);
int hco_ht_insert(hco_hashtable* ht, const int64_t* keys, const uint64_t* vals, uint64_t n);
int hco_ht_get_insert(hco_hashtable* ht, const int64_t* keys, uint64_t* vals, uint64_t n);
void hco_ht_get_mark(const hco_hashtable* ht, const int64_t* keys, uint64_t* vals, uint64_t n);
uint64_t hco_ht_dump(const hco_hashtable* ht, int64_t* keys, uint64_t* vals);

/* ---- key routing (legacy) ------------------------------------------------------------------ */
uint64_t hco_localized_filter(const int64_t* row_offset, const int64_t* keys, int64_t batch,
                              int64_t slot_num, int64_t gid, int64_t gnum, int64_t* out_row_offset,
                              int64_t* out_keys);
uint64_t hco_distributed_filter(const int64_t* row_offset, const int64_t* keys, int64_t batch,
                                int64_t slot_num, int64_t gid, int64_t gnum,
                                int64_t* out_row_offset, int64_t* out_keys);
int64_t hco_slots_on_gpu(int64_t slot_num, int64_t gid, int64_t gnum);

/* ---- forward / backward -------------------------------------------------------------------- */
void hco_forward(int64_t buckets, int64_t D, int combiner, const int64_t* row_offset,
                 const uint64_t* value_index, const float* table, float* out, int threads);
void hco_backward(int64_t buckets, int64_t D, int combiner, const int64_t* row_offset,
                  const float* top_grad, float* wgrad);
void hco_forward_reorder(int64_t batch_per_gpu, int64_t slot_num, int64_t D, int64_t gnum,
                         const float* in, float* out);
void hco_backward_reorder(int64_t batch_per_gpu, int64_t slot_num, int64_t D, int64_t gnum,
                          const float* in, float* out);

/* ---- sparse optimizer ---------------------------------------------------------------------- */
enum { HCO_OPT_ADAM = 0, HCO_OPT_ADAGRAD = 1, HCO_OPT_MOMENTUM = 2, HCO_OPT_NESTEROV = 3,
       HCO_OPT_SGD = 4 };
enum { HCO_UPDATE_LOCAL = 0, HCO_UPDATE_GLOBAL = 1, HCO_UPDATE_LAZY = 2 };

typedef struct {
  int optimizer;
  int update_type;
  float lr;
  float beta1, beta2, epsilon; /* adam; adagrad uses epsilon */
  float momentum_factor;       /* momentum: factor, nesterov: mu */
  float scaler;
  uint64_t times; /* adam step counter AFTER increment (t starts at 1) */
  int state_half; /* OptimizerTensor<__half> (optimizer.hpp:284-296): state stored in fp16 (q6) */
} hco_opt_params;

/* float -> IEEE binary16 (round to nearest even) -> float, what
 * TypeConvertFunc<__half, float> / <float, __half> do around every state store */
float hco_round_half(float x);

/* returns number of unique rows; table/state0/state1/prev_time are [vocab, D] */
int64_t hco_update_params(int64_t buckets, int64_t D, int64_t vocab, const int64_t* row_offset,
                          const uint64_t* value_index, const float* wgrad,
                          const hco_opt_params* opt, float* table, float* state0, float* state1,
                          uint64_t* prev_time, int fast_sort, int threads);

/* ---- dense ops ----------------------------------------------------------------------------- */
void hco_interaction_fwd(int64_t B, int64_t n_emb, int64_t W, const float* mlp, const float* emb,
                         float* out);
void hco_interaction_bwd(int64_t B, int64_t n_emb, int64_t W, const float* mlp, const float* emb,
                         const float* top_grad, float* mlp_grad, float* emb_grad);
void hco_cross_v1_fwd(int64_t B, int64_t w, int layers, const float* x0, const float* kernels,
                      const float* biases, float* outputs, float* hiddens);
void hco_cross_v1_bwd(int64_t B, int64_t w, int layers, const float* x0, const float* kernels,
                      const float* outputs, const float* hiddens, const float* out_grad,
                      float* in_grad, float* kernel_grads, float* bias_grads);
void hco_cross_v2_fwd(int64_t B, int64_t w, int64_t p, int layers, const float* x0,
                      const float* U, const float* V, const float* biases, float* outputs,
                      float* hiddens, float* XUs);
void hco_cross_v2_bwd(int64_t B, int64_t w, int64_t p, int layers, const float* x0,
                      const float* U, const float* V, const float* outputs, const float* hiddens,
                      const float* XUs, const float* out_grad, float* in_grad, float* dU,
                      float* dV, float* db);

/* ---- embedding_collection (EBC) reference -------------------------------------------------- */
void hco_ebc_forward(int64_t batch, int64_t num_lookup, const int32_t* table_ids,
                     const int32_t* ev_sizes, const int32_t* combiners, const int64_t* keys,
                     const int64_t* bucket_range, const int64_t* table_row_start,
                     const int64_t* table_ev_start, const float* tables, int64_t num_gpus,
                     int batch_major, float* out);
void hco_ebc_backward_update(int64_t batch, int64_t num_lookup, const int32_t* table_ids,
                             int64_t ev, const int32_t* combiners, const int64_t* keys,
                             const int64_t* bucket_range, const int64_t* table_row_start,
                             int64_t total_rows, int64_t num_gpus, int batch_major,
                             const float* top_grad, int optimizer /* 0 SGD, 1 AdaGrad, 2 Ftrl */,
                             float lr, float scaler, float epsilon, float* tables, float* accum,
                             float ftrl_lambda1, float ftrl_lambda2, float ftrl_beta,
                             float* ftrl_z);
void hco_keys_to_indices(int64_t n, const int64_t* keys, int64_t table_start, int64_t num_shards,
                         int64_t* idx);

/* ---- synthetic data ------------------------------------------------------------------------ */
void hco_powerlaw_keys(uint32_t seed, int64_t n, int64_t vocab, float alpha, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif
