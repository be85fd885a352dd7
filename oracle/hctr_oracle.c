/*
 * hctr_oracle.c -- CPU restatement of the NVIDIA-Merlin/HugeCTR sparse-embedding hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see hctr_oracle.h).  "parity unpinned" except for the hash KATs.
 * R = /root/reference.  Every function cites the reference lines it restates.  Where the
 * reference's CPU test helper and its GPU product kernel differ by rounding, the GPU product
 * kernel's arithmetic is followed (it is what a user of the reference observes) and the
 * difference is noted.
 */
#include "hctr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* =========================================================================================== */
/* MurmurHash3_x86_32 -- R/HugeCTR/include/hashtable/cudf/hash_functions.cuh:31-113 (seed 0,    */
/* len = sizeof(Key)); public-domain algorithm by Austin Appleby.                               */
/* =========================================================================================== */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

uint32_t hco_murmur3_32(const void* data_, int len, uint32_t seed) {
  const uint8_t* data = (const uint8_t*)data_;
  const int nblocks = len / 4;
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k1;
    memcpy(&k1, data + 4 * i, 4);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = rotl32(k1, 15);
      k1 *= c2;
      h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  return fmix32(h1);
}

/* key_bytes = 4: Key = unsigned int; key_bytes = 8: Key = long long (nv_hashtable.cu:305-306) */
uint32_t hco_hash_key(int64_t key, int key_bytes) {
  if (key_bytes == 4) {
    uint32_t k = (uint32_t)key;
    return hco_murmur3_32(&k, 4, 0);
  }
  return hco_murmur3_32(&key, 8, 0);
}

/* =========================================================================================== */
/* Hash table: sequential restatement of HashTable<Key,size_t>                                  */
/*   R/HugeCTR/src/hashtable/nv_hashtable.cu:169-186 (physical size = capacity / 0.75f, float) */
/*   R/HugeCTR/include/hashtable/cudf/concurrent_unordered_map.cuh:562-655 (find / get_insert) */
/* The reference assigns row indices by atomicAdd in thread-race order; the         */
/* restatement inserts in array order, i.e. first occurrence in CSR order gets the next index.  */
/* =========================================================================================== */
struct hco_hashtable {
  int64_t* keys;
  uint64_t* vals;
  uint64_t size;     /* physical slots */
  uint64_t capacity; /* max_vocabulary_size_per_gpu */
  uint64_t counter;  /* d_counter_ */
  int key_bytes;
  int64_t empty_key;
};

static int64_t empty_key_for(int key_bytes) {
  return key_bytes == 4 ? (int64_t)0xFFFFFFFFu : INT64_MAX; /* numeric_limits<Key>::max() */
}

hco_hashtable* hco_ht_create(uint64_t capacity, int key_bytes) {
  hco_hashtable* ht = (hco_hashtable*)calloc(1, sizeof(*ht));
  ht->capacity = capacity;
  ht->key_bytes = key_bytes;
  ht->empty_key = empty_key_for(key_bytes);
  /* static_cast<size_t>(capacity / LOAD_FACTOR) with `const float LOAD_FACTOR = 0.75f`:
   * size_t / float -> float division (nv_hashtable.cu:178, nv_hashtable.hpp:179) */
  ht->size = (uint64_t)((float)capacity / 0.75f);
  if (ht->size == 0) ht->size = 1;
  ht->keys = (int64_t*)malloc(ht->size * sizeof(int64_t));
  ht->vals = (uint64_t*)malloc(ht->size * sizeof(uint64_t));
  hco_ht_clear(ht);
  return ht;
}

void hco_ht_destroy(hco_hashtable* ht) {
  if (!ht) return;
  free(ht->keys);
  free(ht->vals);
  free(ht);
}

void hco_ht_clear(hco_hashtable* ht) {
  for (uint64_t i = 0; i < ht->size; i++) {
    ht->keys[i] = ht->empty_key;
    ht->vals[i] = HCO_INVALID_INDEX;
  }
  ht->counter = 0;
}

uint64_t hco_ht_table_size(const hco_hashtable* ht) { return ht->size; }
uint64_t hco_ht_value_head(const hco_hashtable* ht) { return ht->counter; }
void hco_ht_set_value_head(hco_hashtable* ht, uint64_t v) { ht->counter = v; }

uint64_t hco_ht_size(const hco_hashtable* ht) { /* size_kernel, nv_hashtable.cu:116-127 */
  uint64_t n = 0;
  for (uint64_t i = 0; i < ht->size; i++) n += (ht->keys[i] != ht->empty_key);
  return n;
}

/* returns slot index or size (== end()) */
static uint64_t ht_find_slot(const hco_hashtable* ht, int64_t key) {
  uint64_t idx = (uint64_t)hco_hash_key(key, ht->key_bytes) % ht->size;
  uint64_t counter = 0;
  for (;;) { /* concurrent_unordered_map.cuh:562-585 */
    int64_t cur = ht->keys[idx];
    if (cur == key) return idx;
    if (cur == ht->empty_key || counter > ht->size) return ht->size;
    idx = (idx + 1) % ht->size;
    ++counter;
  }
}

/* insert(key,val) pairs: insert_kernel, nv_hashtable.cu:35-47 (used by load_parameters) */
int hco_ht_insert(hco_hashtable* ht, const int64_t* keys, const uint64_t* vals, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    uint64_t idx = (uint64_t)hco_hash_key(keys[i], ht->key_bytes) % ht->size;
    uint64_t tries = 0;
    for (;;) {
      if (tries++ >= ht->size) return -1;
      if (ht->keys[idx] == ht->empty_key || ht->keys[idx] == keys[i]) {
        ht->keys[idx] = keys[i];
        ht->vals[idx] = vals[i];
        break;
      }
      idx = (idx + 1) % ht->size;
    }
  }
  return 0;
}

/* get_insert_kernel nv_hashtable.cu:61-72 + concurrent_unordered_map.cuh:587-655 */
int hco_ht_get_insert(hco_hashtable* ht, const int64_t* keys, uint64_t* vals, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    const int64_t key = keys[i];
    uint64_t idx = (uint64_t)hco_hash_key(key, ht->key_bytes) % ht->size;
    uint64_t counter = 0;
    for (;;) {
      if (counter++ >= ht->size) return -1; /* Situation 5: table full -> end() */
      if (ht->keys[idx] == ht->empty_key) { /* Situation 1 */
        ht->keys[idx] = key;
        ht->vals[idx] = ht->counter++;
        break;
      } else if (ht->keys[idx] == key) { /* Situation 3 */
        break;
      }
      idx = (idx + 1) % ht->size; /* Situation 4 */
    }
    vals[i] = ht->vals[idx];
  }
  return 0;
}

/* get_mark_kernel nv_hashtable.cu:74-83: miss -> SIZE_MAX */
void hco_ht_get_mark(const hco_hashtable* ht, const int64_t* keys, uint64_t* vals, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    uint64_t s = ht_find_slot(ht, keys[i]);
    vals[i] = (s == ht->size) ? HCO_INVALID_INDEX : ht->vals[s];
  }
}

/* dump_kernel nv_hashtable.cu:129-163: occupied slots in physical order */
uint64_t hco_ht_dump(const hco_hashtable* ht, int64_t* keys, uint64_t* vals) {
  uint64_t n = 0;
  for (uint64_t i = 0; i < ht->size; i++) {
    if (ht->keys[i] != ht->empty_key) {
      keys[n] = ht->keys[i];
      vals[n] = ht->vals[i];
      n++;
    }
  }
  return n;
}

/* =========================================================================================== */
/* Key routing                                                                                  */
/* =========================================================================================== */
/* R/HugeCTR/include/embeddings/localized_slot_sparse_embedding_hash.hpp:100-122,176-183 */
int64_t hco_slots_on_gpu(int64_t slot_num, int64_t gid, int64_t gnum) {
  return slot_num / gnum + ((gid < slot_num % gnum) ? 1 : 0);
}

/* select_value_and_rowoffset_by_slot_id_kernel + DeviceSelect::Flagged + InclusiveSum,
 * R/HugeCTR/src/embeddings/localized_slot_sparse_embedding_hash.cu:35-54,81-144 */
uint64_t hco_localized_filter(const int64_t* row_offset, const int64_t* keys, int64_t batch,
                              int64_t slot_num, int64_t gid, int64_t gnum, int64_t* out_row_offset,
                              int64_t* out_keys) {
  const int64_t spg = hco_slots_on_gpu(slot_num, gid, gnum);
  uint64_t nnz = 0;
  out_row_offset[0] = 0;
  for (int64_t b = 0; b < batch; b++) {
    for (int64_t s = 0; s < slot_num; s++) {
      if (s % gnum != gid) continue;
      const int64_t t = b * slot_num + s;
      const int64_t res = s / gnum;
      for (int64_t i = row_offset[t]; i < row_offset[t + 1]; i++) out_keys[nnz++] = keys[i];
      out_row_offset[1 + b * spg + res] = (int64_t)nnz;
    }
  }
  return nnz;
}

/* R/HugeCTR/src/embeddings/distributed_slot_sparse_embedding_hash.cu:35-52,94-152:
 * keep keys with key % gnum == gid, row offsets keep all batch*slot_num buckets */
uint64_t hco_distributed_filter(const int64_t* row_offset, const int64_t* keys, int64_t batch,
                                int64_t slot_num, int64_t gid, int64_t gnum,
                                int64_t* out_row_offset, int64_t* out_keys) {
  uint64_t nnz = 0;
  out_row_offset[0] = 0;
  for (int64_t t = 0; t < batch * slot_num; t++) {
    for (int64_t i = row_offset[t]; i < row_offset[t + 1]; i++) {
      if (keys[i] % gnum == gid) out_keys[nnz++] = keys[i];
    }
    out_row_offset[t + 1] = (int64_t)nnz;
  }
  return nnz;
}

/* =========================================================================================== */
/* Forward / backward                                                                           */
/* =========================================================================================== */
/* forward_sum_kernel / forward_mean_kernel, R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu
 * :28-57,103-135 (GPU product arithmetic: SIZE_MAX index contributes 0.0f; mean multiplies the
 * fp32 sum by 1.0f/n when n > 1).  The reference's CPU test helper divides instead
 * (R/test/utest/embedding/sparse_embedding_hash_cpu.hpp:416-439); <= 1 ulp apart.           */
void hco_forward(int64_t buckets, int64_t D, int combiner, const int64_t* row_offset,
                 const uint64_t* value_index, const float* table, float* out, int threads) {
  (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
  for (int64_t u = 0; u < buckets; u++) {
    const int64_t off = row_offset[u];
    const int64_t n = row_offset[u + 1] - off;
    float scaler = 1.0f;
    if (combiner == 1 && n > 1) scaler = 1.0f / (float)n;
    for (int64_t v = 0; v < D; v++) {
      float sum = 0.0f;
      for (int64_t j = 0; j < n; j++) {
        const uint64_t idx = value_index[off + j];
        sum += (idx != HCO_INVALID_INDEX) ? table[idx * (uint64_t)D + (uint64_t)v] : 0.0f;
      }
      out[u * D + v] = (combiner == 1) ? sum * scaler : sum;
    }
  }
}

/* backward_sum_kernel / backward_mean_kernel, R/HugeCTR/src/embeddings/backward_functor.cu:26-104
 * (== cpu_backward_sum/mean, sparse_embedding_hash_cpu.hpp:465-503) */
void hco_backward(int64_t buckets, int64_t D, int combiner, const int64_t* row_offset,
                  const float* top_grad, float* wgrad) {
  for (int64_t u = 0; u < buckets; u++) {
    float scaler = 1.0f;
    if (combiner == 1) {
      const int64_t n = row_offset[u + 1] - row_offset[u];
      if (n > 1) scaler = 1.0f / (float)n;
    }
    for (int64_t v = 0; v < D; v++) {
      wgrad[u * D + v] = (combiner == 1) ? top_grad[u * D + v] * scaler : top_grad[u * D + v];
    }
  }
}

/* forward_reorder_kernel, R/HugeCTR/src/embeddings/forward_reorder_functor.cu:26-58:
 * in  = [gpu g][local sample b][slot j of gpu g][D]  (the all-to-all receive buffer)
 * out = [local sample b][global slot = g + gnum*j][D] */
void hco_forward_reorder(int64_t bpg, int64_t slot_num, int64_t D, int64_t gnum, const float* in,
                         float* out) {
  for (int64_t b = 0; b < bpg; b++) {
    for (int64_t s = 0; s < slot_num; s++) {
      const int64_t g = s % gnum;
      int64_t offset_pre = 0;
      for (int64_t id = 0; id < g; id++) offset_pre += bpg * hco_slots_on_gpu(slot_num, id, gnum);
      const int64_t spg = hco_slots_on_gpu(slot_num, g, gnum);
      const int64_t src = (b * spg + offset_pre + s / gnum) * D;
      const int64_t dst = (b * slot_num + s) * D;
      memcpy(out + dst, in + src, (size_t)D * sizeof(float));
    }
  }
}

/* backward_reorder_kernel, R/HugeCTR/src/embeddings/backward_reorder_functor.cu (inverse map) */
void hco_backward_reorder(int64_t bpg, int64_t slot_num, int64_t D, int64_t gnum, const float* in,
                          float* out) {
  for (int64_t b = 0; b < bpg; b++) {
    for (int64_t s = 0; s < slot_num; s++) {
      const int64_t g = s % gnum;
      int64_t offset_pre = 0;
      for (int64_t id = 0; id < g; id++) offset_pre += bpg * hco_slots_on_gpu(slot_num, id, gnum);
      const int64_t spg = hco_slots_on_gpu(slot_num, g, gnum);
      const int64_t dst = (b * spg + offset_pre + s / gnum) * D;
      const int64_t src = (b * slot_num + s) * D;
      memcpy(out + dst, in + src, (size_t)D * sizeof(float));
    }
  }
}

/* =========================================================================================== */
/* Sparse optimizer: EmbeddingOptimizer::update, R/HugeCTR/src/optimizers/sparse_optimizer.cu  */
/* :622-864, == SparseEmbeddingHashCpu::update_params, sparse_embedding_hash_cpu.hpp:920-1015   */
/* =========================================================================================== */
typedef struct {
  uint64_t idx;
  int64_t sample;
} hco_pair;

/* stable merge sort by idx == cub::DeviceRadixSort::SortPairs (stable) == the reference CPU
 * odd-even transposition sort (sparse_embedding_hash_cpu.hpp:541-561; strict compares => stable) */
static void merge_sort_pairs(hco_pair* a, hco_pair* tmp, int64_t n) {
  for (int64_t width = 1; width < n; width *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * width) {
      int64_t mid = lo + width < n ? lo + width : n;
      int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (a[j].idx < a[i].idx) ? a[j++] : a[i++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, (size_t)n * sizeof(hco_pair));
  }
}

/* the reference's literal O(nnz^2) odd-even sort (cpu_csr_sort), kept for small-size timing and
 * to show it equals the stable sort */
static void odd_even_sort_pairs(hco_pair* a, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    if (i % 2 == 0) {
      for (int64_t j = 1; j < n; j += 2)
        if (a[j].idx < a[j - 1].idx) {
          hco_pair t = a[j];
          a[j] = a[j - 1];
          a[j - 1] = t;
        }
    } else {
      for (int64_t j = 1; j < n - 1; j += 2)
        if (a[j].idx > a[j + 1].idx) {
          hco_pair t = a[j];
          a[j] = a[j + 1];
          a[j + 1] = t;
        }
    }
  }
}

float hco_round_half(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint32_t sign = u & 0x80000000u;
  uint32_t a = u & 0x7FFFFFFFu;
  float out;
  if (a >= 0x7F800000u) return x; /* inf / nan stay */
  if (a >= 0x477FF000u) {         /* >= 65520: rounds to inf in binary16 */
    uint32_t inf = sign | 0x7F800000u;
    memcpy(&out, &inf, 4);
    return out;
  }
  if (a < 0x38800000u) { /* below the smallest normal half (2^-14): subnormal grid of 2^-24 */
    /* adding 0.5f puts the fp32 ulp at 2^-24: the add itself rounds to nearest even */
    float ax;
    memcpy(&ax, &a, 4);
    volatile float t = ax + 0.5f;
    float r = t - 0.5f;
    uint32_t ru;
    memcpy(&ru, &r, 4);
    ru |= sign;
    memcpy(&out, &ru, 4);
    return out;
  }
  /* normal range: keep 10 mantissa bits, round to nearest even on the 13 dropped ones */
  const uint32_t lsb = (a >> 13) & 1u;
  a += 0xFFFu + lsb;
  a &= ~0x1FFFu;
  a |= sign;
  memcpy(&out, &a, 4);
  return out;
}

int64_t hco_update_params(int64_t buckets, int64_t D, int64_t vocab, const int64_t* row_offset,
                          const uint64_t* value_index, const float* wgrad,
                          const hco_opt_params* opt, float* table, float* state0, float* state1,
                          uint64_t* prev_time, int fast_sort, int threads) {
  (void)threads;
  const int64_t nnz = row_offset[buckets];
  if (buckets == 0) return 0;
  hco_pair* pairs = (hco_pair*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(hco_pair));
  hco_pair* tmp = (hco_pair*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(hco_pair));
  /* step1: sample_id_expand_kernel :189-200 / cpu_csr_extend :506-515 */
  for (int64_t i = 0; i < buckets; i++)
    for (int64_t j = row_offset[i]; j < row_offset[i + 1]; j++) {
      pairs[j].idx = value_index[j];
      pairs[j].sample = i;
    }
  /* step3: sort by value_index */
  if (fast_sort) merge_sort_pairs(pairs, tmp, nnz);
  else odd_even_sort_pairs(pairs, nnz);
  /* step4: run starts (value_count_kernel_1/2 :172-219 / cpu_csr_unduplicate :563-583) */
  int64_t* run_off = (int64_t*)malloc((size_t)(nnz + 2) * sizeof(int64_t));
  int64_t nuniq = 0;
  for (int64_t i = 0; i < nnz; i++)
    if (i == 0 || pairs[i].idx != pairs[i - 1].idx) run_off[nuniq++] = i;
  run_off[nuniq] = nnz;

#define ST(x) (opt->state_half ? hco_round_half(x) : (x))
  const float lr = opt->lr, scaler = opt->scaler;
  const float b1 = opt->beta1, b2 = opt->beta2, eps = opt->epsilon, mf = opt->momentum_factor;
  /* alpha_t = lr * sqrt(1-beta2^t)/(1-beta1^t): AdamOptHyperParams::bias(), optimizer.hpp:58-60 */
  const float bias = (float)(sqrt(1.0 - pow((double)b2, (double)opt->times)) /
                             (1.0 - pow((double)b1, (double)opt->times)));
  const float alpha_t = lr * bias; /* sparse_optimizer.cu:706 `opt_params.lr * adam.bias()` */
  const float alpha_t_lazy_common = lr / (1.0f - b1);

  /* Nesterov global: whole-table sweep FIRST (sparse_optimizer.cu:735-748, :310-324) */
  if (opt->optimizer == HCO_OPT_NESTEROV && opt->update_type == HCO_UPDATE_GLOBAL) {
    for (int64_t f = 0; f < vocab * D; f++) {
      float accm = state0[f] * mf;
      state0[f] = ST(accm);
      table[f] += accm * mf;
    }
  }

#pragma omp parallel for schedule(dynamic, 64) num_threads(threads > 0 ? threads : 1)
  for (int64_t r = 0; r < nuniq; r++) {
    const int64_t off = run_off[r], num = run_off[r + 1] - off;
    const uint64_t row = pairs[off].idx;
    for (int64_t j = 0; j < D; j++) {
      /* accumulate_gradients :223-237: ascending position in the sorted list, then / scaler */
      float gi = 0.0f;
      for (int64_t k = 0; k < num; k++) gi += wgrad[pairs[off + k].sample * D + j];
      gi = gi / scaler;
      const uint64_t f = row * (uint64_t)D + (uint64_t)j;
      switch (opt->optimizer) {
        case HCO_OPT_SGD: /* opt_sgd_kernel :497-518 */
          table[f] += -lr * gi;
          break;
        case HCO_OPT_ADAGRAD: { /* opt_adagrad_kernel :410-437 */
          float accum = state0[f] + gi * gi;
          state0[f] = ST(accum);
          table[f] += -lr * gi / (sqrtf(accum) + eps);
        } break;
        case HCO_OPT_ADAM:
          if (opt->update_type == HCO_UPDATE_LOCAL) { /* opt_adam_kernel :379-408 */
            float mi = b1 * state0[f] + (1.0f - b1) * gi;
            float vi = b2 * state1[f] + (1.0f - b2) * gi * gi;
            state0[f] = ST(mi);
            state1[f] = ST(vi);
            table[f] += -alpha_t * mi / (sqrtf(vi) + eps);
          } else if (opt->update_type == HCO_UPDATE_GLOBAL) { /* opt_adam_kernel_global :241-265 */
            state0[f] = ST(state0[f] + (1.0f - b1) * gi / b1);
            state1[f] = ST(state1[f] + (1.0f - b2) * gi * gi / b2);
          } else { /* opt_adam_kernel_lazy :524-561 */
            uint64_t pt = prev_time[f];
            prev_time[f] = opt->times;
            uint64_t skipped = opt->times - pt;
            float b1ps = powf(b1, (float)skipped);
            float a = alpha_t_lazy_common * sqrtf(1.0f - powf(b2, (float)pt)) /
                      (1.0f - powf(b1, (float)pt)) * (1.0f - b1ps);
            float mi = state0[f], vi = state1[f];
            table[f] += -a * mi / (sqrtf(vi) + eps);
            mi = b1ps * mi + (1.0f - b1) * gi;
            vi = powf(b2, (float)skipped) * vi + (1.0f - b2) * gi * gi;
            state0[f] = ST(mi);
            state1[f] = ST(vi);
          }
          break;
        case HCO_OPT_MOMENTUM:
          if (opt->update_type == HCO_UPDATE_LOCAL) { /* opt_momentum_sgd_kernel :440-465 */
            float mo = mf * state0[f] - lr * gi;
            state0[f] = ST(mo);
            table[f] += mo;
          } else { /* opt_momentum_sgd_kernel_global :292-312 */
            state0[f] = ST(state0[f] - lr * gi / mf);
          }
          break;
        case HCO_OPT_NESTEROV:
          if (opt->update_type == HCO_UPDATE_LOCAL) { /* opt_nesterov_kernel :468-494 */
            float accm_old = state0[f];
            float accm_new = mf * accm_old - lr * gi;
            state0[f] = ST(accm_new);
            table[f] += -mf * accm_old + (1.0f + mf) * accm_new;
          } else { /* nesterov_local_update_kernel_global :352-375 */
            float accm = state0[f];
            accm -= lr * gi;
            state0[f] = ST(accm);
            table[f] -= (1.0f + mf) * (lr * gi);
          }
          break;
        default: break;
      }
    }
  }

  /* global sweeps over ALL max_vocabulary_size_per_gpu rows */
  if (opt->update_type == HCO_UPDATE_GLOBAL) {
    if (opt->optimizer == HCO_OPT_ADAM) { /* adam_update_kernel_global :269-288 */
      for (int64_t f = 0; f < vocab * D; f++) {
        float mi = b1 * state0[f];
        float vi = b2 * state1[f];
        state0[f] = ST(mi);
        state1[f] = ST(vi);
        table[f] += -alpha_t * mi / (sqrtf(vi) + eps);
      }
    } else if (opt->optimizer == HCO_OPT_MOMENTUM) { /* momentum_sgd_update_kernel_global */
      for (int64_t f = 0; f < vocab * D; f++) {
        float mo = state0[f];
        mo *= mf;
        table[f] += mo;
        state0[f] = ST(mo);
      }
    }
  }
#undef ST
  free(run_off);
  free(pairs);
  free(tmp);
  return nuniq;
}

/* =========================================================================================== */
/* InteractionLayer: CPU reference inlined in                                                   */
/* R/test/utest/core23_layer_test/interaction_layer_test.cpp:95-282                             */
/*   out[b] = [ mlp[b] (W) | dot(x_n, x_m) for n = 1..n_ins-1, m = 0..n-1 | 0 ]                 */
/*   x_0 = mlp[b], x_i = emb[b][i-1]                                                            */
/* =========================================================================================== */
void hco_interaction_fwd(int64_t B, int64_t n_emb, int64_t W, const float* mlp, const float* emb,
                         float* out) {
  const int64_t n_ins = n_emb + 1;
  const int64_t out_len = W + n_ins * (n_ins - 1) / 2 + 1;
  for (int64_t p = 0; p < B; p++) {
    float* o = out + p * out_len;
    int64_t cur = 0;
    for (int64_t i = 0; i < W; i++) o[cur++] = mlp[p * W + i];
    for (int64_t n = 0; n < n_ins; n++) {
      const float* xn = (n == 0) ? mlp + p * W : emb + (p * n_emb + (n - 1)) * W;
      for (int64_t m = 0; m < n; m++) {
        const float* xm = (m == 0) ? mlp + p * W : emb + (p * n_emb + (m - 1)) * W;
        float accum = 0.0f;
        for (int64_t k = 0; k < W; k++) accum += xm[k] * xn[k];
        o[cur++] = accum;
      }
    }
    o[cur] = 0.0f;
  }
}

/* bprop, interaction_layer_test.cpp:212-282: dM[m][n] = grad of pair (n>m); dX = (dM + dM^T) X;
 * mlp_grad = top_grad[0:W] + dX[0]; emb_grad[i-1] = dX[i]. */
void hco_interaction_bwd(int64_t B, int64_t n_emb, int64_t W, const float* mlp, const float* emb,
                         const float* top_grad, float* mlp_grad, float* emb_grad) {
  const int64_t n_ins = n_emb + 1;
  const int64_t out_len = W + n_ins * (n_ins - 1) / 2 + 1;
  float* mat = (float*)malloc((size_t)(n_ins * n_ins) * sizeof(float));
  for (int64_t p = 0; p < B; p++) {
    const float* g = top_grad + p * out_len;
    int64_t cur = W;
    for (int64_t n = 0; n < n_ins; n++)
      for (int64_t m = 0; m < n_ins; m++) mat[m * n_ins + n] = (n > m) ? g[cur++] : 0.0f;
    for (int64_t m = 0; m < n_ins; m++) {
      for (int64_t n = 0; n < W; n++) {
        float accum = 0.0f;
        for (int64_t k = 0; k < n_ins; k++) {
          const float* xk = (k == 0) ? mlp + p * W : emb + (p * n_emb + (k - 1)) * W;
          accum += (mat[m * n_ins + k] + mat[k * n_ins + m]) * xk[n];
        }
        if (m == 0) mlp_grad[p * W + n] = g[n] + accum;
        else emb_grad[(p * n_emb + (m - 1)) * W + n] = accum;
      }
    }
  }
  free(mat);
}

/* =========================================================================================== */
/* MultiCrossLayer: CPU reference in R/test/utest/core23_layer_test/multi_cross_layer_test.cpp  */
/* v1 :362-372 (fprop) :398-421 (bprop); v2 :375-390 (fprop) :423-470 (bprop)                   */
/* kernels [layers][w], biases [layers][w], outputs [layers][B][w], hiddens v1 [layers][B]      */
/* =========================================================================================== */
void hco_cross_v1_fwd(int64_t B, int64_t w, int layers, const float* x0, const float* kernels,
                      const float* biases, float* outputs, float* hiddens) {
  for (int l = 0; l < layers; l++) {
    const float* xl = (l == 0) ? x0 : outputs + (int64_t)(l - 1) * B * w;
    float* out = outputs + (int64_t)l * B * w;
    float* hid = hiddens + (int64_t)l * B;
    const float* k = kernels + (int64_t)l * w;
    const float* b = biases + (int64_t)l * w;
    for (int64_t r = 0; r < B; r++) {
      float h = 0.0f;
      for (int64_t i = 0; i < w; i++) h = h + xl[r * w + i] * k[i]; /* matrix_vec_mul */
      hid[r] = h;
      for (int64_t i = 0; i < w; i++) {
        float v = x0[r * w + i] * h; /* row_scaling */
        v = v + xl[r * w + i];       /* matrix_add */
        v = v + b[i];                /* matrix_vec_add */
        out[r * w + i] = v;
      }
    }
  }
}

void hco_cross_v1_bwd(int64_t B, int64_t w, int layers, const float* x0, const float* kernels,
                      const float* outputs, const float* hiddens, const float* out_grad,
                      float* in_grad, float* kernel_grads, float* bias_grads) {
  float* t0 = (float*)malloc((size_t)(B * w) * sizeof(float));
  float* t1 = (float*)malloc((size_t)(B * w) * sizeof(float));
  float* tv = (float*)malloc((size_t)B * sizeof(float));
  memset(in_grad, 0, (size_t)(B * w) * sizeof(float));
  for (int l = layers - 1; l >= 0; l--) {
    const float* dY = (l == layers - 1) ? out_grad : t1;
    const float* hid = hiddens + (int64_t)l * B;
    const float* xprev = (l == 0) ? x0 : outputs + (int64_t)(l - 1) * B * w;
    const float* k = kernels + (int64_t)l * w;
    for (int64_t r = 0; r < B; r++)
      for (int64_t i = 0; i < w; i++) in_grad[r * w + i] += dY[r * w + i] * hid[r];
    for (int64_t r = 0; r < B; r++) { /* matrix_pair_mul */
      float s = 0.0f;
      for (int64_t i = 0; i < w; i++) s = s + dY[r * w + i] * x0[r * w + i];
      tv[r] = s;
    }
    for (int64_t i = 0; i < w; i++) { /* row_scaling_sum / rows_sum */
      float s = 0.0f, sb = 0.0f;
      for (int64_t r = 0; r < B; r++) {
        s = s + xprev[r * w + i] * tv[r];
        sb = sb + dY[r * w + i];
      }
      kernel_grads[(int64_t)l * w + i] = s;
      bias_grads[(int64_t)l * w + i] = sb;
    }
    for (int64_t r = 0; r < B; r++) /* out_product + matrix_add */
      for (int64_t i = 0; i < w; i++) t0[r * w + i] = dY[r * w + i] + tv[r] * k[i];
    memcpy(t1, t0, (size_t)(B * w) * sizeof(float));
  }
  for (int64_t i = 0; i < B * w; i++) in_grad[i] += t1[i];
  free(t0);
  free(t1);
  free(tv);
}

/* C[rowA x colB] (+)= op(A) op(B), special_gemm :164-218 */
static void gemm_nn(float* C, const float* A, const float* Bm, int64_t M, int64_t N, int64_t K) {
  for (int64_t r = 0; r < M; r++)
    for (int64_t c = 0; c < N; c++) {
      float acc = 0.f;
      for (int64_t k = 0; k < K; k++) acc = acc + A[r * K + k] * Bm[k * N + c];
      C[r * N + c] = acc;
    }
}
static void gemm_nt(float* C, const float* A, const float* Bm, int64_t M, int64_t N, int64_t K) {
  for (int64_t r = 0; r < M; r++)
    for (int64_t c = 0; c < N; c++) {
      float acc = 0.f;
      for (int64_t k = 0; k < K; k++) acc = acc + A[r * K + k] * Bm[c * K + k];
      C[r * N + c] = acc;
    }
}
static void gemm_tn_acc(float* C, const float* A, const float* Bm, int64_t M, int64_t N,
                        int64_t K) { /* C[M,N] = C*1 + A^T[M,K] B[K,N], A stored [K,M] */
  for (int64_t r = 0; r < M; r++)
    for (int64_t c = 0; c < N; c++) {
      float acc = 0.f;
      for (int64_t k = 0; k < K; k++) acc = acc + A[k * M + r] * Bm[k * N + c];
      C[r * N + c] = C[r * N + c] * 1.0f + acc;
    }
}

/* U [layers][w][p], V [layers][p][w], biases [layers][w]; outputs/hiddens [layers][B][w];
 * XUs [layers][B][p] */
void hco_cross_v2_fwd(int64_t B, int64_t w, int64_t p, int layers, const float* x0,
                      const float* U, const float* V, const float* biases, float* outputs,
                      float* hiddens, float* XUs) {
  for (int l = 0; l < layers; l++) {
    const float* xl = (l == 0) ? x0 : outputs + (int64_t)(l - 1) * B * w;
    float* out = outputs + (int64_t)l * B * w;
    float* hid = hiddens + (int64_t)l * B * w;
    float* xu = XUs + (int64_t)l * B * p;
    gemm_nn(xu, xl, U + (int64_t)l * w * p, B, p, w);
    gemm_nn(hid, xu, V + (int64_t)l * p * w, B, w, p);
    for (int64_t r = 0; r < B; r++)
      for (int64_t i = 0; i < w; i++) {
        hid[r * w + i] = hid[r * w + i] + biases[(int64_t)l * w + i];
        out[r * w + i] = hid[r * w + i] * x0[r * w + i] + xl[r * w + i];
      }
  }
}

void hco_cross_v2_bwd(int64_t B, int64_t w, int64_t p, int layers, const float* x0,
                      const float* U, const float* V, const float* outputs, const float* hiddens,
                      const float* XUs, const float* out_grad, float* in_grad, float* dU,
                      float* dV, float* db) {
  float* t0 = (float*)malloc((size_t)(B * w) * sizeof(float));
  float* t1 = (float*)malloc((size_t)(B * w) * sizeof(float));
  float* t2 = (float*)malloc((size_t)(B * w) * sizeof(float));
  float* t3 = (float*)malloc((size_t)(B * p) * sizeof(float));
  memset(in_grad, 0, (size_t)(B * w) * sizeof(float));
  memset(dU, 0, (size_t)((int64_t)layers * w * p) * sizeof(float));
  memset(dV, 0, (size_t)((int64_t)layers * w * p) * sizeof(float));
  for (int l = layers - 1; l >= 0; l--) {
    const float* dY = (l == layers - 1) ? out_grad : t1;
    const float* hid = hiddens + (int64_t)l * B * w;
    const float* xprev = (l == 0) ? x0 : outputs + (int64_t)(l - 1) * B * w;
    for (int64_t i = 0; i < B * w; i++) {
      t0[i] = dY[i] * x0[i];        /* S0 */
      in_grad[i] += dY[i] * hid[i]; /* dX0 += dY .* hidden */
    }
    for (int64_t i = 0; i < w; i++) { /* db = rows_sum(S0) */
      float s = 0.0f;
      for (int64_t r = 0; r < B; r++) s = s + t0[r * w + i];
      db[(int64_t)l * w + i] = s;
    }
    gemm_tn_acc(dV + (int64_t)l * p * w, XUs + (int64_t)l * B * p, t0, p, w, B); /* dV = XU^T S0 */
    gemm_nt(t3, t0, V + (int64_t)l * p * w, B, p, w);                             /* S1 = S0 V^T */
    gemm_tn_acc(dU + (int64_t)l * w * p, xprev, t3, w, p, B);                     /* dU = H^T S1 */
    gemm_nt(t2, t3, U + (int64_t)l * w * p, B, w, p);                             /* S1 U^T */
    for (int64_t i = 0; i < B * w; i++) t2[i] = dY[i] + t2[i];
    memcpy(t1, t2, (size_t)(B * w) * sizeof(float));
  }
  for (int64_t i = 0; i < B * w; i++) in_grad[i] += t1[i];
  free(t0);
  free(t1);
  free(t2);
  free(t3);
}

/* =========================================================================================== */
/* embedding_collection reference: EmbeddingReferenceCPU::embedding_forward_cpu,                */
/* R/test/utest/embedding_collection/reference_embedding.hpp:72-141; static-table index         */
/* idx = table_start + key / num_shards, R/HugeCTR/embedding/operators/keys_to_indices.cu:24-43 */
/* Sum (0) / Average (1) combiners; bucket_id = lookup_id * batch + b; output per GPU           */
/* feature-major [lookup][local_b][ev] or batch-major [local_b][sum ev].                        */
/* =========================================================================================== */
void hco_keys_to_indices(int64_t n, const int64_t* keys, int64_t table_start, int64_t num_shards,
                         int64_t* idx) {
  for (int64_t i = 0; i < n; i++) idx[i] = table_start + keys[i] / num_shards;
}

void hco_ebc_forward(int64_t batch, int64_t num_lookup, const int32_t* table_ids,
                     const int32_t* ev_sizes, const int32_t* combiners, const int64_t* keys,
                     const int64_t* bucket_range, const int64_t* table_row_start,
                     const int64_t* table_ev_start, const float* tables, int64_t num_gpus,
                     int batch_major, float* out) {
  (void)table_row_start;
  const int64_t bpg = batch / num_gpus;
  int64_t* ev_off = (int64_t*)malloc((size_t)(num_lookup + 1) * sizeof(int64_t));
  ev_off[0] = 0;
  for (int64_t l = 0; l < num_lookup; l++) ev_off[l + 1] = ev_off[l] + ev_sizes[l];
  const int64_t ev_total = ev_off[num_lookup];
  for (int64_t l = 0; l < num_lookup; l++) {
    const int64_t ev = ev_sizes[l];
    const float* tab = tables + table_ev_start[table_ids[l]];
    for (int64_t b = 0; b < batch; b++) {
      const int64_t bucket = l * batch + b, gpu = b / bpg, lb = b % bpg;
      const int64_t s = bucket_range[bucket], e = bucket_range[bucket + 1];
      float* o = out + gpu * ev_total * bpg;
      for (int64_t x = 0; x < ev; x++) {
        float v = 0.f;
        for (int64_t r = s; r < e; r++) v += tab[keys[r] * ev + x];
        if (combiners[l] == 1 && e - s > 0) v /= (float)(e - s);
        const int64_t dst = batch_major ? ev_total * lb + ev_off[l] + x
                                        : ev_off[l] * bpg + lb * ev + x;
        o[dst] = v;
      }
    }
  }
  free(ev_off);
}

/* EmbeddingReferenceCPU::embedding_backward_cpu (reference_embedding.hpp:143-232: per (table,
 * key) the gradients of all buckets holding the key are summed with kahanSum; Average divides
 * the bucket gradient by its key count) + EmbeddingTableCPU::update (embedding_table_cpu.hpp:
 * 94-124, SGD: w += -lr * g / scaler) + AdaGrad as the GPU table does it
 * (R/HugeCTR/embedding_storage/ragged_static_embedding.cu:112-160).  All tables share `ev`.
 * top_grad: per GPU [ev_total * bpg] in the forward's output layout, GPUs concatenated.
 * optimizer: 0 = SGD, 1 = AdaGrad (accum [same shape as tables]). */
void hco_ebc_backward_update(int64_t batch, int64_t num_lookup, const int32_t* table_ids,
                             int64_t ev, const int32_t* combiners, const int64_t* keys,
                             const int64_t* bucket_range, const int64_t* table_row_start,
                             int64_t total_rows, int64_t num_gpus, int batch_major,
                             const float* top_grad, int optimizer, float lr, float scaler,
                             float epsilon, float* tables, float* accum, float ftrl_lambda1,
                             float ftrl_lambda2, float ftrl_beta, float* ftrl_z) {
  const int64_t bpg = batch / num_gpus;
  const int64_t ev_total = ev * num_lookup;
  float* sum = (float*)calloc((size_t)(total_rows * ev), sizeof(float));
  float* comp = (float*)calloc((size_t)(total_rows * ev), sizeof(float)); /* Kahan compensation */
  char* touched = (char*)calloc((size_t)total_rows, 1);
  for (int64_t l = 0; l < num_lookup; l++) {
    const int64_t rs = table_row_start[table_ids[l]];
    for (int64_t b = 0; b < batch; b++) {
      const int64_t bucket = l * batch + b, gpu = b / bpg, lb = b % bpg;
      const int64_t s0 = bucket_range[bucket], e0 = bucket_range[bucket + 1];
      const float* g = top_grad + gpu * ev_total * bpg;
      for (int64_t r = s0; r < e0; r++) {
        const int64_t row = rs + keys[r];
        touched[row] = 1;
        for (int64_t x = 0; x < ev; x++) {
          const int64_t src = batch_major ? lb * ev_total + l * ev + x : l * ev * bpg + lb * ev + x;
          float gi = g[src];
          if (combiners[l] == 1) gi /= (float)(e0 - s0);
          /* kahanSum, reference_embedding.hpp (utils): y = v - c; t = s + y; c = (t - s) - y */
          float y = gi - comp[row * ev + x];
          float t = sum[row * ev + x] + y;
          comp[row * ev + x] = (t - sum[row * ev + x]) - y;
          sum[row * ev + x] = t;
        }
      }
    }
  }
  for (int64_t row = 0; row < total_rows; row++) {
    if (!touched[row]) continue;
    for (int64_t x = 0; x < ev; x++) {
      float gi = sum[row * ev + x];
      if (optimizer == 0) {
        tables[row * ev + x] += -lr * gi / scaler;
      } else if (optimizer == 1) {
        gi = gi / scaler;
        float vi = accum[row * ev + x] + gi * gi;
        accum[row * ev + x] = vi;
        tables[row * ev + x] += -lr * gi / (sqrtf(vi) + epsilon);
      } else {
        /* FtrlOptimizer::update, ragged_static_embedding.cu:159-290 (accum = n, ftrl_z = z) */
        const float l2b = ftrl_lambda2 + ftrl_beta / lr;
        gi = gi / scaler;
        float ni = accum[row * ev + x];
        const float sq = sqrtf(ni + 1.1920929e-07f);
        ni = ni + gi * gi;
        const float sqn = sqrtf(ni + 1.1920929e-07f);
        const float sigma = (sqn - sq) / lr;
        const float w = tables[row * ev + x];
        const float zi = ftrl_z[row * ev + x] + gi - sigma * w;
        const float pq = ((1.f - 2.f * (float)(signbit(zi) != 0)) * ftrl_lambda1 - zi) /
                         (sqn / lr + l2b);
        tables[row * ev + x] = pq * (float)(signbit(ftrl_lambda1 - fabsf(zi)) != 0);
        accum[row * ev + x] = ni;
        ftrl_z[row * ev + x] = zi;
      }
    }
  }
  free(sum);
  free(comp);
  free(touched);
}

/* =========================================================================================== */
/* Synthetic keys: IntPowerLawDataSimulator, R/HugeCTR/include/data_generator.hpp:108-129.      */
/* The reference seeds mt19937 from std::random_device; we fix the seed.  u is drawn as         */
/* std::uniform_real_distribution<float>(0,1) does on libstdc++ (one 32-bit draw / 2^32).       */
/* =========================================================================================== */
typedef struct {
  uint32_t mt[624];
  int idx;
} mt19937_t;

static void mt_seed(mt19937_t* g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937_t* g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

void hco_powerlaw_keys(uint32_t seed, int64_t n, int64_t vocab, float alpha, int64_t* out) {
  mt19937_t g;
  mt_seed(&g, seed);
  const double min_ = 1.0, max_ = (double)vocab; /* min = 0, max = vocab-1 -> max_ = max-min+1 */
  const double offset = -1.0;
  const float one_minus_alpha = 1 - alpha;
  const double a = pow(max_, one_minus_alpha) - pow(min_, one_minus_alpha);
  const double c = pow(min_, one_minus_alpha);
  for (int64_t i = 0; i < n; i++) {
    float u = (float)mt_next(&g) / 4294967296.0f;
    if (u >= 1.0f) u = nextafterf(1.0f, 0.0f);
    double x = u;
    double y = pow(a * x + c, 1.0 / (1.0 - alpha));
    int64_t k = (int64_t)(round(y) + offset);
    if (k < 0) k = 0;
    if (k >= vocab) k = vocab - 1;
    out[i] = k;
  }
}
