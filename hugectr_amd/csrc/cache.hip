// cache.hip -- set-associative LRU embedding cache in HBM and the host<->HBM tiered table on top.
//
// Replaces gpu_cache::gpu_cache (R/gpu_cache/include/nv_gpu_cache.hpp:46-124,
// R/gpu_cache/src/nv_gpu_cache.cu:247-1155: Query / Replace / Update / Dump) and the role of
// gpu_cache::UvmTable (R/gpu_cache/include/uvm_table.hpp:133-174: device table in front of a host
// table) -- BASELINE config 4, SURVEY 8(f)2.  The reference has neither tests nor callers for these
// in tree: parity is against a sequential restatement (oracle/cache_oracle.py), which is itself
// checked against the reference's own kernels stepped through on the host (tests/test_ref_cache_cpu.py).
//
// MI355X-first layout: a slab set of the reference is SET_ASSOCIATIVITY (2) slabs x 32 keys; here
// the whole set is ONE wavefront-wide row of 64 slots, searched with a single 64-lane ballot
// (keys of a set = one 512-byte line).  The reference serialises concurrent sub-warps on a per-set
// mutex, so which key wins a slot depends on scheduling; here the keys of a Replace / Update call
// are sorted by set and every set is walked by exactly one wavefront in position order: results
// are a pure function of the call sequence (= the reference's behaviour under one legal
// interleaving).  The slot rules are the reference's: a key lives in set MurmurHash3_32(key) %
// capacity_in_set, probing starts in slab key % 2; insertion takes the first empty slot in probing
// order, else evicts the least recently used slot (smallest counter; ties: earlier slab in probing
// order, then lower slot); Query / Replace refresh the slot's counter with the global counter,
// which Query advances once per call.
// Calls on one cache are ordered by their streams; calls from unordered streams need external
// ordering (the reference's per-set mutexes are not reproduced).
#include <hip/hip_runtime.h>

#include <cstring>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "hashtable.h"
#include "radix_sort.h"
#include "scan.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kSetSlots = 64;  // SET_ASSOCIATIVITY (2) x SLAB_SIZE (32), nv_gpu_cache.hpp:30-31
constexpr uint32_t kPadSet = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __ballot(p); }

template <typename K>
__device__ __forceinline__ long long widen(K k) {
  return (long long)(sizeof(K) == 4 ? (unsigned long long)(uint32_t)k : (unsigned long long)k);
}

// copy one vector with the 64 lanes of a wavefront (float4 when the caller proved alignment)
template <bool V4>
__device__ __forceinline__ void wave_copy(int lane, int D, float* __restrict__ dst,
                                          const float* __restrict__ src) {
  if (V4) {
    for (int c = lane; c < D / 4; c += 64)
      reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
  } else {
    for (int c = lane; c < D; c += 64) dst[c] = src[c];
  }
}

// update_kernel_overflow_ignore (nv_gpu_cache.cu:225-240): advance the global counter, reset the
// missing length
__global__ void cache_tick_kernel(unsigned long long* global_counter, size_t* d_missing_len) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *global_counter += 1;
    if (d_missing_len) *d_missing_len = 0;
  }
}

// get_kernel (nv_gpu_cache.cu:247-388): one wavefront per key
template <typename K, bool V4>
__global__ void __launch_bounds__(kBlock)
    cache_query_kernel(const K* __restrict__ keys, size_t len, float* __restrict__ values, int D,
                       size_t num_sets, const long long* __restrict__ set_keys,
                       unsigned long long* __restrict__ counters, const float* __restrict__ vals,
                       const unsigned long long* __restrict__ global_counter,
                       uint32_t* __restrict__ miss_flag) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const unsigned long long gc = *global_counter;
  for (size_t i = wave; i < len; i += nwaves) {
    const K key = keys[i];
    const size_t set = (size_t)murmur3_key(key) % num_sets;
    const long long mine = set_keys[set * kSetSlots + lane];
    const unsigned long long hit = ballot64(mine == widen(key));
    if (hit) {
      const int h = __ffsll((long long)hit) - 1;
      const size_t slot = set * kSetSlots + h;
      if (lane == 0) counters[slot] = gc;
      if (values) wave_copy<V4>(lane, D, values + i * (size_t)D, vals + slot * (size_t)D);
    }
    if (lane == 0) miss_flag[i] = hit ? 0u : 1u;
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    cache_emit_missing_kernel(const K* __restrict__ keys, size_t len,
                              const uint32_t* __restrict__ miss_flag,
                              const uint32_t* __restrict__ before, uint64_t* __restrict__ missing_index,
                              K* __restrict__ missing_keys, size_t* __restrict__ d_missing_len,
                              size_t pos_base) {
  // (pos_base: the arrays are a piece of a longer key list -- the indices written are positions of
  //  the whole list; hctr_tiered_lookup queries in pieces)
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len;
       i += (size_t)gridDim.x * kBlock) {
    if (miss_flag[i]) {
      const uint32_t j = before[i];
      if (missing_index) missing_index[j] = pos_base + i;
      if (missing_keys) missing_keys[j] = keys[i];
    }
    if (i == len - 1) *d_missing_len = (size_t)before[i] + miss_flag[i];
  }
}

// set id of every key (padding behind *d_len sorts last), position iota
template <typename K>
__global__ void __launch_bounds__(kBlock)
    cache_setid_kernel(const K* __restrict__ keys, size_t len, const size_t* __restrict__ d_len,
                       size_t num_sets, uint32_t* __restrict__ set_id, uint32_t* __restrict__ pos) {
  const size_t live = d_len ? *d_len : len;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len;
       i += (size_t)gridDim.x * kBlock) {
    set_id[i] = i < live ? (uint32_t)((size_t)murmur3_key(keys[i]) % num_sets) : kPadSet;
    pos[i] = (uint32_t)i;
  }
}

// a cache that fronts a host store with WRITE-BACK (hctr_tiered): a slot whose vector is newer than
// the host row is marked dirty; whoever takes the slot over writes the vector home first
struct WriteBack {
  uint32_t* dirty;   // [slots], nullptr = a plain cache (hctr_cache_*)
  float* host;       // the store as the GPU addresses it, row = key
  size_t host_rows;
};

// insert_replace_kernel (nv_gpu_cache.cu:541-697) / update_kernel (:860-967).  One wavefront per
// run of equal set ids in the sorted list: it keeps the set's 64 keys and counters in registers
// and applies the run's keys one after the other in position order.
//   REPLACE: found -> refresh counter; else first empty slot in probing order; else evict LRU
//   UPDATE : found -> overwrite the vector; else nothing
// value of key at position p: values[(value_index ? value_index[p] : p) * D]
template <typename K, bool REPLACE, bool V4>
__global__ void __launch_bounds__(kBlock)
    cache_modify_kernel(const K* __restrict__ keys, size_t len,
                        const uint32_t* __restrict__ sorted_set, const uint32_t* __restrict__ sorted_pos,
                        const float* __restrict__ values, const uint64_t* __restrict__ value_index,
                        int D, long long* __restrict__ set_keys,
                        unsigned long long* __restrict__ counters, float* __restrict__ vals,
                        const unsigned long long* __restrict__ global_counter, long long empty_key,
                        WriteBack wb) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const unsigned long long gc = *global_counter;
  for (size_t i = wave; i < len; i += nwaves) {
    const uint32_t set = sorted_set[i];
    if (set == kPadSet) continue;
    if (i > 0 && sorted_set[i - 1] == set) continue;  // not the head of its run
    long long my_key = set_keys[(size_t)set * kSetSlots + lane];
    unsigned long long my_cnt = counters[(size_t)set * kSetSlots + lane];
    // the run is consumed 64 entries at a time: one coalesced load of (position, key) per chunk,
    // then the entries are broadcast lane by lane -- a set that receives 100 k copies of a hot
    // key costs a few cycles per copy instead of two dependent global loads
    for (size_t j0 = i;; j0 += 64) {
      const size_t jj = j0 + lane;
      const bool in_run = jj < len && sorted_set[jj] == set;
      const uint32_t my_p = in_run ? sorted_pos[jj] : 0u;
      const long long my_k = in_run ? widen(keys[my_p]) : 0ll;
      const int cnt = __popcll(ballot64(in_run));  // the run is contiguous: lanes [0, cnt)
      for (int e = 0; e < cnt; e++) {
        const uint32_t p = (uint32_t)__shfl((int)my_p, e);
        const long long k64 = __shfl(my_k, e);
        const unsigned long long hit = ballot64(my_key == k64);
        int target = -1;
        if (hit) {
          const int h = __ffsll((long long)hit) - 1;
          if (REPLACE) {
            if (lane == h && my_cnt != gc) {  // (later copies of a key of this call: nothing to do)
              my_cnt = gc;
              counters[(size_t)set * kSetSlots + h] = gc;
            }
          } else {
            target = h;
          }
        } else if (REPLACE) {
          // probing order: slab (key % 2) first (Mod_Hash, nv_gpu_cache.hpp:49), then the other
          const int first_slab = (int)((unsigned long long)k64 & 1ull);
          const unsigned long long empties = ballot64(my_key == empty_key);
          unsigned long long cand = empties;
          if (!cand) {  // LRU: smallest counter, ties by probing order
            unsigned long long mn = my_cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              const unsigned long long other = __shfl_xor(mn, o);
              mn = other < mn ? other : mn;
            }
            cand = ballot64(my_cnt == mn);
          }
          const unsigned long long lo = cand & 0xFFFFFFFFull, hi = cand >> 32;
          const unsigned long long a = first_slab == 0 ? lo : hi;  // first slab in probing order
          const unsigned long long b = first_slab == 0 ? hi : lo;
          if (a)
            target = (__ffsll((long long)a) - 1) + 32 * first_slab;
          else
            target = (__ffsll((long long)b) - 1) + 32 * (1 - first_slab);
          if (REPLACE && wb.dirty != nullptr) {
            // the slot's present tenant goes home first when its vector is the newer one
            const size_t slot = (size_t)set * kSetSlots + target;
            const long long victim = __shfl(my_key, target);
            if (wb.dirty[slot] != 0u) {  // (wave-uniform: every lane reads the same word)
              if (victim >= 0 && (size_t)victim < wb.host_rows)
                wave_copy<V4>(lane, D, wb.host + (size_t)victim * (size_t)D, vals + slot * (size_t)D);
              __builtin_amdgcn_wave_barrier();
              if (lane == 0) wb.dirty[slot] = 0u;
            }
          }
          if (lane == target) {
            my_key = k64;
            my_cnt = gc;
            set_keys[(size_t)set * kSetSlots + target] = k64;
            counters[(size_t)set * kSetSlots + target] = gc;
          }
        }
        if (target >= 0) {
          const float* src = values + (value_index ? (size_t)value_index[p] : (size_t)p) * (size_t)D;
          wave_copy<V4>(lane, D, vals + ((size_t)set * kSetSlots + target) * (size_t)D, src);
        }
      }
      if (cnt < 64) break;
    }
  }
}

// dump_kernel (nv_gpu_cache.cu:1080-1153): keys of the sets [start, end) in (set, slot) order
__global__ void __launch_bounds__(kBlock)
    cache_dump_count_kernel(const long long* __restrict__ set_keys, size_t start, size_t n_sets,
                            long long empty_key, uint32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  for (size_t s = wave; s < n_sets; s += nwaves) {
    const unsigned long long live = ballot64(set_keys[(start + s) * kSetSlots + lane] != empty_key);
    if (lane == 0) counts[s] = (uint32_t)__popcll(live);
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    cache_dump_emit_kernel(const long long* __restrict__ set_keys, size_t start, size_t n_sets,
                           long long empty_key, const uint32_t* __restrict__ offsets,
                           K* __restrict__ out, size_t* __restrict__ d_counter) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  for (size_t s = wave; s < n_sets; s += nwaves) {
    const long long k = set_keys[(start + s) * kSetSlots + lane];
    const unsigned long long live = ballot64(k != empty_key);
    if (k != empty_key) {
      const int rank = __popcll(live & ((1ull << lane) - 1ull));
      out[offsets[s] + rank] = (K)k;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *d_counter = (size_t)offsets[n_sets];
}

__global__ void __launch_bounds__(kBlock)
    cache_init_kernel(size_t slots, long long empty_key, long long* __restrict__ set_keys,
                      unsigned long long* __restrict__ counters) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < slots;
       i += (size_t)gridDim.x * kBlock) {
    set_keys[i] = empty_key;
    counters[i] = 0ull;
  }
}

// ---- tiered table: rows missing in the cache come straight out of pinned host memory -----------
template <bool V4>
__global__ void __launch_bounds__(kBlock)
    tier_fill_kernel(const long long* __restrict__ miss_keys, const uint64_t* __restrict__ miss_index,
                     const size_t* __restrict__ d_missing_len, size_t host_rows, int D,
                     const float* __restrict__ host, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  const size_t n = *d_missing_len;
  for (size_t j = wave; j < n; j += nwaves) {
    const long long key = miss_keys[j];
    float* dst = out + (size_t)miss_index[j] * (size_t)D;
    if (key >= 0 && (size_t)key < host_rows) {
      wave_copy<V4>(lane, D, dst, host + (size_t)key * (size_t)D);
    } else {  // unknown row: zeros (UvmTable's default_value, uvm_table.hpp:137)
      for (int c = lane; c < D; c += 64) dst[c] = 0.0f;
    }
  }
}

// the pieces' miss lists (piece p: entries [p * piece, p * piece + cnt[p]) of the segmented arrays)
// put end to end, in piece order = position order; *d_total = the whole count
__global__ void __launch_bounds__(kBlock)
    tier_concat_kernel(int pieces, size_t piece, const size_t* __restrict__ cnt,
                       const long long* __restrict__ seg_keys, const uint64_t* __restrict__ seg_index,
                       long long* __restrict__ keys, uint64_t* __restrict__ index,
                       size_t* __restrict__ d_total) {
  size_t base = 0;
  for (int p = 0; p < pieces; p++) {
    const size_t n = cnt[p];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
         i += (size_t)gridDim.x * kBlock) {
      keys[base + i] = seg_keys[(size_t)p * piece + i];
      index[base + i] = seg_index[(size_t)p * piece + i];
    }
    base += n;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *d_total = base;
}

// update of unique rows: new = (add ? old : 0) + alpha * value.  A cached row is updated in the
// cache ONLY and marked dirty (it goes home when its slot is taken over, or at hctr_tiered_flush);
// a row that is not cached is updated in the host store.  Write-through (round 4-5) sent every
// distinct row of a batch over the host link -- 128 MB for 1 M power-law keys, 2.3 ms -- although
// the lookup had just put nearly all of them into the cache.
template <bool V4>
__global__ void __launch_bounds__(kBlock)
    tier_scatter_kernel(const long long* __restrict__ keys, size_t len,
                        const float* __restrict__ values, int add, float alpha, size_t host_rows,
                        int D,
                        size_t num_sets, const long long* __restrict__ set_keys,
                        float* __restrict__ vals, float* __restrict__ host,
                        uint32_t* __restrict__ dirty) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  for (size_t i = wave; i < len; i += nwaves) {
    const long long key = keys[i];
    if (key < 0 || (size_t)key >= host_rows) continue;
    const size_t set = (size_t)murmur3_key(key) % num_sets;
    const unsigned long long hit = ballot64(set_keys[set * kSetSlots + lane] == key);
    const size_t slot = set * kSetSlots + (hit ? (size_t)(__ffsll((long long)hit) - 1) : 0);
    float* row = hit ? vals + slot * (size_t)D : host + (size_t)key * (size_t)D;
    const float* v = values + i * (size_t)D;
    if (V4) {
      for (int c = lane; c < D / 4; c += 64) {
        float4 x = reinterpret_cast<const float4*>(v)[c];
        x = make_float4(alpha * x.x, alpha * x.y, alpha * x.z, alpha * x.w);
        if (add) {
          const float4 o = reinterpret_cast<const float4*>(row)[c];
          x = make_float4(o.x + x.x, o.y + x.y, o.z + x.z, o.w + x.w);
        }
        reinterpret_cast<float4*>(row)[c] = x;
      }
    } else {
      for (int c = lane; c < D; c += 64) row[c] = add ? row[c] + alpha * v[c] : alpha * v[c];
    }
    if (hit && lane == 0) dirty[slot] = 1u;
  }
}

// every dirty slot's vector goes home (checkpoints, host-side reads of the store)
template <bool V4>
__global__ void __launch_bounds__(kBlock)
    tier_flush_kernel(size_t slots, int D, const long long* __restrict__ set_keys,
                      const float* __restrict__ vals, uint32_t* __restrict__ dirty,
                      float* __restrict__ host, size_t host_rows) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock;
  for (size_t s0 = wave * 64; s0 < slots; s0 += nwaves * 64) {
    unsigned long long todo = ballot64(s0 + lane < slots && dirty[s0 + lane] != 0u);
    while (todo) {
      const int b = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const size_t slot = s0 + b;
      const long long key = set_keys[slot];
      if (key >= 0 && (size_t)key < host_rows)
        wave_copy<V4>(lane, D, host + (size_t)key * (size_t)D, vals + slot * (size_t)D);
      if (lane == 0) dirty[slot] = 0u;
    }
  }
}

}  // namespace
}  // namespace hctr

using namespace hctr;

struct hctr_cache {
  size_t num_sets = 0;
  int D = 0;
  int key_type = HCTR_KEY_I64;
  long long empty_key = 0;
  long long* set_keys = nullptr;
  unsigned long long* counters = nullptr;
  float* vals = nullptr;
  unsigned long long* global_counter = nullptr;
  WriteBack wb = {nullptr, nullptr, 0};  // set by hctr_tiered_create: the cache fronts a host store
  // scratch, sized for `cap` keys per call (grows on demand)
  size_t cap = 0;
  uint32_t *flags = nullptr, *before = nullptr;
  uint32_t *set_in = nullptr, *set_out = nullptr, *pos_in = nullptr, *pos_out = nullptr;
  void* sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  unsigned long long *tile_sums = nullptr, *d_total = nullptr;

  void free_scratch() {
    void* ptrs[] = {flags, before, set_in, set_out, pos_in, pos_out, sort_temp, tile_sums, d_total};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
    flags = before = set_in = set_out = pos_in = pos_out = nullptr;
    sort_temp = nullptr;
    tile_sums = d_total = nullptr;
    cap = 0;
  }
  int reserve(size_t n, hipStream_t s) {
    if (n <= cap) return HCTR_OK;
    if (cap) HCTR_HIP(hipStreamSynchronize(s));  // earlier calls may still read the old scratch
    free_scratch();
    size_t c = 1024;
    while (c < n) c *= 2;
    HCTR_HIP(hipMalloc(&flags, c * 4));
    HCTR_HIP(hipMalloc(&before, (c + 1) * 4));
    HCTR_HIP(hipMalloc(&set_in, c * 4));
    HCTR_HIP(hipMalloc(&set_out, c * 4));
    HCTR_HIP(hipMalloc(&pos_in, c * 4));
    HCTR_HIP(hipMalloc(&pos_out, c * 4));
    HCTR_HIP(hipMalloc(&tile_sums, (c / 1024 + 2) * 8));
    HCTR_HIP(hipMalloc(&d_total, 8));
    sort_temp_bytes = radix_sort_temp_bytes(c);
    HCTR_HIP(hipMalloc(&sort_temp, sort_temp_bytes));
    cap = c;
    return HCTR_OK;
  }
};

namespace {

bool vec4_ok(int D, const void* a, const void* b) {
  return D % 4 == 0 && reinterpret_cast<uintptr_t>(a) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(b) % 16 == 0;
}

template <typename K>
int cache_query_typed(hctr_cache* c, const K* keys, size_t len, float* values,
                      uint64_t* missing_index, K* missing_keys, size_t* d_missing_len,
                      hipStream_t s, size_t pos_base = 0) {
  const int grid = grid_for(len * 64, kBlock, 1 << 16);
  if (vec4_ok(c->D, values, c->vals))
    hipLaunchKernelGGL((cache_query_kernel<K, true>), dim3(grid), dim3(kBlock), 0, s, keys, len,
                       values, c->D, c->num_sets, c->set_keys, c->counters, c->vals,
                       c->global_counter, c->flags);
  else
    hipLaunchKernelGGL((cache_query_kernel<K, false>), dim3(grid), dim3(kBlock), 0, s, keys, len,
                       values, c->D, c->num_sets, c->set_keys, c->counters, c->vals,
                       c->global_counter, c->flags);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(exclusive_scan_to_offsets<uint32_t>(c->flags, len, c->tile_sums, c->d_total, c->before,
                                               s));
  hipLaunchKernelGGL(cache_emit_missing_kernel<K>, dim3(grid_for(len, kBlock, 4096)), dim3(kBlock),
                     0, s, keys, len, c->flags, c->before, missing_index, missing_keys,
                     d_missing_len, pos_base);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// Replace / Update on up to `len` keys (live count optionally on the device)
template <typename K>
int cache_modify_typed(hctr_cache* c, bool replace, const K* keys, size_t len, const size_t* d_len,
                       const float* values, const uint64_t* value_index, hipStream_t s) {
  HCTR_TRY(c->reserve(len, s));
  hipLaunchKernelGGL(cache_setid_kernel<K>, dim3(grid_for(len, kBlock, 4096)), dim3(kBlock), 0, s,
                     keys, len, d_len, c->num_sets, c->set_in, c->pos_in);
  HCTR_LAUNCH_CHECK();
  // stable sort by set id: inside a set the keys keep their position order.  Key bits: the set
  // ids plus one, so that the padding id (all ones) of positions beyond *d_len sorts last
  int set_bits = 1;
  while (set_bits < 31 && ((size_t)1 << set_bits) < c->num_sets) set_bits++;
  HCTR_TRY(radix_sort_pairs_u32(c->sort_temp, c->sort_temp_bytes, c->set_in, c->set_out, c->pos_in,
                                c->pos_out, len, set_bits + 1, s));
  const int grid = grid_for(len * 64, kBlock, 1 << 16);
  const bool v4 = vec4_ok(c->D, values, c->vals);
#define HCTR_CM(R_, V_)                                                                          \
  hipLaunchKernelGGL((cache_modify_kernel<K, R_, V_>), dim3(grid), dim3(kBlock), 0, s, keys, len, \
                     c->set_out, c->pos_out, values, value_index, c->D, c->set_keys, c->counters, \
                     c->vals, c->global_counter, c->empty_key, c->wb)
  if (replace) {
    if (v4) HCTR_CM(true, true);
    else HCTR_CM(true, false);
  } else {
    if (v4) HCTR_CM(false, true);
    else HCTR_CM(false, false);
  }
#undef HCTR_CM
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // namespace

extern "C" {

int hctr_cache_create(size_t capacity_in_set, int vec_size, int key_type, hctr_cache** out) {
  HCTR_REQUIRE(out && capacity_in_set > 0 && capacity_in_set < 0xFFFFFFF0ull, "capacity_in_set");
  HCTR_REQUIRE(vec_size > 0, "vec_size");
  HCTR_REQUIRE(key_type == HCTR_KEY_U32 || key_type == HCTR_KEY_I64, "key_type");
  hctr_cache* c = new hctr_cache();
  c->num_sets = capacity_in_set;
  c->D = vec_size;
  c->key_type = key_type;
  c->empty_key = key_type == HCTR_KEY_U32 ? KeyTraits<uint32_t>::empty : KeyTraits<long long>::empty;
  const size_t slots = capacity_in_set * kSetSlots;
  bool ok = hipMalloc(&c->set_keys, slots * 8) == hipSuccess &&
            hipMalloc(&c->counters, slots * 8) == hipSuccess &&
            hipMalloc(&c->vals, slots * (size_t)vec_size * sizeof(float)) == hipSuccess &&
            hipMalloc(&c->global_counter, 8) == hipSuccess;
  if (ok) ok = hipMemset(c->global_counter, 0, 8) == hipSuccess;
  if (!ok) {
    set_error("hctr_cache_create: allocation failed");
    if (c->set_keys) (void)hipFree(c->set_keys);
    if (c->counters) (void)hipFree(c->counters);
    if (c->vals) (void)hipFree(c->vals);
    if (c->global_counter) (void)hipFree(c->global_counter);
    delete c;
    return HCTR_ERR_HIP;
  }
  hipLaunchKernelGGL(cache_init_kernel, dim3(grid_for(slots, kBlock, 4096)), dim3(kBlock), 0, 0,
                     slots, c->empty_key, c->set_keys, c->counters);
  (void)hipDeviceSynchronize();
  *out = c;
  return HCTR_OK;
}

int hctr_cache_destroy(hctr_cache* c) {
  if (!c) return HCTR_OK;
  (void)hipDeviceSynchronize();
  c->free_scratch();
  (void)hipFree(c->set_keys);
  (void)hipFree(c->counters);
  (void)hipFree(c->vals);
  (void)hipFree(c->global_counter);
  delete c;
  return HCTR_OK;
}

int hctr_cache_query(hctr_cache* c, const void* keys, size_t len, float* values,
                     uint64_t* missing_index, void* missing_keys, size_t* d_missing_len,
                     hctr_stream_t stream) {
  HCTR_REQUIRE(c && d_missing_len, "null pointer");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(cache_tick_kernel, dim3(1), dim3(64), 0, s, c->global_counter, d_missing_len);
  HCTR_LAUNCH_CHECK();
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(keys, "null pointer");
  HCTR_REQUIRE(len < 0xFFFFFFF0ull, "len");
  HCTR_TRY(c->reserve(len, s));
  if (c->key_type == HCTR_KEY_U32)
    return cache_query_typed<uint32_t>(c, (const uint32_t*)keys, len, values, missing_index,
                                       (uint32_t*)missing_keys, d_missing_len, s);
  return cache_query_typed<long long>(c, (const long long*)keys, len, values, missing_index,
                                      (long long*)missing_keys, d_missing_len, s);
}

int hctr_cache_replace(hctr_cache* c, const void* keys, size_t len, const float* values,
                       hctr_stream_t stream) {
  HCTR_REQUIRE(c, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && values, "null pointer");
  HCTR_REQUIRE(len < 0xFFFFFFF0ull, "len");
  hipStream_t s = as_stream(stream);
  if (c->key_type == HCTR_KEY_U32)
    return cache_modify_typed<uint32_t>(c, true, (const uint32_t*)keys, len, nullptr, values,
                                        nullptr, s);
  return cache_modify_typed<long long>(c, true, (const long long*)keys, len, nullptr, values,
                                       nullptr, s);
}

int hctr_cache_update(hctr_cache* c, const void* keys, size_t len, const float* values,
                      hctr_stream_t stream) {
  HCTR_REQUIRE(c, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && values, "null pointer");
  HCTR_REQUIRE(len < 0xFFFFFFF0ull, "len");
  hipStream_t s = as_stream(stream);
  if (c->key_type == HCTR_KEY_U32)
    return cache_modify_typed<uint32_t>(c, false, (const uint32_t*)keys, len, nullptr, values,
                                        nullptr, s);
  return cache_modify_typed<long long>(c, false, (const long long*)keys, len, nullptr, values,
                                       nullptr, s);
}

int hctr_cache_dump(hctr_cache* c, void* keys, size_t* d_dump_counter, size_t start_set_index,
                    size_t end_set_index, hctr_stream_t stream) {
  HCTR_REQUIRE(c && d_dump_counter, "null pointer");
  HCTR_REQUIRE(start_set_index <= end_set_index && end_set_index <= c->num_sets,
               "set range out of bounds");
  hipStream_t s = as_stream(stream);
  const size_t n = end_set_index - start_set_index;
  if (n == 0) {
    HCTR_HIP(hipMemsetAsync(d_dump_counter, 0, sizeof(size_t), s));
    return HCTR_OK;
  }
  HCTR_REQUIRE(keys, "null pointer");
  HCTR_TRY(c->reserve(n, s));
  const int grid = grid_for(n * 64, kBlock, 1 << 16);
  hipLaunchKernelGGL(cache_dump_count_kernel, dim3(grid), dim3(kBlock), 0, s, c->set_keys,
                     start_set_index, n, c->empty_key, c->flags);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(exclusive_scan_to_offsets<uint32_t>(c->flags, n, c->tile_sums, c->d_total, c->before, s));
  if (c->key_type == HCTR_KEY_U32)
    hipLaunchKernelGGL(cache_dump_emit_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s,
                       c->set_keys, start_set_index, n, c->empty_key, c->before, (uint32_t*)keys,
                       d_dump_counter);
  else
    hipLaunchKernelGGL(cache_dump_emit_kernel<long long>, dim3(grid), dim3(kBlock), 0, s,
                       c->set_keys, start_set_index, n, c->empty_key, c->before, (long long*)keys,
                       d_dump_counter);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_cache_capacity_in_set(const hctr_cache* c) { return c ? c->num_sets : 0; }

}  // extern "C"

// ---- host <-> HBM tiered table -------------------------------------------------------------------
struct hctr_tiered {
  hctr_cache* cache = nullptr;
  size_t rows = 0;
  int D = 0;
  float* host = nullptr;      // pinned, device-mapped
  float* host_dev = nullptr;  // the same memory as the GPU addresses it
  size_t cap = 0;
  long long* miss_keys = nullptr;
  uint64_t* miss_index = nullptr;
  size_t* d_missing_len = nullptr;
  // lookup in pieces (hctr_tiered_lookup): the pieces' own miss lists and counts, the stream the
  // host link's copies run on next to the cache queries of the following pieces
  static constexpr int kPieces = 4;
  long long* seg_keys = nullptr;
  uint64_t* seg_index = nullptr;
  size_t* d_piece_cnt = nullptr;  // [kPieces]
  uint32_t* dirty = nullptr;  // [cache slots] the cached vector is newer than the host row
  hipStream_t fill_stream = nullptr;
  hipEvent_t ev_piece[kPieces] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_filled = nullptr;

  int reserve(size_t n, hipStream_t s) {
    if (n <= cap) return HCTR_OK;
    if (cap) HCTR_HIP(hipStreamSynchronize(s));
    for (void* q : {(void*)miss_keys, (void*)miss_index, (void*)seg_keys, (void*)seg_index})
      if (q) (void)hipFree(q);
    miss_keys = seg_keys = nullptr;
    miss_index = seg_index = nullptr;
    size_t c = 1024;
    while (c < n) c *= 2;
    HCTR_HIP(hipMalloc(&miss_keys, c * 8));
    HCTR_HIP(hipMalloc(&miss_index, c * 8));
    HCTR_HIP(hipMalloc(&seg_keys, c * 8));
    HCTR_HIP(hipMalloc(&seg_index, c * 8));
    cap = c;
    return HCTR_OK;
  }
};

extern "C" {

int hctr_tiered_create(size_t host_rows, int vec_size, size_t cache_capacity_in_set,
                       hctr_tiered** out) {
  HCTR_REQUIRE(out && host_rows > 0 && vec_size > 0 && cache_capacity_in_set > 0, "arguments");
  hctr_tiered* t = new hctr_tiered();
  t->rows = host_rows;
  t->D = vec_size;
  int rc = hctr_cache_create(cache_capacity_in_set, vec_size, HCTR_KEY_I64, &t->cache);
  if (rc != HCTR_OK) {
    delete t;
    return rc;
  }
  const size_t bytes = host_rows * (size_t)vec_size * sizeof(float);
  bool ok = hipHostMalloc((void**)&t->host, bytes, hipHostMallocMapped | hipHostMallocPortable) ==
            hipSuccess;
  if (ok) ok = hipHostGetDevicePointer((void**)&t->host_dev, t->host, 0) == hipSuccess;
  if (ok) ok = hipMalloc(&t->d_missing_len, sizeof(size_t)) == hipSuccess;
  const size_t slots = cache_capacity_in_set * (size_t)kSetSlots;
  if (ok) ok = hipMalloc(&t->dirty, slots * sizeof(uint32_t)) == hipSuccess &&
               hipMemset(t->dirty, 0, slots * sizeof(uint32_t)) == hipSuccess;
  if (ok) ok = hipMalloc(&t->d_piece_cnt, hctr_tiered::kPieces * sizeof(size_t)) == hipSuccess;
  if (ok) ok = hipStreamCreateWithFlags(&t->fill_stream, hipStreamNonBlocking) == hipSuccess;
  for (int p = 0; ok && p < hctr_tiered::kPieces; p++)
    ok = hipEventCreateWithFlags(&t->ev_piece[p], hipEventDisableTiming) == hipSuccess;
  if (ok) ok = hipEventCreateWithFlags(&t->ev_filled, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    set_error("hctr_tiered_create: host / device allocation failed");
    if (t->host) (void)hipHostFree(t->host);
    hctr_cache_destroy(t->cache);
    delete t;
    return HCTR_ERR_HIP;
  }
  memset(t->host, 0, bytes);
  t->cache->wb = {t->dirty, t->host_dev, host_rows};
  *out = t;
  return HCTR_OK;
}

int hctr_tiered_destroy(hctr_tiered* t) {
  if (!t) return HCTR_OK;
  (void)hipDeviceSynchronize();
  hctr_cache_destroy(t->cache);
  for (void* q : {(void*)t->miss_keys, (void*)t->miss_index, (void*)t->seg_keys,
                  (void*)t->seg_index, (void*)t->d_missing_len, (void*)t->d_piece_cnt,
                  (void*)t->dirty})
    if (q) (void)hipFree(q);
  if (t->fill_stream) (void)hipStreamDestroy(t->fill_stream);
  for (hipEvent_t e : t->ev_piece)
    if (e) (void)hipEventDestroy(e);
  if (t->ev_filled) (void)hipEventDestroy(t->ev_filled);
  if (t->host) (void)hipHostFree(t->host);
  delete t;
  return HCTR_OK;
}

float* hctr_tiered_host_rows(hctr_tiered* t) { return t ? t->host : nullptr; }
hctr_cache* hctr_tiered_cache(hctr_tiered* t) { return t ? t->cache : nullptr; }

int hctr_tiered_lookup(hctr_tiered* t, const int64_t* keys, size_t len, float* out,
                       size_t* d_missing_len, hctr_stream_t stream) {
  HCTR_REQUIRE(t, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && out, "null pointer");
  hipStream_t s = as_stream(stream);
  HCTR_TRY(t->reserve(len, s));
  size_t* dml = d_missing_len ? d_missing_len : t->d_missing_len;
  hctr_cache* c = t->cache;
  // The three steps -- (1) hits copied out of the cache, misses listed; (2) missing rows host
  // table -> output over the host link, by the GPU itself; (3) those rows into the cache -- as a
  // pipeline over pieces of the key list (UvmTable::query's double-buffered H2D,
  // R/gpu_cache/include/uvm_table.hpp:128-174): piece p's copy over the link runs on a private
  // stream next to piece p + 1's cache query (an HBM-bound kernel), so the link's time hides behind
  // the queries instead of following them (round 4: query 315 us + fill 225 us + replace 130 us one
  // after the other).  ONE tick of the cache's clock for the whole call and one Replace over the
  // pieces' miss lists put end to end in position order: the cache ends in exactly the state of
  // the unpieced call (hits only refresh their slot's stamp with the call's clock value).
  // HCTR_TIER_PIECES=1: the unpieced form.
  int pieces = hctr_tiered::kPieces;
  if (const char* e = getenv("HCTR_TIER_PIECES")) {
    const int v = atoi(e);
    if (v >= 1 && v <= hctr_tiered::kPieces) pieces = v;
  }
  // (short lists stay whole: four query launches of a few microseconds each would only add gaps;
  //  HCTR_TIER_PIECE_MIN lowers the bound for tests)
  size_t piece_min = 65536;
  if (const char* e = getenv("HCTR_TIER_PIECE_MIN")) piece_min = (size_t)atoll(e);
  if (len < piece_min) pieces = 1;
  hipLaunchKernelGGL(cache_tick_kernel, dim3(1), dim3(64), 0, s, c->global_counter, dml);
  HCTR_LAUNCH_CHECK();
  HCTR_TRY(c->reserve(len, s));
  const size_t piece = ceil_div<size_t>(len, (size_t)pieces);
  const bool v4 = vec4_ok(t->D, out, t->host_dev);
  for (int p = 0; p < pieces; p++) {
    const size_t b = (size_t)p * piece;
    const size_t n = b >= len ? 0 : (len - b < piece ? len - b : piece);
    if (n == 0) {
      HCTR_HIP(hipMemsetAsync(t->d_piece_cnt + p, 0, sizeof(size_t), s));
      continue;
    }
    long long* mk = pieces == 1 ? t->miss_keys : t->seg_keys + b;
    uint64_t* mi = pieces == 1 ? t->miss_index : t->seg_index + b;
    size_t* cnt = pieces == 1 ? dml : t->d_piece_cnt + p;
    HCTR_TRY(cache_query_typed<long long>(c, (const long long*)keys + b, n, out + b * (size_t)t->D,
                                          mi, mk, cnt, s, b));
    hipStream_t fs = pieces == 1 ? s : t->fill_stream;
    if (fs != s) {
      HCTR_HIP(hipEventRecord(t->ev_piece[p], s));
      HCTR_HIP(hipStreamWaitEvent(fs, t->ev_piece[p], 0));
    }
    // in pieces the copy shares the chip with the next piece's query: a few hundred wavefronts keep
    // the host link full (64 GB/s x a few us of latency = a few hundred rows in flight); the 16 k
    // workgroups of the whole-list form, parked on link latency, were measured to hold the query's
    // wave slots (query 84 -> 300 us; lookup 1034 / 957 / 808 us at 1024 / 256 / 64 workgroups).
    // The last piece's copy has nobody beside it and takes the full grid.
    const char* fg_env = getenv("HCTR_TIER_FILL_GRID");
    const int fcap = (pieces == 1 || p == pieces - 1) ? (1 << 14) : (fg_env ? atoi(fg_env) : 64);
    const int grid = grid_for(n * 64, kBlock, fcap);
    if (v4)
      hipLaunchKernelGGL(tier_fill_kernel<true>, dim3(grid), dim3(kBlock), 0, fs, mk, mi, cnt,
                         t->rows, t->D, t->host_dev, out);
    else
      hipLaunchKernelGGL(tier_fill_kernel<false>, dim3(grid), dim3(kBlock), 0, fs, mk, mi, cnt,
                         t->rows, t->D, t->host_dev, out);
    HCTR_LAUNCH_CHECK();
  }
  if (pieces > 1) {
    hipLaunchKernelGGL(tier_concat_kernel, dim3(64), dim3(kBlock), 0, s, pieces, piece,
                       t->d_piece_cnt, t->seg_keys, t->seg_index, t->miss_keys, t->miss_index, dml);
    HCTR_LAUNCH_CHECK();
    HCTR_HIP(hipEventRecord(t->ev_filled, t->fill_stream));
    HCTR_HIP(hipStreamWaitEvent(s, t->ev_filled, 0));
  }
  // (3) the missing rows into the cache (values are read back from the output rows just written)
  return cache_modify_typed<long long>(c, true, t->miss_keys, len, dml, out, t->miss_index, s);
}

int hctr_tiered_scatter(hctr_tiered* t, const int64_t* unique_keys, size_t len, const float* values,
                        int add, float alpha, hctr_stream_t stream) {
  HCTR_REQUIRE(t, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(unique_keys && values, "null pointer");
  hipStream_t s = as_stream(stream);
  const int grid = grid_for(len * 64, kBlock, 1 << 14);
  const hctr_cache* c = t->cache;
  if (vec4_ok(t->D, values, c->vals))
    hipLaunchKernelGGL(tier_scatter_kernel<true>, dim3(grid), dim3(kBlock), 0, s,
                       (const long long*)unique_keys, len, values, add, alpha, t->rows, t->D, c->num_sets,
                       c->set_keys, c->vals, t->host_dev, t->dirty);
  else
    hipLaunchKernelGGL(tier_scatter_kernel<false>, dim3(grid), dim3(kBlock), 0, s,
                       (const long long*)unique_keys, len, values, add, alpha, t->rows, t->D, c->num_sets,
                       c->set_keys, c->vals, t->host_dev, t->dirty);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_tiered_flush(hctr_tiered* t, hctr_stream_t stream) {
  HCTR_REQUIRE(t, "null handle");
  hipStream_t s = as_stream(stream);
  const hctr_cache* c = t->cache;
  const size_t slots = c->num_sets * (size_t)kSetSlots;
  const int grid = grid_for(slots, kBlock, 4096);
  if (vec4_ok(t->D, c->vals, t->host_dev))
    hipLaunchKernelGGL(tier_flush_kernel<true>, dim3(grid), dim3(kBlock), 0, s, slots, t->D,
                       c->set_keys, c->vals, t->dirty, t->host_dev, t->rows);
  else
    hipLaunchKernelGGL(tier_flush_kernel<false>, dim3(grid), dim3(kBlock), 0, s, slots, t->D,
                       c->set_keys, c->vals, t->dirty, t->host_dev, t->rows);
  HCTR_LAUNCH_CHECK();
  HCTR_HIP(hipStreamSynchronize(s));
  return HCTR_OK;
}

}  // extern "C"

// ---- arbitrary keys in front of the tiered table (gpu_cache::UvmTable as a whole) -----------------
// UvmTable maps a key through a device HashBlock to a row of the device store and, failing that,
// through a host HashBlock to a row of the host store (uvm_table.hpp:127-174, uvm_table.cu:318-497).
// Here ONE index answers for both tiers: the path's own open-addressing map (hashtable.hip) hands
// every key of an int64 / uint32 key space -- 10^10 rows of BASELINE configs[3] or anything else --
// a row of the pinned host store on first touch (rows are handed out in order of first occurrence,
// deterministic), and the set-associative cache above holds the hot ROWS.  16 bytes of HBM per slot
// of the index (capacity / 0.75 slots) is what a key costs; its vector costs host memory only.
struct hctr_uvm {
  hctr_tiered* tier = nullptr;
  hctr::HashTable index;
  size_t capacity = 0, max_batch = 0;
  int D = 0, key_type = HCTR_KEY_I64;
  float default_value = 0.f;
  uint64_t* d_rows = nullptr;   // [max_batch] row of every key of the call in flight
  void* d_keys = nullptr;       // [max_batch] staging of add()'s host keys
  float* d_vecs = nullptr;      // [max_batch][D] staging of add()'s host vectors
};

namespace hctr {
namespace {

// positions whose key is unknown to the index (row = SIZE_MAX) read the default vector
__global__ void __launch_bounds__(kBlock)
    uvm_default_kernel(const uint64_t* __restrict__ rows, size_t len, int D, float dv,
                       float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < len * (size_t)D;
       i += (size_t)gridDim.x * kBlock)
    if (rows[i / (size_t)D] == kInvalidIndex) out[i] = dv;
}

}  // namespace
}  // namespace hctr

extern "C" {

int hctr_uvm_create(size_t device_table_capacity, size_t host_table_capacity, size_t max_batch_size,
                    int vec_size, float default_value, int key_type, hctr_uvm** out) {
  HCTR_REQUIRE(out && device_table_capacity > 0 && host_table_capacity > 0 && max_batch_size > 0 &&
                   vec_size > 0,
               "arguments");
  HCTR_REQUIRE(key_type == HCTR_KEY_U32 || key_type == HCTR_KEY_I64, "key_type");
  hctr_uvm* u = new hctr_uvm();
  u->capacity = host_table_capacity;
  u->max_batch = max_batch_size;
  u->D = vec_size;
  u->key_type = key_type;
  u->default_value = default_value;
  int rc = hctr_tiered_create(host_table_capacity, vec_size,
                              ceil_div<size_t>(device_table_capacity, (size_t)kSetSlots), &u->tier);
  if (rc == HCTR_OK) rc = u->index.create(host_table_capacity, key_type);
  if (rc == HCTR_OK) rc = u->index.reserve(max_batch_size);
  const size_t kb = key_type == HCTR_KEY_U32 ? 4 : 8;
  if (rc == HCTR_OK &&
      (hipMalloc(&u->d_rows, max_batch_size * sizeof(uint64_t)) != hipSuccess ||
       hipMalloc(&u->d_keys, max_batch_size * kb) != hipSuccess ||
       hipMalloc(&u->d_vecs, max_batch_size * (size_t)vec_size * sizeof(float)) != hipSuccess)) {
    (void)hipGetLastError();
    set_error("hctr_uvm_create: device allocation failed");
    rc = HCTR_ERR_HIP;
  }
  if (rc != HCTR_OK) {
    hctr_uvm_destroy(u);
    return rc;
  }
  *out = u;
  return HCTR_OK;
}

int hctr_uvm_destroy(hctr_uvm* u) {
  if (!u) return HCTR_OK;
  (void)hipDeviceSynchronize();
  hctr_tiered_destroy(u->tier);
  u->index.destroy();
  for (void* q : {(void*)u->d_rows, u->d_keys, (void*)u->d_vecs})
    if (q) (void)hipFree(q);
  delete u;
  return HCTR_OK;
}

hctr_tiered* hctr_uvm_tier(hctr_uvm* u) { return u ? u->tier : nullptr; }

static int uvm_check(hctr_uvm* u, hipStream_t s) {
  uint32_t f = 0;
  HCTR_TRY(u->index.error_flags(s, &f));
  if (f & 4u) {
    set_error("tiered table: the index's cooperative launch did not get all its workgroups onto the device");
    return HCTR_ERR_HIP;
  }
  if (f != 0u) {
    set_error("tiered table: more distinct keys than host_table_capacity rows");
    return HCTR_ERR_OVERFLOW;
  }
  return HCTR_OK;
}

int hctr_uvm_add(hctr_uvm* u, const void* h_keys, const float* h_vectors, size_t len) {
  HCTR_REQUIRE(u, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(h_keys && h_vectors, "null pointer");
  const size_t kb = u->key_type == HCTR_KEY_U32 ? 4 : 8;
  const size_t D = (size_t)u->D;
  // a key listed twice in one call: its LAST vector stays (UvmTable::add walks the list in order,
  // uvm_table.cu:389-406) -- resolved here, on the host, so that the rows of one scatter are unique
  std::unordered_map<long long, size_t> last;
  last.reserve(len * 2);
  for (size_t i = 0; i < len; i++) {
    const long long k = u->key_type == HCTR_KEY_U32 ? (long long)((const uint32_t*)h_keys)[i]
                                                    : ((const long long*)h_keys)[i];
    last[k] = i;
  }
  std::vector<size_t> pick;  // positions kept, in position order (rows are handed out in this order)
  pick.reserve(last.size());
  for (size_t i = 0; i < len; i++) {
    const long long k = u->key_type == HCTR_KEY_U32 ? (long long)((const uint32_t*)h_keys)[i]
                                                    : ((const long long*)h_keys)[i];
    if (last[k] == i) pick.push_back(i);
  }
  std::vector<unsigned char> kbuf(u->max_batch * kb);
  std::vector<float> vbuf(u->max_batch * D);
  for (size_t b = 0; b < pick.size(); b += u->max_batch) {
    const size_t n = pick.size() - b < u->max_batch ? pick.size() - b : u->max_batch;
    for (size_t j = 0; j < n; j++) {
      memcpy(kbuf.data() + j * kb, (const unsigned char*)h_keys + pick[b + j] * kb, kb);
      memcpy(vbuf.data() + j * D, h_vectors + pick[b + j] * D, D * sizeof(float));
    }
    HCTR_HIP(hipMemcpy(u->d_keys, kbuf.data(), n * kb, hipMemcpyHostToDevice));
    HCTR_HIP(hipMemcpy(u->d_vecs, vbuf.data(), n * D * sizeof(float), hipMemcpyHostToDevice));
    HCTR_TRY(u->index.get_insert(u->d_keys, n, nullptr, u->d_rows, nullptr));
    HCTR_TRY(uvm_check(u, nullptr));
    HCTR_TRY(hctr_tiered_scatter(u->tier, (const int64_t*)u->d_rows, n, u->d_vecs, 0, 1.0f, nullptr));
    HCTR_HIP(hipStreamSynchronize(nullptr));
  }
  return HCTR_OK;
}

// keys -> rows of the host store, in pieces of max_batch; then the tiered lookup of the rows
static int uvm_rows_lookup(hctr_uvm* u, const void* d_keys, size_t len, float* d_vectors, int insert,
                           size_t* d_missing_len, hipStream_t s) {
  const size_t kb = u->key_type == HCTR_KEY_U32 ? 4 : 8;
  for (size_t b = 0; b < len; b += u->max_batch) {
    const size_t n = len - b < u->max_batch ? len - b : u->max_batch;
    const void* kp = (const char*)d_keys + b * kb;
    float* op = d_vectors + b * (size_t)u->D;
    if (insert) HCTR_TRY(u->index.get_insert(kp, n, nullptr, u->d_rows, s));
    else HCTR_TRY(u->index.get_mark(kp, n, nullptr, u->d_rows, s));
    // (a key without a row reads as row SIZE_MAX = -1: outside the store, zeros)
    HCTR_TRY(hctr_tiered_lookup(u->tier, (const int64_t*)u->d_rows, n, op, d_missing_len, s));
    if (u->default_value != 0.f) {
      hipLaunchKernelGGL(uvm_default_kernel, dim3(grid_for(n * (size_t)u->D, kBlock, 4096)),
                         dim3(kBlock), 0, s, u->d_rows, n, u->D, u->default_value, op);
      HCTR_LAUNCH_CHECK();
    }
  }
  return HCTR_OK;
}

int hctr_uvm_query(hctr_uvm* u, const void* d_keys, size_t len, float* d_vectors,
                   hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(d_keys && d_vectors, "null pointer");
  return uvm_rows_lookup(u, d_keys, len, d_vectors, 0, nullptr, as_stream(stream));
}

int hctr_uvm_lookup(hctr_uvm* u, const void* d_keys, size_t len, float* d_vectors,
                    uint64_t* d_row_index, size_t* d_missing_len, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  if (len == 0) return HCTR_OK;
  HCTR_REQUIRE(d_keys && d_vectors, "null pointer");
  HCTR_REQUIRE(d_row_index == nullptr || len <= u->max_batch,
               "row_index is handed back for calls of at most max_batch_size keys");
  hipStream_t s = as_stream(stream);
  HCTR_TRY(uvm_rows_lookup(u, d_keys, len, d_vectors, 1, d_missing_len, s));
  if (d_row_index)
    HCTR_HIP(hipMemcpyAsync(d_row_index, u->d_rows, len * sizeof(uint64_t), hipMemcpyDeviceToDevice, s));
  return HCTR_OK;
}

int hctr_uvm_scatter_rows(hctr_uvm* u, const uint64_t* d_unique_rows, size_t len, const float* values,
                          int add, float alpha, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  return hctr_tiered_scatter(u->tier, (const int64_t*)d_unique_rows, len, values, add, alpha, stream);
}

int hctr_uvm_check_overflow(hctr_uvm* u, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  return uvm_check(u, as_stream(stream));
}

int hctr_uvm_size(hctr_uvm* u, hctr_stream_t stream, size_t* out) {
  HCTR_REQUIRE(u && out, "null pointer");
  return u->index.value_head(as_stream(stream), out);
}

int hctr_uvm_clear(hctr_uvm* u, hctr_stream_t stream) {
  HCTR_REQUIRE(u, "null handle");
  hipStream_t s = as_stream(stream);
  HCTR_TRY(u->index.clear(s));
  hctr_cache* c = u->tier->cache;
  const size_t slots = c->num_sets * (size_t)kSetSlots;
  hipLaunchKernelGGL(cache_init_kernel, dim3(grid_for(slots, kBlock, 4096)), dim3(kBlock), 0, s, slots,
                     c->empty_key, c->set_keys, c->counters);
  HCTR_LAUNCH_CHECK();
  HCTR_HIP(hipMemsetAsync(u->tier->dirty, 0, slots * sizeof(uint32_t), s));
  return HCTR_OK;
}

}  // extern "C"

