// hashtable.hip -- deterministic open-addressing hash map on gfx950.
//
// Semantics follow R/HugeCTR/src/hashtable/nv_hashtable.cu:169-303 +
// R/HugeCTR/include/hashtable/cudf/concurrent_unordered_map.cuh:562-655:
//   slots = (size_t)(capacity / 0.75f), slot = MurmurHash3_32(key) % slots, linear probing,
//   empty key = max(Key), unused value = SIZE_MAX, get_mark miss -> SIZE_MAX.
// Difference by design: the reference hands out row indices with a racing
// atomicAdd; here a new key's index is counter + (rank of its FIRST occurrence among the new
// keys of the batch), i.e. exactly what a sequential insert in array order produces.
//
// get_insert is 5 launches; in steady state (no unseen key) launches 2-5 exit on one scalar load:
//   A probe_insert : find/claim slot (CAS on key); known key -> index; unseen -> atomicMin of
//                    (PENDING | position) into the slot value, out = PENDING | slot
//   B flag_count   : flag positions that are the first occurrence of an unseen key, per-tile count
//   S scan_tiles   : single-workgroup exclusive scan of tile counts, counter bump
//   D1 assign      : first occurrences get counter_base + rank, written to slot + out
//   D2 resolve     : remaining occurrences read the now-final slot value
#include "hashtable.h"

#include "block_prims.h"

namespace hctr {

namespace {

constexpr int kBlock = 256;

template <typename K>
__device__ __forceinline__ long long widen(K k) {
  return (long long)k;
}
template <>
__device__ __forceinline__ long long widen<uint32_t>(uint32_t k) {
  return (long long)(unsigned long long)k;
}

__device__ __forceinline__ size_t live_count(const uint64_t* d_n, size_t n) {
  if (d_n == nullptr) return n;
  uint64_t v = *d_n;
  return v < n ? (size_t)v : n;
}

__global__ void ht_init_kernel(HtEntry* e, uint64_t size, long long empty) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size;
       i += (uint64_t)gridDim.x * blockDim.x) {
    e[i].key = empty;
    e[i].val = kInvalidIndex;
  }
}

// one key, the full protocol: probe (linear), claim an empty slot for an unseen key, hand back the
// row or the pending marker (see the header of this file)
template <typename K>
__device__ __forceinline__ void ht_probe_insert_one(HtEntry* __restrict__ tab, uint64_t size,
                                                    K key, size_t i, uint64_t* __restrict__ out,
                                                    uint32_t* d_pending, uint32_t* d_error) {
  const long long empty = KeyTraits<K>::empty;
  const long long k64 = widen<K>(key);
  uint64_t slot = (uint64_t)murmur3_key(key) % size;
  bool ok = false;
  for (uint64_t probes = 0; probes < size; ++probes) {
    long long cur = tab[slot].key;
    if (cur == k64) {
      ok = true;
      break;
    }
    if (cur == empty) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot].key),
                                         (unsigned long long)empty, (unsigned long long)k64);
      if (old == (unsigned long long)empty || old == (unsigned long long)k64) {
        ok = true;
        break;
      }
    }
    slot = (slot + 1 == size) ? 0 : slot + 1;
  }
  if (!ok) {
    atomicOr(d_error, 1u);
    out[i] = kInvalidIndex;
    return;
  }
  unsigned long long v = tab[slot].val;
  if (v < kPendingBit) {
    out[i] = v;
  } else {
    atomicMin(&tab[slot].val, (unsigned long long)(kPendingBit | (uint64_t)i));
    out[i] = kPendingBit | slot;
    *d_pending = 1u;  // benign race: all writers store 1
  }
}

// Steady state = every key is in the table and sits in its home slot or close to it, so the
// kernel is one dependent 16-byte read per key: a thread takes kHtUnroll keys at a time and has
// their home entries in flight together (key and row in ONE 16-byte load); only a key whose home
// entry is not its own (collision, or unseen) walks the full protocol above.
constexpr int kHtUnroll = 4;

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ht_probe_insert_kernel(HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys,
                           size_t n, const uint64_t* d_n, uint64_t* __restrict__ out,
                           uint32_t* d_pending, uint32_t* d_error) {
  const size_t nl = live_count(d_n, n);
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i0 = blockIdx.x * (size_t)kBlock + threadIdx.x; i0 < nl;
       i0 += nthreads * kHtUnroll) {
    K key[kHtUnroll];
    uint64_t slot[kHtUnroll];
    ulonglong2 ent[kHtUnroll];
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++) {
      const size_t i = i0 + (size_t)u * nthreads;
      key[u] = keys[i < nl ? i : i0];
      slot[u] = (uint64_t)murmur3_key(key[u]) % size;
    }
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++)
      ent[u] = *reinterpret_cast<const ulonglong2*>(&tab[slot[u]]);
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++) {
      const size_t i = i0 + (size_t)u * nthreads;
      if (i >= nl) continue;
      if ((long long)ent[u].x == widen<K>(key[u]) && ent[u].y < kPendingBit)
        out[i] = ent[u].y;
      else
        ht_probe_insert_one<K>(tab, size, key[u], i, out, d_pending, d_error);
    }
  }
}

__device__ __forceinline__ bool is_first_occurrence(const HtEntry* tab, uint64_t o, size_t i) {
  if (o < kPendingBit || o == kInvalidIndex) return false;
  return tab[o & ~kPendingBit].val == (kPendingBit | (uint64_t)i);
}

__global__ void __launch_bounds__(kBlock)
    ht_flag_count_kernel(const HtEntry* __restrict__ tab, const uint64_t* __restrict__ out,
                         size_t n, const uint64_t* d_n, const uint32_t* d_pending,
                         uint32_t* __restrict__ tile_sums, size_t n_tiles) {
  if (*d_pending == 0u) return;
  __shared__ uint32_t smem[kBlock / 64 + 1];
  const size_t nl = live_count(d_n, n);
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < kHtTile / kBlock; r++) {
      size_t i = tile * kHtTile + r * kBlock + threadIdx.x;
      if (i < nl) c += is_first_occurrence(tab, out[i], i) ? 1u : 0u;
    }
    uint32_t tot = block_reduce_sum<uint32_t, kBlock>(c, smem);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}

// single workgroup: exclusive scan of sums[0..m) in place, total -> *d_total (uint64)
__global__ void __launch_bounds__(1024)
    scan_tiles_kernel(uint32_t* sums, size_t m, const uint32_t* d_gate, uint64_t* d_total) {
  if (d_gate != nullptr && *d_gate == 0u) {
    if (threadIdx.x == 0) *d_total = 0;
    return;
  }
  __shared__ uint32_t smem[1024 / 64 + 1];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < m; base += 1024) {
    size_t i = base + threadIdx.x;
    uint32_t v = (i < m) ? sums[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<uint32_t, 1024>(v, smem, &tot);
    uint64_t c = carry;
    // tile counts are bounded by n <= 2^32 positions per batch in practice; keep 32-bit offsets
    if (i < m) sums[i] = (uint32_t)(c + ex);
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *d_total = carry;
}

// get_insert step S: scan the per-tile counts of new keys, hand out the index range
// [counter, counter + new) and latch/reset the "batch has unseen keys" flag so that the next
// get_insert needs no memset.
__global__ void __launch_bounds__(1024)
    ht_scan_bump_kernel(uint32_t* sums, size_t m, uint32_t* d_pending, uint32_t* d_latched,
                        uint64_t* d_counter, uint64_t* d_base, uint64_t* d_new_count,
                        uint64_t capacity, uint32_t* d_error) {
  __shared__ uint32_t smem[1024 / 64 + 1];
  __shared__ uint64_t carry;
  const uint32_t pending = *d_pending;
  if (pending == 0u) {
    if (threadIdx.x == 0) {
      *d_latched = 0u;
      *d_new_count = 0;
      *d_base = *d_counter;
    }
    return;
  }
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < m; base += 1024) {
    size_t i = base + threadIdx.x;
    uint32_t v = (i < m) ? sums[i] : 0u;
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<uint32_t, 1024>(v, smem, &tot);
    uint64_t c = carry;
    if (i < m) sums[i] = (uint32_t)(c + ex);
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const uint64_t c0 = *d_counter;
    *d_base = c0;
    *d_new_count = carry;
    // more unseen keys than free rows: the counter stops at the capacity, ht_assign_kernel gives
    // the keys beyond it no row (kInvalidIndex -> they pool as zeros and are skipped by the
    // update, like an eval miss), and error bit 1 makes check_overflow() fail as the reference's
    // does (localized_slot_sparse_embedding_hash.hpp:552-569) -- nothing is read or written
    // outside the [capacity] row arrays
    *d_counter = (c0 + carry > capacity) ? capacity : c0 + carry;
    if (c0 + carry > capacity) atomicOr(d_error, 2u);
    *d_latched = 1u;
    *d_pending = 0u;
  }
}

// optional: record the slot id of every newly inserted row (store_slot_id_kernel semantics,
// R/HugeCTR/src/embeddings/store_slot_id_functor.cu:26-48, restricted to rows that are new)
struct SlotIdSink {
  uint64_t* slot_id;       // nullptr = off
  const void* row_offset;  // CSR of the batch being resolved (key-typed)
  int key_is_u32;
  size_t buckets;
  int buckets_per_sample, rank, world, localized;
};

__device__ __forceinline__ void record_slot_id(const SlotIdSink& k, uint64_t pos, uint64_t row) {
  size_t lo = 0, hi = k.buckets;
  while (lo < hi) {  // bucket u with ro[u] <= pos < ro[u+1]
    const size_t mid = (lo + hi) >> 1;
    const uint64_t e = k.key_is_u32 ? (uint64_t)((const uint32_t*)k.row_offset)[mid + 1]
                                    : (uint64_t)((const long long*)k.row_offset)[mid + 1];
    if (e <= pos) lo = mid + 1;
    else hi = mid;
  }
  const int j = (int)(lo % (size_t)k.buckets_per_sample);
  k.slot_id[row] = (uint64_t)(k.localized ? k.rank + j * k.world : j);
}

__global__ void __launch_bounds__(kBlock)
    ht_assign_kernel(HtEntry* __restrict__ tab, uint64_t* __restrict__ out, size_t n,
                     const uint64_t* d_n, const uint32_t* d_pending,
                     const uint32_t* __restrict__ tile_sums, size_t n_tiles,
                     const uint64_t* d_base, uint64_t* __restrict__ new_positions,
                     SlotIdSink sink, uint64_t capacity) {
  if (*d_pending == 0u) return;
  __shared__ uint32_t smem[kBlock / 64 + 1];
  const size_t nl = live_count(d_n, n);
  const uint64_t base = *d_base;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t run = tile_sums[tile];
#pragma unroll
    for (int r = 0; r < kHtTile / kBlock; r++) {
      size_t i = tile * kHtTile + r * kBlock + threadIdx.x;
      uint64_t o = (i < nl) ? out[i] : 0;
      bool f = (i < nl) && is_first_occurrence(tab, o, i);
      uint32_t tot;
      uint32_t ex = block_exclusive_scan<uint32_t, kBlock>(f ? 1u : 0u, smem, &tot);
      if (f) {
        uint64_t rank = (uint64_t)run + ex;
        uint64_t fin = base + rank;
        if (fin >= capacity) fin = kInvalidIndex;  // table full: the key gets no row
        tab[o & ~kPendingBit].val = fin;
        out[i] = fin;
        new_positions[rank] = (uint64_t)i;
        if (sink.slot_id != nullptr && fin != kInvalidIndex)
          record_slot_id(sink, (uint64_t)i, fin);
      }
      run += tot;
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    ht_resolve_kernel(const HtEntry* __restrict__ tab, uint64_t* __restrict__ out, size_t n,
                      const uint64_t* d_n, const uint32_t* d_pending) {
  if (*d_pending == 0u) return;
  const size_t nl = live_count(d_n, n);
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < nl;
       i += (size_t)gridDim.x * kBlock) {
    uint64_t o = out[i];
    if (o >= kPendingBit && o != kInvalidIndex) out[i] = tab[o & ~kPendingBit].val;
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ht_find_kernel(const HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys,
                   size_t n, const uint64_t* d_n, uint64_t* __restrict__ out) {
  const size_t nl = live_count(d_n, n);
  const long long empty = KeyTraits<K>::empty;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i0 = blockIdx.x * (size_t)kBlock + threadIdx.x; i0 < nl;
       i0 += nthreads * kHtUnroll) {
    K key[kHtUnroll];
    uint64_t slot[kHtUnroll];
    ulonglong2 ent[kHtUnroll];
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++) {
      const size_t i = i0 + (size_t)u * nthreads;
      key[u] = keys[i < nl ? i : i0];
      slot[u] = (uint64_t)murmur3_key(key[u]) % size;
    }
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++)  // home entries of kHtUnroll keys in flight together
      ent[u] = *reinterpret_cast<const ulonglong2*>(&tab[slot[u]]);
#pragma unroll
    for (int u = 0; u < kHtUnroll; u++) {
      const size_t i = i0 + (size_t)u * nthreads;
      if (i >= nl) continue;
      const long long k64 = widen<K>(key[u]);
      uint64_t res = kInvalidIndex;
      if ((long long)ent[u].x == k64) {
        res = ent[u].y;
      } else if ((long long)ent[u].x != empty) {  // collision: walk on
        uint64_t sl = (slot[u] + 1 == size) ? 0 : slot[u] + 1;
        for (uint64_t probes = 1; probes <= size; ++probes) {
          const long long cur = tab[sl].key;
          if (cur == k64) {
            res = tab[sl].val;
            break;
          }
          if (cur == empty) break;
          sl = (sl + 1 == size) ? 0 : sl + 1;
        }
      }
      out[i] = res;
    }
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    ht_insert_pairs_kernel(HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys,
                           const uint64_t* __restrict__ vals, size_t n, uint32_t* d_error) {
  const long long empty = KeyTraits<K>::empty;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const K key = keys[i];
    const long long k64 = widen<K>(key);
    uint64_t slot = (uint64_t)murmur3_key(key) % size;
    bool ok = false;
    for (uint64_t probes = 0; probes < size; ++probes) {
      long long cur = tab[slot].key;
      if (cur == k64) {
        ok = true;
        break;
      }
      if (cur == empty) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot].key),
                                           (unsigned long long)empty, (unsigned long long)k64);
        if (old == (unsigned long long)empty || old == (unsigned long long)k64) {
          ok = true;
          break;
        }
      }
      slot = (slot + 1 == size) ? 0 : slot + 1;
    }
    if (ok) tab[slot].val = vals[i];
    else atomicOr(d_error, 1u);
  }
}

// occupied-slot compaction (size_kernel / dump_kernel, nv_hashtable.cu:116-163), physical order
__global__ void __launch_bounds__(kBlock)
    ht_occupied_count_kernel(const HtEntry* __restrict__ tab, uint64_t size, long long empty,
                             uint32_t* __restrict__ tile_sums, size_t n_tiles) {
  __shared__ uint32_t smem[kBlock / 64 + 1];
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < kHtTile / kBlock; r++) {
      uint64_t i = tile * (uint64_t)kHtTile + r * kBlock + threadIdx.x;
      // erased entries (dynamic tables) keep a tombstone key with an invalid value
      if (i < size) c += (tab[i].key != empty && tab[i].val != kInvalidIndex) ? 1u : 0u;
    }
    uint32_t tot = block_reduce_sum<uint32_t, kBlock>(c, smem);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}

__global__ void __launch_bounds__(kBlock)
    ht_dump_kernel(const HtEntry* __restrict__ tab, uint64_t size, long long empty,
                   const uint32_t* __restrict__ tile_sums, size_t n_tiles,
                   int64_t* __restrict__ keys, uint64_t* __restrict__ vals) {
  __shared__ uint32_t smem[kBlock / 64 + 1];
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t run = tile_sums[tile];
#pragma unroll
    for (int r = 0; r < kHtTile / kBlock; r++) {
      uint64_t i = tile * (uint64_t)kHtTile + r * kBlock + threadIdx.x;
      HtEntry e;
      bool f = false;
      if (i < size) {
        e = tab[i];
        f = e.key != empty && e.val != kInvalidIndex;
      }
      uint32_t tot;
      uint32_t ex = block_exclusive_scan<uint32_t, kBlock>(f ? 1u : 0u, smem, &tot);
      if (f) {
        keys[run + ex] = e.key;
        vals[run + ex] = e.val;
      }
      run += tot;
    }
  }
}

template <typename K>
__global__ void hash_keys_kernel(const K* keys, size_t n, uint32_t* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = murmur3_key(keys[i]);
}

}  // namespace

int HashTable::create(size_t cap, int kt) {
  HCTR_REQUIRE(kt == HCTR_KEY_U32 || kt == HCTR_KEY_I64, "key_type");
  HCTR_REQUIRE(cap > 0, "capacity must be > 0");
  capacity = cap;
  key_type = kt;
  // static_cast<size_t>(capacity / LOAD_FACTOR), LOAD_FACTOR = 0.75f (float division)
  size = (uint64_t)((float)cap / 0.75f);
  if (size == 0) size = 1;
  HCTR_HIP(hipMalloc(&entries, size * sizeof(HtEntry)));
  uint64_t* scal = nullptr;
  HCTR_HIP(hipMalloc(&scal, 64));
  d_counter = scal;
  d_base = scal + 1;
  d_new_count = scal + 2;
  d_scratch64 = scal + 3;
  d_pending = reinterpret_cast<uint32_t*>(scal + 4);
  d_error = reinterpret_cast<uint32_t*>(scal + 4) + 1;
  d_latched = reinterpret_cast<uint32_t*>(scal + 5);
  HCTR_HIP(hipMemset(scal, 0, 64));
  return clear(nullptr);
}

int HashTable::destroy() {
  if (entries) (void)hipFree(entries);
  if (d_counter) (void)hipFree(d_counter);
  if (tile_sums) (void)hipFree(tile_sums);
  if (new_positions) (void)hipFree(new_positions);
  entries = nullptr;
  d_counter = nullptr;
  tile_sums = nullptr;
  new_positions = nullptr;
  return HCTR_OK;
}

int HashTable::clear(hipStream_t s) {
  const long long empty =
      key_type == HCTR_KEY_U32 ? KeyTraits<uint32_t>::empty : KeyTraits<long long>::empty;
  hipLaunchKernelGGL(ht_init_kernel, dim3(grid_for(size, 256, 8192)), dim3(256), 0, s, entries,
                     size, empty);
  HCTR_LAUNCH_CHECK();
  HCTR_HIP(hipMemsetAsync(d_counter, 0, 64, s));
  return HCTR_OK;
}

int HashTable::reserve(size_t n) {
  // scratch must also cover a dump over `size` slots
  size_t need_tiles = ceil_div<size_t>(n > size ? n : size, kHtTile) + 1;
  if (n <= max_n && tile_sums != nullptr) return HCTR_OK;
  if (tile_sums) (void)hipFree(tile_sums);
  if (new_positions) (void)hipFree(new_positions);
  HCTR_HIP(hipMalloc(&tile_sums, need_tiles * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&new_positions, (n > 0 ? n : 1) * sizeof(uint64_t)));
  max_n = n;
  return HCTR_OK;
}

int HashTable::get_insert(const void* keys, size_t n, const uint64_t* d_n, uint64_t* out,
                          hipStream_t s, const SlotSink* sink_in) {
  if (n == 0) return HCTR_OK;
  HCTR_TRY(reserve(n));
  SlotIdSink sink;
  sink.slot_id = nullptr;
  sink.row_offset = nullptr;
  sink.key_is_u32 = key_type == HCTR_KEY_U32;
  sink.buckets = 0;
  sink.buckets_per_sample = 1;
  sink.rank = 0;
  sink.world = 1;
  sink.localized = 0;
  if (sink_in != nullptr) {
    sink.slot_id = sink_in->slot_id;
    sink.row_offset = sink_in->row_offset;
    sink.buckets = sink_in->buckets;
    sink.buckets_per_sample = sink_in->buckets_per_sample;
    sink.rank = sink_in->rank;
    sink.world = sink_in->world;
    sink.localized = sink_in->localized;
  }
  const int grid = grid_for(ceil_div<size_t>(n, kHtUnroll), kBlock, 1 << 16);
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ht_probe_insert_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const uint32_t*)keys, n, d_n, out, d_pending, d_error);
  } else {
    hipLaunchKernelGGL(ht_probe_insert_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const long long*)keys, n, d_n, out, d_pending, d_error);
  }
  HCTR_LAUNCH_CHECK();
  // steps B..D2 exit on one scalar load when the batch holds no unseen key (steady state);
  // small grids keep those empty launches cheap, grid-stride covers the cold-start case.
  const size_t n_tiles = ceil_div<size_t>(n, kHtTile);
  const int tgrid = (int)(n_tiles < (size_t)512 ? n_tiles : (size_t)512);
  hipLaunchKernelGGL(ht_flag_count_kernel, dim3(tgrid), dim3(kBlock), 0, s, entries, out, n, d_n,
                     d_pending, tile_sums, n_tiles);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ht_scan_bump_kernel, dim3(1), dim3(1024), 0, s, tile_sums, n_tiles, d_pending,
                     d_latched, d_counter, d_base, d_new_count, capacity, d_error);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ht_assign_kernel, dim3(tgrid), dim3(kBlock), 0, s, entries, out, n, d_n,
                     d_latched, tile_sums, n_tiles, d_base, new_positions, sink, capacity);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ht_resolve_kernel, dim3(grid_for(n, kBlock, 512)), dim3(kBlock), 0, s,
                     entries, out, n, d_n, d_latched);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int HashTable::get_mark(const void* keys, size_t n, const uint64_t* d_n, uint64_t* out,
                        hipStream_t s) {
  if (n == 0) return HCTR_OK;
  const int grid = grid_for(n, kBlock);
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ht_find_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, entries, size,
                       (const uint32_t*)keys, n, d_n, out);
  } else {
    hipLaunchKernelGGL(ht_find_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, entries, size,
                       (const long long*)keys, n, d_n, out);
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int HashTable::insert(const void* keys, const uint64_t* vals, size_t n, hipStream_t s) {
  if (n == 0) return HCTR_OK;
  const int grid = grid_for(n, kBlock);
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ht_insert_pairs_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const uint32_t*)keys, vals, n, d_error);
  } else {
    hipLaunchKernelGGL(ht_insert_pairs_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const long long*)keys, vals, n, d_error);
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int HashTable::count(hipStream_t s, size_t* out) {
  HCTR_TRY(reserve(max_n));
  const long long empty =
      key_type == HCTR_KEY_U32 ? KeyTraits<uint32_t>::empty : KeyTraits<long long>::empty;
  const size_t n_tiles = ceil_div<size_t>(size, kHtTile);
  const int tgrid = (int)(n_tiles < (size_t)kMaxGrid ? n_tiles : (size_t)kMaxGrid);
  hipLaunchKernelGGL(ht_occupied_count_kernel, dim3(tgrid), dim3(kBlock), 0, s, entries, size,
                     empty, tile_sums, n_tiles);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, s, tile_sums, n_tiles,
                     (const uint32_t*)nullptr, d_scratch64);
  HCTR_LAUNCH_CHECK();
  uint64_t h = 0;
  HCTR_HIP(hipMemcpyAsync(&h, d_scratch64, sizeof(h), hipMemcpyDeviceToHost, s));
  HCTR_HIP(hipStreamSynchronize(s));
  *out = (size_t)h;
  return HCTR_OK;
}

int HashTable::value_head(hipStream_t s, size_t* out) {
  uint64_t h = 0;
  HCTR_HIP(hipMemcpyAsync(&h, d_counter, sizeof(h), hipMemcpyDeviceToHost, s));
  HCTR_HIP(hipStreamSynchronize(s));
  *out = (size_t)h;
  return HCTR_OK;
}

int HashTable::set_value_head(size_t v, hipStream_t s) {
  uint64_t h = v;
  HCTR_HIP(hipMemcpyAsync(d_counter, &h, sizeof(h), hipMemcpyHostToDevice, s));
  HCTR_HIP(hipStreamSynchronize(s));
  return HCTR_OK;
}

int HashTable::dump(int64_t* d_keys, uint64_t* d_vals, size_t* cnt, hipStream_t s) {
  size_t c = 0;
  HCTR_TRY(count(s, &c));  // leaves exclusive tile offsets in tile_sums
  const long long empty =
      key_type == HCTR_KEY_U32 ? KeyTraits<uint32_t>::empty : KeyTraits<long long>::empty;
  const size_t n_tiles = ceil_div<size_t>(size, kHtTile);
  const int tgrid = (int)(n_tiles < (size_t)kMaxGrid ? n_tiles : (size_t)kMaxGrid);
  hipLaunchKernelGGL(ht_dump_kernel, dim3(tgrid), dim3(kBlock), 0, s, entries, size, empty,
                     tile_sums, n_tiles, d_keys, d_vals);
  HCTR_LAUNCH_CHECK();
  HCTR_HIP(hipStreamSynchronize(s));
  *cnt = c;
  return HCTR_OK;
}

int HashTable::error_flags(hipStream_t s, uint32_t* out) {
  HCTR_HIP(hipMemcpyAsync(out, d_error, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  HCTR_HIP(hipStreamSynchronize(s));
  return HCTR_OK;
}

}  // namespace hctr

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
using namespace hctr;

extern "C" {

int hctr_hash_keys(const void* keys, int key_type, size_t n, uint32_t* out, hctr_stream_t stream) {
  if (n == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && out, "null pointer");
  hipStream_t s = as_stream(stream);
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(hash_keys_kernel<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s,
                       (const uint32_t*)keys, n, out);
  } else if (key_type == HCTR_KEY_I64) {
    hipLaunchKernelGGL(hash_keys_kernel<long long>, dim3(grid_for(n, 256)), dim3(256), 0, s,
                       (const long long*)keys, n, out);
  } else {
    HCTR_REQUIRE(false, "key_type");
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_ht_create(size_t capacity, int key_type, hctr_hashtable** out) {
  HCTR_REQUIRE(out, "out is null");
  hctr_hashtable* h = new hctr_hashtable();
  int rc = h->impl.create(capacity, key_type);
  if (rc != HCTR_OK) {
    h->impl.destroy();
    delete h;
    return rc;
  }
  *out = h;
  return HCTR_OK;
}

int hctr_ht_destroy(hctr_hashtable* ht) {
  if (!ht) return HCTR_OK;
  ht->impl.destroy();
  delete ht;
  return HCTR_OK;
}

int hctr_ht_clear(hctr_hashtable* ht, hctr_stream_t stream) {
  HCTR_REQUIRE(ht, "null handle");
  return ht->impl.clear(as_stream(stream));
}

int hctr_ht_get_insert(hctr_hashtable* ht, const void* keys, size_t n, const uint64_t* d_n,
                       uint64_t* value_index, hctr_stream_t stream) {
  HCTR_REQUIRE(ht && (n == 0 || (keys && value_index)), "null pointer");
  return ht->impl.get_insert(keys, n, d_n, value_index, as_stream(stream), nullptr);
}

int hctr_ht_get_mark(hctr_hashtable* ht, const void* keys, size_t n, const uint64_t* d_n,
                     uint64_t* value_index, hctr_stream_t stream) {
  HCTR_REQUIRE(ht && (n == 0 || (keys && value_index)), "null pointer");
  return ht->impl.get_mark(keys, n, d_n, value_index, as_stream(stream));
}

int hctr_ht_insert(hctr_hashtable* ht, const void* keys, const uint64_t* vals, size_t n,
                   hctr_stream_t stream) {
  HCTR_REQUIRE(ht && (n == 0 || (keys && vals)), "null pointer");
  return ht->impl.insert(keys, vals, n, as_stream(stream));
}

int hctr_ht_size(hctr_hashtable* ht, hctr_stream_t stream, size_t* out) {
  HCTR_REQUIRE(ht && out, "null pointer");
  return ht->impl.count(as_stream(stream), out);
}

int hctr_ht_value_head(hctr_hashtable* ht, hctr_stream_t stream, size_t* out) {
  HCTR_REQUIRE(ht && out, "null pointer");
  return ht->impl.value_head(as_stream(stream), out);
}

int hctr_ht_set_value_head(hctr_hashtable* ht, size_t v, hctr_stream_t stream) {
  HCTR_REQUIRE(ht, "null handle");
  return ht->impl.set_value_head(v, as_stream(stream));
}

size_t hctr_ht_table_size(const hctr_hashtable* ht) { return ht ? (size_t)ht->impl.size : 0; }

int hctr_ht_dump(hctr_hashtable* ht, int64_t* d_keys, uint64_t* d_vals, size_t* count,
                 hctr_stream_t stream) {
  HCTR_REQUIRE(ht && d_keys && d_vals && count, "null pointer");
  return ht->impl.dump(d_keys, d_vals, count, as_stream(stream));
}

}  // extern "C"
