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
// get_insert is 2 launches (round 3; five before):
//   1 probe_insert : find/claim slot (CAS on key); known key -> index; unseen -> out = PENDING |
//                    slot and the slot value becomes PENDING | (smallest position of the key).
//                    Two forms, chosen per batch from the number of keys the PREVIOUS batch
//                    inserted (a device word: no host decision).  Few unseen keys (steady state):
//                    every occurrence lowers the value with an atomic min.  Many (a first epoch):
//                    the thread that CLAIMED the slot stores PENDING | its position (an atomic
//                    store, no read-modify-write: one memory-side atomic per unseen key instead of
//                    two), a later occurrence that SEES a pending value lowers it with an atomic
//                    min (the store it saw precedes its min in the word's modification order), and
//                    one that sees "no row yet" (the claimer's store still on its way, or an erased
//                    key, which nobody claims) defers its min to the finish kernel (list flag; the
//                    finish kernel then runs those first and takes a second grid barrier).  Its
//                    threads also make the caller's private copy of the row offsets (world == 1).
//   2 finish       : exits on one scalar load when the batch held no unseen key.  Otherwise, in
//                    kHtFinishBlocks co-resident workgroups with ONE grid barrier between them (two
//                    when the batch deferred any min: (0) those run first, then a barrier):
//                    (A) per 64 positions a mask of the FIRST occurrences of unseen keys + counts
//                    per workgroup; barrier; (S) every workgroup scans the counts, workgroup 0
//                    bumps the row counter; (D) every pending position computes its key's row =
//                    counter + (first occurrences before the key's first position) from the
//                    masks -- no second pass over the table, no dependency between workgroups.
//                    Also posts the row counter / error flags to pinned host words and presets
//                    the caller's one-hot flag of the next batch (no copy / memset launches).
#include "hashtable.h"

#include <map>
#include <mutex>

#include <cstdlib>

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

// a key that met a full table owns no row: it reads as a miss
__device__ __forceinline__ uint64_t row_of(uint64_t v) { return v == kNoRow ? kInvalidIndex : v; }

__global__ void ht_init_kernel(HtEntry* e, uint64_t size, long long empty) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < size;
       i += (uint64_t)gridDim.x * blockDim.x) {
    e[i].key = empty;
    e[i].val = kInvalidIndex;
  }
}

// one key, the full protocol: probe (linear), claim an empty slot for an unseen key, hand back the
// row or the pending marker (see the header of this file).  Returns true when the position is left
// pending (the caller appends it to the batch's list: the finish kernel works on that list only).
// list entry of a pending position: bit 31 = its atomic min on the slot value is still to be done
constexpr uint32_t kListDefer = 0x80000000u;

template <typename T>
__device__ __forceinline__ T ld_agent(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Returns 0: resolved (out[i] written), 1: left pending, 2: left pending and its min deferred.
template <typename K>
__device__ __forceinline__ int ht_probe_insert_one(HtEntry* __restrict__ tab, uint64_t size,
                                                   K key, size_t i, uint64_t* __restrict__ out,
                                                   uint32_t* d_error, uint64_t slot, long long cur,
                                                   bool store_form) {
  // `cur` = the key of the home entry as the caller has just read it: an unseen key whose home slot
  // is empty goes straight to its claim.
  const long long empty = KeyTraits<K>::empty;
  const long long k64 = widen<K>(key);
  bool ok = false, claimed = false;
  for (uint64_t probes = 0; probes < size; ++probes) {
    if (cur == k64) {
      ok = true;
      break;
    }
    if (cur == empty) {
      unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot].key),
                                         (unsigned long long)empty, (unsigned long long)k64);
      if (old == (unsigned long long)empty || old == (unsigned long long)k64) {
        ok = true;
        claimed = old == (unsigned long long)empty;
        break;
      }
    }
    slot = (slot + 1 == size) ? 0 : slot + 1;
    cur = tab[slot].key;
  }
  if (!ok) {
    atomicOr(d_error, 1u);
    out[i] = kInvalidIndex;
    return 0;
  }
  out[i] = kPendingBit | slot;  // (overwritten below when the key turns out to own a row)
  if (store_form) {
    if (claimed) {
      // the ONE claimer of the slot: nobody lowers the value before seeing this store (below)
      st_agent(&tab[slot].val, (unsigned long long)(kPendingBit | (uint64_t)i));
      return 1;
    }
    // (an atomic load: what it returns is ordered against this thread's atomic min on the word)
    const unsigned long long v = ld_agent(&tab[slot].val);
    if (v < kPendingBit) {
      out[i] = row_of(v);
      return 0;
    }
    if (v == kInvalidIndex) return 2;  // no pending value to lower yet: the finish kernel's phase 0
  } else if (!claimed) {
    // (a slot this thread has just claimed holds no row yet: no read needed)
    const unsigned long long v = tab[slot].val;
    if (v < kPendingBit) {
      out[i] = row_of(v);
      return 0;
    }
  }
  // no-return atomic: nothing waits for it
  (void)__hip_atomic_fetch_min(&tab[slot].val, (unsigned long long)(kPendingBit | (uint64_t)i),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return 1;
}

// Steady state = every key is in the table and sits in its home slot or close to it, so the
// kernel is one dependent 16-byte read per key: a thread takes kHtUnroll keys at a time and has
// their home entries in flight together (key and row in ONE 16-byte load); only a key whose home
// entry is not its own (collision, or unseen) walks the full protocol above.
constexpr int kHtUnroll = 4;

// (8 waves per SIMD = at most 64 VGPRs: the grid of a 1.7 M-key batch is 6.5 waves per SIMD and must
//  be resident at once -- at 74 VGPRs / 6 waves the steady probe was measured 27.6 -> 33.1 us)
template <typename K>
__global__ void __launch_bounds__(kBlock, 8)
    ht_probe_insert_kernel(HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys,
                           size_t n, const uint64_t* d_n, uint64_t* __restrict__ out,
                           uint32_t* d_pending, uint32_t* __restrict__ d_list,
                           uint32_t* __restrict__ block_cnt, uint32_t seg_cap, uint32_t* d_error,
                           uint32_t* d_defer, const uint64_t* __restrict__ d_prev_new,
                           const K* __restrict__ ro_src, K* __restrict__ ro_dst, size_t n_offsets,
                           uint32_t* __restrict__ one_hot) {
  // Positions left pending go to THIS workgroup's segment of the batch's list, d_list[blockIdx *
  // seg_cap ...), counted in LDS; the count is published once at the end.  No global counter: a
  // single address takes ~10 ns per atomic whoever issues it (26 k wave-level appends to one
  // counter were measured at 250 us).
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0u;
  __syncthreads();
  const size_t nl = live_count(d_n, n);
  // more than an eighth of the previous batch's keys were unseen: the claimers store (header)
  const bool store_form = *d_prev_new * 8ull > (uint64_t)nl;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  const size_t gtid = blockIdx.x * (size_t)kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t* const my_list = d_list + (size_t)blockIdx.x * seg_cap;
  if (ro_src != nullptr) {
    // the caller's private copy of the row offsets; lengths all 1 and ro[0] == 0  <=>
    // ro[i] == i for every i: otherwise the gather must not take its offset-free one-hot loop
    bool bad = false;
    for (size_t i = gtid; i < n_offsets; i += nthreads) {
      const K v = ro_src[i];
      ro_dst[i] = v;
      bad |= v != (K)i;
    }
    if (__ballot(bad) != 0ull && lane == 0) *one_hot = 0u;
  }
  for (size_t i0 = gtid; i0 < nl; i0 += nthreads * kHtUnroll) {
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
      int pst = 0;
      if (i < nl) {
        if ((long long)ent[u].x == widen<K>(key[u]) && ent[u].y < kPendingBit)
          out[i] = row_of(ent[u].y);
        else  // (issuing the four keys' claims together was measured: no gain with 4 % unseen
              //  keys, and the steady state lost 5 us to the extra registers)
          pst = ht_probe_insert_one<K>(tab, size, key[u], i, out, d_error, slot[u],
                                        (long long)ent[u].x, store_form);
      }
      const bool pend = pst != 0;
      if (pst == 2) *d_defer = 1u;  // (benign race: every writer stores 1)
      const unsigned long long m = __ballot(pend);
      if (m != 0ull) {  // one LDS atomic per wavefront and unroll step that has any
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(&s_cnt, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (pend) {
          const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          // (seg_cap = every key of this workgroup)
          if (k < seg_cap) my_list[k] = (uint32_t)i | (pst == 2 ? kListDefer : 0u);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t c = s_cnt < seg_cap ? s_cnt : seg_cap;
    block_cnt[blockIdx.x] = c;
    if (c != 0u) *d_pending = 1u;  // (benign race: every writer stores 1)
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
  if (pos < k.buckets) {  // one key per bucket: bucket == position (two reads instead of ~20)
    const uint64_t a = k.key_is_u32 ? (uint64_t)((const uint32_t*)k.row_offset)[pos]
                                    : (uint64_t)((const long long*)k.row_offset)[pos];
    const uint64_t e = k.key_is_u32 ? (uint64_t)((const uint32_t*)k.row_offset)[pos + 1]
                                    : (uint64_t)((const long long*)k.row_offset)[pos + 1];
    if (a <= pos && pos < e) lo = hi = (size_t)pos;
  }
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

// ---- get_insert, launch 2 ------------------------------------------------------------------------
constexpr int kFinBlock = 1024;
constexpr int kFinRegions = 4096;  // (their bases live in LDS: 16 KB; 1.7 M positions: regions of
                                    //  448 = 7 mask words, which phase D loads in one batch)
constexpr int kProbeMaxBlocks = 4096;  // workgroups of the probe kernel (their list segments'
                                       // first entries live in the finish kernel's LDS: 16 KB)
constexpr uint32_t kSpinLimit = 1u << 24;  // (a barrier that never opens raises error bit 2^2)
// generation word of the grid barrier: bit 31 = a workgroup gave up waiting.  Giving up and opening
// are both compare-and-swaps on this ONE word, so a batch's barrier either opens for every
// workgroup or fails for every workgroup -- never rows for some positions and "no row" for others
constexpr uint32_t kBarAbort = 1u << 31;

struct FinishCtl {
  uint32_t *pending, *latched, *error, *barrier;
  uint32_t* barrier_odd;  // arrivals of the barriers with an odd generation (grid_barrier)
  uint32_t* defer;  // set by the probe kernel when some position left its atomic min to phase 0
  uint32_t* snap;  // two-launch form: {unseen keys?, mask parity, row counter lo, hi} of the first half
  uint64_t *counter, *base, *new_count;
  // positions whose key was not in the table: segment b of the list = [b * seg_cap, ... +
  // block_cnt[b]), written by workgroup b of the probe kernel
  const uint32_t* list;
  const uint32_t* block_cnt;
  uint32_t probe_blocks, seg_cap;
  uint32_t* region_cnt;         // 2 x [kFinRegions] first occurrences per region (as masks2)
  // two buffers of [mask_words] first-occurrence masks: buffer *parity is all zero on entry and
  // takes this batch's bits, the other one is zeroed here for the next batch that inserts
  unsigned long long* masks2;
  uint32_t* parity;
  size_t mask_words;
  uint32_t* one_hot_next;
  uint64_t *host_rows, *host_seq;
  uint64_t seq;
  uint32_t* host_error;
  uint32_t spin_limit;  // polls of the grid barrier before a workgroup gives up (HCTR_HT_SPIN_LIMIT)
};

__device__ __forceinline__ void post_to_host(const FinishCtl& c, uint64_t rows) {
  if (c.one_hot_next != nullptr) *c.one_hot_next = 1u;
  if (c.host_error != nullptr) *c.host_error = *c.error;
  if (c.host_rows != nullptr) {
    *c.host_rows = rows;
    __threadfence_system();  // rows before seq: whoever sees seq sees rows at least this new
    *c.host_seq = c.seq;
  }
}

// Data that crosses workgroups inside the finish kernel (masks, group prefixes, region counts)
// moves through agent-scope atomic loads / stores: they are coherent at the memory side, so the
// barrier itself needs no cache write-back / invalidate (an agent-scope release fence writes the
// XCD's whole L2 back -- measured ~20 us per barrier with the probe kernel's 14 MB of fresh
// stores sitting there).

// all workgroups of the grid are resident (grid <= kHtFinishBlocks, far below what 256 CUs hold):
// sense-reversing barrier on {arrived, generation}.  __syncthreads: every wave has waited for
// its (write-through) stores before thread 0 arrives.  `gen` = the generation word as thread 0 read
// it when the kernel began (it cannot move before this workgroup has arrived; a second barrier of
// the same launch passes gen + 1): no read of it sits between the phase in front and the arrival.
// Arrivals are counted on one of TWO words, picked by the generation's parity: the workgroup that
// arrives last puts its word back to zero WITHOUT waiting for that store before it opens the
// barrier -- the next barrier counts on the other word, and the one after that cannot open before
// this workgroup has arrived at it, which it does with every memory operation of its own drained.
__device__ __forceinline__ bool grid_barrier(uint32_t* bar, uint32_t* cnt_odd, uint32_t nblocks,
                                             uint32_t spin_limit, uint32_t gen) {
  // every thread waits for ITS OWN outstanding memory operations first (the no-return atomics on
  // masks / region counts of phase A included): the workgroup barrier alone orders the waves, not
  // the arrival of their atomics at the memory side -- a per-wave wait, no cache write-back
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ uint32_t ok;
  if (threadIdx.x == 0) {
    ok = 1u;
    if (gen & kBarAbort) {
      ok = 0u;  // given up before this workgroup arrived (or never cleared by the host)
    } else {
      uint32_t* const cnt = (gen & 1u) ? cnt_odd : bar;
      const uint32_t old =
          __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == nblocks - 1u) {
        st_agent(cnt, 0u);
        // opens the barrier -- unless a waiter has given up in the meantime
        uint32_t expect = gen;
        if (!__hip_atomic_compare_exchange_strong(bar + 1, &expect, (gen + 1u) & ~kBarAbort,
                                                  __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT))
          ok = 0u;
      } else {
        uint32_t spins = 0;
        for (;;) {
          uint32_t cur = ld_agent(bar + 1);
          if (cur == gen) {
            if (spins++ < spin_limit) {
              __builtin_amdgcn_s_sleep(1);
              continue;
            }
            // gives up: poisons the generation, or learns that the barrier has just opened
            uint32_t expect = gen;
            if (__hip_atomic_compare_exchange_strong(bar + 1, &expect, gen | kBarAbort,
                                                     __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
              ok = 0u;
              break;
            }
            cur = expect;
          }
          ok = (cur & kBarAbort) ? 0u : 1u;
          break;
        }
      }
    }
  }
  __syncthreads();
  return ok != 0u;
}

// PHASE 3: phase 0 alone (the deferred atomic mins), in front of PHASE 1 in the two-launch form.
// PHASE 0: the whole kernel, one grid barrier between its halves (every workgroup resident: the
// in-line index stage).  PHASE 1 / 2: the same two halves as two launches, no barrier -- for an
// index stage that runs beside other work (hctr_emb_index_ahead under the dense tower's GEMMs: a
// grid that waits for workgroups the GEMMs keep off the chip was measured at + 1 ms per step).
// The second launch takes the row counter / mask parity the first one saw from c.snap (workgroup
// 0 of the second half moves the live words).
template <int PHASE>
__global__ void __launch_bounds__(kFinBlock)
    ht_finish_kernel(HtEntry* __restrict__ tab, uint64_t* __restrict__ out, size_t n,
                     const uint64_t* d_n, FinishCtl c, uint64_t* __restrict__ new_positions,
                     SlotIdSink sink, uint64_t capacity) {
  if constexpr (PHASE == 2) {
    if (c.snap[0] == 0u) return;
  } else if constexpr (PHASE == 3) {
    if (*c.pending == 0u || *c.defer == 0u) return;
  } else {
    if (*c.pending == 0u) {  // steady state: no unseen key in this batch
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t cnt = *c.counter;
        *c.latched = 0u;
        *c.new_count = 0;
        *c.base = cnt;
        if constexpr (PHASE == 1) c.snap[0] = 0u;
        post_to_host(c, cnt);
      }
      return;
    }
  }
  // (thread 0: the barrier's generation, read here so that its latency hides behind phase A)
  uint32_t bar_gen = 0u;
  if constexpr (PHASE == 0) {
    if (threadIdx.x == 0) bar_gen = ld_agent(c.barrier + 1);
  }
  __shared__ uint32_t smem[kFinBlock / 64 + 1];
  // v2: a finish workgroup takes the list segments of ITS share of the probe workgroups (a few
  // tens of counts to scan instead of all of them, a search over those only, and no total of the
  // whole list is needed); positions are spread evenly over the probe workgroups, so the unseen
  // keys of a batch are spread over the finish workgroups as they are over the batch
  __shared__ uint32_t seg_first[kProbeMaxBlocks + 1];
  const uint32_t G = gridDim.x, b = blockIdx.x;
  const uint32_t own = (c.probe_blocks + G - 1u) / G;
  const uint32_t q0 = b * own < c.probe_blocks ? b * own : c.probe_blocks;
  const uint32_t q1 = q0 + own < c.probe_blocks ? q0 + own : c.probe_blocks;
  const uint32_t nq = q1 - q0;
  {
    uint32_t run = 0u;
    for (uint32_t b0 = 0; b0 < nq; b0 += kFinBlock) {
      const uint32_t pb = b0 + threadIdx.x;
      const uint32_t v = pb < nq ? c.block_cnt[q0 + pb] : 0u;
      uint32_t tot;
      const uint32_t ex = block_exclusive_scan<uint32_t, kFinBlock>(v, smem, &tot);
      if (pb < nq) seg_first[pb] = run + ex;
      run += tot;
    }
    if (threadIdx.x == 0) seg_first[nq] = run;
    __syncthreads();
  }
  const uint32_t P = seg_first[nq];  // entries of this workgroup
  auto entry = [&](size_t k) -> uint64_t {
    uint32_t lo = 0u, hi = nq;  // largest b with seg_first[b] <= k
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (seg_first[mid] <= (uint32_t)k) lo = mid;
      else hi = mid;
    }
    return c.list[(size_t)(q0 + lo) * c.seg_cap + ((uint32_t)k - seg_first[lo])];
  };
  // ---- 0: the atomic mins the probe kernel deferred (a position that met its key's slot before
  //         the claimer's store had arrived, or an erased key: nobody claims those).  The probe
  //         kernel is over, so every claimer's store is in place; a barrier, then phase A reads
  //         the smallest position of every key ------------------------------------------------------
  auto give_up = [&]() {
    for (size_t k = (size_t)threadIdx.x; k < P; k += kFinBlock)
      out[entry(k) & ~kListDefer] = kInvalidIndex;
    if (threadIdx.x == 0) {
      const uint32_t e = atomicOr(c.error, 4u) | 4u;
      if (c.host_error != nullptr) *c.host_error = e;
      if (b == 0) *c.new_count = 0;  // (nobody may initialise "the rows this batch created")
    }
  };
  if constexpr (PHASE == 0 || PHASE == 3) {
    const bool deferred = PHASE == 3 || *c.defer != 0u;  // (uniform over the grid)
    if (deferred) {
      for (size_t k = (size_t)threadIdx.x; k < P; k += kFinBlock) {
        const uint32_t raw = entry(k);
        if ((raw & kListDefer) != 0u) {
          const uint64_t i = (uint64_t)(raw & ~kListDefer);
          const uint64_t slot = out[i] & ~kPendingBit;
          (void)__hip_atomic_fetch_min(&tab[slot].val, (unsigned long long)(kPendingBit | i),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if constexpr (PHASE == 3) return;  // (the launch boundary is the barrier)
      if (!grid_barrier(c.barrier, c.barrier_odd, G, c.spin_limit, bar_gen)) {
        give_up();
        return;
      }
      bar_gen = (bar_gen + 1u) & ~kBarAbort;
      if (b == 0 && threadIdx.x == 0) *c.defer = 0u;  // (every workgroup has read it)
    }
  }
  if constexpr (PHASE == 3) return;
  if constexpr (PHASE == 1) {
    if (b == 0 && threadIdx.x == 0) *c.defer = 0u;  // (read by the launch in front of this one)
  }
  // (workgroup 0 moves the counter and flips the parity only behind the barrier / in the second
  //  launch, which reads the first launch's snapshot)
  const uint64_t c0 = PHASE == 2 ? ((uint64_t)c.snap[3] << 32 | c.snap[2]) : *c.counter;
  const size_t nl = live_count(d_n, n);
  const size_t gtid = (size_t)b * kFinBlock + threadIdx.x, gthreads = (size_t)G * kFinBlock;
  const uint32_t par = PHASE == 2 ? c.snap[1] : (*c.parity & 1u);
  if constexpr (PHASE == 1) {
    if (b == 0 && threadIdx.x == 0) {
      c.snap[0] = 1u;
      c.snap[1] = par;
      c.snap[2] = (uint32_t)c0;
      c.snap[3] = (uint32_t)(c0 >> 32);
    }
  }
  unsigned long long* const masks = c.masks2 + (size_t)par * c.mask_words;
  if constexpr (PHASE != 2) {  // the other mask buffer is nobody's at the moment: all zero for the next inserting batch
    unsigned long long* const other = c.masks2 + (size_t)(1u - par) * c.mask_words;
    for (size_t w = gtid; w < c.mask_words; w += gthreads) other[w] = 0ull;
    for (size_t r = gtid; r < (size_t)kFinRegions; r += gthreads)
      c.region_cnt[(1u - par) * kFinRegions + r] = 0u;
  }
  uint32_t* const region_cnt = c.region_cnt + par * kFinRegions;  // (zero on entry, like masks)
  // regions of `per` consecutive positions (a multiple of 64, at most kFinRegions of them)
  size_t per = (nl + kFinRegions - 1) / kFinRegions;
  per = (per + 63) / 64 * 64;
  const uint32_t R = (uint32_t)((nl + per - 1) / per);
  // ---- A: the pending positions that hold the FIRST occurrence of their key set their bit and
  //         count themselves into their region --------------------------------------------------
  // (the first kKeep entries of a thread stay in registers for phase D: two dependent loads less)
  // kept per entry: position, slot and the slot's value as phase A read it -- PENDING | first
  // position of the key.  Nobody publishes a row in front of the barrier, so phase D ranks from
  // the kept value and never reads the table again (one random read per unseen key less; a dense
  // list of 6656 entries per workgroup fits the eight kept rounds whole).
  constexpr int kKeep = PHASE == 0 ? 8 : 0;  // (two launches: nothing survives in registers)
  uint32_t keep_i[kKeep > 0 ? kKeep : 1];
  uint64_t keep_slot[kKeep > 0 ? kKeep : 1], keep_v[kKeep > 0 ? kKeep : 1];
  if constexpr (PHASE != 2) {
    // A DENSE list (most keys of the batch unseen: a first epoch): the 64 entries of a wavefront
    // are consecutive positions of one probe workgroup, i.e. one or two mask words, and 64 atomics
    // on one address serialise at the memory side.  The lanes that share a word then OR their bits
    // across the wavefront and one of them issues the two atomics (all positions of a word lie in
    // one region: `per` is a multiple of 64).  For sparse lists that form was measured slower
    // (round 3), hence the threshold: two entries per thread.
    const bool dense = P >= 2u * (uint32_t)kFinBlock;
    const int lane = threadIdx.x & 63;
    auto mark = [&](bool first, uint64_t i) {  // (called by every lane of the workgroup)
      if (!dense) {
        if (first) {
          atomicOr(&masks[i >> 6], 1ull << (i & 63));
          atomicAdd(&region_cnt[i / per], 1u);
        }
        return;
      }
      const uint32_t word = (uint32_t)(i >> 6);
      unsigned long long todo = __ballot(first);
      while (todo != 0ull) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t w0 = (uint32_t)__shfl((int)word, leader, 64);
        const bool mine = first && word == w0;
        const unsigned long long same = __ballot(mine);
        unsigned long long bits = mine ? (1ull << (i & 63)) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bits, d, 64);
          const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bits >> 32), d, 64);
          bits |= ((unsigned long long)hi << 32) | lo;
        }
        if (lane == leader) {
          atomicOr(&masks[w0], bits);
          atomicAdd(&region_cnt[((size_t)w0 << 6) / per], (uint32_t)__popcll(bits));
        }
        todo &= ~same;
      }
    };
#pragma unroll
    for (int j = 0; j < kKeep; j++) {
      const size_t k = (size_t)threadIdx.x + (size_t)j * kFinBlock;
      keep_i[j] = 0u;
      keep_slot[j] = 0;
      keep_v[j] = 0;
      bool first = false;
      if (k < P) {
        const uint64_t i = (uint64_t)(entry(k) & ~kListDefer);
        const uint64_t slot = out[i] & ~kPendingBit;
        keep_i[j] = (uint32_t)i;
        keep_slot[j] = slot;
        // (memory-side atomics of this launch may have written the word: past the caches)
        keep_v[j] = ld_agent(&tab[slot].val);
        first = keep_v[j] == (kPendingBit | i);
      }
      mark(first, (uint64_t)keep_i[j]);
    }
    for (size_t k0 = (size_t)kKeep * kFinBlock; k0 < P; k0 += kFinBlock) {  // (uniform trips)
      const size_t k = k0 + threadIdx.x;
      bool first = false;
      uint64_t i = 0;
      if (k < P) {
        i = (uint64_t)(entry(k) & ~kListDefer);
        const uint64_t slot = out[i] & ~kPendingBit;
        first = ld_agent(&tab[slot].val) == (kPendingBit | i);
      }
      mark(first, i);
    }
  }
  if constexpr (PHASE == 1) return;  // (the launch boundary is the barrier)
  if constexpr (PHASE == 0) {
    if (!grid_barrier(c.barrier, c.barrier_odd, G, c.spin_limit, bar_gen)) {
      // the barrier did not open (the grid was not resident as a whole: a partitioned or masked
      // device, or other work holding the CUs): no row is handed out to ANY position of the batch
      // (grid_barrier: all workgroups fail together).  The pending positions of this workgroup's
      // share get "no row" instead of keeping PENDING | slot, error bit 4 tells the host, and the
      // generation word stays poisoned: every later batch fails the same way until the host has
      // put the table right (HashTable::recover, or clear) -- the slots this batch claimed still
      // hold PENDING | position, which a later batch must not mistake for its own
      give_up();
      return;
    }
  }
  // ---- S: every workgroup scans the region counts (into LDS); workgroup 0 hands out the row
  //         range ------------------------------------------------------------------------------------
  __shared__ uint32_t region_base[kFinRegions];
  {
    // thread t: regions [4 t, 4 t + 4) -- four independent loads, one scan of the workgroup
    static_assert(kFinRegions == 4 * kFinBlock, "one scan covers every region");
    uint32_t v4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t r = 4u * threadIdx.x + (uint32_t)q;
      v4[q] = r < R ? ld_agent(&region_cnt[r]) : 0u;
    }
    uint32_t run;
    uint32_t ex = block_exclusive_scan<uint32_t, kFinBlock>(v4[0] + v4[1] + v4[2] + v4[3], smem, &run);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      region_base[4u * threadIdx.x + (uint32_t)q] = ex;
      ex += v4[q];
    }
    const uint32_t total = run;
    __syncthreads();
    if (b == 0) {
      if (threadIdx.x == 0) {
        *c.base = c0;
        *c.new_count = total;
        // more unseen keys than free rows: the counter stops at the capacity, the keys beyond it
        // get no row (they pool as zeros and are skipped by the update, like an eval miss), and
        // error bit 1 makes check_overflow() fail as the reference's does
        // (localized_slot_sparse_embedding_hash.hpp:552-569) -- nothing is read or written
        // outside the [capacity] row arrays
        const uint64_t head = (c0 + total > capacity) ? capacity : c0 + total;
        *c.counter = head;
        if (c0 + total > capacity) atomicOr(c.error, 2u);
        *c.latched = 1u;
        *c.pending = 0u;
        *c.parity = 1u - par;
        post_to_host(c, head);
      }
    }
  }
  // ---- D: rows.  rank of a first position fp = firsts in the regions before its own + firsts of
  //         its own region in front of it (mask words, a few tens at most) -------------------------
  auto resolve = [&](uint64_t i, uint64_t slot, uint64_t v) {
    if (v < kPendingBit) {  // the key's first occurrence has already published the row
      out[i] = row_of(v);
      return;
    }
    const uint64_t fp = v & ~kPendingBit;
    const uint32_t r = (uint32_t)(fp / per);
    const size_t w0 = ((size_t)r * per) >> 6, wf = fp >> 6;
    uint32_t rank = region_base[r] +
                    (uint32_t)__popcll(ld_agent(&masks[wf]) & ((1ull << (fp & 63)) - 1ull));
    for (size_t w = w0; w < wf; w += 8) {  // (independent loads, clamped: eight in flight --
                                           //  a region of the Criteo-1TB batch in one trip)
      unsigned long long m[8];
#pragma unroll
      for (int q = 0; q < 8; q++) m[q] = ld_agent(&masks[w + q < wf ? w + q : w]);
#pragma unroll
      for (int q = 0; q < 8; q++) rank += w + q < wf ? (uint32_t)__popcll(m[q]) : 0u;
    }
    uint64_t fin = c0 + rank;
    if (fin >= capacity) fin = kInvalidIndex;  // table full: the key gets no row
    out[i] = fin;
    if (i == fp) {
      tab[slot].val = fin == kInvalidIndex ? kNoRow : fin;
      new_positions[rank] = i;
      if (sink.slot_id != nullptr && fin != kInvalidIndex) record_slot_id(sink, i, fin);
    }
  };
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    const size_t k = (size_t)threadIdx.x + (size_t)j * kFinBlock;
    if (k < P) resolve((uint64_t)keep_i[j], keep_slot[j], keep_v[j]);
  }
  for (size_t k = (size_t)threadIdx.x + (size_t)kKeep * kFinBlock; k < P; k += kFinBlock) {
    const uint64_t i = (uint64_t)(entry(k) & ~kListDefer);
    const uint64_t slot = out[i] & ~kPendingBit;
    resolve(i, slot, tab[slot].val);
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
        res = row_of(ent[u].y);
      } else if ((long long)ent[u].x != empty) {  // collision: walk on
        uint64_t sl = (slot[u] + 1 == size) ? 0 : slot[u] + 1;
        for (uint64_t probes = 1; probes <= size; ++probes) {
          const long long cur = tab[sl].key;
          if (cur == k64) {
            res = row_of(tab[sl].val);
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

// after a finish kernel that gave up (error bit 4): the slots of the batch's keys that still hold
// PENDING | position go back to "no row" (what an erased key holds: the next get_insert of the key
// hands it a row), so that the batch can be resolved again
template <typename K>
__global__ void __launch_bounds__(kBlock)
    ht_unpend_kernel(HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys, size_t n,
                     uint32_t* d_error) {
  const long long empty = KeyTraits<K>::empty;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const K key = keys[i];
    const long long k64 = widen<K>(key);
    uint64_t slot = (uint64_t)murmur3_key(key) % size;
    for (uint64_t probes = 0; probes < size; ++probes) {
      const long long cur = tab[slot].key;
      if (cur == k64) {
        const unsigned long long v = tab[slot].val;
        if (v >= kPendingBit && v != kInvalidIndex) tab[slot].val = kInvalidIndex;  // (same value from every duplicate)
        break;
      }
      if (cur == empty) break;
      slot = (slot + 1 == size) ? 0 : slot + 1;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAnd(d_error, ~4u);
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
      if (i < size)
        c += (tab[i].key != empty && tab[i].val != kInvalidIndex && tab[i].val != kNoRow) ? 1u : 0u;
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
        f = e.key != empty && e.val != kInvalidIndex && e.val != kNoRow;
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
  HCTR_HIP(hipMalloc(&scal, 96));
  d_counter = scal;
  d_base = scal + 1;
  d_new_count = scal + 2;
  d_scratch64 = scal + 3;
  d_pending = reinterpret_cast<uint32_t*>(scal + 4);
  d_error = reinterpret_cast<uint32_t*>(scal + 4) + 1;
  d_latched = reinterpret_cast<uint32_t*>(scal + 5);
  d_barrier = reinterpret_cast<uint32_t*>(scal + 6);
  d_parity = reinterpret_cast<uint32_t*>(scal + 7);
  d_snap = reinterpret_cast<uint32_t*>(scal + 8);
  d_defer = reinterpret_cast<uint32_t*>(scal + 10);
  d_barrier_odd = reinterpret_cast<uint32_t*>(scal + 11);
  HCTR_HIP(hipMemset(scal, 0, 96));
  return clear(nullptr);
}

int HashTable::destroy() {
  if (entries) (void)hipFree(entries);
  if (d_counter) (void)hipFree(d_counter);
  if (tile_sums) (void)hipFree(tile_sums);
  if (new_positions) (void)hipFree(new_positions);
  if (fin_masks) (void)hipFree(fin_masks);
  if (pend_list) (void)hipFree(pend_list);
  fin_masks = nullptr;
  pend_list = nullptr;
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
  HCTR_HIP(hipMemsetAsync(d_defer, 0, 16, s));  // (and the odd generations' arrival word behind it)
  if (fin_masks)
    HCTR_HIP(hipMemsetAsync(fin_masks, 0, mask_words * 2 * sizeof(unsigned long long) + 2 * kFinRegions * 4, s));
  return HCTR_OK;
}

int HashTable::reserve(size_t n) {
  // scratch must also cover a dump over `size` slots
  size_t need_tiles = ceil_div<size_t>(n > size ? n : size, kHtTile) + 1;
  if (n <= max_n && tile_sums != nullptr) return HCTR_OK;
  if (tile_sums) (void)hipFree(tile_sums);
  if (new_positions) (void)hipFree(new_positions);
  if (fin_masks) (void)hipFree(fin_masks);
  if (pend_list) (void)hipFree(pend_list);
  // (+ the finish kernel's per-workgroup counts and their scan)
  HCTR_HIP(hipMalloc(&tile_sums, (need_tiles + 2 * kHtFinishBlocksMax) * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&new_positions, (n > 0 ? n : 1) * sizeof(uint64_t)));
  mask_words = n / 64 + 2;
  HCTR_HIP(hipMalloc(&fin_masks, mask_words * 2 * sizeof(unsigned long long) + 2 * kFinRegions * 4));
  HCTR_HIP(hipMemset(fin_masks, 0, mask_words * 2 * sizeof(unsigned long long) + 2 * kFinRegions * 4));
  region_cnt = reinterpret_cast<uint32_t*>(fin_masks + 2 * mask_words);
  HCTR_HIP(hipMemset(d_parity, 0, sizeof(uint32_t)));
  {  // list segments: every probe workgroup's share of n, rounded up to whole passes
    const size_t blocks = ceil_div<size_t>(n > 0 ? n : 1, (size_t)kBlock * kHtUnroll);
    const size_t g = blocks < (size_t)kProbeMaxBlocks ? blocks : (size_t)kProbeMaxBlocks;
    const size_t entries_ = (n > 0 ? n : 1) + (g + 1) * kBlock * kHtUnroll;
    HCTR_HIP(hipMalloc(&pend_list, (entries_ + kProbeMaxBlocks) * sizeof(uint32_t)));
    block_cnt = pend_list + entries_;
  }
  max_n = n;
  return HCTR_OK;
}

int HashTable::get_insert(const void* keys, size_t n, const uint64_t* d_n, uint64_t* out,
                          hipStream_t s, const SlotSink* sink_in, const IndexExtras* ex) {
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
  IndexExtras none;
  const IndexExtras& x = ex ? *ex : none;
  // (bit 31 of a list entry is the "min deferred" flag)
  HCTR_REQUIRE(n < 0x80000000ull, "get_insert: more than 2^31 - 1 keys in one call");
  const size_t work = n > x.n_offsets ? n : x.n_offsets;
  const int grid = grid_for(ceil_div<size_t>(work, kHtUnroll), kBlock, kProbeMaxBlocks);
  // every key a workgroup may leave pending has a place in its segment of the list
  const size_t per_pass = (size_t)grid * kBlock * kHtUnroll;
  const uint32_t seg_cap = (uint32_t)(ceil_div<size_t>(n, per_pass) * kBlock * kHtUnroll);
  if (key_type == HCTR_KEY_U32) {
    hipLaunchKernelGGL(ht_probe_insert_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const uint32_t*)keys, n, d_n, out, d_pending, pend_list, block_cnt,
                       seg_cap, d_error, d_defer, d_new_count, (const uint32_t*)x.ro_src, (uint32_t*)x.ro_dst,
                       x.n_offsets, x.one_hot);
  } else {
    hipLaunchKernelGGL(ht_probe_insert_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, entries,
                       size, (const long long*)keys, n, d_n, out, d_pending, pend_list, block_cnt,
                       seg_cap, d_error, d_defer, d_new_count, (const long long*)x.ro_src, (long long*)x.ro_dst,
                       x.n_offsets, x.one_hot);
  }
  HCTR_LAUNCH_CHECK();
  FinishCtl c;
  c.pending = d_pending;
  c.list = pend_list;
  c.block_cnt = block_cnt;
  c.probe_blocks = (uint32_t)grid;
  c.seg_cap = seg_cap;
  c.latched = d_latched;
  c.error = d_error;
  c.barrier = d_barrier;
  c.defer = d_defer;
  c.barrier_odd = d_barrier_odd;
  c.counter = d_counter;
  c.base = d_base;
  c.new_count = d_new_count;
  c.region_cnt = region_cnt;
  c.masks2 = fin_masks;
  c.parity = d_parity;
  c.mask_words = mask_words;
  c.one_hot_next = x.one_hot_next;
  c.host_rows = x.host_rows;
  c.host_seq = x.host_seq;
  c.seq = x.seq;
  c.host_error = x.host_error;
  {  // (read per call: a test narrows it to force the give-up path)
    const char* e = getenv("HCTR_HT_SPIN_LIMIT");
    c.spin_limit = e ? (uint32_t)strtoul(e, nullptr, 10) : kSpinLimit;
  }
  // few positions: fewer workgroups (every one of them takes part in the barrier), and never more
  // than the device can hold at once (a CPX partition or a CU-masked device has far fewer than
  // 256 CUs: a grid that is not resident as a whole could only time out at its barrier)
  // (per device: a process may drive partitions / CU masks of different sizes)
  size_t resident = (size_t)kHtFinishBlocks;
  {
    static std::mutex mu;
    static std::map<int, size_t> per_device;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      std::lock_guard<std::mutex> lk(mu);
      auto it = per_device.find(dev);
      if (it == per_device.end()) {
        int per_cu = 0, cus = 0;
        size_t r = (size_t)kHtFinishBlocks;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ht_finish_kernel<0>, kFinBlock,
                                                         0) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            per_cu >= 1 && cus >= 1)
          r = (size_t)per_cu * (size_t)cus;
        else
          (void)hipGetLastError();
        it = per_device.emplace(dev, r).first;
      }
      resident = it->second;
    } else {
      (void)hipGetLastError();
    }
  }
  // (HCTR_HT_FINISH_BLOCKS: the cap, for measurements; at most kHtFinishBlocksMax)
  static const size_t cap = [] {
    const char* e = getenv("HCTR_HT_FINISH_BLOCKS");
    long v = e ? atol(e) : kHtFinishBlocks;
    if (v < 1) v = 1;
    if (v > kHtFinishBlocksMax) v = kHtFinishBlocksMax;
    return (size_t)v;
  }();
  size_t fg = ceil_div<size_t>(n, (size_t)kFinBlock * 4);
  if (fg > cap) fg = cap;
  if (fg > resident) fg = resident;
  if (fg < 1) fg = 1;
  c.snap = d_snap;
  if (x.two_launches) {
    hipLaunchKernelGGL(ht_finish_kernel<3>, dim3((int)fg), dim3(kFinBlock), 0, s, entries, out, n,
                       d_n, c, new_positions, sink, capacity);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ht_finish_kernel<1>, dim3((int)fg), dim3(kFinBlock), 0, s, entries, out, n,
                       d_n, c, new_positions, sink, capacity);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ht_finish_kernel<2>, dim3((int)fg), dim3(kFinBlock), 0, s, entries, out, n,
                       d_n, c, new_positions, sink, capacity);
  } else {
    hipLaunchKernelGGL(ht_finish_kernel<0>, dim3((int)fg), dim3(kFinBlock), 0, s, entries, out, n,
                       d_n, c, new_positions, sink, capacity);
  }
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

int HashTable::recover(const void* keys, size_t n, hipStream_t s) {
  if (n > 0) {
    const int grid = grid_for(n, kBlock);
    if (key_type == HCTR_KEY_U32)
      hipLaunchKernelGGL(ht_unpend_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, entries, size,
                         (const uint32_t*)keys, n, d_error);
    else
      hipLaunchKernelGGL(ht_unpend_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, entries, size,
                         (const long long*)keys, n, d_error);
    HCTR_LAUNCH_CHECK();
  }
  // the scalars of the insert protocol (not the row counter, not the error word): pending,
  // latched, barrier {arrived, generation}, mask parity -- and both mask buffers / region counts
  HCTR_HIP(hipMemsetAsync(d_pending, 0, sizeof(uint32_t), s));
  HCTR_HIP(hipMemsetAsync(d_latched, 0, sizeof(uint32_t), s));
  HCTR_HIP(hipMemsetAsync(d_barrier, 0, 2 * sizeof(uint32_t), s));
  HCTR_HIP(hipMemsetAsync(d_barrier_odd, 0, sizeof(uint32_t), s));
  HCTR_HIP(hipMemsetAsync(d_parity, 0, sizeof(uint32_t), s));
  HCTR_HIP(hipMemsetAsync(d_defer, 0, sizeof(uint32_t), s));
  if (fin_masks)
    HCTR_HIP(hipMemsetAsync(fin_masks, 0, mask_words * 2 * sizeof(unsigned long long) + 2 * kFinRegions * 4, s));
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

int hctr_ht_error_flags(hctr_hashtable* ht, hctr_stream_t stream, uint32_t* out) {
  HCTR_REQUIRE(ht && out, "null pointer");
  return ht->impl.error_flags(as_stream(stream), out);
}

int hctr_ht_recover(hctr_hashtable* ht, const void* keys, size_t n, hctr_stream_t stream) {
  HCTR_REQUIRE(ht && (n == 0 || keys), "null pointer");
  return ht->impl.recover(keys, n, as_stream(stream));
}

}  // extern "C"
