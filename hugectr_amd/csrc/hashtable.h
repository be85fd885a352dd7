// hashtable.h -- device open-addressing key -> row-index map (internal C++ view).
// Replaces HashTable<Key,size_t> (R/HugeCTR/include/hashtable/nv_hashtable.hpp:31-189).
#pragma once
#include "common.h"

namespace hctr {

struct HtEntry {
  long long key;           // u32 keys are zero-extended; empty = KeyTraits<K>::empty
  unsigned long long val;  // row index; kInvalidIndex = unused
};

constexpr uint64_t kPendingBit = 1ull << 63;
// value of a key that met a full table: it keeps its hash slot but owns no row.  Below
// kPendingBit, so later batches resolve it on the fast path (it reads as kInvalidIndex) instead
// of re-entering the insert protocol and raising the overflow flag again
constexpr uint64_t kNoRow = kPendingBit - 1;
constexpr int kHtTile = 1024;  // positions per compaction tile
// workgroups of the cooperative finish kernel (co-resident; one per 4096 positions up to this cap).
// 256: a batch of 1.7 M unseen keys 772 -> 647 us against 128 (512: no further gain), nothing
// changes for batches with few unseen keys (the kernel is a latency chain then)
constexpr int kHtFinishBlocks = 256;
constexpr int kHtFinishBlocksMax = 512;  // upper bound of HCTR_HT_FINISH_BLOCKS

// what get_insert can do on the side of its two launches (all optional)
struct IndexExtras {
  // world == 1: private copy of the batch's row offsets + "every bucket holds one key" check,
  // done by the probe kernel's threads (no launch of its own)
  const void* ro_src = nullptr;
  void* ro_dst = nullptr;
  size_t n_offsets = 0;
  uint32_t* one_hot = nullptr;       // cleared when ro_src[i] != i for some i
  uint32_t* one_hot_next = nullptr;  // preset to 1 for the NEXT batch by the finish kernel
  // pinned host words the finish kernel posts to (no copy launch, no event): rows handed out so
  // far, then `seq` -- the host reads seq first, so the row count it pairs with it is never older
  uint64_t* host_rows = nullptr;
  uint64_t* host_seq = nullptr;
  uint64_t seq = 0;
  uint32_t* host_error = nullptr;    // the error flags
  // the finish kernel as two launches instead of one with a grid barrier: for an index stage that
  // runs beside other work (its workgroups cannot count on being resident together)
  bool two_launches = false;
};

// where get_insert records the slot id of newly inserted rows (embedding dump needs it)
struct SlotSink {
  uint64_t* slot_id;
  const void* row_offset;  // key-typed CSR row offsets of the batch
  size_t buckets;
  int buckets_per_sample, rank, world, localized;
};

struct HashTable {
  HtEntry* entries = nullptr;
  uint64_t size = 0;      // physical slots = (size_t)(capacity / 0.75f)
  uint64_t capacity = 0;  // max_vocabulary_size_per_gpu
  int key_type = HCTR_KEY_I64;
  // device scalars
  uint64_t* d_counter = nullptr;    // value head (next row index)
  uint64_t* d_base = nullptr;       // counter snapshot used by the current get_insert
  uint32_t* d_pending = nullptr;    // positions of the current batch that hold an unseen key
  uint32_t* d_latched = nullptr;    // d_pending as seen by the scan step of this get_insert
  uint32_t* d_error = nullptr;      // bit0: probe overflow (table full) bit1: counter > capacity
                                    // bit2: the finish kernel's grid barrier did not open (recover)
  uint64_t* d_new_count = nullptr;  // number of keys inserted by the last get_insert
  // scratch sized for max_n positions
  size_t max_n = 0;
  uint32_t* d_barrier = nullptr;    // {arrived, generation} of the finish kernel's grid barrier
  uint32_t* tile_sums = nullptr;   // [ceil(max_n / kHtTile) + 1] (+ kHtFinishBlocks block totals)
  uint64_t* new_positions = nullptr;  // [max_n] positions (into keys) of newly inserted keys
  unsigned long long* fin_masks = nullptr;  // 2 x [mask_words] first-occurrence masks + prefixes
  size_t mask_words = 0;
  uint32_t* region_cnt = nullptr;   // [2048] first occurrences per region of positions (finish)
  uint32_t* pend_list = nullptr;    // positions of the batch whose key was not in the table, one
                                    // segment per workgroup of the probe kernel
  uint32_t* block_cnt = nullptr;    // entries of every segment (behind the list)
  uint32_t* d_parity = nullptr;     // which mask buffer the next inserting batch takes
  uint32_t* d_snap = nullptr;       // two-launch finish: what its first half saw (FinishCtl::snap)
  uint32_t* d_barrier_odd = nullptr;  // arrivals of the odd generations (d_barrier[0]: the even ones)
  uint32_t* d_defer = nullptr;      // the probe kernel left some atomic mins to the finish kernel
  uint64_t* d_scratch64 = nullptr;  // 1 element

  int create(size_t capacity, int key_type);
  int destroy();
  int clear(hipStream_t s);
  int reserve(size_t n);  // scratch for batches up to n keys
  int get_insert(const void* keys, size_t n, const uint64_t* d_n, uint64_t* out, hipStream_t s,
                 const SlotSink* sink = nullptr, const IndexExtras* extras = nullptr);
  int get_mark(const void* keys, size_t n, const uint64_t* d_n, uint64_t* out, hipStream_t s);
  int insert(const void* keys, const uint64_t* vals, size_t n, hipStream_t s);
  int count(hipStream_t s, size_t* out);
  int value_head(hipStream_t s, size_t* out);
  int set_value_head(size_t v, hipStream_t s);
  int dump(int64_t* d_keys, uint64_t* d_vals, size_t* count, hipStream_t s);
  int error_flags(hipStream_t s, uint32_t* out);
  // after error bit 4 (the finish kernel's grid barrier did not open: NO position of that batch got
  // a row): puts back the slots the batch's keys left pending and the insert protocol's scalars /
  // masks, clears bit 4.  The batch can then be resolved again (IndexExtras::two_launches needs no
  // barrier).  `keys` = the n keys of the failed get_insert.
  int recover(const void* keys, size_t n, hipStream_t s);
};

}  // namespace hctr

struct hctr_hashtable {
  hctr::HashTable impl;
};
