// sparse_update.h -- fused backward + sparse optimizer (internal C++ view).
#pragma once
#include "common.h"

namespace hctr {

struct OptState {
  int optimizer = HCTR_OPT_SGD;
  int update_type = HCTR_UPDATE_LOCAL;
  float lr = 0.f;
  float beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-7f;
  float momentum_factor = 0.f;
  float scaler = 1.f;
  int atomic_update = 0;
  float ftrl_lambda1 = 0.f, ftrl_lambda2 = 0.f, ftrl_beta = 0.f;  // EBC static tables only
  uint64_t times = 0;  // Adam step counter (incremented before each update)
  int state_half = 0;  // optimizer state holds fp16 values (fp16 embeddings)
};

constexpr int kOptStoreSumId = 1000;  // internal: table[row] = per-row gradient sum

struct SparseUpdater {
  size_t max_nnz = 0;
  size_t max_vocab = 0;
  int D = 0;
  bool key32 = true;  // sort keys fit 32 bits
  // rows handed out so far are < row_bound (0 = unknown): the sort then covers log2(row_bound)
  // key bits instead of log2(max_vocab) -- one digit pass less while a table is filling up
  size_t row_bound = 0;
  // sort buffers
  void* sort_keys_in = nullptr;
  void* sort_keys_out = nullptr;
  uint32_t* sort_vals_in = nullptr;
  uint32_t* sort_vals_out = nullptr;
  void* sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  // run detection
  uint32_t* tile_sums = nullptr;
  uint32_t* run_start = nullptr;  // [max_nnz + 1]
  uint64_t* d_num_runs = nullptr;
  // tile-based segmented reduce: partial sums of runs that cross tile borders
  float* seg_head = nullptr;    // [tiles][D]
  float* seg_tail = nullptr;    // [tiles][D]
  float* gsum = nullptr;           // [max_nnz][D] per-run gradient sums, indexed by run start
  uint32_t* span_list = nullptr;   // [tiles] tiles in which a long (multi-tile) run starts
  uint32_t* span_count = nullptr;  // device counters: [0] span_list, [2..3] big runs / chunks (u64)
  uint32_t* big_list = nullptr;    // [4][tiles] runs longer than kCombBigTiles tiles (seg_combine_big)
  size_t big_stride = 0;
  Profiler* prof = nullptr;
  bool allow_ftrl = false;  // the legacy embedding rejects Ftrl as the reference does (q9)
  // the (row, bucket) sort needs only the index stage's output, not the gradients: presort() runs
  // it on a side stream while the caller's stream does the gather and the dense tower
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_sorted = nullptr;
  // presorted mode: the caller supplies the sorted (row, bucket) list (unique-row exchange)
  const uint32_t* ext_rows = nullptr;
  const uint32_t* ext_buckets = nullptr;
  // mean combiner: CSR whose bucket lengths are the divisor, when it is not row_offset itself
  // (distributed embedding on N > 1 GPUs: the unfiltered full-batch offsets); same offset type
  const void* scale_row_offset = nullptr;
  // gradient map (embedding_collection on one GPU, batch-major output): bucket u's gradient row is
  // (u % map_inner) * map_outer + u / map_inner of `grad` -- the [sample][lookup] gradient read in
  // place instead of being transposed into bucket order first.  0 = identity.  Sum combiner only.
  uint32_t map_inner = 0, map_outer = 0;
  // device word, non-zero <=> every bucket of the batch holds exactly one key (the index stage's
  // one-hot flag; world == 1): the sort then reads the rows where they lie and the pair expansion
  // is skipped (RsFirst).  nullptr = unknown: the pairs are expanded.
  const uint32_t* one_hot_flag = nullptr;
  // hot rows of one-hot batches (sparse_update.hip, "hot rows of a one-hot batch"): positions whose
  // row is < hot_rows are summed per (stream, chunk) by hot_chunk_kernel and never sorted.
  // hot_streams: 0 = layout unknown (no hot path); S = the batch is sample-major with S buckets
  // per sample (all positions p with the same p % S come from one table); 1 = positions as they lie
  uint32_t hot_streams = 0;
  uint32_t hot_rows = 0;        // H; 0 = off (HCTR_HOT_ROWS, default 8192, at most 16384)
  hipStream_t hot_side = nullptr;  // the cold pairs' chain (default priority)
  bool hot_serial = false;         // HCTR_HOT_SERIAL=1: no side stream (measurements)
  size_t hot_min_n = 0;         // batches with fewer positions keep the plain path (HCTR_HOT_MIN)
  uint32_t hot_chunks_max = 0;  // chunks the tables below have room for
  uint16_t* hot_loc = nullptr;        // [hot_rows][hot_chunks_max] partial number of (row, chunk)
  uint32_t* hot_S = nullptr;           // [hot_chunks_max][4096] a chunk's hot entries, sorted
  uint32_t* hot_meta = nullptr;        // [hot_chunks_max][2] entries, first pool slot
  uint32_t* hot_tpref = nullptr;       // [hot_chunks_max][129] run starts in front of a tile
  uint32_t* hot_items = nullptr;       // tiles with entries (work list of hot_reduce_kernel)
  uint32_t* hot_loc_blk = nullptr;     // [hot_rows] blocks of 32 chunks that hold a partial of the row
  uint32_t* hot_joins = nullptr;       // [.][3] runs that cross tile borders (hot_join_kernel)
  uint32_t* hot_counts = nullptr;      // two alternating sets {pool slots, items, joins, -} + [8] pairs the sort kept
  uint32_t hot_parity = 0;             // set the next update takes
  float* hot_head = nullptr;           // [hot_chunks_max * 128][D] tile partials of runs that
  float* hot_tail = nullptr;           //   cross tile borders inside a chunk
  // cold rows of the same batches (sparse_update.hip, "cold rows of a one-key-per-position batch"):
  // counted per row instead of sorted.  HCTR_COLD_COUNT=0 keeps the filtering radix sort + segmented
  // reduce of round 4 (measurements, the bit-equality tests).
  bool cold_count = true;
  uint32_t* cold_cnt = nullptr;     // [max_vocab] per-row counter / base word, zero between updates
  uint32_t* cold_rank = nullptr;    // [max_nnz]
  uint32_t* cold_plist = nullptr;   // [max_nnz]
  uint32_t* cold_bkt = nullptr;     // [max_nnz]
  void* cold_dlist = nullptr;       // [max_nnz] uint2
  void* cold_singles = nullptr;     // [max_nnz] uint2
  void* cold_segs = nullptr;        // [max_nnz / 2 + 1] uint4
  void* cold_longs = nullptr;       // [max_nnz / 2 + 1] uint4 (runs longer than short_max >= 8)
  uint32_t* cold_counts = nullptr;  // two alternating sets of counters
  void* pre_plan = nullptr;         // geometry / buffers / events of the batch in hand (PrePlan)
  size_t early_n = 0;  // > 0: sort_*_out hold the sorted pairs of (early_vi, early_buckets)
  const uint64_t* early_vi = nullptr;
  size_t early_buckets = 0;

  // eager_hot: the owner feeds one-hot batches with a device flag (the legacy embedding): the
  // hot-row path's tables and stream are set up here.  Others get them on the first update that
  // takes the path, if any (hot_buffers) -- measured: set up inside a step of the uniform-key leg
  // (213 GB of tables already resident) every random-access kernel of the step ran 25-90 % slower
  // from then on (profiles/r4_uniform_leg_lazy_hot_buffers.txt), so the one owner that does take
  // the path never does it that way.
  int create(size_t max_nnz, size_t max_vocab, int D, bool eager_hot = false);
  int hot_buffers(hipStream_t s);
  int destroy();
  // optional: start sorting n >= live nnz (row, bucket) pairs now, concurrently with stream s
  int presort(size_t buckets, size_t n, const void* row_offset, int key_type,
              const uint64_t* value_index, hipStream_t s);
  // optional: the grouping work of the hot-row path (hot rows' chunk sort; cold rows' count / base
  // / scatter) needs the rows only -- start it now on the side streams, behind what s holds; the
  // update of the same (value_index, nnz, buckets) then runs the reduces alone.  A no-op when the
  // batch would not take that path.  one_hot_flag / map / hot_streams as for update().
  int prework(size_t buckets, size_t nnz, int combiner, const void* row_offset, int key_type,
              const uint64_t* value_index, hipStream_t s);
  // row_offset/key_type as in the forward; top_grad [buckets][D] of grad_dtype.
  int update(size_t buckets, size_t nnz, int combiner, const void* row_offset, int key_type,
             const uint64_t* value_index, const void* top_grad, int grad_dtype, const OptState& opt,
             float* table, float* state0, float* state1, uint64_t* prev_time, hipStream_t s);
};

int materialize_wgrad(size_t buckets, int D, int combiner, const void* row_offset, int key_type,
                      const void* top_grad, void* wgrad, int dtype, hipStream_t s);

}  // namespace hctr
