// det.hip -- dynamic embedding table: key -> embedding vector maps that grow on demand.
//
// Replaces det::DynamicEmbeddingTable<Key, float>
// (R/third_party/dynamic_embedding_table/dynamic_embedding_table.hpp:25-66, .cu) and the fused
// optimizer step of embedding::DynamicEmbeddingTable::update
// (R/HugeCTR/embedding_storage/dynamic_embedding.cu:176-330, optimizers.cuh:29-233).
//
// MI355X-first layout instead of cuco's chain of sub-maps with inline vectors: per class (= one
// embedding dimension) an open-addressing index (the path's HashTable: 16-B {key,row} entries,
// deterministic first-occurrence row numbers) in front of ONE dense row store [capacity][dim]
// fp32.  Lookups are then the same coalesced row gathers as the static path, new rows are dense at
// the tail (initialised by a counter-based RNG, so results do not depend on thread scheduling or
// on std::random_device as in the reference), and growth is a doubling re-allocation + re-index.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "hashtable.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// cuco::initializer (initializer.cuh:36-55): a constant, or curand_uniform -> (0, 1]
__device__ __forceinline__ float det_init_value(int mode, float val, uint64_t seed, uint64_t row,
                                                uint32_t e) {
  if (mode == 0) return val;
  const uint64_t h = splitmix64(seed ^ splitmix64(row * 0x100000001B3ull + e));
  return ((float)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);  // 24 random bits -> (0, 1]
}

// rows created by the get_insert that just ran: positions (into the key batch) of the first
// occurrence of every unseen key are in new_positions[0 .. *d_new_count)
__global__ void __launch_bounds__(kBlock)
    det_init_rows_kernel(const uint64_t* __restrict__ new_positions,
                         const uint64_t* __restrict__ d_new_count,
                         const uint64_t* __restrict__ idx, float* __restrict__ rows, int dim,
                         int mode, float val, uint64_t seed) {
  const uint64_t n_new = *d_new_count;
  const uint64_t total = n_new * (uint64_t)dim;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t row = idx[new_positions[i / dim]];
    const uint32_t e = (uint32_t)(i % dim);
    rows[row * dim + e] = det_init_value(mode, val, seed, row, e);
  }
}

__global__ void __launch_bounds__(kBlock)
    det_gather_kernel(const uint64_t* __restrict__ idx, size_t n, const float* __restrict__ rows,
                      int dim, float* __restrict__ out) {
  const uint64_t total = (uint64_t)n * dim;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t r = idx[i / dim];
    out[i] = r != kInvalidIndex ? rows[r * dim + (i % dim)] : 0.0f;
  }
}

// dim % 4 == 0 and 16-byte aligned buffers: one float4 per thread (the common case)
__global__ void __launch_bounds__(kBlock)
    det_gather4_kernel(const uint64_t* __restrict__ idx, size_t n, const float4* __restrict__ rows,
                       int d4, float4* __restrict__ out) {
  const uint64_t total = (uint64_t)n * d4;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t k = i / (uint32_t)d4;
    const uint32_t c = (uint32_t)(i - k * (uint32_t)d4);
    const uint64_t r = idx[k];
    out[i] = r != kInvalidIndex ? rows[r * (uint32_t)d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

static void launch_det_gather(const uint64_t* idx, size_t n, const float* rows, int dim, float* out,
                              hipStream_t s) {
  const bool v4 = dim % 4 == 0 && reinterpret_cast<uintptr_t>(rows) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(out) % 16 == 0;
  if (v4)
    hipLaunchKernelGGL(det_gather4_kernel, dim3(grid_for(n * (size_t)(dim / 4), kBlock, 8192)),
                       dim3(kBlock), 0, s, idx, n, (const float4*)rows, dim / 4, (float4*)out);
  else
    hipLaunchKernelGGL(det_gather_kernel, dim3(grid_for(n * (size_t)dim, kBlock, 8192)),
                       dim3(kBlock), 0, s, idx, n, rows, dim, out);
}

__global__ void __launch_bounds__(kBlock)
    det_ptr_kernel(const uint64_t* __restrict__ idx, size_t n, float* rows, int dim,
                   float** __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock)
    out[i] = idx[i] != kInvalidIndex ? rows + idx[i] * (uint64_t)dim : nullptr;
}

// ---- one launch over all id spaces of a call (instead of one hash pass per id space) -------------
struct DetClassDesc {
  const HtEntry* tab;
  uint64_t size;
  float* rows;
  uint64_t row_base;
  int dim;
};

// segment (id space) of position i: the last s with seg_off[s] <= i
__device__ __forceinline__ int det_segment_of(const uint64_t* __restrict__ seg_off, int n_seg,
                                              uint64_t i) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg_off[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// find-only probe of every key in its own class; classes that miss a key are flagged so that only
// they go through the inserting path
template <typename K>
__global__ void __launch_bounds__(kBlock)
    det_find_multi_kernel(const DetClassDesc* __restrict__ cls, const uint32_t* __restrict__ seg_class,
                          const uint64_t* __restrict__ seg_off, int n_seg,
                          const K* __restrict__ keys, size_t n, uint64_t* __restrict__ idx,
                          uint32_t* __restrict__ miss, uint64_t* __restrict__ out_row) {
  const long long empty = KeyTraits<K>::empty;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const uint32_t c = seg_class[det_segment_of(seg_off, n_seg, i)];
    const HtEntry* __restrict__ tab = cls[c].tab;
    const uint64_t size = cls[c].size;
    const K key = keys[i];
    const long long k64 = (long long)(sizeof(K) == 4 ? (unsigned long long)(uint32_t)key
                                                     : (unsigned long long)key);
    uint64_t slot = (uint64_t)murmur3_key(key) % size;
    uint64_t res = kInvalidIndex;
    for (uint64_t probes = 0; probes <= size; ++probes) {
      const long long cur = tab[slot].key;
      if (cur == k64) {
        res = tab[slot].val;
        break;
      }
      if (cur == empty) break;
      slot = (slot + 1 == size) ? 0 : slot + 1;
    }
    idx[i] = res;
    if (res == kInvalidIndex) miss[c] = 1u;
    // (the table-wide row number right away: final unless a class has to insert and may grow)
    if (out_row) out_row[i] = res != kInvalidIndex ? cls[c].row_base + res : kInvalidIndex;
  }
}

__global__ void __launch_bounds__(kBlock)
    det_rows_multi_kernel(const DetClassDesc* __restrict__ cls, const uint32_t* __restrict__ seg_class,
                          const uint64_t* __restrict__ seg_off, int n_seg,
                          const uint64_t* __restrict__ idx, size_t n, float** __restrict__ out_ptr,
                          uint64_t* __restrict__ out_row) {
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const DetClassDesc d = cls[seg_class[det_segment_of(seg_off, n_seg, i)]];
    const uint64_t r = idx[i];
    if (out_ptr) out_ptr[i] = r != kInvalidIndex ? d.rows + r * (uint64_t)d.dim : nullptr;
    if (out_row) out_row[i] = r != kInvalidIndex ? d.row_base + r : kInvalidIndex;
  }
}

// dynamic_map_kernels.cuh:143-183: keys that are not in the map are skipped
template <bool ADD>
__global__ void __launch_bounds__(kBlock)
    det_scatter_kernel(const uint64_t* __restrict__ idx, size_t n, float* __restrict__ rows,
                       int dim, const float* __restrict__ upd) {
  const uint64_t total = (uint64_t)n * dim;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t r = idx[i / dim];
    if (r == kInvalidIndex) continue;
    float* dst = rows + r * dim + (i % dim);
    if (ADD) unsafeAtomicAdd(dst, upd[i]);
    else *dst = upd[i];
  }
}

template <typename K>
__global__ void __launch_bounds__(kBlock)
    det_erase_kernel(HtEntry* __restrict__ tab, uint64_t size, const K* __restrict__ keys,
                     size_t n, long long tomb, unsigned long long* __restrict__ d_erased) {
  const long long empty = KeyTraits<K>::empty;
  for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const K key = keys[i];
    const long long k64 = (long long)(sizeof(K) == 4 ? (unsigned long long)(uint32_t)key
                                                     : (unsigned long long)key);
    uint64_t slot = (uint64_t)murmur3_key(key) % size;
    for (uint64_t probes = 0; probes <= size; ++probes) {
      const long long cur = tab[slot].key;
      if (cur == k64) {
        // duplicates of one key in the batch race for the slot; one of them wins the CAS
        const unsigned long long old =
            atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot].key),
                      (unsigned long long)k64, (unsigned long long)tomb);
        if (old == (unsigned long long)k64) {
          tab[slot].val = kInvalidIndex;
          atomicAdd(d_erased, 1ull);
        }
        break;
      }
      if (cur == empty) break;
      slot = (slot + 1 == size) ? 0 : slot + 1;
    }
  }
}

struct DetOpt {
  int optimizer;
  float lr, beta1, beta2, epsilon, momentum, scaler;
  float lr_scaled_bias;               // adam
  float rms_beta;                     // rmsprop
  float lambda1, lambda2_plus_beta_div_lr;  // ftrl
};

// One thread per element of one unique key's vector.  Formulas: optimizers.cuh:29-233 -- the
// reference turns wgrad into the weight delta in place and then scatter_adds it; here the delta is
// applied to the row directly (same arithmetic, one pass).
// one element of one row: state update + weight delta (optimizers.cuh:29-233)
__device__ __forceinline__ void det_apply(const DetOpt& o, int dim, int e, float gi, float* w,
                                          float* st) {
  float delta;
  switch (o.optimizer) {
    case HCTR_OPT_SGD:
      delta = -o.lr * gi;
      break;
    case HCTR_OPT_MOMENTUM_SGD: {
      const float mi = o.momentum * st[e] - o.lr * gi;
      st[e] = mi;
      delta = mi;
    } break;
    case HCTR_OPT_NESTEROV: {
      const float prev = st[e];
      const float mi = o.momentum * prev - o.lr * gi;
      st[e] = mi;
      delta = mi + o.momentum * mi - o.momentum * prev;
    } break;
    case HCTR_OPT_ADAGRAD: {
      const float vi = st[e] + gi * gi;
      st[e] = vi;
      delta = -o.lr * gi / (sqrtf(vi) + o.epsilon);
    } break;
    case HCTR_OPT_RMSPROP: {
      const float vi = o.rms_beta * st[e] + (1.f - o.rms_beta) * gi * gi;
      st[e] = vi;
      delta = -o.lr * gi / (sqrtf(vi) + o.epsilon);
    } break;
    case HCTR_OPT_ADAM: {
      const float mi = o.beta1 * st[e] + (1.f - o.beta1) * gi;
      const float vi = o.beta2 * st[dim + e] + (1.f - o.beta2) * gi * gi;
      st[e] = mi;
      st[dim + e] = vi;
      delta = -o.lr_scaled_bias * mi / (sqrtf(vi) + o.epsilon);
    } break;
    default: {  // HCTR_OPT_FTRL
      float ni = st[e];
      const float ni_prev_sqrt = sqrtf(ni + 1.1920929e-07f);  // FLT_EPSILON
      ni = ni + gi * gi;
      st[e] = ni;
      const float ni_sqrt = sqrtf(ni + 1.1920929e-07f);
      const float sigma = (ni_sqrt - ni_prev_sqrt) / o.lr;
      const float wi = *w;
      const float zi = st[dim + e] + gi - sigma * wi;
      st[dim + e] = zi;
      const float p = (1.f - 2.f * (float)signbit(zi)) * o.lambda1 - zi;
      const float q = ni_sqrt / o.lr + o.lambda2_plus_beta_div_lr;
      delta = (p / q) * (float)signbit(o.lambda1 - fabsf(zi)) - wi;
    } break;
  }
  *w += delta;
}

__global__ void __launch_bounds__(kBlock)
    det_update_kernel(DetOpt o, size_t n, int dim, const uint64_t* __restrict__ idx_w,
                      const uint64_t* __restrict__ idx_s, float* __restrict__ rows_w,
                      float* __restrict__ rows_s, const uint32_t* __restrict__ ev_start,
                      const float* __restrict__ wgrad) {
  const uint64_t total = (uint64_t)n * dim;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const size_t k = (size_t)(i / dim);
    const int e = (int)(i % dim);
    const uint64_t rw = idx_w[k];
    if (rw == kInvalidIndex) continue;  // scatter_add skips keys that are not in the table
    const float gi = wgrad[ev_start[k] + e] / o.scaler;
    const int sdim = dim * (o.optimizer == HCTR_OPT_ADAM || o.optimizer == HCTR_OPT_FTRL ? 2 : 1);
    float* st = (rows_s != nullptr && idx_s != nullptr) ? rows_s + idx_s[k] * (uint64_t)sdim
                                                         : nullptr;
    det_apply(o, dim, e, gi, rows_w + rw * dim + e, st);
  }
}

// the same step through per-key row addresses: one launch for keys of any number of classes
// (dynamic_embedding.cu:227-317: *_update_grad_kernel(ev_start_indices, ..., float** state,
// float** weight))
__global__ void __launch_bounds__(kBlock)
    det_update_ptr_kernel(DetOpt o, size_t n, int dim, float* const* __restrict__ wptr,
                          float* const* __restrict__ sptr, const uint32_t* __restrict__ ev_start,
                          const float* __restrict__ wgrad) {
  const uint64_t total = (uint64_t)n * dim;
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < total;
       i += (uint64_t)gridDim.x * kBlock) {
    const size_t k = (size_t)(i / dim);
    const int e = (int)(i % dim);
    float* w = wptr[k];
    if (w == nullptr) continue;  // scatter_add skips keys that are not in the table
    const float gi = wgrad[ev_start[k] + e] / o.scaler;
    det_apply(o, dim, e, gi, w + e, sptr ? sptr[k] : nullptr);
  }
}

// HCTR_DET_TRACE=1: wall time of the pieces of a growth step, to stderr
struct GrowTrace {
  bool on;
  double t0;
  const char* what;
  static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  explicit GrowTrace(const char* w) : what(w) {
    static const bool e = [] { const char* v = getenv("HCTR_DET_TRACE"); return v && v[0] == '1'; }();
    on = e;
    t0 = on ? now() : 0.0;
  }
  void lap(const char* piece) {
    if (!on) return;
    const double t = now();
    fprintf(stderr, "[det] %s: %s %.3f ms\n", what, piece, (t - t0) * 1e3);
    t0 = t;
  }
};

// ---- growing arrays --------------------------------------------------------------------------
// A table's rows (and, for the flat row store's users, its optimizer state) live in ONE reserved
// address range per array, class c in the region [c * stride_rows rows, (c + 1) * stride_rows rows);
// physical memory is mapped behind the rows a class holds in pieces of one size as the class
// grows (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess).  A row never moves:
// growth copies nothing, frees nothing, allocates the increment only.  Why not hipMalloc + copy +
// hipFree (rounds 3-5): a single hipMalloc of 8-20 GB on a device that has seen some traffic was
// measured at 1.6 - 3.3 s every few calls (profiles/r6_vmm_probe.txt, r6_dyn_growth_trace.txt),
// next to 3 us + 20 us - 3 ms for creating and mapping a piece.  Measured rules of this runtime
// (ROCm 7.2, tools/vmm_probe.hip): the pieces of a range must all have ONE size (hipMemSetAccess
// fails with "invalid argument" at the first piece larger than its predecessors), so the piece
// size is a process-wide constant; reserving, freeing and reserving again is fine.
inline size_t vm_piece_bytes() {
  static const size_t v = [] {
    const char* e = getenv("HCTR_DET_PIECE_MB");  // (a power of two; experiments only)
    size_t mb = e ? (size_t)atol(e) : 64;
    if (mb < 2) mb = 2;
    size_t p = 2;
    while (p < mb) p *= 2;
    return p << 20;
  }();
  return v;
}

struct VmArray {
  char* base = nullptr;
  size_t va_bytes = 0;
  struct Piece {
    size_t off;
    hipMemGenericAllocationHandle_t h;
  };
  std::vector<Piece> pieces;
};

int vm_reserve(VmArray& a, size_t bytes) {
  hipDeviceptr_t p = nullptr;
  if (hipMemAddressReserve(&p, bytes, 0, nullptr, 0) != hipSuccess || p == nullptr) {
    (void)hipGetLastError();
    set_error("dynamic table: hipMemAddressReserve of " + std::to_string(bytes >> 30) + " GiB failed");
    return HCTR_ERR_HIP;
  }
  a.base = (char*)p;
  a.va_bytes = bytes;
  return HCTR_OK;
}

// physical pieces behind [off, off + bytes) (both multiples of the piece size, nothing mapped there yet)
int vm_map(VmArray& a, size_t off, size_t bytes, bool zero, hipStream_t s) {
  const size_t piece = vm_piece_bytes();
  int dev = 0;
  HCTR_HIP(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t o = off; o < off + bytes; o += piece) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      set_error("dynamic table: out of device memory (hipMemCreate of a " +
                std::to_string(piece >> 20) + " MiB piece)");
      return HCTR_ERR_HIP;
    }
    if (hipMemMap(a.base + o, piece, 0, h, 0) != hipSuccess ||
        hipMemSetAccess(a.base + o, piece, &acc, 1) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipMemRelease(h);
      set_error("dynamic table: hipMemMap / hipMemSetAccess failed");
      return HCTR_ERR_HIP;
    }
    a.pieces.push_back({o, h});
  }
  if (zero && bytes) HCTR_HIP(hipMemsetAsync(a.base + off, 0, bytes, s));
  return HCTR_OK;
}

void vm_release(VmArray& a) {
  const size_t piece = vm_piece_bytes();
  for (const VmArray::Piece& p : a.pieces) {
    (void)hipMemUnmap(a.base + p.off, piece);
    (void)hipMemRelease(p.h);
  }
  a.pieces.clear();
  if (a.base) (void)hipMemAddressFree(a.base, a.va_bytes);
  a.base = nullptr;
  a.va_bytes = 0;
}

struct DetClass {
  HashTable ht;
  float* rows = nullptr;  // the class's region of hctr_det::rows_va (fixed for the table's life)
  size_t mapped = 0;      // bytes of the region that have memory behind them (a multiple of the piece)
  size_t cap = 0;         // rows the class may hand out == ht.capacity; cap * row bytes <= mapped
  int dim = 0;
  size_t head_bound = 0;  // host upper bound of the row counter (value head)
  unsigned long long* d_erased = nullptr;
};

}  // namespace
}  // namespace hctr

using namespace hctr;

struct hctr_det {
  std::vector<DetClass> cls;
  int key_type = HCTR_KEY_I64;
  int init_mode = 1;  // 0 constant, 1 uniform (0, 1]
  float init_val = 0.f;
  uint64_t seed = 0;
  uint64_t adam_times = 0;
  // Row store: class c owns rows [c * stride_rows, c * stride_rows + cap_c) of rows_va (VmArray
  // above).  Classes of ONE dimension (an embedding_collection group always: one ev_size): the
  // table-wide row numbers that hctr_det_lookup_rows hands out are then rows of one flat
  // [arena_rows = classes * stride_rows][dim] table (sparsely backed), and the static tables'
  // gather and sparse-update kernels run on a dynamic table as they are (hctr_det_row_store).
  VmArray rows_va;
  size_t stride_rows = 0;             // a power of two; classes * stride_rows < 2^32 - 16
  std::vector<size_t> region_off;     // byte offset of every class's region in rows_va (+ total)
  float* arena = nullptr;             // rows_va.base when the classes share a dimension, else null
  size_t arena_rows = 0;
  // optimizer state of the flat row store's users (hctr_det_state_store): n_state arrays laid out
  // and backed like the rows, zero for a row nobody updated yet -- what the reference's state table
  // ("zeros" initializer, one entry per updated key, dynamic_embedding.cu:227-317) holds for a key,
  // at the key's weight row: the state needs no probe of its own.
  VmArray state_va[2];
  float* state_arena[2] = {nullptr, nullptr};
  int n_state = 0;
  uint64_t* idx = nullptr;  // scratch row indices
  uint64_t* idx2 = nullptr;
  size_t idx_cap = 0;
  // device copies of the per-call id-space table and the class descriptors (multi-space lookups)
  float** ptr_w = nullptr;  // per-key row addresses of a multi-class update
  float** ptr_s = nullptr;
  size_t ptr_cap = 0;
  DetClassDesc* d_desc = nullptr;
  uint32_t* d_seg_class = nullptr;
  uint64_t* d_seg_off = nullptr;
  uint32_t* d_miss = nullptr;
  size_t seg_cap = 0;
  // host copies of what d_desc / d_seg_class / d_seg_off hold (det_upload_spaces)
  std::vector<unsigned char> up_desc;
  std::vector<uint32_t> up_seg_class;
  std::vector<uint64_t> up_seg_off;
  hipStream_t up_stream = nullptr;  // the stream those copies were queued on
  // pinned host word per class: the error flags of the class's hash index as its last inserting
  // lookup left them (posted by the finish kernel, IndexExtras::host_error)
  uint32_t* h_err = nullptr;
  uint64_t repairs = 0;  // inserting lookups that were repaired and issued again
};

namespace {

int det_scratch(hctr_det* h, size_t n) {
  if (n <= h->idx_cap) return HCTR_OK;
  if (h->idx) (void)hipFree(h->idx);
  if (h->idx2) (void)hipFree(h->idx2);
  h->idx = h->idx2 = nullptr;
  size_t c = h->idx_cap ? h->idx_cap : 1024;
  while (c < n) c *= 2;
  HCTR_HIP(hipMalloc(&h->idx, c * sizeof(uint64_t)));
  HCTR_HIP(hipMalloc(&h->idx2, c * sizeof(uint64_t)));
  h->idx_cap = c;
  return HCTR_OK;
}

int class_create(DetClass& c, size_t cap, int dim, int key_type) {
  c.dim = dim;
  c.cap = cap;
  HCTR_TRY(c.ht.create(cap, key_type));
  HCTR_HIP(hipMalloc(&c.d_erased, sizeof(unsigned long long)));
  HCTR_HIP(hipMemset(c.d_erased, 0, sizeof(unsigned long long)));
  c.head_bound = 0;
  return HCTR_OK;
}

void class_destroy(DetClass& c) {
  c.ht.destroy();
  if (c.d_erased) (void)hipFree(c.d_erased);
  c.rows = nullptr;
  c.d_erased = nullptr;
}

// memory behind rows [0, rows) of class ci, in every array the table keeps (state: zero-filled)
int class_map(hctr_det* h, size_t ci, size_t rows, hipStream_t s) {
  DetClass& c = h->cls[ci];
  if (rows > h->stride_rows) {
    set_error("dynamic table: class " + std::to_string(ci) + " needs " + std::to_string(rows) +
              " rows, its address range holds " + std::to_string(h->stride_rows) +
              " (2^32 row numbers shared by " + std::to_string(h->cls.size()) + " classes)");
    return HCTR_ERR_INVALID_ARG;
  }
  const size_t piece = vm_piece_bytes();
  const size_t need = ceil_div<size_t>(rows * (size_t)c.dim * sizeof(float), piece) * piece;
  if (need <= c.mapped) return HCTR_OK;
  HCTR_TRY(vm_map(h->rows_va, h->region_off[ci] + c.mapped, need - c.mapped, false, s));
  for (int k = 0; k < h->n_state; k++)
    HCTR_TRY(vm_map(h->state_va[k], h->region_off[ci] + c.mapped, need - c.mapped, true, s));
  c.mapped = need;
  return HCTR_OK;
}

// make room for n more rows (cuco::dynamic_map::reserve): the index is re-built at twice the
// capacity (tombstones are dropped); the rows stay where they are and get more memory behind them
int class_reserve(hctr_det* h, DetClass& c, size_t n, int key_type, hipStream_t s) {
  if (c.head_bound + n <= c.cap) return HCTR_OK;
  size_t head = 0;
  GrowTrace tr("class_reserve");
  HCTR_TRY(c.ht.value_head(s, &head));  // synchronises: the exact row counter
  c.head_bound = head;
  tr.lap("value_head (stream drained)");
  if (head + n <= c.cap) return HCTR_OK;
  size_t ncap = c.cap * 2;
  while (ncap < head + n) ncap *= 2;
  if (ncap > h->stride_rows && head + n <= h->stride_rows) ncap = h->stride_rows;
  const size_t ci = (size_t)(&c - h->cls.data());
  HCTR_TRY(class_map(h, ci, ncap, s));
  tr.lap("pieces mapped");
  // live (key, row) pairs of the old index
  int64_t* d_keys = nullptr;
  uint64_t* d_vals = nullptr;
  const size_t slots = (size_t)c.ht.size;
  HCTR_HIP(hipMalloc(&d_keys, slots * sizeof(int64_t)));
  HCTR_HIP(hipMalloc(&d_vals, slots * sizeof(uint64_t)));
  size_t live = 0;
  HCTR_TRY(c.ht.dump(d_keys, d_vals, &live, s));
  HashTable nht;
  HCTR_TRY(nht.create(ncap, key_type));
  if (live > 0) {
    if (key_type == HCTR_KEY_U32) {
      // dump widens keys to int64; the u32 insert path wants 32-bit keys
      std::vector<int64_t> hk(live);
      HCTR_HIP(hipMemcpy(hk.data(), d_keys, live * sizeof(int64_t), hipMemcpyDeviceToHost));
      std::vector<uint32_t> hk32(live);
      for (size_t i = 0; i < live; i++) hk32[i] = (uint32_t)hk[i];
      HCTR_HIP(hipMemcpy(d_keys, hk32.data(), live * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    HCTR_TRY(nht.insert(d_keys, d_vals, live, s));
  }
  HCTR_TRY(nht.set_value_head(head, s));
  tr.lap("re-index");
  (void)hipFree(d_keys);
  (void)hipFree(d_vals);
  c.ht.destroy();
  c.ht = nht;
  c.cap = ncap;
  tr.lap("old index freed");
  // rows of erased keys stay allocated (indices are never reused); only the index forgets them
  return HCTR_OK;
}

struct Range {
  size_t cls, off, n;
};

int ranges_of(const hctr_det* h, size_t num_keys, const size_t* id_spaces,
              const size_t* id_space_offsets, size_t num_id_spaces, std::vector<Range>* out) {
  HCTR_REQUIRE(num_id_spaces == 0 || (id_spaces && id_space_offsets), "null id_space arrays");
  for (size_t i = 0; i < num_id_spaces; i++) {
    HCTR_REQUIRE(id_spaces[i] < h->cls.size(), "id_space out of range");
    HCTR_REQUIRE(id_space_offsets[i] <= id_space_offsets[i + 1] &&
                     id_space_offsets[i + 1] <= num_keys,
                 "id_space_offsets must be ascending and end at num_keys");
    out->push_back({id_spaces[i], id_space_offsets[i], id_space_offsets[i + 1] - id_space_offsets[i]});
  }
  return HCTR_OK;
}

inline const void* key_at(const hctr_det* h, const void* keys, size_t off) {
  return (const char*)keys + off * (h->key_type == HCTR_KEY_U32 ? 4 : 8);
}

// an inserting lookup that was queued and is not verified yet (det_verify_inserts)
struct Inserted {
  size_t cls;
  const void* keys;
  size_t n;
  uint64_t* idx;
};

// lookup with insertion of unseen keys; leaves row indices in idx[0..n).  `log`: the call is noted
// for det_verify_inserts, which every caller runs before it reads idx.
int class_lookup_insert(hctr_det* h, DetClass& c, size_t cls_index, const void* keys, size_t n,
                        uint64_t* idx, hipStream_t s, std::vector<Inserted>* log,
                        bool two_launches = false) {
  HCTR_TRY(class_reserve(h, c, n, h->key_type, s));
  IndexExtras ex;
  ex.host_error = h->h_err + cls_index;
  ex.two_launches = two_launches;
  HCTR_TRY(c.ht.get_insert(keys, n, nullptr, idx, s, nullptr, &ex));
  if (log != nullptr) log->push_back({cls_index, keys, n, idx});
  c.head_bound += n;
  hipLaunchKernelGGL(det_init_rows_kernel, dim3(grid_for(n * (size_t)c.dim, kBlock, 2048)),
                     dim3(kBlock), 0, s, c.ht.new_positions, c.ht.d_new_count, idx, c.rows, c.dim,
                     h->init_mode, h->init_val, h->seed + 0x9E3779B97F4A7C15ull * (cls_index + 1));
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// The hash index reports through its error word, never through the row it hands out: a finish
// kernel whose grid barrier did not open leaves "no row" at every unseen key of its batch (they
// would pool as zeros and be skipped by the update, silently).  Every inserting lookup is
// therefore verified before its rows are used: one wait for the stream, the pinned error words of
// the classes that inserted.  A barrier that gave up (bit 4: all-or-nothing, hashtable.hip) is
// repaired and the class's lookups are queued again in their order with the two-launch finish
// (no barrier) -- the rows are the ones an undisturbed run hands out (the row counter had not
// moved).  Anything else (a full index: a bug in class_reserve's accounting) is an error.
int det_verify_inserts(hctr_det* h, std::vector<Inserted>& ins, hipStream_t s) {
  if (ins.empty()) return HCTR_OK;
  HCTR_HIP(hipStreamSynchronize(s));
  const volatile uint32_t* err = h->h_err;
  std::vector<char> redo(h->cls.size(), 0);
  bool any = false;
  for (const Inserted& it : ins) {
    const uint32_t e = err[it.cls];
    if (e & ~4u) {
      set_error("dynamic table: hash index of class " + std::to_string(it.cls) +
                " reports error flags " + std::to_string(e));
      return HCTR_ERR_HIP;
    }
    if (e & 4u) redo[it.cls] = 1, any = true;
  }
  if (!any) {
    ins.clear();
    return HCTR_OK;
  }
  for (const Inserted& it : ins)
    if (redo[it.cls]) HCTR_TRY(h->cls[it.cls].ht.recover(it.keys, it.n, s));
  for (const Inserted& it : ins)
    if (redo[it.cls]) {
      DetClass& c = h->cls[it.cls];
      c.head_bound -= it.n < c.head_bound ? it.n : c.head_bound;  // (counted again below)
      h->repairs++;
      HCTR_TRY(class_lookup_insert(h, c, it.cls, it.keys, it.n, it.idx, s, nullptr, true));
    }
  HCTR_HIP(hipStreamSynchronize(s));
  for (size_t ci = 0; ci < redo.size(); ci++)
    if (redo[ci] && err[ci] != 0u) {
      set_error("dynamic table: hash index of class " + std::to_string(ci) +
                " could not be repaired after a grid barrier that did not open (error flags " +
                std::to_string(err[ci]) + ")");
      return HCTR_ERR_HIP;
    }
  ins.clear();
  return HCTR_OK;
}

}  // namespace

extern "C" {

int hctr_det_create(size_t num_classes, const size_t* dimension_per_class, const char* initializer,
                    size_t initial_capacity_per_class, int key_type, uint64_t seed,
                    hctr_det** out) {
  HCTR_REQUIRE(out && dimension_per_class && num_classes > 0, "null pointer / no classes");
  HCTR_REQUIRE(key_type == HCTR_KEY_U32 || key_type == HCTR_KEY_I64, "key_type");
  hctr_det* h = new hctr_det();
  h->key_type = key_type;
  h->seed = seed;
  // dynamic_embedding_table.cu:66-84: "ones", "zeros", a float literal, anything else = random
  const std::string ini = initializer ? initializer : "";
  h->init_mode = 1;
  if (ini == "ones") {
    h->init_mode = 0;
    h->init_val = 1.0f;
  } else if (ini == "zeros") {
    h->init_mode = 0;
    h->init_val = 0.0f;
  } else if (!ini.empty()) {
    char* end = nullptr;
    const float v = strtof(ini.c_str(), &end);
    if (end != ini.c_str()) {
      h->init_mode = 0;
      h->init_val = v;
    }
  }
  const size_t cap = initial_capacity_per_class ? initial_capacity_per_class : 1048576;
  if (hipHostMalloc(&h->h_err, num_classes * sizeof(uint32_t)) != hipSuccess) {
    (void)hipGetLastError();
    delete h;
    set_error("hipHostMalloc (error words)");
    return HCTR_ERR_HIP;
  }
  memset(h->h_err, 0, num_classes * sizeof(uint32_t));
  h->cls.resize(num_classes);
  bool flat = true;  // one dimension: the row numbers index one flat table (hctr_det::arena)
  for (size_t i = 1; i < num_classes; i++) flat = flat && dimension_per_class[i] == dimension_per_class[0];
  auto fail = [&](int rc) {
    for (auto& c : h->cls) class_destroy(c);
    vm_release(h->rows_va);
    (void)hipHostFree(h->h_err);
    delete h;
    return rc;
  };
  {
    int vmm = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev) != hipSuccess ||
        vmm == 0) {
      (void)hipGetLastError();
      set_error("dynamic table: the device does not support virtual memory management");
      return fail(HCTR_ERR_HIP);
    }
  }
  size_t max_row_bytes = 0;
  for (size_t i = 0; i < num_classes; i++) {
    if (dimension_per_class[i] == 0 || dimension_per_class[i] > (1u << 14)) {
      set_error("dimension_per_class out of range (1 .. 16384)");
      return fail(HCTR_ERR_INVALID_ARG);
    }
    max_row_bytes = std::max(max_row_bytes, dimension_per_class[i] * sizeof(float));
  }
  // row numbers are 32-bit for the sparse update's sort: classes * stride_rows < 2^32 - 16; a
  // region of at most 1 TiB, all regions of an array at most 16 TiB of addresses; a region is a
  // whole number of pieces (stride_rows >= 2^24 rows of >= 4 bytes)
  size_t stride = (size_t)1 << 28;
  while (stride * num_classes > 0xFFF00000ull) stride >>= 1;
  while (stride * max_row_bytes > ((size_t)1 << 40)) stride >>= 1;
  while (stride > ((size_t)1 << 24) && stride * max_row_bytes * num_classes > ((size_t)16 << 40)) stride >>= 1;
  if (stride < ((size_t)1 << 24) || stride * max_row_bytes * num_classes > ((size_t)16 << 40) ||
      (stride * sizeof(float)) % vm_piece_bytes() != 0) {
    set_error("dynamic table: too many classes for one table (at most 255, fewer for very long vectors)");
    return fail(HCTR_ERR_INVALID_ARG);
  }
  if (cap > stride) {
    set_error("dynamic table: initial capacity above the " + std::to_string(stride) +
              " rows a class of this table can hold");
    return fail(HCTR_ERR_INVALID_ARG);
  }
  h->stride_rows = stride;
  h->region_off.assign(num_classes + 1, 0);
  for (size_t i = 0; i < num_classes; i++)
    h->region_off[i + 1] = h->region_off[i] + stride * dimension_per_class[i] * sizeof(float);
  {
    const int rc = vm_reserve(h->rows_va, h->region_off[num_classes]);
    if (rc != HCTR_OK) return fail(rc);
  }
  for (size_t i = 0; i < num_classes; i++) {
    int rc = class_create(h->cls[i], cap, (int)dimension_per_class[i], key_type);
    h->cls[i].rows = reinterpret_cast<float*>(h->rows_va.base + h->region_off[i]);
    if (rc == HCTR_OK) rc = class_map(h, i, cap, nullptr);
    if (rc != HCTR_OK) return fail(rc);
  }
  if (flat) {
    h->arena = reinterpret_cast<float*>(h->rows_va.base);
    h->arena_rows = stride * num_classes;
  }
  (void)hipDeviceSynchronize();
  *out = h;
  return HCTR_OK;
}

int hctr_det_destroy(hctr_det* h) {
  if (!h) return HCTR_OK;
  (void)hipDeviceSynchronize();
  for (auto& c : h->cls) class_destroy(c);
  vm_release(h->rows_va);
  for (VmArray& a : h->state_va) vm_release(a);
  if (h->idx) (void)hipFree(h->idx);
  if (h->idx2) (void)hipFree(h->idx2);
  if (h->ptr_w) (void)hipFree(h->ptr_w);
  if (h->ptr_s) (void)hipFree(h->ptr_s);
  if (h->d_desc) (void)hipFree(h->d_desc);
  if (h->d_seg_class) (void)hipFree(h->d_seg_class);
  if (h->d_seg_off) (void)hipFree(h->d_seg_off);
  if (h->d_miss) (void)hipFree(h->d_miss);
  if (h->h_err) (void)hipHostFree(h->h_err);
  delete h;
  return HCTR_OK;
}

int hctr_det_lookup(hctr_det* h, const void* keys, float* elements, size_t num_keys,
                    const size_t* id_spaces, const size_t* id_space_offsets, size_t num_id_spaces,
                    hctr_stream_t stream) {
  HCTR_REQUIRE(h, "null handle");
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && elements, "null pointer");
  hipStream_t s = as_stream(stream);
  std::vector<Range> rs;
  HCTR_TRY(ranges_of(h, num_keys, id_spaces, id_space_offsets, num_id_spaces, &rs));
  HCTR_TRY(det_scratch(h, num_keys));
  // every insertion (and the check that it handed out rows) before a row is read
  std::vector<Inserted> ins;
  for (const Range& r : rs)
    if (r.n)
      HCTR_TRY(class_lookup_insert(h, h->cls[r.cls], r.cls, key_at(h, keys, r.off), r.n,
                                   h->idx + r.off, s, &ins));
  HCTR_TRY(det_verify_inserts(h, ins, s));
  size_t out_off = 0;
  for (const Range& r : rs) {
    DetClass& c = h->cls[r.cls];
    if (r.n == 0) continue;
    launch_det_gather(h->idx + r.off, r.n, c.rows, c.dim, elements + out_off, s);
    HCTR_LAUNCH_CHECK();
    out_off += r.n * (size_t)c.dim;
  }
  return HCTR_OK;
}

int hctr_det_lookup_unsafe(hctr_det* h, const void* keys, float** elements, size_t num_keys,
                           const size_t* id_spaces, const size_t* id_space_offsets,
                           size_t num_id_spaces, hctr_stream_t stream) {
  HCTR_REQUIRE(h, "null handle");
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && elements, "null pointer");
  hipStream_t s = as_stream(stream);
  std::vector<Range> rs;
  HCTR_TRY(ranges_of(h, num_keys, id_spaces, id_space_offsets, num_id_spaces, &rs));
  HCTR_TRY(det_scratch(h, num_keys));
  // two passes: every insertion (and so every re-allocation) happens before a pointer is taken
  // (a class may own several ranges: reserve their sum, or a later range could move the store)
  std::vector<size_t> need(h->cls.size(), 0);
  for (const Range& r : rs) need[r.cls] += r.n;
  for (size_t ci = 0; ci < need.size(); ci++)
    if (need[ci]) HCTR_TRY(class_reserve(h, h->cls[ci], need[ci], h->key_type, s));
  std::vector<Inserted> ins;
  for (const Range& r : rs)
    if (r.n)
      HCTR_TRY(class_lookup_insert(h, h->cls[r.cls], r.cls, key_at(h, keys, r.off), r.n,
                                   h->idx + r.off, s, &ins));
  HCTR_TRY(det_verify_inserts(h, ins, s));
  for (const Range& r : rs) {
    DetClass& c = h->cls[r.cls];
    if (r.n == 0) continue;
    hipLaunchKernelGGL(det_ptr_kernel, dim3(grid_for(r.n, kBlock, 1024)), dim3(kBlock), 0, s,
                       h->idx + r.off, r.n, c.rows, c.dim, elements + r.off);
    HCTR_LAUNCH_CHECK();
  }
  return HCTR_OK;
}

// upload the id-space table of a call and the current class descriptors
static int det_upload_spaces(hctr_det* h, const std::vector<Range>& rs,
                             const std::vector<uint64_t>& base, hipStream_t s) {
  const size_t ncls = h->cls.size();
  if (!h->d_desc) {
    HCTR_HIP(hipMalloc(&h->d_desc, ncls * sizeof(DetClassDesc)));
    HCTR_HIP(hipMalloc(&h->d_miss, ncls * sizeof(uint32_t)));
  }
  if (rs.size() > h->seg_cap) {
    if (h->d_seg_class) (void)hipFree(h->d_seg_class);
    if (h->d_seg_off) (void)hipFree(h->d_seg_off);
    h->d_seg_class = nullptr;
    h->d_seg_off = nullptr;
    size_t c = 64;
    while (c < rs.size()) c *= 2;
    HCTR_HIP(hipMalloc(&h->d_seg_class, c * sizeof(uint32_t)));
    HCTR_HIP(hipMalloc(&h->d_seg_off, c * sizeof(uint64_t)));
    h->seg_cap = c;
    h->up_seg_class.clear();
    h->up_seg_off.clear();
  }
  std::vector<DetClassDesc> desc(ncls);
  memset((void*)desc.data(), 0, ncls * sizeof(DetClassDesc));
  for (size_t ci = 0; ci < ncls; ci++) {
    const DetClass& c = h->cls[ci];
    desc[ci].tab = c.ht.entries;
    desc[ci].size = c.ht.size;
    desc[ci].rows = c.rows;
    desc[ci].row_base = base[ci];
    desc[ci].dim = c.dim;
  }
  std::vector<uint32_t> sc(rs.size());
  std::vector<uint64_t> so(rs.size());
  for (size_t i = 0; i < rs.size(); i++) {
    sc[i] = (uint32_t)rs[i].cls;
    so[i] = rs[i].off;
  }
  // what the device holds already stays (a training loop repeats the same id spaces over tables
  // that stopped growing: three small copies less in front of every lookup)
  auto same = [](const auto& a, const auto& b) {
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0);
  };
  // (the shadows say what the device holds in the order of ONE stream: a call on another stream
  //  could run ahead of the copies still queued on the first -- it uploads again)
  if (h->up_stream != s) {
    h->up_desc.clear();
    h->up_seg_class.clear();
    h->up_seg_off.clear();
    h->up_stream = s;
  }
  std::vector<unsigned char> dbytes(ncls * sizeof(DetClassDesc));
  memcpy(dbytes.data(), desc.data(), dbytes.size());  // (padding bytes of the struct: value-initialised above)
  // (pageable sources: the runtime has consumed them when these calls return)
  if (!same(dbytes, h->up_desc)) {
    HCTR_HIP(hipMemcpyAsync(h->d_desc, desc.data(), ncls * sizeof(DetClassDesc),
                            hipMemcpyHostToDevice, s));
    h->up_desc = dbytes;
  }
  if (!same(sc, h->up_seg_class)) {
    HCTR_HIP(hipMemcpyAsync(h->d_seg_class, sc.data(), sc.size() * sizeof(uint32_t),
                            hipMemcpyHostToDevice, s));
    h->up_seg_class = sc;
  }
  if (!same(so, h->up_seg_off)) {
    HCTR_HIP(hipMemcpyAsync(h->d_seg_off, so.data(), so.size() * sizeof(uint64_t),
                            hipMemcpyHostToDevice, s));
    h->up_seg_off = so;
  }
  return HCTR_OK;
}

int hctr_det_lookup_rows(hctr_det* h, const void* keys, size_t num_keys, const size_t* id_spaces,
                         const size_t* id_space_offsets, size_t num_id_spaces, int insert,
                         float** elements, uint64_t* row_index, uint64_t* class_row_base,
                         hctr_stream_t stream) {
  HCTR_REQUIRE(h, "null handle");
  hipStream_t s = as_stream(stream);
  std::vector<Range> all, rs;
  HCTR_TRY(ranges_of(h, num_keys, id_spaces, id_space_offsets, num_id_spaces, &all));
  for (const Range& r : all)
    if (r.n) rs.push_back(r);
  if (num_keys) {
    HCTR_REQUIRE(keys && (elements || row_index), "null pointer");
    HCTR_REQUIRE(!rs.empty() && rs.front().off == 0 &&
                     rs.back().off + rs.back().n == num_keys,
                 "id spaces must cover keys[0, num_keys)");
    for (size_t i = 1; i < rs.size(); i++)
      HCTR_REQUIRE(rs[i].off == rs[i - 1].off + rs[i - 1].n, "id spaces must be contiguous");
    HCTR_TRY(det_scratch(h, num_keys));
  }
  const size_t ncls = h->cls.size();
  std::vector<uint64_t> base(ncls + 1, 0);
  auto rebase = [&] {  // (fixed for the table's life: a class's region never moves)
    for (size_t ci = 0; ci < ncls; ci++) base[ci + 1] = base[ci] + h->stride_rows;
  };
  rebase();
  if (num_keys) {
    // 1. one find-only pass over all id spaces; classes that met an unseen key are flagged
    HCTR_TRY(det_upload_spaces(h, rs, base, s));
    HCTR_HIP(hipMemsetAsync(h->d_miss, 0, ncls * sizeof(uint32_t), s));
    const int grid = grid_for(num_keys, kBlock, 8192);
    // row numbers only (the flat row store's callers): the probe writes them itself, and the
    // address / row pass below runs only when a class inserted (its rows, and after a growth every
    // class's base, are then different ones)
    uint64_t* early_rows = elements == nullptr ? row_index : nullptr;
    bool rows_done = early_rows != nullptr;
    if (h->key_type == HCTR_KEY_U32)
      hipLaunchKernelGGL(det_find_multi_kernel<uint32_t>, dim3(grid), dim3(kBlock), 0, s, h->d_desc,
                         h->d_seg_class, h->d_seg_off, (int)rs.size(), (const uint32_t*)keys,
                         num_keys, h->idx, h->d_miss, early_rows);
    else
      hipLaunchKernelGGL(det_find_multi_kernel<long long>, dim3(grid), dim3(kBlock), 0, s, h->d_desc,
                         h->d_seg_class, h->d_seg_off, (int)rs.size(), (const long long*)keys,
                         num_keys, h->idx, h->d_miss, early_rows);
    HCTR_LAUNCH_CHECK();
    if (insert) {
      // 2. only those classes take the inserting path (and may grow: every insertion happens
      //    before a pointer or a row base is taken)
      std::vector<uint32_t> miss(ncls, 0);
      HCTR_HIP(hipMemcpyAsync(miss.data(), h->d_miss, ncls * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, s));
      HCTR_HIP(hipStreamSynchronize(s));
      std::vector<size_t> need(ncls, 0);
      bool any = false;
      for (const Range& r : rs)
        if (miss[r.cls]) {
          need[r.cls] += r.n;
          any = true;
        }
      if (any) {
        for (size_t ci = 0; ci < ncls; ci++)
          if (need[ci]) HCTR_TRY(class_reserve(h, h->cls[ci], need[ci], h->key_type, s));
        std::vector<Inserted> ins;
        for (const Range& r : rs)
          if (miss[r.cls])
            HCTR_TRY(class_lookup_insert(h, h->cls[r.cls], r.cls, key_at(h, keys, r.off), r.n,
                                         h->idx + r.off, s, &ins));
        HCTR_TRY(det_verify_inserts(h, ins, s));
        rebase();
        HCTR_TRY(det_upload_spaces(h, rs, base, s));
        rows_done = false;
      }
    }
    // 3. addresses / table-wide row numbers of all keys
    if (!rows_done) {
      hipLaunchKernelGGL(det_rows_multi_kernel, dim3(grid_for(num_keys, kBlock, 8192)), dim3(kBlock),
                         0, s, h->d_desc, h->d_seg_class, h->d_seg_off, (int)rs.size(), h->idx,
                         num_keys, elements, row_index);
      HCTR_LAUNCH_CHECK();
    }
  }
  if (class_row_base)
    for (size_t ci = 0; ci <= ncls; ci++) class_row_base[ci] = base[ci];
  return HCTR_OK;
}

static int det_scatter(hctr_det* h, const void* keys, const float* elements, size_t num_keys,
                       const size_t* id_spaces, const size_t* id_space_offsets,
                       size_t num_id_spaces, bool add, hipStream_t s) {
  HCTR_REQUIRE(h, "null handle");
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && elements, "null pointer");
  std::vector<Range> rs;
  HCTR_TRY(ranges_of(h, num_keys, id_spaces, id_space_offsets, num_id_spaces, &rs));
  HCTR_TRY(det_scratch(h, num_keys));
  size_t off = 0;
  for (const Range& r : rs) {
    DetClass& c = h->cls[r.cls];
    if (r.n == 0) continue;
    HCTR_TRY(c.ht.get_mark(key_at(h, keys, r.off), r.n, nullptr, h->idx, s));
    const dim3 grid(grid_for(r.n * (size_t)c.dim, kBlock, 4096));
    if (add)
      hipLaunchKernelGGL(det_scatter_kernel<true>, grid, dim3(kBlock), 0, s, h->idx, r.n, c.rows,
                         c.dim, elements + off);
    else
      hipLaunchKernelGGL(det_scatter_kernel<false>, grid, dim3(kBlock), 0, s, h->idx, r.n, c.rows,
                         c.dim, elements + off);
    HCTR_LAUNCH_CHECK();
    off += r.n * (size_t)c.dim;
  }
  return HCTR_OK;
}

int hctr_det_scatter_add(hctr_det* h, const void* keys, const float* elements, size_t num_keys,
                         const size_t* id_spaces, const size_t* id_space_offsets,
                         size_t num_id_spaces, hctr_stream_t stream) {
  return det_scatter(h, keys, elements, num_keys, id_spaces, id_space_offsets, num_id_spaces, true,
                     as_stream(stream));
}

int hctr_det_scatter_update(hctr_det* h, const void* keys, const float* elements, size_t num_keys,
                            const size_t* id_spaces, const size_t* id_space_offsets,
                            size_t num_id_spaces, hctr_stream_t stream) {
  return det_scatter(h, keys, elements, num_keys, id_spaces, id_space_offsets, num_id_spaces,
                     false, as_stream(stream));
}

int hctr_det_remove(hctr_det* h, const void* keys, size_t num_keys, const size_t* id_spaces,
                    const size_t* id_space_offsets, size_t num_id_spaces, hctr_stream_t stream) {
  HCTR_REQUIRE(h, "null handle");
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys, "null pointer");
  hipStream_t s = as_stream(stream);
  std::vector<Range> rs;
  HCTR_TRY(ranges_of(h, num_keys, id_spaces, id_space_offsets, num_id_spaces, &rs));
  for (const Range& r : rs) {
    DetClass& c = h->cls[r.cls];
    if (r.n == 0) continue;
    const dim3 grid(grid_for(r.n, kBlock, 2048));
    if (h->key_type == HCTR_KEY_U32)
      hipLaunchKernelGGL(det_erase_kernel<uint32_t>, grid, dim3(kBlock), 0, s, c.ht.entries,
                         c.ht.size, (const uint32_t*)key_at(h, keys, r.off), r.n,
                         KeyTraits<uint32_t>::empty - 1, c.d_erased);
    else
      hipLaunchKernelGGL(det_erase_kernel<long long>, grid, dim3(kBlock), 0, s, c.ht.entries,
                         c.ht.size, (const long long*)key_at(h, keys, r.off), r.n,
                         KeyTraits<long long>::empty - 1, c.d_erased);
    HCTR_LAUNCH_CHECK();
  }
  return HCTR_OK;
}

int hctr_det_export(hctr_det* h, size_t class_index, void* keys, float* values, size_t num_keys,
                    size_t* exported, hctr_stream_t stream) {
  HCTR_REQUIRE(h && class_index < h->cls.size(), "class_index");
  hipStream_t s = as_stream(stream);
  DetClass& c = h->cls[class_index];
  int64_t* d_keys = nullptr;
  uint64_t* d_vals = nullptr;
  const size_t slots = (size_t)c.ht.size;
  HCTR_HIP(hipMalloc(&d_keys, slots * sizeof(int64_t)));
  HCTR_HIP(hipMalloc(&d_vals, slots * sizeof(uint64_t)));
  size_t live = 0;
  int rc = c.ht.dump(d_keys, d_vals, &live, s);
  if (rc == HCTR_OK) {
    const size_t n = live < num_keys ? live : num_keys;
    if (exported) *exported = n;
    if (n > 0) {
      HCTR_REQUIRE(keys && values, "null pointer");
      if (h->key_type == HCTR_KEY_U32) {
        std::vector<int64_t> hk(n);
        std::vector<uint32_t> hk32(n);
        (void)hipMemcpy(hk.data(), d_keys, n * sizeof(int64_t), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; i++) hk32[i] = (uint32_t)hk[i];
        (void)hipMemcpy(keys, hk32.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
      } else {
        (void)hipMemcpyAsync(keys, d_keys, n * sizeof(int64_t), hipMemcpyDeviceToDevice, s);
      }
      launch_det_gather(d_vals, n, c.rows, c.dim, values, s);
      (void)hipStreamSynchronize(s);
    }
  }
  (void)hipFree(d_keys);
  (void)hipFree(d_vals);
  return rc;
}

int hctr_det_lookup_index(hctr_det* h, size_t class_index, const void* keys, size_t num_keys,
                          int insert, uint64_t* row_index, hctr_stream_t stream) {
  HCTR_REQUIRE(h && class_index < h->cls.size(), "class_index");
  if (num_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(keys && row_index, "null pointer");
  DetClass& c = h->cls[class_index];
  if (insert) {
    std::vector<Inserted> ins;
    HCTR_TRY(class_lookup_insert(h, c, class_index, keys, num_keys, row_index, as_stream(stream), &ins));
    return det_verify_inserts(h, ins, as_stream(stream));
  }
  return c.ht.get_mark(keys, num_keys, nullptr, row_index, as_stream(stream));
}

int hctr_det_rows(hctr_det* h, size_t class_index, float** rows, size_t* capacity) {
  HCTR_REQUIRE(h && class_index < h->cls.size() && rows, "class_index");
  *rows = h->cls[class_index].rows;
  if (capacity) *capacity = h->cls[class_index].cap;
  return HCTR_OK;
}

int hctr_det_row_store(hctr_det* h, float** rows, uint64_t* total_rows) {
  HCTR_REQUIRE(h && rows, "null pointer");
  *rows = h->arena;
  if (total_rows) *total_rows = h->arena ? (uint64_t)h->arena_rows : 0;
  return HCTR_OK;
}

int hctr_det_state_store(hctr_det* h, int num_state, float** state0, float** state1,
                         hctr_stream_t stream) {
  HCTR_REQUIRE(h && state0, "null pointer");
  HCTR_REQUIRE(num_state >= 1 && num_state <= 2, "num_state");
  HCTR_REQUIRE(h->arena != nullptr, "the table keeps no flat row store (classes of several dimensions)");
  hipStream_t s = as_stream(stream);
  while (h->n_state < num_state) {
    VmArray& a = h->state_va[h->n_state];
    HCTR_TRY(vm_reserve(a, h->rows_va.va_bytes));
    for (size_t ci = 0; ci < h->cls.size(); ci++)  // behind what the rows have mapped so far
      HCTR_TRY(vm_map(a, h->region_off[ci], h->cls[ci].mapped, true, s));
    h->state_arena[h->n_state++] = reinterpret_cast<float*>(a.base);
  }
  *state0 = h->state_arena[0];
  if (state1) *state1 = h->state_arena[1];
  return HCTR_OK;
}

int hctr_det_clear(hctr_det* h, hctr_stream_t stream) {
  HCTR_REQUIRE(h, "null handle");
  hipStream_t s = as_stream(stream);
  // (rows are handed out from 0 again: their state starts from zero again)
  for (int k = 0; k < h->n_state; k++)
    for (size_t ci = 0; ci < h->cls.size(); ci++)
      HCTR_HIP(hipMemsetAsync(h->state_va[k].base + h->region_off[ci], 0, h->cls[ci].mapped, s));
  for (auto& c : h->cls) {
    HCTR_TRY(c.ht.clear(s));
    HCTR_HIP(hipMemsetAsync(c.d_erased, 0, sizeof(unsigned long long), s));
    c.head_bound = 0;
  }
  return HCTR_OK;
}

int hctr_det_size_per_class(hctr_det* h, size_t* sizes, hctr_stream_t stream) {
  HCTR_REQUIRE(h && sizes, "null pointer");
  hipStream_t s = as_stream(stream);
  for (size_t i = 0; i < h->cls.size(); i++) {
    size_t head = 0;
    HCTR_TRY(h->cls[i].ht.value_head(s, &head));
    unsigned long long er = 0;
    HCTR_HIP(hipMemcpy(&er, h->cls[i].d_erased, sizeof(er), hipMemcpyDeviceToHost));
    h->cls[i].head_bound = head;
    sizes[i] = head - (size_t)er;
  }
  return HCTR_OK;
}

int hctr_det_repair_count(const hctr_det* h, uint64_t* out) {
  HCTR_REQUIRE(h && out, "null pointer");
  *out = h->repairs;
  return HCTR_OK;
}

int hctr_det_capacity_per_class(const hctr_det* h, size_t* caps) {
  HCTR_REQUIRE(h && caps, "null pointer");
  for (size_t i = 0; i < h->cls.size(); i++) caps[i] = h->cls[i].cap;
  return HCTR_OK;
}

size_t hctr_det_num_classes(const hctr_det* h) { return h ? h->cls.size() : 0; }

int hctr_det_update(hctr_det* weights, hctr_det* states, const hctr_det_opt_params* p,
                    const void* unique_keys, size_t num_unique_keys, const size_t* id_spaces,
                    const size_t* id_space_offsets, size_t num_id_spaces,
                    const uint32_t* ev_start_indices, const float* wgrad, hctr_stream_t stream) {
  HCTR_REQUIRE(weights && p, "null pointer");
  if (num_unique_keys == 0) return HCTR_OK;
  HCTR_REQUIRE(unique_keys && ev_start_indices && wgrad, "null pointer");
  const int opt = p->optimizer;
  HCTR_REQUIRE(opt == HCTR_OPT_SGD || opt == HCTR_OPT_ADAM || opt == HCTR_OPT_ADAGRAD ||
                   opt == HCTR_OPT_MOMENTUM_SGD || opt == HCTR_OPT_NESTEROV ||
                   opt == HCTR_OPT_RMSPROP || opt == HCTR_OPT_FTRL,
               "optimizer");
  const bool needs_state = opt != HCTR_OPT_SGD;
  HCTR_REQUIRE(!needs_state || states, "this optimizer needs a state table");
  hipStream_t s = as_stream(stream);
  hctr_det* h = weights;
  std::vector<Range> rs;
  HCTR_TRY(ranges_of(h, num_unique_keys, id_spaces, id_space_offsets, num_id_spaces, &rs));
  HCTR_TRY(det_scratch(h, num_unique_keys));
  DetOpt o{};
  o.optimizer = opt;
  o.lr = p->lr;
  o.beta1 = p->beta1;
  o.beta2 = p->beta2;
  o.epsilon = p->epsilon;
  o.momentum = p->momentum_factor;
  o.scaler = p->scaler;
  o.rms_beta = p->rmsprop_beta;
  o.lambda1 = p->ftrl_lambda1;
  o.lambda2_plus_beta_div_lr = p->ftrl_lambda2 + p->ftrl_beta / p->lr;
  if (opt == HCTR_OPT_ADAM) {
    // ++adam.times; lr * adam.bias()  (dynamic_embedding.cu:239-240, optimizer.hpp:58-60)
    const uint64_t t = ++weights->adam_times;
    o.lr_scaled_bias = p->lr * (float)(std::sqrt(1.0 - std::pow((double)p->beta2, (double)t)) /
                                       (1.0 - std::pow((double)p->beta1, (double)t)));
  }
  const int smul = (opt == HCTR_OPT_ADAM || opt == HCTR_OPT_FTRL) ? 2 : 1;
  // several id spaces of one vector size that tile the key list (the embedding_collection case):
  // row addresses of all keys by the multi-class lookup, then one optimizer launch
  bool tiled = rs.size() > 1 && rs.front().off == 0;
  size_t covered = 0;
  for (const Range& r : rs) {
    tiled = tiled && r.off == covered && h->cls[r.cls].dim == h->cls[rs[0].cls].dim;
    if (needs_state)
      tiled = tiled && r.cls < states->cls.size() &&
              states->cls[r.cls].dim == h->cls[r.cls].dim * smul;
    covered += r.n;
  }
  if (tiled && covered == num_unique_keys) {
    if (num_unique_keys > h->ptr_cap) {
      if (h->ptr_w) (void)hipFree(h->ptr_w);
      if (h->ptr_s) (void)hipFree(h->ptr_s);
      h->ptr_w = h->ptr_s = nullptr;
      size_t c = h->ptr_cap ? h->ptr_cap : 1024;
      while (c < num_unique_keys) c *= 2;
      HCTR_HIP(hipMalloc(&h->ptr_w, c * sizeof(float*)));
      HCTR_HIP(hipMalloc(&h->ptr_s, c * sizeof(float*)));
      h->ptr_cap = c;
    }
    // (the reference looks the weights up with insertion only for Ftrl)
    HCTR_TRY(hctr_det_lookup_rows(h, unique_keys, num_unique_keys, id_spaces, id_space_offsets,
                                  num_id_spaces, opt == HCTR_OPT_FTRL ? 1 : 0, h->ptr_w, nullptr,
                                  nullptr, stream));
    if (needs_state)
      HCTR_TRY(hctr_det_lookup_rows(states, unique_keys, num_unique_keys, id_spaces,
                                    id_space_offsets, num_id_spaces, 1, h->ptr_s, nullptr, nullptr,
                                    stream));
    const int dim = h->cls[rs[0].cls].dim;
    hipLaunchKernelGGL(det_update_ptr_kernel,
                       dim3(grid_for(num_unique_keys * (size_t)dim, kBlock, 8192)), dim3(kBlock), 0,
                       s, o, num_unique_keys, dim, h->ptr_w, needs_state ? h->ptr_s : nullptr,
                       ev_start_indices, wgrad);
    HCTR_LAUNCH_CHECK();
    return HCTR_OK;
  }
  for (const Range& r : rs) {
    if (r.n == 0) continue;
    DetClass& cw = h->cls[r.cls];
    const void* kp = key_at(h, unique_keys, r.off);
    float* rows_s = nullptr;
    if (needs_state) {
      HCTR_REQUIRE(r.cls < states->cls.size() && states->cls[r.cls].dim == cw.dim * smul,
                   "state table: dimension must be ev_size * num_parameters_per_weight");
      DetClass& cs = states->cls[r.cls];
      std::vector<Inserted> ins;
      HCTR_TRY(class_lookup_insert(states, cs, r.cls, kp, r.n, h->idx2, s, &ins));
      HCTR_TRY(det_verify_inserts(states, ins, s));
      rows_s = cs.rows;
    }
    if (opt == HCTR_OPT_FTRL) {  // the reference looks the weights up (inserting) for Ftrl
      std::vector<Inserted> ins;
      HCTR_TRY(class_lookup_insert(h, cw, r.cls, kp, r.n, h->idx, s, &ins));
      HCTR_TRY(det_verify_inserts(h, ins, s));
    } else
      HCTR_TRY(cw.ht.get_mark(kp, r.n, nullptr, h->idx, s));
    hipLaunchKernelGGL(det_update_kernel, dim3(grid_for(r.n * (size_t)cw.dim, kBlock, 4096)),
                       dim3(kBlock), 0, s, o, r.n, cw.dim, h->idx, needs_state ? h->idx2 : nullptr,
                       cw.rows, rows_s, ev_start_indices + r.off, wgrad);
    HCTR_LAUNCH_CHECK();
  }
  return HCTR_OK;
}

}  // extern "C"
