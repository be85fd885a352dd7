// sparse_update.hip -- fused embedding backward + sparse optimizer update on gfx950.
//
// Replaces backward_sum/backward_mean (R/HugeCTR/src/embeddings/backward_functor.cu:26-104) and
// EmbeddingOptimizer::update (R/HugeCTR/src/optimizers/sparse_optimizer.cu:622-864).
// Reference pipeline: wgrad copy -> sample-id expand -> radix sort (row index -> bucket id) ->
// run flags -> scan -> BLOCKING D2H of the run count -> one block per unique row.
// Here: no wgrad tensor (the top gradient is read in place, the mean scale 1/n is applied while
// accumulating), the run count stays on the device (persistent grid-stride over runs), and a
// "group" of D/4 lanes owns a row with 16-byte accesses.  Gradient accumulation per row is in
// ascending bucket id, exactly the reference's order (stable sort), then / scaler.
#include "sparse_update.h"
#include "radix_sort.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "block_prims.h"

namespace hctr {
namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;

template <typename GradT>
struct Load4;
template <>
struct Load4<float> {
  typedef float4 raw;  // 4 elements as they sit in memory
  __device__ __forceinline__ static raw ld_raw(const float* p) {
    return *reinterpret_cast<const float4*>(p);
  }
  __device__ __forceinline__ static float4 cvt(raw r) { return r; }
  __device__ __forceinline__ static float4 ld(const float* p) {
    return *reinterpret_cast<const float4*>(p);
  }
  __device__ __forceinline__ static float ld1(const float* p) { return *p; }
  __device__ __forceinline__ static float rnd(float v) { return v; }
};
template <>
struct Load4<__half> {
  typedef uint2 raw;
  __device__ __forceinline__ static raw ld_raw(const __half* p) {
    return *reinterpret_cast<const uint2*>(p);
  }
  __device__ __forceinline__ static float4 cvt(raw u) {
    __half2 a = *reinterpret_cast<__half2*>(&u.x), b = *reinterpret_cast<__half2*>(&u.y);
    float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
  }
  __device__ __forceinline__ static float4 ld(const __half* p) { return cvt(ld_raw(p)); }
  __device__ __forceinline__ static float ld1(const __half* p) { return __half2float(*p); }
  __device__ __forceinline__ static float rnd(float v) { return __half2float(__float2half_rn(v)); }
};
template <>
struct Load4<__hip_bfloat16> {
  typedef uint2 raw;
  __device__ __forceinline__ static raw ld_raw(const __hip_bfloat16* p) {
    return *reinterpret_cast<const uint2*>(p);
  }
  __device__ __forceinline__ static float4 cvt(raw u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
  }
  __device__ __forceinline__ static float4 ld(const __hip_bfloat16* p) { return cvt(ld_raw(p)); }
  __device__ __forceinline__ static float ld1(const __hip_bfloat16* p) {
    return __bfloat162float(*p);
  }
  __device__ __forceinline__ static float rnd(float v) {
    return __bfloat162float(__float2bfloat16(v));
  }
};

// ---- step 1: (row index, bucket id) pairs (sample_id_expand_kernel :189-200) ------------------
template <typename OffT, typename SortK>
__global__ void __launch_bounds__(kBlock)
    expand_pairs_kernel(size_t buckets, size_t n_sort, const OffT* __restrict__ row_offset,
                        const uint64_t* __restrict__ value_index, SortK* __restrict__ keys,
                        uint32_t* __restrict__ vals, uint32_t* __restrict__ span_count,
                        uint32_t map_inner, uint32_t map_outer,
                        const uint32_t* __restrict__ skip_flag) {
  const size_t nnz = (size_t)row_offset[buckets];
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (tid == 0 && blockIdx.y == 0)  // long-run lists of seg_reduce / seg_combine
    span_count[0] = span_count[1] = span_count[2] = span_count[3] = 0u;
  // one-hot batch: the sort's first pass takes rows and payloads from where they lie (RsFirst)
  if (skip_flag != nullptr && *skip_flag != 0u) return;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  // key-parallel (block_prims.h): the payload is the gradient row of the key's bucket
  // (SparseUpdater::map_inner)
  for_each_key_wave(buckets, row_offset, [&](size_t u, size_t j) {
    if (j >= n_sort) return;
    keys[j] = (SortK)value_index[j];
    vals[j] = map_inner ? ((uint32_t)u % map_inner) * map_outer + (uint32_t)u / map_inner
                        : (uint32_t)u;
  });
  // padding (host upper bound > live nnz): sorts to the end, never forms a counted run
  if (blockIdx.y != 0) return;
  for (size_t j = nnz + tid; j < n_sort; j += nthreads) {
    keys[j] = (SortK)~(SortK)0;
    vals[j] = 0xFFFFFFFFu;
  }
}

// ---- step 2: run starts ------------------------------------------------------------------------
template <typename SortK>
__device__ __forceinline__ bool is_run_start(const SortK* k, size_t i, size_t nnz) {
  if (i >= nnz) return false;
  return i == 0 || k[i] != k[i - 1];
}

template <typename OffT, typename SortK>
__global__ void __launch_bounds__(kBlock)
    run_count_kernel(const SortK* __restrict__ keys, const OffT* __restrict__ row_offset,
                     size_t buckets, size_t n_tiles, uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t smem[kBlock / 64 + 1];
  const size_t nnz = (size_t)row_offset[buckets];
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t c = 0;
#pragma unroll
    for (int r = 0; r < kTile / kBlock; r++) {
      size_t i = tile * kTile + r * kBlock + threadIdx.x;
      c += is_run_start(keys, i, nnz) ? 1u : 0u;
    }
    uint32_t tot = block_reduce_sum<uint32_t, kBlock>(c, smem);
    if (threadIdx.x == 0) tile_sums[tile] = tot;
  }
}

__global__ void __launch_bounds__(1024)
    scan_tiles_u32_kernel(uint32_t* sums, size_t m, uint64_t* d_total) {
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
    if (i < m) sums[i] = (uint32_t)(c + ex);
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *d_total = carry;
}

template <typename OffT, typename SortK>
__global__ void __launch_bounds__(kBlock)
    run_write_kernel(const SortK* __restrict__ keys, const OffT* __restrict__ row_offset,
                     size_t buckets, size_t n_tiles, const uint32_t* __restrict__ tile_sums,
                     const uint64_t* __restrict__ d_num_runs, uint32_t* __restrict__ run_start) {
  __shared__ uint32_t smem[kBlock / 64 + 1];
  const size_t nnz = (size_t)row_offset[buckets];
  if (blockIdx.x == 0 && threadIdx.x == 0) run_start[*d_num_runs] = (uint32_t)nnz;
  for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t run = tile_sums[tile];
#pragma unroll
    for (int r = 0; r < kTile / kBlock; r++) {
      size_t i = tile * kTile + r * kBlock + threadIdx.x;
      bool f = is_run_start(keys, i, nnz);
      uint32_t tot;
      uint32_t ex = block_exclusive_scan<uint32_t, kBlock>(f ? 1u : 0u, smem, &tot);
      if (f) run_start[run + ex] = (uint32_t)i;
      run += tot;
    }
  }
}

// ---- step 3: per-row ordered reduce + optimizer -------------------------------------------------
struct OptConst {
  int optimizer, update_type;
  float lr, beta1, beta2, epsilon, mf, scaler;
  float alpha_t;         // lr * adam.bias()
  float alpha_t_common;  // lr / (1 - beta1) (lazy adam)
  float ftrl_l1, ftrl_l2b;  // lambda1, lambda2 + beta / lr
  unsigned long long times;
  int state_half;  // optimizer state carries fp16 values
};

// OptimizerTensor<TypeEmbeddingComp> (R/HugeCTR/include/optimizer.hpp:284-296): with fp16 embeddings
// the reference keeps m / v / accumulators in fp16 -- every kernel converts the stored value to
// float, computes in float and converts the result back on the store; the weight update of the
// same step uses the unrounded float.  Here too (round 4): with state_half the state arrays ARE
// __half arrays (half the footprint and the traffic of the fp32 arrays rounds 1-3 kept); the
// pointers travel as float* and are re-typed where they are dereferenced (ld_state / st_state).
__device__ __forceinline__ float state_store(int state_half, float x) {
  if (!state_half) return x;
  // the fp32 result first, THEN the conversion (two roundings, as the reference's float math +
  // TypeConvertFunc does): without the barrier the compiler folds a preceding multiply into one
  // mixed-precision instruction that rounds the exact product straight to fp16
  asm volatile("" : "+v"(x));
  return __half2float(__float2half_rn(x));
}

// element f of a state array / the four elements from f on (f a multiple of 4)
__device__ __forceinline__ float ld_state1(const float* base, size_t f, int half) {
  return half ? __half2float(reinterpret_cast<const __half*>(base)[f]) : base[f];
}
__device__ __forceinline__ void st_state1(float* base, size_t f, int half, float v) {
  if (half) reinterpret_cast<__half*>(base)[f] = __float2half_rn(v);  // (v is fp16-valued: exact)
  else base[f] = v;
}
__device__ __forceinline__ float4 ld_state4(const float* base, size_t f, int half) {
  if (half)
    return Load4<__half>::cvt(
        *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(base) + f));
  return *reinterpret_cast<const float4*>(base + f);
}
__device__ __forceinline__ void st_state4(float* base, size_t f, int half, const float4& v) {
  if (half) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 u;
    u.x = *reinterpret_cast<const uint32_t*>(&a);
    u.y = *reinterpret_cast<const uint32_t*>(&b);
    *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(base) + f) = u;
  } else {
    *reinterpret_cast<float4*>(base + f) = v;
  }
}

// internal pseudo-optimizer of hctr_updater_reduce_presorted: table[row] = gradient sum (no read)
constexpr int kOptStoreSum = 1000;

// one element of one row; formulas cite sparse_optimizer.cu
__device__ __forceinline__ void apply_opt(const OptConst& o, float gi, float& w, float* s0p,
                                          float* s1p, unsigned long long* ptp) {
  switch (o.optimizer) {
    case HCTR_OPT_SGD:  // opt_sgd_kernel :497-518
      w += -o.lr * gi;
      break;
    case kOptStoreSum:
      w = gi;
      break;
    case HCTR_OPT_FTRL: {  // FtrlOptimizer::update, ragged_static_embedding.cu:159-290 (s0 = n, s1 = z)
      float ni = *s0p;
      const float sq = sqrtf(ni + 1.1920929e-07f);
      ni = ni + gi * gi;
      const float sqn = sqrtf(ni + 1.1920929e-07f);
      const float sigma = (sqn - sq) / o.lr;
      const float zi = *s1p + gi - sigma * w;
      const float p = (1.f - 2.f * (float)signbit(zi)) * o.ftrl_l1 - zi;
      const float q = sqn / o.lr + o.ftrl_l2b;
      w = p / q * (float)signbit(o.ftrl_l1 - fabsf(zi));
      *s0p = state_store(o.state_half, ni);
      *s1p = state_store(o.state_half, zi);
    } break;
    case HCTR_OPT_ADAGRAD: {  // opt_adagrad_kernel :410-437 (Global == Local)
      float accum = *s0p + gi * gi;
      *s0p = state_store(o.state_half, accum);
      w += -o.lr * gi / (sqrtf(accum) + o.epsilon);
    } break;
    case HCTR_OPT_ADAM:
      if (o.update_type == HCTR_UPDATE_LOCAL) {  // opt_adam_kernel :379-408
        float mi = o.beta1 * *s0p + (1.0f - o.beta1) * gi;
        float vi = o.beta2 * *s1p + (1.0f - o.beta2) * gi * gi;
        *s0p = state_store(o.state_half, mi);
        *s1p = state_store(o.state_half, vi);
        w += -o.alpha_t * mi / (sqrtf(vi) + o.epsilon);
      } else if (o.update_type == HCTR_UPDATE_GLOBAL) {  // opt_adam_kernel_global :241-265
        *s0p = state_store(o.state_half, *s0p + (1.0f - o.beta1) * gi / o.beta1);
        *s1p = state_store(o.state_half, *s1p + (1.0f - o.beta2) * gi * gi / o.beta2);
      } else {  // opt_adam_kernel_lazy :524-561
        unsigned long long pt = *ptp;
        *ptp = o.times;
        unsigned long long skipped = o.times - pt;
        float b1ps = powf(o.beta1, (float)skipped);
        float a = o.alpha_t_common * sqrtf(1.0f - powf(o.beta2, (float)pt)) /
                  (1.0f - powf(o.beta1, (float)pt)) * (1.0f - b1ps);
        float mi = *s0p, vi = *s1p;
        w += -a * mi / (sqrtf(vi) + o.epsilon);
        mi = b1ps * mi + (1.0f - o.beta1) * gi;
        vi = powf(o.beta2, (float)skipped) * vi + (1.0f - o.beta2) * gi * gi;
        *s0p = state_store(o.state_half, mi);
        *s1p = state_store(o.state_half, vi);
      }
      break;
    case HCTR_OPT_MOMENTUM_SGD:
      if (o.update_type == HCTR_UPDATE_LOCAL) {  // opt_momentum_sgd_kernel :440-465
        float mo = o.mf * *s0p - o.lr * gi;
        *s0p = state_store(o.state_half, mo);
        w += mo;
      } else {  // opt_momentum_sgd_kernel_global :292-312
        *s0p = state_store(o.state_half, *s0p - o.lr * gi / o.mf);
      }
      break;
    case HCTR_OPT_NESTEROV:
      if (o.update_type == HCTR_UPDATE_LOCAL) {  // opt_nesterov_kernel :468-494
        float accm_old = *s0p;
        float accm_new = o.mf * accm_old - o.lr * gi;
        *s0p = state_store(o.state_half, accm_new);
        w += -o.mf * accm_old + (1.0f + o.mf) * accm_new;
      } else {  // nesterov_local_update_kernel_global :352-375
        float accm = *s0p;
        accm -= o.lr * gi;
        *s0p = state_store(o.state_half, accm);
        w -= (1.0f + o.mf) * (o.lr * gi);
      }
      break;
    default: break;
  }
}

__device__ __forceinline__ bool needs_s0(const OptConst& o) {
  return o.optimizer != HCTR_OPT_SGD && o.optimizer != kOptStoreSum;
}
__device__ __forceinline__ bool needs_s1(const OptConst& o) {
  return o.optimizer == HCTR_OPT_ADAM || o.optimizer == HCTR_OPT_FTRL;
}
__device__ __forceinline__ bool needs_pt(const OptConst& o) {
  return o.optimizer == HCTR_OPT_ADAM && o.update_type == HCTR_UPDATE_LAZY_GLOBAL;
}

// Row update shared by seg_apply_kernel and seg_combine_kernel, split in load / compute / store so
// that callers can keep several rows in flight: gi = acc / scaler, then the optimizer on the 4
// elements this lane owns.
struct RowRegs {
  float4 w, s0, s1;
  unsigned long long pt[4];
};

template <int LPR>
__device__ __forceinline__ void row_load(const OptConst& o, uint64_t row, int l, RowRegs& r,
                                         const float* __restrict__ table,
                                         const float* __restrict__ state0,
                                         const float* __restrict__ state1,
                                         const unsigned long long* __restrict__ prev_time) {
  constexpr int D = LPR * 4;
  const size_t f = row * (uint64_t)D + l * 4;
  r.s0 = make_float4(0.f, 0.f, 0.f, 0.f);
  r.w = r.s0;
  if (o.optimizer != kOptStoreSum) r.w = *reinterpret_cast<const float4*>(table + f);
  r.s1 = r.s0;
  r.pt[0] = r.pt[1] = r.pt[2] = r.pt[3] = 1ull;
  if (needs_s0(o)) r.s0 = ld_state4(state0, f, o.state_half);
  if (needs_s1(o)) r.s1 = ld_state4(state1, f, o.state_half);
  if (needs_pt(o)) {
#pragma unroll
    for (int t = 0; t < 4; t++) r.pt[t] = prev_time[f + t];
  }
}

__device__ __forceinline__ void row_compute(const OptConst& o, float4 gi, RowRegs& r) {
  gi.x /= o.scaler;
  gi.y /= o.scaler;
  gi.z /= o.scaler;
  gi.w /= o.scaler;
  apply_opt(o, gi.x, r.w.x, &r.s0.x, &r.s1.x, &r.pt[0]);
  apply_opt(o, gi.y, r.w.y, &r.s0.y, &r.s1.y, &r.pt[1]);
  apply_opt(o, gi.z, r.w.z, &r.s0.z, &r.s1.z, &r.pt[2]);
  apply_opt(o, gi.w, r.w.w, &r.s0.w, &r.s1.w, &r.pt[3]);
}

template <int LPR>
__device__ __forceinline__ void row_store(const OptConst& o, uint64_t row, int l, const RowRegs& r,
                                          float* __restrict__ table, float* __restrict__ state0,
                                          float* __restrict__ state1,
                                          unsigned long long* __restrict__ prev_time) {
  constexpr int D = LPR * 4;
  const size_t f = row * (uint64_t)D + l * 4;
  const bool w_written = !((o.optimizer == HCTR_OPT_ADAM || o.optimizer == HCTR_OPT_MOMENTUM_SGD) &&
                           o.update_type == HCTR_UPDATE_GLOBAL);
  if (w_written) *reinterpret_cast<float4*>(table + f) = r.w;
  if (needs_s0(o)) st_state4(state0, f, o.state_half, r.s0);
  if (needs_s1(o)) st_state4(state1, f, o.state_half, r.s1);
  if (needs_pt(o)) {
#pragma unroll
    for (int t = 0; t < 4; t++) prev_time[f + t] = r.pt[t];
  }
}

// A key that found no row (hash table overflow, or an unseen key of an index-only call) carries
// kInvalidIndex; as a 32-bit sort key that is 0xFFFFFFFF, which create() keeps out of the legal row
// range.  Such positions sort behind every live row and their run is dropped by every writer.
constexpr uint64_t kNoRow = 0xFFFFFFFFull;

template <int LPR>
__device__ __forceinline__ void apply_row_vec4(const OptConst& o, uint64_t row, int l, float4 gi,
                                               float* __restrict__ table,
                                               float* __restrict__ state0,
                                               float* __restrict__ state1,
                                               unsigned long long* __restrict__ prev_time) {
  if (row == kNoRow) return;
  RowRegs r;
  row_load<LPR>(o, row, l, r, table, state0, state1, prev_time);
  row_compute(o, gi, r);
  row_store<LPR>(o, row, l, r, table, state0, state1, prev_time);
}

// Tile-based segmented reduce + optimizer.  The sorted (row, bucket) list is cut into tiles of
// kSegTile positions; a group of LPR lanes walks one tile in order, so every group performs about
// the same number of gradient-row reads no matter how skewed the key distribution is (the
// reference gives one block to each unique row, sparse_optimizer.cu:223-237 -- a power-law head
// row with 20k duplicates is then one serial 20k-iteration loop).
//   * A run (= all gradients of one row) that starts in tile t is OWNED by tile t's group.  The
//     owner follows it up to one tile past its own tile end; the next tile's group skips those
//     leading positions.  So every run that ends before the end of tile t+1 is reduced by one group
//     in ascending bucket order (the reference's order, stable sort) and applied at once.
//   * A run that reaches beyond tile t+1 is "long": the owner stores the sum of its own part in
//     tail[t] and appends t to span_list; every later tile the run touches stores its part in
//     head[t'].  seg_combine_kernel adds tail + heads in a fixed order (deterministic).
constexpr int kSegTile = 32;

// number of keys in bucket b (the mean combiner's divisor)
__device__ __forceinline__ int bucket_len(const void* row_offset_v, bool off_is_u32, uint32_t b) {
  if (off_is_u32) {
    const uint32_t* ro = (const uint32_t*)row_offset_v;
    return (int)(ro[b + 1] - ro[b]);
  }
  const long long* ro = (const long long*)row_offset_v;
  return (int)(ro[b + 1] - ro[b]);
}

template <typename GradT>
__device__ __forceinline__ float4 scaled_grad(typename Load4<GradT>::raw r, int combiner, int n) {
  float4 v = Load4<GradT>::cvt(r);
  if (combiner == 1) {
    // backward_mean_align2_kernel (backward_functor.cu:83-104): the scaler is rounded to the
    // gradient type before the multiply; fp32 gradients: rnd() is the identity
    const float sc = Load4<GradT>::rnd(n > 1 ? 1.0f / (float)n : 1.0f);
    v.x = Load4<GradT>::rnd(v.x * sc);
    v.y = Load4<GradT>::rnd(v.y * sc);
    v.z = Load4<GradT>::rnd(v.z * sc);
    v.w = Load4<GradT>::rnd(v.w * sc);
  }
  return v;
}

// Phase A: segmented sums.  Pure load/accumulate/store -- no read-modify-write of table rows
// inside the walk.  The kernel is bound by DEPENDENT memory round trips per tile, not by bytes, so
// everything a tile may need is fetched in as few trips as possible:
//   trip 1: the tile's 32 (row, bucket) pairs, one per lane (coalesced), the NEXT tile's pairs
//           (for the run that overhangs the tile end) and the four neighbour rows that decide
//           ownership -- run starts / overhang length become 32-bit ballot masks;
//   trips 2..: the 32 gradient rows of the tile plus the first kSegAhead rows of the overhang,
//           issued back to back in batches of QB raw (unconverted) fragments, clamped to a row the
//           batch reads anyway where a position is not needed.
// The only sequential part is the fp32 add chain, which is what fixes the summation order.
// The sum of a run its owner finishes goes to gsum[start position]; seg_apply_kernel picks it up.
constexpr int kSegAhead = 8;

// row id of tile position q (0..31): the metadata lane that holds it broadcasts it to the group
template <int NPL, int ML>
__device__ __forceinline__ uint32_t seg_row_at(const uint32_t (&mrow)[NPL], int q, int gshift) {
  uint32_t src = mrow[0];
#pragma unroll
  for (int j = 1; j < NPL; j++) src = (q / ML == j) ? mrow[j] : src;
  return (uint32_t)__shfl((int)src, gshift + (q % ML), 64);
}

constexpr int kFuseNone = 0, kFuseSgd = 1, kFuseAdaGrad = 2;

template <int LPR, typename OffT, typename SortK, typename GradT, int kFuse>
__global__ void __launch_bounds__(kBlock)
    seg_reduce_kernel(size_t buckets, const OffT* __restrict__ row_offset,
                      const SortK* __restrict__ sorted_rows,
                      const uint32_t* __restrict__ sorted_buckets, int combiner,
                      const GradT* __restrict__ grad, float* __restrict__ gsum,
                      float* __restrict__ head, float* __restrict__ tail,
                      uint32_t* __restrict__ span_list, uint32_t* __restrict__ span_count,
                      float* __restrict__ direct_out, const OffT* __restrict__ scale_ro,
                      OptConst fuse_o, float* __restrict__ fuse_state0,
                      const uint32_t* __restrict__ n_live) {
  // kFuse (kFuseSgd / kFuseAdaGrad): the optimizer applied where a run's sum is complete --
  // e.g. table[row] += -lr * (sum / scaler) -- right here (direct_out = the table) instead of
  // parking the sum in gsum for seg_apply.  Every row is one run owned by one lane group, so
  // nobody else touches it; the arithmetic is seg_apply's (apply_opt), bit for bit, without the
  // gsum round trip (2 x D x 4 bytes per unique row).  Optimizers with two state vectors or
  // time stamps keep the two-pass form (their row registers would cost the gather its occupancy).
  // Measured (MI355X): one-hot Criteo-1TB update 231 -> 209 us, embedding_collection one-hot
  // backward+update 365 -> 295 us, multi-hot MLPerf shape 2.15 -> 1.84 ms.  (No-return fp32
  // atomic adds in place of the read-modify-write were 2x SLOWER: 496 us / 4.2 ms.)
  // scale_ro: the CSR whose bucket lengths divide a mean gradient.  The distributed embedding
  // divides by the bucket's key count over ALL GPUs (backward() with the all-reduced row offsets,
  // distributed_slot_sparse_embedding_hash.hpp:216-221), not by this rank's filtered count.
  // direct_out != nullptr (hctr_updater_reduce_presorted): the sum of a finished run goes to
  // direct_out[row] instead of gsum[run start] -- no apply pass is needed afterwards
  typedef typename Load4<GradT>::raw Raw;
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int T = kSegTile;
  constexpr int LA = kSegAhead;
  constexpr int ML = LPR < T ? LPR : T;  // lanes of a group that carry tile metadata
  constexpr int NPL = T / ML;            // metadata entries per such lane
  constexpr int QB = sizeof(Raw) == 8 ? 20 : 10;  // fragments in flight per lane: 40 VGPRs
  constexpr bool kOff32 = sizeof(OffT) == 4;
  static_assert(T == 32 && (T + LA) % QB == 0, "masks are 32-bit; batches tile T + LA");
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const int gshift = ((threadIdx.x & 63) / LPR) * LPR;  // first lane of my group in the wave
  constexpr unsigned long long kGroupMask = ML >= 64 ? ~0ull : ((1ull << ML) - 1ull);
  // (n_live: the sorted list holds the cold rows' positions only -- the sort's first pass left the
  //  hot rows to hot_chunk_kernel and posted how many pairs it kept)
  const size_t nnz = n_live != nullptr ? (size_t)*n_live : (size_t)row_offset[buckets];
  const size_t n_tiles = (nnz + T - 1) / T;
  // kFuse: the row update of a finished run is completed when the NEXT run finishes -- its row
  // (and accumulator) read travels while the next run's gradients are added, instead of stalling
  // the lane group (one-hot update 205 -> 195 us, multi-hot backward + update 1.73 -> 1.59 ms)
  uint32_t pend_row = 0xFFFFFFFFu;
  float4 pend_w = make_float4(0.f, 0.f, 0.f, 0.f), pend_d = pend_w;
  RowRegs pend_rr;  // kFuseAdaGrad: row + accumulator in flight, pend_d = the run's gradient sum
  auto pend_flush = [&]() {
    if (pend_row != 0xFFFFFFFFu) {
      if constexpr (kFuse == kFuseAdaGrad) {
        OptConst oo = fuse_o;
        oo.optimizer = HCTR_OPT_ADAGRAD;  // (compile-time: the state loads / stores fold)
        row_compute(oo, pend_d, pend_rr);
        row_store<LPR>(oo, (uint64_t)pend_row, l, pend_rr, direct_out, fuse_state0, nullptr, nullptr);
      } else {
        pend_w.x += pend_d.x;
        pend_w.y += pend_d.y;
        pend_w.z += pend_d.z;
        pend_w.w += pend_d.w;
        *reinterpret_cast<float4*>(direct_out + (size_t)pend_row * D + l * 4) = pend_w;
      }
    }
  };
  for (size_t tile = (size_t)blockIdx.x * GPB + g; tile < n_tiles;
       tile += (size_t)gridDim.x * GPB) {
    const size_t base = tile * T;
    const size_t end = (base + T < nnz) ? base + T : nnz;
    const size_t limit = (end + T < nnz) ? end + T : nnz;
    const int nvalid = (int)(end - base);
    // ---- trip 1: all metadata --------------------------------------------------------------
    uint32_t mrow[NPL], mbkt[NPL], prow[NPL], nrow[NPL], nbkt[NPL];
#pragma unroll
    for (int j = 0; j < NPL; j++) {
      const size_t pos = base + (size_t)j * ML + l;
      const bool valid = l < ML && pos < end;
      mrow[j] = valid ? (uint32_t)sorted_rows[pos] : 0xFFFFFFFFu;
      mbkt[j] = valid ? sorted_buckets[pos] : 0u;
      prow[j] = (valid && pos > 0) ? (uint32_t)sorted_rows[pos - 1] : 0xFFFFFFFFu;
      const size_t np = end + (size_t)j * ML + l;
      const bool nval = l < ML && np < limit;
      nrow[j] = nval ? (uint32_t)sorted_rows[np] : 0xFFFFFFFFu;
      nbkt[j] = nval ? sorted_buckets[np] : 0u;
    }
    // rows at base-T, base-T-1 (who owns a run that enters this tile) and at limit (does the
    // overhanging run reach beyond tile+1); 0xFFFFFFFF never equals a live row
    const uint32_t row_pt = base >= (size_t)T ? (uint32_t)sorted_rows[base - T] : 0xFFFFFFFFu;
    const uint32_t row_pt1 = base > (size_t)T ? (uint32_t)sorted_rows[base - T - 1] : 0xFFFFFFFFu;
    const uint32_t row_lim = limit < nnz ? (uint32_t)sorted_rows[limit] : 0xFFFFFFFFu;

    uint32_t startmask = 0u;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
      const size_t pos = base + (size_t)j * ML + l;
      const bool valid = l < ML && pos < end;
      const bool is_start = valid && (pos == 0 || prow[j] != mrow[j]);
      const unsigned long long bal = __ballot(is_start);
      startmask |= (uint32_t)((bal >> gshift) & kGroupMask) << (j * ML);
    }
    const uint32_t row0 = (uint32_t)__shfl((int)mrow[0], gshift, 64);
    const uint32_t cur_row =
        (uint32_t)__shfl((int)mrow[(nvalid - 1) / ML], gshift + ((nvalid - 1) % ML), 64);
    const uint32_t next_row0 = (uint32_t)__shfl((int)nrow[0], gshift, 64);
#define HCTR_RUN_DST(q_)                                                                        \
  ((direct_out != nullptr && seg_row_at<NPL, ML>(mrow, (q_), gshift) != 0xFFFFFFFFu)             \
       ? direct_out + (size_t)seg_row_at<NPL, ML>(mrow, (q_), gshift) * D                       \
       : gsum + (base + (size_t)(q_)) * D) /* a run of keys without a row has no output row */
    auto emit_run = [&](int q_run, const float4& a) {
      if constexpr (kFuse == kFuseSgd) {
        const uint32_t r = seg_row_at<NPL, ML>(mrow, q_run, gshift);
        if (r != 0xFFFFFFFFu) {
          pend_flush();
          pend_d.x = -fuse_o.lr * (a.x / fuse_o.scaler);
          pend_d.y = -fuse_o.lr * (a.y / fuse_o.scaler);
          pend_d.z = -fuse_o.lr * (a.z / fuse_o.scaler);
          pend_d.w = -fuse_o.lr * (a.w / fuse_o.scaler);
          pend_row = r;
          pend_w = *reinterpret_cast<const float4*>(direct_out + (size_t)r * D + l * 4);
        }
      } else if constexpr (kFuse == kFuseAdaGrad) {
        const uint32_t r = seg_row_at<NPL, ML>(mrow, q_run, gshift);
        if (r != 0xFFFFFFFFu) {
          pend_flush();
          OptConst oo = fuse_o;
          oo.optimizer = HCTR_OPT_ADAGRAD;
          pend_d = a;
          pend_row = r;
          row_load<LPR>(oo, (uint64_t)r, l, pend_rr, direct_out, fuse_state0, nullptr, nullptr);
        }
      } else {
        *reinterpret_cast<float4*>(HCTR_RUN_DST(q_run) + l * 4) = a;
      }
    };
    const bool ends_at_tile_end = end == nnz || next_row0 != cur_row;
    int q0 = 0;
    bool head_mode = false;
    if (base > 0 && (startmask & 1u) == 0u) {
      // the tile starts inside a run begun earlier: owned by the previous tile AND ending inside
      // this tile -> its owner reduces it, skip it; otherwise it is (part of) a long run.
      const bool owner_prev = row_pt != row0 || base == (size_t)T || row_pt1 != row0;
      const bool whole_tile = startmask == 0u;
      const bool ends_inside = !whole_tile || ends_at_tile_end;
      if (owner_prev && ends_inside) q0 = whole_tile ? nvalid : __ffs((int)startmask) - 1;
      else head_mode = true;
    }
    if (q0 >= nvalid) continue;  // the whole tile belonged to the previous tile's run
    // overhang: leading positions of the next tile that continue this tile's last run
    uint32_t matchmask = 0u;
#pragma unroll
    for (int j = 0; j < NPL; j++) {
      const unsigned long long bal = __ballot(nrow[j] == cur_row);
      matchmask |= (uint32_t)((bal >> gshift) & kGroupMask) << (j * ML);
    }
    const bool whole_head = head_mode && startmask == 0u;  // one earlier run covers the tile
    int cnt = (~matchmask == 0u) ? T : __ffs((int)~matchmask) - 1;  // leading ones
    if (ends_at_tile_end || whole_head) cnt = 0;
    const int cnt_la = cnt < LA ? cnt : LA;

    // ---- trips 2..: gradient rows, QB fragments in flight ---------------------------------
    int run_start = q0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 own_part = acc;
    const uint32_t b_q0 = (uint32_t)__shfl((int)mbkt[q0 / ML], gshift + (q0 % ML), 64);
#pragma unroll
    for (int qb = 0; qb < T + LA; qb += QB) {
      Raw v[QB];
      int nb[QB];
#pragma unroll
      for (int k = 0; k < QB; k++) {
        const int q = qb + k;
        uint32_t bsel;
        if (q < T) {
          const uint32_t bq = (uint32_t)__shfl((int)mbkt[q / ML], gshift + (q % ML), 64);
          bsel = (q >= q0 && q < nvalid) ? bq : b_q0;
        } else {
          const uint32_t bq =
              (uint32_t)__shfl((int)nbkt[(q - T) / ML], gshift + ((q - T) % ML), 64);
          bsel = (q - T) < cnt_la ? bq : b_q0;
        }
        v[k] = Load4<GradT>::ld_raw(grad + (size_t)bsel * D + l * 4);
        nb[k] = combiner == 1 ? bucket_len(scale_ro, kOff32, bsel) : 1;
      }
#pragma unroll
      for (int k = 0; k < QB; k++) {
        const int q = qb + k;
        if (q == T) own_part = acc;
        if (q < T) {
          if (q >= q0 && q < nvalid) {
            if (((startmask >> q) & 1u) != 0u && q != q0) {
              if (head_mode) *reinterpret_cast<float4*>(head + tile * D + l * 4) = acc;
              else emit_run(run_start, acc);
              acc = make_float4(0.f, 0.f, 0.f, 0.f);
              run_start = q;
              head_mode = false;
            }
            const float4 f = scaled_grad<GradT>(v[k], combiner, nb[k]);
            acc.x += f.x;
            acc.y += f.y;
            acc.z += f.z;
            acc.w += f.w;
          }
        } else if ((q - T) < cnt_la) {
          const float4 f = scaled_grad<GradT>(v[k], combiner, nb[k]);
          acc.x += f.x;
          acc.y += f.y;
          acc.z += f.z;
          acc.w += f.w;
        }
      }
    }
    if (head_mode) {  // one run covers the whole tile
      *reinterpret_cast<float4*>(head + tile * D + l * 4) = acc;
      continue;
    }
    if (cnt == 0) {  // the last run ends with the tile
      emit_run(run_start, acc);
      continue;
    }
    // the last run of this tile continues: this group owns it and follows it through tile+1
    if (cnt > LA) {
      constexpr int QC = 8;
#pragma unroll 1
      for (int qb = LA; qb < cnt; qb += QC) {
        Raw v[QC];
        int nb[QC];
#pragma unroll
        for (int k = 0; k < QC; k++) {
          const int q = (qb + k) < cnt ? qb + k : cnt - 1;
          // NPL > 1: the register index is dynamic here -> select with a small unrolled scan
          uint32_t src = nbkt[0];
#pragma unroll
          for (int j = 1; j < NPL; j++) src = (q / ML == j) ? nbkt[j] : src;
          const uint32_t bsel = (uint32_t)__shfl((int)src, gshift + (q % ML), 64);
          v[k] = Load4<GradT>::ld_raw(grad + (size_t)bsel * D + l * 4);
          nb[k] = combiner == 1 ? bucket_len(scale_ro, kOff32, bsel) : 1;
        }
#pragma unroll
        for (int k = 0; k < QC; k++) {
          if (qb + k < cnt) {
            const float4 f = scaled_grad<GradT>(v[k], combiner, nb[k]);
            acc.x += f.x;
            acc.y += f.y;
            acc.z += f.z;
            acc.w += f.w;
          }
        }
      }
    }
    // long <=> the run reaches beyond the end of tile+1
    const bool runs_on = cnt == T && limit < nnz && row_lim == cur_row;
    if (!runs_on) {
      emit_run(run_start, acc);
    } else {
      *reinterpret_cast<float4*>(tail + tile * D + l * 4) = own_part;
      if (l == 0) span_list[atomicAdd(span_count, 1u)] = (uint32_t)tile;
    }
  }
  if constexpr (kFuse != kFuseNone) pend_flush();
}
#undef HCTR_RUN_DST

// Phase B: one lane inspects one sorted position; run starts of runs that are not "long" are
// compacted with a wave ballot and handed to lane groups, which read the run's gradient sum from
// gsum[position] and apply the optimizer to the row (one coalesced D*4-byte RMW per row).
// kSgd: plain SGD known at compile time -- one float4 of state per row instead of the generic
// RowRegs (w, two state vectors, four time stamps: 148 VGPRs, 3 waves per SIMD), 8 rows per lane
// group in flight instead of 4 (93 VGPRs).  Same arithmetic, same bits; seg_apply 98 -> 70 us at
// the bench shape.
template <int LPR, typename OffT, typename SortK, bool kSgd>
__global__ void __launch_bounds__(kBlock)
    seg_apply_kernel(size_t buckets, const OffT* __restrict__ row_offset,
                     const SortK* __restrict__ sorted_rows, const float* __restrict__ gsum,
                     OptConst o, float* __restrict__ table, float* __restrict__ state0,
                     float* __restrict__ state1, unsigned long long* __restrict__ prev_time,
                     const uint32_t* __restrict__ n_live) {
  constexpr int D = LPR * 4;
  constexpr int G = 64 / LPR;  // groups per wavefront
  constexpr int T = kSegTile;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR;
  const int l = lane % LPR;
  const size_t nnz = n_live != nullptr ? (size_t)*n_live : (size_t)row_offset[buckets];
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t c0 = wave * 64; c0 < nnz; c0 += nwaves * 64) {
    const size_t p = c0 + lane;
    SortK row = 0;
    bool active = false;
    if (p < nnz) {
      row = sorted_rows[p];
      const bool is_start = p == 0 || sorted_rows[p - 1] != row;
      if (is_start) {
        const size_t e2 = (p / T + 2) * T;  // first position after the tile following p's tile
        const bool is_long = e2 < nnz && sorted_rows[e2] == row;
        active = !is_long && (uint64_t)row != kNoRow;
      }
    }
    unsigned long long mask = __ballot(active);
    // R rows per group per step: all gsum / table / state reads of a step are issued before the
    // first optimizer evaluation
    constexpr int R = kSgd ? 8 : 4;
    while (mask != 0ull) {
      int src[R];
#pragma unroll
      for (int k = 0; k < R; k++) {
        src[k] = -1;
#pragma unroll
        for (int q = 0; q < G; q++) {
          if (mask != 0ull) {
            const int bit = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            if (q == g) src[k] = bit;
          }
        }
      }
      uint32_t r2[R];
      float4 gi[R];
      if constexpr (kSgd) {
        float4 w[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
          r2[k] = (uint32_t)__shfl((int)row, src[k] < 0 ? 0 : src[k], 64);
          if (src[k] >= 0) {
            gi[k] = *reinterpret_cast<const float4*>(gsum + (c0 + src[k]) * D + l * 4);
            w[k] = *reinterpret_cast<const float4*>(table + (uint64_t)r2[k] * D + l * 4);
          }
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
          if (src[k] >= 0) {  // row_compute + apply_opt(HCTR_OPT_SGD): w += -lr * (g / scaler)
            w[k].x += -o.lr * (gi[k].x / o.scaler);
            w[k].y += -o.lr * (gi[k].y / o.scaler);
            w[k].z += -o.lr * (gi[k].z / o.scaler);
            w[k].w += -o.lr * (gi[k].w / o.scaler);
            *reinterpret_cast<float4*>(table + (uint64_t)r2[k] * D + l * 4) = w[k];
          }
        }
      } else {
        RowRegs rr[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
          r2[k] = (uint32_t)__shfl((int)row, src[k] < 0 ? 0 : src[k], 64);
          if (src[k] >= 0) {
            gi[k] = *reinterpret_cast<const float4*>(gsum + (c0 + src[k]) * D + l * 4);
            row_load<LPR>(o, (uint64_t)r2[k], l, rr[k], table, state0, state1, prev_time);
          }
        }
#pragma unroll
        for (int k = 0; k < R; k++) {
          if (src[k] >= 0) {
            row_compute(o, gi[k], rr[k]);
            row_store<LPR>(o, (uint64_t)r2[k], l, rr[k], table, state0, state1, prev_time);
          }
        }
      }
    }
  }
}

// Long runs (listed in span_list by the tile they start in): tail[t0] + head[t0+1] + head[t0+2] ...
// With power-law keys most long runs are a few tiles long while a handful (the rows of 3- or
// 10-row tables) span hundreds of tiles.  seg_combine_kernel gives one lane group to each run: it
// measures the run (how many following tiles begin with the same row) and adds the head partials
// in order, 8 reads in flight; runs of more than kCombBigTiles tiles are parked in big_list and
// taken by seg_combine_big_kernel, one 1024-thread workgroup per run: group q adds heads q,
// q+GPB, ...; the GPB sums are added in the fixed order q = 0..GPB-1.  Both orders are fixed, so
// the result does not depend on scheduling.
constexpr int kCombBigTiles = 64;
constexpr int kCombBlock = 1024;
constexpr int kCombBigChunk = 2048;  // tile partials one workgroup of the big kernel adds

template <int LPR, typename OffT, typename SortK>
__global__ void __launch_bounds__(kBlock)
    seg_combine_kernel(size_t buckets, const OffT* __restrict__ row_offset,
                       const SortK* __restrict__ sorted_rows, OptConst o,
                       float* __restrict__ table, float* __restrict__ state0,
                       float* __restrict__ state1, unsigned long long* __restrict__ prev_time,
                       const float* __restrict__ head, const float* __restrict__ tail,
                       const uint32_t* __restrict__ span_list, uint32_t* __restrict__ span_count,
                       uint32_t* __restrict__ big_list, size_t big_stride,
                       const uint32_t* __restrict__ n_live) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int CU = 8;
  constexpr unsigned long long kGroupMask = LPR >= 64 ? ~0ull : ((1ull << LPR) - 1ull);
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const int gshift = ((threadIdx.x & 63) / LPR) * LPR;
  const size_t nnz = n_live != nullptr ? (size_t)*n_live : (size_t)row_offset[buckets];
  const size_t n_tiles = (nnz + kSegTile - 1) / kSegTile;
  const uint32_t n_span = span_count[0];
  for (size_t si = (size_t)blockIdx.x * GPB + g; si < n_span; si += (size_t)gridDim.x * GPB) {
    const size_t t0 = span_list[si];
    const SortK row = sorted_rows[(t0 + 1) * kSegTile - 1];
    float4 acc = *reinterpret_cast<const float4*>(tail + t0 * D + l * 4);
    size_t n_heads = 0;
    bool parked = false;
    for (;;) {
      const size_t tt = t0 + 1 + n_heads + l;
      const bool match = tt < n_tiles && sorted_rows[tt * kSegTile] == row;
      const unsigned long long gm = (__ballot(match) >> gshift) & kGroupMask;
      const int ld = gm == kGroupMask ? LPR : __ffsll((long long)~gm) - 1;
      n_heads += (size_t)ld;
      if (ld < LPR) break;
      if (n_heads > (size_t)kCombBigTiles) {
        parked = true;
        break;
      }
    }
    if (parked) {
      // a big run: measure it to the end (LPR evenly spaced probes per round; tiles < lo begin
      // with `row`, tile hi does not) and register its chunks of kCombBigChunk tile partials --
      // seg_combine_big_kernel gives every chunk a workgroup of its own
      size_t lo = t0 + 1 + n_heads, hi = n_tiles;
      while (lo < hi) {
        const size_t step = (hi - lo + LPR - 1) / LPR;
        const size_t probe = lo + (size_t)l * step;
        const bool match = probe < hi && sorted_rows[probe * kSegTile] == row;
        const unsigned long long gm = (__ballot(match) >> gshift) & kGroupMask;
        const int m = gm == kGroupMask ? LPR : __ffsll((long long)~gm) - 1;
        if (m == 0) {
          hi = lo;
        } else {
          const size_t first_miss = lo + (size_t)m * step;
          lo = lo + (size_t)(m - 1) * step + 1;
          if (first_miss < hi) hi = first_miss;
        }
      }
      if (l == 0) {
        const size_t n = lo - (t0 + 1);
        const unsigned long long nch = (n + kCombBigChunk - 1) / kCombBigChunk;
        // one 64-bit counter: runs in the upper half, chunks in the lower -- the chunk bases
        // then ascend with the slot numbers (binary search in the big kernel)
        const unsigned long long old = atomicAdd(
            reinterpret_cast<unsigned long long*>(span_count + 2), (1ull << 32) | nch);
        const size_t slot = (size_t)(old >> 32);
        big_list[slot] = (uint32_t)t0;
        big_list[big_stride + slot] = (uint32_t)n;
        big_list[2 * big_stride + slot] = (uint32_t)(old & 0xFFFFFFFFull);
      }
      continue;
    }
    for (size_t i = 0; i < n_heads; i += CU) {
      float4 h[CU];
#pragma unroll
      for (int c = 0; c < CU; c++) {
        const size_t tt = t0 + 1 + (i + c < n_heads ? i + c : i);  // clamp: always a legal read
        h[c] = *reinterpret_cast<const float4*>(head + tt * D + l * 4);
      }
#pragma unroll
      for (int c = 0; c < CU; c++) {
        if (i + c < n_heads) {
          acc.x += h[c].x;
          acc.y += h[c].y;
          acc.z += h[c].z;
          acc.w += h[c].w;
        }
      }
    }
    apply_row_vec4<LPR>(o, (uint64_t)row, l, acc, table, state0, state1, prev_time);
  }
}

template <int LPR, typename OffT, typename SortK>
__global__ void __launch_bounds__(kCombBlock)
    seg_combine_big_kernel(size_t buckets, const OffT* __restrict__ row_offset,
                           const SortK* __restrict__ sorted_rows, OptConst o,
                           float* __restrict__ table, float* __restrict__ state0,
                           float* __restrict__ state1, unsigned long long* __restrict__ prev_time,
                           float* head, const float* __restrict__ tail, uint32_t* big_list,
                           size_t big_stride, const uint32_t* __restrict__ span_count) {
  // Work item = one chunk (kCombBigChunk tile partials) of one big run.  A row with a million
  // gradients is 30 000 partials: one workgroup adding them all was the tail of the whole update
  // (a single CU's bandwidth); now its chunks run side by side.  Every chunk sum has a fixed order
  // (group q adds partials q, q + GPB, ...; the GPB group sums are added q = 0..GPB-1), a chunk's
  // sum is parked in the slot of its own first partial, and the workgroup that finishes LAST (a
  // counter per run) adds tail + chunk sums in chunk order and applies the optimizer: the result
  // does not depend on which workgroup that is.
  constexpr int D = LPR * 4;
  constexpr int GPB = kCombBlock / LPR;
  constexpr int CU = 8;
  __shared__ float4 part[kCombBlock];
  __shared__ int is_last;
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  const unsigned long long ctr = *reinterpret_cast<const unsigned long long*>(span_count + 2);
  const uint32_t n_big = (uint32_t)(ctr >> 32);
  const uint32_t total = (uint32_t)(ctr & 0xFFFFFFFFull);
  const uint32_t* big_t0 = big_list;
  const uint32_t* big_len = big_list + big_stride;
  const uint32_t* big_base = big_list + 2 * big_stride;
  uint32_t* big_done = big_list + 3 * big_stride;
  for (uint32_t w = blockIdx.x; w < total; w += gridDim.x) {
    uint32_t lo = 0, hi = n_big;  // the run whose chunks include w: last slot with base <= w
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (big_base[mid] <= w) lo = mid;
      else hi = mid;
    }
    const uint32_t slot = lo;
    const size_t t0 = big_t0[slot];
    const size_t n_heads = big_len[slot];
    const uint32_t c = w - big_base[slot];
    const uint32_t nch = (uint32_t)((n_heads + kCombBigChunk - 1) / kCombBigChunk);
    const SortK row = sorted_rows[(t0 + 1) * kSegTile - 1];
    const size_t h0 = (size_t)c * kCombBigChunk;
    const size_t h1 = h0 + kCombBigChunk < n_heads ? h0 + kCombBigChunk : n_heads;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = h0 + (size_t)g; i < h1; i += (size_t)GPB * CU) {
      float4 h[CU];
#pragma unroll
      for (int k = 0; k < CU; k++) {
        const size_t ii = i + (size_t)k * GPB;
        const size_t tt = t0 + 1 + (ii < h1 ? ii : i);
        h[k] = *reinterpret_cast<const float4*>(head + tt * D + l * 4);
      }
#pragma unroll
      for (int k = 0; k < CU; k++) {
        if (i + (size_t)k * GPB < h1) {
          acc.x += h[k].x;
          acc.y += h[k].y;
          acc.z += h[k].z;
          acc.w += h[k].w;
        }
      }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0) {
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nch == 1u) tot = *reinterpret_cast<const float4*>(tail + t0 * D + l * 4);
#pragma unroll 8
      for (int q = 0; q < GPB; q++) {
        const float4 pq = part[q * LPR + l];
        tot.x += pq.x;
        tot.y += pq.y;
        tot.z += pq.z;
        tot.w += pq.w;
      }
      if (nch == 1u) {
        apply_row_vec4<LPR>(o, (uint64_t)row, l, tot, table, state0, state1, prev_time);
      } else {  // every partial of this chunk has been read (the barrier above): reuse slot h0
        *reinterpret_cast<float4*>(head + (t0 + 1 + h0) * D + l * 4) = tot;
        __threadfence();
      }
    }
    __syncthreads();
    if (nch > 1u) {
      if (threadIdx.x == 0) is_last = atomicAdd(big_done + slot, 1u) == nch - 1u ? 1 : 0;
      __syncthreads();
      if (is_last != 0) {
        if (g == 0) {
          __threadfence();
          float4 tot = *reinterpret_cast<const float4*>(tail + t0 * D + l * 4);
          for (uint32_t c2 = 0; c2 < nch; c2++) {
            float* p = head + (t0 + 1 + (size_t)c2 * kCombBigChunk) * D + l * 4;
            // (sums other workgroups parked: read past this CU's vector cache)
            tot.x += __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot.y += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot.z += __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tot.w += __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          apply_row_vec4<LPR>(o, (uint64_t)row, l, tot, table, state0, state1, prev_time);
        }
        if (threadIdx.x == 0) big_done[slot] = 0u;  // clean for the next update
      }
      __syncthreads();
    }
  }
}

// ---- hot rows of a one-hot batch ----------------------------------------------------------------
// Power-law batches are bimodal: a few thousand rows -- the rows of the tiny tables and the heads
// of the big ones, which the table handed out first and which therefore carry the LOWEST row
// numbers -- take two thirds of a batch's positions (Criteo-1TB shape, alpha 1.1: rows < 8192 take
// 71 % of the 1.7 M positions).  Sending those positions through a global radix sort only to cut
// the result into tiles again is what made the update latency-bound.  For a batch with one key per
// bucket (device flag of the index stage) the hot positions never enter the sort:
//   * position p belongs to stream p % G (G = slots per sample of a sample-major batch: all
//     positions of a stream come from ONE table, so its hot rows recur inside the stream; G = 1:
//     the positions as they lie) and a chunk is kHotChunk consecutive positions of one stream;
//   * hot_sort_kernel, one workgroup per chunk: the chunk's positions whose row is < H are sorted
//     by row inside LDS (two stable 7-bit passes, ascending position inside a row) and cut into
//     tiles of 32 like the global list is;
//   * hot_reduce_kernel, one lane group per tile (all chunks' tiles in one flat list): every run
//     (= one row's gradients inside the chunk) is summed in ascending position order; the pieces
//     of runs that cross tile borders are added in tile order by hot_join_kernel.  A chunk's
//     partial sums land in a pool (the far end of gsum, which the sorted list cannot reach: cold
//     pairs + hot partials <= nnz) and loc[row][chunk] says where;
//   * hot_apply_kernel, one lane group per hot row: partials in ascending chunk order, then the
//     optimizer -- a fixed association, so the result does not depend on scheduling;
//   * the sort's first pass leaves out keys < H (RsFirst::skip_below); sort and segmented reduce
//     of the cold pairs (n_live) run on a side stream next to the hot rows' kernels.
// A batch that is not one-hot (flag 0) makes these kernels exit and the sort keeps every pair.
// Measured (MI355X, Criteo-1TB shape, round 4): one workgroup doing sort AND reduce of its chunk
// (8 tiles per lane group, one after the other) took 132 us for 310 MB -- a latency chain on 3
// waves per SIMD; hence the flat tile list.
constexpr int kHotChunk = 4096;
constexpr int kHotBlock = 512;
constexpr int kHotWaves = kHotBlock / 64;
constexpr int kHotRounds = kHotChunk / kHotBlock;  // entries per thread
constexpr int kHotBits = 7;                        // digit of one LDS pass; two passes
constexpr int kHotBins = 1 << kHotBits;
constexpr int kHotMaxRows = 1 << (2 * kHotBits);   // 16384
constexpr int kHotTile = 32;
constexpr int kHotTiles = kHotChunk / kHotTile;    // 128
constexpr int kHotPosBits = 12;                    // entry = row << 12 | position inside the chunk
static_assert((1 << kHotPosBits) == kHotChunk, "entry layout");
constexpr uint32_t kHotNone = 0xFFFFu;             // loc[][]: the row has no partial in this chunk
constexpr uint32_t kHotMaxStreams = 64;            // more slots per sample: the plain path

struct HotGeom {
  uint32_t n;          // positions (= buckets: one key each)
  uint32_t G;          // streams
  uint32_t cpg;        // chunks per stream
  uint32_t rows;       // H
  uint32_t map_inner, map_outer;  // gradient row of bucket u (SparseUpdater::map_inner)
  uint32_t loc_stride;  // chunks the loc table has room for, per row
};

// lanes of this wavefront whose digit equals mine (valid lanes only)
__device__ __forceinline__ unsigned long long hot_match(uint32_t d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < kHotBits; bit++) {
    const bool one = ((d >> bit) & 1u) != 0u;
    const unsigned long long bal = __ballot(one);
    m &= one ? bal : ~bal;
  }
  return valid ? m : 0ull;
}

// One stable LDS split of the workgroup's entries by the digit (e >> shift) & 127.  "Wavefront,
// then round, then lane" is the input order (wavefront w holds entries [w * 512, (w + 1) * 512) of
// it), and ranks are handed out in that nesting.  Returns the number of valid entries.
__device__ __forceinline__ uint32_t hot_lds_pass(const uint32_t (&e)[kHotRounds], uint32_t vmask,
                                                 int shift, uint32_t* __restrict__ dst,
                                                 uint32_t (*wh)[kHotBins], uint32_t* scan_smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kHotWaves * kHotBins; i += kHotBlock) (&wh[0][0])[i] = 0u;
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t info[kHotRounds];  // rank inside the match group | group size << 8
#pragma unroll
  for (int r = 0; r < kHotRounds; r++) {
    const bool valid = ((vmask >> r) & 1u) != 0u;
    const uint32_t d = (e[r] >> shift) & (kHotBins - 1);
    const unsigned long long m = hot_match(d, valid);
    const uint32_t rank = (uint32_t)__popcll(m & lt), cnt = (uint32_t)__popcll(m);
    info[r] = rank | (cnt << 8);
    if (valid && rank == 0u) atomicAdd(&wh[wave][d], cnt);
  }
  __syncthreads();
  uint32_t c = 0u;
  if (threadIdx.x < kHotBins) {
#pragma unroll
    for (int w = 0; w < kHotWaves; w++) c += wh[w][threadIdx.x];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<uint32_t, kHotBlock>(c, scan_smem, &total);
  if (threadIdx.x < kHotBins) {
#pragma unroll
    for (int w = 0; w < kHotWaves; w++) {
      const uint32_t cw = wh[w][threadIdx.x];
      wh[w][threadIdx.x] = run;
      run += cw;
    }
  }
  __syncthreads();
  volatile uint32_t* cur = wh[wave];
#pragma unroll
  for (int r = 0; r < kHotRounds; r++) {
    const bool valid = ((vmask >> r) & 1u) != 0u;
    const uint32_t d = (e[r] >> shift) & (kHotBins - 1);
    uint32_t first = 0u;
    if (valid) first = cur[d];                        // every lane of the match group reads ...
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const uint32_t rank = info[r] & 0xFFu;
      if (rank == 0u) cur[d] = first + (info[r] >> 8);  // ... before its leader advances
      dst[first + rank] = e[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  return total;
}

// per-chunk results of hot_sort_kernel
struct HotBufs {
  uint32_t* S;       // [chunks][kHotChunk] the chunk's hot entries, sorted by row
  uint32_t* meta;    // [chunks][2] entries, first pool slot
  uint32_t* tpref;   // [chunks][kHotTiles + 1] run starts in front of a tile
  uint32_t* items;   // tiles that hold entries: chunk * kHotTiles + tile (any order)
  uint32_t* loc_blk;  // [hot rows] bit b: some chunk in [32 b, 32 b + 32) holds a partial
  uint32_t* joins;   // [.][3] runs that cross tile borders: chunk << 14 | first tile << 7 | last
                     //        tile, partial number, row
  uint32_t* counts;  // this update's counters: [0] pool slots taken, [1] items, [2] joins
  uint32_t* counts_next;  // the next update's (the other parity): zeroed by hot_sort_kernel
  uint16_t* loc;     // [hot rows][loc_stride] partial number of (row, chunk), kHotNone = none
  float* head;       // [chunks * kHotTiles][D] partial of the run that enters a tile
  float* tail;       // [chunks * kHotTiles][D] partial of the run that leaves a tile (its owner's)
};

// one workgroup per chunk: the chunk's hot entries sorted by row (LDS), run starts per tile, a
// block of pool slots for its partials, the work lists of hot_reduce_kernel / hot_join_kernel
__global__ void __launch_bounds__(kHotBlock)
    hot_sort_kernel(HotGeom hg, const uint32_t* __restrict__ one_hot,
                    const uint64_t* __restrict__ value_index, HotBufs hb) {
  // (before the flag is looked at: the counters alternate between two sets, and a batch that is not
  //  one-hot must leave the next one a clean set too)
  if (blockIdx.x == 0 && threadIdx.x == 0)
    hb.counts_next[0] = hb.counts_next[1] = hb.counts_next[2] = 0u;
  if (*one_hot == 0u) return;
  __shared__ uint32_t list[2][kHotChunk];
  __shared__ uint32_t wh[kHotWaves][kHotBins];
  __shared__ uint32_t scan_smem[kHotWaves + 1];
  __shared__ uint32_t tile_pref[kHotTiles + 2];  // run starts in front of a tile
  __shared__ uint32_t sh_ibase, sh_jbase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t chunk = blockIdx.x;
  const uint32_t g = chunk / hg.cpg, c = chunk % hg.cpg;
  const uint32_t len_g = hg.n > g ? (hg.n - g + hg.G - 1u) / hg.G : 0u;  // positions of my stream
  const uint32_t c0 = c * (uint32_t)kHotChunk;
  // ---- the chunk's hot entries, sorted by row (stable: ascending position inside a row) -------
  uint32_t e[kHotRounds], vmask = 0u;
#pragma unroll
  for (int r = 0; r < kHotRounds; r++) {
    const uint32_t i = (uint32_t)(wave * (64 * kHotRounds) + r * 64 + lane);
    e[r] = 0u;
    if (c0 + i < len_g) {
      const uint64_t row = value_index[(size_t)(c0 + i) * hg.G + g];
      if (row < (uint64_t)hg.rows) {
        e[r] = ((uint32_t)row << kHotPosBits) | i;
        vmask |= 1u << r;
      }
    }
  }
  const uint32_t nh = hot_lds_pass(e, vmask, kHotPosBits, list[0], wh, scan_smem);
  if (threadIdx.x == 0) hb.meta[2 * chunk] = nh;
  if (nh == 0u) return;  // (uniform: every thread holds the block total)
  vmask = 0u;
#pragma unroll
  for (int r = 0; r < kHotRounds; r++) {
    const uint32_t j = (uint32_t)(wave * (64 * kHotRounds) + r * 64 + lane);
    e[r] = j < nh ? list[0][j] : 0u;
    if (j < nh) vmask |= 1u << r;
  }
  hot_lds_pass(e, vmask, kHotPosBits + kHotBits, list[1], wh, scan_smem);
  const uint32_t* S = list[1];
  const uint32_t nt = (nh + kHotTile - 1) / kHotTile;
  // ---- run starts per tile -> the run number of every tile's first entry ----------------------
#pragma unroll
  for (int r = 0; r < kHotRounds; r++) {
    const uint32_t j = (uint32_t)(wave * (64 * kHotRounds) + r * 64 + lane);
    const bool st = j < nh && (j == 0u || (S[j] >> kHotPosBits) != (S[j - 1] >> kHotPosBits));
    // (every run of the chunk becomes one partial of its row: hot_apply_kernel looks at the
    //  blocks of 32 chunks marked here only)
    if (st) atomicOr(hb.loc_blk + (S[j] >> kHotPosBits), 1u << (chunk >> 5));
    const unsigned long long bal = __ballot(st);
    if (lane == 0) {
      tile_pref[j / kHotTile] = (uint32_t)__popcll(bal & 0xFFFFFFFFull);
      tile_pref[j / kHotTile + 1] = (uint32_t)__popcll(bal >> 32);
    }
    if (j < nh) hb.S[(size_t)chunk * kHotChunk + j] = S[j];
  }
  __syncthreads();
  // a run that leaves tile t and began in it (or exactly at its start) is pieced together by
  // hot_join_kernel: tail of t + the heads of the tiles after it, through the last one it reaches
  uint32_t t2 = 0u, jrow = 0u;
  bool owner = false;
  if (threadIdx.x + 1u < nt) {
    const uint32_t t = threadIdx.x, base = t * kHotTile;
    jrow = S[base + kHotTile - 1u] >> kHotPosBits;
    owner = (S[base + kHotTile] >> kHotPosBits) == jrow &&
            !((S[base] >> kHotPosBits) == jrow && t > 0u && (S[base - 1u] >> kHotPosBits) == jrow);
    if (owner) {  // last entry of the run: the list is sorted by row
      uint32_t lo = base + kHotTile, hi = nh;  // S[lo] belongs to the run, S[hi] (if any) does not
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((S[mid] >> kHotPosBits) == jrow) lo = mid;
        else hi = mid;
      }
      t2 = lo / kHotTile;
    }
  }
  {
    const uint32_t cnt = threadIdx.x < kHotTiles ? tile_pref[threadIdx.x] : 0u;
    uint32_t nr, nj;
    const uint32_t ex = block_exclusive_scan<uint32_t, kHotBlock>(cnt, scan_smem, &nr);
    const uint32_t jx = block_exclusive_scan<uint32_t, kHotBlock>(owner ? 1u : 0u, scan_smem, &nj);
    if (threadIdx.x == 0) {
      // (which block of slots / list entries the chunk gets does not matter)  One 64-bit add
      // takes both: pool slots in the low word, work items in the high word
      const unsigned long long old =
          atomicAdd(reinterpret_cast<unsigned long long*>(hb.counts),
                    ((unsigned long long)nt << 32) | (unsigned long long)nr);
      hb.meta[2 * chunk + 1] = (uint32_t)old;
      sh_ibase = (uint32_t)(old >> 32);
      sh_jbase = nj > 0u ? atomicAdd(hb.counts + 2, nj) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < kHotTiles) {
      hb.tpref[(size_t)chunk * (kHotTiles + 1) + threadIdx.x] = ex;
      if (threadIdx.x < nt) hb.items[sh_ibase + threadIdx.x] = chunk * kHotTiles + threadIdx.x;
      if (owner) {
        uint32_t* jp = hb.joins + 3 * (size_t)(sh_jbase + jx);
        jp[0] = (chunk << 14) | (threadIdx.x << 7) | t2;
        // number of the last run that starts at or before the end of the tile
        jp[1] = ex + cnt - 1u;
        jp[2] = jrow;
      }
    }
    if (threadIdx.x == 0) hb.tpref[(size_t)chunk * (kHotTiles + 1) + kHotTiles] = nr;
  }
}

// one lane group per tile of 32 sorted entries (any chunk): runs summed in ascending position order;
// a run inside the tile is a finished partial of its (row, chunk), the piece of a run that enters
// / leaves the tile goes to head / tail
template <int LPR, typename GradT>
__global__ void __launch_bounds__(kBlock, 5)  // (<= 102 VGPRs: leaves room for the other chain)
    hot_reduce_kernel(HotGeom hg, const uint32_t* __restrict__ one_hot,
                      const GradT* __restrict__ grad, float* __restrict__ pool_end, HotBufs hb) {
  typedef typename Load4<GradT>::raw Raw;
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  // raw fragments in flight per lane: half a tile, 32 VGPRs.  Measured and dropped (round 4): the
  // whole tile in one batch at 4 waves per SIMD, and one 40-word record per tile written by
  // hot_sort_kernel (one fetch instead of item -> entries / counts): both 79 -> 90 us
  constexpr int QB = sizeof(Raw) == 8 ? 16 : 8;
  static_assert(kHotTile % QB == 0, "batches tile the tile");
  if (*one_hot == 0u) return;
  // the tile's entries + the one in front + the one behind, per lane group
  __shared__ uint32_t ent[GPB][kHotTile + 2];
  const int gq = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t n_items = hb.counts[1];
  constexpr uint32_t kNoneRow = 0xFFFFFFFFu;
  for (uint32_t it = blockIdx.x * GPB + gq; it < n_items; it += gridDim.x * GPB) {
    const uint32_t item = hb.items[it];
    const uint32_t chunk = item / kHotTiles, t = item % kHotTiles;
    const uint32_t nh = hb.meta[2 * chunk], pbase = hb.meta[2 * chunk + 1];
    const uint32_t g = chunk / hg.cpg, c0 = (chunk % hg.cpg) * (uint32_t)kHotChunk;
    const uint32_t base = t * kHotTile;
    const uint32_t cnt = nh - base < (uint32_t)kHotTile ? nh - base : (uint32_t)kHotTile;
    const uint32_t* Sg = hb.S + (size_t)chunk * kHotChunk;
    __builtin_amdgcn_wave_barrier();  // (the previous item's reads of ent are done)
    for (int q = l; q < kHotTile + 2; q += LPR) {
      // ent[q] = entry base - 1 + q; outside the chunk's list: no row
      const bool in = (q > 0 || base > 0u) && base + (uint32_t)q < nh + 1u;
      ent[gq][q] = in ? Sg[base + (uint32_t)q - 1u] : kNoneRow;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t* E = ent[gq] + 1;  // E[q]: entry q of the tile
    const uint32_t prev_row = base > 0u ? E[-1] >> kHotPosBits : kNoneRow;
    const uint32_t next_row = base + cnt < nh ? E[cnt] >> kHotPosBits : kNoneRow;
    uint32_t cur = E[0] >> kHotPosBits;
    bool from_prev = cur == prev_row;
    uint32_t ri = hb.tpref[(size_t)chunk * (kHotTiles + 1) + t] - (from_prev ? 1u : 0u);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t slot = (size_t)chunk * kHotTiles + t;
    auto emit_partial = [&](uint32_t row, uint32_t rix, const float4& a) {
      *reinterpret_cast<float4*>(pool_end - ((size_t)(pbase + rix) + 1u) * D + l * 4) = a;
      if (l == 0) hb.loc[(size_t)row * hg.loc_stride + chunk] = (uint16_t)rix;
    };
#pragma unroll
    for (int qb = 0; qb < kHotTile; qb += QB) {
      if ((uint32_t)qb >= cnt) break;
      Raw v[QB];
#pragma unroll
      for (int k = 0; k < QB; k++) {
        const uint32_t q = (uint32_t)(qb + k) < cnt ? (uint32_t)(qb + k) : cnt - 1u;
        const uint32_t u = (c0 + (E[q] & (uint32_t)(kHotChunk - 1))) * hg.G + g;
        const uint32_t b = hg.map_inner ? (u % hg.map_inner) * hg.map_outer + u / hg.map_inner : u;
        v[k] = Load4<GradT>::ld_raw(grad + (size_t)b * D + l * 4);
      }
#pragma unroll
      for (int k = 0; k < QB; k++) {
        const uint32_t q = (uint32_t)(qb + k);
        if (q < cnt) {
          const uint32_t row = E[q] >> kHotPosBits;
          if (row != cur) {  // the run in hand ends here (it may have begun in an earlier tile)
            if (from_prev) *reinterpret_cast<float4*>(hb.head + slot * D + l * 4) = acc;
            else emit_partial(cur, ri, acc);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            cur = row;
            from_prev = false;
            ri++;
          }
          const float4 f = Load4<GradT>::cvt(v[k]);
          acc.x += f.x;
          acc.y += f.y;
          acc.z += f.z;
          acc.w += f.w;
        }
      }
    }
    const bool to_next = cur == next_row;
    if (from_prev) *reinterpret_cast<float4*>(hb.head + slot * D + l * 4) = acc;
    else if (to_next) *reinterpret_cast<float4*>(hb.tail + slot * D + l * 4) = acc;
    else emit_partial(cur, ri, acc);
  }
}

// runs that cross tile borders inside a chunk: tail of the tile they start in + the heads after it
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    hot_join_kernel(HotGeom hg, const uint32_t* __restrict__ one_hot, float* __restrict__ pool_end,
                    HotBufs hb) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int CU = 16;
  if (*one_hot == 0u) return;
  const int gq = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const uint32_t n_joins = hb.counts[2];
  for (uint32_t it = blockIdx.x * GPB + gq; it < n_joins; it += gridDim.x * GPB) {
    const uint32_t w = hb.joins[3 * (size_t)it], ri = hb.joins[3 * (size_t)it + 1];
    const uint32_t row = hb.joins[3 * (size_t)it + 2];
    const uint32_t chunk = w >> 14, t = (w >> 7) & 127u, t2 = w & 127u;
    const size_t slot0 = (size_t)chunk * kHotTiles;
    float4 acc = *reinterpret_cast<const float4*>(hb.tail + (slot0 + t) * D + l * 4);
    for (uint32_t i = t + 1u; i <= t2; i += CU) {
      float4 h[CU];
#pragma unroll
      for (int k = 0; k < CU; k++) {
        const uint32_t tt = i + (uint32_t)k <= t2 ? i + (uint32_t)k : i;
        h[k] = *reinterpret_cast<const float4*>(hb.head + (slot0 + tt) * D + l * 4);
      }
#pragma unroll
      for (int k = 0; k < CU; k++) {
        if (i + (uint32_t)k <= t2) {
          acc.x += h[k].x;
          acc.y += h[k].y;
          acc.z += h[k].z;
          acc.w += h[k].w;
        }
      }
    }
    const uint32_t pbase = hb.meta[2 * chunk + 1];
    *reinterpret_cast<float4*>(pool_end - ((size_t)(pbase + ri) + 1u) * D + l * 4) = acc;
    if (l == 0) hb.loc[(size_t)row * hg.loc_stride + chunk] = (uint16_t)ri;
  }
}

// one lane group per hot row: its partials in ascending chunk order, then the optimizer
constexpr int kHotApplyChunks = 1024;  // first pool slots of the chunks, staged in LDS
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    hot_apply_kernel(HotGeom hg, uint32_t n_chunks, const uint32_t* __restrict__ one_hot,
                     OptConst o, float* __restrict__ table, float* __restrict__ state0,
                     float* __restrict__ state1, unsigned long long* __restrict__ prev_time,
                     const float* __restrict__ pool_end, HotBufs hb) {
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int CU = 8;
  constexpr int LU = 4;  // loc words per lane per trip
  constexpr unsigned long long kGroupMask = LPR >= 64 ? ~0ull : ((1ull << LPR) - 1ull);
  if (*one_hot == 0u) return;
  __shared__ uint32_t cbase[kHotApplyChunks];
  for (uint32_t i = threadIdx.x; i < n_chunks; i += kBlock) cbase[i] = hb.meta[2 * i + 1];
  __syncthreads();
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const int gshift = ((threadIdx.x & 63) / LPR) * LPR;
  for (uint32_t row = blockIdx.x * GPB + g; row < hg.rows; row += gridDim.x * GPB) {
    // blocks of 32 chunks that hold a partial of this row (a row of one stream: one or two)
    uint32_t blk = hb.loc_blk[row];
    __builtin_amdgcn_wave_barrier();  // (every lane of the group has read the word ...)
    if (blk == 0u) continue;
    if (l == 0) hb.loc_blk[row] = 0u;  // (... before it is cleaned for the next update)
    uint16_t* lrow = hb.loc + (size_t)row * hg.loc_stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    while (blk != 0u) {
      const uint32_t b32 = (uint32_t)__ffs((int)blk) - 1u;
      blk &= blk - 1u;
      for (uint32_t cb = b32 * 32u; cb < b32 * 32u + 32u && cb < n_chunks; cb += LPR * LU) {
        uint32_t v[LU];
#pragma unroll
        for (int u = 0; u < LU; u++) {
          const uint32_t cc = cb + (uint32_t)(u * LPR + l);
          v[u] = (cc < n_chunks && cc < b32 * 32u + 32u) ? (uint32_t)lrow[cc] : kHotNone;
        }
#pragma unroll
        for (int u = 0; u < LU; u++) {
          const uint32_t cc = cb + (uint32_t)(u * LPR + l);
          unsigned long long m = (__ballot(v[u] != kHotNone) >> gshift) & kGroupMask;
          uint32_t myslot = 0u;
          if (v[u] != kHotNone) {
            lrow[cc] = (uint16_t)kHotNone;  // clean for the next update
            myslot = cbase[cc] + v[u];
          }
          while (m != 0ull) {
            uint32_t slot[CU];
            int nk = 0;
#pragma unroll
            for (int k = 0; k < CU; k++) {
              int bit = 0;
              if (m != 0ull) {
                bit = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                nk = k + 1;
              }
              slot[k] = (uint32_t)__shfl((int)myslot, gshift + bit, 64);
            }
            float4 h[CU];
#pragma unroll
            for (int k = 0; k < CU; k++) {
              if (k < nk)
                h[k] = *reinterpret_cast<const float4*>(pool_end - ((size_t)slot[k] + 1u) * D + l * 4);
            }
#pragma unroll
            for (int k = 0; k < CU; k++) {
              if (k < nk) {
                acc.x += h[k].x;
                acc.y += h[k].y;
                acc.z += h[k].z;
                acc.w += h[k].w;
              }
            }
            any = true;
          }
        }
      }
    }
    if (any) apply_row_vec4<LPR>(o, (uint64_t)row, l, acc, table, state0, state1, prev_time);
  }
}

// ---- cold rows of a one-key-per-position batch: counted per row, never sorted ---------------------
// The positions the hot-row kernels leave (rows >= H; all of them when the batch's flag says the
// offsets were ragged) used to go through a global radix sort whose only job was to put the
// gradients of a row side by side: 70 us of latency-bound passes for 490 k pairs of which a third
// are alone in their row (Criteo-1TB shape, alpha 1.1: 221 k cold rows, 167 k of them met once,
// 53 k met 2 .. 32 times, 1.2 k more often).  Here the rows are COUNTED instead:
//   * cold_count_kernel, one thread per position: rank = atomicAdd(cnt[row], 1) -- the order of
//     arrival, whatever it is; the position that arrives first announces the row (dlist);
//   * cold_base_kernel, one thread per announced row: a row met once goes to the list of singles
//     with its position; a row met c > 1 times gets c consecutive entries of plist (handed out in
//     any order) and cnt[row] = base | kColdBased; runs of more than short_max go to their own list;
//   * cold_scatter_kernel, one thread per position: plist[base + rank] = position;
//   * cold_reduce_kernel: singles -- gradient row and table row of several rows in flight per lane
//     group, no list at all; short runs -- a lane group sorts the run's positions ASCENDING (LDS
//     rank count: the arrival order never reaches the arithmetic) and adds the gradients in that
//     order, the reference's order (stable sort by row, sparse_optimizer.cu:657-676); long runs -- a
//     workgroup sorts the positions (LDS bitonic up to kColdLds; beyond that it re-derives them in
//     order by scanning the batch's rows), lane groups sum pieces of 32 consecutive entries, the
//     pieces are added in order.  Every association is a function of the run alone, so the result
//     does not depend on scheduling.  The kernel leaves cnt[] zero again.
// The gradient row of position p: the bucket p itself (through the gradient map) when the batch's
// flag says one key per bucket, else the bucket found by a search of the offsets (bkt[]).
constexpr uint32_t kColdBased = 0x80000000u;
constexpr uint32_t kColdNone = 0xFFFFFFFFu;
constexpr int kColdPer = 8;      // positions per thread (count / scatter)
constexpr int kColdLds = 2048;   // longest run sorted inside LDS
constexpr int kColdPiece = 32;   // entries of a long run summed by one lane group at a time

struct ColdGeom {
  uint32_t n;          // positions
  uint32_t hot_rows;   // H (rows below it belong to the hot kernels while the flag is set)
  uint32_t max_vocab;
  uint32_t map_inner, map_outer;
  uint32_t short_max;  // longest run a lane group sorts (cold_reduce_kernel<LPR>: kShortMax)
  int off_is_u32, combiner;
  size_t buckets;
};

struct ColdBufs {
  uint32_t* cnt;      // [max_vocab] zero between updates
  uint32_t* rank;     // [max_nnz] order of arrival of a position inside its row
  uint32_t* plist;    // [max_nnz] positions of the rows met more than once, row by row
  uint32_t* bkt;      // [max_nnz] bucket of a position (ragged batches only)
  uint2* dlist;       // [max_nnz] (row, first position to arrive)
  uint2* singles;     // [max_nnz] (row, position)
  uint4* segs;        // [max_nnz / 2] short runs: row, base, length
  uint4* longs;       // long runs: row, base, length
  // this update's counters, 128 bytes apart (same-word device-scope atomics cost ~11 ns each and
  // words of one line share that queue): [kCcRows] rows announced; [kCcPl] 64 bits: plist entries
  // handed out | long runs << 32; [kCcSs] 64 bits: short runs | singles << 32
  uint32_t* counts;
  uint32_t* counts_next;
};
constexpr int kCcRows = 0, kCcPl = 32, kCcSs = 64, kCcWords = 96;
constexpr int kColdBlock = 1024;  // count / base / scatter: few, large workgroups = few counter atomics
constexpr int kColdBasePer = 4;   // announced rows per thread of cold_base_kernel

// bucket of key position j: the last u with offset[u] <= j (empty buckets skipped)
__device__ __forceinline__ uint32_t cold_bucket_of(const void* ro_v, bool u32, size_t buckets,
                                                   uint32_t j) {
  size_t lo = 0, hi = buckets;  // offset[lo] <= j < offset[hi]
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) >> 1;
    const unsigned long long v = u32 ? (unsigned long long)((const uint32_t*)ro_v)[mid]
                                     : (unsigned long long)((const long long*)ro_v)[mid];
    if (v <= (unsigned long long)j) lo = mid;
    else hi = mid;
  }
  return (uint32_t)lo;
}

__global__ void __launch_bounds__(kColdBlock)
    cold_count_kernel(ColdGeom cg, const uint32_t* __restrict__ one_hot,
                      const void* __restrict__ row_offset, const uint64_t* __restrict__ value_index,
                      ColdBufs cb) {
  __shared__ uint32_t smem[kColdBlock / 64 + 1];
  __shared__ uint32_t sh_base;
  if (blockIdx.x == 0 && threadIdx.x < kCcWords) cb.counts_next[threadIdx.x] = 0u;
  const bool oh = *one_hot != 0u;
  const uint64_t H = oh ? (uint64_t)cg.hot_rows : 0ull;
  const uint32_t p0 = blockIdx.x * (uint32_t)(kColdBlock * kColdPer) + threadIdx.x;
  uint32_t row[kColdPer], rk[kColdPer];
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    const uint32_t p = p0 + (uint32_t)(r * kColdBlock);
    row[r] = kColdNone;
    if (p < cg.n) {
      const uint64_t v = value_index[p];
      if (v >= H && v < (uint64_t)cg.max_vocab) row[r] = (uint32_t)v;
    }
  }
#pragma unroll
  for (int r = 0; r < kColdPer; r++)
    rk[r] = row[r] != kColdNone ? atomicAdd(cb.cnt + row[r], 1u) : 1u;
  uint32_t nlead = 0u;
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    const uint32_t p = p0 + (uint32_t)(r * kColdBlock);
    if (row[r] != kColdNone) {
      cb.rank[p] = rk[r];
      if (!oh) cb.bkt[p] = cold_bucket_of(row_offset, cg.off_is_u32 != 0, cg.buckets, p);
      nlead += rk[r] == 0u ? 1u : 0u;
    }
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan<uint32_t, kColdBlock>(nlead, smem, &tot);
  if (threadIdx.x == 0) sh_base = tot > 0u ? atomicAdd(cb.counts + kCcRows, tot) : 0u;
  __syncthreads();
  ex += sh_base;
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    if (row[r] != kColdNone && rk[r] == 0u)
      cb.dlist[ex++] = make_uint2(row[r], p0 + (uint32_t)(r * kColdBlock));
  }
}

__global__ void __launch_bounds__(kColdBlock) cold_base_kernel(ColdGeom cg, ColdBufs cb) {
  __shared__ unsigned long long smem64[kColdBlock / 64 + 1];
  __shared__ unsigned long long smem64b[kColdBlock / 64 + 1];
  __shared__ unsigned long long sh[2];
  constexpr uint32_t kTrip = (uint32_t)(kColdBlock * kColdBasePer);
  const uint32_t nd = cb.counts[kCcRows];
  for (uint32_t i0 = blockIdx.x * kTrip; i0 < nd; i0 += gridDim.x * kTrip) {
    // thread t takes kColdBasePer CONSECUTIVE rows of the trip (one scan covers them)
    uint2 e[kColdBasePer];
    uint32_t c[kColdBasePer];
#pragma unroll
    for (int k = 0; k < kColdBasePer; k++) {
      const uint32_t i = i0 + threadIdx.x * (uint32_t)kColdBasePer + (uint32_t)k;
      e[k] = make_uint2(0u, 0u);
      if (i < nd) e[k] = cb.dlist[i];
    }
#pragma unroll
    for (int k = 0; k < kColdBasePer; k++) {
      const uint32_t i = i0 + threadIdx.x * (uint32_t)kColdBasePer + (uint32_t)k;
      c[k] = i < nd ? cb.cnt[e[k].x] : 0u;
    }
    // two 64-bit scans: plist entries | long runs << 32, and short runs | singles << 32
    unsigned long long a = 0ull, b = 0ull;
#pragma unroll
    for (int k = 0; k < kColdBasePer; k++) {
      if (c[k] > cg.short_max) a += (unsigned long long)c[k] | (1ull << 32);
      else if (c[k] > 1u) {
        a += (unsigned long long)c[k];
        b += 1ull;
      } else if (c[k] == 1u) b += 1ull << 32;
    }
    unsigned long long ta, tb;
    unsigned long long xa = block_exclusive_scan<unsigned long long, kColdBlock>(a, smem64, &ta);
    unsigned long long xb = block_exclusive_scan<unsigned long long, kColdBlock>(b, smem64b, &tb);
    if (threadIdx.x == 0) {
      sh[0] = ta != 0ull ? atomicAdd(reinterpret_cast<unsigned long long*>(cb.counts + kCcPl), ta) : 0ull;
      sh[1] = tb != 0ull ? atomicAdd(reinterpret_cast<unsigned long long*>(cb.counts + kCcSs), tb) : 0ull;
    }
    __syncthreads();
    xa += sh[0];
    xb += sh[1];
#pragma unroll
    for (int k = 0; k < kColdBasePer; k++) {
      if (c[k] == 1u) {
        cb.singles[(uint32_t)(xb >> 32)] = e[k];
        xb += 1ull << 32;
      } else if (c[k] > 1u) {
        const uint32_t base = (uint32_t)xa;
        cb.cnt[e[k].x] = base | kColdBased;
        const uint4 seg = make_uint4(e[k].x, base, c[k], 0u);
        if (c[k] > cg.short_max) {
          cb.longs[(uint32_t)(xa >> 32)] = seg;
          xa += (unsigned long long)c[k] | (1ull << 32);
        } else {
          cb.segs[(uint32_t)xb] = seg;
          xa += (unsigned long long)c[k];
          xb += 1ull;
        }
      }
    }
    __syncthreads();  // (sh is rewritten by the next trip)
  }
}

__global__ void __launch_bounds__(kColdBlock)
    cold_scatter_kernel(ColdGeom cg, const uint32_t* __restrict__ one_hot,
                        const uint64_t* __restrict__ value_index, ColdBufs cb) {
  const uint64_t H = *one_hot != 0u ? (uint64_t)cg.hot_rows : 0ull;
  const uint32_t p0 = blockIdx.x * (uint32_t)(kColdBlock * kColdPer) + threadIdx.x;
  uint32_t row[kColdPer], w[kColdPer], rk[kColdPer];
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    const uint32_t p = p0 + (uint32_t)(r * kColdBlock);
    row[r] = kColdNone;
    if (p < cg.n) {
      const uint64_t v = value_index[p];
      if (v >= H && v < (uint64_t)cg.max_vocab) row[r] = (uint32_t)v;
    }
  }
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    w[r] = row[r] != kColdNone ? cb.cnt[row[r]] : 0u;
    rk[r] = row[r] != kColdNone ? cb.rank[p0 + (uint32_t)(r * kColdBlock)] : 0u;
  }
#pragma unroll
  for (int r = 0; r < kColdPer; r++) {
    if ((w[r] & kColdBased) != 0u)
      cb.plist[(w[r] & ~kColdBased) + rk[r]] = p0 + (uint32_t)(r * kColdBlock);
  }
}

template <int LPR>
struct ColdShape {
  // LDS per workgroup: three arrays of GPB x kEMax words = 24 KB whatever LPR is
  static constexpr int kEMax = 8 * LPR < 256 ? 8 * LPR : 256;   // entries of a slab of short runs
  static constexpr int kShortMax = kEMax < 32 ? kEMax : 32;     // longest short run
  static constexpr int kNS = kEMax / 32 > 0 ? kEMax / 32 : 1;   // short runs per slab
};

template <int LPR, typename GradT, bool kSgd>
__global__ void __launch_bounds__(kBlock)
    cold_reduce_kernel(ColdGeom cg, const uint32_t* __restrict__ one_hot,
                       const void* __restrict__ row_offset,
                       const uint64_t* __restrict__ value_index, const GradT* __restrict__ grad,
                       OptConst o, float* __restrict__ table, float* __restrict__ state0,
                       float* __restrict__ state1, unsigned long long* __restrict__ prev_time,
                       float* __restrict__ gsum, ColdBufs cb, uint32_t parts) {
  typedef typename Load4<GradT>::raw Raw;
  constexpr int D = LPR * 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int EMAX = ColdShape<LPR>::kEMax;
  constexpr int NSS = ColdShape<LPR>::kNS;
  constexpr int QB = sizeof(Raw) == 8 ? 16 : 8;
  constexpr int NS1 = kSgd ? 8 : 4;  // singles in flight per lane group
  static_assert(GPB * EMAX * 3 <= 6144 && kColdLds <= 6144, "LDS budget");
  __shared__ uint32_t lds[6144];
  __shared__ uint32_t scan_smem[kBlock / 64 + 1];
  const bool oh = *one_hot != 0u;
  const bool mean = cg.combiner == 1 && !oh;  // (one key per bucket: the mean is the sum)
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  auto grad_row = [&](uint32_t p) -> uint32_t {
    if (!oh) return cb.bkt[p];
    return cg.map_inner ? (p % cg.map_inner) * cg.map_outer + p / cg.map_inner : p;
  };
  auto cvt = [&](const Raw& r, uint32_t b) -> float4 {
    return scaled_grad<GradT>(r, mean ? 1 : 0,
                              mean ? bucket_len(row_offset, cg.off_is_u32 != 0, b) : 1);
  };
  auto add = [](float4& a, const float4& f) {
    a.x += f.x;
    a.y += f.y;
    a.z += f.z;
    a.w += f.w;
  };
  // plain SGD: the row of a finished run is completed when the next one finishes (its read
  // travels meanwhile), as in seg_reduce_kernel
  uint32_t pend_row = kColdNone;
  float4 pend_w = make_float4(0.f, 0.f, 0.f, 0.f), pend_d = pend_w;
  auto pend_flush = [&]() {
    if (pend_row != kColdNone) {
      add(pend_w, pend_d);
      *reinterpret_cast<float4*>(table + (size_t)pend_row * D + l * 4) = pend_w;
      pend_row = kColdNone;
    }
  };
  auto emit = [&](uint32_t row, const float4& a) {
    if constexpr (kSgd) {
      pend_flush();
      pend_d.x = -o.lr * (a.x / o.scaler);
      pend_d.y = -o.lr * (a.y / o.scaler);
      pend_d.z = -o.lr * (a.z / o.scaler);
      pend_d.w = -o.lr * (a.w / o.scaler);
      pend_row = row;
      pend_w = *reinterpret_cast<const float4*>(table + (size_t)row * D + l * 4);
    } else {
      apply_row_vec4<LPR>(o, (uint64_t)row, l, a, table, state0, state1, prev_time);
    }
    if (l == 0) cb.cnt[row] = 0u;  // clean for the next update
  };

  // ---- long runs: one workgroup each --------------------------------------------------------------
  const uint32_t n3 = (parts & 4u) ? cb.counts[kCcPl + 1] : 0u;
  for (uint32_t ir = blockIdx.x; ir < n3; ir += gridDim.x) {
    const uint4 e = cb.longs[ir];
    const uint32_t row = e.x, base = e.y, c = e.z;
    __syncthreads();  // (LDS of the part above / of the previous run is no longer read)
    const uint32_t* sp;  // the run's positions, ascending
    if (c <= 512u) {
      // one pass: an entry's place = the number of entries below it (positions are distinct)
      for (uint32_t q = threadIdx.x; q < c; q += (uint32_t)kBlock) lds[2048 + q] = cb.plist[base + q];
      __syncthreads();
      for (uint32_t q = threadIdx.x; q < c; q += (uint32_t)kBlock) {
        const uint32_t mine = lds[2048 + q];
        uint32_t rnk = 0u;
        for (uint32_t t = 0u; t < c; t++) rnk += lds[2048 + t] < mine ? 1u : 0u;
        lds[rnk] = mine;
      }
      __syncthreads();
      sp = lds;
    } else if (c <= (uint32_t)kColdLds) {
      uint32_t N = 1024u;
      while (N < c) N <<= 1;
      for (uint32_t q = threadIdx.x; q < N; q += (uint32_t)kBlock)
        lds[q] = q < c ? cb.plist[base + q] : kColdNone;
      __syncthreads();
      for (uint32_t k = 2u; k <= N; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
          for (uint32_t t = threadIdx.x; t < (N >> 1); t += (uint32_t)kBlock) {
            const uint32_t lo = ((t / j) * (j << 1)) + (t % j), hi = lo + j;
            const bool up = (lo & k) == 0u;
            const uint32_t a = lds[lo], b = lds[hi];
            if ((a > b) == up) {
              lds[lo] = b;
              lds[hi] = a;
            }
          }
          __syncthreads();
        }
      }
      sp = lds;
    } else {
      // longer than the LDS list: the batch's rows are walked in order and the positions of this
      // row written back over the run's (unordered) entries as they come
      uint32_t filled = 0u;
      for (uint32_t q0 = 0u; q0 < cg.n; q0 += (uint32_t)(kBlock * 4)) {
        const uint32_t p = q0 + threadIdx.x * 4u;
        uint32_t m = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (p + (uint32_t)r < cg.n && value_index[p + (uint32_t)r] == (uint64_t)row) m |= 1u << r;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<uint32_t, kBlock>((uint32_t)__popc(m), scan_smem, &tot);
#pragma unroll
        for (int r = 0; r < 4; r++)
          if ((m >> r) & 1u) cb.plist[base + filled + ex++] = p + (uint32_t)r;
        filled += tot;
      }
      __syncthreads();
      sp = cb.plist + base;
    }
    // pieces of kColdPiece entries, lane group by lane group; the sums wait in gsum[base + first
    // entry] (rows of gsum the hot rows' pool cannot reach: cold entries + hot partials <= n)
    const uint32_t np = (c + (uint32_t)kColdPiece - 1u) / (uint32_t)kColdPiece;
    for (uint32_t k = (uint32_t)g; k < np; k += (uint32_t)GPB) {
      const uint32_t q0 = k * (uint32_t)kColdPiece;
      const uint32_t cntp = c - q0 < (uint32_t)kColdPiece ? c - q0 : (uint32_t)kColdPiece;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int qb = 0; qb < kColdPiece; qb += QB) {
        if ((uint32_t)qb >= cntp) break;
        Raw v[QB];
        uint32_t bb[QB];
#pragma unroll
        for (int t = 0; t < QB; t++) {
          const uint32_t q = (uint32_t)(qb + t) < cntp ? (uint32_t)(qb + t) : cntp - 1u;
          bb[t] = grad_row(sp[q0 + q]);
        }
#pragma unroll
        for (int t = 0; t < QB; t++) v[t] = Load4<GradT>::ld_raw(grad + (size_t)bb[t] * D + l * 4);
#pragma unroll
        for (int t = 0; t < QB; t++)
          if ((uint32_t)(qb + t) < cntp) add(acc, cvt(v[t], bb[t]));
      }
      *reinterpret_cast<float4*>(gsum + (size_t)(base + q0) * D + l * 4) = acc;
    }
    __syncthreads();
    if (g == 0) {
      float4 tot = *reinterpret_cast<const float4*>(gsum + (size_t)base * D + l * 4);
      constexpr int CU = 8;
      for (uint32_t k = 1u; k < np; k += (uint32_t)CU) {
        float4 h[CU];
#pragma unroll
        for (int t = 0; t < CU; t++) {
          const uint32_t kk = k + (uint32_t)t < np ? k + (uint32_t)t : k;
          h[t] = *reinterpret_cast<const float4*>(
              gsum + (size_t)(base + kk * (uint32_t)kColdPiece) * D + l * 4);
        }
#pragma unroll
        for (int t = 0; t < CU; t++)
          if (k + (uint32_t)t < np) add(tot, h[t]);
      }
      apply_row_vec4<LPR>(o, (uint64_t)row, l, tot, table, state0, state1, prev_time);
      if (l == 0) cb.cnt[row] = 0u;
    }
  }
  __syncthreads();  // (the long runs' LDS lists are no longer read)

  // ---- rows met once ---------------------------------------------------------------------------
  const uint32_t n1 = (parts & 1u) ? cb.counts[kCcSs + 1] : 0u;
  for (uint32_t i0 = (blockIdx.x * (uint32_t)GPB + (uint32_t)g) * (uint32_t)NS1; i0 < n1;
       i0 += gridDim.x * (uint32_t)(GPB * NS1)) {
    uint32_t row[NS1], b[NS1];
    Raw v[NS1];
    float4 w[NS1];
#pragma unroll
    for (int k = 0; k < NS1; k++) {
      const uint32_t i = i0 + (uint32_t)k < n1 ? i0 + (uint32_t)k : n1 - 1u;
      const uint2 e = cb.singles[i];
      row[k] = e.x;
      b[k] = grad_row(e.y);
    }
#pragma unroll
    for (int k = 0; k < NS1; k++) {
      v[k] = Load4<GradT>::ld_raw(grad + (size_t)b[k] * D + l * 4);
      if constexpr (kSgd) w[k] = *reinterpret_cast<const float4*>(table + (size_t)row[k] * D + l * 4);
    }
#pragma unroll
    for (int k = 0; k < NS1; k++) {
      if (i0 + (uint32_t)k < n1) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        add(a, cvt(v[k], b[k]));
        if constexpr (kSgd) {
          w[k].x += -o.lr * (a.x / o.scaler);
          w[k].y += -o.lr * (a.y / o.scaler);
          w[k].z += -o.lr * (a.z / o.scaler);
          w[k].w += -o.lr * (a.w / o.scaler);
          *reinterpret_cast<float4*>(table + (size_t)row[k] * D + l * 4) = w[k];
        } else {
          apply_row_vec4<LPR>(o, (uint64_t)row[k], l, a, table, state0, state1, prev_time);
        }
        if (l == 0) cb.cnt[row[k]] = 0u;
      }
    }
  }

  // ---- short runs: NSS of them per lane group and trip ------------------------------------------
  {
    uint32_t* raw = lds + (size_t)g * (3 * EMAX);
    uint32_t* srt = raw + EMAX;
    uint32_t* erow = srt + EMAX;
    const uint32_t n2 = (parts & 2u) ? cb.counts[kCcSs] : 0u;
    for (uint32_t i0 = (blockIdx.x * (uint32_t)GPB + (uint32_t)g) * (uint32_t)NSS; i0 < n2;
         i0 += gridDim.x * (uint32_t)(GPB * NSS)) {
      uint32_t srow[NSS], sbase[NSS], soff[NSS + 1];
      soff[0] = 0u;
#pragma unroll
      for (int j = 0; j < NSS; j++) {
        srow[j] = kColdNone;
        sbase[j] = 0u;
        soff[j + 1] = soff[j];
        if (i0 + (uint32_t)j < n2) {
          const uint4 e = cb.segs[i0 + (uint32_t)j];
          srow[j] = e.x;
          sbase[j] = e.y;
          soff[j + 1] = soff[j] + e.z;
        }
      }
      const uint32_t E = soff[NSS];
      __builtin_amdgcn_wave_barrier();  // (the previous trip's reads of the lists are done)
      for (uint32_t q = (uint32_t)l; q < E; q += (uint32_t)LPR) {
        uint32_t bs = sbase[0], of = 0u, rw = srow[0];
#pragma unroll
        for (int j = 1; j < NSS; j++) {
          if (q >= soff[j]) {
            bs = sbase[j];
            of = soff[j];
            rw = srow[j];
          }
        }
        raw[q] = cb.plist[bs + (q - of)];
        erow[q] = rw;
      }
      __builtin_amdgcn_wave_barrier();
      // the run's positions in ascending order: rank = entries of the run below mine
      for (uint32_t q = (uint32_t)l; q < E; q += (uint32_t)LPR) {
        uint32_t s0 = 0u, s1 = soff[1];
#pragma unroll
        for (int j = 1; j < NSS; j++) {
          if (q >= soff[j]) {
            s0 = soff[j];
            s1 = soff[j + 1];
          }
        }
        const uint32_t mine = raw[q];
        uint32_t rnk = 0u;
        for (uint32_t t = s0; t < s1; t++) rnk += raw[t] < mine ? 1u : 0u;
        srt[s0 + rnk] = mine;
      }
      __builtin_amdgcn_wave_barrier();
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t cur = erow[0];
#pragma unroll 1
      for (uint32_t qb = 0; qb < E; qb += (uint32_t)QB) {
        Raw v[QB];
        uint32_t bb[QB];
#pragma unroll
        for (int k = 0; k < QB; k++) {
          const uint32_t q = qb + (uint32_t)k < E ? qb + (uint32_t)k : E - 1u;
          bb[k] = grad_row(srt[q]);
        }
#pragma unroll
        for (int k = 0; k < QB; k++) v[k] = Load4<GradT>::ld_raw(grad + (size_t)bb[k] * D + l * 4);
#pragma unroll
        for (int k = 0; k < QB; k++) {
          const uint32_t q = qb + (uint32_t)k;
          if (q < E) {
            const uint32_t rw = erow[q];
            if (rw != cur) {
              emit(cur, acc);
              acc = make_float4(0.f, 0.f, 0.f, 0.f);
              cur = rw;
            }
            add(acc, cvt(v[k], bb[k]));
          }
        }
      }
      emit(cur, acc);
    }
    if constexpr (kSgd) pend_flush();
  }

}

// HCTR_SORT=rocprim selects the library's one-sweep sort (A/B measurements); default: radix_sort.hip
inline bool use_library_sort() {
  static const bool v = [] {
    const char* e = getenv("HCTR_SORT");
    return e != nullptr && e[0] == 'r';
  }();
  return v;
}

// What the hot / cold kernels of one batch share: geometry, buffers, counter sets.  Built once per
// batch -- by SparseUpdater::prework() right after the index stage (the grouping work needs the
// rows only, not the gradients: hot_sort_kernel and the cold rows' count / base / scatter then run
// on side streams under the dense tower) or by the update itself.
struct PrePlan {
  bool valid = false;  // the grouping kernels of (vi, n, buckets) are enqueued; the reduces are not
  const uint64_t* vi = nullptr;
  size_t n = 0, buckets = 0;
  size_t n_chunks = 0;
  int lpr = 0;
  HotGeom hg;
  HotBufs hb;
  ColdGeom cg;
  ColdBufs cb;
  hipEvent_t ev_hot = nullptr, ev_cold = nullptr;
};

__global__ void __launch_bounds__(kBlock) cold_clear_kernel(ColdBufs cb) {
  const uint32_t nd = cb.counts[kCcRows];
  for (uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x; i < nd;
       i += gridDim.x * (uint32_t)kBlock)
    cb.cnt[cb.dlist[i].x] = 0u;
}

// everything the path asks of a batch except what only the update knows (gradient alignment,
// store-only mode)
inline bool plan_possible(const SparseUpdater& u, size_t buckets, size_t nnz) {
  const int D = u.D, lpr = D / 4;
  const bool lpr_ok = D % 4 == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;
  const uint32_t G = u.hot_streams;
  if (!(u.hot_rows > 0 && u.one_hot_flag != nullptr && G > 0 && G <= kHotMaxStreams &&
        nnz == buckets && nnz >= u.hot_min_n && nnz < 0x7FFFFFF0ull && lpr_ok &&
        u.scale_row_offset == nullptr))
    return false;
  // (a gradient map -- the embedding_collection's transposed read -- is applied by the cold chain
  //  for one-hot batches only, cold_reduce_kernel's grad_row(); a ragged batch with nnz == buckets
  //  would read unmapped rows there: the sorting path, which maps in every case, takes those)
  if (u.map_inner != 0u) return false;
  const size_t per_g = ceil_div<size_t>(nnz, (size_t)G);
  const size_t n_chunks = (size_t)G * ceil_div<size_t>(per_g, (size_t)kHotChunk);
  const char* ip_env = getenv("HCTR_SORT_IN_PLACE");
  return n_chunks <= (size_t)u.hot_chunks_max && n_chunks <= (size_t)kHotApplyChunks &&
         !use_library_sort() && !(ip_env && ip_env[0] == '0');
}

inline int cold_short_max(int lpr) {
  switch (lpr) {
    case 1: return ColdShape<1>::kShortMax;
    case 2: return ColdShape<2>::kShortMax;
    case 4: return ColdShape<4>::kShortMax;
    case 8: return ColdShape<8>::kShortMax;
    case 16: return ColdShape<16>::kShortMax;
    case 32: return ColdShape<32>::kShortMax;
    default: return ColdShape<64>::kShortMax;
  }
}

inline void plan_build(SparseUpdater& u, PrePlan& pp, size_t buckets, size_t nnz, int combiner,
                       bool off_is_u32, const uint64_t* vi) {
  const uint32_t G = u.hot_streams;
  const size_t per_g = ceil_div<size_t>(nnz, (size_t)G);
  const size_t cpg = ceil_div<size_t>(per_g, (size_t)kHotChunk);
  pp.vi = vi;
  pp.n = nnz;
  pp.buckets = buckets;
  pp.n_chunks = (size_t)G * cpg;
  pp.lpr = u.D / 4;
  pp.hg.n = (uint32_t)nnz;
  pp.hg.G = G;
  pp.hg.cpg = (uint32_t)cpg;
  pp.hg.rows = u.hot_rows;
  pp.hg.map_inner = u.map_inner;
  pp.hg.map_outer = u.map_outer;
  pp.hg.loc_stride = u.hot_chunks_max;
  pp.hb.S = u.hot_S;
  pp.hb.meta = u.hot_meta;
  pp.hb.tpref = u.hot_tpref;
  pp.hb.items = u.hot_items;
  pp.hb.loc_blk = u.hot_loc_blk;
  pp.hb.joins = u.hot_joins;
  // counter sets alternate: [0..3] / [4..7]; [8] = pairs the sort kept
  pp.hb.counts = u.hot_counts + 4 * (u.hot_parity & 1u);
  pp.hb.counts_next = u.hot_counts + 4 * ((u.hot_parity + 1u) & 1u);
  pp.hb.loc = u.hot_loc;
  pp.hb.head = u.hot_head;
  pp.hb.tail = u.hot_tail;
  pp.cg.n = (uint32_t)nnz;
  pp.cg.hot_rows = u.hot_rows;
  pp.cg.max_vocab = (uint32_t)u.max_vocab;
  pp.cg.map_inner = u.map_inner;
  pp.cg.map_outer = u.map_outer;
  pp.cg.short_max = (uint32_t)cold_short_max(pp.lpr);
  pp.cg.off_is_u32 = off_is_u32 ? 1 : 0;
  pp.cg.combiner = combiner;
  pp.cg.buckets = buckets;
  pp.cb.cnt = u.cold_cnt;
  pp.cb.rank = u.cold_rank;
  pp.cb.plist = u.cold_plist;
  pp.cb.bkt = u.cold_bkt;
  pp.cb.dlist = (uint2*)u.cold_dlist;
  pp.cb.singles = (uint2*)u.cold_singles;
  pp.cb.segs = (uint4*)u.cold_segs;
  pp.cb.longs = (uint4*)u.cold_longs;
  pp.cb.counts = u.cold_counts + kCcWords * (u.hot_parity & 1u);
  pp.cb.counts_next = u.cold_counts + kCcWords * ((u.hot_parity + 1u) & 1u);
  u.hot_parity++;
}

// the grouping kernels of a planned batch: the hot rows' chunk sort on hs, the cold rows' count /
// base / scatter on cs
inline int plan_launch_grouping(SparseUpdater& u, PrePlan& pp, const void* ro, hipStream_t hs,
                                hipStream_t cs) {
  hipLaunchKernelGGL(hot_sort_kernel, dim3((unsigned)pp.n_chunks), dim3(kHotBlock), 0, hs, pp.hg,
                     u.one_hot_flag, pp.vi, pp.hb);
  HCTR_LAUNCH_CHECK();
  const unsigned pgrid = (unsigned)ceil_div<size_t>(pp.n, (size_t)(kColdBlock * kColdPer));
  const unsigned bgrid = (unsigned)grid_for(pp.n, kColdBlock * kColdBasePer, 256);
  hipLaunchKernelGGL(cold_count_kernel, dim3(pgrid), dim3(kColdBlock), 0, cs, pp.cg,
                     u.one_hot_flag, ro, pp.vi, pp.cb);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(cold_base_kernel, dim3(bgrid), dim3(kColdBlock), 0, cs, pp.cg, pp.cb);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(cold_scatter_kernel, dim3(pgrid), dim3(kColdBlock), 0, cs, pp.cg,
                     u.one_hot_flag, pp.vi, pp.cb);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// a batch whose grouping kernels ran ahead but whose update takes another path (or never comes):
// the per-row words go back to zero
inline int plan_discard(SparseUpdater& u, PrePlan& pp, hipStream_t s) {
  if (!pp.valid) return HCTR_OK;
  HCTR_HIP(hipStreamWaitEvent(s, pp.ev_hot, 0));
  HCTR_HIP(hipStreamWaitEvent(s, pp.ev_cold, 0));
  hipLaunchKernelGGL(cold_clear_kernel, dim3(grid_for(pp.n, kBlock, 1024)), dim3(kBlock), 0, s,
                     pp.cb);
  HCTR_LAUNCH_CHECK();
  pp.valid = false;
  return HCTR_OK;
}

// any D: one wavefront per run, lanes stride over the vector
template <typename OffT, typename SortK, typename GradT>
__global__ void __launch_bounds__(kBlock)
    update_rows_generic_kernel(const uint64_t* __restrict__ d_num_runs,
                               const uint32_t* __restrict__ run_start,
                               const SortK* __restrict__ sorted_rows,
                               const uint32_t* __restrict__ sorted_buckets,
                               const OffT* __restrict__ scale_ro, int combiner, int D,
                               const GradT* __restrict__ grad, OptConst o,
                               float* __restrict__ table, float* __restrict__ state0,
                               float* __restrict__ state1,
                               unsigned long long* __restrict__ prev_time) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  const size_t num_runs = (size_t)*d_num_runs;
  for (size_t r = wave; r < num_runs; r += nwaves) {
    const uint32_t off = run_start[r];
    const uint32_t cnt = run_start[r + 1] - off;
    const uint64_t row = (uint64_t)sorted_rows[off];
    if (row == kNoRow) continue;
    auto grad_of = [&](uint32_t k, int v) -> float {
      const uint32_t b = sorted_buckets[off + k];
      float gv = Load4<GradT>::ld1(grad + (size_t)b * D + v);
      if (combiner == 1) {
        long long n = (long long)scale_ro[b + 1] - (long long)scale_ro[b];
        if (n > 1) {
          const float sc = 1.0f / (float)n;  // even sizes: align2 rule (16-bit scaler)
          gv = Load4<GradT>::rnd(gv * (D % 2 == 0 ? Load4<GradT>::rnd(sc) : sc));
        }
      }
      return gv;
    };
    // A LONG run of a short vector (the wide tables of Wide & Deep: D = 1, a hot row met tens of
    // thousands of times): with one lane per element the wavefront would walk the run with 1 .. 32
    // lanes, two dependent loads per entry (measured: 1.75 ms per step for three D = 1 tables at
    // batch 16384).  Instead 64 / L lane groups (L = D rounded up to a power of two) take every
    // (64 / L)-th entry each, in ascending order, and their partial sums are combined by a fixed
    // butterfly -- an association that depends on the run's length only.  Runs of up to 64 entries
    // keep the plain ascending sum.
    if (D <= 32 && cnt > 64u) {
      int L = 1;
      while (L < D) L <<= 1;
      const int G = 64 / L, g = lane / L, v = lane % L;
      float part = 0.0f;
      if (v < D)
        for (uint32_t k = (uint32_t)g; k < cnt; k += (uint32_t)G) part += grad_of(k, v);
      for (int ofs = 32; ofs >= L; ofs >>= 1) part += __shfl_xor(part, ofs, 64);
      if (g == 0 && v < D) {
        const float gi = part / o.scaler;
        const size_t f = row * (uint64_t)D + v;
        float w = table[f];
        float s0 = needs_s0(o) ? ld_state1(state0, f, o.state_half) : 0.f;
        float s1 = needs_s1(o) ? ld_state1(state1, f, o.state_half) : 0.f;
        unsigned long long pt = needs_pt(o) ? prev_time[f] : 1ull;
        apply_opt(o, gi, w, &s0, &s1, &pt);
        table[f] = w;
        if (needs_s0(o)) st_state1(state0, f, o.state_half, s0);
        if (needs_s1(o)) st_state1(state1, f, o.state_half, s1);
        if (needs_pt(o)) prev_time[f] = pt;
      }
      continue;
    }
    for (int v = lane; v < D; v += 64) {
      float gi = 0.0f;
      for (uint32_t k = 0; k < cnt; k++) gi += grad_of(k, v);
      gi /= o.scaler;
      const size_t f = row * (uint64_t)D + v;
      float w = table[f];
      float s0 = needs_s0(o) ? ld_state1(state0, f, o.state_half) : 0.f;
      float s1 = needs_s1(o) ? ld_state1(state1, f, o.state_half) : 0.f;
      unsigned long long pt = needs_pt(o) ? prev_time[f] : 1ull;
      apply_opt(o, gi, w, &s0, &s1, &pt);
      table[f] = w;
      if (needs_s0(o)) st_state1(state0, f, o.state_half, s0);
      if (needs_s1(o)) st_state1(state1, f, o.state_half, s1);
      if (needs_pt(o)) prev_time[f] = pt;
    }
  }
}

// SGD with atomic_update (opt_sgd_atomic_kernel :564-582): w[idx] += -(lr/scaler) * wgrad[bucket]
template <typename OffT, typename GradT>
__global__ void __launch_bounds__(kBlock)
    sgd_atomic_kernel(size_t buckets, int D, int combiner, const OffT* __restrict__ row_offset,
                      const uint64_t* __restrict__ value_index, const GradT* __restrict__ grad,
                      float lr_scale, float* __restrict__ table,
                      const OffT* __restrict__ scale_ro) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t u = wave; u < buckets; u += nwaves) {
    const long long off = (long long)row_offset[u];
    const int n = (int)((long long)row_offset[u + 1] - off);
    // (the scaling CSR is read for the mean combiner only)
    const int ns = combiner == 1 ? (int)((long long)scale_ro[u + 1] - (long long)scale_ro[u]) : 1;
    float sc = (combiner == 1 && ns > 1) ? 1.0f / (float)ns : 1.0f;
    if (D % 2 == 0) sc = Load4<GradT>::rnd(sc);  // align2 rule (backward_functor.cu:83-104)
    for (int v = lane; v < D; v += 64) {
      float gv = Load4<GradT>::ld1(grad + u * (size_t)D + v);
      if (combiner == 1) gv = Load4<GradT>::rnd(gv * sc);
      const float dw = -lr_scale * gv;
      for (int j = 0; j < n; j++) {
        const uint64_t idx = value_index[off + j];
        if (idx != kInvalidIndex) unsafeAtomicAdd(table + idx * (uint64_t)D + v, dw);
      }
    }
  }
}

// ---- global (whole-table) sweeps ----------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    adam_global_sweep_kernel(size_t n, float beta1, float beta2, float eps, float alpha_t,
                             int state_half, float* __restrict__ m, float* __restrict__ v,
                             float* __restrict__ w) {
  // adam_update_kernel_global :269-288
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    float mi = beta1 * ld_state1(m, i, state_half);
    float vi = beta2 * ld_state1(v, i, state_half);
    st_state1(m, i, state_half, state_store(state_half, mi));
    st_state1(v, i, state_half, state_store(state_half, vi));
    w[i] += -alpha_t * mi / (sqrtf(vi) + eps);
  }
}

__global__ void __launch_bounds__(kBlock)
    momentum_global_sweep_kernel(size_t n, float factor, int state_half, float* __restrict__ mo,
                                 float* __restrict__ w) {
  // momentum_sgd_update_kernel_global :316-329
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    float m = ld_state1(mo, i, state_half);
    m *= factor;
    w[i] += m;
    st_state1(mo, i, state_half, state_store(state_half, m));
  }
}

__global__ void __launch_bounds__(kBlock)
    nesterov_global_sweep_kernel(size_t n, float mu, int state_half, float* __restrict__ accm,
                                 float* __restrict__ w) {
  // nesterov_global_update_kernel_global :333-347
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    float a = ld_state1(accm, i, state_half);
    a *= mu;
    st_state1(accm, i, state_half, state_store(state_half, a));
    w[i] += a * mu;
  }
}

// ---- wgrad materialisation (tests / get_wgrad) --------------------------------------------------
template <typename OffT, typename GradT>
__global__ void __launch_bounds__(kBlock)
    wgrad_kernel(size_t buckets, int D, int combiner, const OffT* __restrict__ row_offset,
                 const GradT* __restrict__ top, GradT* __restrict__ wgrad) {
  const size_t total = buckets * (size_t)D;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    const size_t u = i / D;
    float g = Load4<GradT>::ld1(top + i);
    if (combiner == 1) {
      long long n = (long long)row_offset[u + 1] - (long long)row_offset[u];
      if (n > 1) {
        const float sc = 1.0f / (float)n;
        g = g * (D % 2 == 0 ? Load4<GradT>::rnd(sc) : sc);
      }
    }
    if constexpr (std::is_same<GradT, float>::value) wgrad[i] = g;
    else if constexpr (std::is_same<GradT, __half>::value) wgrad[i] = __float2half_rn(g);
    else wgrad[i] = __float2bfloat16(g);
  }
}


template <typename SortK>
int sort_pairs(void* temp, size_t& temp_bytes, const SortK* kin, SortK* kout, const uint32_t* vin,
               uint32_t* vout, size_t n, int end_bit, hipStream_t s, const RsFirst* first = nullptr) {
  static_assert(sizeof(SortK) == 4, "32-bit sort keys");
  if (temp == nullptr) {  // size query: room for either implementation
    size_t lib = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, lib, kin, kout, vin, vout, n, 0,
                                             (unsigned)end_bit, s, false);
    if (e != hipSuccess) {
      set_error(std::string("rocprim::radix_sort_pairs: ") + hipGetErrorString(e));
      return HCTR_ERR_HIP;
    }
    const size_t own = radix_sort_temp_bytes(n);
    temp_bytes = lib > own ? lib : own;
    return HCTR_OK;
  }
  if (!use_library_sort())
    return radix_sort_pairs_u32(temp, temp_bytes, (const uint32_t*)kin, (uint32_t*)kout, vin, vout,
                                n, end_bit, s, first);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0,
                                           (unsigned)end_bit, s, false);
  if (e != hipSuccess) {
    set_error(std::string("rocprim::radix_sort_pairs: ") + hipGetErrorString(e));
    return HCTR_ERR_HIP;
  }
  return HCTR_OK;
}

// (row, bucket) pairs -> stable radix sort by row (sparse_optimizer.cu:657-676)
// skip_below / n_kept: the hot path's filter (RsFirst); `timed`: stage 2 of the profiler brackets
// the sort here (the hot path brackets the fork .. join on the caller's stream instead)
template <typename OffT, typename SortK>
int sort_stage(SparseUpdater& u, size_t buckets, size_t n, const OffT* ro, const uint64_t* vi,
               hipStream_t s, uint32_t skip_below = 0u, uint32_t* n_kept = nullptr,
               bool timed = true) {
  SortK* kin = (SortK*)u.sort_keys_in;
  SortK* kout = (SortK*)u.sort_keys_out;
  // wavefronts per 64-bucket chunk = the average bucket length (for_each_key_wave): one for
  // one-hot input, 8 for the MLPerf multi-hot shape whose 100-hot table would otherwise be the tail
  const size_t avg = buckets > 0 ? (n + buckets - 1) / buckets : 1;
  const unsigned parts = (unsigned)(avg < 1 ? 1 : (avg > 16 ? 16 : avg));
  // one key per bucket on the host's count AND on the device's word (the index stage checked the
  // offsets): rows and payloads are read in place by the sort's first pass
  RsFirst first;
  first.keys64 = vi;
  first.flag = u.one_hot_flag;
  first.map_inner = u.map_inner;
  first.map_outer = u.map_outer;
  first.skip_below = skip_below;
  first.n_kept = n_kept;
  const char* ip_env = getenv("HCTR_SORT_IN_PLACE");
  const bool in_place = u.one_hot_flag != nullptr && n == buckets && !use_library_sort() &&
                        !(ip_env && ip_env[0] == '0');
  hipLaunchKernelGGL((expand_pairs_kernel<OffT, SortK>), dim3(grid_for(buckets, kBlock), parts),
                     dim3(kBlock), 0, s, buckets, n, ro, vi, kin, u.sort_vals_in, u.span_count,
                     u.map_inner, u.map_outer, in_place ? u.one_hot_flag : nullptr);
  HCTR_LAUNCH_CHECK();
  // end_bit = log2(max_vocab)+1 (sparse_optimizer.cu:663).  The padding key and the key of a
  // position without a row are all ones: inside end_bit bits they are 2^end_bit - 1, above every
  // live row (rows < top <= 2^end_bit - 1), so they sort last without a bit of their own
  int end_bit = 1;
  // (row_bound: the caller may know that only the first row_bound rows of the table exist yet)
  const size_t top = (u.row_bound > 0 && u.row_bound < u.max_vocab) ? u.row_bound : u.max_vocab;
  while (end_bit < (int)sizeof(SortK) * 8 && ((size_t)1 << end_bit) <= top) end_bit++;
  size_t tb = u.sort_temp_bytes;
  if (u.prof && timed) u.prof->begin(2, s);
  HCTR_TRY(sort_pairs<SortK>(u.sort_temp, tb, kin, kout, u.sort_vals_in, u.sort_vals_out, n,
                             end_bit, s, in_place ? &first : nullptr));
  if (u.prof && timed) u.prof->end(2, s);
  return HCTR_OK;
}

template <typename OffT, typename SortK, typename GradT>
int update_typed(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const OffT* ro,
                 const uint64_t* vi, const GradT* grad, const OptState& opt, float* table,
                 float* state0, float* state1, uint64_t* prev_time, hipStream_t s) {
  const int D = u.D;
  const OffT* sro = u.scale_row_offset ? (const OffT*)u.scale_row_offset : ro;
  OptConst o;
  o.optimizer = opt.optimizer;
  o.update_type = opt.update_type;
  o.lr = opt.lr;
  o.beta1 = opt.beta1;
  o.beta2 = opt.beta2;
  o.epsilon = opt.epsilon;
  o.mf = opt.momentum_factor;
  o.scaler = opt.scaler;
  o.times = opt.times;
  // AdamOptHyperParams::bias() (optimizer.hpp:58-60): double pow, rounded to float, times lr
  const float bias = (float)(std::sqrt(1.0 - std::pow((double)opt.beta2, (double)opt.times)) /
                             (1.0 - std::pow((double)opt.beta1, (double)opt.times)));
  o.alpha_t = opt.lr * bias;
  o.alpha_t_common = opt.lr / (1.0f - opt.beta1);
  o.ftrl_l1 = opt.ftrl_lambda1;
  o.ftrl_l2b = opt.ftrl_lambda2 + opt.ftrl_beta / opt.lr;
  o.state_half = opt.state_half;
  // Global update types sweep the table (sparse_optimizer.cu:269-347 run over all
  // max_vocabulary_size_per_gpu rows).  A row that was never handed out has zero state,
  // and zero state is a fixed point of every sweep (m = v = 0 stay 0, w += -alpha * 0 / (0 + eps)
  // leaves w's bits alone): sweeping the rows handed out so far -- row_bound, the same upper bound
  // the sort's key width uses -- gives the identical table for a fraction of the traffic while a
  // table fills up (DeepFM / Criteo-Kaggle, 33.7 M rows x 16: 12.9 GB per step down to the live rows).
  const size_t live_rows = (u.row_bound > 0 && u.row_bound < u.max_vocab) ? u.row_bound : u.max_vocab;
  const size_t table_elems = live_rows * (size_t)D;

  if (u.map_inner != 0u) {
    if (combiner != 0 || u.ext_rows != nullptr || (opt.optimizer == HCTR_OPT_SGD && opt.atomic_update) ||
        (size_t)u.map_inner * u.map_outer != buckets) {
      set_error("gradient map: sum combiner, sorted update, samples * lookups == buckets only");
      return HCTR_ERR_INVALID_ARG;
    }
  }

  if (opt.optimizer == HCTR_OPT_SGD && opt.atomic_update) {
    const float lr_scale = opt.lr / opt.scaler;
    hipLaunchKernelGGL((sgd_atomic_kernel<OffT, GradT>), dim3(grid_for(buckets * 64, kBlock)),
                       dim3(kBlock), 0, s, buckets, D, combiner, ro, vi, grad, lr_scale, table, sro);
    HCTR_LAUNCH_CHECK();
    return HCTR_OK;
  }

  if (opt.optimizer == HCTR_OPT_NESTEROV && opt.update_type == HCTR_UPDATE_GLOBAL) {
    hipLaunchKernelGGL(nesterov_global_sweep_kernel, dim3(grid_for(table_elems, kBlock)),
                       dim3(kBlock), 0, s, table_elems, opt.momentum_factor, opt.state_half, state0,
                       table);
    HCTR_LAUNCH_CHECK();
  }

  if (nnz > 0) {
    SortK* kout = (SortK*)u.sort_keys_out;
    const uint32_t* vout = u.sort_vals_out;
    bool need_sort = false;
    if (u.ext_rows != nullptr) {
      // presorted by the caller: only the long-run counters need a reset
      static_assert(sizeof(SortK) == 4, "presorted lists carry 32-bit rows");
      kout = (SortK*)const_cast<uint32_t*>(u.ext_rows);
      vout = u.ext_buckets;
      HCTR_HIP(hipMemsetAsync(u.span_count, 0, 4 * sizeof(uint32_t), s));
    } else if (u.early_n >= nnz && u.early_vi == vi && u.early_buckets == buckets) {
      // (row, bucket) pairs of this batch were sorted on the side stream right after the index
      // stage (SparseUpdater::presort); padding keys sit behind the live ones
      HCTR_HIP(hipStreamWaitEvent(s, u.ev_sorted, 0));
      nnz = u.early_n;
    } else {
      need_sort = true;
    }
    u.early_n = 0;
    const bool a16 = reinterpret_cast<uintptr_t>(grad) % 16 == 0;
    bool done = false;
    // store-only mode: finished runs are written straight to their output row
    float* direct = (opt.optimizer == kOptStoreSum && opt.scaler == 1.0f) ? table : nullptr;
    // ---- hot rows (one key per bucket, device flag): see hot_sort_kernel ------------------------
    const uint32_t* n_live = nullptr;
    hipStream_t ss = s;  // stream of the segmented reduce
    bool hot_taken = false, cold_taken = false;
    int seg_grid_cap = 1 << 20;
    {
      const int lpr = D / 4;
      PrePlan* pp = (PrePlan*)u.pre_plan;
      const bool hot = need_sort && a16 && direct == nullptr && plan_possible(u, buckets, nnz);
      // (prework() only ever runs for an updater whose cold rows are counted)
      const bool pre = hot && pp != nullptr && pp->valid && pp->vi == vi && pp->n == nnz &&
                       pp->buckets == buckets;
      if (pp != nullptr && pp->valid && !pre) HCTR_TRY(plan_discard(u, *pp, s));
      if (hot) HCTR_TRY(u.hot_buffers(s));
      if (hot) {
        pp = (PrePlan*)u.pre_plan;
        const bool cold = u.cold_count && u.cold_cnt != nullptr;
        if (!pre) plan_build(u, *pp, buckets, nnz, combiner, sizeof(OffT) == 4, vi);
        const HotGeom& hg = pp->hg;
        const HotBufs& hb = pp->hb;
        const size_t n_chunks = pp->n_chunks;
        // Two chains side by side: the cold rows (count / base / scatter / reduce -- or, with
        // HCTR_COLD_COUNT=0, the filtering sort and the segmented reduce over what it kept) on the
        // side stream, the hot rows on the caller's; they touch disjoint rows.  Stage 2 of the
        // profiler = fork .. join, all of the update.
        hipStream_t cs = u.hot_serial ? s : u.hot_side;  // the cold chain's stream
        if (u.prof) u.prof->begin(2, s);
        if (cs != s) {
          HCTR_HIP(hipEventRecord(u.ev_fork, s));
          HCTR_HIP(hipStreamWaitEvent(cs, u.ev_fork, 0));
        }
        if (pre) {  // grouped ahead (prework): the reduces wait for their chain's kernels only
          HCTR_HIP(hipStreamWaitEvent(s, pp->ev_hot, 0));
          HCTR_HIP(hipStreamWaitEvent(cs, pp->ev_cold, 0));
          pp->valid = false;
        } else if (cold) {
          HCTR_TRY(plan_launch_grouping(u, *pp, (const void*)ro, s, cs));
          // HCTR_HOT_AFTER_GROUPING=1 (measurements): the hot rows' reduce waits for the cold
          // rows' count / base / scatter -- three latency-bound launches that run at a third of
          // their speed beside a streaming kernel
          static const bool hot_waits = [] {
            const char* e = getenv("HCTR_HOT_AFTER_GROUPING");
            return e != nullptr && e[0] == '1';
          }();
          if (hot_waits && cs != s) {
            HCTR_HIP(hipEventRecord(pp->ev_cold, cs));
            HCTR_HIP(hipStreamWaitEvent(s, pp->ev_cold, 0));
          }
        } else {
          hipLaunchKernelGGL(hot_sort_kernel, dim3((unsigned)n_chunks), dim3(kHotBlock), 0, s, hg,
                             u.one_hot_flag, vi, hb);
          HCTR_LAUNCH_CHECK();
        }
        if (cold) {
          const ColdGeom& cg = pp->cg;
          const ColdBufs& cb = pp->cb;
          // HCTR_COLD_SPLIT=1 (measurements): the three parts of the reduce as three launches
          const char* sp_env = getenv("HCTR_COLD_SPLIT");
          const bool split = sp_env != nullptr && sp_env[0] == '1';
          const char* cg_env = getenv("HCTR_COLD_GRID");
          const int cold_grid = cg_env ? atoi(cg_env) : 2048;
          const bool sgd = o.optimizer == HCTR_OPT_SGD;
#define HCTR_COLD_CASE(LPR_)                                                                      \
  {                                                                                               \
    for (uint32_t part = split ? 1u : 7u; part <= 7u; part = split && part < 4u ? part << 1 : 8u) { \
      if (sgd)                                                                                    \
        hipLaunchKernelGGL((cold_reduce_kernel<LPR_, GradT, true>), dim3(cold_grid), dim3(kBlock), \
                           0, cs, cg, u.one_hot_flag, (const void*)ro, vi, grad, o, table, state0, \
                           state1, (unsigned long long*)prev_time, u.gsum, cb, part);             \
      else                                                                                        \
        hipLaunchKernelGGL((cold_reduce_kernel<LPR_, GradT, false>), dim3(cold_grid),             \
                           dim3(kBlock), 0, cs, cg, u.one_hot_flag, (const void*)ro, vi, grad, o, \
                           table, state0, state1, (unsigned long long*)prev_time, u.gsum, cb,     \
                           part);                                                                 \
      HCTR_LAUNCH_CHECK();                                                                        \
    }                                                                                             \
  }
          switch (lpr) {
            case 1: HCTR_COLD_CASE(1) break;
            case 2: HCTR_COLD_CASE(2) break;
            case 4: HCTR_COLD_CASE(4) break;
            case 8: HCTR_COLD_CASE(8) break;
            case 16: HCTR_COLD_CASE(16) break;
            case 32: HCTR_COLD_CASE(32) break;
            default: HCTR_COLD_CASE(64) break;
          }
#undef HCTR_COLD_CASE
          cold_taken = true;
        } else {
          HCTR_TRY((sort_stage<OffT, SortK>(u, buckets, nnz, ro, vi, cs, u.hot_rows,
                                            u.hot_counts + 8, false)));
        }
        float* pool_end = u.gsum + u.max_nnz * (size_t)D;
        const size_t items_max = nnz / kHotTile + n_chunks;
        // the hot rows' reduce is a grid-stride loop over a bounded number of workgroups: a kernel
        // that queues one workgroup per tile fills every wave slot of the device and the other
        // chain only trickles in (measured, round 4: its scatter 21 -> 96 us; 768 workgroups: 60).
        // HCTR_SEG_GRID bounds the sorting path's segmented reduce for measurements.
        const char* hg_env = getenv("HCTR_HOT_GRID");
        const char* sg_env = getenv("HCTR_SEG_GRID");
        const int hot_grid = hg_env ? atoi(hg_env) : 768;
        if (sg_env) seg_grid_cap = atoi(sg_env);
#define HCTR_HOT_CASE(LPR_)                                                                       \
  {                                                                                               \
    constexpr int GPB = kBlock / LPR_;                                                            \
    hipLaunchKernelGGL((hot_reduce_kernel<LPR_, GradT>), dim3(grid_for(items_max, GPB, hot_grid)), \
                       dim3(kBlock), 0, s, hg, u.one_hot_flag, grad, pool_end, hb);               \
    HCTR_LAUNCH_CHECK();                                                                          \
    hipLaunchKernelGGL((hot_join_kernel<LPR_>), dim3(grid_for(items_max / 8 + 1, GPB, 2048)),     \
                       dim3(kBlock), 0, s, hg, u.one_hot_flag, pool_end, hb);                     \
    HCTR_LAUNCH_CHECK();                                                                          \
    hipLaunchKernelGGL((hot_apply_kernel<LPR_>), dim3(grid_for(u.hot_rows, GPB)), dim3(kBlock),   \
                       0, s, hg, (uint32_t)n_chunks, u.one_hot_flag, o, table, state0, state1,    \
                       (unsigned long long*)prev_time, (const float*)pool_end, hb);               \
    HCTR_LAUNCH_CHECK();                                                                          \
  }
        switch (lpr) {
          case 1: HCTR_HOT_CASE(1) break;
          case 2: HCTR_HOT_CASE(2) break;
          case 4: HCTR_HOT_CASE(4) break;
          case 8: HCTR_HOT_CASE(8) break;
          case 16: HCTR_HOT_CASE(16) break;
          case 32: HCTR_HOT_CASE(32) break;
          default: HCTR_HOT_CASE(64) break;
        }
#undef HCTR_HOT_CASE
        n_live = u.hot_counts + 8;
        ss = cs;  // the segmented reduce follows the sort on the side stream
        hot_taken = true;
      } else if (need_sort) {
        HCTR_TRY((sort_stage<OffT, SortK>(u, buckets, nnz, ro, vi, s)));
      }
    }
    if (u.prof && !hot_taken) u.prof->begin(3, s);
    // plain SGD: the apply pass folds into the reduce (seg_reduce_kernel<.., kFuseSgd>);
    // HCTR_SGD_FUSED=0 keeps the two-pass form (measurements, the bit-equality test)
    const char* fuse_env = getenv("HCTR_SGD_FUSED");  // (read per call: tests flip it in-process)
    int fuse = kFuseNone;
    if (!(fuse_env && fuse_env[0] == '0') && direct == nullptr) {
      if (o.optimizer == HCTR_OPT_SGD) fuse = kFuseSgd;
      if (o.optimizer == HCTR_OPT_ADAGRAD) fuse = kFuseAdaGrad;
    }
#define HCTR_SEG_REDUCE(LPR_, FUSE_, OUT_)                                                        \
  hipLaunchKernelGGL((seg_reduce_kernel<LPR_, OffT, SortK, GradT, FUSE_>),                        \
                     dim3(grid_for(seg_tiles, GPB, seg_grid_cap)), dim3(kBlock), 0, ss, buckets,   \
                     ro,                                                                          \
                     kout, vout, combiner, grad, u.gsum, u.seg_head, u.seg_tail, u.span_list,     \
                     u.span_count, OUT_, sro, o, state0, n_live)
#define HCTR_SEG_CASE(LPR_)                                                                       \
  {                                                                                               \
    constexpr int GPB = kBlock / LPR_;                                                            \
    const size_t seg_tiles = ceil_div<size_t>(nnz, (size_t)kSegTile);                             \
    if (fuse == kFuseSgd) HCTR_SEG_REDUCE(LPR_, kFuseSgd, table);                                     \
    else if (fuse == kFuseAdaGrad) HCTR_SEG_REDUCE(LPR_, kFuseAdaGrad, table);                        \
    else HCTR_SEG_REDUCE(LPR_, kFuseNone, direct);                                                    \
    HCTR_LAUNCH_CHECK();                                                                          \
    if (direct == nullptr && fuse == kFuseNone) {                                                 \
      if (o.optimizer == HCTR_OPT_SGD)                                                            \
        hipLaunchKernelGGL((seg_apply_kernel<LPR_, OffT, SortK, true>),                           \
                           dim3(grid_for(nnz, kBlock, 256 * 8)), dim3(kBlock), 0, ss, buckets, ro, \
                           kout, u.gsum, o, table, state0, state1,                                \
                           (unsigned long long*)prev_time, n_live);                               \
      else                                                                                        \
        hipLaunchKernelGGL((seg_apply_kernel<LPR_, OffT, SortK, false>),                          \
                           dim3(grid_for(nnz, kBlock, 256 * 8)), dim3(kBlock), 0, ss, buckets, ro, \
                           kout, u.gsum, o, table, state0, state1,                                \
                           (unsigned long long*)prev_time, n_live);                               \
      HCTR_LAUNCH_CHECK();                                                                        \
    }                                                                                             \
    hipLaunchKernelGGL((seg_combine_kernel<LPR_, OffT, SortK>),                                   \
                       dim3(grid_for(seg_tiles, GPB * 4, 1024)), dim3(kBlock), 0, ss, buckets, ro, \
                       kout, o, table, state0, state1, (unsigned long long*)prev_time, u.seg_head, \
                       u.seg_tail, u.span_list, u.span_count, u.big_list, u.big_stride, n_live);  \
    HCTR_LAUNCH_CHECK();                                                                          \
    hipLaunchKernelGGL((seg_combine_big_kernel<LPR_, OffT, SortK>), dim3(256), dim3(kCombBlock),  \
                       0, ss, buckets, ro, kout, o, table, state0, state1,                         \
                       (unsigned long long*)prev_time, u.seg_head, u.seg_tail, u.big_list,        \
                       u.big_stride, u.span_count);                                               \
  }
    if (cold_taken) {
      done = true;  // (the cold rows' chain above applied its rows itself)
    } else if (a16 && D % 4 == 0) {
      done = true;
      switch (D / 4) {
        case 1: HCTR_SEG_CASE(1) break;
        case 2: HCTR_SEG_CASE(2) break;
        case 4: HCTR_SEG_CASE(4) break;
        case 8: HCTR_SEG_CASE(8) break;
        case 16: HCTR_SEG_CASE(16) break;
        case 32: HCTR_SEG_CASE(32) break;
        case 64: HCTR_SEG_CASE(64) break;
        default: done = false;
      }
    }
#undef HCTR_SEG_CASE
#undef HCTR_SEG_REDUCE
    if (!done) {
      // generic embedding_vec_size: run detection + one wavefront per unique row
      const size_t n_tiles = ceil_div<size_t>(nnz, kTile);
      const int tgrid = (int)(n_tiles < (size_t)kMaxGrid ? n_tiles : (size_t)kMaxGrid);
      hipLaunchKernelGGL((run_count_kernel<OffT, SortK>), dim3(tgrid), dim3(kBlock), 0, s, kout,
                         ro, buckets, n_tiles, u.tile_sums);
      HCTR_LAUNCH_CHECK();
      hipLaunchKernelGGL(scan_tiles_u32_kernel, dim3(1), dim3(1024), 0, s, u.tile_sums, n_tiles,
                         u.d_num_runs);
      HCTR_LAUNCH_CHECK();
      hipLaunchKernelGGL((run_write_kernel<OffT, SortK>), dim3(tgrid), dim3(kBlock), 0, s, kout,
                         ro, buckets, n_tiles, u.tile_sums, u.d_num_runs, u.run_start);
      HCTR_LAUNCH_CHECK();
      hipLaunchKernelGGL((update_rows_generic_kernel<OffT, SortK, GradT>),
                         dim3(grid_for(nnz * 64, kBlock)), dim3(kBlock), 0, s, u.d_num_runs,
                         u.run_start, kout, vout, sro, combiner, D, grad, o, table,
                         state0, state1, (unsigned long long*)prev_time);
    }
    HCTR_LAUNCH_CHECK();
    if (hot_taken) {  // join: the cold chain's end is ordered before whatever follows on s
      if (ss != s) {
        HCTR_HIP(hipEventRecord(u.ev_sorted, ss));
        HCTR_HIP(hipStreamWaitEvent(s, u.ev_sorted, 0));
      }
      if (u.prof) u.prof->end(2, s);
    } else if (u.prof) {
      u.prof->end(3, s);
    }
  }

  if (opt.update_type == HCTR_UPDATE_GLOBAL) {
    if (opt.optimizer == HCTR_OPT_ADAM) {
      hipLaunchKernelGGL(adam_global_sweep_kernel, dim3(grid_for(table_elems, kBlock)),
                         dim3(kBlock), 0, s, table_elems, opt.beta1, opt.beta2, opt.epsilon,
                         o.alpha_t, opt.state_half, state0, state1, table);
      HCTR_LAUNCH_CHECK();
    } else if (opt.optimizer == HCTR_OPT_MOMENTUM_SGD) {
      hipLaunchKernelGGL(momentum_global_sweep_kernel, dim3(grid_for(table_elems, kBlock)),
                         dim3(kBlock), 0, s, table_elems, opt.momentum_factor, opt.state_half,
                         state0, table);
      HCTR_LAUNCH_CHECK();
    }
  }
  return HCTR_OK;
}

template <typename OffT, typename GradT>
int update_sortk(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const OffT* ro,
                 const uint64_t* vi, const GradT* grad, const OptState& opt, float* table,
                 float* s0, float* s1, uint64_t* pt, hipStream_t s) {
  // row indices are sorted as 32-bit keys; create() rejects tables with >= 2^32 rows per GPU
  return update_typed<OffT, uint32_t, GradT>(u, buckets, nnz, combiner, ro, vi, grad, opt, table,
                                             s0, s1, pt, s);
}

template <typename OffT>
int update_grad(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const OffT* ro,
                const uint64_t* vi, const void* grad, int grad_dtype, const OptState& opt,
                float* table, float* s0, float* s1, uint64_t* pt, hipStream_t s) {
  switch (grad_dtype) {
    case HCTR_EMB_F32:
      return update_sortk<OffT, float>(u, buckets, nnz, combiner, ro, vi, (const float*)grad, opt,
                                       table, s0, s1, pt, s);
    case HCTR_EMB_F16:
      return update_sortk<OffT, __half>(u, buckets, nnz, combiner, ro, vi, (const __half*)grad,
                                        opt, table, s0, s1, pt, s);
    case HCTR_EMB_BF16:
      return update_sortk<OffT, __hip_bfloat16>(u, buckets, nnz, combiner, ro, vi,
                                                (const __hip_bfloat16*)grad, opt, table, s0, s1,
                                                pt, s);
  }
  set_error("grad dtype");
  return HCTR_ERR_INVALID_ARG;
}

}  // namespace

// This file is compiled three times (Makefile): HCTR_SU_PART 0 = everything but the segmented
// update's kernel instantiations, 1 / 2 = those for 32-bit / 64-bit row offsets (63 instances of
// seg_reduce_kernel each) -- the three objects build side by side instead of one 3-minute TU.
#ifndef HCTR_SU_PART
#define HCTR_SU_PART 0
#endif
int update_grad_u32(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const uint32_t* ro,
                    const uint64_t* vi, const void* grad, int grad_dtype, const OptState& opt,
                    float* table, float* s0, float* s1, uint64_t* pt, hipStream_t s);
int update_grad_i64(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const long long* ro,
                    const uint64_t* vi, const void* grad, int grad_dtype, const OptState& opt,
                    float* table, float* s0, float* s1, uint64_t* pt, hipStream_t s);
#if HCTR_SU_PART == 1
int update_grad_u32(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const uint32_t* ro,
                    const uint64_t* vi, const void* grad, int grad_dtype, const OptState& opt,
                    float* table, float* s0, float* s1, uint64_t* pt, hipStream_t s) {
  return update_grad<uint32_t>(u, buckets, nnz, combiner, ro, vi, grad, grad_dtype, opt, table, s0,
                               s1, pt, s);
}
#elif HCTR_SU_PART == 2
int update_grad_i64(SparseUpdater& u, size_t buckets, size_t nnz, int combiner, const long long* ro,
                    const uint64_t* vi, const void* grad, int grad_dtype, const OptState& opt,
                    float* table, float* s0, float* s1, uint64_t* pt, hipStream_t s) {
  return update_grad<long long>(u, buckets, nnz, combiner, ro, vi, grad, grad_dtype, opt, table,
                                s0, s1, pt, s);
}
#else

int SparseUpdater::create(size_t max_nnz_, size_t max_vocab_, int D_, bool eager_hot) {
  max_nnz = max_nnz_ > 0 ? max_nnz_ : 1;
  max_vocab = max_vocab_;
  D = D_;
  key32 = true;
  if (max_vocab >= 0xFFFFFFF0ull) {
    set_error("more than 2^32 - 16 rows per GPU are not supported by the sparse update");
    return HCTR_ERR_UNSUPPORTED;
  }
  const size_t ksz = 4;
  HCTR_HIP(hipMalloc(&sort_keys_in, max_nnz * ksz));
  HCTR_HIP(hipMalloc(&sort_keys_out, max_nnz * ksz));
  HCTR_HIP(hipMalloc(&sort_vals_in, max_nnz * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&sort_vals_out, max_nnz * sizeof(uint32_t)));
  size_t tb = 0;
  HCTR_TRY(sort_pairs<uint32_t>(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, nullptr,
                                nullptr, max_nnz, 32, nullptr));
  sort_temp_bytes = tb > 0 ? tb : 16;
  HCTR_HIP(hipMalloc(&sort_temp, sort_temp_bytes));
  HCTR_HIP(hipMalloc(&tile_sums, (ceil_div<size_t>(max_nnz, kTile) + 1) * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&run_start, (max_nnz + 2) * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&d_num_runs, sizeof(uint64_t)));
  HCTR_HIP(hipMemset(d_num_runs, 0, sizeof(uint64_t)));
  {
    int lo = 0, hi = 0;  // hi = numerically lowest = most urgent
    HCTR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* pr = getenv("HCTR_PRESORT_PRIO");  // "low": fill gaps only (measurements)
    HCTR_HIP(hipStreamCreateWithPriority(&side, hipStreamNonBlocking,
                                         (pr && pr[0] == 'l') ? lo : hi));
    HCTR_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HCTR_HIP(hipEventCreateWithFlags(&ev_sorted, hipEventDisableTiming));
  }
  const size_t seg_tiles = ceil_div<size_t>(max_nnz, (size_t)kSegTile) + 1;
  HCTR_HIP(hipMalloc(&seg_head, seg_tiles * (size_t)D * sizeof(float)));
  HCTR_HIP(hipMalloc(&seg_tail, seg_tiles * (size_t)D * sizeof(float)));
  HCTR_HIP(hipMalloc(&gsum, max_nnz * (size_t)D * sizeof(float)));
  HCTR_HIP(hipMalloc(&span_list, seg_tiles * sizeof(uint32_t)));
  // [0] long runs, [1] unused, [2..3] one 64-bit counter: big runs (upper half) / their chunks
  HCTR_HIP(hipMalloc(&span_count, 4 * sizeof(uint32_t)));
  HCTR_HIP(hipMemset(span_count, 0, 4 * sizeof(uint32_t)));
  // per big run: start tile, length in tile partials, first chunk number, finished-chunk counter
  // (the counters start at zero and every update leaves them at zero)
  big_stride = seg_tiles;
  HCTR_HIP(hipMalloc(&big_list, 4 * seg_tiles * sizeof(uint32_t)));
  HCTR_HIP(hipMemset(big_list, 0, 4 * seg_tiles * sizeof(uint32_t)));
  // hot rows of one-hot batches.  HCTR_HOT_ROWS: rows below it are hot (0 = off);
  // HCTR_HOT_MIN: batches with fewer positions keep the plain path (a small batch is launch-bound:
  // two more kernels and a stream fork cost more than its sort)
  {
    const char* hr = getenv("HCTR_HOT_ROWS");
    long rows = hr ? atol(hr) : 8192;
    if (rows < 0) rows = 0;
    if (rows > kHotMaxRows) rows = kHotMaxRows;
    const char* hm = getenv("HCTR_HOT_MIN");
    hot_min_n = hm ? (size_t)atoll(hm) : (size_t)262144;
    hot_rows = 0;
    // (the tables themselves: here for an owner that announces one-hot batches (eager_hot), else by
    //  the first update that takes the path -- hot_buffers(); see sparse_update.h)
    if (rows > 0 && max_nnz >= hot_min_n && D % 4 == 0 && D / 4 <= 64 && ((D / 4) & (D / 4 - 1)) == 0) {
      hot_rows = (uint32_t)rows;
      const char* hs = getenv("HCTR_HOT_SERIAL");  // "1": both chains on the caller's stream (measurements)
      hot_serial = hs != nullptr && hs[0] == '1';
      hot_chunks_max = (uint32_t)(ceil_div<size_t>(max_nnz, (size_t)kHotChunk) + kHotMaxStreams);
      if (eager_hot) {
        HCTR_TRY(hot_buffers(nullptr));
        HCTR_HIP(hipDeviceSynchronize());  // (the clears above ran on the null stream)
      }
    }
  }
  return HCTR_OK;
}

int SparseUpdater::hot_buffers(hipStream_t s) {
  if (hot_loc != nullptr) return HCTR_OK;
  {
    int lo = 0, hi = 0;
    HCTR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const char* hp = getenv("HCTR_HOT_PRIO");  // "high" / "low": priority of the cold chain
    const int pr = (hp && hp[0] == 'h') ? hi : ((hp && hp[0] == 'l') ? lo : 0);
    HCTR_HIP(hipStreamCreateWithPriority(&hot_side, hipStreamNonBlocking, pr));
  }
  const size_t C = hot_chunks_max;
  HCTR_HIP(hipMalloc(&hot_S, C * kHotChunk * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&hot_loc_blk, (size_t)hot_rows * sizeof(uint32_t)));
  HCTR_HIP(hipMemsetAsync(hot_loc_blk, 0, (size_t)hot_rows * sizeof(uint32_t), s));
  HCTR_HIP(hipMalloc(&hot_meta, C * 2 * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&hot_tpref, C * (kHotTiles + 1) * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&hot_items, (max_nnz / kHotTile + C + 1) * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&hot_joins, 3 * C * kHotTiles * sizeof(uint32_t)));
  HCTR_HIP(hipMalloc(&hot_counts, 12 * sizeof(uint32_t)));
  HCTR_HIP(hipMemsetAsync(hot_counts, 0, 12 * sizeof(uint32_t), s));
  const size_t part = C * kHotTiles * (size_t)D * sizeof(float);
  HCTR_HIP(hipMalloc(&hot_head, part));
  HCTR_HIP(hipMalloc(&hot_tail, part));
  {
    const char* cc = getenv("HCTR_COLD_COUNT");
    cold_count = !(cc != nullptr && cc[0] == '0');
  }
  if (cold_count) {
    HCTR_HIP(hipMalloc(&cold_cnt, max_vocab * sizeof(uint32_t)));
    HCTR_HIP(hipMemsetAsync(cold_cnt, 0, max_vocab * sizeof(uint32_t), s));
    HCTR_HIP(hipMalloc(&cold_rank, max_nnz * sizeof(uint32_t)));
    HCTR_HIP(hipMalloc(&cold_plist, max_nnz * sizeof(uint32_t)));
    HCTR_HIP(hipMalloc(&cold_bkt, max_nnz * sizeof(uint32_t)));
    HCTR_HIP(hipMalloc(&cold_dlist, max_nnz * sizeof(uint2)));
    HCTR_HIP(hipMalloc(&cold_singles, max_nnz * sizeof(uint2)));
    HCTR_HIP(hipMalloc(&cold_segs, (max_nnz / 2 + 1) * sizeof(uint4)));
    HCTR_HIP(hipMalloc(&cold_longs, (max_nnz / 2 + 1) * sizeof(uint4)));
    HCTR_HIP(hipMalloc(&cold_counts, 2 * kCcWords * sizeof(uint32_t)));
    HCTR_HIP(hipMemsetAsync(cold_counts, 0, 2 * kCcWords * sizeof(uint32_t), s));
  }
  {
    PrePlan* pp = new PrePlan();
    HCTR_HIP(hipEventCreateWithFlags(&pp->ev_hot, hipEventDisableTiming));
    HCTR_HIP(hipEventCreateWithFlags(&pp->ev_cold, hipEventDisableTiming));
    pre_plan = pp;
  }
  // (last: its presence is what marks the set complete)
  const size_t loc_bytes = (size_t)hot_rows * hot_chunks_max * sizeof(uint16_t);
  HCTR_HIP(hipMalloc(&hot_loc, loc_bytes));
  // kHotNone everywhere; hot_apply keeps it so.  (On the caller's stream: the kernels that follow
  // on it, and on the side stream behind its fork event, see the tables initialised)
  HCTR_HIP(hipMemsetAsync(hot_loc, 0xFF, loc_bytes, s));
  return HCTR_OK;
}

int SparseUpdater::destroy() {
  void* ptrs[] = {sort_keys_in, sort_keys_out, sort_vals_in, sort_vals_out, sort_temp, tile_sums,
                  run_start,    d_num_runs,    seg_head,     seg_tail,      span_list, span_count,
                  gsum,         big_list,      hot_loc,      hot_counts,    hot_head,  hot_tail,
                  hot_S,        hot_meta,      hot_tpref,    hot_items,     hot_joins,
                  hot_loc_blk,  cold_cnt,      cold_rank,    cold_plist,    cold_bkt,
                  cold_dlist,   cold_singles,  cold_segs,    cold_longs,    cold_counts};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (pre_plan) {
    PrePlan* pp = (PrePlan*)pre_plan;
    if (pp->ev_hot) (void)hipEventDestroy(pp->ev_hot);
    if (pp->ev_cold) (void)hipEventDestroy(pp->ev_cold);
    delete pp;
    pre_plan = nullptr;
  }
  if (hot_side) {
    (void)hipStreamSynchronize(hot_side);
    (void)hipStreamDestroy(hot_side);
    hot_side = nullptr;
  }
  if (side) {
    (void)hipStreamSynchronize(side);
    (void)hipStreamDestroy(side);
    (void)hipEventDestroy(ev_fork);
    (void)hipEventDestroy(ev_sorted);
    side = nullptr;
  }
  early_n = 0;
  sort_keys_in = sort_keys_out = sort_temp = nullptr;
  sort_vals_in = sort_vals_out = tile_sums = run_start = nullptr;
  d_num_runs = nullptr;
  seg_head = seg_tail = nullptr;
  span_list = span_count = big_list = nullptr;
  gsum = nullptr;
  hot_loc = nullptr;
  hot_counts = hot_S = hot_meta = hot_tpref = hot_items = hot_joins = hot_loc_blk = nullptr;
  hot_head = hot_tail = nullptr;
  cold_cnt = cold_rank = cold_plist = cold_bkt = cold_counts = nullptr;
  cold_dlist = cold_singles = cold_segs = cold_longs = nullptr;
  hot_rows = hot_chunks_max = 0;
  return HCTR_OK;
}

int SparseUpdater::presort(size_t buckets, size_t n, const void* row_offset, int key_type,
                           const uint64_t* value_index, hipStream_t s) {
  early_n = 0;
  if (buckets == 0 || n == 0 || n > max_nnz || buckets > 0xFFFFFFF0ull || !side) return HCTR_OK;
  HCTR_HIP(hipEventRecord(ev_fork, s));
  HCTR_HIP(hipStreamWaitEvent(side, ev_fork, 0));
  int rc;
  if (key_type == HCTR_KEY_U32)
    rc = sort_stage<uint32_t, uint32_t>(*this, buckets, n, (const uint32_t*)row_offset,
                                        value_index, side);
  else
    rc = sort_stage<long long, uint32_t>(*this, buckets, n, (const long long*)row_offset,
                                         value_index, side);
  if (rc != HCTR_OK) return rc;
  HCTR_HIP(hipEventRecord(ev_sorted, side));
  early_n = n;
  early_vi = value_index;
  early_buckets = buckets;
  return HCTR_OK;
}

int SparseUpdater::prework(size_t buckets, size_t nnz, int combiner, const void* row_offset,
                            int key_type, const uint64_t* value_index, hipStream_t s) {
  PrePlan* pp = (PrePlan*)pre_plan;
  if (pp == nullptr || !cold_count || cold_cnt == nullptr || hot_loc == nullptr || !side ||
      !hot_side || hot_serial)
    return HCTR_OK;
  if (pp->valid) HCTR_TRY(plan_discard(*this, *pp, s));  // (a batch that was never updated)
  if (buckets == 0 || nnz == 0 || nnz > max_nnz || !plan_possible(*this, buckets, nnz))
    return HCTR_OK;
  plan_build(*this, *pp, buckets, nnz, combiner, key_type == HCTR_KEY_U32, value_index);
  // both chains behind what s has enqueued so far (the index stage), next to what follows on it
  HCTR_HIP(hipEventRecord(ev_fork, s));
  HCTR_HIP(hipStreamWaitEvent(side, ev_fork, 0));
  HCTR_HIP(hipStreamWaitEvent(hot_side, ev_fork, 0));
  HCTR_TRY(plan_launch_grouping(*this, *pp, row_offset, side, hot_side));
  HCTR_HIP(hipEventRecord(pp->ev_hot, side));
  HCTR_HIP(hipEventRecord(pp->ev_cold, hot_side));
  pp->valid = true;
  return HCTR_OK;
}

int SparseUpdater::update(size_t buckets, size_t nnz, int combiner, const void* row_offset,
                          int key_type, const uint64_t* value_index, const void* top_grad,
                          int grad_dtype, const OptState& opt, float* table, float* state0,
                          float* state1, uint64_t* prev_time, hipStream_t s) {
  if (buckets == 0) return HCTR_OK;
  if (nnz > max_nnz) {
    set_error("update: nnz exceeds the workspace (batch_size * max_feature_num)");
    return HCTR_ERR_INVALID_ARG;
  }
  if (buckets > 0xFFFFFFF0ull) {
    set_error("update: more than 2^32 buckets");
    return HCTR_ERR_UNSUPPORTED;
  }
  switch (opt.optimizer) {
    case HCTR_OPT_SGD:
    case HCTR_OPT_ADAM:
    case HCTR_OPT_ADAGRAD:
    case HCTR_OPT_MOMENTUM_SGD:
    case HCTR_OPT_NESTEROV:
    case kOptStoreSum: break;
    case HCTR_OPT_FTRL:
      if (allow_ftrl) break;
      [[fallthrough]];
    default:
      // Ftrl / RMSProp are not implemented by the reference's GPU update either
      set_error("sparse optimizer not supported (reference: sparse_optimizer.cu:821-826)");
      return HCTR_ERR_UNSUPPORTED;
  }
  if (opt.update_type == HCTR_UPDATE_LAZY_GLOBAL && opt.optimizer != HCTR_OPT_ADAM) {
    set_error("lazy global update is only implemented for Adam (sparse_optimizer.cu:829-850)");
    return HCTR_ERR_UNSUPPORTED;
  }
  if (key_type == HCTR_KEY_U32)
    return update_grad_u32(*this, buckets, nnz, combiner, (const uint32_t*)row_offset, value_index,
                           top_grad, grad_dtype, opt, table, state0, state1, prev_time, s);
  if (key_type == HCTR_KEY_I64)
    return update_grad_i64(*this, buckets, nnz, combiner, (const long long*)row_offset, value_index,
                           top_grad, grad_dtype, opt, table, state0, state1, prev_time, s);
  set_error("key_type");
  return HCTR_ERR_INVALID_ARG;
}

int materialize_wgrad(size_t buckets, int D, int combiner, const void* ro, int key_type,
                      const void* top, void* wgrad, int dtype, hipStream_t s) {
  if (buckets == 0) return HCTR_OK;
  const int grid = grid_for(buckets * (size_t)D, kBlock);
#define HCTR_WG(OffT, GradT)                                                                  \
  hipLaunchKernelGGL((wgrad_kernel<OffT, GradT>), dim3(grid), dim3(kBlock), 0, s, buckets, D, \
                     combiner, (const OffT*)ro, (const GradT*)top, (GradT*)wgrad)
  if (key_type == HCTR_KEY_U32) {
    if (dtype == HCTR_EMB_F32) HCTR_WG(uint32_t, float);
    else if (dtype == HCTR_EMB_F16) HCTR_WG(uint32_t, __half);
    else HCTR_WG(uint32_t, __hip_bfloat16);
  } else {
    if (dtype == HCTR_EMB_F32) HCTR_WG(long long, float);
    else if (dtype == HCTR_EMB_F16) HCTR_WG(long long, __half);
    else HCTR_WG(long long, __hip_bfloat16);
  }
#undef HCTR_WG
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}
#endif  // HCTR_SU_PART

}  // namespace hctr
