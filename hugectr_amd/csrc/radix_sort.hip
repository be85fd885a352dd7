// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, sized for the
// sparse update's (row, bucket) lists: n <= 2^22 .. 2^24 pairs, keys of <= 32 bits.
//
// The reference sorts with cub::DeviceRadixSort::SortPairs (R/HugeCTR/src/optimizers/
// sparse_optimizer.cu:663-668); the library counterpart here (rocprim::radix_sort_pairs) is a
// one-sweep sort whose decoupled look-back chain makes every digit pass latency-bound at this size
// (~46 us per pass at 1.7 M pairs, 3 passes + a histogram launch = 190 us -- profiles/, round 1).
// This sort has no inter-workgroup dependency inside a launch:
//
//   per 10-bit digit:  rs_hist_kernel     per-tile digit histograms (LDS atomics) -> global
//                      rs_colscan_kernel  exclusive scan over tiles, per digit value
//                      rs_scatter_kernel  stable rank of every key inside its tile + global offset
//
// A tile is 4096 consecutive keys; its four wavefronts own 1024 consecutive keys each, so
// "tile, then wavefront, then round, then lane" IS the input order and ranks assigned in that
// nesting are stable.  Inside a wavefront the lanes that hold the same digit are found with ten
// ballots; their rank is the running per-wavefront counter of that digit (LDS) plus the number of
// lower lanes in the match mask.  Three launches of a few microseconds per digit instead of one
// 46-us launch; keys of b bits take ceil(b / 10) digits.
// The first digit pass can read its keys straight from the index stage's 64-bit rows and make up
// the payload (position, or its gradient-map image) itself -- the sparse update's pair expansion
// drops out for one-hot batches -- and can leave out the keys below a bound (RsFirst).  Measured
// and dropped (round 4, MI355X, 1.7 M pairs): tiles of 2048 / 1024 keys and a 128- / 64-workgroup
// column scan -- every combination slower than 4096-key tiles with a 32-workgroup scan (sort
// 104 us -> 107 .. 137 us).
#include "radix_sort.h"

#include <cstdlib>

#include "block_prims.h"

namespace hctr {
namespace {

constexpr int kRsBlock = 256;
constexpr int kRsWaves = kRsBlock / 64;
constexpr int kRsRounds = 16;                       // keys per lane
constexpr int kRsWaveKeys = 64 * kRsRounds;         // 1024 consecutive keys per wavefront
constexpr int kRsTile = kRsWaveKeys * kRsWaves;     // 4096
// digit width: 10 bits, or 11 when that saves a whole pass (keys of 21-22 bits: a table that has
// handed out fewer than 4 M rows -- two passes instead of three)
constexpr int kRsMaxBits = 11;
constexpr int kRsMaxBins = 1 << kRsMaxBits;
constexpr int kRsScanBlock = 1024;                  // colscan: 32 tile chunks x 32 digit values
constexpr int kRsScanBins = 32;

// key / payload of position i in the first pass (see RsFirst)
struct RsSrc {
  const uint64_t* k64;
  const uint32_t* flag;
  uint32_t map_inner, map_outer;
  // first pass of a one-hot batch: keys below skip_below are left out of the sort altogether (the
  // sparse update's hot rows, summed by hot_chunk_kernel); the pass that filters posts the number
  // of keys it kept to *n_kept, and the later passes take their length from *n_live
  uint32_t skip_below;
  uint32_t* n_kept;
  const uint32_t* n_live;
};
__device__ __forceinline__ bool rs_first64(const RsSrc& f) {
  return f.k64 != nullptr && *f.flag != 0u;
}
__device__ __forceinline__ size_t rs_len(const RsSrc& f, size_t n) {
  return f.n_live != nullptr ? (size_t)*f.n_live : n;
}
__device__ __forceinline__ uint32_t rs_payload(const RsSrc& f, size_t i) {
  const uint32_t u = (uint32_t)i;
  return f.map_inner ? (u % f.map_inner) * f.map_outer + u / f.map_inner : u;
}

// lanes of this wavefront that hold the same digit as mine (valid lanes only): ten ballots
template <int BITS>
__device__ __forceinline__ unsigned long long rs_match(uint32_t d, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < BITS; bit++) {
    const bool one = ((d >> bit) & 1u) != 0u;
    const unsigned long long bal = __ballot(one);
    m &= one ? bal : ~bal;
  }
  return m;
}

template <int BITS>
__global__ void __launch_bounds__(kRsBlock)
    rs_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int shift,
                   uint32_t* __restrict__ hist, RsSrc src) {
  constexpr int kRsBins = 1 << BITS;
  n = rs_len(src, n);
  if ((size_t)blockIdx.x * kRsTile >= n) return;  // (a tile past the live length: colscan skips it)
  const bool f64 = rs_first64(src);
  const uint32_t skip = f64 ? src.skip_below : 0u;
  __shared__ uint32_t h[kRsBins];
  for (int b = threadIdx.x; b < kRsBins; b += kRsBlock) h[b] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * kRsTile + (size_t)wave * kRsWaveKeys + lane;
  uint32_t key[kRsRounds];
#pragma unroll
  for (int r = 0; r < kRsRounds; r++) {
    const size_t i = base + (size_t)r * 64;
    key[r] = i < n ? (f64 ? (uint32_t)src.k64[i] : keys[i]) : 0xFFFFFFFFu;
  }
  // Row ids of a power-law batch are small numbers: in the upper digits most lanes of a wavefront
  // hold the SAME value, and 64 LDS atomics on one address serialise.  The lanes that agree with
  // lane 0 are counted with one ballot and added once; the rest take the plain atomic.
#pragma unroll
  for (int r = 0; r < kRsRounds; r++) {
    const bool valid = base + (size_t)r * 64 < n && key[r] >= skip;
    const uint32_t d = (key[r] >> shift) & (kRsBins - 1);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
    const unsigned long long same = __ballot(valid && d == d0);
    if (valid && d != d0) atomicAdd(&h[d], 1u);
    if (lane == 0 && same != 0ull) atomicAdd(&h[d0], (uint32_t)__popcll(same));
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kRsBins; b += kRsBlock) {
    hist[(size_t)blockIdx.x * kRsBins + b] = h[b];
  }
}

// hist[t][b] <- sum of hist[t'][b] over t' < t, for the 32 digit values of this workgroup
template <int BITS>
__global__ void __launch_bounds__(kRsScanBlock)
    rs_colscan_kernel(uint32_t* __restrict__ hist, size_t tiles, uint32_t* __restrict__ total,
                      const uint32_t* __restrict__ n_live, uint32_t tile_keys) {
  constexpr int kRsBins = 1 << BITS;
  if (n_live != nullptr) tiles = ((size_t)*n_live + tile_keys - 1) / tile_keys;
  constexpr int kChunks = kRsScanBlock / kRsScanBins;
  __shared__ uint32_t part[kChunks][kRsScanBins];
  const int c = threadIdx.x % kRsScanBins, q = threadIdx.x / kRsScanBins;
  const size_t b = (size_t)blockIdx.x * kRsScanBins + c;
  const size_t per = (tiles + kChunks - 1) / kChunks;
  const size_t t0 = (size_t)q * per < tiles ? (size_t)q * per : tiles;
  const size_t t1 = t0 + per < tiles ? t0 + per : tiles;
  uint32_t sum = 0u;
#pragma unroll 4
  for (size_t t = t0; t < t1; t++) sum += hist[t * kRsBins + b];
  part[q][c] = sum;
  __syncthreads();
  uint32_t run = 0u;
  for (int k = 0; k < q; k++) run += part[k][c];
#pragma unroll 4
  for (size_t t = t0; t < t1; t++) {
    const uint32_t v = hist[t * kRsBins + b];
    hist[t * kRsBins + b] = run;
    run += v;
  }
  if (q == kChunks - 1) total[b] = run;  // keys of this digit value in all tiles
}

// MASK: the lanes of a round that share a digit are found through a per-wavefront LDS lane mask
// per digit value (10-bit digits: 32 KB of masks fit beside the tile); otherwise with BITS ballots
// (11-bit digits: the masks would halve the resident workgroups)
template <int BITS, bool MASK>
__global__ void __launch_bounds__(kRsBlock)
    rs_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                      uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, size_t n,
                      int shift, const uint32_t* __restrict__ tile_offs,
                      const uint32_t* __restrict__ total, RsSrc src) {
  constexpr int kRsBins = 1 << BITS;
  n = rs_len(src, n);
  if ((size_t)blockIdx.x * kRsTile >= n) return;
  const bool f64 = rs_first64(src);
  const uint32_t skip = f64 ? src.skip_below : 0u;
  constexpr int kPerThread = kRsBins / kRsBlock;  // digit values per thread in the offset step
  // The tile is first sorted INSIDE LDS (stable, by this digit), then streamed out: consecutive
  // threads write consecutive sorted elements, and elements of one digit value are consecutive
  // in the output too, so the stores of a digit run coalesce.  (Storing each key straight from
  // the ranking loop -- one 4-byte store per lane to 64 unrelated lines -- cost 19 of the 33 us.)
  __shared__ uint32_t wh[kRsWaves][kRsBins];  // per-wavefront digit counters, then local cursors
  __shared__ int32_t delta[kRsBins];          // global position - local position, per digit value
  // used twice: while ranking (MASK), per wavefront one 64-bit lane mask per digit value (the
  // lanes of a round that hold it: built with ds_or, read back, cleared by the group's first
  // lane -- three LDS operations instead of ten ballots and ~80 vector instructions per round);
  // afterwards the tile's sorted keys and values
  constexpr int kStage = (MASK && kRsWaves * kRsBins > kRsTile) ? kRsWaves * kRsBins : kRsTile;
  __shared__ unsigned long long stage[kStage];
  uint32_t* lkey = reinterpret_cast<uint32_t*>(stage);
  uint32_t* lval = lkey + kRsTile;
  volatile unsigned long long* mm = stage + (size_t)(threadIdx.x >> 6) * (MASK ? kRsBins : 0);
  __shared__ uint32_t scan_smem[kRsBlock / 64 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kRsWaves * kRsBins; i += kRsBlock) {
    (&wh[0][0])[i] = 0u;
    if (MASK) stage[i] = 0ull;
  }
  const size_t tile_base = (size_t)blockIdx.x * kRsTile;
  const size_t base = tile_base + (size_t)wave * kRsWaveKeys + lane;
  uint32_t key[kRsRounds];
#pragma unroll
  for (int r = 0; r < kRsRounds; r++) {
    const size_t i = base + (size_t)r * 64;
    key[r] = i < n ? (f64 ? (uint32_t)src.k64[i] : kin[i]) : 0xFFFFFFFFu;
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t info[kRsRounds];  // rank inside the match group | group size << 8
#pragma unroll
  for (int r = 0; r < kRsRounds; r++) {
    const bool valid = base + (size_t)r * 64 < n && key[r] >= skip;
    const uint32_t d = (key[r] >> shift) & (kRsBins - 1);
    unsigned long long m = 0ull;
    if (MASK) {
      // LDS operations of one wavefront complete in issue order; the wave barriers keep the
      // COMPILER from moving the OR, the read-back and the clear across each other (they sit in
      // divergent branches it could otherwise reschedule)
      if (valid) atomicOr(const_cast<unsigned long long*>(&mm[d]), 1ull << lane);
      __builtin_amdgcn_wave_barrier();
      if (valid) m = mm[d];
      __builtin_amdgcn_wave_barrier();
    } else {
      m = rs_match<BITS>(d, valid);
      if (!valid) m = 0ull;
    }
    const uint32_t rank = (uint32_t)__popcll(m & lt), cnt = (uint32_t)__popcll(m);
    info[r] = rank | (cnt << 8);
    if (valid && rank == 0u) {
      if (MASK) mm[d] = 0ull;
      atomicAdd(&wh[wave][d], cnt);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();  // (the masks are all zero again: stage becomes the sorted tile)
  // per digit value (kPerThread per thread): keys of this tile, global first position, local
  // first position
  uint32_t c4[kPerThread], t4[kPerThread], cs = 0u, ts = 0u;
#pragma unroll
  for (int j = 0; j < kPerThread; j++) {
    const int b = threadIdx.x * kPerThread + j;
    c4[j] = 0u;
#pragma unroll
    for (int w = 0; w < kRsWaves; w++) c4[j] += wh[w][b];
    t4[j] = total[b];
    cs += c4[j];
    ts += t4[j];
  }
  uint32_t all, tile_kept;  // keys of this tile that take part / of all tiles
  uint32_t lex = block_exclusive_scan<uint32_t, kRsBlock>(cs, scan_smem, &tile_kept);  // local
  uint32_t gex = block_exclusive_scan<uint32_t, kRsBlock>(ts, scan_smem, &all);        // global
  if (src.n_kept != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *src.n_kept = all;
  const int tile_n = (int)tile_kept;
#pragma unroll
  for (int j = 0; j < kPerThread; j++) {
    const int b = threadIdx.x * kPerThread + j;
    delta[b] = (int32_t)(gex + tile_offs[(size_t)blockIdx.x * kRsBins + b]) - (int32_t)lex;
    uint32_t run = lex;
    lex += c4[j];
    gex += t4[j];
#pragma unroll
    for (int w = 0; w < kRsWaves; w++) {
      const uint32_t c = wh[w][b];
      wh[w][b] = run;
      run += c;
    }
  }
  __syncthreads();
  volatile uint32_t* cur = wh[wave];
#pragma unroll
  for (int r = 0; r < kRsRounds; r++) {
    const size_t i = base + (size_t)r * 64;
    const bool valid = i < n && key[r] >= skip;
    const uint32_t d = (key[r] >> shift) & (kRsBins - 1);
    const uint32_t v = valid ? (f64 ? rs_payload(src, i) : vin[i]) : 0u;
    uint32_t first = 0u;
    if (valid) first = cur[d];                       // every lane of the match group reads ...
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const uint32_t rank = info[r] & 0xFFu;
      if (rank == 0u) cur[d] = first + (info[r] >> 8);  // ... before its leader advances
      lkey[first + rank] = key[r];
      lval[first + rank] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRsTile / kRsBlock; k++) {
    const int j = k * kRsBlock + threadIdx.x;
    if (j < tile_n) {
      const uint32_t kk = lkey[j];
      const uint32_t g = (uint32_t)(j + delta[(kk >> shift) & (kRsBins - 1)]);
      kout[g] = kk;
      vout[g] = lval[j];
    }
  }
}

}  // namespace

// digit width of a sort of end_bit-bit keys: 11 bits where that saves a pass over 10-bit digits
static int radix_sort_bits(int end_bit) {
  const int p10 = (end_bit + 9) / 10, p11 = (end_bit + 10) / 11;
  return p11 < p10 ? 11 : 10;
}

int radix_sort_passes(int end_bit) {
  const int bits = radix_sort_bits(end_bit);
  int p = (end_bit + bits - 1) / bits;
  return p < 1 ? 1 : p;
}

size_t radix_sort_temp_bytes(size_t n) {
  const size_t tiles = ceil_div<size_t>(n > 0 ? n : 1, (size_t)kRsTile);
  return 2 * n * sizeof(uint32_t) + tiles * kRsMaxBins * sizeof(uint32_t) +
         4 * kRsMaxBins * sizeof(uint32_t) + 256;
}

template <int BITS, bool MASK>
static int radix_sort_run(uint32_t* ktmp, uint32_t* vtmp, uint32_t* hist, uint32_t* total,
                          const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                          size_t n, int passes, hipStream_t s, const RsFirst* first) {
  constexpr int kRsBins = 1 << BITS;
  const size_t tiles = ceil_div<size_t>(n, (size_t)kRsTile);
  const uint32_t* sk = kin;
  const uint32_t* sv = vin;
  for (int p = 0; p < passes; p++) {
    const bool to_out = ((passes - 1 - p) % 2) == 0;  // the last pass lands in the caller's buffers
    uint32_t* dk = to_out ? kout : ktmp;
    uint32_t* dv = to_out ? vout : vtmp;
    const int shift = p * BITS;
    RsSrc src;
    src.k64 = (p == 0 && first) ? first->keys64 : nullptr;
    src.flag = (p == 0 && first) ? first->flag : nullptr;
    src.map_inner = first ? first->map_inner : 0u;
    src.map_outer = first ? first->map_outer : 0u;
    // a filtering first pass posts how many keys it kept; the later passes (and the caller's
    // kernels) work on that many -- their grids are launched for n, tiles past the end exit
    const bool filtered = first != nullptr && first->skip_below != 0u && first->n_kept != nullptr;
    src.skip_below = (p == 0 && filtered) ? first->skip_below : 0u;
    src.n_kept = (p == 0 && filtered) ? first->n_kept : nullptr;
    src.n_live = (p > 0 && filtered) ? first->n_kept : nullptr;
    hipLaunchKernelGGL((rs_hist_kernel<BITS>), dim3((unsigned)tiles), dim3(kRsBlock), 0, s,
                       sk, n, shift, hist, src);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL((rs_colscan_kernel<BITS>), dim3(kRsBins / kRsScanBins),
                       dim3(kRsScanBlock), 0, s, hist, tiles, total, src.n_live,
                       (uint32_t)kRsTile);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL((rs_scatter_kernel<BITS, MASK>), dim3((unsigned)tiles),
                       dim3(kRsBlock), 0, s, sk, sv, dk, dv, n, shift, hist, total, src);
    HCTR_LAUNCH_CHECK();
    sk = dk;
    sv = dv;
  }
  return HCTR_OK;
}

int radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout,
                         const uint32_t* vin, uint32_t* vout, size_t n, int end_bit,
                         hipStream_t s, const RsFirst* first) {
  if (n == 0) return HCTR_OK;
  if (temp == nullptr || temp_bytes < radix_sort_temp_bytes(n)) {
    set_error("radix_sort_pairs_u32: workspace too small");
    return HCTR_ERR_INVALID_ARG;
  }
  if (n > 0xFFFFFFF0ull || end_bit < 1 || end_bit > 32) {
    set_error("radix_sort_pairs_u32: n / end_bit out of range");
    return HCTR_ERR_INVALID_ARG;
  }
  const int passes = radix_sort_passes(end_bit);
  uint32_t* ktmp = (uint32_t*)temp;
  uint32_t* vtmp = ktmp + n;
  uint32_t* hist = vtmp + n;
  uint32_t* total = hist + ceil_div<size_t>(n, (size_t)kRsTile) * kRsMaxBins;
  if (radix_sort_bits(end_bit) == 11)
    return radix_sort_run<11, false>(ktmp, vtmp, hist, total, kin, kout, vin, vout, n, passes, s,
                                     first);
  return radix_sort_run<10, true>(ktmp, vtmp, hist, total, kin, kout, vin, vout, n, passes, s,
                                  first);
}

}  // namespace hctr

extern "C" {

size_t hctr_radix_sort_temp_bytes(size_t n) { return hctr::radix_sort_temp_bytes(n); }

int hctr_radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in,
                              uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                              size_t n, int end_bit, hctr_stream_t stream) {
  using namespace hctr;
  HCTR_REQUIRE(n == 0 || (keys_in && keys_out && vals_in && vals_out), "null pointer");
  return radix_sort_pairs_u32(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit,
                              as_stream(stream));
}
}
