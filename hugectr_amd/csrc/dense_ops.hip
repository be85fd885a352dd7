// dense_ops.hip -- the dense ops on the hot path: DLRM dot-interaction and DCN cross layers.
//
// InteractionLayer<T>::fprop/bprop: R/HugeCTR/src/layers/interaction_layer.cu:1046-1237
//   (generic path = concat kernel + cublasGemmStridedBatched X.X^T + gather kernel, three
//   round trips through HBM; fused WMMA path only for fp16).  Here one wavefront owns one sample:
//   the 27x128 tile is staged once in LDS, X.X^T runs on the fp32 MFMA (v_mfma_f32_32x32x2_f32,
//   exact fp32 fma chain), the strict lower triangle is gathered in LDS and the 480-float output
//   row leaves as 16-byte stores.  No `concat` / `mat` intermediates exist.
// MultiCrossLayer<T> (DCN v1): R/HugeCTR/src/layers/multi_cross_layer.cu:582-601 (fprop functor),
//   :671-812 (bprop) -- 4 element-wise kernels + a gemv per layer in the reference; here all layers
//   run in one launch with x0/x_l held in registers (one wavefront per row).
#include <hip/hip_bf16.h>
#include <cstdlib>
#include <cstring>
#include <hip/hip_fp16.h>

#include "block_prims.h"
#include "common.h"

namespace hctr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;

// ================================================================================================
// Interaction forward, fp32, MFMA path: n_ins <= 32, W % 8 == 0, W <= 256
// LDS per wave: X tile [32][W+4] floats (rows >= n_ins stay zero) + out row staging
// ================================================================================================
template <int W>
struct InterCfg {
  static constexpr int LD = W + 4;             // row stride (floats): +16 B breaks b128 conflicts
  static constexpr int XT = 32 * LD;           // X tile floats
};

__device__ __forceinline__ int tri_index(int n, int m) { return n * (n - 1) / 2 + m; }  // n > m

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 -> (hi, lo) bf16 pair: x ~= hi + lo with |x - hi - lo| <= 2^-17 |x|.  Three bf16 MFMAs
// (hi*hi + hi*lo + lo*hi) then reproduce the fp32 product to ~2^-16 relative, at 3/16 of the
// fp32-MFMA cycle cost -- which is what lets the kernel run at the HBM roofline instead of the
// fp32 matrix-pipe limit (MI355X_MICROARCH: f32 MFMA = 1/16 of the bf16 rate).
__device__ __forceinline__ void split8(const float4& p, const float4& q, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// registers <- one sample's [n_ins][W] tile (row 0 = mlp, rows 1.. = emb), 16 B per lane per load.
// Lanes past the tile re-read element 0 so that `pre` stays in registers (no predicated array
// writes -> no scratch).
template <int W, int NPRE>
__device__ __forceinline__ void load_sample_tile(f32x4 (&pre)[NPRE], const float* __restrict__ mlp,
                                                 const float* __restrict__ emb, size_t b, int n_emb,
                                                 int n_vec, int lane) {
  constexpr int W4 = W / 4;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(mlp + b * W);
  const f32x4* e4 = reinterpret_cast<const f32x4*>(emb + b * (size_t)n_emb * W) - W4;
#pragma unroll
  for (int q = 0; q < NPRE; q++) {
    int i = lane + 64 * q;
    i = i < n_vec ? i : 0;
    const f32x4* src = (i < W4) ? m4 : e4;
    pre[q] = src[i];
  }
}

// One wavefront (= one 64-thread workgroup) per sample, software-pipelined over samples:
//   registers <- global (sample i+1, 16-byte coalesced)   ||   MFMA on the LDS tile of sample i
// LDS per workgroup: X tile [(n_ins+1) rows][W+4] (last row = zeros for the padded MFMA rows) +
// the staged output row.
template <int W>
__global__ void __launch_bounds__(64, 2)
    interaction_fwd_mfma_kernel(size_t batch, int n_emb, const float* __restrict__ mlp,
                                const float* __restrict__ emb, float* __restrict__ out,
                                int out_len) {
  using C = InterCfg<W>;
  HCTR_DYN_LDS16(float, smem);
  const int lane = threadIdx.x;
  const int n_ins = n_emb + 1;
  float* xt = smem;
  float* stage = smem + (n_ins + 1) * C::LD;
  constexpr int W4 = W / 4;
  constexpr int NPRE = (32 * W4 + 63) / 64;  // float4 per lane for up to 32 rows
  const int n_vec = n_ins * W4;
  for (int i = lane; i < C::LD; i += 64) xt[n_ins * C::LD + i] = 0.f;  // the zero row

  const int r = lane & 31, h = lane >> 5;
  const int rr = r < n_ins ? r : n_ins;
  f32x4 pre[NPRE];
  size_t b = blockIdx.x;
  if (b < batch) load_sample_tile<W, NPRE>(pre, mlp, emb, b, n_emb, n_vec, lane);
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W4, c4 = i % W4;
        *reinterpret_cast<f32x4*>(xt + row * C::LD + c4 * 4) = pre[q];
      }
    }
    __syncthreads();
    const size_t nb = b + gridDim.x;
    if (nb < batch) load_sample_tile<W, NPRE>(pre, mlp, emb, nb, n_emb, n_vec, lane);

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // lane (r,h) feeds X[r][h*W/2 + 8t .. +7] at k-step t; A == B fragment (product is X.X^T)
    const float* xr = xt + rr * C::LD + h * (W / 2);
#pragma unroll
    for (int t = 0; t < W / 16; t++) {
      const float4 p = *reinterpret_cast<const float4*>(xr + t * 8);
      const float4 q = *reinterpret_cast<const float4*>(xr + t * 8 + 4);
      bf16x8 hi, lo;
      split8(p, q, hi, lo);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, lo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, hi, acc, 0, 0, 0);
    }
    // C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row > r && row < n_ins) stage[W + tri_index(row, r)] = acc[reg];
    }
    for (int i = lane; i < W; i += 64) stage[i] = xt[i];  // mlp passthrough
    if (lane == 0) stage[out_len - 1] = 0.f;              // zero pad column
    __syncthreads();
    float* o = out + b * (size_t)out_len;
    if ((out_len & 3) == 0) {
      for (int i = lane; i < out_len / 4; i += 64)
        reinterpret_cast<float4*>(o)[i] = reinterpret_cast<const float4*>(stage)[i];
    } else {
      for (int i = lane; i < out_len; i += 64) o[i] = stage[i];
    }
    __syncthreads();
  }
}

// ================================================================================================
// 16-bit (bf16 / fp16) interaction: same one-wavefront-per-sample pipeline, a single MFMA chain
// (inputs are already 16-bit), fp32 accumulate, 16-bit output.  This is the reference's mixed
// precision mode (InteractionLayer<__half>, interaction_layer.cu:47-756) with bf16 added.
// ================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool BF>
struct H16;
template <>
struct H16<true> {
  typedef bf16x8 vec8;
  __device__ __forceinline__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  __device__ __forceinline__ static unsigned short from_f32(float v) {
    __bf16 h = (__bf16)v;
    return *reinterpret_cast<unsigned short*>(&h);
  }
  __device__ __forceinline__ static float to_f32(unsigned short u) {
    return __uint_as_float((unsigned)u << 16);
  }
};
template <>
struct H16<false> {
  typedef f16x8 vec8;
  __device__ __forceinline__ static f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  __device__ __forceinline__ static unsigned short from_f32(float v) {
    _Float16 h = (_Float16)v;
    return *reinterpret_cast<unsigned short*>(&h);
  }
  __device__ __forceinline__ static float to_f32(unsigned short u) {
    _Float16 h = *reinterpret_cast<_Float16*>(&u);
    return (float)h;
  }
};

template <int W>
struct InterCfg16 {
  static constexpr int LD = W + 8;   // row stride in 16-bit elements (+16 B against conflicts)
  static constexpr int XT = 32 * LD; // elements
};

// row_of == nullptr: emb is the dense [batch][n_emb][W] tensor.  row_of != nullptr (unique-row
// exchange): emb is a table of distinct rows [R][W] and embedding row s of sample b is
// emb[row_of[b * n_emb + s]] -- the receiver never materialises the expanded tensor.
template <int W, int NPRE>
__device__ __forceinline__ void load_sample_tile16(u32x4 (&pre)[NPRE],
                                                   const unsigned short* __restrict__ mlp,
                                                   const unsigned short* __restrict__ emb,
                                                   const uint32_t* __restrict__ row_of, size_t b,
                                                   int n_emb, int n_vec, int lane) {
  constexpr int W8 = W / 8;
  const u32x4* m4 = reinterpret_cast<const u32x4*>(mlp + b * W);
  if (row_of == nullptr) {
    const u32x4* e4 = reinterpret_cast<const u32x4*>(emb + b * (size_t)n_emb * W) - W8;
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      int i = lane + 64 * q;
      i = i < n_vec ? i : 0;
      const u32x4* src = (i < W8) ? m4 : e4;
      pre[q] = src[i];
    }
  } else {
    const u32x4* r4 = reinterpret_cast<const u32x4*>(emb);
    const uint32_t* ro = row_of + b * (size_t)n_emb;
    uint32_t idx[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      int i = lane + 64 * q;
      i = i < n_vec ? i : 0;
      const int row = i / W8;
      idx[q] = ro[row > 0 ? row - 1 : 0];
    }
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      int i = lane + 64 * q;
      i = i < n_vec ? i : 0;
      const int row = i / W8, c8 = i % W8;
      pre[q] = (row == 0) ? m4[c8] : r4[(size_t)idx[q] * W8 + c8];
    }
  }
}

template <int W, bool BF>
__global__ void __launch_bounds__(64, 2)
    interaction_fwd16_kernel(size_t batch, int n_emb, const unsigned short* __restrict__ mlp,
                             const unsigned short* __restrict__ emb,
                             const uint32_t* __restrict__ row_of,
                             unsigned short* __restrict__ out, int out_len) {
  using C = InterCfg16<W>;
  using H = H16<BF>;
  HCTR_DYN_LDS16(unsigned short, smem16);
  const int lane = threadIdx.x;
  const int n_ins = n_emb + 1;
  unsigned short* xt = smem16;
  unsigned short* stage = smem16 + (n_ins + 1) * C::LD;
  constexpr int W8 = W / 8;
  constexpr int NPRE = (32 * W8 + 63) / 64;
  const int n_vec = n_ins * W8;
  for (int i = lane; i < C::LD; i += 64) xt[n_ins * C::LD + i] = 0;  // zero row

  const int r = lane & 31, h = lane >> 5;
  const int rr = r < n_ins ? r : n_ins;
  u32x4 pre[NPRE];
  size_t b = blockIdx.x;
  if (b < batch) load_sample_tile16<W, NPRE>(pre, mlp, emb, row_of, b, n_emb, n_vec, lane);
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W8, c8 = i % W8;
        *reinterpret_cast<u32x4*>(xt + row * C::LD + c8 * 8) = pre[q];
      }
    }
    __syncthreads();
    const size_t nb = b + gridDim.x;
    if (nb < batch) load_sample_tile16<W, NPRE>(pre, mlp, emb, row_of, nb, n_emb, n_vec, lane);

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned short* xr = xt + rr * C::LD + h * (W / 2);
#pragma unroll
    for (int t = 0; t < W / 16; t++) {
      const typename H::vec8 f = *reinterpret_cast<const typename H::vec8*>(xr + t * 8);
      acc = H::mfma(f, f, acc);
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row > r && row < n_ins) stage[W + tri_index(row, r)] = H::from_f32(acc[reg]);
    }
    for (int i = lane; i < W; i += 64) stage[i] = xt[i];
    if (lane == 0) stage[out_len - 1] = 0;
    __syncthreads();
    unsigned short* o = out + b * (size_t)out_len;
    if ((out_len & 7) == 0) {
      for (int i = lane; i < out_len / 8; i += 64)
        reinterpret_cast<u32x4*>(o)[i] = reinterpret_cast<const u32x4*>(stage)[i];
    } else {
      for (int i = lane; i < out_len; i += 64) o[i] = stage[i];
    }
    __syncthreads();
  }
}

// ---- gather fused into the interaction (one GPU, one key per bucket, sum combiner) ---------------
// The pooled vector of a one-hot bucket IS its table row (rounded to the 16-bit type), and the
// interaction stages each sample's rows in LDS anyway: this kernel reads the fp32 table rows
// through value_index straight into the tile, writes the pooled [batch][n_emb][W] vectors once (the
// backward and the top-gradient layout need them) and runs the same MFMA chain and output stage
// as interaction_fwd16_kernel -- bit-identical to pool_vec4_kernel + interaction_fwd16_kernel,
// without the pass that re-reads the pooled vectors (B * n_emb * W * 2 bytes).
// A missing row (kInvalidIndex: evaluation miss / full table) pools as zeros.
template <int W, int NPRE>
__device__ __forceinline__ void load_gather_idx(uint64_t (&idx)[NPRE],
                                                const uint64_t* __restrict__ value_index,
                                                size_t b, int n_emb, int n_vec, int lane) {
  constexpr int W8 = W / 8;
  const uint64_t* vi = value_index + b * (size_t)n_emb;
#pragma unroll
  for (int q = 0; q < NPRE; q++) {
    int i = lane + 64 * q;
    i = i < n_vec ? i : 0;
    const int row = i / W8;
    idx[q] = vi[row > 0 ? row - 1 : 0];
  }
}

template <int W, int NPRE>
__device__ __forceinline__ void load_gather_rows(f32x4 (&lo)[NPRE], f32x4 (&hi)[NPRE],
                                                 const uint64_t (&idx)[NPRE],
                                                 const unsigned short* __restrict__ mlp,
                                                 const float* __restrict__ table, size_t b,
                                                 int n_vec, int lane) {
  constexpr int W8 = W / 8;
  const f32x4* m4 = reinterpret_cast<const f32x4*>(mlp + b * W);  // (16 bytes = 8 halves)
#pragma unroll
  for (int q = 0; q < NPRE; q++) {
    int i = lane + 64 * q;
    i = i < n_vec ? i : 0;
    const int row = i / W8, c8 = i % W8;
    const uint64_t r = idx[q] != kInvalidIndex ? idx[q] : 0ull;  // always a legal read
    const f32x4* t4 = reinterpret_cast<const f32x4*>(table + r * (uint64_t)W + c8 * 8);
    lo[q] = (row == 0) ? m4[c8] : t4[0];
    hi[q] = (row == 0) ? m4[c8] : t4[1];
  }
}

template <int W, bool BF>
__global__ void __launch_bounds__(64, 2)
    interaction_fwd16_gather_kernel(size_t batch, int n_emb,
                                    const unsigned short* __restrict__ mlp,
                                    const float* __restrict__ table,
                                    const uint64_t* __restrict__ value_index,
                                    unsigned short* __restrict__ pooled,
                                    unsigned short* __restrict__ out, int out_len) {
  using C = InterCfg16<W>;
  using H = H16<BF>;
  HCTR_DYN_LDS16(unsigned short, smem16);
  const int lane = threadIdx.x;
  const int n_ins = n_emb + 1;
  unsigned short* xt = smem16;
  unsigned short* stage = smem16 + (n_ins + 1) * C::LD;
  constexpr int W8 = W / 8;
  constexpr int NPRE = (32 * W8 + 63) / 64;
  const int n_vec = n_ins * W8;
  for (int i = lane; i < C::LD; i += 64) xt[n_ins * C::LD + i] = 0;  // zero row

  const int r = lane & 31, h = lane >> 5;
  const int rr = r < n_ins ? r : n_ins;
  f32x4 lo[NPRE], hi[NPRE];
  uint64_t idx[NPRE], idx_nxt[NPRE];
  size_t b = blockIdx.x;
  const size_t last = batch - 1;
  // row indices run one sample ahead of the rows, the rows one sample ahead of the MFMA chain
  load_gather_idx<W, NPRE>(idx, value_index, b < batch ? b : last, n_emb, n_vec, lane);
  load_gather_rows<W, NPRE>(lo, hi, idx, mlp, table, b < batch ? b : last, n_vec, lane);
  {
    const size_t nb = b + gridDim.x;
    load_gather_idx<W, NPRE>(idx_nxt, value_index, nb < batch ? nb : last, n_emb, n_vec, lane);
  }
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W8, c8 = i % W8;
        u32x4 v;
        if (row == 0) {
          v = *reinterpret_cast<const u32x4*>(&lo[q]);
        } else {
          const bool live = idx[q] != kInvalidIndex;
          float f[8] = {lo[q][0], lo[q][1], lo[q][2], lo[q][3], hi[q][0], hi[q][1], hi[q][2], hi[q][3]};
          unsigned short u[8];
#pragma unroll
          for (int k = 0; k < 8; k++) u[k] = H::from_f32(0.f + (live ? f[k] : 0.f));
          v[0] = (uint32_t)u[0] | ((uint32_t)u[1] << 16);
          v[1] = (uint32_t)u[2] | ((uint32_t)u[3] << 16);
          v[2] = (uint32_t)u[4] | ((uint32_t)u[5] << 16);
          v[3] = (uint32_t)u[6] | ((uint32_t)u[7] << 16);
          *reinterpret_cast<u32x4*>(pooled + (b * (size_t)n_emb + (row - 1)) * W + c8 * 8) = v;
        }
        *reinterpret_cast<u32x4*>(xt + row * C::LD + c8 * 8) = v;
      }
    }
    __syncthreads();
    const size_t nb = b + gridDim.x, nb2 = nb + gridDim.x;
#pragma unroll
    for (int q = 0; q < NPRE; q++) idx[q] = idx_nxt[q];
    if (nb < batch) load_gather_rows<W, NPRE>(lo, hi, idx, mlp, table, nb, n_vec, lane);
    load_gather_idx<W, NPRE>(idx_nxt, value_index, nb2 < batch ? nb2 : last, n_emb, n_vec, lane);

    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned short* xr = xt + rr * C::LD + h * (W / 2);
#pragma unroll
    for (int t = 0; t < W / 16; t++) {
      const typename H::vec8 f = *reinterpret_cast<const typename H::vec8*>(xr + t * 8);
      acc = H::mfma(f, f, acc);
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row > r && row < n_ins) stage[W + tri_index(row, r)] = H::from_f32(acc[reg]);
    }
    for (int i = lane; i < W; i += 64) stage[i] = xt[i];
    if (lane == 0) stage[out_len - 1] = 0;
    __syncthreads();
    unsigned short* o = out + b * (size_t)out_len;
    if ((out_len & 7) == 0) {
      for (int i = lane; i < out_len / 8; i += 64)
        reinterpret_cast<u32x4*>(o)[i] = reinterpret_cast<const u32x4*>(stage)[i];
    } else {
      for (int i = lane; i < out_len; i += 64) o[i] = stage[i];
    }
    __syncthreads();
  }
}

template <int W, bool BF>
__global__ void __launch_bounds__(64, 2)
    interaction_bwd16_kernel(size_t batch, int n_emb, const unsigned short* __restrict__ mlp,
                             const unsigned short* __restrict__ emb,
                             const uint32_t* __restrict__ row_of,
                             const unsigned short* __restrict__ top_grad,
                             unsigned short* __restrict__ mlp_grad,
                             unsigned short* __restrict__ emb_grad, int out_len,
                             const uint32_t* __restrict__ grad_map) {
  // grad_map != nullptr: the gradient of embedding s of sample b goes to row
  // grad_map[b * n_emb + s] of emb_grad (the all-to-all send layout: no reorder pass behind it)
  using C = InterCfg16<W>;
  using H = H16<BF>;
  constexpr int GS = 40;  // G row stride (16-bit elements): 80 B -> 16 distinct 16-B slots
  HCTR_DYN_LDS16(unsigned short, smem16);
  const int lane = threadIdx.x;
  const int n_ins = n_emb + 1;
  unsigned short* xt = smem16;            // [32][LD] X, reused for dX
  unsigned short* gm = smem16 + C::XT;    // [32][GS]
  unsigned short* pair_nm = gm + 32 * GS; // [n_pairs]
  constexpr int W8 = W / 8;
  constexpr int NT = W / 32;
  constexpr int NPRE = (32 * W8 + 63) / 64;
  constexpr int NG = 8;
  const int n_vec = n_ins * W8;
  const int n_pairs = n_ins * (n_ins - 1) / 2;
  for (int i = lane; i < C::XT + 32 * GS; i += 64) smem16[i] = 0;
  for (int p = lane; p < n_pairs; p += 64) {
    int n = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (n * (n - 1) / 2 > p) n--;
    while ((n + 1) * n / 2 <= p) n++;
    pair_nm[p] = (unsigned short)((n << 8) | (p - n * (n - 1) / 2));
  }
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  u32x4 pre[NPRE];
  unsigned short gpre[NG];
  size_t b = blockIdx.x;
#define HCTR_BWD16_PREFETCH(bb)                                                  \
  {                                                                              \
    load_sample_tile16<W, NPRE>(pre, mlp, emb, row_of, (bb), n_emb, n_vec, lane); \
    const unsigned short* g__ = top_grad + (bb) * (size_t)out_len + W;           \
    _Pragma("unroll") for (int q = 0; q < NG; q++) {                             \
      int p__ = lane + 64 * q;                                                   \
      p__ = p__ < n_pairs ? p__ : 0;                                             \
      gpre[q] = g__[p__];                                                        \
    }                                                                            \
  }
  if (b < batch) HCTR_BWD16_PREFETCH(b)
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W8, c8 = i % W8;
        *reinterpret_cast<u32x4*>(xt + row * C::LD + c8 * 8) = pre[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int p = lane + 64 * q;
      if (p < n_pairs) {
        const int nm = pair_nm[p];
        const int n = nm >> 8, m = nm & 0xFF;
        gm[n * GS + m] = gpre[q];
        gm[m * GS + n] = gpre[q];
      }
    }
    __syncthreads();
    const size_t nb = b + gridDim.x;
    if (nb < batch) HCTR_BWD16_PREFETCH(nb)

    constexpr int HP = NT >= 2 ? 2 : 1;
#pragma unroll
    for (int pn = 0; pn < NT / HP; pn++) {
      f32x16 acc[HP];
#pragma unroll
      for (int t = 0; t < HP; t++)
        acc[t] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const int k0 = 16 * s2 + 8 * h;
        const typename H::vec8 af = *reinterpret_cast<const typename H::vec8*>(gm + r * GS + k0);
#pragma unroll
        for (int t = 0; t < HP; t++) {
          const int col = (pn * HP + t) * 32 + r;
          unsigned int w4[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const unsigned lo = xt[(k0 + 2 * e) * C::LD + col];
            const unsigned hi = xt[(k0 + 2 * e + 1) * C::LD + col];
            w4[e] = lo | (hi << 16);
          }
          const u32x4 packed = {w4[0], w4[1], w4[2], w4[3]};
          const typename H::vec8 bfv = *reinterpret_cast<const typename H::vec8*>(&packed);
          acc[t] = H::mfma(af, bfv, acc[t]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < HP; t++) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (row < n_ins) xt[row * C::LD + (pn * HP + t) * 32 + r] = H::from_f32(acc[t][reg]);
        }
      }
    }
    __syncthreads();
    const unsigned short* gtop = top_grad + b * (size_t)out_len;
    u32x4* mg4 = reinterpret_cast<u32x4*>(mlp_grad + b * W);
    u32x4* eg4 = reinterpret_cast<u32x4*>(emb_grad + b * (size_t)n_emb * W);
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W8, c8 = i % W8;
        u32x4 v = *reinterpret_cast<const u32x4*>(xt + row * C::LD + c8 * 8);
        if (row == 0) {
          const u32x4 gt = reinterpret_cast<const u32x4*>(gtop)[c8];
          u32x4 o4;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float a0 = H::to_f32((unsigned short)(v[e] & 0xFFFFu)) +
                             H::to_f32((unsigned short)(gt[e] & 0xFFFFu));
            const float a1 = H::to_f32((unsigned short)(v[e] >> 16)) +
                             H::to_f32((unsigned short)(gt[e] >> 16));
            o4[e] = (unsigned)H::from_f32(a0) | ((unsigned)H::from_f32(a1) << 16);
          }
          mg4[c8] = o4;
        } else if (grad_map == nullptr) {
          eg4[i - W8] = v;
        } else {
          const uint32_t dst = grad_map[b * (size_t)n_emb + (row - 1)];
          reinterpret_cast<u32x4*>(emb_grad + (size_t)dst * W)[c8] = v;
        }
      }
    }
    __syncthreads();
  }
#undef HCTR_BWD16_PREFETCH
}

// any shape / dtype: one wavefront per sample, VALU dot products (fp32 accumulate)
template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <>
__device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) {
  return __bfloat162float(v);
}
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half_rn(v); }
template <>
__device__ __forceinline__ __hip_bfloat16 from_f32<__hip_bfloat16>(float v) {
  return __float2bfloat16(v);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
    interaction_fwd_generic_kernel(size_t batch, int n_emb, int W, const T* __restrict__ mlp,
                                   const T* __restrict__ emb, T* __restrict__ out, int out_len) {
  HCTR_DYN_LDS16(float, smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ins = n_emb + 1;
  float* xt = smem + wave * (n_ins * (W + 1));
  const size_t waves_total = (size_t)gridDim.x * kWavesPerBlock;
  const size_t iters = (batch + waves_total - 1) / waves_total;
  const int n_pairs = n_ins * (n_ins - 1) / 2;
  for (size_t it = 0; it < iters; it++) {
    const size_t b = it * waves_total + (size_t)blockIdx.x * kWavesPerBlock + wave;
    const bool valid = b < batch;
    if (valid) {
      for (int i = lane; i < n_ins * W; i += 64) {
        const int row = i / W, c = i % W;
        xt[row * (W + 1) + c] =
            to_f32<T>(row == 0 ? mlp[b * W + c] : emb[(b * n_emb + (row - 1)) * (size_t)W + c]);
      }
    }
    __syncthreads();
    if (valid) {
      T* o = out + b * (size_t)out_len;
      for (int i = lane; i < W; i += 64) o[i] = from_f32<T>(xt[i]);
      for (int p = lane; p < n_pairs; p += 64) {
        // invert p = n(n-1)/2 + m
        int n = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (n * (n - 1) / 2 > p) n--;
        while ((n + 1) * n / 2 <= p) n++;
        const int m = p - n * (n - 1) / 2;
        float a = 0.f;
        for (int k = 0; k < W; k++) a += xt[m * (W + 1) + k] * xt[n * (W + 1) + k];
        o[W + p] = from_f32<T>(a);
      }
      if (lane == 0) o[out_len - 1] = from_f32<T>(0.f);
    }
    __syncthreads();
  }
}

// ================================================================================================
// Interaction backward, fp32 MFMA path.  G = dM + dM^T (symmetric, zero diagonal) in LDS,
// dX = G . X : M = 32 (n_ins padded), N = W, K = 32.
//   mlp_grad[b] = top_grad[b][0:W] + dX[0];  emb_grad[b][i-1] = dX[i]
// ================================================================================================
template <int W>
__global__ void __launch_bounds__(64, 2)
    interaction_bwd_mfma_kernel(size_t batch, int n_emb, const float* __restrict__ mlp,
                                const float* __restrict__ emb, const float* __restrict__ top_grad,
                                float* __restrict__ mlp_grad, float* __restrict__ emb_grad,
                                int out_len) {
  using C = InterCfg<W>;
  constexpr int GS = 36;  // G row stride (floats): 16 lanes of a ds_read_b128 group -> 16 slots
  HCTR_DYN_LDS16(float, smem);
  const int lane = threadIdx.x;
  const int n_ins = n_emb + 1;
  float* xt = smem;                 // [32][LD]  X, later reused for dX
  float* gm = smem + C::XT;         // [32][GS]  G = dM + dM^T
  unsigned short* pair_nm = reinterpret_cast<unsigned short*>(gm + 32 * GS);  // [n_pairs]
  constexpr int W4 = W / 4;
  constexpr int NT = W / 32;
  constexpr int NPRE = (32 * W4 + 63) / 64;
  constexpr int NG = 8;  // ceil(496 / 64) gradient words per lane
  const int n_vec = n_ins * W4;
  const int n_pairs = n_ins * (n_ins - 1) / 2;
  for (int i = lane; i < C::XT + 32 * GS; i += 64) smem[i] = 0.f;
  for (int p = lane; p < n_pairs; p += 64) {
    int n = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (n * (n - 1) / 2 > p) n--;
    while ((n + 1) * n / 2 <= p) n++;
    pair_nm[p] = (unsigned short)((n << 8) | (p - n * (n - 1) / 2));
  }
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  f32x4 pre[NPRE];
  float gpre[NG];
  size_t b = blockIdx.x;
#define HCTR_BWD_PREFETCH(bb)                                                   \
  {                                                                             \
    load_sample_tile<W, NPRE>(pre, mlp, emb, (bb), n_emb, n_vec, lane);         \
    const float* g__ = top_grad + (bb) * (size_t)out_len + W;                   \
    _Pragma("unroll") for (int q = 0; q < NG; q++) {                            \
      int p__ = lane + 64 * q;                                                  \
      p__ = p__ < n_pairs ? p__ : 0;                                            \
      gpre[q] = g__[p__];                                                       \
    }                                                                           \
  }
  if (b < batch) HCTR_BWD_PREFETCH(b)
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W4, c4 = i % W4;
        *reinterpret_cast<f32x4*>(xt + row * C::LD + c4 * 4) = pre[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NG; q++) {
      const int p = lane + 64 * q;
      if (p < n_pairs) {
        const int nm = pair_nm[p];
        const int n = nm >> 8, m = nm & 0xFF;
        gm[n * GS + m] = gpre[q];
        gm[m * GS + n] = gpre[q];
      }
    }
    __syncthreads();
    const size_t nb = b + gridDim.x;
    if (nb < batch) HCTR_BWD_PREFETCH(nb)

    // dX = G . X : A = G [32 x 32], B = X [32 x W]; K = 32 -> two k-steps of 16.  The output is
    // produced in column panels of HP*32 columns (32 accumulator registers instead of 64); panel
    // p only reads columns of X that earlier panels did not overwrite, so dX replaces X in place.
    constexpr int HP = NT >= 2 ? 2 : 1;  // N tiles per panel
#pragma unroll
    for (int pn = 0; pn < NT / HP; pn++) {
      f32x16 acc[HP];
#pragma unroll
      for (int t = 0; t < HP; t++)
        acc[t] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const int k0 = 16 * s2 + 8 * h;
        const float4 ga = *reinterpret_cast<const float4*>(gm + r * GS + k0);
        const float4 gb = *reinterpret_cast<const float4*>(gm + r * GS + k0 + 4);
        bf16x8 ah, al;
        split8(ga, gb, ah, al);
#pragma unroll
        for (int t = 0; t < HP; t++) {
          const int col = (pn * HP + t) * 32 + r;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; e++) xv[e] = xt[(k0 + e) * C::LD + col];
          bf16x8 bh, bl;
          split8(make_float4(xv[0], xv[1], xv[2], xv[3]), make_float4(xv[4], xv[5], xv[6], xv[7]),
                 bh, bl);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
        }
      }
      __syncthreads();  // every lane is done reading this panel's columns of X
#pragma unroll
      for (int t = 0; t < HP; t++) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (row < n_ins) xt[row * C::LD + (pn * HP + t) * 32 + r] = acc[t][reg];
        }
      }
    }
    __syncthreads();
    const float4* gtop4 = reinterpret_cast<const float4*>(top_grad + b * (size_t)out_len);
    float4* mg4 = reinterpret_cast<float4*>(mlp_grad + b * W);
    float4* eg4 = reinterpret_cast<float4*>(emb_grad + b * (size_t)n_emb * W);
#pragma unroll
    for (int q = 0; q < NPRE; q++) {
      const int i = lane + 64 * q;
      if (i < n_vec) {
        const int row = i / W4, c4 = i % W4;
        float4 v = *reinterpret_cast<const float4*>(xt + row * C::LD + c4 * 4);
        if (row == 0) {
          const float4 gt = gtop4[c4];
          v.x += gt.x;
          v.y += gt.y;
          v.z += gt.z;
          v.w += gt.w;
          mg4[c4] = v;
        } else {
          eg4[i - W4] = v;
        }
      }
    }
    __syncthreads();
    // rows >= n_ins of the tile were never written; rows < n_ins are rewritten next iteration
  }
#undef HCTR_BWD_PREFETCH
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
    interaction_bwd_generic_kernel(size_t batch, int n_emb, int W, const T* __restrict__ mlp,
                                   const T* __restrict__ emb, const T* __restrict__ top_grad,
                                   T* __restrict__ mlp_grad, T* __restrict__ emb_grad,
                                   int out_len) {
  HCTR_DYN_LDS16(float, smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ins = n_emb + 1;
  float* xt = smem + wave * (n_ins * (W + 1) + n_ins * n_ins);
  float* gm = xt + n_ins * (W + 1);
  const size_t waves_total = (size_t)gridDim.x * kWavesPerBlock;
  const size_t iters = (batch + waves_total - 1) / waves_total;
  for (size_t it = 0; it < iters; it++) {
    const size_t b = it * waves_total + (size_t)blockIdx.x * kWavesPerBlock + wave;
    const bool valid = b < batch;
    if (valid) {
      for (int i = lane; i < n_ins * W; i += 64) {
        const int row = i / W, c = i % W;
        xt[row * (W + 1) + c] =
            to_f32<T>(row == 0 ? mlp[b * W + c] : emb[(b * n_emb + (row - 1)) * (size_t)W + c]);
      }
      const T* g = top_grad + b * (size_t)out_len + W;
      for (int i = lane; i < n_ins * n_ins; i += 64) {
        const int m = i / n_ins, n = i % n_ins;
        float v = 0.f;
        if (m != n) {
          const int hi = m > n ? m : n, lo = m > n ? n : m;
          v = to_f32<T>(g[hi * (hi - 1) / 2 + lo]);
        }
        gm[i] = v;
      }
    }
    __syncthreads();
    if (valid) {
      const T* gtop = top_grad + b * (size_t)out_len;
      for (int i = lane; i < n_ins * W; i += 64) {
        const int m = i / W, n = i % W;
        float a = 0.f;
        for (int k = 0; k < n_ins; k++) a += gm[m * n_ins + k] * xt[k * (W + 1) + n];
        if (m == 0) mlp_grad[b * W + n] = from_f32<T>(to_f32<T>(gtop[n]) + a);
        else emb_grad[(b * n_emb + (m - 1)) * (size_t)W + n] = from_f32<T>(a);
      }
    }
    __syncthreads();
  }
}

// ================================================================================================
// DCN v1 cross layers: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l   (one wavefront per row)
// ================================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <int NPL>
__global__ void __launch_bounds__(kBlock)
    cross_v1_fwd_kernel(size_t batch, int w, int layers, const float* __restrict__ x0,
                        const float* __restrict__ kernels, const float* __restrict__ biases,
                        float* __restrict__ outputs, float* __restrict__ hiddens) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  for (size_t row = wave; row < batch; row += nwaves) {
    float a0[NPL], xl[NPL];
#pragma unroll
    for (int t = 0; t < NPL; t++) {
      const int i = lane + 64 * t;
      a0[t] = (i < w) ? x0[row * w + i] : 0.f;
      xl[t] = a0[t];
    }
    for (int l = 0; l < layers; l++) {
      const float* k = kernels + (size_t)l * w;
      const float* bi = biases + (size_t)l * w;
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < NPL; t++) {
        const int i = lane + 64 * t;
        if (i < w) part += xl[t] * k[i];
      }
      const float hsum = wave_sum(part);
      if (lane == 0) hiddens[(size_t)l * batch + row] = hsum;
      float* o = outputs + ((size_t)l * batch + row) * w;
#pragma unroll
      for (int t = 0; t < NPL; t++) {
        const int i = lane + 64 * t;
        if (i < w) {
          float v = a0[t] * hsum;
          v = v + xl[t];
          v = v + bi[i];
          xl[t] = v;
          o[i] = v;
        }
      }
    }
  }
}

// backward: per row local math; dW/db column sums go to per-wave partial rows in `partials`
// [num_waves][layers][2][w], reduced in fixed order by cross_v1_reduce_kernel (deterministic).
// LDS = true: one wavefront per workgroup keeps its dW/db partial rows in LDS (2*layers*w floats)
// instead of read-modify-writing them in global memory for every row -- the per-row RMW chain
// through L2 made the backward 12x slower than the forward at the DCN shape.
template <int NPL, bool LDS>
__global__ void __launch_bounds__(kBlock)
    cross_v1_bwd_kernel(size_t batch, int w, int layers, const float* __restrict__ x0,
                        const float* __restrict__ kernels, const float* __restrict__ outputs,
                        const float* __restrict__ hiddens, const float* __restrict__ out_grad,
                        float* __restrict__ in_grad, float* __restrict__ partials) {
  HCTR_DYN_LDS(float, cross_lds);
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  float* gmy = partials + wave * (size_t)layers * 2 * w;
  float* my = LDS ? cross_lds : gmy;
  for (int i = lane; i < layers * 2 * w; i += 64) my[i] = 0.f;
  // (entry i is cleared / copied out by lane i % 64 but accumulated by the lane that owns its
  //  COLUMN: the partial rows cross lanes at both ends of the row loop)
  __builtin_amdgcn_wave_barrier();
  for (size_t row = wave; row < batch; row += nwaves) {
    float a0[NPL], dy[NPL], dx0[NPL];
#pragma unroll
    for (int t = 0; t < NPL; t++) {
      const int i = lane + 64 * t;
      a0[t] = (i < w) ? x0[row * w + i] : 0.f;
      dy[t] = (i < w) ? out_grad[row * w + i] : 0.f;
      dx0[t] = 0.f;
    }
    for (int l = layers - 1; l >= 0; l--) {
      const float hsum = hiddens[(size_t)l * batch + row];
      const float* k = kernels + (size_t)l * w;
      const float* xprev = (l == 0) ? x0 + row * w : outputs + ((size_t)(l - 1) * batch + row) * w;
      float part = 0.f;
#pragma unroll
      for (int t = 0; t < NPL; t++) {
        dx0[t] += dy[t] * hsum;  // row_scaling + matrix_add
        part += dy[t] * a0[t];   // matrix_pair_mul
      }
      const float tv = wave_sum(part);
      float* dwp = my + (size_t)l * 2 * w;
      float* dbp = dwp + w;
#pragma unroll
      for (int t = 0; t < NPL; t++) {
        const int i = lane + 64 * t;
        if (i < w) {
          dwp[i] += xprev[i] * tv;  // row_scaling_sum
          dbp[i] += dy[t];          // rows_sum
          dy[t] = dy[t] + tv * k[i];  // out_product + matrix_add
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NPL; t++) {
      const int i = lane + 64 * t;
      if (i < w) in_grad[row * w + i] = dx0[t] + dy[t];
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (LDS)
    for (int i = lane; i < layers * 2 * w; i += 64) gmy[i] = my[i];
}

// column sums of the per-wave partial rows: a workgroup owns 32 columns x 8 row groups; the 8
// group sums are added in fixed order (deterministic)
__global__ void __launch_bounds__(kBlock)
    cross_v1_reduce_kernel(size_t nwaves, int w, int layers, const float* __restrict__ partials,
                           float* __restrict__ kernel_grads, float* __restrict__ bias_grads) {
  __shared__ float red[kBlock];
  const size_t total = (size_t)layers * 2 * w;
  const size_t i = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
  const int tg = threadIdx.x >> 5;
  float s = 0.f;
  if (i < total) {
#pragma unroll 8
    for (size_t q = tg; q < nwaves; q += 8) s += partials[q * total + i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tg == 0 && i < total) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) sum += red[k * 32 + threadIdx.x];
    const int l = (int)(i / (2 * w)), rem = (int)(i % (2 * w));
    if (rem < w) kernel_grads[(size_t)l * w + rem] = sum;
    else bias_grads[(size_t)l * w + (rem - w)] = sum;
  }
}

// ================================================================================================
// Fused ReLU backward + bias gradient for the MLP layers around the path:
//   dz[b][n] = dy[b][n] * (y[b][n] > 0),   db[n] = sum_b dz[b][n]
// PyTorch issues threshold_backward (read dy,y / write dz) and a column-sum reduction (read dz)
// as two launches; fused, dz is consumed from registers.  Two-stage, fixed-order column sums
// (deterministic): stage 1 writes partial[row_tile][n], stage 2 adds the tiles in order.
// ================================================================================================
constexpr int kRbRows = 128;  // rows per workgroup

// rows per workgroup of cross_v2_bwd_step_kernel: 128 at most, fewer (a power of two >= 16) while
// the grid would stay below ~2 k workgroups -- 64 tiles of 128 rows at batch 8192 left three
// quarters of the CUs idle
inline int cross_step_rows_per_tile(size_t batch, int width) {
  const int n8 = width / 8;
  const int cw = n8 < kBlock ? n8 : kBlock;
  const size_t col_blocks = (size_t)((n8 + cw - 1) / cw);
  int rpt = 128;
  while (rpt > 16 && ((batch + rpt - 1) / rpt) * col_blocks < 2048) rpt >>= 1;
  return rpt;
}

// thread t of a workgroup owns 16-byte column vector (t % cw) and walks rows (t / cw), +rg, ...;
// cw = min(n/8, 256), rg = 256 / cw row groups; the rg partial sums meet in LDS in fixed order.
template <bool BF>
__global__ void __launch_bounds__(kBlock)
    relu_bwd_bias_kernel(size_t rows, int n, int cw, int rg, const unsigned short* __restrict__ dy,
                         const unsigned short* __restrict__ y, unsigned short* __restrict__ dz,
                         float* __restrict__ partial) {
  using H = H16<BF>;
  __shared__ float red[kBlock * 8];
  const int n8 = n / 8;
  const int c8 = threadIdx.x % cw, g = threadIdx.x / cw;
  const bool live = g < rg;
  const size_t r0 = (size_t)blockIdx.x * kRbRows;
  const size_t r1 = r0 + kRbRows < rows ? r0 + kRbRows : rows;
  for (int cc = c8; cc - c8 < n8; cc += cw) {  // uniform trip count: every thread reaches the barriers
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live && cc < n8) {
#pragma unroll 4
      for (size_t r = r0 + g; r < r1; r += rg) {
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + r * n + cc * 8);
        const u32x4 av = *reinterpret_cast<const u32x4*>(y + r * n + cc * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const unsigned short g0 = (unsigned short)(gv[e] & 0xFFFFu);
          const unsigned short g1 = (unsigned short)(gv[e] >> 16);
          const float y0 = H::to_f32((unsigned short)(av[e] & 0xFFFFu));
          const float y1 = H::to_f32((unsigned short)(av[e] >> 16));
          const unsigned short z0 = y0 > 0.f ? g0 : (unsigned short)0;
          const unsigned short z1 = y1 > 0.f ? g1 : (unsigned short)0;
          acc[2 * e] += H::to_f32(z0);
          acc[2 * e + 1] += H::to_f32(z1);
          o[e] = (unsigned)z0 | ((unsigned)z1 << 16);
        }
        *reinterpret_cast<u32x4*>(dz + r * n + cc * 8) = o;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[e * kBlock + threadIdx.x] = acc[e];
    __syncthreads();
    if (g == 0 && cc < n8) {
      float* p = partial + (size_t)blockIdx.x * n + cc * 8;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float sum = 0.f;
        for (int k = 0; k < rg; k++) sum += red[e * kBlock + k * cw + c8];
        p[e] = sum;
      }
    }
    __syncthreads();
  }
}

// ================================================================================================
// DCN-v2 cross layer, the elementwise step of one layer's backward in the activations' 16-bit type
// (MultiCrossBackwardFunctorv2: fused_mul_fma3, R/HugeCTR/src/layers/multi_cross_layer.cu:391-424,
// 127-165 -- S0 = dY .* X0, dX += dY .* H, one rounding per element as its paired-half kernel
// rounds -- and the bias gradient db = column sums of S0, which the reference takes in the
// epilogue of the dV GEMM, :770-776): dY is read once for both products, S0 is summed from
// registers.  Thread layout and the two-stage fixed-order column sums are relu_bwd_bias_kernel's.
// FIRST (the last layer, visited first): dX = dY .* H, the accumulator is not read (nor cleared
// beforehand).
// ================================================================================================
template <bool BF, bool FIRST>
__global__ void __launch_bounds__(kBlock)
    cross_v2_bwd_step_kernel(size_t rows, int n, int cw, int rg, int rpt,
                             const unsigned short* __restrict__ dy,
                             const unsigned short* __restrict__ x0,
                             const unsigned short* __restrict__ hm, unsigned short* __restrict__ acc_io,
                             unsigned short* __restrict__ s0, float* __restrict__ partial) {
  using H = H16<BF>;
  __shared__ float red[kBlock * 8];
  const int n8 = n / 8;
  const int c8 = threadIdx.x % cw, g = threadIdx.x / cw;
  const bool live = g < rg;
  // a workgroup = rpt rows x cw 16-byte column vectors (blockIdx.y picks the column block): the
  // grid is (row tiles, column blocks), sized by the host for >= ~1 k workgroups at any batch
  const size_t r0 = (size_t)blockIdx.x * (size_t)rpt;
  const size_t r1 = r0 + (size_t)rpt < rows ? r0 + (size_t)rpt : rows;
  const int cc = (int)blockIdx.y * cw + c8;
  {
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live && cc < n8) {
#pragma unroll 4
      for (size_t r = r0 + g; r < r1; r += rg) {
        const size_t at = r * n + cc * 8;
        const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + at);
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x0 + at);
        const u32x4 hv = *reinterpret_cast<const u32x4*>(hm + at);
        u32x4 av = {0u, 0u, 0u, 0u};
        if (!FIRST) av = *reinterpret_cast<const u32x4*>(acc_io + at);
        u32x4 so, ao;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float g0 = H::to_f32((unsigned short)(gv[e] & 0xFFFFu));
          const float g1 = H::to_f32((unsigned short)(gv[e] >> 16));
          // (the product of two 16-bit values is exact in fp32: one rounding, to the 16-bit type)
          const unsigned short p0 = H::from_f32(g0 * H::to_f32((unsigned short)(xv[e] & 0xFFFFu)));
          const unsigned short p1 = H::from_f32(g1 * H::to_f32((unsigned short)(xv[e] >> 16)));
          sum[2 * e] += H::to_f32(p0);
          sum[2 * e + 1] += H::to_f32(p1);
          so[e] = (unsigned)p0 | ((unsigned)p1 << 16);
          float a0 = g0 * H::to_f32((unsigned short)(hv[e] & 0xFFFFu));
          float a1 = g1 * H::to_f32((unsigned short)(hv[e] >> 16));
          if (!FIRST) {
            a0 += H::to_f32((unsigned short)(av[e] & 0xFFFFu));
            a1 += H::to_f32((unsigned short)(av[e] >> 16));
          }
          ao[e] = (unsigned)H::from_f32(a0) | ((unsigned)H::from_f32(a1) << 16);
        }
        *reinterpret_cast<u32x4*>(s0 + at) = so;
        *reinterpret_cast<u32x4*>(acc_io + at) = ao;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[e * kBlock + threadIdx.x] = sum[e];
    __syncthreads();
    if (g == 0 && cc < n8) {
      float* p = partial + (size_t)blockIdx.x * n + cc * 8;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float t = 0.f;
        for (int k = 0; k < rg; k++) t += red[e * kBlock + k * cw + c8];
        p[e] = t;
      }
    }
    __syncthreads();
  }
}

// db[c..c+3] = sum over tiles of partial[tile][c..c+3]: a workgroup owns 8 float4 column groups x 32
// tile groups; the 32 group sums are added in fixed order through LDS.
__global__ void __launch_bounds__(kBlock)
    colsum_partials_kernel(size_t tiles, int n, const float* __restrict__ partial,
                           float* __restrict__ db) {
  __shared__ f32x4 red[kBlock];
  const int c4 = blockIdx.x * 8 + (threadIdx.x & 7), tg = threadIdx.x >> 3;
  const bool live = c4 * 4 < n;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (live) {
#pragma unroll 8
    for (size_t t = tg; t < tiles; t += 32)
      s += *reinterpret_cast<const f32x4*>(partial + t * n + c4 * 4);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (tg == 0 && live) {
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 32; k++) sum += red[k * 8 + threadIdx.x];
    *reinterpret_cast<f32x4*>(db + c4 * 4) = sum;
  }
}

// out[i] = sum_g in[g][i] (fixed order), 16-bit partial products of a split-K GEMM -> fp32
template <bool BF>
__global__ void __launch_bounds__(kBlock)
    sum_groups_kernel(int groups, size_t n8, const unsigned short* __restrict__ in,
                      float* __restrict__ out) {
  using H = H16<BF>;
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n8) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int g = 0; g < groups; g++) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(in + ((size_t)g * n8 + i) * 8);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      acc[2 * e] += H::to_f32((unsigned short)(v[e] & 0xFFFFu));
      acc[2 * e + 1] += H::to_f32((unsigned short)(v[e] >> 16));
    }
  }
  f32x4* o = reinterpret_cast<f32x4*>(out + i * 8);
  o[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
  o[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// dense SGD step fused with the refresh of the 16-bit compute copy: w -= lr * g; w16 = (T16)w
template <bool BF>
__global__ void __launch_bounds__(kBlock)
    sgd_shadow_kernel(size_t n4, float lr, float grad_scale, float* __restrict__ w,
                      const float* __restrict__ g, unsigned short* __restrict__ w16) {
  using H = H16<BF>;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
       i += (size_t)gridDim.x * kBlock) {
    f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
    wv -= lr * grad_scale * gv;
    reinterpret_cast<f32x4*>(w)[i] = wv;
    const unsigned lo = (unsigned)H::from_f32(wv[0]) | ((unsigned)H::from_f32(wv[1]) << 16);
    const unsigned hi = (unsigned)H::from_f32(wv[2]) | ((unsigned)H::from_f32(wv[3]) << 16);
    reinterpret_cast<uint2*>(w16)[i] = make_uint2(lo, hi);
  }
}

// ================================================================================================
// BinaryCrossEntropyLoss (R/HugeCTR/src/loss.cu:231-262): per-sample stable BCE-with-logits,
// the logit gradient scaled by grad_scale (= scaler / batch / total_gpu_count in the reference),
// and the mean loss.  The reference accumulates the block sums with atomicAdd; here the block
// partials are added in fixed order by the last launch (deterministic).
// ================================================================================================
constexpr int kBceBlocks = 256;

template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p, size_t i);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p, size_t i) { return p[i]; }
template <>
__device__ __forceinline__ float ld_as_f32<__half>(const __half* p, size_t i) {
  return __half2float(p[i]);
}
template <>
__device__ __forceinline__ float ld_as_f32<__hip_bfloat16>(const __hip_bfloat16* p, size_t i) {
  return __bfloat162float(p[i]);
}
__device__ __forceinline__ void st_from_f32(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st_from_f32(__half* p, size_t i, float v) { p[i] = __float2half(v); }
__device__ __forceinline__ void st_from_f32(__hip_bfloat16* p, size_t i, float v) {
  p[i] = __float2bfloat16(v);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
    bce_kernel(size_t batch, const T* __restrict__ logit, const float* __restrict__ label,
               float grad_scale, T* __restrict__ dlogit, float* __restrict__ partial) {
  __shared__ float smem[kBlock / 64];
  float val = 0.f;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < batch;
       i += (size_t)gridDim.x * kBlock) {
    const float x = ld_as_f32(logit, i);
    const float y = label[i];
    float g;
    if (x >= 0.f) {
      const float e = expf(-x);
      g = (1.f - y) - e / (1.f + e);
      val += x * (1.f - y) + logf(1.f + e);
    } else {
      const float e = expf(x);
      g = -y + e / (1.f + e);
      val += -x * y + logf(1.f + e);
    }
    if (dlogit) st_from_f32(dlogit, i, g * grad_scale);
  }
  const float tot = block_reduce_sum<float, kBlock>(val, smem);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kBlock)
    bce_finish_kernel(int blocks, size_t batch, const float* __restrict__ partial,
                      float* __restrict__ loss) {
  __shared__ float smem[kBlock / 64];
  const float v = (int)threadIdx.x < blocks ? partial[threadIdx.x] : 0.f;
  const float tot = block_reduce_sum<float, kBlock>(v, smem);
  if (threadIdx.x == 0) *loss = tot / (float)batch;
}

// ---- logit head: the last fully connected layer (K -> 1), BinaryCrossEntropyLoss and their
//      backward in ONE pass over the activations (MLPLayer's last GEMM + loss.cu:231-262).  As
//      library calls this is a GEMV forward, a rank-1 dgrad and a K = batch reduction for the weight
//      gradient -- three badly shaped GEMMs (16 + 17 + 69 us at batch 65536, K = 256) around the
//      loss kernels; here every row of x is read once: z = x.w + b, loss term, dz = (sigmoid(z) -
//      y) * grad_scale, dx = dz * w written back, dw / db / loss accumulated per wavefront and
//      reduced in fixed order (deterministic).
constexpr int kHeadBlocks = 1024;
constexpr int kHeadMaxSeg = 8;  // K <= 64 lanes * 4 elements * 8 segments = 2048

template <typename T>
__device__ __forceinline__ float4 ld4_as_f32(const T* p);
template <>
__device__ __forceinline__ float4 ld4_as_f32<__hip_bfloat16>(const __hip_bfloat16* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u),
                     __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
}
template <>
__device__ __forceinline__ float4 ld4_as_f32<__half>(const __half* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  const __half2 a = *reinterpret_cast<const __half2*>(&r.x), b = *reinterpret_cast<const __half2*>(&r.y);
  const float2 fa = __half22float2(a), fb = __half22float2(b);
  return make_float4(fa.x, fa.y, fb.x, fb.y);
}
template <typename T>
__device__ __forceinline__ void st4_from_f32(T* p, float4 v);
template <>
__device__ __forceinline__ void st4_from_f32<__hip_bfloat16>(__hip_bfloat16* p, float4 v) {
  __hip_bfloat16 h[4] = {__float2bfloat16(v.x), __float2bfloat16(v.y), __float2bfloat16(v.z),
                         __float2bfloat16(v.w)};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(h);
}
template <>
__device__ __forceinline__ void st4_from_f32<__half>(__half* p, float4 v) {
  __half h[4] = {__float2half_rn(v.x), __float2half_rn(v.y), __float2half_rn(v.z),
                 __float2half_rn(v.w)};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(h);
}

// partial layout per block: [K dw][1 db][1 loss].  A wavefront takes ROWS rows per iteration (ROWS
// independent row loads in flight); NSEG = ceil(K / 256) segments of 64 lanes x 4 elements.
template <typename T, int NSEG, int ROWS>
__global__ void __launch_bounds__(kBlock)
    logit_head_kernel(size_t batch, int K, const T* __restrict__ x, const T* __restrict__ w,
                      const T* __restrict__ bias, const float* __restrict__ label,
                      float grad_scale, T* __restrict__ dx, float* __restrict__ partial) {
  HCTR_DYN_LDS(float, lds);  // [waves][K + 2]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wr[NSEG], acc[NSEG];
#pragma unroll
  for (int j = 0; j < NSEG; j++) {
    const int k = j * 256 + lane * 4;
    wr[j] = k < K ? ld4_as_f32<T>(w + k) : zero4;
    acc[j] = zero4;
  }
  const float b0 = ld_as_f32(bias, 0);
  float db = 0.f, loss = 0.f;
  const size_t nw = (size_t)gridDim.x * (kBlock / 64);
  for (size_t r0 = ((size_t)blockIdx.x * (kBlock / 64) + wv) * ROWS; r0 < batch; r0 += nw * ROWS) {
    float4 xv[ROWS][NSEG];
    float dot[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
      const size_t r = r0 + i;
      dot[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NSEG; j++) {
        const int k = j * 256 + lane * 4;
        xv[i][j] = (r < batch && k < K) ? ld4_as_f32<T>(x + r * (size_t)K + k) : zero4;
        dot[i] += xv[i][j].x * wr[j].x + xv[i][j].y * wr[j].y + xv[i][j].z * wr[j].z +
                  xv[i][j].w * wr[j].w;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < ROWS; i++) dot[i] += __shfl_xor(dot[i], o);
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
      const size_t r = r0 + i;
      if (r >= batch) break;  // wave-uniform
      const float z = dot[i] + b0;
      const float y = label[r];
      float g, l;
      if (z >= 0.f) {
        const float e = expf(-z);
        g = (1.f - y) - e / (1.f + e);
        l = z * (1.f - y) + logf(1.f + e);
      } else {
        const float e = expf(z);
        g = -y + e / (1.f + e);
        l = -z * y + logf(1.f + e);
      }
      const float dz = g * grad_scale;
      loss += l;
      db += dz;
#pragma unroll
      for (int j = 0; j < NSEG; j++) {
        const int k = j * 256 + lane * 4;
        if (k < K) {
          acc[j].x += dz * xv[i][j].x;
          acc[j].y += dz * xv[i][j].y;
          acc[j].z += dz * xv[i][j].z;
          acc[j].w += dz * xv[i][j].w;
          if (dx)
            st4_from_f32<T>(dx + r * (size_t)K + k, make_float4(dz * wr[j].x, dz * wr[j].y,
                                                                 dz * wr[j].z, dz * wr[j].w));
        }
      }
    }
  }
  // wavefronts of the block -> LDS -> one partial per block (waves added in order)
  float* mine = lds + wv * (K + 2);
#pragma unroll
  for (int j = 0; j < NSEG; j++) {
    const int k = j * 256 + lane * 4;
    if (k < K) {
      mine[k] = acc[j].x;
      mine[k + 1] = acc[j].y;
      mine[k + 2] = acc[j].z;
      mine[k + 3] = acc[j].w;
    }
  }
  if (lane == 0) {
    mine[K] = db;
    mine[K + 1] = loss;
  }
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * (K + 2);
  for (int k = threadIdx.x; k < K + 2; k += kBlock) {
    float t = 0.f;
    for (int q = 0; q < kBlock / 64; q++) t += lds[q * (K + 2) + k];
    out[k] = t;
  }
}

// fixed-order sum of the block partials: a 1024-thread workgroup owns 64 columns; 16 row groups sum
// consecutive chunks of the partial blocks (coalesced 256-byte reads), then the group sums are
// added in group order
__global__ void __launch_bounds__(1024)
    logit_head_finish_kernel(int blocks, int K, size_t batch, const float* __restrict__ partial,
                             float* __restrict__ dw, float* __restrict__ db,
                             float* __restrict__ loss) {
  __shared__ float part[16][64];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const int chunk = (blocks + 15) / 16;
  const int b0 = grp * chunk, b1 = min(blocks, b0 + chunk);
  float t = 0.f;
  if (k < K + 2) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {  // 8 independent loads in flight, added in order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(b + u) * (K + 2) + k];
#pragma unroll
      for (int u = 0; u < 8; u++) t += v[u];
    }
    for (; b < b1; b++) t += partial[(size_t)b * (K + 2) + k];
  }
  part[grp][c] = t;
  __syncthreads();
  if (grp == 0 && k < K + 2) {
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g++) tot += part[g][c];
    if (k < K) dw[k] = tot;
    else if (k == K) *db = tot;
    else *loss = tot / (float)batch;
  }
}

// ---- first layer of the bottom MLP: y = relu(x W^T + b) with a handful of input features (K <= 16,
//      DLRM: 13 dense features -> 512).  As a GEMM it is all output traffic and no arithmetic
//      (library kernel: 30 us forward; ReLU backward + bias + a K = batch weight-gradient GEMM with
//      13 columns: 72 us).  Here 128 lanes own one row at a time, a lane 4 consecutive outputs
//      with their K weights in registers; the backward folds dz = dy * (y > 0) into the
//      weight-gradient sum, so dz is never written (the layer has no data gradient).
constexpr int kSkinnyK = 16;       // padded K
constexpr int kSkinnyBlocks = 256;
constexpr int kSkinnyLanes = 128;  // lanes per row, 4 outputs each (N <= 512)
constexpr int kSkinnyBwdBlock = 1024;
constexpr bool kSkinnyBwdMfmaDefault = true;
constexpr int kSkinnyUnroll = 2;   // rows per step and row group in the backward (x2 in flight)

template <typename T>
__device__ __forceinline__ float round16(float v);
template <>
__device__ __forceinline__ float round16<__hip_bfloat16>(float v) {
  return __bfloat162float(__float2bfloat16(v));
}
template <>
__device__ __forceinline__ float round16<__half>(float v) {
  return __half2float(__float2half_rn(v));
}

template <typename T>
__device__ __forceinline__ float4 cvt4_as_f32(uint2 r);
template <>
__device__ __forceinline__ float4 cvt4_as_f32<__hip_bfloat16>(uint2 r) {
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u),
                     __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
}
template <>
__device__ __forceinline__ float4 cvt4_as_f32<__half>(uint2 r) {
  const float2 fa = __half22float2(*reinterpret_cast<const __half2*>(&r.x));
  const float2 fb = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
  return make_float4(fa.x, fa.y, fb.x, fb.y);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// x fp32 [B][K] (rounded to T on load, as the 16-bit GEMM path does), w T [N][K], bias T [N].
// A block owns R consecutive rows: their features are staged (rounded, zero-padded to kSkinnyK)
// in LDS once, so the row loop has no global load to wait on -- broadcast LDS reads, packed FMAs
// and a stream of stores.
template <typename T>
__global__ void __launch_bounds__(kBlock)
    skinny_fc_fwd_kernel(size_t batch, int K, int N, int R, const float* __restrict__ x,
                         const T* __restrict__ w, const T* __restrict__ bias, T* __restrict__ y) {
  HCTR_DYN_LDS(float, xs);  // [R][kSkinnyK]
  const int col = threadIdx.x % kSkinnyLanes;
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kSkinnyLanes);
  const bool live = col * 4 < N;
  const int n0 = live ? col * 4 : 0;
  const size_t rb = (size_t)blockIdx.x * R;
  v2f wr[4][kSkinnyK / 2];
  float br[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    br[j] = ld_as_f32(bias, (size_t)(n0 + j));
#pragma unroll
    for (int k = 0; k < kSkinnyK; k++) {
      // clamped index + select: 64 independent loads, no branch (and no wait) per element
      const float t = ld_as_f32(w, (size_t)(n0 + j) * K + (k < K ? k : K - 1));
      const float v = k < K ? t : 0.f;
      if (k & 1) wr[j][k / 2].y = v;
      else wr[j][k / 2].x = v;
    }
  }
  const size_t last = batch - 1;
  for (int idx = threadIdx.x; idx < R * kSkinnyK; idx += kBlock) {
    const int row = idx / kSkinnyK, k = idx % kSkinnyK;
    const float t = x[min(rb + row, last) * (size_t)K + (k < K ? k : K - 1)];
    xs[idx] = k < K ? round16<T>(t) : 0.f;
  }
  __syncthreads();
  constexpr int kRows = kBlock / kSkinnyLanes;
  for (int row = sub; row < R; row += kRows) {
    const size_t r = rb + row;
    if (r >= batch) break;  // scalar condition
    const float4* xp = reinterpret_cast<const float4*>(xs + row * kSkinnyK);
    v2f acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kSkinnyK / 4; q++) {
      const float4 xv = xp[q];  // same address in every lane: broadcast
      const v2f lo = v2f{xv.x, xv.y}, hi = v2f{xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[j] += lo * wr[j][2 * q];
        acc[j] += hi * wr[j][2 * q + 1];
      }
    }
    if (live)
      st4_from_f32<T>(y + r * (size_t)N + n0,
                      make_float4(fmaxf(acc[0].x + acc[0].y + br[0], 0.f),
                                  fmaxf(acc[1].x + acc[1].y + br[1], 0.f),
                                  fmaxf(acc[2].x + acc[2].y + br[2], 0.f),
                                  fmaxf(acc[3].x + acc[3].y + br[3], 0.f)));
  }
}

// partial per block: [N][kSkinnyK] dw and [N] db interleaved as [N][kSkinnyK + 1].  The loads of the
// next kSkinnyUnroll rows (raw 16-bit words) are issued before the current ones are consumed; row
// numbers are scalar, rows past the end are clamped for the loads and count with weight zero.
template <typename T>
__global__ void __launch_bounds__(kSkinnyBwdBlock)
    skinny_fc_bwd_kernel(size_t batch, int K, int N, const float* __restrict__ x,
                         const T* __restrict__ dy, const T* __restrict__ y,
                         float* __restrict__ partial) {
  HCTR_DYN_LDS(float, lds);  // [N * (kSkinnyK + 1)]
  const int lane = threadIdx.x & 63;
  const int col = threadIdx.x % kSkinnyLanes;
  const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x / kSkinnyLanes);
  const bool live = col * 4 < N;
  const int n0 = live ? col * 4 : 0;
  const int kl = lane < K ? lane : 0;
  v2f acc[4][kSkinnyK / 2];
  float dbv[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    dbv[j] = 0.f;
#pragma unroll
    for (int k = 0; k < kSkinnyK / 2; k++) acc[j][k] = v2f{0.f, 0.f};
  }
  constexpr int kRows = kSkinnyBwdBlock / kSkinnyLanes;
  const size_t step = (size_t)gridDim.x * kRows;
  const size_t first = (size_t)blockIdx.x * kRows + sub;
  const size_t last = batch - 1;
  uint2 g[kSkinnyUnroll], a[kSkinnyUnroll];
  float xl[kSkinnyUnroll];
#pragma unroll
  for (int u = 0; u < kSkinnyUnroll; u++) {
    const size_t r = min(first + (size_t)u * step, last);
    g[u] = *reinterpret_cast<const uint2*>(dy + r * (size_t)N + n0);
    a[u] = *reinterpret_cast<const uint2*>(y + r * (size_t)N + n0);
    xl[u] = x[r * (size_t)K + kl];
  }
  for (size_t r0 = first; r0 < batch; r0 += step * kSkinnyUnroll) {
    uint2 gn[kSkinnyUnroll], an[kSkinnyUnroll];
    float xn[kSkinnyUnroll];
#pragma unroll
    for (int u = 0; u < kSkinnyUnroll; u++) {
      const size_t r = min(r0 + (size_t)(u + kSkinnyUnroll) * step, last);
      gn[u] = *reinterpret_cast<const uint2*>(dy + r * (size_t)N + n0);
      an[u] = *reinterpret_cast<const uint2*>(y + r * (size_t)N + n0);
      xn[u] = x[r * (size_t)K + kl];
    }
#pragma unroll
    for (int u = 0; u < kSkinnyUnroll; u++) {
      if (r0 + (size_t)u * step < batch) {  // scalar condition
        // dz as the unfused path hands it to its GEMM
        const float4 gf = cvt4_as_f32<T>(g[u]), af = cvt4_as_f32<T>(a[u]);
        const float d[4] = {af.x > 0.f ? gf.x : 0.f, af.y > 0.f ? gf.y : 0.f,
                            af.z > 0.f ? gf.z : 0.f, af.w > 0.f ? gf.w : 0.f};
        const float xr = lane < K ? round16<T>(xl[u]) : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) dbv[j] += d[j];
#pragma unroll
        for (int k = 0; k < kSkinnyK; k += 2) {
          const v2f xk = v2f{__shfl(xr, k), __shfl(xr, k + 1)};
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j][k / 2] += xk * d[j];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kSkinnyUnroll; u++) {
      g[u] = gn[u];
      a[u] = an[u];
      xl[u] = xn[u];
    }
  }
  // the block's row groups add their sums in a fixed order through LDS
  const int stride = kSkinnyK + 1;
  for (int q = 0; q < kRows; q++) {
    if (sub == q && live) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float* d = lds + (size_t)(n0 + j) * stride;
#pragma unroll
        for (int k = 0; k < kSkinnyK; k++) {
          const float v = (k & 1) ? acc[j][k / 2].y : acc[j][k / 2].x;
          d[k] = (q == 0 ? 0.f : d[k]) + v;
        }
        d[kSkinnyK] = (q == 0 ? 0.f : d[kSkinnyK]) + dbv[j];
      }
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * N * stride;
  for (int i = threadIdx.x; i < N * stride; i += kSkinnyBwdBlock) out[i] = lds[i];
}

// The same backward on the matrix cores.  v_mfma_f32_16x16x4_f32 takes ONE f32 per lane for A and
// for B, so nothing has to be transposed: with D[i][j] = sum_k A[i][k] B[k][j], k = 4 batch rows,
// j = input feature (16 columns: K features, then a column of ones whose sum is db), and
// i = 16 outputs, a lane that loaded 8 consecutive outputs of row (lane >> 4) feeds them to 8
// instructions -- instruction t owns the outputs {8 i + t}: which outputs form a tile is free.
// Products of 16-bit values are exact in f32; the sums are f32 in a fixed order.  A wavefront owns
// 128 outputs x a share of the rows (8 accumulators of 4 registers), 16 rows per step in flight;
// the wavefronts of a block that share outputs add up through LDS in a fixed order.
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kMfmaUnroll = 4;  // 4-row groups per step and wavefront

template <typename T>
__global__ void __launch_bounds__(kSkinnyBwdBlock)
    skinny_fc_bwd_mfma_kernel(size_t batch, int K, int N, const float* __restrict__ x,
                              const T* __restrict__ dy, const T* __restrict__ y,
                              float* __restrict__ partial) {
  HCTR_DYN_LDS(float, lds);  // [N * (kSkinnyK + 1)]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int kWaves = kSkinnyBwdBlock / 64;
  const int ngroups = (N + 127) / 128;
  const int rsets = kWaves / ngroups;
  const int ng = wv % ngroups, rs = wv / ngroups;
  const bool active = rs < rsets;  // wave-uniform
  const int c = lane & 15, ri = lane >> 4;
  const int n0 = ng * 128 + 8 * c;
  const bool has_n = n0 < N;
  const int n0c = has_n ? n0 : 0;
  const int kc = c < K ? c : K - 1;
  v4f acc[8];
#pragma unroll
  for (int t = 0; t < 8; t++) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  const size_t last = batch - 1;
  const size_t step = (size_t)gridDim.x * rsets;
  if (active) {
    for (size_t g16 = (size_t)blockIdx.x * rsets + rs; g16 * 16 < batch; g16 += step) {
      uint4 gq[kMfmaUnroll], aq[kMfmaUnroll];
      float xv[kMfmaUnroll];
#pragma unroll
      for (int u = 0; u < kMfmaUnroll; u++) {
        const size_t r = min(g16 * 16 + u * 4 + ri, last);
        gq[u] = *reinterpret_cast<const uint4*>(dy + r * (size_t)N + n0c);
        aq[u] = *reinterpret_cast<const uint4*>(y + r * (size_t)N + n0c);
        xv[u] = x[r * (size_t)K + kc];
      }
#pragma unroll
      for (int u = 0; u < kMfmaUnroll; u++) {
        const bool ok = has_n && g16 * 16 + u * 4 + ri < batch;
        const float4 g0 = cvt4_as_f32<T>(make_uint2(gq[u].x, gq[u].y));
        const float4 g1 = cvt4_as_f32<T>(make_uint2(gq[u].z, gq[u].w));
        const float4 a0 = cvt4_as_f32<T>(make_uint2(aq[u].x, aq[u].y));
        const float4 a1 = cvt4_as_f32<T>(make_uint2(aq[u].z, aq[u].w));
        const float d[8] = {(ok && a0.x > 0.f) ? g0.x : 0.f, (ok && a0.y > 0.f) ? g0.y : 0.f,
                            (ok && a0.z > 0.f) ? g0.z : 0.f, (ok && a0.w > 0.f) ? g0.w : 0.f,
                            (ok && a1.x > 0.f) ? g1.x : 0.f, (ok && a1.y > 0.f) ? g1.y : 0.f,
                            (ok && a1.z > 0.f) ? g1.z : 0.f, (ok && a1.w > 0.f) ? g1.w : 0.f};
        const float b = c < K ? round16<T>(xv[u]) : (c == K ? 1.f : 0.f);
#pragma unroll
        for (int t = 0; t < 8; t++)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[t], b, acc[t], 0, 0, 0);
      }
    }
  }
  // acc[t][reg] = column c of output ng * 128 + 8 * (4 * ri + reg) + t; column K carries db
  const int stride = kSkinnyK + 1;
  const int col = c == K ? kSkinnyK : c;
  for (int q = 0; q < rsets; q++) {
    if (active && rs == q && c <= K) {
#pragma unroll
      for (int t = 0; t < 8; t++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
          const int n = ng * 128 + 8 * (4 * ri + reg) + t;
          if (n < N) {
            float* d = lds + (size_t)n * stride + col;
            *d = (q == 0 ? 0.f : *d) + acc[t][reg];
          }
        }
      }
    }
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * N * stride;
  for (int i = threadIdx.x; i < N * stride; i += kSkinnyBwdBlock) out[i] = lds[i];
}

// dw [N][K] and db [N] from the block partials: 64 elements per block, 16 groups of lanes each sum
// a consecutive chunk of the partial blocks (8 loads in flight), group sums added in group order
__global__ void __launch_bounds__(1024)
    skinny_fc_finish_kernel(int blocks, int K, int N, const float* __restrict__ partial,
                            float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float part[16][64];
  const int stride = kSkinnyK + 1;
  const int total = N * stride;
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const int chunk = (blocks + 15) / 16;
  const int b0 = grp * chunk, b1 = min(blocks, b0 + chunk);
  float t = 0.f;
  if (i < total) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = partial[(size_t)(b + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; u++) t += v[u];
    }
    for (; b < b1; b++) t += partial[(size_t)b * total + i];
  }
  part[grp][c] = t;
  __syncthreads();
  if (grp == 0 && i < total) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) tot += part[q][c];
    const int n = i / stride, k = i % stride;
    if (k == kSkinnyK) db[n] = tot;
    else if (k < K) dw[(size_t)n * K + k] = tot;
  }
}

constexpr int kCrossBwdWaves = 256 * 4;  // waves used by the cross backward (deterministic reduce)

}  // namespace
}  // namespace hctr

using namespace hctr;

namespace hctr {
namespace {
// single-wavefront workgroups per CU of the 16-bit interaction kernels (which: 0 forward,
// 1 backward); HCTR_INTER_WAVES=f,b overrides (measurements)
int inter_waves_per_cu(int which) {
  static const int v[2] = {[] {
                             const char* e = getenv("HCTR_INTER_WAVES");
                             int f = 8;
                             if (e) f = atoi(e);
                             return f > 0 ? f : 8;
                           }(),
                           [] {
                             const char* e = getenv("HCTR_INTER_WAVES");
                             const char* c = e ? strchr(e, ',') : nullptr;
                             int b = c ? atoi(c + 1) : 8;
                             return b > 0 ? b : 8;
                           }()};
  return v[which];
}
}  // namespace
}  // namespace hctr

extern "C" {

static int interaction_fwd_impl(size_t batch, int n_emb, int width, const void* mlp,
                                const void* emb, const uint32_t* row_of, void* out, int dtype,
                                hctr_stream_t stream) {
  HCTR_REQUIRE(n_emb >= 1 && width >= 1, "shape");
  HCTR_REQUIRE(row_of == nullptr ||
                   ((dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16) && n_emb + 1 <= 32 &&
                    (width == 128 || width == 64 || width == 32) &&
                    reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(emb) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(out) % 16 == 0),
               "indexed interaction: 16-bit rows, width 32/64/128, <= 31 embeddings, 16-byte "
               "aligned buffers");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(mlp && emb && out, "null pointer");
  hipStream_t s = as_stream(stream);
  const int n_ins = n_emb + 1;
  const int out_len = width + n_ins * (n_ins - 1) / 2 + 1;
  const bool a16 = reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(emb) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0;
  const int grid = grid_for(batch, kWavesPerBlock, 256 * 2);
  if (dtype == HCTR_EMB_F32 && n_ins <= 32 && a16 &&
      (width == 128 || width == 64 || width == 32 || width == 16)) {
    const int stage_len = (out_len + 3) & ~3;
    const int grid1 = (int)(batch < (size_t)(256 * 8) ? batch : (size_t)(256 * 8));
#define HCTR_IFWD(W_)                                                                        \
  {                                                                                          \
    const size_t lds = (size_t)((n_ins + 1) * InterCfg<W_>::LD + stage_len) * 4;             \
    hipLaunchKernelGGL(interaction_fwd_mfma_kernel<W_>, dim3(grid1), dim3(64), lds, s, batch, \
                       n_emb, (const float*)mlp, (const float*)emb, (float*)out, out_len);   \
  }
    switch (width) {
      case 128: HCTR_IFWD(128) break;
      case 64: HCTR_IFWD(64) break;
      case 32: HCTR_IFWD(32) break;
      default: HCTR_IFWD(16) break;
    }
#undef HCTR_IFWD
  } else if ((dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16) && n_ins <= 32 && a16 &&
             (width == 128 || width == 64 || width == 32 || width == 16)) {
    const int stage_len = (out_len + 7) & ~7;
    const size_t gmax = (size_t)256 * (size_t)inter_waves_per_cu(0);
    const int grid1 = (int)(batch < gmax ? batch : gmax);
    const bool bf = dtype == HCTR_EMB_BF16;
#define HCTR_IFWD16(W_)                                                                         \
  {                                                                                             \
    const size_t lds = (size_t)((n_ins + 1) * InterCfg16<W_>::LD + stage_len) * 2;              \
    if (bf)                                                                                     \
      hipLaunchKernelGGL((interaction_fwd16_kernel<W_, true>), dim3(grid1), dim3(64), lds, s,   \
                         batch, n_emb, (const unsigned short*)mlp, (const unsigned short*)emb,  \
                         row_of, (unsigned short*)out, out_len);                                \
    else                                                                                        \
      hipLaunchKernelGGL((interaction_fwd16_kernel<W_, false>), dim3(grid1), dim3(64), lds, s,  \
                         batch, n_emb, (const unsigned short*)mlp, (const unsigned short*)emb,  \
                         row_of, (unsigned short*)out, out_len);                                \
  }
    switch (width) {
      case 128: HCTR_IFWD16(128) break;
      case 64: HCTR_IFWD16(64) break;
      case 32: HCTR_IFWD16(32) break;
      default: HCTR_IFWD16(16) break;
    }
#undef HCTR_IFWD16
  } else {
    const size_t lds = (size_t)kWavesPerBlock * n_ins * (width + 1) * 4;
    HCTR_REQUIRE(lds <= 160 * 1024, "interaction: tile does not fit LDS");
    if (dtype == HCTR_EMB_F32)
      hipLaunchKernelGGL(interaction_fwd_generic_kernel<float>, dim3(grid), dim3(kBlock), lds, s,
                         batch, n_emb, width, (const float*)mlp, (const float*)emb, (float*)out,
                         out_len);
    else if (dtype == HCTR_EMB_F16)
      hipLaunchKernelGGL(interaction_fwd_generic_kernel<__half>, dim3(grid), dim3(kBlock), lds, s,
                         batch, n_emb, width, (const __half*)mlp, (const __half*)emb, (__half*)out,
                         out_len);
    else if (dtype == HCTR_EMB_BF16)
      hipLaunchKernelGGL(interaction_fwd_generic_kernel<__hip_bfloat16>, dim3(grid), dim3(kBlock),
                         lds, s, batch, n_emb, width, (const __hip_bfloat16*)mlp,
                         (const __hip_bfloat16*)emb, (__hip_bfloat16*)out, out_len);
    else
      HCTR_REQUIRE(false, "dtype");
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_interaction_fwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         void* out, int dtype, hctr_stream_t stream) {
  return interaction_fwd_impl(batch, n_emb, width, mlp, emb, nullptr, out, dtype, stream);
}

int hctr_interaction_fwd_indexed(size_t batch, int n_emb, int width, const void* mlp,
                                 const void* rows, const uint32_t* row_of, void* out, int dtype,
                                 hctr_stream_t stream) {
  HCTR_REQUIRE(row_of, "null pointer");
  return interaction_fwd_impl(batch, n_emb, width, mlp, rows, row_of, out, dtype, stream);
}

int hctr_interaction_fwd_gather(size_t batch, int n_emb, int width, const void* mlp,
                                const float* table, const uint64_t* value_index, void* pooled,
                                void* out, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(n_emb >= 1 && n_emb <= 31, "interaction_fwd_gather: 1 .. 31 embeddings");
  HCTR_REQUIRE(dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16,
               "interaction_fwd_gather: 16-bit vectors (fp16 / bf16)");
  HCTR_REQUIRE(width == 128 || width == 64 || width == 32 || width == 16,
               "interaction_fwd_gather: width 16 / 32 / 64 / 128");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(mlp && table && value_index && pooled && out, "null pointer");
  HCTR_REQUIRE(reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(pooled) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0,
               "interaction_fwd_gather: 16-byte aligned buffers");
  hipStream_t s = as_stream(stream);
  const int n_ins = n_emb + 1;
  const int out_len = width + n_ins * (n_ins - 1) / 2 + 1;
  const int stage_len = (out_len + 7) & ~7;
  // resident wavefronts per CU: every one keeps a sample's rows (14 x 16 B per lane) in flight,
  // and with random rows an iteration lasts as long as that round trip -- more waves, more of
  // them overlapped (HCTR_GATHER_WAVES overrides)
  static const int waves = [] {
    const char* e = getenv("HCTR_GATHER_WAVES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 8;
  }();
  const size_t gmax = (size_t)256 * (size_t)waves;
  const int grid1 = (int)(batch < gmax ? batch : gmax);
  const bool bf = dtype == HCTR_EMB_BF16;
#define HCTR_IFG16(W_)                                                                          \
  {                                                                                             \
    const size_t lds = (size_t)((n_ins + 1) * InterCfg16<W_>::LD + stage_len) * 2;              \
    if (bf)                                                                                     \
      hipLaunchKernelGGL((interaction_fwd16_gather_kernel<W_, true>), dim3(grid1), dim3(64),    \
                         lds, s, batch, n_emb, (const unsigned short*)mlp, table, value_index,  \
                         (unsigned short*)pooled, (unsigned short*)out, out_len);               \
    else                                                                                        \
      hipLaunchKernelGGL((interaction_fwd16_gather_kernel<W_, false>), dim3(grid1), dim3(64),   \
                         lds, s, batch, n_emb, (const unsigned short*)mlp, table, value_index,  \
                         (unsigned short*)pooled, (unsigned short*)out, out_len);               \
  }
  switch (width) {
    case 128: HCTR_IFG16(128) break;
    case 64: HCTR_IFG16(64) break;
    case 32: HCTR_IFG16(32) break;
    default: HCTR_IFG16(16) break;
  }
#undef HCTR_IFG16
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

static int interaction_bwd_impl(size_t batch, int n_emb, int width, const void* mlp,
                                const void* emb, const uint32_t* row_of, const void* top_grad,
                                void* mlp_grad, void* emb_grad, int dtype, hctr_stream_t stream,
                                const uint32_t* grad_map = nullptr) {
  HCTR_REQUIRE(n_emb >= 1 && width >= 1, "shape");
  {
    const int n_ins_ = n_emb + 1;
    const int out_len_ = width + n_ins_ * (n_ins_ - 1) / 2 + 1;
    HCTR_REQUIRE(row_of == nullptr ||
                     ((dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16) && n_ins_ <= 32 &&
                      (width == 128 || width == 64 || width == 32) && out_len_ % 8 == 0 &&
                      reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(emb) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(top_grad) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(mlp_grad) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(emb_grad) % 16 == 0),
                 "indexed interaction: 16-bit rows, width 32/64/128, <= 31 embeddings, output "
                 "length % 8 == 0, 16-byte aligned buffers");
  }
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(mlp && emb && top_grad && mlp_grad && emb_grad, "null pointer");
  hipStream_t s = as_stream(stream);
  const int n_ins = n_emb + 1;
  const int out_len = width + n_ins * (n_ins - 1) / 2 + 1;
  const bool a16 = reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(emb) % 16 == 0;
  const int grid = grid_for(batch, kWavesPerBlock, 256 * 2);
  const bool g16 = (out_len % 4 == 0) && reinterpret_cast<uintptr_t>(top_grad) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(mlp_grad) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(emb_grad) % 16 == 0;
  if (dtype == HCTR_EMB_F32 && n_ins <= 32 && a16 && g16 &&
      (width == 128 || width == 64 || width == 32)) {
    const int grid1 = (int)(batch < (size_t)(256 * 8) ? batch : (size_t)(256 * 8));
    const int n_pairs = n_ins * (n_ins - 1) / 2;
#define HCTR_IBWD(W_)                                                                          \
  {                                                                                            \
    const size_t lds = (size_t)(InterCfg<W_>::XT + 32 * 36) * 4 + (size_t)((n_pairs + 7) & ~7) * 2; \
    hipLaunchKernelGGL(interaction_bwd_mfma_kernel<W_>, dim3(grid1), dim3(64), lds, s, batch,  \
                       n_emb, (const float*)mlp, (const float*)emb, (const float*)top_grad,    \
                       (float*)mlp_grad, (float*)emb_grad, out_len);                           \
  }
    switch (width) {
      case 128: HCTR_IBWD(128) break;
      case 64: HCTR_IBWD(64) break;
      default: HCTR_IBWD(32) break;
    }
#undef HCTR_IBWD
  } else if ((dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16) && n_ins <= 32 && a16 &&
             (out_len % 8 == 0) && reinterpret_cast<uintptr_t>(top_grad) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(mlp_grad) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(emb_grad) % 16 == 0 &&
             (width == 128 || width == 64 || width == 32)) {
    const size_t gmax = (size_t)256 * (size_t)inter_waves_per_cu(1);
    const int grid1 = (int)(batch < gmax ? batch : gmax);
    const int n_pairs = n_ins * (n_ins - 1) / 2;
    const bool bf = dtype == HCTR_EMB_BF16;
#define HCTR_IBWD16(W_)                                                                          \
  {                                                                                              \
    const size_t lds = (size_t)(InterCfg16<W_>::XT + 32 * 40 + ((n_pairs + 7) & ~7)) * 2;        \
    if (bf)                                                                                      \
      hipLaunchKernelGGL((interaction_bwd16_kernel<W_, true>), dim3(grid1), dim3(64), lds, s,    \
                         batch, n_emb, (const unsigned short*)mlp, (const unsigned short*)emb,   \
                         row_of, (const unsigned short*)top_grad, (unsigned short*)mlp_grad,     \
                         (unsigned short*)emb_grad, out_len, grad_map);                          \
    else                                                                                         \
      hipLaunchKernelGGL((interaction_bwd16_kernel<W_, false>), dim3(grid1), dim3(64), lds, s,   \
                         batch, n_emb, (const unsigned short*)mlp, (const unsigned short*)emb,   \
                         row_of, (const unsigned short*)top_grad, (unsigned short*)mlp_grad,     \
                         (unsigned short*)emb_grad, out_len, grad_map);                          \
  }
    switch (width) {
      case 128: HCTR_IBWD16(128) break;
      case 64: HCTR_IBWD16(64) break;
      default: HCTR_IBWD16(32) break;
    }
#undef HCTR_IBWD16
  } else {
    const size_t lds = (size_t)kWavesPerBlock * (n_ins * (width + 1) + n_ins * n_ins) * 4;
    HCTR_REQUIRE(lds <= 160 * 1024, "interaction: tile does not fit LDS");
    if (dtype == HCTR_EMB_F32)
      hipLaunchKernelGGL(interaction_bwd_generic_kernel<float>, dim3(grid), dim3(kBlock), lds, s,
                         batch, n_emb, width, (const float*)mlp, (const float*)emb,
                         (const float*)top_grad, (float*)mlp_grad, (float*)emb_grad, out_len);
    else if (dtype == HCTR_EMB_F16)
      hipLaunchKernelGGL(interaction_bwd_generic_kernel<__half>, dim3(grid), dim3(kBlock), lds, s,
                         batch, n_emb, width, (const __half*)mlp, (const __half*)emb,
                         (const __half*)top_grad, (__half*)mlp_grad, (__half*)emb_grad, out_len);
    else if (dtype == HCTR_EMB_BF16)
      hipLaunchKernelGGL(interaction_bwd_generic_kernel<__hip_bfloat16>, dim3(grid), dim3(kBlock),
                         lds, s, batch, n_emb, width, (const __hip_bfloat16*)mlp,
                         (const __hip_bfloat16*)emb, (const __hip_bfloat16*)top_grad,
                         (__hip_bfloat16*)mlp_grad, (__hip_bfloat16*)emb_grad, out_len);
    else
      HCTR_REQUIRE(false, "dtype");
  }
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_interaction_bwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         const void* top_grad, void* mlp_grad, void* emb_grad, int dtype,
                         hctr_stream_t stream) {
  return interaction_bwd_impl(batch, n_emb, width, mlp, emb, nullptr, top_grad, mlp_grad, emb_grad,
                              dtype, stream);
}

int hctr_interaction_bwd_indexed(size_t batch, int n_emb, int width, const void* mlp,
                                 const void* rows, const uint32_t* row_of, const void* top_grad,
                                 void* mlp_grad, void* emb_grad, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(row_of, "null pointer");
  return interaction_bwd_impl(batch, n_emb, width, mlp, rows, row_of, top_grad, mlp_grad, emb_grad,
                              dtype, stream);
}

int hctr_interaction_bwd_indexed_scatter(size_t batch, int n_emb, int width, const void* mlp,
                                         const void* rows, const uint32_t* row_of,
                                         const void* top_grad, void* mlp_grad, void* grad_rows,
                                         int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(row_of, "null pointer");
  return interaction_bwd_impl(batch, n_emb, width, mlp, rows, row_of, top_grad, mlp_grad, grad_rows,
                              dtype, stream, row_of);
}

#define HCTR_CROSS_DISPATCH(MACRO)       \
  if (npl <= 1) MACRO(1)                 \
  else if (npl <= 2) MACRO(2)            \
  else if (npl <= 4) MACRO(4)            \
  else if (npl <= 8) MACRO(8)            \
  else if (npl <= 16) MACRO(16)          \
  else if (npl <= 32) MACRO(32)          \
  else MACRO(64)

int hctr_cross_v1_fwd(size_t batch, int width, int layers, const float* x0, const float* kernels,
                      const float* biases, float* outputs, float* hiddens, hctr_stream_t stream) {
  HCTR_REQUIRE(width >= 1 && layers >= 1, "shape");
  HCTR_REQUIRE(width <= 64 * 64, "cross v1: width > 4096 not supported");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(x0 && kernels && biases && outputs && hiddens, "null pointer");
  hipStream_t s = as_stream(stream);
  const int npl = (width + 63) / 64;
  const int grid = grid_for(batch * 64, kBlock);
#define HCTR_CF(N_)                                                                            \
  hipLaunchKernelGGL(cross_v1_fwd_kernel<N_>, dim3(grid), dim3(kBlock), 0, s, batch, width,    \
                     layers, x0, kernels, biases, outputs, hiddens);
  HCTR_CROSS_DISPATCH(HCTR_CF)
#undef HCTR_CF
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_cross_v1_bwd_workspace_bytes(size_t batch, int width, int layers) {
  (void)batch;
  return (size_t)kCrossBwdWaves * layers * 2 * width * sizeof(float);
}

int hctr_cross_v1_bwd(size_t batch, int width, int layers, const float* x0, const float* kernels,
                      const float* outputs, const float* hiddens, const float* out_grad,
                      float* in_grad, float* kernel_grads, float* bias_grads, float* workspace,
                      hctr_stream_t stream) {
  HCTR_REQUIRE(width >= 1 && layers >= 1, "shape");
  HCTR_REQUIRE(width <= 64 * 64, "cross v1: width > 4096 not supported");
  HCTR_REQUIRE(x0 && kernels && outputs && hiddens && out_grad && in_grad && kernel_grads &&
                   bias_grads && workspace,
               "null pointer");
  hipStream_t s = as_stream(stream);
  const int npl = (width + 63) / 64;
  const int grid = kCrossBwdWaves / kWavesPerBlock;
  const size_t lds_bytes = (size_t)layers * 2 * width * sizeof(float);
  if (lds_bytes <= 60 * 1024) {
#define HCTR_CB(N_)                                                                           \
  hipLaunchKernelGGL((cross_v1_bwd_kernel<N_, true>), dim3(kCrossBwdWaves), dim3(64), lds_bytes, \
                     s, batch, width, layers, x0, kernels, outputs, hiddens, out_grad, in_grad,  \
                     workspace);
    HCTR_CROSS_DISPATCH(HCTR_CB)
#undef HCTR_CB
  } else {
#define HCTR_CB(N_)                                                                           \
  hipLaunchKernelGGL((cross_v1_bwd_kernel<N_, false>), dim3(grid), dim3(kBlock), 0, s, batch, \
                     width, layers, x0, kernels, outputs, hiddens, out_grad, in_grad, workspace);
    HCTR_CROSS_DISPATCH(HCTR_CB)
#undef HCTR_CB
  }
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(cross_v1_reduce_kernel,
                     dim3((unsigned)ceil_div<size_t>((size_t)layers * 2 * width, 32)),
                     dim3(kBlock), 0, s, (size_t)kCrossBwdWaves, width, layers, workspace,
                     kernel_grads, bias_grads);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_relu_bwd_bias_workspace_bytes(size_t rows, int n) {
  return ceil_div<size_t>(rows, (size_t)kRbRows) * (size_t)n * sizeof(float);
}

int hctr_relu_bwd_bias(size_t rows, int n, const void* dy, const void* y, void* dz, float* db,
                       float* workspace, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(n > 0 && n % 8 == 0, "n must be a multiple of 8");
  HCTR_REQUIRE(dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16, "16-bit dtypes only");
  if (rows == 0) return HCTR_OK;
  HCTR_REQUIRE(dy && y && dz && db && workspace, "null pointer");
  hipStream_t s = as_stream(stream);
  const size_t tiles = ceil_div<size_t>(rows, (size_t)kRbRows);
  const int cw = n / 8 < kBlock ? n / 8 : kBlock;
  const int rg = kBlock / cw;
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(relu_bwd_bias_kernel<true>, dim3((unsigned)tiles), dim3(kBlock), 0, s, rows,
                       n, cw, rg, (const unsigned short*)dy, (const unsigned short*)y,
                       (unsigned short*)dz, workspace);
  else
    hipLaunchKernelGGL(relu_bwd_bias_kernel<false>, dim3((unsigned)tiles), dim3(kBlock), 0, s,
                       rows, n, cw, rg, (const unsigned short*)dy, (const unsigned short*)y,
                       (unsigned short*)dz, workspace);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div<int>(n, 32)), dim3(kBlock), 0, s,
                     tiles, n, workspace, db);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_cross_v2_bwd_step_workspace_bytes(size_t batch, int width) {
  if (batch == 0 || width <= 0) return 0;
  return ceil_div<size_t>(batch, (size_t)cross_step_rows_per_tile(batch, width)) * (size_t)width *
         sizeof(float);
}

int hctr_cross_v2_bwd_step(size_t batch, int width, const void* dy, const void* x0, const void* h,
                           void* acc, void* s0, float* db, float* workspace, int first, int dtype,
                           hctr_stream_t stream) {
  HCTR_REQUIRE(width > 0 && width % 8 == 0, "width must be a multiple of 8");
  HCTR_REQUIRE(dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16, "16-bit dtypes only");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(dy && x0 && h && acc && s0 && db && workspace, "null pointer");
  hipStream_t s = as_stream(stream);
  const int cw = width / 8 < kBlock ? width / 8 : kBlock;
  const int rg = kBlock / cw;
  const int rpt = cross_step_rows_per_tile(batch, width);
  const size_t tiles = ceil_div<size_t>(batch, (size_t)rpt);
  const unsigned col_blocks = (unsigned)ceil_div<int>(width / 8, cw);
#define HCTR_CROSS_STEP(BF_, FIRST_)                                                              \
  hipLaunchKernelGGL((cross_v2_bwd_step_kernel<BF_, FIRST_>), dim3((unsigned)tiles, col_blocks),   \
                     dim3(kBlock), 0, s, batch, width, cw, rg, rpt, (const unsigned short*)dy,     \
                     (const unsigned short*)x0, (const unsigned short*)h, (unsigned short*)acc,    \
                     (unsigned short*)s0, workspace)
  if (dtype == HCTR_EMB_BF16) {
    if (first) HCTR_CROSS_STEP(true, true);
    else HCTR_CROSS_STEP(true, false);
  } else {
    if (first) HCTR_CROSS_STEP(false, true);
    else HCTR_CROSS_STEP(false, false);
  }
#undef HCTR_CROSS_STEP
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(ceil_div<int>(width, 32)), dim3(kBlock), 0, s,
                     tiles, width, workspace, db);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_sum_groups(int groups, size_t n, const void* in, int dtype, float* out,
                    hctr_stream_t stream) {
  HCTR_REQUIRE(groups > 0 && n % 8 == 0, "n must be a multiple of 8");
  HCTR_REQUIRE(dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16, "16-bit dtypes only");
  if (n == 0) return HCTR_OK;
  HCTR_REQUIRE(in && out, "null pointer");
  hipStream_t s = as_stream(stream);
  const size_t n8 = n / 8;
  const dim3 grid((unsigned)ceil_div<size_t>(n8, (size_t)kBlock));
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(sum_groups_kernel<true>, grid, dim3(kBlock), 0, s, groups, n8,
                       (const unsigned short*)in, out);
  else
    hipLaunchKernelGGL(sum_groups_kernel<false>, grid, dim3(kBlock), 0, s, groups, n8,
                       (const unsigned short*)in, out);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_sgd_shadow(size_t n, float lr, float grad_scale, float* w, const float* g, void* w16,
                    int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(n % 4 == 0, "n must be a multiple of 4");
  HCTR_REQUIRE(dtype == HCTR_EMB_BF16 || dtype == HCTR_EMB_F16, "16-bit shadow dtypes only");
  if (n == 0) return HCTR_OK;
  HCTR_REQUIRE(w && g && w16, "null pointer");
  hipStream_t s = as_stream(stream);
  const dim3 grid(grid_for(n / 4, kBlock, 4096));
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(sgd_shadow_kernel<true>, grid, dim3(kBlock), 0, s, n / 4, lr, grad_scale, w,
                       g, (unsigned short*)w16);
  else
    hipLaunchKernelGGL(sgd_shadow_kernel<false>, grid, dim3(kBlock), 0, s, n / 4, lr, grad_scale,
                       w, g, (unsigned short*)w16);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_bce_loss_workspace_bytes(void) { return kBceBlocks * sizeof(float); }

int hctr_bce_loss(size_t batch, const void* logit, const float* label, float grad_scale,
                  void* dlogit, float* loss, float* workspace, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(batch > 0, "empty batch");
  HCTR_REQUIRE(logit && label && loss && workspace, "null pointer");
  hipStream_t s = as_stream(stream);
  const int blocks = (int)(ceil_div<size_t>(batch, (size_t)kBlock) < (size_t)kBceBlocks
                               ? ceil_div<size_t>(batch, (size_t)kBlock)
                               : (size_t)kBceBlocks);
  switch (dtype) {
    case HCTR_EMB_F32:
      hipLaunchKernelGGL(bce_kernel<float>, dim3(blocks), dim3(kBlock), 0, s, batch,
                         (const float*)logit, label, grad_scale, (float*)dlogit, workspace);
      break;
    case HCTR_EMB_F16:
      hipLaunchKernelGGL(bce_kernel<__half>, dim3(blocks), dim3(kBlock), 0, s, batch,
                         (const __half*)logit, label, grad_scale, (__half*)dlogit, workspace);
      break;
    case HCTR_EMB_BF16:
      hipLaunchKernelGGL(bce_kernel<__hip_bfloat16>, dim3(blocks), dim3(kBlock), 0, s, batch,
                         (const __hip_bfloat16*)logit, label, grad_scale, (__hip_bfloat16*)dlogit,
                         workspace);
      break;
    default:
      HCTR_REQUIRE(false, "bad dtype");
  }
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(kBlock), 0, s, blocks, batch, workspace, loss);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_logit_head_workspace_bytes(int k) { return (size_t)kHeadBlocks * (k + 2) * sizeof(float); }

int hctr_logit_head(size_t batch, int k, const void* x, const void* w, const void* bias,
                    const float* label, float grad_scale, void* dx, float* dw, float* db,
                    float* loss, float* workspace, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(batch > 0 && k >= 4 && k % 4 == 0 && k <= 256 * kHeadMaxSeg,
               "logit head: K must be a multiple of 4 and <= 2048");
  HCTR_REQUIRE(dtype == HCTR_EMB_F16 || dtype == HCTR_EMB_BF16, "16-bit activations");
  HCTR_REQUIRE(x && w && bias && label && dw && db && loss && workspace, "null pointer");
  HCTR_REQUIRE(reinterpret_cast<uintptr_t>(x) % 8 == 0 && reinterpret_cast<uintptr_t>(w) % 8 == 0 &&
                   (dx == nullptr || reinterpret_cast<uintptr_t>(dx) % 8 == 0),
               "8-byte aligned buffers");
  hipStream_t s = as_stream(stream);
  const int nseg = (k + 255) / 256;
  const int rows = nseg <= 1 ? 8 : (nseg <= 2 ? 4 : (nseg <= 4 ? 2 : 1));
  const int blocks = (int)std::min<size_t>(
      (size_t)kHeadBlocks, ceil_div<size_t>(batch, (size_t)(kBlock / 64) * rows));
  const size_t lds = (size_t)(kBlock / 64) * (k + 2) * sizeof(float);
#define HCTR_HEAD(T_, NSEG_, ROWS_)                                                               \
  hipLaunchKernelGGL((logit_head_kernel<T_, NSEG_, ROWS_>), dim3(blocks), dim3(kBlock), lds, s,   \
                     batch, k, (const T_*)x, (const T_*)w, (const T_*)bias, label, grad_scale,    \
                     (T_*)dx, workspace)
#define HCTR_HEAD_T(T_)                       \
  if (nseg <= 1) HCTR_HEAD(T_, 1, 8);         \
  else if (nseg <= 2) HCTR_HEAD(T_, 2, 4);    \
  else if (nseg <= 4) HCTR_HEAD(T_, 4, 2);    \
  else HCTR_HEAD(T_, 8, 1);
  if (dtype == HCTR_EMB_BF16) {
    HCTR_HEAD_T(__hip_bfloat16)
  } else {
    HCTR_HEAD_T(__half)
  }
#undef HCTR_HEAD_T
#undef HCTR_HEAD
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(logit_head_finish_kernel, dim3(ceil_div<int>(k + 2, 64)), dim3(1024), 0, s,
                     blocks, k, batch, workspace, dw, db, loss);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

static int skinny_check(size_t batch, int k, int n, int dtype) {
  HCTR_REQUIRE(batch > 0 && k >= 1 && k <= kSkinnyK, "skinny fc: 1 <= K <= 16");
  HCTR_REQUIRE(n >= 4 && n % 4 == 0 && n <= 512, "skinny fc: N a multiple of 4, <= 512");
  HCTR_REQUIRE(dtype == HCTR_EMB_F16 || dtype == HCTR_EMB_BF16, "16-bit weights / activations");
  return HCTR_OK;
}

int hctr_skinny_fc_fwd(size_t batch, int k, int n, const float* x, const void* w, const void* bias,
                       void* y, int dtype, hctr_stream_t stream) {
  HCTR_TRY(skinny_check(batch, k, n, dtype));
  HCTR_REQUIRE(x && w && bias && y, "null pointer");
  HCTR_REQUIRE(reinterpret_cast<uintptr_t>(y) % 8 == 0, "8-byte aligned output");
  hipStream_t s = as_stream(stream);
  // R consecutive rows per block (even, <= 256: 16 KB of LDS), about 4 blocks per CU
  size_t rows = ceil_div<size_t>(batch, 1024);
  rows = std::min<size_t>(256, (rows + 1) / 2 * 2);
  const int blocks = (int)ceil_div<size_t>(batch, rows);
  const size_t lds = rows * kSkinnyK * sizeof(float);
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(skinny_fc_fwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(kBlock), lds, s,
                       batch, k, n, (int)rows, x, (const __hip_bfloat16*)w,
                       (const __hip_bfloat16*)bias, (__hip_bfloat16*)y);
  else
    hipLaunchKernelGGL(skinny_fc_fwd_kernel<__half>, dim3(blocks), dim3(kBlock), lds, s, batch, k, n,
                       (int)rows, x, (const __half*)w, (const __half*)bias, (__half*)y);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

size_t hctr_skinny_fc_bwd_workspace_bytes(int n) {
  return (size_t)kSkinnyBlocks * n * (kSkinnyK + 1) * sizeof(float);
}

int hctr_skinny_fc_bwd(size_t batch, int k, int n, const float* x, const void* dy, const void* y,
                       float* dw, float* db, float* workspace, int dtype, hctr_stream_t stream) {
  HCTR_TRY(skinny_check(batch, k, n, dtype));
  HCTR_REQUIRE(x && dy && y && dw && db && workspace, "null pointer");
  HCTR_REQUIRE(reinterpret_cast<uintptr_t>(dy) % 8 == 0 && reinterpret_cast<uintptr_t>(y) % 8 == 0,
               "8-byte aligned activations");
  hipStream_t s = as_stream(stream);
  const size_t lds = (size_t)n * (kSkinnyK + 1) * sizeof(float);
  // matrix-core form: needs a spare input column for db and 16-byte rows (HCTR_SKINNY_BWD=valu|mfma)
  const char* mode = getenv("HCTR_SKINNY_BWD");
  const bool want_mfma = mode ? strcmp(mode, "valu") != 0 : kSkinnyBwdMfmaDefault;
  if (want_mfma && k < kSkinnyK && n % 8 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(y) % 16 == 0) {
    const int rsets = (kSkinnyBwdBlock / 64) / ceil_div<int>(n, 128);
    const int blocks = (int)std::min<size_t>((size_t)kSkinnyBlocks,
                                             ceil_div<size_t>(batch, (size_t)rsets * 16));
    if (dtype == HCTR_EMB_BF16)
      hipLaunchKernelGGL(skinny_fc_bwd_mfma_kernel<__hip_bfloat16>, dim3(blocks),
                         dim3(kSkinnyBwdBlock), lds, s, batch, k, n, x, (const __hip_bfloat16*)dy,
                         (const __hip_bfloat16*)y, workspace);
    else
      hipLaunchKernelGGL(skinny_fc_bwd_mfma_kernel<__half>, dim3(blocks), dim3(kSkinnyBwdBlock), lds,
                         s, batch, k, n, x, (const __half*)dy, (const __half*)y, workspace);
    HCTR_LAUNCH_CHECK();
    hipLaunchKernelGGL(skinny_fc_finish_kernel, dim3(ceil_div<int>(n * (kSkinnyK + 1), 64)),
                       dim3(1024), 0, s, blocks, k, n, workspace, dw, db);
    HCTR_LAUNCH_CHECK();
    return HCTR_OK;
  }
  const int blocks = (int)std::min<size_t>(
      (size_t)kSkinnyBlocks, ceil_div<size_t>(batch, kSkinnyBwdBlock / kSkinnyLanes));
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(skinny_fc_bwd_kernel<__hip_bfloat16>, dim3(blocks), dim3(kSkinnyBwdBlock), lds, s,
                       batch, k, n, x, (const __hip_bfloat16*)dy, (const __hip_bfloat16*)y,
                       workspace);
  else
    hipLaunchKernelGGL(skinny_fc_bwd_kernel<__half>, dim3(blocks), dim3(kSkinnyBwdBlock), lds, s, batch, k, n,
                       x, (const __half*)dy, (const __half*)y, workspace);
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(skinny_fc_finish_kernel, dim3(ceil_div<int>(n * (kSkinnyK + 1), 64)),
                     dim3(1024), 0, s, blocks, k, n, workspace, dw, db);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // extern "C"
