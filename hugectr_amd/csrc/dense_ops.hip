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
#include <hip/hip_fp16.h>

#include "common.h"

namespace hctr {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

template <int W>
__global__ void __launch_bounds__(kBlock)
    interaction_fwd_mfma_kernel(size_t batch, int n_emb, const float* __restrict__ mlp,
                                const float* __restrict__ emb, float* __restrict__ out,
                                int out_len) {
  using C = InterCfg<W>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ins = n_emb + 1;
  const int stage_len = (out_len + 3) & ~3;
  float* xt = smem + wave * (C::XT + stage_len);
  float* stage = xt + C::XT;
  // zero the whole X tile once: pad rows n_ins..31 must read as 0 forever
  for (int i = lane; i < C::XT; i += 64) xt[i] = 0.f;
  __syncthreads();

  const size_t waves_total = (size_t)gridDim.x * kWavesPerBlock;
  const size_t iters = (batch + waves_total - 1) / waves_total;
  const int r = lane & 31, h = lane >> 5;
  constexpr int W4 = W / 4;
  for (size_t it = 0; it < iters; it++) {
    const size_t b = it * waves_total + (size_t)blockIdx.x * kWavesPerBlock + wave;
    const bool valid = b < batch;
    if (valid) {
      // coalesced 16-byte loads: row 0 = mlp[b], rows 1.. = emb[b]
      const float4* m4 = reinterpret_cast<const float4*>(mlp + b * W);
      const float4* e4 = reinterpret_cast<const float4*>(emb + b * (size_t)n_emb * W);
      for (int i = lane; i < n_ins * W4; i += 64) {
        const int row = i / W4, c4 = i % W4;
        float4 v = (row == 0) ? m4[c4] : e4[(size_t)(row - 1) * W4 + c4];
        *reinterpret_cast<float4*>(xt + row * C::LD + c4 * 4) = v;
      }
    }
    __syncthreads();
    if (valid) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // lane (r,h) feeds X[r][h*W/2 + k]; A == B operand because the product is X.X^T.
      const float* xr = xt + r * C::LD + h * (W / 2);
#pragma unroll
      for (int t = 0; t < W / 8; t++) {
        const float4 v = *reinterpret_cast<const float4*>(xr + t * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, v.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, v.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, v.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, v.w, acc, 0, 0, 0);
      }
      // C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
      const int col = r;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (row > col && row < n_ins) stage[W + tri_index(row, col)] = acc[reg];
      }
      for (int i = lane; i < W; i += 64) stage[i] = xt[i];  // mlp passthrough
      if (lane == 0) stage[out_len - 1] = 0.f;              // zero pad column
    }
    __syncthreads();
    if (valid) {
      float* o = out + b * (size_t)out_len;
      if ((out_len & 3) == 0) {
        for (int i = lane; i < out_len / 4; i += 64)
          reinterpret_cast<float4*>(o)[i] = reinterpret_cast<const float4*>(stage)[i];
      } else {
        for (int i = lane; i < out_len; i += 64) o[i] = stage[i];
      }
    }
    __syncthreads();
  }
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
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
__global__ void __launch_bounds__(kBlock)
    interaction_bwd_mfma_kernel(size_t batch, int n_emb, const float* __restrict__ mlp,
                                const float* __restrict__ emb, const float* __restrict__ top_grad,
                                float* __restrict__ mlp_grad, float* __restrict__ emb_grad,
                                int out_len) {
  using C = InterCfg<W>;
  constexpr int GS = 33;  // G row stride (floats)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ins = n_emb + 1;
  float* xt = smem + wave * (C::XT + 32 * GS);
  float* gm = xt + C::XT;
  for (int i = lane; i < C::XT + 32 * GS; i += 64) xt[i] = 0.f;
  __syncthreads();

  const size_t waves_total = (size_t)gridDim.x * kWavesPerBlock;
  const size_t iters = (batch + waves_total - 1) / waves_total;
  const int r = lane & 31, h = lane >> 5;
  constexpr int W4 = W / 4;
  constexpr int NT = W / 32;  // N tiles
  const int n_pairs = n_ins * (n_ins - 1) / 2;
  for (size_t it = 0; it < iters; it++) {
    const size_t b = it * waves_total + (size_t)blockIdx.x * kWavesPerBlock + wave;
    const bool valid = b < batch;
    if (valid) {
      const float4* m4 = reinterpret_cast<const float4*>(mlp + b * W);
      const float4* e4 = reinterpret_cast<const float4*>(emb + b * (size_t)n_emb * W);
      for (int i = lane; i < n_ins * W4; i += 64) {
        const int row = i / W4, c4 = i % W4;
        float4 v = (row == 0) ? m4[c4] : e4[(size_t)(row - 1) * W4 + c4];
        *reinterpret_cast<float4*>(xt + row * C::LD + c4 * 4) = v;
      }
      const float* g = top_grad + b * (size_t)out_len + W;
      for (int p = lane; p < n_pairs; p += 64) {
        int n = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (n * (n - 1) / 2 > p) n--;
        while ((n + 1) * n / 2 <= p) n++;
        const int m = p - n * (n - 1) / 2;
        const float v = g[p];
        gm[n * GS + m] = v;
        gm[m * GS + n] = v;
      }
    }
    __syncthreads();
    if (valid) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; t++)
        acc[t] = (f32x16){0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int ksteps = (n_ins + 1) / 2;  // K = n_ins rounded up to 2
      for (int s = 0; s < ksteps; s++) {
        const int k = 2 * s + h;
        const float a = gm[r * GS + k];  // A[i = r][k]
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const float bv = xt[k * C::LD + t * 32 + r];  // B[k][j = 32 t + r]
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
        }
      }
      const float* gtop = top_grad + b * (size_t)out_len;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int col = t * 32 + r;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          if (row == 0) mlp_grad[b * W + col] = gtop[col] + acc[t][reg];
          else if (row < n_ins) emb_grad[(b * n_emb + (row - 1)) * (size_t)W + col] = acc[t][reg];
        }
      }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
    interaction_bwd_generic_kernel(size_t batch, int n_emb, int W, const T* __restrict__ mlp,
                                   const T* __restrict__ emb, const T* __restrict__ top_grad,
                                   T* __restrict__ mlp_grad, T* __restrict__ emb_grad,
                                   int out_len) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
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
template <int NPL>
__global__ void __launch_bounds__(kBlock)
    cross_v1_bwd_kernel(size_t batch, int w, int layers, const float* __restrict__ x0,
                        const float* __restrict__ kernels, const float* __restrict__ outputs,
                        const float* __restrict__ hiddens, const float* __restrict__ out_grad,
                        float* __restrict__ in_grad, float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * kBlock) >> 6;
  float* my = partials + wave * (size_t)layers * 2 * w;
  for (int i = lane; i < layers * 2 * w; i += 64) my[i] = 0.f;
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
}

__global__ void __launch_bounds__(kBlock)
    cross_v1_reduce_kernel(size_t nwaves, int w, int layers, const float* __restrict__ partials,
                           float* __restrict__ kernel_grads, float* __restrict__ bias_grads) {
  const size_t total = (size_t)layers * 2 * w;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (size_t)gridDim.x * kBlock) {
    float s = 0.f;
    for (size_t q = 0; q < nwaves; q++) s += partials[q * total + i];
    const int l = (int)(i / (2 * w)), rem = (int)(i % (2 * w));
    if (rem < w) kernel_grads[(size_t)l * w + rem] = s;
    else bias_grads[(size_t)l * w + (rem - w)] = s;
  }
}

__global__ void __launch_bounds__(kBlock)
    cross_v2_epilogue_kernel(size_t n, int w, const float* __restrict__ x0,
                             const float* __restrict__ xl, const float* __restrict__ hmat,
                             const float* __restrict__ bias, float* __restrict__ hidden_out,
                             float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (size_t)gridDim.x * kBlock) {
    const float hv = hmat[i] + bias[i % w];
    if (hidden_out) hidden_out[i] = hv;
    out[i] = hv * x0[i] + xl[i];
  }
}

constexpr int kCrossBwdWaves = 256 * 4;  // waves used by the cross backward (deterministic reduce)

}  // namespace
}  // namespace hctr

using namespace hctr;

extern "C" {

int hctr_interaction_fwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         void* out, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(n_emb >= 1 && width >= 1, "shape");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(mlp && emb && out, "null pointer");
  hipStream_t s = as_stream(stream);
  const int n_ins = n_emb + 1;
  const int out_len = width + n_ins * (n_ins - 1) / 2 + 1;
  const bool a16 = reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(emb) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(out) % 16 == 0;
  const int grid = grid_for(batch, kWavesPerBlock, 256 * 2);
  if (dtype == HCTR_EMB_F32 && n_ins <= 32 && a16 && (width == 128 || width == 64 || width == 32 || width == 16)) {
    const int stage_len = (out_len + 3) & ~3;
#define HCTR_IFWD(W_)                                                                        \
  {                                                                                          \
    const size_t lds = (size_t)kWavesPerBlock * (InterCfg<W_>::XT + stage_len) * 4;          \
    HCTR_HIP(hipFuncSetAttribute((const void*)interaction_fwd_mfma_kernel<W_>,                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
    hipLaunchKernelGGL(interaction_fwd_mfma_kernel<W_>, dim3(grid), dim3(kBlock), lds, s,    \
                       batch, n_emb, (const float*)mlp, (const float*)emb, (float*)out,      \
                       out_len);                                                             \
  }
    switch (width) {
      case 128: HCTR_IFWD(128) break;
      case 64: HCTR_IFWD(64) break;
      case 32: HCTR_IFWD(32) break;
      default: HCTR_IFWD(16) break;
    }
#undef HCTR_IFWD
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

int hctr_interaction_bwd(size_t batch, int n_emb, int width, const void* mlp, const void* emb,
                         const void* top_grad, void* mlp_grad, void* emb_grad, int dtype,
                         hctr_stream_t stream) {
  HCTR_REQUIRE(n_emb >= 1 && width >= 1, "shape");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(mlp && emb && top_grad && mlp_grad && emb_grad, "null pointer");
  hipStream_t s = as_stream(stream);
  const int n_ins = n_emb + 1;
  const int out_len = width + n_ins * (n_ins - 1) / 2 + 1;
  const bool a16 = reinterpret_cast<uintptr_t>(mlp) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(emb) % 16 == 0;
  const int grid = grid_for(batch, kWavesPerBlock, 256 * 2);
  if (dtype == HCTR_EMB_F32 && n_ins <= 32 && a16 && (width == 128 || width == 64 || width == 32)) {
#define HCTR_IBWD(W_)                                                                         \
  {                                                                                           \
    const size_t lds = (size_t)kWavesPerBlock * (InterCfg<W_>::XT + 32 * 33) * 4;             \
    HCTR_HIP(hipFuncSetAttribute((const void*)interaction_bwd_mfma_kernel<W_>,                 \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL(interaction_bwd_mfma_kernel<W_>, dim3(grid), dim3(kBlock), lds, s,     \
                       batch, n_emb, (const float*)mlp, (const float*)emb,                    \
                       (const float*)top_grad, (float*)mlp_grad, (float*)emb_grad, out_len);  \
  }
    switch (width) {
      case 128: HCTR_IBWD(128) break;
      case 64: HCTR_IBWD(64) break;
      default: HCTR_IBWD(32) break;
    }
#undef HCTR_IBWD
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
#define HCTR_CB(N_)                                                                           \
  hipLaunchKernelGGL(cross_v1_bwd_kernel<N_>, dim3(grid), dim3(kBlock), 0, s, batch, width,   \
                     layers, x0, kernels, outputs, hiddens, out_grad, in_grad, workspace);
  HCTR_CROSS_DISPATCH(HCTR_CB)
#undef HCTR_CB
  HCTR_LAUNCH_CHECK();
  hipLaunchKernelGGL(cross_v1_reduce_kernel, dim3(grid_for((size_t)layers * 2 * width, kBlock)),
                     dim3(kBlock), 0, s, (size_t)kCrossBwdWaves, width, layers, workspace,
                     kernel_grads, bias_grads);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

int hctr_cross_v2_epilogue(size_t batch, int width, const float* x0, const float* xl,
                           const float* h, const float* bias, float* hidden_out, float* out,
                           hctr_stream_t stream) {
  HCTR_REQUIRE(width >= 1, "shape");
  if (batch == 0) return HCTR_OK;
  HCTR_REQUIRE(x0 && xl && h && bias && out, "null pointer");
  const size_t n = batch * (size_t)width;
  hipLaunchKernelGGL(cross_v2_epilogue_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), n, width, x0, xl, h, bias, hidden_out, out);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // extern "C"
