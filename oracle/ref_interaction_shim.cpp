// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's device code of InteractionLayer
// (R/HugeCTR/src/layers/interaction_layer.cu:31-955, the anonymous namespace: the fused fp16 kernels
// dotBasedInteractFwdKernel / BwdKernel (+ NonAligned) on nvcuda::wmma tiles with their launch
// wrappers dotBasedInteractFwd / Bwd, and the kernels of the generic path -- concat_kernel,
// gather_concat_fprop / bprop_kernel, transpose_and_add_oneshot), cut out of the checkout by
// oracle/Makefile and executed by the host interpreter of tests/emu (32-lane warps; wmma by its
// contract, ref_shims/cuda/wmma_emu.h).  The generic path's three steps -- concat, the strided
// batched GEMM X * X^T (cuBLAS in the reference: a plain fp32 triple loop here), gather -- are
// launched in the order and with the block shapes of InteractionLayer<T>::fprop_generic /
// bprop_generic (:1046-1110, :1134-1215).
#define REFSHIM_TRIVIAL_HALF
#define REFSHIM_HALF_ARITH
#include <common.hpp>  // oracle/ref_shims/common.hpp

#include <limits>
#include <memory>
#include <type_traits>
#include <vector>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"
#include "ref_shims/cuda/wmma_emu.h"

typedef __half half;
typedef __half2 half2;
#define __align__(n) alignas(n)
#define __launch_bounds__(...)
#define __restrict

namespace HugeCTR {
#include "_ref/gen/interaction_kernels.inc"

namespace {
template <typename T>
void gemm_xxt(const T* x, T* mat, int h, int n_ins, int w) {  // mat[b] = X[b] * X[b]^T
  for (int b = 0; b < h; b++)
    for (int i = 0; i < n_ins; i++)
      for (int j = 0; j < n_ins; j++) {
        float acc = 0.f;
        for (int k = 0; k < w; k++)
          acc += (float)x[((size_t)b * n_ins + i) * w + k] * (float)x[((size_t)b * n_ins + j) * w + k];
        mat[((size_t)b * n_ins + i) * n_ins + j] = (T)acc;
      }
}
}  // namespace
}  // namespace HugeCTR

using namespace HugeCTR;

extern "C" {
// fp16, n_ins < 32: the fused kernel.  out [B][W + n_ins (n_ins - 1) / 2 + 1]
void refinter_fwd16(const void* mlp, const void* emb, void* out, unsigned B, unsigned n_ins,
                    unsigned W) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  dotBasedInteractFwd(mlp, emb, out, B, n_ins, W, nullptr);
  hipemu::set_wave_width(64);
}
// in place as InteractionLayer<__half>::bprop does it: mlp_io / emb_io hold the forward inputs and
// receive their gradients
void refinter_bwd16(void* ugrad, void* mlp_io, void* emb_io, unsigned B, unsigned n_ins, unsigned W) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  dotBasedInteractBwd(ugrad, mlp_io, emb_io, B, n_ins, W, nullptr);
  hipemu::set_wave_width(64);
}
// fp32: the generic path of fprop_generic
void refinter_fwd32(float* mlp, float* emb, float* out, int B, int n_ins, int W) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  const int n_emb = n_ins - 1, out_w = n_ins * W;
  std::vector<float> concat((size_t)B * out_w), mat((size_t)B * n_ins * n_ins);
  const int sm = 4;
  REFEMU_LAUNCH((concat_kernel), (dim3(n_ins, sm, 1), dim3(W <= 128 ? 128 : (W <= 256 ? 256 : 512), 1, 1)),
                true, concat.data(), mlp, emb, B, out_w, W, n_emb);
  gemm_xxt(concat.data(), mat.data(), B, n_ins, W);
  const size_t smem = sizeof(float) * (n_ins * (n_ins + 1) / 2 - n_ins);
  REFEMU_LAUNCH((gather_concat_fprop_kernel), (dim3(sm * 8, 1, 1), dim3(16, 16, 1), smem), out,
                (const float*)mlp, (const float*)mat.data(), B, n_ins, W);
  hipemu::set_wave_width(64);
}
}
