/* TEST INFRASTRUCTURE ONLY -- the CPU references of InteractionLayer and MultiCrossLayer that live
 * INLINE in the reference's CUDA-bound gtest files, compiled into oracle/_ref/libref_layers.so:
 *   R/test/utest/core23_layer_test/interaction_layer_test.cpp:95-282  (concat, X X^T, lower-
 *       triangle gather; backward: scatter, (dM + dM^T) X, un-concat with the mlp pass-through)
 *   R/test/utest/core23_layer_test/multi_cross_layer_test.cpp:152-432 (helpers, cpu_fprop_,
 *       cpu_fprop_v2_, cpu_bprop_, cpu_bprop_v2_)
 * Those files cannot be compiled (gtest, core23, the GPU layer under test), so the build recipe
 * (oracle/Makefile `ref`) cuts the host-only statement blocks out with sed -- by their first and
 * last lines -- into oracle/_ref/gen/*.inc (generated, git-ignored) and this file supplies only the
 * variables those blocks name.  Not one arithmetic statement below is ours.
 * oracle/hctr_oracle.c's interaction / cross references are pinned against this library in
 * tests/test_ref_layers_cpu.py. */
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

namespace core23 {
template <typename To, typename From>
struct TypeConverter {
  static To value(From v) { return static_cast<To>(v); }
};
}  // namespace core23
template <typename To, typename From>
struct TypeConvert {
  static To convert(From v) { return static_cast<To>(v); }
};
#define HCTR_LOG(...) do { } while (0)

namespace {
template <typename T>
#include "_ref/gen/interaction_get_accum.inc"
template <typename T>
#include "_ref/gen/interaction_get_sec_accum.inc"

/* interaction_layer_test<T>() with the device calls left out: forward into `top`, then backward
 * with `top_grad` as the upstream gradient (the reference back-propagates its own output because
 * its layer works in place; any gradient exercises the same statements) */
template <typename T>
void interaction_ref(size_t height, size_t n_emb, size_t in_width, std::vector<T>& h_bottom_mlp,
                     std::vector<T>& h_bottom_emb, std::vector<T>& top, const T* top_grad) {
#include "_ref/gen/interaction_concat.inc"
  concat_op(true);
#include "_ref/gen/interaction_matmul.inc"
#include "_ref/gen/interaction_gather.inc"
  top = h_ref;
  if (top_grad == nullptr) return;
  std::copy(top_grad, top_grad + h_ref.size(), h_ref.begin());
#include "_ref/gen/interaction_bprop.inc"
  concat_op(false);
}

template <typename T>
struct CrossRef {
  size_t batchsize_, w_;
  int layers_;
  size_t projection_dim_;
  std::vector<T> h_input_, h_input_grad_, h_output_grad_;
  std::vector<std::vector<T>> XUs, h_kernels_, h_biases_, h_outputs_, h_hiddens_, h_kernel_grads_,
      h_bias_grads_;
#include "_ref/gen/cross_cpu_fprop.inc"
#include "_ref/gen/cross_cpu_bprop.inc"
};
}  // namespace

extern "C" {

/* mlp [B][W], emb [B][n_emb][W] -> top [B][W + n_ins (n_ins - 1) / 2 + 1]; when top_grad is given
 * mlp / emb are overwritten with their gradients (as the reference's in-place layer does) */
int ref_interaction(size_t B, size_t n_emb, size_t W, float* mlp, float* emb, float* top,
                    const float* top_grad) {
  std::vector<float> m(mlp, mlp + B * W), e(emb, emb + B * n_emb * W), t;
  interaction_ref<float>(B, n_emb, W, m, e, t, top_grad);
  std::memcpy(top, t.data(), t.size() * sizeof(float));
  if (top_grad) {
    std::memcpy(mlp, m.data(), m.size() * sizeof(float));
    std::memcpy(emb, e.data(), e.size() * sizeof(float));
  }
  return 0;
}

/* DCN cross layers.  projection_dim == 0 (v1): kernels [L][w]; else (v2): kernels [2L] = U_l [w][p],
 * V_l [p][w] back to back per layer.  Outputs: out [B][w], in_grad [B][w], kernel_grads (same
 * shape as kernels), bias_grads [L][w].  out_grad may be NULL (forward only). */
int ref_cross(size_t B, size_t w, int L, size_t p, const float* x, const float* kernels,
              const float* biases, const float* out_grad, float* out, float* in_grad,
              float* kernel_grads, float* bias_grads) {
  CrossRef<float> c;
  c.batchsize_ = B;
  c.w_ = w;
  c.layers_ = L;
  c.projection_dim_ = p;
  c.h_input_.assign(x, x + B * w);
  c.h_input_grad_.assign(B * w, 0.f);
  const size_t ksz = p ? w * p : w;
  const int nk = p ? 2 * L : L;
  for (int i = 0; i < nk; i++) {
    c.h_kernels_.emplace_back(kernels + (size_t)i * ksz, kernels + (size_t)(i + 1) * ksz);
    c.h_kernel_grads_.emplace_back(ksz, 0.f);
  }
  for (int i = 0; i < L; i++) {
    c.h_biases_.emplace_back(biases + (size_t)i * w, biases + (size_t)(i + 1) * w);
    c.h_bias_grads_.emplace_back(w, 0.f);
    c.h_outputs_.emplace_back(B * w, 0.f);
    c.h_hiddens_.emplace_back(p ? B * w : B, 0.f);
    c.XUs.emplace_back(p ? B * p : 1, 0.f);
  }
  if (p) c.cpu_fprop_v2_();
  else c.cpu_fprop_();
  std::memcpy(out, c.h_outputs_.back().data(), B * w * sizeof(float));
  if (out_grad == nullptr) return 0;
  c.h_output_grad_.assign(out_grad, out_grad + B * w);
  if (p) c.cpu_bprop_v2_();
  else c.cpu_bprop_();
  std::memcpy(in_grad, c.h_input_grad_.data(), B * w * sizeof(float));
  for (int i = 0; i < nk; i++)
    std::memcpy(kernel_grads + (size_t)i * ksz, c.h_kernel_grads_[i].data(), ksz * sizeof(float));
  for (int i = 0; i < L; i++)
    std::memcpy(bias_grads + (size_t)i * w, c.h_bias_grads_[i].data(), w * sizeof(float));
  return 0;
}
}
