// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's device code of MultiCrossLayer
// (R/HugeCTR/src/layers/multi_cross_layer.cu:54-563: the kernels vector_fma4 / vector_fma4_align8 /
// vector_fma3_align8 / vector_mul_fma3_align (+ their paired-half forms), matrix_pair_mul_kernel,
// mm_1d, row_scaling_sum_kernel and the host functions that launch them) and the two functors of
// the v1 layer that compose them (MultiCrossForwardFunctor / MultiCrossBackwardFunctor,
// :582-600 / :698-732) -- cut out of the checkout by oracle/Makefile and executed by the host
// interpreter of tests/emu (32-lane warps).
//
// What the file calls but does not hold is supplied here by its documented contract, no reference
// code: cuBLAS gemm (column-major, C = alpha op(A) op(B) + beta C; fp32 accumulation in index order,
// one rounding for binary16 -- cuBLAS does not specify its order, so comparisons of the GEMV
// results carry a tolerance), and MLCommon::LinAlg::matrixVectorOp / binaryOp / reduce (cuML
// primitives that the checkout includes from an un-vendored path; their index rule is the one the
// reference's own test kernel states, R/test/utest/prims/matrix_vector_op.h:24-41; reduce adds the
// rows in index order).  The v2 layer's GEMMs are cublasLt calls; its own kernels are the
// elementwise ones, driven here one by one (refcross_v2_*).
#define REFSHIM_TRIVIAL_HALF
#define REFSHIM_HALF_ARITH
#include <common.hpp>  // oracle/ref_shims/common.hpp

#include <array>
#include <cassert>
#include <type_traits>
#include <vector>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"

typedef __half half;
typedef __half2 half2;
#define __restrict__
#define __inline__ inline
#define WARP_SIZE 32

namespace cuda {
namespace std {
using ::std::array;
}
}  // namespace cuda

// the paired-half fused multiply-add and the binary16 add (one rounding each, cuda_fp16.h)
static inline __half2 __hfma2(__half2 a, __half2 b, __half2 c) {
  __half2 r;
  r.x = __float2half((float)((double)__half2float(a.x) * (double)__half2float(b.x) + (double)__half2float(c.x)));
  r.y = __float2half((float)((double)__half2float(a.y) * (double)__half2float(b.y) + (double)__half2float(c.y)));
  return r;
}
static inline __half& operator+=(__half& a, const __half& b) {  // (cuda_fp16.hpp: a = a + b, one rounding)
  a = a + b;
  return a;
}
static inline __half __hadd(__half a, __half b) { return __float2half(__half2float(a) + __half2float(b)); }
template <typename T>
static inline T __shfl_xor_sync(unsigned, T v, int mask, int width = 32, int site = __builtin_LINE()) {
  uint64_t b = 0;
  static_assert(sizeof(T) <= 8, "shuffles move at most 64 bits");
  memcpy(&b, &v, sizeof(T));
  b = hipemu::collective(hipemu::OP_SHFL_XOR, b, mask, width, site);
  T r;
  memcpy(&r, &b, sizeof(T));
  return r;
}

// ---- cuBLAS by its contract (column-major) -----------------------------------------------------
typedef struct refemu_cublas* cublasHandle_t;
enum cublasOperation_t { CUBLAS_OP_N = 0, CUBLAS_OP_T = 1 };
#define CUBLAS_CHECK(x) (x)
static inline int cublasSetStream(cublasHandle_t, cudaStream_t) { return 0; }
template <typename T>
static int refemu_gemm(cublasOperation_t ta, cublasOperation_t tb, int m, int n, int k, const T* alpha,
                       const T* A, int lda, const T* B, int ldb, const T* beta, T* C, int ldc) {
  for (int j = 0; j < n; j++)
    for (int i = 0; i < m; i++) {
      float acc = 0.f;
      for (int p = 0; p < k; p++) {
        const float a = (float)(ta == CUBLAS_OP_N ? A[i + (size_t)p * lda] : A[p + (size_t)i * lda]);
        const float b = (float)(tb == CUBLAS_OP_N ? B[p + (size_t)j * ldb] : B[j + (size_t)p * ldb]);
        acc += a * b;
      }
      const float prev = (float)*beta == 0.f ? 0.f : (float)*beta * (float)C[i + (size_t)j * ldc];
      C[i + (size_t)j * ldc] = (T)((float)*alpha * acc + prev);
    }
  return 0;
}
static inline int cublasSgemm(cublasHandle_t, cublasOperation_t ta, cublasOperation_t tb, int m, int n,
                              int k, const float* alpha, const float* A, int lda, const float* B,
                              int ldb, const float* beta, float* C, int ldc) {
  return refemu_gemm<float>(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}
static inline int cublasHgemm(cublasHandle_t, cublasOperation_t ta, cublasOperation_t tb, int m, int n,
                              int k, const __half* alpha, const __half* A, int lda, const __half* B,
                              int ldb, const __half* beta, __half* C, int ldc) {
  return refemu_gemm<__half>(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc);
}

// ---- the cuML primitives by their contract ------------------------------------------------------
namespace MLCommon {
namespace LinAlg {
template <typename T, typename Op>
void matrixVectorOp(T* out, const T* mat, const T* vec, int D, int N, bool rowMajor,
                    bool bcastAlongRows, Op op, cudaStream_t) {
  const int len = N * D;
  for (int idx = 0; idx < len; idx++) {
    int col;
    if (rowMajor && bcastAlongRows) col = idx % D;
    else if (!rowMajor && !bcastAlongRows) col = idx % N;
    else if (rowMajor && !bcastAlongRows) col = idx / D;
    else col = idx / N;
    out[idx] = op(mat[idx], vec[col]);
  }
}
template <typename T, typename Op>
void binaryOp(T* out, const T* a, const T* b, int len, Op op, cudaStream_t) {
  for (int i = 0; i < len; i++) out[i] = op(a[i], b[i]);
}
// the one shape the file uses: (h, w, init, rowMajor = false, alongRows = true) over a [h][w] array
// = the sum of the h rows, per column
template <typename T, typename Op>
void reduce(T* out, const T* mat, int h, int w, T init, bool rowMajor, bool alongRows, cudaStream_t,
            bool inplace, Op main_op) {
  assert(!rowMajor && alongRows && !inplace);
  for (int c = 0; c < w; c++) {
    T acc = init;
    for (int r = 0; r < h; r++) acc = acc + main_op(mat[(size_t)r * w + c], r);
    out[c] = acc;
  }
}
}  // namespace LinAlg
}  // namespace MLCommon

// ---- core23::Tensor as the file uses it: a typed pointer with a two-dimensional shape ------------
namespace HugeCTR {
namespace core23 {
struct Shape {
  int64_t d[2];
  int dims() const { return 2; }
  int64_t size(int i) const { return d[i]; }
  int64_t operator[](int i) const { return d[i]; }
};
struct Tensor {
  void* p = nullptr;
  Shape s{{0, 0}};
  size_t elem = 4;
  Tensor() = default;
  Tensor(void* p_, int64_t h, int64_t w, size_t elem_) : p(p_), s{{h, w}}, elem(elem_) {}
  template <typename T>
  T* data() const { return (T*)p; }
  void* data() const { return p; }
  const Shape& shape() const { return s; }
  size_t num_bytes() const { return (size_t)(s.d[0] * s.d[1]) * elem; }
};
}  // namespace core23

// (declarations of R/HugeCTR/include/layers/multi_cross_layer.hpp:25-37, 87-101)
template <typename T>
struct MultiCrossForwardFunctor {
  void operator()(cudaStream_t stream, cublasHandle_t cublas_handle, const core23::Tensor& input_tensor,
                  const std::vector<core23::Tensor>& kernel_tensors,
                  const std::vector<core23::Tensor>& bias_tensors,
                  std::vector<core23::Tensor>& layer_output_tensors,
                  std::vector<core23::Tensor>& layer_hidden_tensors, int num_layers) const;
};
template <typename T>
struct MultiCrossBackwardFunctor {
  void operator()(cudaStream_t stream, const core23::Tensor& input_tensor,
                  const std::vector<core23::Tensor>& kernel_tensors,
                  const std::vector<core23::Tensor>& layer_output_tensors,
                  const std::vector<core23::Tensor>& layer_hidden_tensors,
                  const core23::Tensor& grad_tensor, core23::Tensor& output_tensor,
                  std::vector<core23::Tensor>& kernel_output_tensors,
                  std::vector<core23::Tensor>& bias_output_tensors, core23::Tensor& tmp_vec_tensor,
                  core23::Tensor tmp_mat_tensors[], int num_layers) const;
};

#include "_ref/gen/utils_typefunc.inc"    // TypeFunc<T>
#include "_ref/gen/utils_warp_reduce.inc" // warpReduceSum
#include "_ref/gen/cross_kernels.inc"     // :54-563
template <typename T>
#include "_ref/gen/cross_v1_fwd.inc"      // MultiCrossForwardFunctor<T>::operator()
template <typename T>
#include "_ref/gen/cross_v1_bwd.inc"      // MultiCrossBackwardFunctor<T>::operator()
}  // namespace HugeCTR

using namespace HugeCTR;
namespace {
struct Warp32 {
  Warp32() {
    hipemu::set_wave_width(32);
    hipemu::set_max_workers(0);
  }
  ~Warp32() { hipemu::set_wave_width(64); }
};
template <typename T>
core23::Tensor tn(const T* p, int64_t h, int64_t w) {
  return core23::Tensor((void*)p, h, w, sizeof(T));
}
}  // namespace

extern "C" {
// v1 forward, fp32: x0 [B][w], kernels / biases [L][w] -> outputs [L][B][w], hiddens [L][B]
void refcross_v1_fwd(int B, int w, int L, const float* x0, const float* kernels, const float* biases,
                     float* outputs, float* hiddens) {
  Warp32 g;
  std::vector<core23::Tensor> ks, bs, outs, hid;
  for (int l = 0; l < L; l++) {
    ks.push_back(tn(kernels + (size_t)l * w, 1, w));
    bs.push_back(tn(biases + (size_t)l * w, 1, w));
    outs.push_back(tn(outputs + (size_t)l * B * w, B, w));
    hid.push_back(tn(hiddens + (size_t)l * B, B, 1));
  }
  MultiCrossForwardFunctor<float>()(nullptr, nullptr, tn(x0, B, w), ks, bs, outs, hid, L);
}
// v1 backward, fp32 (kernel / bias gradients are ACCUMULATED by row_scaling_sum_kernel's `+=`: the
// caller hands them in zeroed, as the layer's wgrad buffers are)
void refcross_v1_bwd(int B, int w, int L, const float* x0, const float* kernels, const float* outputs,
                     const float* hiddens, const float* out_grad, float* in_grad, float* kernel_grads,
                     float* bias_grads) {
  Warp32 g;
  std::vector<core23::Tensor> ks, outs, hid, dk, db;
  for (int l = 0; l < L; l++) {
    ks.push_back(tn(kernels + (size_t)l * w, 1, w));
    outs.push_back(tn(outputs + (size_t)l * B * w, B, w));
    hid.push_back(tn(hiddens + (size_t)l * B, B, 1));
    dk.push_back(tn(kernel_grads + (size_t)l * w, 1, w));
    db.push_back(tn(bias_grads + (size_t)l * w, 1, w));
  }
  std::vector<float> t0((size_t)B * w), t1((size_t)B * w), t2((size_t)B * w), tv(B);
  core23::Tensor tmp[3] = {tn(t0.data(), B, w), tn(t1.data(), B, w), tn(t2.data(), B, w)};
  core23::Tensor tvec = tn(tv.data(), B, 1), og = tn(in_grad, B, w);
  MultiCrossBackwardFunctor<float>()(nullptr, tn(x0, B, w), ks, outs, hid, tn(out_grad, B, w), og, dk,
                                     db, tvec, tmp, L);
}
// v2: x_{l+1} = x0 .* h + x_l with h = x_l U V + b from the GEMM (fused_matrix_elementwise_dot_add,
// :426-464).  half != 0: binary16 arrays (len % 8 == 0 takes the paired-half kernels; out == xl
// the in-place fma3 form)
void refcross_v2_dot_add(int B, int w, int half, void* out, const void* h, const void* x0, const void* xl) {
  Warp32 g;
  if (half) {
    core23::Tensor o = tn((__half*)out, B, w);
    fused_matrix_elementwise_dot_add<__half>(o, tn((const __half*)h, B, w), tn((const __half*)x0, B, w),
                                             tn((const __half*)xl, B, w), nullptr);
  } else {
    core23::Tensor o = tn((float*)out, B, w);
    fused_matrix_elementwise_dot_add<float>(o, tn((const float*)h, B, w), tn((const float*)x0, B, w),
                                            tn((const float*)xl, B, w), nullptr);
  }
}
// v2 backward's elementwise step (fused_mul_fma3, :391-424): S0 = dY .* X0, dX += dY .* H
void refcross_v2_mul_fma3(int B, int w, int half, void* s0, void* dx_acc, const void* dy, const void* x0,
                          const void* hmat) {
  Warp32 g;
  if (half) {
    core23::Tensor y0 = tn((__half*)s0, B, w), y1 = tn((__half*)dx_acc, B, w);
    fused_mul_fma3<__half>(y0, y1, tn((const __half*)dy, B, w), tn((const __half*)x0, B, w),
                           tn((const __half*)hmat, B, w), nullptr);
  } else {
    core23::Tensor y0 = tn((float*)s0, B, w), y1 = tn((float*)dx_acc, B, w);
    fused_mul_fma3<float>(y0, y1, tn((const float*)dy, B, w), tn((const float*)x0, B, w),
                          tn((const float*)hmat, B, w), nullptr);
  }
}
}
