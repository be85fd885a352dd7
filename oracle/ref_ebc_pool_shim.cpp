// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's device code of the
// embedding_collection pooling / reduction operators -- R/HugeCTR/embedding/operators/
// generic_lookup.cuh whole (namespace embedding: the 4-wide vector type, the multi_to_one_* /
// one_to_multi_* kernels every EBC forward / backward operator launches, the MultiToOne / OneToOne
// descriptors and the host functions copy_multi_to_one / copy_one_to_multi that choose the kernel
// by vector size), cut out of the checkout by oracle/Makefile and executed by the host interpreter
// of tests/emu (32-lane warps).  The descriptors' address lambdas below describe plain
// back-to-back layouts (the operators' own lambdas, model_forward.cu / network_forward.cu, differ
// only in where vectors live, which the reference's CPU code already pins); what is exercised is
// the ARITHMETIC of the reference: fp32 accumulation in source order, the division by the Average
// factor, the rounding to the destination type -- for fp32 and fp16 sources / destinations.
#define REFSHIM_TRIVIAL_HALF
#include <common.hpp>  // oracle/ref_shims/common.hpp

#include <cassert>
#include <limits>
#include <vector>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"

#define DEVICE_INLINE __device__ __forceinline__
#define HOST_DEVICE_INLINE __host__ __device__ __forceinline__

namespace HugeCTR {
#include "_ref/gen/gpu_type_convert_func.inc"
namespace core23 {
struct KernelParams {  // (R/HugeCTR/include/core23/kernel_params.hpp: the three fields read here)
  int num_sms = 8;
  int max_thread_per_sm = 2048;
  int warp_size = 32;
};
}  // namespace core23
}  // namespace HugeCTR

#include "_ref/gen/ebc_generic_lookup.inc"

using namespace embedding;

namespace {
template <typename S, typename D>
void run_multi_to_one(int num_vec, const int* offsets, const int* factor, int vec_length,
                      const S* const* src, D* dst, int max_ev_size) {
  auto desc = make_MultiToOne<S, D>(
      num_vec, [=](int i) { return offsets[i]; }, [=](int i) { return factor[i]; },
      [=](int) { return vec_length; }, [=](int i) { return src[i]; },
      [=](int i) { return dst + (size_t)i * vec_length; });
  copy_multi_to_one(desc, max_ev_size, nullptr);
}
template <typename S, typename D>
void run_one_to_multi(int num_vec, const int* offsets, const int* factor, int vec_length,
                      const S* src, D* const* dst, int max_ev_size) {
  auto desc = make_MultiToOne<S, D>(
      num_vec, [=](int i) { return offsets[i]; }, [=](int i) { return factor[i]; },
      [=](int) { return vec_length; }, [=](int i) { return src + (size_t)i * vec_length; },
      [=](int j) { return dst[j]; });
  copy_one_to_multi(desc, max_ev_size, nullptr);
}
}  // namespace

extern "C" {
// dst[r][:] = round_D( float(src[i][:]) / factor[i] ) for r in [offsets[i], offsets[i+1]) --
// NetworkBackward: the gradient of an output vector goes to every shard of its lookup, divided by
// the bucket's key count for Average lookups (network_backward.cu:56-100)
void refebc_one_to_multi(int half, int num_vec, const int* offsets, const int* factor,
                         int vec_length, const void* src, void* const* dst, int max_ev_size) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  if (half)
    run_one_to_multi<__half, __half>(num_vec, offsets, factor, vec_length, (const __half*)src, (__half* const*)dst, max_ev_size);
  else
    run_one_to_multi<float, float>(num_vec, offsets, factor, vec_length, (const float*)src, (float* const*)dst, max_ev_size);
  hipemu::set_wave_width(64);
}

// dst[i][:] = round_D( (sum_{r in [offsets[i], offsets[i+1])} float(src[r][:])) / factor[i] )
// (factor <= 0: no division) -- ModelForward pools a bucket's rows this way (factor 1),
// NetworkForward sums the shards' partial vectors and divides Average lookups by the bucket's key
// count.  src_half / dst_half: binary16 vectors, else fp32.
void refebc_multi_to_one(int src_half, int dst_half, int num_vec, const int* offsets,
                         const int* factor, int vec_length, const void* const* src, void* dst,
                         int max_ev_size) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  if (!src_half && !dst_half)
    run_multi_to_one<float, float>(num_vec, offsets, factor, vec_length, (const float* const*)src, (float*)dst, max_ev_size);
  else if (!src_half && dst_half)
    run_multi_to_one<float, __half>(num_vec, offsets, factor, vec_length, (const float* const*)src, (__half*)dst, max_ev_size);
  else if (src_half && dst_half)
    run_multi_to_one<__half, __half>(num_vec, offsets, factor, vec_length, (const __half* const*)src, (__half*)dst, max_ev_size);
  else
    run_multi_to_one<__half, float>(num_vec, offsets, factor, vec_length, (const __half* const*)src, (float*)dst, max_ev_size);
  hipemu::set_wave_width(64);
}
}
