// TEST INFRASTRUCTURE ONLY: the REFERENCE's ModelForward and NetworkForward operators of a
// model-parallel embedding_collection group, address arithmetic included -- the descriptor block of
// ModelForward::sparse_forward (R/HugeCTR/embedding/operators/model_forward.cu:182-206: every
// (local lookup, global sample) bucket's rows, reached through the float** that ILookup::lookup
// filled, summed into the buffer of the GPU that owns the sample) and the descriptor blocks of
// network_forward_to_batch_major_output / network_forward_to_feature_major_output
// (R/HugeCTR/embedding/operators/network_forward.cu:272-321, :353-404: the five device lambdas that
// say which partial vectors of the received buffers make up an output vector, by what count an
// Average is divided, and where the result goes, with the copy_multi_to_one launch behind them),
// cut out of the checkout by oracle/Makefile and compiled inside a function that declares the
// variables the lambdas capture, under the names the reference gives them; the kernels they feed
// are generic_lookup.cuh's (oracle/_ref/gen/ebc_generic_lookup.inc, as in ref_ebc_pool_shim.cpp);
// all of it executed by the host interpreter of tests/emu (32-lane warps).
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
struct KernelParams {  // (R/HugeCTR/include/core23/kernel_params.hpp: the three fields read)
  int num_sms = 8;
  int max_thread_per_sm = 2048;
  int warp_size = 32;
};
}  // namespace core23
}  // namespace HugeCTR

#include "_ref/gen/ebc_generic_lookup.inc"

namespace embedding {
enum class Combiner : char { Sum, Average, Concat };  // (R/HugeCTR/embedding/common.hpp:129)

struct NetArgs {
  int batch_size_per_gpu, num_lookup, num_network_dst_lookup_ids, max_ev_size;
  const void* dp_num_keys_per_bucket;
  const int *network_ids, *network_gpu_ids, *network_offsets, *network_dst_lookup_ids;
  const int* const* network_ev_sizes;
  const int* const* network_ev_offsets;
  const void* const* network_comm_buffer;
  const int* dst_ev_start_indices;
  const char* dst_combiner;
  void* output_buffer;
};

template <typename offset_t, typename emb_t, typename dst_emb_t, bool kBatchMajor>
void run(const NetArgs& a) {
  const HugeCTR::core23::KernelParams kernel_params;
  cudaStream_t stream = nullptr;
  const int batch_size_per_gpu = a.batch_size_per_gpu;
  const int max_ev_size = a.max_ev_size;
  const int num_lookup = a.num_lookup;
  (void)num_lookup;
  const offset_t* dp_num_keys_per_bucket_ptr = (const offset_t*)a.dp_num_keys_per_bucket;
  const int* network_ids_ptr = a.network_ids;
  const int* network_gpu_ids_ptr = a.network_gpu_ids;
  const int* network_offsets_ptr = a.network_offsets;
  const int* network_dst_lookup_ids_ptr = a.network_dst_lookup_ids;
  const int** network_ev_sizes_ptr = (const int**)a.network_ev_sizes;
  const int** network_ev_offsets_ptr = (const int**)a.network_ev_offsets;
  const emb_t** network_comm_buffer_ptr = (const emb_t**)a.network_comm_buffer;
  const int* dst_ev_start_indices_ptr = a.dst_ev_start_indices;
  const char* dst_combiner_ptr = a.dst_combiner;
  dst_emb_t* output_buffer_ptr = (dst_emb_t*)a.output_buffer;
  int num_network_dst_lookup_ids = a.num_network_dst_lookup_ids;
  if constexpr (kBatchMajor) {
#include "_ref/gen/network_forward_batch_major.inc"
  } else {
#include "_ref/gen/network_forward_feature_major.inc"
  }
}

// NetworkBackward (network_backward.cu:55-98 feature-major top gradient, :132-174 batch-major): the
// gradient of an output vector goes to every shard of its lookup, divided by the bucket's key
// count for Average lookups
struct NetBwdArgs {
  int batch_size_per_gpu, num_network_dst_lookup_ids, max_ev_size;
  const void* dp_num_keys_per_bucket;
  const int *network_ids, *network_gpu_ids, *network_offsets, *network_dst_lookup_ids;
  int* const* network_ev_sizes;
  int* const* network_ev_offsets;
  const int* d_ev_size_offset;
  const void* top_grad;
  void* const* network_comm_buffer;
  const char* combiner;
};
template <typename offset_t, typename emb_t, typename dst_emb_t, bool kBatchMajor>
void run_bwd(const NetBwdArgs& a) {
  const HugeCTR::core23::KernelParams kernel_params;
  cudaStream_t stream = nullptr;
  const int batch_size_per_gpu = a.batch_size_per_gpu;
  const int max_ev_size = a.max_ev_size;
  const offset_t* dp_num_keys_per_bucket_ptr = (const offset_t*)a.dp_num_keys_per_bucket;
  const int* network_ids_ptr = a.network_ids;
  const int* network_gpu_ids_ptr = a.network_gpu_ids;
  const int* network_offsets_ptr = a.network_offsets;
  const int* network_dst_lookup_ids_ptr = a.network_dst_lookup_ids;
  int** network_ev_sizes_ptr = (int**)a.network_ev_sizes;
  int** network_ev_offsets_ptr = (int**)a.network_ev_offsets;
  const int* d_ev_size_offset_ptr = a.d_ev_size_offset;
  const emb_t* top_grad_ptr = (const emb_t*)a.top_grad;
  dst_emb_t** network_comm_buffer_ptr = (dst_emb_t**)a.network_comm_buffer;
  const char* combiner_ptr = a.combiner;
  int num_network_dst_lookup_ids = a.num_network_dst_lookup_ids;
  int num_lookup = a.num_network_dst_lookup_ids;  // (the batch-major block's name for the row width's index)
  (void)num_lookup;
  if constexpr (kBatchMajor) {
#include "_ref/gen/network_backward_batch_major.inc"
  } else {
#include "_ref/gen/network_backward_feature_major.inc"
  }
}

struct ModelArgs {
  int batch_size, batch_size_per_gpu, num_lookup, max_ev_size;
  const void* bucket_range;
  const int *id_to_ev_size, *id_to_ev_start_indices;
  const float* const* mp_ev;
  void* const* model_comm_buffer;
};
template <typename offset_t, typename emb_t>
void run_model(const ModelArgs& a) {
  // (the three names the launch statement of the cut block reads besides the captured pointers)
  struct Core {
    HugeCTR::core23::KernelParams get_kernel_param() const { return HugeCTR::core23::KernelParams(); }
  } core_obj, *core_ = &core_obj;
  struct {
    struct {
      int max_ev_size;
    } attr;
  } model_comm_buffer{{a.max_ev_size}};
  cudaStream_t stream = nullptr;
  const int batch_size = a.batch_size, batch_size_per_gpu = a.batch_size_per_gpu;
  const int num_lookup = a.num_lookup;
  const offset_t* bucket_range_ptr = (const offset_t*)a.bucket_range;
  const int* id_to_ev_size_ptr = a.id_to_ev_size;
  const int* id_to_ev_start_indices_ptr = a.id_to_ev_start_indices;
  const float** mp_ev_ptr = (const float**)a.mp_ev;
  emb_t** model_comm_buffer_ptr = (emb_t**)a.model_comm_buffer;
#include "_ref/gen/model_forward_sparse.inc"
}
}  // namespace embedding

extern "C" {
void refnet_backward(int batch_major, int half, int batch_size_per_gpu, int num_dst, int max_ev_size,
                     const long long* dp_num_keys_per_bucket, const int* network_ids,
                     const int* network_gpu_ids, const int* network_offsets,
                     const int* network_dst_lookup_ids, int* const* network_ev_sizes,
                     int* const* network_ev_offsets, const int* d_ev_size_offset, const void* top_grad,
                     void* const* network_comm_buffer, const char* combiner) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  embedding::NetBwdArgs a{batch_size_per_gpu, num_dst, max_ev_size, dp_num_keys_per_bucket, network_ids,
                          network_gpu_ids, network_offsets, network_dst_lookup_ids, network_ev_sizes,
                          network_ev_offsets, d_ev_size_offset, top_grad, network_comm_buffer, combiner};
  if (half) {
    if (batch_major) embedding::run_bwd<long long, __half, __half, true>(a);
    else embedding::run_bwd<long long, __half, __half, false>(a);
  } else {
    if (batch_major) embedding::run_bwd<long long, float, float, true>(a);
    else embedding::run_bwd<long long, float, float, false>(a);
  }
  hipemu::set_wave_width(64);
}

// bucket_range: int64 [num_lookup * batch_size + 1] over buckets (local lookup, global sample);
// mp_ev[j]: the fp32 row of key j; model_comm_buffer[g]: [local lookup][batch_size_per_gpu][ev] of
// binary16 (half != 0) or fp32 vectors for GPU g
void refmodel_forward(int half, int batch_size, int batch_size_per_gpu, int num_lookup, int max_ev_size,
                      const long long* bucket_range, const int* id_to_ev_size,
                      const int* id_to_ev_start_indices, const float* const* mp_ev,
                      void* const* model_comm_buffer) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  embedding::ModelArgs a{batch_size, batch_size_per_gpu, num_lookup, max_ev_size, bucket_range,
                         id_to_ev_size, id_to_ev_start_indices, mp_ev, model_comm_buffer};
  if (half) embedding::run_model<long long, __half>(a);
  else embedding::run_model<long long, float>(a);
  hipemu::set_wave_width(64);
}

// half != 0: binary16 buffers and output, else fp32.  dp_num_keys_per_bucket: int64
// [num_lookup][batch_size_per_gpu]
void refnet_forward(int batch_major, int half, int batch_size_per_gpu, int num_lookup, int num_dst,
                    int max_ev_size, const long long* dp_num_keys_per_bucket, const int* network_ids,
                    const int* network_gpu_ids, const int* network_offsets,
                    const int* network_dst_lookup_ids, const int* const* network_ev_sizes,
                    const int* const* network_ev_offsets, const void* const* network_comm_buffer,
                    const int* dst_ev_start_indices, const char* dst_combiner, void* output_buffer) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  embedding::NetArgs a{batch_size_per_gpu, num_lookup, num_dst, max_ev_size, dp_num_keys_per_bucket,
                       network_ids, network_gpu_ids, network_offsets, network_dst_lookup_ids,
                       network_ev_sizes, network_ev_offsets, network_comm_buffer,
                       dst_ev_start_indices, dst_combiner, output_buffer};
  if (half) {
    if (batch_major) embedding::run<long long, __half, __half, true>(a);
    else embedding::run<long long, __half, __half, false>(a);
  } else {
    if (batch_major) embedding::run<long long, float, float, true>(a);
    else embedding::run<long long, float, float, false>(a);
  }
  hipemu::set_wave_width(64);
}
}
