// TEST INFRASTRUCTURE ONLY: the REFERENCE's device code of LocalReduce's index calculation -- the
// last component of the embedding_collection backward whose oracle was pinned against CPU code
// only -- R/HugeCTR/embedding/operators/index_calculation.cu: replicate_bucket_range_kernel
// (:337-375: every key's source bucket and table id), cal_table_range_kernel + LessThan (:397-416:
// where a table's keys begin in the partitioned list), get_keys_flag / get_unique_key (:647-684:
// first occurrences, unique keys, the key -> unique-key map), with bs_upper_bound_sub_one
// (R/HugeCTR/embedding/view.hpp:25-38) and CUDA_1D_KERNEL_LOOP (R/HugeCTR/include/utils.cuh:23-25),
// all cut out of the checkout by oracle/Makefile into oracle/_ref/gen/ and executed by the host
// interpreter of tests/emu.  The launches below restate LocalReduceIndexCalculation::
// cal_for_sparse_input's sequence (:877-904: partition_by_table_id -> intra_partition_sort ->
// unique_keys) for the case num_table == num_lookup with the launch geometry of the call sites
// (grid = num_sms * max_thread_per_block / 256, :385-387; dynamic shared memory (256 + 1) words,
// :388); cub's device-wide primitives act by their documented contracts
// (ref_shims/cuda/cuda_device_extras.h; the segmented sort = a stable sort of every table's range,
// SegmentedSortDevice's pre-2.2 branch :604-611).
#include <common.hpp>  // oracle/ref_shims/common.hpp

#include <cassert>
#include <cstdint>
#include <limits>
#include <numeric>
#include <vector>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"

#define DEVICE_INLINE __device__ __forceinline__
#define HOST_DEVICE_INLINE __host__ __device__ __forceinline__

#include "_ref/gen/cuda_1d_kernel_loop.inc"

namespace embedding {
#include "_ref/gen/bs_upper_bound_sub_one.inc"
namespace {
#include "_ref/gen/index_calculation_kernels.inc"
}  // namespace
}  // namespace embedding

using namespace embedding;

namespace {

template <typename key_t, typename offset_t>
void run(int batch_size, int num_lookup, const key_t* keys, const offset_t* bucket_range,
         const int* sorted_table_ids, const int* table_id_to_ev_size, key_t* sorted_keys,
         uint32_t* sorted_src_ids, int* table_ids, key_t* unique_keys, int* unique_table_ids,
         uint32_t* dst_ids, uint64_t* num_unique) {
  cudaStream_t stream = nullptr;
  const int block_size = 256;
  const int grid_size = 8 * 1024 / block_size;  // (num_sms * max_thread_per_block / block_size)
  const size_t max_keys = (size_t)bucket_range[(size_t)num_lookup * batch_size];
  std::vector<int> sorted_lookup_ids(num_lookup);
  std::iota(sorted_lookup_ids.begin(), sorted_lookup_ids.end(), 0);
  // ---- partition_by_table_id (num_table == num_lookup): keys as they lie, replicate_bucket_range
  std::vector<uint32_t> src_ids(max_keys + 1);
  std::vector<int> ev_sizes(max_keys + 1);
  uint64_t num_key = 0;
  REFEMU_LAUNCH((replicate_bucket_range_kernel<offset_t>),
                (grid_size, block_size, sizeof(uint32_t) * (block_size + 1), stream), bucket_range,
                sorted_lookup_ids.data(), sorted_table_ids, table_id_to_ev_size, num_lookup,
                batch_size, src_ids.data(), table_ids, ev_sizes.data(), &num_key);
  assert(num_key == max_keys);
  // ---- intra_partition_sort = SegmentedSortDevice::operator() ----------------------------------
  std::vector<int> temp_lookup_range(num_lookup + 1), partitioned_table_range(num_lookup + 2);
  REFEMU_LAUNCH((cal_table_range_kernel<offset_t>), (grid_size, block_size, 0, stream),
                bucket_range, sorted_table_ids, num_lookup, temp_lookup_range.data(), batch_size);
  int num_selected = 0;
  {
    LessThan select_op(std::numeric_limits<int>::max());
    char tmp[16];
    size_t tb = sizeof(tmp);
    cub::DeviceSelect::If(tmp, tb, temp_lookup_range.data(), partitioned_table_range.data(),
                          &num_selected, temp_lookup_range.size(), select_op, stream);
  }
  {
    char tmp[16];
    size_t tb = sizeof(tmp);
    cub::DeviceSegmentedRadixSort::SortPairs(tmp, tb, keys, sorted_keys, src_ids.data(),
                                             sorted_src_ids, num_key, num_selected - 1,
                                             partitioned_table_range.data(),
                                             partitioned_table_range.data() + 1, 0,
                                             (int)sizeof(key_t) * 8, stream);
  }
  // ---- unique_keys = SegmentdUnique::operator() -------------------------------------------------
  std::vector<uint32_t> key_flag(max_keys + 1);
  REFEMU_LAUNCH((get_keys_flag<key_t>), (grid_size, block_size, 0, stream), sorted_keys, table_ids,
                &num_key, key_flag.data());
  {
    char tmp[16];
    size_t tb = sizeof(tmp);
    cub::DeviceScan::InclusiveSum(tmp, tb, key_flag.data(), key_flag.data(), (size_t)num_key,
                                  stream);
  }
  REFEMU_LAUNCH((get_unique_key<key_t>), (grid_size, block_size, 0, stream), sorted_keys, table_ids,
                key_flag.data(), &num_key, unique_keys, unique_table_ids, num_unique, dst_ids);
}

}  // namespace

extern "C" {

// keys [nnz] int64, bucket_range [num_lookup * batch + 1] (offset_is_u32: uint32, else int64),
// feature-major buckets (bucket = lookup * batch + sample), lookup l reads table sorted_table_ids[l]
// (ascending).  Outputs sized nnz: sorted_keys, sorted_src_ids (the source BUCKET of every sorted
// key), table_ids (of every position of the partitioned list), unique_keys, unique_table_ids,
// dst_ids (sorted position -> unique key); *num_unique.
int refidx_local_reduce_indices(int batch_size, int num_lookup, int offset_is_u32,
                                const long long* keys, const void* bucket_range,
                                const int* sorted_table_ids, const int* table_id_to_ev_size,
                                long long* sorted_keys, uint32_t* sorted_src_ids, int* table_ids,
                                long long* unique_keys, int* unique_table_ids, uint32_t* dst_ids,
                                uint64_t* num_unique) {
  hipemu::set_wave_width(32);
  if (offset_is_u32)
    run<long long, uint32_t>(batch_size, num_lookup, keys, (const uint32_t*)bucket_range,
                             sorted_table_ids, table_id_to_ev_size, sorted_keys, sorted_src_ids,
                             table_ids, unique_keys, unique_table_ids, dst_ids, num_unique);
  else
    run<long long, long long>(batch_size, num_lookup, keys, (const long long*)bucket_range,
                              sorted_table_ids, table_id_to_ev_size, sorted_keys, sorted_src_ids,
                              table_ids, unique_keys, unique_table_ids, dst_ids, num_unique);
  hipemu::set_wave_width(64);
  return 0;
}

}  // extern "C"
