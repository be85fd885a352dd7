// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's device code of the static
// embedding_collection table, embedding::RaggedStaticEmbeddingTable
// (R/HugeCTR/embedding_storage/ragged_static_embedding.cu:29-355: the lookup kernel, the key -> row
// functor, the SGD / AdaGrad / Ftrl optimizer functors and update_kernel / update4_kernel), with the
// 4-wide vector type those functors load and store through (R/HugeCTR/embedding/operators/
// generic_lookup.cuh:29-291), the binary search of R/HugeCTR/embedding/view.hpp:25-39 and
// keys_to_indices_kernel (R/HugeCTR/embedding/operators/keys_to_indices.cu:23-43), cut out
// of the checkout into _ref/gen/ by oracle/Makefile and executed by the host interpreter of
// tests/emu (32-lane warps: the update kernels hand a key's row from lane to lane with
// __shfl_sync).  The launch lines below follow RaggedStaticEmbeddingTable::lookup / update
// (:553-700: block 256; the vectorized kernel when every vector size divides by 4).
#define REFSHIM_TRIVIAL_HALF
#include <common.hpp>  // oracle/ref_shims/common.hpp: enums, binary16 __half

#include <cassert>
#include <cfloat>
#include <cmath>
#include <limits>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"

using std::abs;
using std::signbit;
#define DEVICE_INLINE __device__ __forceinline__
#define HOST_DEVICE_INLINE __host__ __device__ __forceinline__

namespace HugeCTR {
#include "_ref/gen/gpu_type_convert_func.inc"
}
#include "_ref/gen/static_loop_macro.inc"

namespace embedding {
#include "_ref/gen/static_view_bsearch.inc"
#include "_ref/gen/static_vec4.inc"
#include "_ref/gen/static_table_kernels.inc"
#include "_ref/gen/static_keys_to_indices.inc"

namespace {
template <typename Opt>
void run_update(bool vec4, const long long* keys, const size_t* n, const int* table_ids,
                const float* wgrad, const uint32_t* ev_start, RaggedKeyToIndicesFunc<long long, uint64_t> f,
                float* table, Opt opt, float lr, float scaler) {
  const int grid = 6, block = 256;
  if (vec4)
    REFEMU_LAUNCH((update4_kernel<long long, uint64_t, float, Opt, decltype(f)>), (grid, block), keys,
                  n, table_ids, wgrad, ev_start, f, table, opt, lr, scaler);
  else
    REFEMU_LAUNCH((update_kernel<long long, uint64_t, float, Opt, decltype(f)>), (grid, block), keys,
                  n, table_ids, wgrad, ev_start, f, table, opt, lr, scaler);
}
}  // namespace
}  // namespace embedding

using namespace embedding;

extern "C" {
// tables of one GPU back to back in `emb_table`: table t holds keys [0, rows_t), its first row at
// element ev_offset[t], row offset table key_offset[t] (cumulative rows), vector size ev_size[t]
// optimizer: 0 SGD, 1 AdaGrad (state0 = accumulator), 2 Ftrl (state0 = n, state1 = z)
void refstatic_update(int optimizer, size_t num_keys, const long long* keys, const int* key_table,
                      const float* wgrad, const uint32_t* wgrad_start, int num_tables,
                      int* table_ids, int* ev_size, uint64_t* key_offset, uint64_t* ev_offset,
                      float* emb_table, float* state0, float* state1, float lr, float scaler,
                      float epsilon, float lambda1, float lambda2, float beta) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  bool vec4 = true;
  for (int t = 0; t < num_tables; t++) vec4 = vec4 && ev_size[t] % 4 == 0;
  RaggedKeyToIndicesFunc<long long, uint64_t> f{table_ids, ev_size, num_tables, key_offset, ev_offset};
  if (optimizer == 0) {
    run_update(vec4, keys, &num_keys, key_table, wgrad, wgrad_start, f, emb_table,
               SGDOptimizer<float>{}, lr, scaler);
  } else if (optimizer == 1) {
    run_update(vec4, keys, &num_keys, key_table, wgrad, wgrad_start, f, emb_table,
               AdaGradOptimizer<float, float>{state0, epsilon}, lr, scaler);
  } else {
    run_update(vec4, keys, &num_keys, key_table, wgrad, wgrad_start, f, emb_table,
               FtrlOptimizer<float, float>{state1, state0, beta, lambda1, lambda2}, lr, scaler);
  }
  hipemu::set_wave_width(64);
}

// KeysToIndicesConverter::convert: keys of `num_lookups` lookups back to back (lookup_offset), in
// place: index = first row of the lookup's table on this GPU + key / num_shards[table]
void refstatic_keys_to_indices(long long* keys, size_t num_keys, const uint64_t* lookup_offset,
                               int num_lookups, const int* table_of_lookup, const int* local_tables,
                               int num_local_tables, const uint64_t* table_row_offset,
                               const int* num_shards) {
  hipemu::set_wave_width(32);
  if (num_keys == 0) return;
  REFEMU_LAUNCH((keys_to_indices_kernel), ((num_keys - 1) / 256 + 1, 256), keys, num_keys,
                lookup_offset, num_lookups, table_of_lookup, local_tables, num_local_tables,
                table_row_offset, num_shards);
  hipemu::set_wave_width(64);
}

// the address of every key's vector (keys grouped by id space: id_space_offset[num_spaces + 1])
void refstatic_lookup(size_t num_keys, const long long* keys, const uint64_t* id_space_offset,
                      size_t num_offsets, const int* id_space_list, int num_tables, const int* table_ids,
                      const int* ev_size, const uint64_t* key_offset, const uint64_t* ev_offset,
                      float* emb_table, float** emb_vec) {
  hipemu::set_wave_width(32);
  if (num_keys == 0) return;
  REFEMU_LAUNCH((ragged_static_embedding_table_lookup_kernel), ((num_keys - 1) / 256 + 1, 256), keys,
                num_keys, id_space_offset, num_offsets, id_space_list, table_ids,
                (size_t)num_tables, key_offset, emb_table, ev_offset, ev_size, emb_vec);
  hipemu::set_wave_width(64);
}
}
