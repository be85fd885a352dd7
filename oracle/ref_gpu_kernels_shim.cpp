// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's own DEVICE code of the legacy sparse
// embedding, executed by the host interpreter of tests/emu (CUDA threads = fibers, host memory):
//   forward_sum / forward_mean   + their launch wrappers (fp32, and the paired-half "align2" forms)
//                                  R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu:22-243
//   do_forward_scale                R/HugeCTR/src/embeddings/forward_scale_functor.cu:23-101
//   backward_sum / backward_mean    R/HugeCTR/src/embeddings/backward_functor.cu:23-158
//   do_forward_reorder / do_backward_reorder (the layout change around the localized embedding's
//                                  all-to-all)  R/HugeCTR/src/embeddings/forward_reorder_functor.cu:22-120,
//                                  backward_reorder_functor.cu:22-123
//   select_value_and_rowoffset_by_slot_id_kernel / select_rowoffset + HashOp (filter_keys_per_gpu of the
//                                  localized / distributed embedding; the three library calls
//                                  around them follow localized_slot_sparse_embedding_hash.cu:
//                                  113-146 and distributed_slot_sparse_embedding_hash.cu:107-147)
//   store_slot_id_kernel           R/HugeCTR/src/embeddings/store_slot_id_functor.cu:22-50
//   EmbeddingOptimizer::update     R/HugeCTR/src/optimizers/sparse_optimizer.cu:170-612 (kernels),
//                                  :622-864 (the method: expansion, sort, run counting, optimizer)
// The blocks are cut out of the checkout by oracle/Makefile (sed by their first / last lines, the
// <<<>>> launches rewritten by ref_launch_rewrite.py) into _ref/gen/ (generated, never committed).
// What is written here is declarations only: the class shell whose member NAMES update() refers to
// (after R/HugeCTR/include/optimizer.hpp:284-340), a pointer-and-size Tensor2, and the C wrappers.
// cub's sort / scan are the contract-level stand-ins of ref_shims/cuda/cuda_device_extras.h.
// tests/test_ref_gpu_kernels_cpu.py compares oracle/hctr_oracle.c AND the HIP kernels' source
// (tests/emu build of hugectr_amd/csrc) with these on the same inputs.
#include <common.hpp>  // oracle/ref_shims/common.hpp: enums, OptParams, binary16 __half

#include <cmath>
#include <limits>
#include <memory>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cuda_device_extras.h"

#define HCTR_LIB_THROW(expr) \
  do { if ((expr) != cudaSuccess) throw std::runtime_error("cuda stand-in reported an error"); } while (0)

namespace HugeCTR {

template <typename T>
class Tensor2 {
  T* p_ = nullptr;
  size_t bytes_ = 0;

 public:
  Tensor2() = default;
  Tensor2(T* p, size_t bytes) : p_(p), bytes_(bytes) {}
  T* get_ptr() const { return p_; }
  size_t get_size_in_bytes() const { return bytes_; }
};

struct SparseEmbeddingHashParams {
  OptParams opt_params;
};

template <typename TypeEmbeddingComp>
struct OptimizerTensor {
  Tensor2<TypeEmbeddingComp> opt_z_tensors_, opt_n_tensors_, opt_m_tensors_, opt_v_tensors_;
  Tensor2<uint64_t> opt_prev_time_tensors_;
  Tensor2<TypeEmbeddingComp> opt_momentum_tensors_, opt_accm_tensors_;
};

template <typename TypeHashKey, typename TypeEmbeddingComp>
class EmbeddingOptimizer {
 public:
  Tensor2<void> temp_storage_sort_tensors_, temp_storage_scan_tensors_;
  Tensor2<TypeHashKey> sample_id_tensors_, sample_id_sort_tensors_;
  Tensor2<size_t> hash_value_index_sort_tensors_;
  Tensor2<uint32_t> new_hash_value_flag_tensors_, hash_value_flag_sumed_tensors_,
      hash_value_index_count_offset_tensors_, hash_value_index_count_counter_tensors_;
  SparseEmbeddingHashParams& param;
  OptimizerTensor<TypeEmbeddingComp> opt_tensors_;
  explicit EmbeddingOptimizer(SparseEmbeddingHashParams& p) : param(p) {}
  void update(size_t batch_size, size_t slot_num, size_t embedding_vec_size,
              size_t max_vocabulary_size_per_gpu, size_t nnz, const Tensor2<TypeHashKey>& row_offset,
              Tensor2<size_t>& hash_value_index, const Tensor2<TypeEmbeddingComp>& wgrad,
              Tensor2<float>& hash_table_value, size_t sm_count, cudaStream_t stream);
};

// ---- the reference's text from here ----------------------------------------------------------------
#include "_ref/gen/gpu_type_convert_func.inc"
#include "_ref/gen/gpu_forward.inc"
#include "_ref/gen/gpu_forward_scale.inc"
#include "_ref/gen/gpu_backward.inc"
#include "_ref/gen/gpu_forward_reorder.inc"
#include "_ref/gen/gpu_backward_reorder.inc"
#include "_ref/gen/gpu_filter_localized.inc"
#include "_ref/gen/gpu_filter_distributed.inc"
#include "_ref/gen/gpu_store_slot_id.inc"
#include "_ref/gen/gpu_opt_kernels.inc"
#include "_ref/gen/gpu_opt_update.inc"
// ---- to here -----------------------------------------------------------------------------------------

namespace {
template <typename K, typename E>
void run_forward(int combiner, size_t batch, size_t slots, size_t D, const K* ro, const size_t* vi,
                 const float* table, E* out) {
  if (combiner == 0)
    forward_sum(batch, slots, D, ro, vi, table, out, (cudaStream_t) nullptr);
  else
    forward_mean(batch, slots, D, ro, vi, table, out, (cudaStream_t) nullptr);
}
template <typename K, typename E>
void run_backward(int combiner, size_t batch, size_t slots, size_t D, const K* ro, const E* top,
                  E* wgrad) {
  if (combiner == 0)
    backward_sum(batch, slots, D, top, wgrad, (cudaStream_t) nullptr);
  else
    backward_mean(batch, slots, D, ro, top, wgrad, (cudaStream_t) nullptr);
}
template <typename K, typename E>
void run_update(const OptParams& op, size_t batch, size_t slots, size_t D, size_t vocab, size_t nnz,
                const K* ro, size_t* vi, const E* wgrad, float* table, E* s0, E* s1,
                uint64_t* prev_time) {
  SparseEmbeddingHashParams p;
  p.opt_params = op;
  EmbeddingOptimizer<K, E> o(p);
  const size_t n = nnz > 0 ? nnz : 1;
  std::vector<K> sid(n), sids(n);
  std::vector<size_t> vis(n);
  std::vector<uint32_t> flag(n), sumed(n), off(n + 1), counter(1);
  std::vector<char> tmp(64);
  o.temp_storage_sort_tensors_ = Tensor2<void>(tmp.data(), tmp.size());
  o.temp_storage_scan_tensors_ = Tensor2<void>(tmp.data(), tmp.size());
  o.sample_id_tensors_ = Tensor2<K>(sid.data(), n * sizeof(K));
  o.sample_id_sort_tensors_ = Tensor2<K>(sids.data(), n * sizeof(K));
  o.hash_value_index_sort_tensors_ = Tensor2<size_t>(vis.data(), n * sizeof(size_t));
  o.new_hash_value_flag_tensors_ = Tensor2<uint32_t>(flag.data(), n * 4);
  o.hash_value_flag_sumed_tensors_ = Tensor2<uint32_t>(sumed.data(), n * 4);
  o.hash_value_index_count_offset_tensors_ = Tensor2<uint32_t>(off.data(), (n + 1) * 4);
  o.hash_value_index_count_counter_tensors_ = Tensor2<uint32_t>(counter.data(), 4);
  const size_t sb = vocab * D * sizeof(E);
  switch (op.optimizer) {
    case Optimizer_t::Adam:
      o.opt_tensors_.opt_m_tensors_ = Tensor2<E>(s0, sb);
      o.opt_tensors_.opt_v_tensors_ = Tensor2<E>(s1, sb);
      o.opt_tensors_.opt_prev_time_tensors_ = Tensor2<uint64_t>(prev_time, vocab * D * 8);
      break;
    case Optimizer_t::MomentumSGD:
      o.opt_tensors_.opt_momentum_tensors_ = Tensor2<E>(s0, sb);
      break;
    default:
      o.opt_tensors_.opt_accm_tensors_ = Tensor2<E>(s0, sb);
  }
  Tensor2<K> t_ro(const_cast<K*>(ro), (batch * slots + 1) * sizeof(K));
  Tensor2<size_t> t_vi(vi, n * sizeof(size_t));
  Tensor2<E> t_wg(const_cast<E*>(wgrad), batch * slots * D * sizeof(E));
  Tensor2<float> t_tab(table, vocab * D * 4);
  o.update(batch, slots, D, vocab, nnz, t_ro, t_vi, t_wg, t_tab, /*sm_count=*/8, nullptr);
}
// filter_keys_per_gpu: the reference's kernels between the library calls its host code makes
// (cudaMemset of the flags, the kernel, DeviceSelect, InclusiveSum), in its order
template <typename K>
size_t run_filter(int distributed, size_t batch, size_t slots, size_t gid, size_t gnum, const K* ro,
                  const K* keys, size_t nnz, K* ro_out, K* keys_out) {
  std::vector<char> tmp(64);
  size_t tb = tmp.size(), picked = 0;
  const size_t num = batch * slots;
  if (!distributed) {
    const size_t spg = slots / gnum + (gid < slots % gnum ? 1 : 0);
    std::vector<char> flag(nnz + 1, 0);
    std::vector<K> sel(batch * spg + 1, 0);
    REFEMU_LAUNCH((localized_filter_keys_kernel::select_value_and_rowoffset_by_slot_id_kernel),
                  ((num - 1) / 256 + 1, 256), ro, num, flag.data(), sel.data(), spg, slots, gid,
                  gnum);
    cub::DeviceSelect::Flagged(tmp.data(), tb, keys, flag.data(), keys_out, &picked, nnz);
    cub::DeviceScan::InclusiveSum(tmp.data(), tb, sel.data(), ro_out, batch * spg + 1);
  } else {
    distributed_embedding_kernels::HashOp<K> op{gid, gnum};
    cub::DeviceSelect::If(tmp.data(), tb, keys, keys_out, &picked, nnz, op);
    std::vector<K> sel(num + 1, 0);
    REFEMU_LAUNCH((distributed_embedding_kernels::select_rowoffset), ((num - 1) / 512 + 1, 512), ro,
                  num, keys, sel.data(), gid, gnum);
    cub::DeviceScan::InclusiveSum(tmp.data(), tb, sel.data(), ro_out, num + 1);
  }
  return picked;
}
}  // namespace
}  // namespace HugeCTR

using namespace HugeCTR;

extern "C" {
// key_bytes 8 | 4, fp16 0 | 1; row_offset is key-typed (as the reference's), value_index size_t
void refgpu_forward(int key_bytes, int fp16, int combiner, size_t batch, size_t slots, size_t D,
                    const void* ro, const size_t* vi, const float* table, void* out) {
  hipemu::set_wave_width(64);
  hipemu::set_max_workers(0);
  if (key_bytes == 8 && !fp16) run_forward(combiner, batch, slots, D, (const long long*)ro, vi, table, (float*)out);
  if (key_bytes == 8 && fp16) run_forward(combiner, batch, slots, D, (const long long*)ro, vi, table, (__half*)out);
  if (key_bytes == 4 && !fp16) run_forward(combiner, batch, slots, D, (const unsigned*)ro, vi, table, (float*)out);
  if (key_bytes == 4 && fp16) run_forward(combiner, batch, slots, D, (const unsigned*)ro, vi, table, (__half*)out);
}

// distributed embedding, mean: the division after the reduce-scatter (in place)
void refgpu_forward_scale(int key_bytes, int fp16, size_t batch, size_t slots, size_t D,
                          const void* ro, void* feature) {
  if (key_bytes == 8 && !fp16) do_forward_scale(batch, slots, D, (const long long*)ro, (float*)feature, (cudaStream_t) nullptr);
  if (key_bytes == 8 && fp16) do_forward_scale(batch, slots, D, (const long long*)ro, (__half*)feature, (cudaStream_t) nullptr);
  if (key_bytes == 4 && !fp16) do_forward_scale(batch, slots, D, (const unsigned*)ro, (float*)feature, (cudaStream_t) nullptr);
  if (key_bytes == 4 && fp16) do_forward_scale(batch, slots, D, (const unsigned*)ro, (__half*)feature, (cudaStream_t) nullptr);
}

void refgpu_backward(int key_bytes, int fp16, int combiner, size_t batch, size_t slots, size_t D,
                     const void* ro, const void* top, void* wgrad) {
  if (key_bytes == 8 && !fp16) run_backward(combiner, batch, slots, D, (const long long*)ro, (const float*)top, (float*)wgrad);
  if (key_bytes == 8 && fp16) run_backward(combiner, batch, slots, D, (const long long*)ro, (const __half*)top, (__half*)wgrad);
  if (key_bytes == 4 && !fp16) run_backward(combiner, batch, slots, D, (const unsigned*)ro, (const float*)top, (float*)wgrad);
  if (key_bytes == 4 && fp16) run_backward(combiner, batch, slots, D, (const unsigned*)ro, (const __half*)top, (__half*)wgrad);
}

// forward: all-to-all receive buffer [peer][b][slot in peer][D] -> [b][slot][D]; backward: the inverse
void refgpu_reorder(int fp16, int backward, size_t bpg, size_t slots, size_t D, size_t gpus,
                    const void* in, void* out) {
  if (!backward) {
    if (fp16) do_forward_reorder(bpg, slots, D, gpus, (const __half*)in, (__half*)out, (cudaStream_t) nullptr);
    else do_forward_reorder(bpg, slots, D, gpus, (const float*)in, (float*)out, (cudaStream_t) nullptr);
  } else {
    if (fp16) do_backward_reorder(bpg, slots, D, gpus, (const __half*)in, (__half*)out, (cudaStream_t) nullptr);
    else do_backward_reorder(bpg, slots, D, gpus, (const float*)in, (float*)out, (cudaStream_t) nullptr);
  }
}

// keys of the full-batch CSR that GPU `gid` of `gnum` resolves: localized = its slots (slot % gnum ==
// gid), distributed = its keys (key % gnum == gid); returns the key count
size_t refgpu_filter_keys(int key_bytes, int distributed, size_t batch, size_t slots, size_t gid,
                          size_t gnum, const void* ro, const void* keys, size_t nnz, void* ro_out,
                          void* keys_out) {
  if (key_bytes == 8)
    return run_filter(distributed, batch, slots, gid, gnum, (const long long*)ro,
                      (const long long*)keys, nnz, (long long*)ro_out, (long long*)keys_out);
  return run_filter(distributed, batch, slots, gid, gnum, (const unsigned*)ro,
                    (const unsigned*)keys, nnz, (unsigned*)ro_out, (unsigned*)keys_out);
}

// slot id of every row met by this batch (the localized embedding's dump needs it)
void refgpu_store_slot_id(int key_bytes, size_t batch, int slots, int slots_per_gpu, int gnum,
                          int gid, const void* ro, const size_t* vi, size_t* slot_id) {
  const size_t n = batch * (size_t)slots_per_gpu;
  if (n == 0) return;
  if (key_bytes == 8)
    REFEMU_LAUNCH((store_slot_id_kernel), ((n - 1) / 64 + 1, 64), batch, slots, slots_per_gpu, gnum,
                  gid, (const long long*)ro, vi, slot_id);
  else
    REFEMU_LAUNCH((store_slot_id_kernel), ((n - 1) / 64 + 1, 64), batch, slots, slots_per_gpu, gnum,
                  gid, (const unsigned*)ro, vi, slot_id);
}

// one EmbeddingOptimizer::update; optimizer / update_type = the reference's enum values
// (common.hpp:82-94); `times` = adam.times AFTER the increment the embedding does before update.
// Optimizer state has the embedding's output type (fp32 or fp16), as in the reference.
int refgpu_update(int key_bytes, int fp16, int optimizer, int update_type, int atomic_sgd, float lr,
                  float scaler, float beta1, float beta2, float epsilon, float momentum_or_mu,
                  unsigned long long times, size_t batch, size_t slots, size_t D, size_t vocab,
                  size_t nnz, const void* ro, size_t* vi, const void* wgrad, float* table, void* s0,
                  void* s1, uint64_t* prev_time) {
  OptParams op;
  op.optimizer = static_cast<Optimizer_t>(optimizer);
  op.update_type = static_cast<Update_t>(update_type);
  op.lr = lr;
  op.scaler = scaler;
  op.hyperparams.adam.beta1 = beta1;
  op.hyperparams.adam.beta2 = beta2;
  op.hyperparams.adam.epsilon = epsilon;
  op.hyperparams.adam.times = times;
  op.hyperparams.adagrad.epsilon = epsilon;
  op.hyperparams.momentum.factor = momentum_or_mu;
  op.hyperparams.nesterov.mu = momentum_or_mu;
  op.hyperparams.sgd.atomic_update = atomic_sgd != 0;
  try {
    if (key_bytes == 8 && !fp16) run_update(op, batch, slots, D, vocab, nnz, (const long long*)ro, vi, (const float*)wgrad, table, (float*)s0, (float*)s1, prev_time);
    if (key_bytes == 8 && fp16) run_update(op, batch, slots, D, vocab, nnz, (const long long*)ro, vi, (const __half*)wgrad, table, (__half*)s0, (__half*)s1, prev_time);
    if (key_bytes == 4 && !fp16) run_update(op, batch, slots, D, vocab, nnz, (const unsigned*)ro, vi, (const float*)wgrad, table, (float*)s0, (float*)s1, prev_time);
    if (key_bytes == 4 && fp16) run_update(op, batch, slots, D, vocab, nnz, (const unsigned*)ro, vi, (const __half*)wgrad, table, (__half*)s0, (__half*)s1, prev_time);
  } catch (const std::exception& e) {
    fprintf(stderr, "refgpu_update: %s\n", e.what());
    return 1;
  }
  return 0;
}
}
