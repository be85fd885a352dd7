/* TEST INFRASTRUCTURE ONLY -- stand-in for the reference's core23/logger.hpp (and, through it, for
 * the few core23 / embedding declarations its CPU mirror of the dynamic embedding table needs), so
 * that R/HugeCTR/embedding_storage/dynamic_embedding_cpu.hpp and optimizers.hpp compile from where
 * they lie with g++ into oracle/_ref/libref_det.so (oracle/Makefile `ref`).  The real headers pull
 * in CUDA.  Declarations only: the interface the class overrides
 * (R/HugeCTR/embedding/embedding_table.hpp:22-33, R/HugeCTR/embedding_storage/
 * embedding_table.hpp:25-78), the two parameter structs it reads (embedding_storage/common.hpp:
 * 76-94, embedding/common.hpp:171-226: only the fields used), a host-memory Tensor with the three
 * members the class calls, and cudaMemcpy as memcpy.  Every optimizer formula and all table logic
 * in the library are the reference's. */
#pragma once
#include <cfloat>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <common.hpp>  // oracle/ref_shims/common.hpp: HugeCTR::OptParams, Optimizer_t, Error_t

#define HCTR_CHECK(cond) \
  do { if (!(cond)) throw std::runtime_error("check failed: " #cond); } while (0)
#define HCTR_LIB_THROW(expr) \
  do { if ((expr) != 0) throw std::runtime_error("call failed: " #expr); } while (0)

enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
static inline int cudaMemcpy(void* dst, const void* src, size_t bytes, cudaMemcpyKind) {
  std::memcpy(dst, src, bytes);
  return 0;
}

namespace core23 {
class Tensor {  // a view of host memory
 public:
  Tensor() = default;
  Tensor(void* p, size_t n) : p_(p), n_(n) {}
  size_t num_elements() const { return n_; }
  template <typename T>
  T* data() const { return static_cast<T*>(p_); }

 private:
  void* p_ = nullptr;
  size_t n_ = 0;
};
template <typename T>
void copy_sync(std::vector<T>& dst, const Tensor& src) {
  if (!dst.empty()) std::memcpy(dst.data(), src.data<T>(), dst.size() * sizeof(T));
}
}  // namespace core23

namespace embedding {
struct EmbeddingTableParam {
  int table_id;
  int64_t max_vocabulary_size;
  int ev_size;
  HugeCTR::OptParams opt_param;
};
struct GroupedTableParam {
  std::vector<int> table_ids;
};
struct EmbeddingCollectionParam {
  std::vector<GroupedTableParam> grouped_table_params;
};

class ILookup {
 public:
  virtual ~ILookup() = default;
  virtual void lookup(const core23::Tensor& keys, size_t num_keys,
                      const core23::Tensor& num_keys_per_table_offset, size_t num_table_offset,
                      const core23::Tensor& table_id_list, core23::Tensor& embedding_vec) = 0;
};

class IGroupedEmbeddingTable : public ILookup {
 public:
  virtual void update(const core23::Tensor& unique_keys, const core23::Tensor& num_unique_keys,
                      const core23::Tensor& table_ids, const core23::Tensor& ev_start_indices,
                      const core23::Tensor& wgrad) = 0;
  virtual void assign(const core23::Tensor& unique_key, size_t num_unique_key,
                      const core23::Tensor& num_unique_key_per_table_offset,
                      size_t num_table_offset, const core23::Tensor& table_id_list,
                      core23::Tensor& embeding_vector,
                      const core23::Tensor& embedding_vector_offset) = 0;
  virtual void load(core23::Tensor& keys, core23::Tensor& id_space_offset,
                    core23::Tensor& embedding_table, core23::Tensor& ev_size_list,
                    core23::Tensor& id_space) = 0;
  virtual void dump(core23::Tensor* keys, core23::Tensor* id_space_offset,
                    core23::Tensor* embedding_table, core23::Tensor* ev_size_list,
                    core23::Tensor* id_space) = 0;
  virtual void dump_by_id(core23::Tensor* h_keys_tensor, core23::Tensor* h_embedding_table,
                          int table_id) = 0;
  virtual void load_by_id(core23::Tensor* h_keys_tensor, core23::Tensor* h_embedding_table,
                          int table_id) = 0;
  virtual size_t size() const = 0;
  virtual size_t capacity() const = 0;
  virtual size_t key_num() const = 0;
  virtual std::vector<size_t> size_per_table() const = 0;
  virtual std::vector<size_t> capacity_per_table() const = 0;
  virtual std::vector<size_t> key_num_per_table() const = 0;
  virtual std::vector<int> table_ids() const = 0;
  virtual std::vector<int> table_evsize() const = 0;
  virtual void clear() = 0;
  virtual void set_learning_rate(float lr) = 0;
};

class IDynamicEmbeddingTable : public IGroupedEmbeddingTable {
 public:
  virtual void evict(const core23::Tensor& keys, size_t num_keys,
                     const core23::Tensor& id_space_offset, size_t num_id_space_offset,
                     const core23::Tensor& id_space_list) = 0;
};
}  // namespace embedding
