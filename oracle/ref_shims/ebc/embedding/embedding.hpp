/* TEST INFRASTRUCTURE ONLY -- stand-in for the reference's embedding/embedding.hpp (+ what it pulls
 * in) so that the CPU reference of embedding_collection in its unit tests
 * (R/test/utest/embedding_collection/embedding_table_cpu.hpp:27-125 and the class
 * EmbeddingReferenceCPU of reference_embedding.hpp:32-237) compiles with g++ into
 * oracle/_ref/libref_ebc.so.  Declarations only, restated after R/HugeCTR/embedding/common.hpp:
 * 129-226 (Combiner, EmbeddingLayout, LookupParam, the fields of EmbeddingCollectionParam the
 * class reads), embedding_storage/common.hpp:76-94 (EmbeddingTableParam) and
 * embedding_storage/embedding_table.hpp:25-49 (the dump() the class calls).  Every line of
 * forward / backward / update arithmetic in the library is the reference's. */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include <common.hpp>  // oracle/ref_shims/common.hpp: OptParams, Optimizer_t, Error_t, __half

#ifndef HCTR_CHECK
#define HCTR_CHECK(cond) \
  do { if (!(cond)) throw std::runtime_error("check failed: " #cond); } while (0)
#endif

namespace HugeCTR {
template <typename To, typename From>
struct TypeConvert;
template <>
struct TypeConvert<float, float> {
  static float convert(float v) { return v; }
};
template <>
struct TypeConvert<__half, float> {
  static __half convert(float v) { return __float2half(v); }
};
template <>
struct TypeConvert<float, __half> {
  static float convert(__half v) { return __half2float(v); }
};
}  // namespace HugeCTR

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
enum class Combiner : char { Sum, Average, Concat };
enum class EmbeddingLayout : int8_t { FeatureMajor, BatchMajor };

struct LookupParam {
  int lookup_id;
  int table_id;
  Combiner combiner;
  int max_hotness;
  int ev_size;
};
struct EmbeddingCollectionParam {
  int num_lookup;
  std::vector<LookupParam> lookup_params;
};
struct EmbeddingTableParam {
  int table_id;
  int64_t max_vocabulary_size;
  int ev_size;
  HugeCTR::OptParams opt_param;
};
class IGroupedEmbeddingTable {
 public:
  virtual ~IGroupedEmbeddingTable() = default;
  virtual void dump(core23::Tensor* keys, core23::Tensor* id_space_offset,
                    core23::Tensor* embedding_table, core23::Tensor* ev_size_list,
                    core23::Tensor* id_space) = 0;
};
}  // namespace embedding
