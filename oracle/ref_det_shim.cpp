/* TEST INFRASTRUCTURE ONLY -- C entry points around the reference's CPU mirror of the dynamic
 * embedding table, embedding::DynamicEmbeddingTableCPU<long long>
 * (R/HugeCTR/embedding_storage/dynamic_embedding_cpu.hpp:32-485, with its optimizer formulas in
 * R/HugeCTR/embedding_storage/optimizers.hpp:25-199), compiled from the reference checkout into
 * oracle/_ref/libref_det.so (oracle/Makefile, target `ref`).  oracle/det_oracle.py (the restated
 * oracle the GPU tests of hctr_det_* compare against) is pinned against it in
 * tests/test_ref_det_cpu.py. */
#include <core23/logger.hpp>  // oracle/ref_shims/det/core23/logger.hpp

#include <embedding_storage/dynamic_embedding_cpu.hpp>

namespace {
using Table = embedding::DynamicEmbeddingTableCPU<long long>;
struct Handle {
  std::vector<embedding::EmbeddingTableParam> tp;
  embedding::EmbeddingCollectionParam ebc;
  std::vector<int> ev;
  std::unique_ptr<Table> t;
};
core23::Tensor view(const void* p, size_t n) { return core23::Tensor(const_cast<void*>(p), n); }
}  // namespace

extern "C" {

void* ref_det_create(int num_tables, const int* ev_sizes, int optimizer, float lr, float scaler,
                     float beta1, float beta2, float epsilon, float momentum, float rms_beta,
                     float lambda1, float lambda2, float ftrl_beta) {
  HugeCTR::OptParams p;
  p.optimizer = static_cast<HugeCTR::Optimizer_t>(optimizer);
  p.lr = lr;
  p.scaler = scaler;
  p.hyperparams.adam.beta1 = beta1;
  p.hyperparams.adam.beta2 = beta2;
  p.hyperparams.adam.epsilon = epsilon;
  p.hyperparams.adagrad.epsilon = epsilon;
  p.hyperparams.rmsprop.beta = rms_beta;
  p.hyperparams.rmsprop.epsilon = epsilon;
  p.hyperparams.momentum.factor = momentum;
  p.hyperparams.nesterov.mu = momentum;
  p.hyperparams.ftrl.lambda1 = lambda1;
  p.hyperparams.ftrl.lambda2 = lambda2;
  p.hyperparams.ftrl.beta = ftrl_beta;
  auto* h = new Handle();
  embedding::GroupedTableParam g;
  for (int i = 0; i < num_tables; i++) {
    embedding::EmbeddingTableParam t;
    t.table_id = i;
    t.max_vocabulary_size = -1;
    t.ev_size = ev_sizes[i];
    t.opt_param = p;
    h->tp.push_back(t);
    h->ev.push_back(ev_sizes[i]);
    g.table_ids.push_back(i);
  }
  h->ebc.grouped_table_params.push_back(g);
  h->t.reset(new Table(h->tp, h->ebc, 0, p));
  return h;
}

void ref_det_destroy(void* hv) { delete static_cast<Handle*>(hv); }

/* keys [n] grouped by table: table_ids[i] owns keys[offsets[i] .. offsets[i+1]) */
int ref_det_load(void* hv, const long long* keys, size_t n, const uint32_t* offsets,
                 size_t n_offsets, const int32_t* table_ids, const float* vectors) {
  Handle* h = static_cast<Handle*>(hv);
  try {
    std::vector<uint32_t> sizes(n);
    size_t total = 0;
    for (size_t i = 0; i + 1 < n_offsets; i++)
      for (uint32_t j = offsets[i]; j < offsets[i + 1]; j++) {
        sizes[j] = (uint32_t)h->ev[table_ids[i]];
        total += sizes[j];
      }
    core23::Tensor k = view(keys, n), o = view(offsets, n_offsets), v = view(vectors, total),
                   s = view(sizes.data(), n), t = view(table_ids, n_offsets - 1);
    h->t->load(k, o, v, s, t);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_det_load: %s\n", ex.what());
    return 1;
  }
  return 0;
}

/* ILookup::lookup: out receives the vectors back to back in key order */
int ref_det_lookup(void* hv, const long long* keys, size_t n, const uint32_t* offsets,
                   size_t n_offsets, const int32_t* table_ids, float* out) {
  Handle* h = static_cast<Handle*>(hv);
  try {
    std::vector<float*> ptrs(n);
    size_t pos = 0;
    for (size_t i = 0; i + 1 < n_offsets; i++)
      for (uint32_t j = offsets[i]; j < offsets[i + 1]; j++) {
        ptrs[j] = out + pos;
        pos += (size_t)h->ev[table_ids[i]];
      }
    core23::Tensor ev = view(ptrs.data(), n);
    h->t->lookup(view(keys, n), n, view(offsets, n_offsets), n_offsets,
                 view(table_ids, n_offsets - 1), ev);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_det_lookup: %s\n", ex.what());
    return 1;
  }
  return 0;
}

/* IGroupedEmbeddingTable::update with the reference's Wgrad pieces
 * (R/HugeCTR/embedding/common.hpp:352-373): unique_keys [n], table id per key [n] (ascending),
 * ev_start_indices [n + 1], data [ev_start_indices[n]] */
int ref_det_update(void* hv, const long long* unique_keys, size_t n, const int* table_id_per_key,
                   const uint32_t* ev_start_indices, const float* wgrad) {
  Handle* h = static_cast<Handle*>(hv);
  try {
    uint64_t num = n;
    h->t->update(view(unique_keys, n), view(&num, 1), view(table_id_per_key, n),
                 view(ev_start_indices, n + 1), view(wgrad, ev_start_indices[n]));
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_det_update: %s\n", ex.what());
    return 1;
  }
  return 0;
}

size_t ref_det_size(void* hv) { return static_cast<Handle*>(hv)->t->size(); }
}
