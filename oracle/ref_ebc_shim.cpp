/* TEST INFRASTRUCTURE ONLY -- C entry points around the reference's CPU reference of
 * embedding_collection: EmbeddingTableCPU (R/test/utest/embedding_collection/
 * embedding_table_cpu.hpp:27-125, included from where it lies) and EmbeddingReferenceCPU
 * (reference_embedding.hpp:32-237).  The latter shares its header with CUDA test fixtures, so the
 * build recipe (oracle/Makefile `ref`) cuts the class -- and kahanSum, embedding_collection_utils.
 * hpp:93-118 -- out of the reference files with sed into oracle/_ref/gen/ (generated, git-ignored)
 * and this file includes the cuts.  oracle/hctr_oracle.c's hco_ebc_forward /
 * hco_ebc_backward_update are pinned against this library in tests/test_ref_ebc_cpu.py. */
#include <cstdio>

#include <utest/embedding_collection/embedding_table_cpu.hpp>  // from the reference checkout

#include "_ref/gen/kahan_sum.inc"

template <typename key_t, typename offset_t, typename index_t, typename emb_t>
#include "_ref/gen/embedding_reference_cpu_class.inc"

namespace {
/* the storage the reference class reads its initial tables from (IGroupedEmbeddingTable::dump) */
struct HostStorage : embedding::IGroupedEmbeddingTable {
  std::vector<long long> keys;
  std::vector<uint32_t> offsets;
  std::vector<float> vectors;
  std::vector<int> ev_sizes, table_ids;
  void dump(core23::Tensor* k, core23::Tensor* o, core23::Tensor* t, core23::Tensor* e,
            core23::Tensor* ids) override {
    *k = core23::Tensor(keys.data(), keys.size());
    *o = core23::Tensor(offsets.data(), offsets.size());
    *t = core23::Tensor(vectors.data(), vectors.size());
    *e = core23::Tensor(ev_sizes.data(), ev_sizes.size());
    *ids = core23::Tensor(table_ids.data(), table_ids.size());
  }
};
template <typename emb_t>
struct Handle {
  HostStorage st;
  std::unique_ptr<EmbeddingReferenceCPU<long long, uint32_t, uint32_t, emb_t>> ref;
  int num_gpus;
};
using H32 = Handle<float>;
using H16 = Handle<__half>;

template <typename emb_t>
void* create(int num_gpus, int num_table, const int* ev_sizes, float lr, float scaler,
             int num_lookup, const int* lookup_table, const int* combiner, const int* max_hotness,
             int batch_major, const long long* keys, const uint32_t* key_offsets,
             const float* vectors) {
  auto* h = new Handle<emb_t>();
  h->num_gpus = num_gpus;
  std::vector<embedding::EmbeddingTableParam> tp(num_table);
  size_t nvec = 0;
  for (int t = 0; t < num_table; t++) {
    tp[t].table_id = t;
    tp[t].max_vocabulary_size = key_offsets[t + 1] - key_offsets[t];
    tp[t].ev_size = ev_sizes[t];
    tp[t].opt_param.optimizer = HugeCTR::Optimizer_t::SGD;
    tp[t].opt_param.lr = lr;
    tp[t].opt_param.scaler = scaler;
    h->st.ev_sizes.push_back(ev_sizes[t]);
    h->st.table_ids.push_back(t);
    nvec += (size_t)(key_offsets[t + 1] - key_offsets[t]) * ev_sizes[t];
  }
  h->st.keys.assign(keys, keys + key_offsets[num_table]);
  h->st.offsets.assign(key_offsets, key_offsets + num_table + 1);
  h->st.vectors.assign(vectors, vectors + nvec);
  embedding::EmbeddingCollectionParam ebc;
  ebc.num_lookup = num_lookup;
  for (int l = 0; l < num_lookup; l++)
    ebc.lookup_params.push_back({l, lookup_table[l], static_cast<embedding::Combiner>(combiner[l]),
                                 max_hotness[l], ev_sizes[lookup_table[l]]});
  // every GPU hands the class one storage; the tables are read once (duplicates are checked equal)
  std::vector<std::vector<embedding::IGroupedEmbeddingTable*>> storages(num_gpus);
  storages[0].push_back(&h->st);
  h->ref.reset(new EmbeddingReferenceCPU<long long, uint32_t, uint32_t, emb_t>(
      num_gpus, ebc, num_table, tp, storages,
      batch_major ? embedding::EmbeddingLayout::BatchMajor : embedding::EmbeddingLayout::FeatureMajor));
  return h;
}
}  // namespace

extern "C" {

void* ref_ebc_create(int fp16, int num_gpus, int num_table, const int* ev_sizes, float lr,
                     float scaler, int num_lookup, const int* lookup_table, const int* combiner,
                     const int* max_hotness, int batch_major, const long long* keys,
                     const uint32_t* key_offsets, const float* vectors) {
  try {
    if (fp16)
      return create<__half>(num_gpus, num_table, ev_sizes, lr, scaler, num_lookup, lookup_table,
                            combiner, max_hotness, batch_major, keys, key_offsets, vectors);
    return create<float>(num_gpus, num_table, ev_sizes, lr, scaler, num_lookup, lookup_table,
                         combiner, max_hotness, batch_major, keys, key_offsets, vectors);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_ebc_create: %s\n", ex.what());
    return nullptr;
  }
}

void ref_ebc_destroy(void* h, int fp16) {
  if (fp16) delete static_cast<H16*>(h);
  else delete static_cast<H32*>(h);
}

/* forward: keys [n] lookup-major (bucket = lookup * batch + b), bucket_range [num_lookup * batch
 * + 1]; out = the per-GPU outputs back to back, each of per_gpu_len floats (fp16 widened) */
int ref_ebc_forward(void* hv, int fp16, const long long* keys, size_t n,
                    const uint32_t* bucket_range, size_t n_range, float* out, size_t per_gpu_len) {
  try {
    std::vector<long long> k(keys, keys + n);
    std::vector<uint32_t> br(bucket_range, bucket_range + n_range);
    auto run = [&](auto* h) {
      h->ref->embedding_forward_cpu(k, br);
      for (int g = 0; g < h->num_gpus; g++) {
        if (h->ref->embedding_vec_[g].size() != per_gpu_len) throw std::runtime_error("out size");
        for (size_t i = 0; i < per_gpu_len; i++)
          out[(size_t)g * per_gpu_len + i] = (float)h->ref->embedding_vec_[g][i];
      }
    };
    if (fp16) run(static_cast<H16*>(hv));
    else run(static_cast<H32*>(hv));
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_ebc_forward: %s\n", ex.what());
    return 1;
  }
  return 0;
}

/* backward + SGD update; top_grad = per-GPU gradients back to back in the forward's layout */
int ref_ebc_backward_update(void* hv, int fp16, const float* top_grad, size_t per_gpu_len,
                            const long long* keys, size_t n, const uint32_t* bucket_range,
                            size_t n_range) {
  try {
    std::vector<long long> k(keys, keys + n);
    std::vector<uint32_t> br(bucket_range, bucket_range + n_range);
    auto run = [&](auto* h, auto zero) {
      using emb_t = decltype(zero);
      std::vector<std::vector<emb_t>> tg(h->num_gpus);
      for (int g = 0; g < h->num_gpus; g++)
        for (size_t i = 0; i < per_gpu_len; i++)
          tg[g].push_back(HugeCTR::TypeConvert<emb_t, float>::convert(
              top_grad[(size_t)g * per_gpu_len + i]));
      h->ref->embedding_backward_cpu(tg, k, br);
      h->ref->embedding_update_cpu();
    };
    if (fp16) run(static_cast<H16*>(hv), __half());
    else run(static_cast<H32*>(hv), 0.0f);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_ebc_backward_update: %s\n", ex.what());
    return 1;
  }
  return 0;
}

/* current vector of (table, key) */
int ref_ebc_get(void* hv, int fp16, int table, long long key, float* out) {
  auto get = [&](auto* h) {
    auto& m = h->ref->emb_table_cpu_.emb_table_list_[table];
    auto it = m.find(key);
    if (it == m.end()) return 1;
    std::memcpy(out, it->second.data(), it->second.size() * sizeof(float));
    return 0;
  };
  return fp16 ? get(static_cast<H16*>(hv)) : get(static_cast<H32*>(hv));
}
}
