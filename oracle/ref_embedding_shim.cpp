/* TEST INFRASTRUCTURE ONLY -- C entry points around the reference's own CPU oracle of the legacy
 * sparse embedding, SparseEmbeddingHashCpu<long long | unsigned, float / __half>
 * (R/test/utest/embedding/sparse_embedding_hash_cpu.hpp:52-1015), compiled from the reference
 * checkout into oracle/_ref/libref_embedding.so (oracle/Makefile, target `ref`).  One step is what
 * the reference tests drive (localized_slot_sparse_embedding_hash_test.cu:181-519): read a batch
 * of the Norm dataset, forward, backward with the forward output as the top gradient, update. */
#include <common.hpp>  // oracle/ref_shims/common.hpp

#include <chrono>

#include <utest/embedding/sparse_embedding_hash_cpu.hpp>

namespace {
template <typename Emb, typename Key = long long>
struct Handle {
  SparseEmbeddingHashCpu<Key, Emb> e;
  int batch, slots, dim, vocab;
  // wall seconds spent in the reference's three calls since creation (stages
  // timed separately): forward() = read_a_batch + hash get/insert + pooling, backward(),
  // update_params() = sort / unduplicate + optimizer
  double stage_s[3] = {0.0, 0.0, 0.0};
  template <typename... A>
  Handle(int b, int s, int d, int v, A&&... a)
      : e(std::forward<A>(a)...), batch(b), slots(s), dim(d), vocab(v) {}
};
using H32 = Handle<float>;
using H16 = Handle<__half>;
using HU32 = Handle<float, unsigned int>;  // `fp16` argument == 2: u32 keys, fp32 vectors
}  // namespace

extern "C" {

void* ref_emb_create(int fp16, int batch, int max_feature_num, int vocab, int dim, int slot_num,
                     int label_dim, int dense_dim, int check_sum, long long num_records,
                     int combiner, int optimizer, int update_type, float lr, float scaler,
                     float beta1, float beta2, float epsilon, float momentum_or_mu,
                     const char* file_list, const char* model_dir) {
  OptParams p;
  p.optimizer = static_cast<Optimizer_t>(optimizer);
  p.update_type = static_cast<Update_t>(update_type);
  p.lr = lr;
  p.scaler = scaler;
  p.hyperparams.adam.beta1 = beta1;
  p.hyperparams.adam.beta2 = beta2;
  p.hyperparams.adam.epsilon = epsilon;
  p.hyperparams.adagrad.epsilon = epsilon;
  p.hyperparams.rmsprop.beta = beta2;
  p.hyperparams.rmsprop.epsilon = epsilon;
  p.hyperparams.momentum.factor = momentum_or_mu;
  p.hyperparams.nesterov.mu = momentum_or_mu;
  const Check_t chk = check_sum ? Check_t::Sum : Check_t::None;
  try {
    if (fp16 == 2)
      return new HU32(batch, slot_num, dim, vocab, batch, max_feature_num, vocab, dim, slot_num,
                      label_dim, dense_dim, chk, num_records, combiner, p, std::string(file_list),
                      std::string(model_dir), SparseEmbedding_t::Localized);
    if (fp16)
      return new H16(batch, slot_num, dim, vocab, batch, max_feature_num, vocab, dim, slot_num,
                     label_dim, dense_dim, chk, num_records, combiner, p, std::string(file_list),
                     std::string(model_dir), SparseEmbedding_t::Localized);
    return new H32(batch, slot_num, dim, vocab, batch, max_feature_num, vocab, dim, slot_num,
                   label_dim, dense_dim, chk, num_records, combiner, p, std::string(file_list),
                   std::string(model_dir), SparseEmbedding_t::Localized);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_emb_create: %s\n", ex.what());
    return nullptr;
  }
}

void ref_emb_destroy(void* h, int fp16) {
  if (fp16 == 2) delete static_cast<HU32*>(h);
  else if (fp16) delete static_cast<H16*>(h);
  else delete static_cast<H32*>(h);
}

/* forward (reads the next batch), then optionally backward and update.  fwd / wgrad: float
 * [batch][slot][dim] (the fp16 instance's values widened); either may be NULL */
int ref_emb_step(void* h, int fp16, int train, float* fwd, float* wgrad) {
  try {
    if (fp16 == 1) {
      H16* o = static_cast<H16*>(h);
      const size_t n = (size_t)o->batch * o->slots * o->dim;
      o->e.forward();
      if (fwd) for (size_t i = 0; i < n; i++) fwd[i] = __half2float(o->e.get_forward_results()[i]);
      if (train) {
        o->e.backward();
        if (wgrad)
          for (size_t i = 0; i < n; i++) wgrad[i] = __half2float(o->e.get_backward_results()[i]);
        o->e.update_params();
      }
    } else {
      auto run = [&](auto* o) {
        using clk = std::chrono::steady_clock;
        auto secs = [](clk::time_point a, clk::time_point b) {
          return std::chrono::duration<double>(b - a).count();
        };
        const size_t n = (size_t)o->batch * o->slots * o->dim;
        auto t0 = clk::now();
        o->e.forward();
        auto t1 = clk::now();
        o->stage_s[0] += secs(t0, t1);
        if (fwd) std::memcpy(fwd, o->e.get_forward_results(), n * sizeof(float));
        if (train) {
          t0 = clk::now();
          o->e.backward();
          t1 = clk::now();
          o->stage_s[1] += secs(t0, t1);
          if (wgrad) std::memcpy(wgrad, o->e.get_backward_results(), n * sizeof(float));
          t0 = clk::now();
          o->e.update_params();
          o->stage_s[2] += secs(t0, clk::now());
        }
      };
      if (fp16 == 2) run(static_cast<HU32*>(h));
      else run(static_cast<H32*>(h));
    }
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "ref_emb_step: %s\n", ex.what());
    return 1;
  }
  return 0;
}

/* seconds in forward / backward / update_params since creation (fp32 instances only) */
void ref_emb_stage_seconds(void* h, int fp16, double* out3) {
  const double* s = fp16 == 2 ? static_cast<HU32*>(h)->stage_s
                    : fp16    ? static_cast<H16*>(h)->stage_s
                              : static_cast<H32*>(h)->stage_s;
  for (int i = 0; i < 3; i++) out3[i] = s[i];
}

/* the table: keys [vocab] and vectors [vocab][dim] in the oracle's row order */
void ref_emb_table(void* h, int fp16, long long* keys, float* values) {
  auto get = [&](auto* o) {
    for (int i = 0; i < o->vocab; i++) keys[i] = (long long)o->e.get_hash_table_key_ptr()[i];
    std::memcpy(values, o->e.get_hash_table_value_ptr(), (size_t)o->vocab * o->dim * sizeof(float));
  };
  if (fp16 == 2) get(static_cast<HU32*>(h));
  else if (fp16) get(static_cast<H16*>(h));
  else get(static_cast<H32*>(h));
}
}
