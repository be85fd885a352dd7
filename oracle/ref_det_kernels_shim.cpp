// TEST INFRASTRUCTURE ONLY: C entry point over the REFERENCE's device code of the dynamic embedding
// table's fused optimizer step (R/HugeCTR/embedding_storage/optimizers.cuh:29-233: sgd / momentum /
// nesterov / ada_grad / rms_prop / adam / ftrl _update_grad_kernel), cut out of the checkout by
// oracle/Makefile (from `namespace embedding {` on: the include of core23/data_type_helpers.cuh in
// front of it is replaced by the two-line TypeConverter below) and executed by the host interpreter
// of tests/emu with the launch shape of DynamicEmbeddingTable::update
// (R/HugeCTR/embedding_storage/dynamic_embedding.cu:222-317: one thread per unique key, blocks of
// 256).  The kernels rewrite wgrad in place into the weight DELTA (the caller scatter-adds it) and
// update the state vectors through per-key pointers.
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <cmath>

#include "ref_shims/cuda/cuda_runtime_api.h"

namespace core23 {
template <typename Dst, typename Src>
struct TypeConverter {
  static Dst value(Src s) { return (Dst)s; }  // (wgrad_t = float here: the identity)
};
}  // namespace core23
using std::abs;
using std::signbit;

#include "_ref/gen/det_optimizers.inc"

extern "C" {
// opt: the reference's Optimizer_t values as det_oracle uses them (Ftrl 0, Adam 1, RMSProp 2,
// AdaGrad 3, Nesterov 4, MomentumSGD 5, SGD 6; R/HugeCTR/include/common.hpp:82-92).
// a, b, c: the kernel's hyper-parameters in its own argument order after lr (see the switch).
int refdetk_update(int opt, uint32_t num_ev, const uint32_t* ev_offsets, float** states,
                   float** weights, float lr, float a, float b, float c, float scaler, float* g) {
  using namespace embedding;
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(0);
  const int block = 256, grid = (int)((num_ev - 1) / block + 1);
  int rc = 0;
  switch (opt) {
    case 6: REFEMU_LAUNCH((sgd_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, scaler, g); break;
    case 5: REFEMU_LAUNCH((momentum_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, a, states, scaler, g); break;
    case 4: REFEMU_LAUNCH((nesterov_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, a, states, scaler, g); break;
    case 3: REFEMU_LAUNCH((ada_grad_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, states, a, scaler, g); break;
    case 2: REFEMU_LAUNCH((rms_prop_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, a, states, b, scaler, g); break;
    // adam: lr = lr * bias() (dynamic_embedding.cu:238), a = beta1, b = beta2, c = epsilon
    case 1: REFEMU_LAUNCH((adam_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, a, b, states, c, scaler, g); break;
    // ftrl: a = lambda1, b = lambda2 + beta / lr (dynamic_embedding.cu:220-221)
    case 0: REFEMU_LAUNCH((ftrl_update_grad_kernel<float>), (grid, block), ev_offsets, num_ev, lr, a, b, states, weights, scaler, g); break;
    default: rc = 1;
  }
  hipemu::set_wave_width(64);
  return rc;
}
}
