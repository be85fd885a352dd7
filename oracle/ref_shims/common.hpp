/* TEST INFRASTRUCTURE ONLY -- stand-in for the reference's HugeCTR/include/common.hpp so that the
 * reference's own CPU oracle of the sparse embedding (R/test/utest/embedding/
 * sparse_embedding_hash_cpu.hpp, with cpu_hashtable.hpp and the data_readers/ headers it pulls in)
 * compiles from where it lies with g++: the real common.hpp drags in cuBLAS / cuRAND / NVML / MPI /
 * NCCL.  Only declarations are restated here (enum and field NAMES the oracle's code refers to,
 * after R/HugeCTR/include/common.hpp:67-94,184-191 and optimizer.hpp:30-160); every function body
 * that ends up in oracle/_ref/libref_embedding.so is the reference's.  See oracle/Makefile `ref`. */
#pragma once
#include <immintrin.h>

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace HugeCTR {

enum class Error_t {
  Success, FileCannotOpen, BrokenFile, OutOfBound, OutOfMemory, WrongInput, IllegalCall,
  NotInitialized, UnSupportedFormat, InvalidEnv, DataCheckError, UnspecificError, EndOfFile
};
enum class Check_t { Sum, None, Unknown };
enum class Optimizer_t { Ftrl, Adam, RMSProp, AdaGrad, Nesterov, MomentumSGD, SGD, DEFAULT,
                         NOT_INITIALIZED };
enum class Update_t { Local, Global, LazyGlobal };

typedef struct DataSetHeader_ {
  long long error_check;
  long long number_of_records;
  long long label_dim;
  long long dense_dim;
  long long slot_num;
  long long reserved[3];
} DataSetHeader;

struct FtrlOptHyperParams { float beta = 0.f, lambda1 = 0.f, lambda2 = 0.f; };
struct AdamOptHyperParams {
  uint64_t times = 0; float beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-7f;
  // optimizer.hpp:58-60 (used by the dynamic-table CPU mirror, oracle/ref_det_shim.cpp)
  inline float bias() const {
    return std::sqrt(1 - std::pow(beta2, times)) / (1 - std::pow(beta1, times));
  }
};
struct RMSPropOptHyperParams { float beta = 0.9f, epsilon = 1e-7f; };
struct AdaGradOptHyperParams { float initial_accu_value = 0.f, epsilon = 1e-7f; };
struct MomentumSGDOptHyperParams { float factor = 0.1f; };
struct NesterovOptHyperParams { float mu = 0.9f; };
struct SGDOptHyperParams { bool atomic_update = false; };
struct OptHyperParams {
  FtrlOptHyperParams ftrl;
  AdamOptHyperParams adam;
  RMSPropOptHyperParams rmsprop;
  AdaGradOptHyperParams adagrad;
  MomentumSGDOptHyperParams momentum;
  NesterovOptHyperParams nesterov;
  SGDOptHyperParams sgd;
};
struct OptParams {
  Optimizer_t optimizer{Optimizer_t::SGD};
  float lr{};
  OptHyperParams hyperparams;
  Update_t update_type{Update_t::Local};
  float scaler{};
  // state vectors per weight: optimizer.hpp:30-140 (`num_parameters_per_weight` of each
  // hyper-parameter struct: Ftrl 2, Adam 2, RMSProp / AdaGrad / MomentumSGD / Nesterov 1, SGD 0)
  inline size_t num_parameters_per_weight() const {
    switch (optimizer) {
      case Optimizer_t::Ftrl:
      case Optimizer_t::Adam: return 2;
      case Optimizer_t::RMSProp:
      case Optimizer_t::AdaGrad:
      case Optimizer_t::MomentumSGD:
      case Optimizer_t::Nesterov: return 1;
      default: return 0;
    }
  }
};

}  // namespace HugeCTR

#define HCTR_OWN_THROW(err, msg) \
  do { (void)(err); throw std::runtime_error(std::string(msg)); } while (0)
#define HCTR_LOCATION() ""
#define HCTR_LOG_S(level, where) std::cerr
#define HCTR_LOG(level, where, ...) std::fprintf(stderr, __VA_ARGS__)
#define HCTR_PRINT_FUNC_NAME_() do { } while (0)
#define HCTR_CHECK_HINT(cond, ...) \
  do { if (!(cond)) throw std::runtime_error("check failed: " #cond); } while (0)

/* IEEE binary16 with round-to-nearest-even conversions (what __float2half / __half2float do) */
struct __half {  // with the implicit float conversions cuda_fp16.h gives host code
  unsigned short bits;
#ifdef REFSHIM_TRIVIAL_HALF  // (a member of unions in the reference's device code, as cuda_fp16's is)
  __half() = default;
#else
  __half() : bits(0) {}
#endif
  __half(float v) : bits(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT)) {}
  __half(int v) : bits(_cvtss_sh((float)v, _MM_FROUND_TO_NEAREST_INT)) {}
  operator float() const { return _cvtsh_ss(bits); }
};
static inline __half __float2half(float v) {
  __half h;
  h.bits = _cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT);
  return h;
}
static inline float __half2float(__half h) { return _cvtsh_ss(h.bits); }
