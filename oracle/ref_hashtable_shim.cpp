// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's own GPU hash table of the legacy
// embedding -- HugeCTR::HashTable<KeyType, size_t> (R/HugeCTR/src/hashtable/nv_hashtable.cu, whole:
// insert / get_insert / get_mark / get / size / dump kernels and the host class) on the cuDF-derived
// concurrent_unordered_map (R/HugeCTR/include/hashtable/cudf/concurrent_unordered_map.cuh: atomicCAS
// claim of a bucket, linear probing, row numbers from an atomic counter) with its MurmurHash3_32
// (cudf/hash_functions.cuh) -- compiled from the checkout as plain C++ and executed by the host
// interpreter of tests/emu (CUDA threads = fibers; thread blocks in block order and threads in
// thread order, i.e. keys are inserted in position order: the interleaving the oracle restates).
// The two files that contain launches are rewritten into _ref/gen/ (ref_launch_rewrite.py: launch
// syntax, dynamic shared memory); the other headers are included where they lie.  ref_shims/
// supplies declaration-only stand-ins for the CUDA, thrust and core23 headers they name.
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <iterator>
#include <limits>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "ref_shims/cuda/cuda_runtime_api.h"

#include <common.hpp>  // oracle/ref_shims/common.hpp

#define HCTR_LIB_THROW(expr) \
  do { if ((expr) != cudaSuccess) throw std::runtime_error("cuda stand-in reported an error"); } while (0)

#include "_ref/gen/nv_hashtable.gen.cpp"

using namespace HugeCTR;

namespace {
struct Handle {
  int key_bytes;
  HashTable<long long, size_t>* t64 = nullptr;
  HashTable<unsigned int, size_t>* t32 = nullptr;
};
template <typename F64, typename F32>
void both(void* hv, F64 f64, F32 f32) {
  Handle* h = (Handle*)hv;
  if (h->key_bytes == 8)
    f64(h->t64);
  else
    f32(h->t32);
}
}  // namespace

extern "C" {
void* refht_create(size_t capacity, int key_bytes) {
  hipemu::set_wave_width(64);
  hipemu::set_max_workers(1);
  Handle* h = new Handle;
  h->key_bytes = key_bytes;
  if (key_bytes == 8)
    h->t64 = new HashTable<long long, size_t>(capacity);
  else
    h->t32 = new HashTable<unsigned int, size_t>(capacity);
  return h;
}
void refht_destroy(void* hv) {
  Handle* h = (Handle*)hv;
  delete h->t64;
  delete h->t32;
  delete h;
}
// workers = 1: blocks in order (keys in position order); 0: one OS thread per block
void refht_schedule(size_t workers) { hipemu::set_max_workers(workers); }
void refht_get_insert(void* hv, const void* keys, size_t* vals, size_t n) {
  both(hv, [&](auto* t) { t->get_insert((const long long*)keys, vals, n, nullptr); },
       [&](auto* t) { t->get_insert((const unsigned*)keys, vals, n, nullptr); });
}
void refht_get_mark(void* hv, const void* keys, size_t* vals, size_t n) {
  both(hv, [&](auto* t) { t->get_mark((const long long*)keys, vals, n, nullptr); },
       [&](auto* t) { t->get_mark((const unsigned*)keys, vals, n, nullptr); });
}
void refht_insert(void* hv, const void* keys, const size_t* vals, size_t n) {
  both(hv, [&](auto* t) { t->insert((const long long*)keys, vals, n, nullptr); },
       [&](auto* t) { t->insert((const unsigned*)keys, vals, n, nullptr); });
}
size_t refht_size(void* hv) {
  size_t r = 0;
  both(hv, [&](auto* t) { r = t->get_size(nullptr); }, [&](auto* t) { r = t->get_size(nullptr); });
  return r;
}
size_t refht_value_head(void* hv) {
  size_t r = 0;
  both(hv, [&](auto* t) { r = t->get_value_head(nullptr); },
       [&](auto* t) { r = t->get_value_head(nullptr); });
  return r;
}
size_t refht_capacity(void* hv) {
  size_t r = 0;
  both(hv, [&](auto* t) { r = t->get_capacity(); }, [&](auto* t) { r = t->get_capacity(); });
  return r;
}
// (key, row) pairs in bucket order, as dump_kernel gathers them
size_t refht_dump(void* hv, void* keys, size_t* vals) {
  size_t n = 0;
  both(hv, [&](auto* t) { t->dump((long long*)keys, vals, &n, nullptr); },
       [&](auto* t) { t->dump((unsigned*)keys, vals, &n, nullptr); });
  return n;
}
}
