// TEST INFRASTRUCTURE ONLY: C entry points over the REFERENCE's embedding cache, compiled from where
// it lies (R/gpu_cache/src/nv_gpu_cache.cu + R/gpu_cache/include/*.hpp, the non-libcu++ branch) as
// plain C++ and executed by the host interpreter of tests/emu (hipemu: CUDA threads = fibers,
// 32-lane tiles = wavefronts of width 32, thread blocks one after the other in block order, so the
// per-set mutexes are taken in key-position order -- the interleaving oracle/cache_oracle.py
// restates).  tests/test_ref_cache_cpu.py drives random Query / Replace / Update / Dump sequences
// through it and through the oracle and compares results AND internal state (keys per slot, LRU
// counters, vectors).  The only edit made to the reference's text is the launch syntax
// (oracle/ref_launch_rewrite.py -> _ref/gen/nv_gpu_cache.gen.cpp, generated, never committed); the
// CUDA headers it includes are the declaration-only stand-ins of ref_shims/cuda/.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "ref_shims/cuda/cuda_runtime_api.h"
#include "ref_shims/cuda/cooperative_groups.h"

// (state comparison needs the members: keys_, vals_, slot_counter_, global_counter_)
#define private public
#include "_ref/gen/nv_gpu_cache.gen.cpp"
#undef private

namespace {
using Cache64 = gpu_cache::gpu_cache<long long, uint64_t, std::numeric_limits<long long>::max(),
                                     SET_ASSOCIATIVITY, SLAB_SIZE>;
using Cache32 = gpu_cache::gpu_cache<unsigned int, uint64_t,
                                     std::numeric_limits<unsigned int>::max(), SET_ASSOCIATIVITY,
                                     SLAB_SIZE>;
struct Handle {
  int key_bytes;
  Cache64* c64 = nullptr;
  Cache32* c32 = nullptr;
  size_t sets, vec;
};
template <typename F64, typename F32>
void both(Handle* h, F64 f64, F32 f32) {
  if (h->key_bytes == 8)
    f64(h->c64);
  else
    f32(h->c32);
}
}  // namespace

extern "C" {
// workers = 1: thread blocks strictly in block order (deterministic, = key-position order);
// 0: one OS thread per block while the grid is small (the mutexes are really contended)
void refcache_schedule(size_t workers) {
  hipemu::set_wave_width(32);
  hipemu::set_max_workers(workers);
}

void* refcache_create(size_t capacity_in_set, size_t vec, int key_bytes) {
  refcache_schedule(1);
  Handle* h = new Handle;
  h->key_bytes = key_bytes;
  h->sets = capacity_in_set;
  h->vec = vec;
  if (key_bytes == 8)
    h->c64 = new Cache64(capacity_in_set, vec);
  else
    h->c32 = new Cache32(capacity_in_set, vec);
  return h;
}

void refcache_destroy(void* hv) {
  Handle* h = (Handle*)hv;
  delete h->c64;
  delete h->c32;
  delete h;
}

void refcache_query(void* hv, const void* keys, size_t len, float* values, uint64_t* miss_index,
                    void* miss_keys, size_t* miss_len) {
  both((Handle*)hv,
       [&](Cache64* c) {
         c->Query((const long long*)keys, len, values, miss_index, (long long*)miss_keys, miss_len,
                  nullptr);
       },
       [&](Cache32* c) {
         c->Query((const unsigned*)keys, len, values, miss_index, (unsigned*)miss_keys, miss_len,
                  nullptr);
       });
}

void refcache_replace(void* hv, const void* keys, size_t len, const float* values) {
  both((Handle*)hv, [&](Cache64* c) { c->Replace((const long long*)keys, len, values, nullptr); },
       [&](Cache32* c) { c->Replace((const unsigned*)keys, len, values, nullptr); });
}

void refcache_update(void* hv, const void* keys, size_t len, const float* values) {
  both((Handle*)hv, [&](Cache64* c) { c->Update((const long long*)keys, len, values, nullptr); },
       [&](Cache32* c) { c->Update((const unsigned*)keys, len, values, nullptr); });
}

void refcache_dump(void* hv, void* keys, size_t* count, size_t start_set, size_t end_set) {
  both((Handle*)hv,
       [&](Cache64* c) { c->Dump((long long*)keys, count, start_set, end_set, nullptr); },
       [&](Cache32* c) { c->Dump((unsigned*)keys, count, start_set, end_set, nullptr); });
}

// internal state, slot = (set * SET_ASSOCIATIVITY + slab) * SLAB_SIZE + lane: keys widened to
// int64 with `empty` flags, LRU counters, vectors, the global counter
void refcache_state(void* hv, long long* keys, unsigned char* empty, uint64_t* counters,
                    float* vals, uint64_t* global_counter) {
  Handle* h = (Handle*)hv;
  const size_t slots = h->sets * SET_ASSOCIATIVITY * SLAB_SIZE;
  both(h,
       [&](Cache64* c) {
         const long long* k = (const long long*)c->keys_;
         for (size_t i = 0; i < slots; i++) {
           keys[i] = k[i];
           empty[i] = k[i] == std::numeric_limits<long long>::max();
         }
         memcpy(counters, c->slot_counter_, slots * sizeof(uint64_t));
         memcpy(vals, c->vals_, slots * h->vec * sizeof(float));
         *global_counter = *c->global_counter_;
       },
       [&](Cache32* c) {
         const unsigned* k = (const unsigned*)c->keys_;
         for (size_t i = 0; i < slots; i++) {
           keys[i] = (long long)k[i];
           empty[i] = k[i] == std::numeric_limits<unsigned>::max();
         }
         memcpy(counters, c->slot_counter_, slots * sizeof(uint64_t));
         memcpy(vals, c->vals_, slots * h->vec * sizeof(float));
         *global_counter = *c->global_counter_;
       });
}

// the two hash functions the kernels use, for the oracle's set / slab choice
size_t refcache_set_of(long long key, size_t capacity_in_set, int key_bytes) {
  return key_bytes == 8 ? MurmurHash3_32<long long>::hash(key) % capacity_in_set
                        : MurmurHash3_32<unsigned>::hash((unsigned)key) % capacity_in_set;
}
}
