// Shim that compiles the REFERENCE's own hash functors from where they lie under /root/reference
// (nothing is copied into this repository) into oracle/_ref/libref_hash.so:
//   * HugeCTR/include/hashtable/cudf/hash_functions.cuh : MurmurHash3_32<Key> -- the hash of the
//     embedding hash table
//   * gpu_cache/include/hash_functions.cuh              : MurmurHash3_32<Key, seed>::hash and
//     Mod_Hash -- set / slab hash of the embedding cache (nv_gpu_cache.hpp:46-50)
// Both headers are plain C++ apart from the CUDA function qualifiers, defined away here.
// TEST INFRASTRUCTURE ONLY: the oracle (oracle/hctr_oracle.c, oracle/cache_oracle.py) is checked
// against this library in tests/test_ref_hash_cpu.py; the product never loads it.
#include <cstddef>
#include <cstdint>

#define __forceinline__ inline
#define __host__
#define __device__

#include "hashtable/cudf/hash_functions.cuh"  // -I <reference>/HugeCTR/include

namespace gpu_cache_ref {
#include "hash_functions.cuh"  // -I <reference>/gpu_cache/include (own namespace: same struct name)
}

extern "C" {
uint32_t ref_murmur3_u32(uint32_t key) { return MurmurHash3_32<uint32_t>()(key); }
uint32_t ref_murmur3_i64(long long key) { return MurmurHash3_32<long long>()(key); }
uint32_t ref_cache_murmur3_u32(uint32_t key) {
  return gpu_cache_ref::MurmurHash3_32<uint32_t>::hash(key);
}
uint32_t ref_cache_murmur3_i64(long long key) {
  return gpu_cache_ref::MurmurHash3_32<long long>::hash(key);
}
// slab_hasher = Mod_Hash<key_type, size_t> (nv_gpu_cache.hpp:49); first slab = hash % associativity
size_t ref_cache_mod_hash_i64(long long key) {
  return gpu_cache_ref::Mod_Hash<long long, size_t>::hash(key);
}
void ref_murmur3_i64_many(const long long* keys, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; i++) out[i] = MurmurHash3_32<long long>()(keys[i]);
}
void ref_murmur3_u32_many(const uint32_t* keys, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; i++) out[i] = MurmurHash3_32<uint32_t>()(keys[i]);
}
}
