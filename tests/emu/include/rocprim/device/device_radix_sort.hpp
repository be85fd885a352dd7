// <rocprim/device/device_radix_sort.hpp> of the host interpreter -- TEST INFRASTRUCTURE ONLY.
// radix_sort_pairs as a stable host sort on the key bits [begin_bit, end_bit).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <typename K, typename V>
static inline hipError_t radix_sort_pairs(void* temp, size_t& temp_bytes, const K* kin, K* kout,
                                          const V* vin, V* vout, size_t n, unsigned begin_bit,
                                          unsigned end_bit, hipStream_t = nullptr,
                                          bool = false) {
  if (temp == nullptr) {
    temp_bytes = 256;
    return hipSuccess;
  }
  typedef typename std::make_unsigned<K>::type U;
  const unsigned bits = end_bit - begin_bit;
  const U mask = bits >= sizeof(U) * 8 ? (U)~(U)0 : (U)((((U)1) << bits) - 1);
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
    return (((U)kin[a] >> begin_bit) & mask) < (((U)kin[b] >> begin_bit) & mask);
  });
  std::vector<K> k(n);
  std::vector<V> v(n);
  for (size_t i = 0; i < n; i++) {
    k[i] = kin[idx[i]];
    v[i] = vin[idx[i]];
  }
  std::copy(k.begin(), k.end(), kout);
  std::copy(v.begin(), v.end(), vout);
  return hipSuccess;
}
}  // namespace rocprim
