// <thrust/pair.h> stand-in -- TEST INFRASTRUCTURE ONLY (see ../cuda_runtime_api.h): the two-member
// aggregate and make_pair that the reference's hash map stores in its buckets.
#pragma once
namespace thrust {
template <typename T1, typename T2>
struct pair {
  using first_type = T1;
  using second_type = T2;
  T1 first;
  T2 second;
};
template <typename T1, typename T2>
static inline pair<T1, T2> make_pair(T1 a, T2 b) {
  return pair<T1, T2>{a, b};
}
}  // namespace thrust
