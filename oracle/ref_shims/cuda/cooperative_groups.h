// <cooperative_groups.h> stand-in -- TEST INFRASTRUCTURE ONLY (see cuda_runtime_api.h): the tile
// interface the reference's cache kernels use (tiled_partition<32>(this_thread_block()),
// thread_rank / meta_group_rank / size / sync / ballot / shfl / shfl_xor), on the interpreter's
// wavefront collectives.  A tile is one wavefront of the interpreter (set_wave_width(32)); the call
// site number is the caller's source line, so lanes that wait at different calls are told apart.
#pragma once
#include "cuda_runtime_api.h"

namespace cooperative_groups {
struct thread_block {};
static inline thread_block this_thread_block() { return thread_block{}; }

namespace detail {
template <typename T>
static inline uint64_t to_bits(T v) {
  static_assert(sizeof(T) <= 8, "shuffles move at most 64 bits");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template <typename T>
static inline T from_bits(uint64_t b) {
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}
}  // namespace detail

template <int Size>
class thread_block_tile {
  static_assert(Size == 32, "the interpreter runs this build with 32-lane wavefronts");

 public:
  unsigned thread_rank() const { return threadIdx.x % Size; }
  unsigned meta_group_rank() const { return threadIdx.x / Size; }
  static constexpr unsigned size() { return Size; }
  void sync(int site = __builtin_LINE()) const {
    (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, Size, site);
  }
  unsigned ballot(int pred, int site = __builtin_LINE()) const {
    return (unsigned)hipemu::collective(hipemu::OP_BALLOT, pred ? 1 : 0, 0, Size, site);
  }
  template <typename T>
  T shfl(T v, int src, int site = __builtin_LINE()) const {
    return detail::from_bits<T>(
        hipemu::collective(hipemu::OP_SHFL, detail::to_bits(v), src, Size, site));
  }
  template <typename T>
  T shfl_xor(T v, int mask, int site = __builtin_LINE()) const {
    return detail::from_bits<T>(
        hipemu::collective(hipemu::OP_SHFL_XOR, detail::to_bits(v), mask, Size, site));
  }
};
template <int Size>
static inline thread_block_tile<Size> tiled_partition(const thread_block&) {
  return thread_block_tile<Size>{};
}
}  // namespace cooperative_groups
