// radix_sort.h -- stable LSD radix sort of (u32 key, u32 value) pairs (radix_sort.hip)
#pragma once
#include "common.h"

namespace hctr {

// bytes of workspace for n pairs (two ping-pong arrays + per-tile digit histograms)
size_t radix_sort_temp_bytes(size_t n);
int radix_sort_passes(int end_bit);
// optional source of the FIRST pass: when *flag != 0 (a device word: "the batch is one-hot") key i
// is the low 32 bits of keys64[i] and its payload is i (through the gradient map when map_inner
// != 0: (i % map_inner) * map_outer + i / map_inner) -- kin / vin are not read at all
// skip_below != 0 (with n_kept): when *flag != 0 the keys below skip_below are left out of the sort;
// *n_kept (device) receives the number of keys kept and the result holds that many pairs
// (*flag == 0: nothing is left out, *n_kept = n)
struct RsFirst {
  const uint64_t* keys64;
  const uint32_t* flag;
  uint32_t map_inner, map_outer;
  uint32_t skip_below = 0;
  uint32_t* n_kept = nullptr;
};
// sorts by key bits [0, 10 * ceil(end_bit / 10)), stable; kin / vin are left untouched; kout /
// vout receive the result.  No host synchronisation, no inter-workgroup waits.
int radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout,
                         const uint32_t* vin, uint32_t* vout, size_t n, int end_bit,
                         hipStream_t s, const RsFirst* first = nullptr);

}  // namespace hctr
