// nvcuda::wmma stand-in -- TEST INFRASTRUCTURE ONLY (see cuda_runtime_api.h): the 16x16x16 half ->
// float warp matrix operations the reference's dot-interaction kernels use, by their documented
// contract.  A fragment is opaque on the device (its element-to-lane mapping is unspecified), so
// here every lane holds the whole tile: load / mma / store are computed redundantly per lane and
// need no data from other lanes.  They are still warp-SYNCHRONOUS operations -- every lane's loads
// of a tile happen before any lane's later store to (possibly the same) shared memory -- so each
// one is a wavefront barrier of the interpreter first.  Products of two binary16 values are exact
// in binary32; the sums are fp32 in k order (the hardware's order is unspecified: compare with a
// tolerance of one output ulp).
#pragma once
#include "cuda_device_extras.h"

static inline void __syncwarp(unsigned = 0xffffffffu, int site = __builtin_LINE()) {
  (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 32, site);
}

namespace nvcuda {
namespace wmma {
struct matrix_a {};
struct matrix_b {};
struct accumulator {};
struct row_major {};
struct col_major {};
enum layout_t { mem_row_major, mem_col_major };

template <typename Use, int M, int N, int K, typename T, typename Layout = void>
struct fragment {
  static_assert(M == 16 && N == 16 && K == 16, "16x16x16 tiles");
  static constexpr int num_elements = 256;
  T x[256];  // [row][col] of the logical tile (A: m x k, B: k x n, accumulator: m x n)
};

template <typename Use, typename T, typename Layout, typename V>
static inline void fill_fragment(fragment<Use, 16, 16, 16, T, Layout>& f, V v) {
  for (int i = 0; i < 256; i++) f.x[i] = (T)v;
}
static inline void load_matrix_sync(fragment<matrix_a, 16, 16, 16, __half, row_major>& f,
                                    const __half* p, unsigned ldm, int site = __builtin_LINE()) {
  (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 32, site);
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < 16; c++) f.x[r * 16 + c] = p[r * ldm + c];
}
static inline void load_matrix_sync(fragment<matrix_b, 16, 16, 16, __half, col_major>& f,
                                    const __half* p, unsigned ldm, int site = __builtin_LINE()) {
  (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 32, site);
  for (int k = 0; k < 16; k++)
    for (int n = 0; n < 16; n++) f.x[k * 16 + n] = p[n * ldm + k];
}
static inline void load_matrix_sync(fragment<matrix_b, 16, 16, 16, __half, row_major>& f,
                                    const __half* p, unsigned ldm, int site = __builtin_LINE()) {
  (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 32, site);
  for (int k = 0; k < 16; k++)
    for (int n = 0; n < 16; n++) f.x[k * 16 + n] = p[k * ldm + n];
}
template <typename LA, typename LB>
static inline void mma_sync(fragment<accumulator, 16, 16, 16, float>& d,
                            const fragment<matrix_a, 16, 16, 16, __half, LA>& a,
                            const fragment<matrix_b, 16, 16, 16, __half, LB>& b,
                            const fragment<accumulator, 16, 16, 16, float>& c) {
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float acc = c.x[i * 16 + j];
      for (int k = 0; k < 16; k++)
        acc += __half2float(a.x[i * 16 + k]) * __half2float(b.x[k * 16 + j]);
      d.x[i * 16 + j] = acc;
    }
}
static inline void store_matrix_sync(float* p, const fragment<accumulator, 16, 16, 16, float>& f,
                                     unsigned ldm, layout_t layout, int site = __builtin_LINE()) {
  (void)hipemu::collective(hipemu::OP_WAVE_BARRIER, 0, 0, 32, site);
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++)
      p[layout == mem_row_major ? i * ldm + j : j * ldm + i] = f.x[i * 16 + j];
}
}  // namespace wmma
}  // namespace nvcuda
