// cross_gemm.hip -- the GEMMs of MultiCrossLayer v2 (DCN v2) as this library's own MFMA kernel, with
// the elementwise work the reference fuses into its GEMM epilogues.
//
// Reference: MultiCrossForwardFunctorv2 / MultiCrossBackwardFunctorv2
// (R/HugeCTR/src/layers/multi_cross_layer.cu:582-700, 732-812): per layer  P = X_l U,
// H = P V + b (bias in the second GEMM's epilogue, :640-700),  X_{l+1} = X_0 .* H + X_l
// (fused_mul_fma3, :391-424); backward S1 = S0 V^T and dY_{l-1} = S1 U^T + dY_l (the residual in
// the GEMM's epilogue, :760-812).  The reference searches cuBLASLt algorithms per shape; the shapes
// are skinny (K = 3456 -> N = 512 and K = 512 -> N = 3456 at the MLPerf DCNv2 size), which a
// library's 256 x 256 tiles fill badly (64 tiles for B = 8192 on 256 CUs: measured 0.16 of the
// matrix peak, round 4).
//
// One kernel, "NT" form:  C[M][N] = A[M][K] . Bt[N][K]^T  (+ epilogue), 16-bit operands (fp16 or
// bf16), fp32 accumulation on v_mfma_f32_32x32x16_{f16,bf16}.  Both operands are K-contiguous, so
// every MFMA fragment is one 16-byte LDS read; the forward's weights arrive transposed (they are
// converted to the 16-bit type once per step anyway), the backward's are used as they lie.
//   * workgroup = 256 threads = 4 wavefronts as 2 x 2, tile BM x 128 (BM = 128 or 64), K step 64;
//     a wavefront owns (BM / 2) x 64 of C: (BM / 64) x 2 MFMA tiles of 32 x 32;
//   * staging: global_load_lds 16 bytes per lane (no VGPR round trip) into a ring of STAGES LDS
//     buffers (dynamic LDS: 3 x 32 KB at BM = 128), one barrier per K step: tiles k + 1 ..
//     k + STAGES - 1 stream in while tile k is multiplied -- with two buffers a K step cannot be
//     shorter than a trip to L2 / HBM (measured, X1 shape: 545 TFLOP/s at K = 3456), so the wait
//     in front of the barrier is a COUNTED one (vmcnt = the loads of the tiles still allowed in
//     flight) and the barrier a raw s_barrier that does not drain the queue.  The LDS image of a
//     tile is row-major [rows][64] 16-bit = 128 bytes a row, 16-byte chunk c of row r stored at
//     chunk position c ^ ((r >> 1) & 7): the 16 rows a quarter-wave reads together fall into 16
//     different bank groups.  The DMA writes lane-linear, so the permutation is applied to the
//     lanes' SOURCE addresses (inside one 128-byte row segment: coalescing is untouched);
//   * epilogue: the accumulators go through LDS (fp32, the K loop's buffers) and leave as
//     row-contiguous 16-byte stores, which is also where bias / X_0 / X_l / the residual are read
//     as 16-byte vectors:
//       EPI_PLAIN     C = (T)acc
//       EPI_CROSS     H = (T)(acc + b[n]);  C = (T)(X_l + X_0 * H)      (H is stored too: backward)
//       EPI_RESIDUAL  C = (T)(acc + R)
//   * block id -> tile: column tiles of one row panel are consecutive on ONE XCD (block b runs on
//     XCD b % 8), so the panel of A is fetched from HBM once per XCD L2.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace hctr {
namespace {

typedef _Float16 cg_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float cg_f32x16 __attribute__((ext_vector_type(16)));
typedef float cg_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cg_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmThreads = 256;
constexpr int kGemmBN = 128;
constexpr int kGemmBK = 64;  // 16-bit elements: 128 bytes per LDS row
constexpr int kEpiPlain = 0, kEpiCross = 1, kEpiResidual = 2;

template <bool BF>
struct Cg16;
template <>
struct Cg16<false> {
  __device__ __forceinline__ static cg_f32x16 mfma(cg_u32x4 a, cg_u32x4 b, cg_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<cg_f16x8*>(&a),
                                                  *reinterpret_cast<cg_f16x8*>(&b), c, 0, 0, 0);
  }
  __device__ __forceinline__ static float to_f32(unsigned short u) {
    _Float16 h = *reinterpret_cast<_Float16*>(&u);
    return (float)h;
  }
  __device__ __forceinline__ static unsigned short from_f32(float v) {
    _Float16 h = (_Float16)v;
    return *reinterpret_cast<unsigned short*>(&h);
  }
};
template <>
struct Cg16<true> {
  __device__ __forceinline__ static cg_f32x16 mfma(cg_u32x4 a, cg_u32x4 b, cg_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<cg_bf16x8*>(&a),
                                                   *reinterpret_cast<cg_bf16x8*>(&b), c, 0, 0, 0);
  }
  __device__ __forceinline__ static float to_f32(unsigned short u) {
    return __uint_as_float((unsigned)u << 16);
  }
  __device__ __forceinline__ static unsigned short from_f32(float v) {
    __bf16 h = (__bf16)v;
    return *reinterpret_cast<unsigned short*>(&h);
  }
};

// 16 bytes per lane from global memory straight into LDS: lane l's bytes land at lds + 16 l
__device__ __forceinline__ void cg_load_lds16(const unsigned short* g, unsigned short* lds_wave_base) {
  HCTR_GLOBAL_LOAD_LDS16(g, lds_wave_base);
}

struct GemmArgs {
  int M, N, K;
  const unsigned short* A;   // [M][lda]
  const unsigned short* Bt;  // [N][ldb]
  unsigned short* C;         // [M][ldc]
  int lda, ldb, ldc;
  const unsigned short* bias;  // EPI_CROSS: [N] (16-bit)
  const unsigned short* X0;    // EPI_CROSS: [M][ldc]
  const unsigned short* XL;    // EPI_CROSS: [M][ldc]; EPI_RESIDUAL: R [M][ldc]
  unsigned short* H;           // EPI_CROSS: [M][ldc]
  int tiles_n;                 // N / 128
  int tiles_m;                 // ceil(M / BM)
};

// rows [row0, row0 + ROWS) x K step k0 of a row-major matrix into an LDS tile image (see the
// header): ROWS / 8 DMA instructions of 8 rows each, spread over the NW wavefronts
template <int ROWS, int NW = 4>
__device__ __forceinline__ void cg_stage(const unsigned short* __restrict__ src, int ld, int row0,
                                         int rows_total, int k0, unsigned short* tile, int wave,
                                         int lane) {
  constexpr int PER_WAVE = ROWS / 8 / NW;  // DMA instructions per wavefront
  static_assert(ROWS % (8 * NW) == 0, "tile rows");
#pragma unroll
  for (int j = 0; j < PER_WAVE; j++) {
    const int r8 = (wave * PER_WAVE + j) * 8;  // first row of this instruction's 8 (tile-local)
    const int r = r8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);  // the chunk this lane's LDS position holds
    int gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;  // (rows past the end: a legal read, never stored)
    cg_load_lds16(src + (size_t)gr * ld + k0 + c * 8, tile + (size_t)r8 * kGemmBK);
  }
}

template <int BM, bool BF, int EPI, int STAGES>
__global__ void __launch_bounds__(kGemmThreads)
    cross_gemm_nt16_kernel(GemmArgs g) {
  using H16 = Cg16<BF>;
  constexpr int MI = BM / 64;           // 32-row MFMA tiles per wavefront along M
  constexpr int A_TILE = BM * kGemmBK;  // 16-bit elements
  constexpr int B_TILE = kGemmBN * kGemmBK;
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int LOADS = (BM + kGemmBN) / 32;  // DMA instructions of one thread per tile
  static_assert(LOADS == 8 || LOADS == 6, "counted waits below");
  // one object for all of it (a second __shared__ array makes hipcc drain the DMA queue before
  // every LDS read): the staging ring, reused by the epilogue as [BM][128] fp32
  HCTR_DYN_LDS16(unsigned char, lds_raw);
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- block -> tile: XCD x takes row panels x, x + 8, ...; a panel's column tiles are
  //      consecutive blocks of that XCD ---------------------------------------------------------
  int tile_m, tile_n;
  {
    const int b = blockIdx.x;
    if ((g.tiles_m & 7) == 0) {
      const int xcd = b & 7, slot = b >> 3;  // slot-th block of this XCD
      tile_m = (slot / g.tiles_n) * 8 + xcd;
      tile_n = slot % g.tiles_n;
    } else {  // (panel count not a multiple of 8: plain row-major order)
      tile_m = b / g.tiles_n;
      tile_n = b % g.tiles_n;
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * kGemmBN;
  const int KT = g.K / kGemmBK;

  cg_f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment addresses inside a tile image: row r, logical 16-byte chunk c -> (r, c ^ ((r>>1)&7))
  const int fr = lane & 31, fk = lane >> 5;
  auto stage_tile = [&](int kt) {
    unsigned short* buf = lds + (kt % STAGES) * STAGE;
    cg_stage<BM>(g.A, g.lda, m0, g.M, kt * kGemmBK, buf, wave, lane);
    cg_stage<kGemmBN>(g.Bt, g.ldb, n0, g.N, kt * kGemmBK, buf + A_TILE, wave, lane);
  };
#pragma unroll
  for (int p = 0; p < STAGES - 1; p++)
    if (p < KT) stage_tile(p);
  for (int kt = 0; kt < KT; kt++) {
    // tile kt has landed once at most (tiles behind it that were issued) x LOADS of this wave's
    // DMA instructions are still out; every wave waits for its own, the barrier makes it everybody's
    const int behind = (KT - 1 - kt) < (STAGES - 2) ? (KT - 1 - kt) : (STAGES - 2);
    if constexpr (STAGES == 2) {
      HCTR_WAIT_VMCNT(0);
    } else if constexpr (LOADS == 8) {
      if (behind >= 2) HCTR_WAIT_VMCNT(16);
      else if (behind == 1) HCTR_WAIT_VMCNT(8);
      else HCTR_WAIT_VMCNT(0);
    } else {
      if (behind >= 2) HCTR_WAIT_VMCNT(12);
      else if (behind == 1) HCTR_WAIT_VMCNT(6);
      else HCTR_WAIT_VMCNT(0);
    }
    HCTR_RAW_BARRIER();  // (also: everybody is done reading the buffer tile kt + STAGES - 1 takes)
    if (kt + STAGES - 1 < KT) stage_tile(kt + STAGES - 1);
    const unsigned short* at = lds + (kt % STAGES) * STAGE;
    const unsigned short* bt = at + A_TILE;
#pragma unroll
    for (int s = 0; s < kGemmBK / 16; s++) {
      cg_u32x4 af[MI], bf[2];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int r = wm * (BM / 2) + i * 32 + fr;
        const int c = (2 * s + fk) ^ ((r >> 1) & 7);
        af[i] = *reinterpret_cast<const cg_u32x4*>(at + r * kGemmBK + c * 8);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int r = wn * 64 + j * 32 + fr;
        const int c = (2 * s + fk) ^ ((r >> 1) & 7);
        bf[j] = *reinterpret_cast<const cg_u32x4*>(bt + r * kGemmBK + c * 8);
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = H16::mfma(af[i], bf[j], acc[i][j]);
    }
  }
  // ---- epilogue: accumulators -> LDS [BM][128] fp32 -> row-contiguous 16-byte stores ------------
  // What the epilogue reads from memory (X_0, X_l / the residual, the bias) is requested FIRST, all
  // chunks of this thread at once: one trip that overlaps the accumulators' way through LDS (a
  // load -> wait -> store chain per chunk was measured at 8 dependent trips, 4 x the K loop of a
  // K = 512 product).  Raw barriers: they do not wait for these loads (no DMA is outstanding any
  // more: the last K step waited for all of it).
  constexpr int CHUNKS = BM * kGemmBN / 8;  // 8 columns each
  constexpr int NCH = CHUNKS / kGemmThreads;
  const int cc = tid & 15;                  // (q & 15 of every chunk q = i * 256 + tid)
  const int gn = n0 + cc * 8;
  cg_u32x4 ex0[NCH], exl[NCH], bv;
  if constexpr (EPI != kEpiPlain) {
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      int gm = m0 + ((i * kGemmThreads + tid) >> 4);
      gm = gm < g.M ? gm : g.M - 1;  // (clamped: a legal read; the store below is predicated)
      const size_t off = (size_t)gm * g.ldc + gn;
      exl[i] = *reinterpret_cast<const cg_u32x4*>(g.XL + off);
      if constexpr (EPI == kEpiCross) ex0[i] = *reinterpret_cast<const cg_u32x4*>(g.X0 + off);
    }
    if constexpr (EPI == kEpiCross) bv = *reinterpret_cast<const cg_u32x4*>(g.bias + gn);
  }
  HCTR_RAW_BARRIER();  // (every wave is done reading the last tile)
  float* ct = reinterpret_cast<float*>(lds_raw);
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        const int col = wn * 64 + j * 32 + fr;
        ct[row * kGemmBN + col] = acc[i][j][r];
      }
  HCTR_RAW_BARRIER();
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int row = (i * kGemmThreads + tid) >> 4;
    const int gm = m0 + row;
    const cg_f32x4 v0 = *reinterpret_cast<const cg_f32x4*>(ct + row * kGemmBN + cc * 8);
    const cg_f32x4 v1 = *reinterpret_cast<const cg_f32x4*>(ct + row * kGemmBN + cc * 8 + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    const size_t off = (size_t)gm * g.ldc + gn;
    unsigned short o[8];
    if constexpr (EPI == kEpiPlain) {
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = H16::from_f32(v[e]);
    } else if constexpr (EPI == kEpiResidual) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const unsigned short ru = (unsigned short)(exl[i][e >> 1] >> ((e & 1) * 16));
        o[e] = H16::from_f32(v[e] + H16::to_f32(ru));
      }
    } else {
      unsigned short h[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const unsigned short bu = (unsigned short)(bv[e >> 1] >> ((e & 1) * 16));
        const unsigned short x0u = (unsigned short)(ex0[i][e >> 1] >> ((e & 1) * 16));
        const unsigned short xlu = (unsigned short)(exl[i][e >> 1] >> ((e & 1) * 16));
        h[e] = H16::from_f32(v[e] + H16::to_f32(bu));
        o[e] = H16::from_f32(H16::to_f32(xlu) + H16::to_f32(x0u) * H16::to_f32(h[e]));
      }
      cg_u32x4 hv;
#pragma unroll
      for (int e = 0; e < 4; e++) hv[e] = (uint32_t)h[2 * e] | ((uint32_t)h[2 * e + 1] << 16);
      if (gm < g.M) *reinterpret_cast<cg_u32x4*>(g.H + off) = hv;
    }
    cg_u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; e++) ov[e] = (uint32_t)o[2 * e] | ((uint32_t)o[2 * e + 1] << 16);
    if (gm < g.M) *reinterpret_cast<cg_u32x4*>(g.C + off) = ov;
  }
}

// The same product on 256 x 256 tiles (plain epilogue): 512 threads = 8 wavefronts as 2 x 4, a
// wavefront owns 128 x 64 of C (4 x 2 MFMA tiles: 6 fragment reads per 8 MFMAs instead of 4 per 4),
// two staging buffers of 64 KB.  What it is for: the K = 3456 -> N = 512 products at batch 65536
// are bound by what the L2 can feed the LDS, not by the matrix pipe -- 2048 tiles of 128 x 128 move
// 2048 x 54 x 32 KB = 3.6 GB through the L2 (12 TB/s at the measured 302 us); tiles of 256 x 256
// move half of that.  One workgroup per CU (128 KB of LDS), so it is taken only while there are at
// least as many tiles as CUs.  The accumulators leave through LDS in two passes of 128 rows.
constexpr int kBigBM = 256, kBigBN = 256, kBigThreads = 512;

template <bool BF>
__global__ void __launch_bounds__(kBigThreads)
    cross_gemm_nt16_big_kernel(GemmArgs g) {
  using H16 = Cg16<BF>;
  constexpr int MI = 4, NJ = 2;
  constexpr int A_TILE = kBigBM * kGemmBK, B_TILE = kBigBN * kGemmBK, STAGE = A_TILE + B_TILE;
  HCTR_DYN_LDS16(unsigned char, lds_raw);
  unsigned short* lds = reinterpret_cast<unsigned short*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  int tile_m, tile_n;
  {
    const int b = blockIdx.x;
    if ((g.tiles_m & 7) == 0) {
      const int xcd = b & 7, slot = b >> 3;
      tile_m = (slot / g.tiles_n) * 8 + xcd;
      tile_n = slot % g.tiles_n;
    } else {
      tile_m = b / g.tiles_n;
      tile_n = b % g.tiles_n;
    }
  }
  const int m0 = tile_m * kBigBM, n0 = tile_n * kBigBN;
  const int KT = g.K / kGemmBK;
  cg_f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;
  auto stage_tile = [&](int kt) {
    unsigned short* buf = lds + (kt & 1) * STAGE;
    cg_stage<kBigBM, 8>(g.A, g.lda, m0, g.M, kt * kGemmBK, buf, wave, lane);
    cg_stage<kBigBN, 8>(g.Bt, g.ldb, n0, g.N, kt * kGemmBK, buf + A_TILE, wave, lane);
  };
  stage_tile(0);
  for (int kt = 0; kt < KT; kt++) {
    HCTR_WAIT_VMCNT(0);
    HCTR_RAW_BARRIER();
    if (kt + 1 < KT) stage_tile(kt + 1);
    const unsigned short* at = lds + (kt & 1) * STAGE;
    const unsigned short* bt = at + A_TILE;
#pragma unroll
    for (int s = 0; s < kGemmBK / 16; s++) {
      cg_u32x4 af[MI], bf[NJ];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int r = wm * 128 + i * 32 + fr;
        const int c = (2 * s + fk) ^ ((r >> 1) & 7);
        af[i] = *reinterpret_cast<const cg_u32x4*>(at + r * kGemmBK + c * 8);
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int r = wn * 64 + j * 32 + fr;
        const int c = (2 * s + fk) ^ ((r >> 1) & 7);
        bf[j] = *reinterpret_cast<const cg_u32x4*>(bt + r * kGemmBK + c * 8);
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = H16::mfma(af[i], bf[j], acc[i][j]);
    }
  }
  // epilogue: rows [128 p, 128 p + 128) of the tile through LDS as [128][256] fp32, p = 0, 1 (the
  // wavefronts with wm == p hold them), then row-contiguous 16-byte stores by all 512 threads
  float* ct = reinterpret_cast<float*>(lds_raw);
  constexpr int NCH = 128 * kBigBN / 8 / kBigThreads;  // 8 chunks of 8 columns per thread and pass
  const int cc = tid & 31;
#pragma unroll 1
  for (int p = 0; p < 2; p++) {
    HCTR_RAW_BARRIER();  // (the K loop's / the previous pass's reads of the buffer are done)
    if (wm == p) {
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
            const int col = wn * 64 + j * 32 + fr;
            ct[row * kBigBN + col] = acc[i][j][r];
          }
    }
    HCTR_RAW_BARRIER();
#pragma unroll
    for (int i = 0; i < NCH; i++) {
      const int row = (i * kBigThreads + tid) >> 5;
      const int gm = m0 + p * 128 + row, gn = n0 + cc * 8;
      const cg_f32x4 v0 = *reinterpret_cast<const cg_f32x4*>(ct + row * kBigBN + cc * 8);
      const cg_f32x4 v1 = *reinterpret_cast<const cg_f32x4*>(ct + row * kBigBN + cc * 8 + 4);
      const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      cg_u32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; e++)
        ov[e] = (uint32_t)H16::from_f32(v[2 * e]) | ((uint32_t)H16::from_f32(v[2 * e + 1]) << 16);
      if (gm < g.M) *reinterpret_cast<cg_u32x4*>(g.C + (size_t)gm * g.ldc + gn) = ov;
    }
  }
}

template <bool BF>
int launch_big(const GemmArgs& g, hipStream_t s) {
  constexpr int lds = 2 * (kBigBM + kBigBN) * kGemmBK * 2;  // 128 KB
  // (per device and cheap: asked on every launch rather than remembered per process)
  HCTR_HIP(hipFuncSetAttribute((const void*)cross_gemm_nt16_big_kernel<BF>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((cross_gemm_nt16_big_kernel<BF>), dim3((unsigned)(g.tiles_m * g.tiles_n)),
                     dim3(kBigThreads), lds, s, g);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

// fp32 master weights [batch][rows][cols] -> the 16-bit copy as it lies AND its transpose
// [batch][cols][rows] in one pass (the forward wants both operands K-contiguous, the backward takes
// the weights as they lie; torch's generic transpose was 28 us per tensor at the MLPerf size)
template <bool BF>
__global__ void __launch_bounds__(256)
    convert_transpose16_kernel(int rows, int cols, const float* __restrict__ src,
                               unsigned short* __restrict__ dst, unsigned short* __restrict__ dst_t) {
  using H16 = Cg16<BF>;
  __shared__ unsigned short tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) {
      const unsigned short v = H16::from_f32(src[base + (size_t)r * cols + c]);
      if (dst != nullptr) dst[base + (size_t)r * cols + c] = v;
      tile[ty + 8 * k][tx] = v;
    }
  }
  __syncthreads();
  if (dst_t == nullptr) return;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < rows && c < cols) dst_t[base + (size_t)c * rows + r] = tile[tx][ty + 8 * k];
  }
}

template <int BM, int STAGES>
constexpr int gemm_lds_bytes() {
  constexpr int ring = STAGES * (BM + kGemmBN) * kGemmBK * 2, epi = BM * kGemmBN * 4;
  return ring > epi ? ring : epi;
}

template <int BM, bool BF, int EPI, int STAGES>
int launch_one(const GemmArgs& g, hipStream_t s) {
  constexpr int lds = gemm_lds_bytes<BM, STAGES>();
  if (lds > 65536)  // (above 64 KB the kernel has to be told -- per device, so on every launch)
    HCTR_HIP(hipFuncSetAttribute((const void*)cross_gemm_nt16_kernel<BM, BF, EPI, STAGES>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((cross_gemm_nt16_kernel<BM, BF, EPI, STAGES>),
                     dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(kGemmThreads), lds, s, g);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

template <int BM, bool BF, int STAGES>
int launch_epi(const GemmArgs& g, int epi, hipStream_t s) {
  switch (epi) {
    case kEpiPlain: return launch_one<BM, BF, kEpiPlain, STAGES>(g, s);
    case kEpiCross: return launch_one<BM, BF, kEpiCross, STAGES>(g, s);
    default: return launch_one<BM, BF, kEpiResidual, STAGES>(g, s);
  }
}

template <int BM, bool BF>
int launch_stages(const GemmArgs& g, int epi, int stages, hipStream_t s) {
  if (stages == 4) return launch_epi<BM, BF, 4>(g, epi, s);
  if (stages == 3) return launch_epi<BM, BF, 3>(g, epi, s);
  return launch_epi<BM, BF, 2>(g, epi, s);
}

}  // namespace
}  // namespace hctr

using namespace hctr;

extern "C" {

int hctr_gemm_nt16(size_t m, int n, int k, const void* a, int lda, const void* bt, int ldb, void* c,
                   int ldc, int epilogue, const void* bias, const void* x0, const void* xl,
                   void* h_out, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(dtype == HCTR_EMB_F16 || dtype == HCTR_EMB_BF16, "gemm_nt16: 16-bit types only");
  HCTR_REQUIRE(epilogue >= kEpiPlain && epilogue <= kEpiResidual, "gemm_nt16: epilogue");
  if (m == 0) return HCTR_OK;
  HCTR_REQUIRE(a && bt && c, "null pointer");
  HCTR_REQUIRE(n > 0 && k > 0 && n % kGemmBN == 0 && k % kGemmBK == 0,
               "gemm_nt16: N % 128 == 0 and K % 64 == 0");
  HCTR_REQUIRE(lda >= k && ldb >= k && ldc >= n && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0,
               "gemm_nt16: leading dimensions (multiples of 8 elements)");
  HCTR_REQUIRE(m <= (size_t)0x7FFFFF00, "gemm_nt16: M");
  auto al16 = [](const void* p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  HCTR_REQUIRE(al16(a) && al16(bt) && al16(c) && al16(bias) && al16(x0) && al16(xl) && al16(h_out),
               "gemm_nt16: 16-byte aligned buffers");
  if (epilogue == kEpiCross) HCTR_REQUIRE(bias && x0 && xl && h_out, "gemm_nt16: cross epilogue operands");
  if (epilogue == kEpiResidual) HCTR_REQUIRE(xl, "gemm_nt16: residual operand");
  GemmArgs g;
  g.M = (int)m;
  g.N = n;
  g.K = k;
  g.A = (const unsigned short*)a;
  g.Bt = (const unsigned short*)bt;
  g.C = (unsigned short*)c;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.bias = (const unsigned short*)bias;
  g.X0 = (const unsigned short*)x0;
  g.XL = (const unsigned short*)xl;
  g.H = (unsigned short*)h_out;
  g.tiles_n = n / kGemmBN;
  // 128-row tiles while they give every CU two workgroups, 64-row tiles for the small products
  // (B = 8192 x N = 512: 256 tiles of 128 rows = one per CU, nothing to overlap a barrier with)
  const size_t tiles128 = ceil_div<size_t>(m, 128) * (size_t)g.tiles_n;
  const char* bm_env = getenv("HCTR_GEMM_BM");
  const bool bm64 = bm_env ? atoi(bm_env) == 64 : tiles128 < 512;
  hipStream_t s = as_stream(stream);
  const bool bf = dtype == HCTR_EMB_BF16;
  // plain products with at least a tile of 256 x 256 per CU: half the L2 -> LDS traffic per flop
  // (HCTR_GEMM_BM=256 forces, any other value keeps the 128-wide tiles)
  {
    const size_t tiles256 = ceil_div<size_t>(m, 256) * (size_t)(n / 256);
    const bool big = epilogue == kEpiPlain && n % 256 == 0 &&
                     (bm_env ? atoi(bm_env) == 256 : tiles256 >= 256);
    if (big) {
      g.tiles_n = n / 256;
      g.tiles_m = (int)ceil_div<size_t>(m, 256);
      return bf ? launch_big<true>(g, s) : launch_big<false>(g, s);
    }
  }
  // depth of the staging ring (HCTR_GEMM_STAGES: 2 / 3 / 4, measurements)
  const char* st_env = getenv("HCTR_GEMM_STAGES");
  int stages = st_env ? atoi(st_env) : 2;
  if (stages != 3 && stages != 4) stages = 2;
  if (bm64) {
    g.tiles_m = (int)ceil_div<size_t>(m, 64);
    return bf ? launch_stages<64, true>(g, epilogue, stages, s)
              : launch_stages<64, false>(g, epilogue, stages, s);
  }
  g.tiles_m = (int)ceil_div<size_t>(m, 128);
  if (stages == 4) stages = 3;  // (4 x 32 KB + nothing else: 128 KB of the CU's 160 -- not built)
  return bf ? launch_stages<128, true>(g, epilogue, stages, s)
            : launch_stages<128, false>(g, epilogue, stages, s);
}

int hctr_convert_transpose16(size_t batch, int rows, int cols, const float* src, void* dst,
                             void* dst_t, int dtype, hctr_stream_t stream) {
  HCTR_REQUIRE(dtype == HCTR_EMB_F16 || dtype == HCTR_EMB_BF16, "convert_transpose16: 16-bit types only");
  if (batch == 0 || rows <= 0 || cols <= 0) return HCTR_OK;
  HCTR_REQUIRE(src && (dst || dst_t), "null pointer");
  HCTR_REQUIRE(batch <= 65535, "convert_transpose16: batch");
  const dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32), (unsigned)batch);
  if (dtype == HCTR_EMB_BF16)
    hipLaunchKernelGGL(convert_transpose16_kernel<true>, grid, dim3(256), 0, as_stream(stream), rows,
                       cols, src, (unsigned short*)dst, (unsigned short*)dst_t);
  else
    hipLaunchKernelGGL(convert_transpose16_kernel<false>, grid, dim3(256), 0, as_stream(stream), rows,
                       cols, src, (unsigned short*)dst, (unsigned short*)dst_t);
  HCTR_LAUNCH_CHECK();
  return HCTR_OK;
}

}  // extern "C"
