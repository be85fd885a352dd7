// membench.hip -- what the MI355X memory system gives for the access patterns of the embedding
// path: sequential read, sequential copy, and random-row reads of R-byte rows (read-only and
// gather-copy).  Build: hipcc --offload-arch=gfx950 -O3 membench.hip -o membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void seq_read(const float4* __restrict__ in, size_t n4, float* out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = in[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
__global__ void seq_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// LPR lanes per row (row = LPR*16 bytes); each group handles U rows per iteration
template <int LPR, int U, bool COPY>
__global__ void __launch_bounds__(256) rand_rows(const float4* __restrict__ in, const unsigned* __restrict__ idx,
                                                 size_t nrows, float4* __restrict__ out, float* sink) {
  constexpr int GPB = 256 / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t r0 = ((size_t)blockIdx.x * GPB + g) * U; r0 < nrows; r0 += (size_t)gridDim.x * GPB * U) {
    unsigned id[U];
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) id[k] = (r0 + k < nrows) ? idx[r0 + k] : 0;
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = in[(size_t)id[k] * LPR + l];
#pragma unroll
    for (int k = 0; k < U; k++) {
      if (COPY) { if (r0 + k < nrows) out[(r0 + k) * LPR + l] = v[k]; }
      else { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
  }
  if (!COPY && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = 1.f;
}

template <typename F>
double time_ms(F f, int iters = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

template <int LPR>
void run_rand(const float4* in, float4* out, float* sink, size_t bytes, unsigned* d_idx, std::vector<unsigned>& h_idx, std::mt19937& rng, int grid) {
  const size_t nrows = bytes / (LPR * 16);
  h_idx.resize(nrows);
  for (size_t i = 0; i < nrows; i++) h_idx[i] = (unsigned)i;
  std::shuffle(h_idx.begin(), h_idx.end(), rng);
  CK(hipMemcpy(d_idx, h_idx.data(), nrows * 4, hipMemcpyHostToDevice));
  double t1 = time_ms([&] { hipLaunchKernelGGL((rand_rows<LPR, 4, false>), dim3(grid), dim3(256), 0, 0, in, d_idx, nrows, out, sink); });
  double t2 = time_ms([&] { hipLaunchKernelGGL((rand_rows<LPR, 8, false>), dim3(grid), dim3(256), 0, 0, in, d_idx, nrows, out, sink); });
  double t3 = time_ms([&] { hipLaunchKernelGGL((rand_rows<LPR, 4, true>), dim3(grid), dim3(256), 0, 0, in, d_idx, nrows, out, sink); });
  printf("rand rows %5d B: read-only U4 %7.1f GB/s  U8 %7.1f GB/s | gather-copy U4 %7.1f GB/s (r+w)\n", LPR * 16,
         bytes / t1 / 1e6, bytes / t2 / 1e6, 2.0 * bytes / t3 / 1e6);
}

int main(int argc, char** argv) {
  size_t bytes = (argc > 1 ? atof(argv[1]) : 0.872) * 1e9;
  bytes = bytes / 4096 * 4096;
  int grid = argc > 2 ? atoi(argv[2]) : 2048;
  float4 *in, *out; float* sink; unsigned* d_idx;
  CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&d_idx, bytes / 16 * 4 / 4 + 64));
  CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
  const size_t n4 = bytes / 16;
  printf("buffer %.3f GB grid %d\n", bytes / 1e9, grid);
  double t = time_ms([&] { hipLaunchKernelGGL(seq_read, dim3(grid), dim3(256), 0, 0, in, n4, sink); });
  printf("seq read        : %7.1f GB/s\n", bytes / t / 1e6);
  t = time_ms([&] { hipLaunchKernelGGL(seq_copy, dim3(grid), dim3(256), 0, 0, in, out, n4); });
  printf("seq copy (r+w)  : %7.1f GB/s\n", 2.0 * bytes / t / 1e6);
  std::vector<unsigned> h_idx; std::mt19937 rng(1);
  run_rand<4>(in, out, sink, bytes, d_idx, h_idx, rng, grid);
  run_rand<8>(in, out, sink, bytes, d_idx, h_idx, rng, grid);
  run_rand<16>(in, out, sink, bytes, d_idx, h_idx, rng, grid);
  run_rand<32>(in, out, sink, bytes, d_idx, h_idx, rng, grid);
  run_rand<64>(in, out, sink, bytes, d_idx, h_idx, rng, grid);
  return 0;
}
