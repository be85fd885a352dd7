// lds_atomic_probe.hip -- rate of fp32 / u32 atomic adds into LDS on gfx950 as a function of how the
// 64 lanes of a wavefront spread over addresses (the hot rows' accumulators, sparse_update.hip).
// hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_probe tools/lds_atomic_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

// MODE 0: every lane its own word (64 consecutive words: no conflict)
// MODE 1: 16 lanes x float4-strided words, 4 groups on 4 different rows (the kernel's pattern, no sharing)
// MODE 2: same, the 4 groups on the SAME row (a 3-row table)
// MODE 3: all waves of the workgroup on the same row as well
// T = float / unsigned
template <typename T, int MODE, bool ATOMIC>
__global__ void __launch_bounds__(512) probe(int iters, float* out) {
  __shared__ T acc[112 * 64];
  for (int i = threadIdx.x; i < 112 * 64; i += 512) acc[i] = T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, l = lane & 15;
  for (int it = 0; it < iters; it++) {
    int row;
    if (MODE == 0) row = (wave * 8 + (it & 7));
    else if (MODE == 1) row = (wave * 14 + g * 3 + (it % 3));
    else if (MODE == 2) row = wave * 14 + (it % 3);
    else row = it % 3;
    T* a = acc + row * 64 + (MODE == 0 ? lane : l * 4);
#pragma unroll
    for (int k = 0; k < (MODE == 0 ? 1 : 4); k++) {
      if (ATOMIC) atomicAdd(a + k, T(1));
      else a[k] += T(1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)acc[0];
}

template <typename T, int MODE, bool ATOMIC>
void run(const char* name, float* d_out) {
  const int iters = 4096, grid = 512;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((probe<T, MODE, ATOMIC>), dim3(grid), dim3(512), 0, 0, 16, d_out);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((probe<T, MODE, ATOMIC>), dim3(grid), dim3(512), 0, 0, iters, d_out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double lane_ops = (double)grid * 512 * iters * (MODE == 0 ? 1 : 4);
  // 256 CUs at ~2.4 GHz
  printf("%-52s %8.1f us  %7.2f G lane-ops/s  %6.2f lane-ops / clk / CU\n", name, ms * 1e3,
         lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float* d_out;
  hipMalloc(&d_out, 4096);
  run<float, 0, true>("f32 atomic, 64 consecutive words", d_out);
  run<float, 0, false>("f32 plain rmw, 64 consecutive words", d_out);
  run<unsigned, 0, true>("u32 atomic, 64 consecutive words", d_out);
  run<float, 1, true>("f32 atomic, 4 groups x 16 lanes, 4 rows", d_out);
  run<float, 1, false>("f32 plain rmw, 4 groups x 16 lanes, 4 rows", d_out);
  run<unsigned, 1, true>("u32 atomic, 4 groups x 16 lanes, 4 rows", d_out);
  run<float, 2, true>("f32 atomic, 4 groups on ONE row (per wave)", d_out);
  run<unsigned, 2, true>("u32 atomic, 4 groups on ONE row (per wave)", d_out);
  run<unsigned long long, 1, true>("u64 atomic, 4 groups x 16 lanes, 4 rows", d_out);
  run<unsigned long long, 2, true>("u64 atomic, 4 groups on ONE row (per wave)", d_out);
  run<unsigned long long, 3, true>("u64 atomic, the whole workgroup on one row", d_out);
  run<float, 3, true>("f32 atomic, the whole workgroup on one row", d_out);
  run<unsigned, 3, true>("u32 atomic, the whole workgroup on one row", d_out);
  return 0;
}
