// vmm_probe.hip -- what growing a row store costs on this box: hipMalloc / hipFree of a fresh array
// against hipMemCreate + hipMemMap + hipMemSetAccess of an increment inside one reserved address
// range (the dynamic table's growth, det.hip).  hipcc --offload-arch=gfx950 -O2 -o tools/vmm_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("FAIL %s -> %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);      \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void touch(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = v;
}
__global__ void sum(const float* p, size_t n, size_t stride, double* out) {
  double a = 0;
  for (size_t i = threadIdx.x; i * stride < n; i += blockDim.x) a += p[i * stride];
  atomicAdd(out, a);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t only_mb = argc > 1 ? (size_t)atol(argv[1]) : 0;  // one chunk size only
  const size_t align_chunk = argc > 2 ? (size_t)atol(argv[2]) : 0;  // reserve with alignment = chunk
  const bool skip_malloc = argc > 3;
  int dev = 0;
  CK(hipSetDevice(dev));
  int vmm = 0;
  CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  printf("virtual memory management supported: %d\n", vmm);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: minimum %zu recommended %zu\n", gmin, grec);
  // plain allocation
  for (size_t gb : {1, 4, 16, 32}) {
    if (skip_malloc) break;
    void* p = nullptr;
    double t0 = now();
    CK(hipMalloc(&p, gb << 30));
    double t1 = now();
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)p, (gb << 30) / 4, 1.f);
    CK(hipDeviceSynchronize());
    double t2 = now();
    CK(hipFree(p));
    double t3 = now();
    printf("hipMalloc %2zu GiB: malloc %.1f ms, first touch %.1f ms, free %.1f ms\n", gb,
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  // reserved ranges grown by equal chunks (the chunk size is the experiment)
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  double* d_out = nullptr;
  CK(hipMalloc(&d_out, 8));
  for (size_t chunk_mb : {2, 64, 256, 1024, 4096}) {
    if (only_mb && chunk_mb != only_mb) continue;
    const size_t chunk = chunk_mb << 20;
    const size_t va = 1ull << 40;  // 1 TiB
    hipDeviceptr_t base = nullptr;
    double t0 = now();
    if (hipMemAddressReserve(&base, va, align_chunk ? chunk : 0, nullptr, 0) != hipSuccess) {
      printf("chunk %zu MiB: reserve failed\n", chunk_mb);
      (void)hipGetLastError();
      continue;
    }
    printf("chunk %4zu MiB: reserve 1 TiB (alignment = chunk) %.2f ms -> %p\n", chunk_mb, (now() - t0) * 1e3, base);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    size_t off = 0;
    const int n = chunk_mb >= 1024 ? 8 : 16;
    double tc = 0, tm = 0, ta = 0, tt = 0;
    bool ok = true;
    for (int i = 0; i < n && ok; i++) {
      hipMemGenericAllocationHandle_t h;
      double a = now();
      hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
      double b = now();
      if (e == hipSuccess) e = hipMemMap((char*)base + off, chunk, 0, h, 0);
      double c = now();
      if (e == hipSuccess) e = hipMemSetAccess((char*)base + off, chunk, &acc, 1);
      double d = now();
      if (e != hipSuccess) {
        printf("  step %d at offset %zu MiB failed: %s\n", i, off >> 20, hipGetErrorString(e));
        (void)hipGetLastError();
        ok = false;
        break;
      }
      hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)((char*)base + off), chunk / 4, 2.f);
      CK(hipDeviceSynchronize());
      double f = now();
      tc += b - a; tm += c - b; ta += d - c; tt += f - d;
      hs.push_back(h);
      off += chunk;
    }
    if (!hs.empty())
      printf("  %zu chunks: per chunk create %.3f ms, map %.3f ms, set_access %.3f ms, first touch %.3f ms\n",
             hs.size(), tc / hs.size() * 1e3, tm / hs.size() * 1e3, ta / hs.size() * 1e3, tt / hs.size() * 1e3);
    if (ok) {
      CK(hipMemset(d_out, 0, 8));
      hipLaunchKernelGGL(sum, dim3(1), dim3(1024), 0, 0, (const float*)base, off / 4, (size_t)1 << 18, d_out);
      double hsum = 0;
      CK(hipMemcpy(&hsum, d_out, 8, hipMemcpyDeviceToHost));
      printf("  strided sum over %zu MiB mapped: %.1f (expect %.1f)\n", off >> 20, hsum,
             2.0 * (double)((off / 4 + (1 << 18) - 1) >> 18));
      // map one more chunk while a kernel writes the mapped range
      hipStream_t s;
      CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (float*)base, (off / 4), 3.f);
      hipMemGenericAllocationHandle_t h;
      double a = now();
      CK(hipMemCreate(&h, chunk, &prop, 0));
      CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
      CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1));
      double b = now();
      CK(hipStreamSynchronize(s));
      printf("  one more chunk while a kernel writes the mapped range: %.2f ms (kernel ok)\n", (b - a) * 1e3);
      hs.push_back(h);
      off += chunk;
      CK(hipMemsetAsync(base, 0, 1 << 20, 0));
      CK(hipMemcpyAsync((char*)base + (1 << 20), base, 1 << 20, hipMemcpyDeviceToDevice, 0));
      CK(hipDeviceSynchronize());
      printf("  memset / memcpy on the mapped range: ok\n");
    }
    double u0 = now();
    for (size_t i = 0; i < hs.size(); i++) {
      CK(hipMemUnmap((char*)base + i * chunk, chunk));
      CK(hipMemRelease(hs[i]));
    }
    CK(hipMemAddressFree(base, va));
    printf("  unmap + release %zu MiB + free range: %.1f ms\n", off >> 20, (now() - u0) * 1e3);
  }
  if (only_mb && only_mb != 9999) return 0;
  auto fill_check = [&](float* p, size_t bytes, float v, const char* what) -> bool {
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, p, bytes / 4, v);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", what); return false; }
    (void)hipMemset(d_out, 0, 8);
    hipLaunchKernelGGL(sum, dim3(1), dim3(1024), 0, 0, (const float*)p, bytes / 4, (size_t)1 << 14, d_out);
    double hs = 0;
    (void)hipMemcpy(&hs, d_out, 8, hipMemcpyDeviceToHost);
    const double want = (double)v * (double)((bytes / 4 + (1 << 14) - 1) >> 14);
    printf("%s: sum %.1f expect %.1f %s\n", what, hs, want, hs == want ? "ok" : "WRONG");
    return hs == want;
  };
  const int test = argc > 4 ? atoi(argv[4]) : 0;
  if (test == 1) {
    // ONE reservation: chunks of different sizes, then unmap the tail and map it again with new handles
    const size_t va = 1ull << 38;
    hipDeviceptr_t base = nullptr;
    CK(hipMemAddressReserve(&base, va, 0, nullptr, 0));
    size_t off = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    std::vector<size_t> offs, szs;
    for (size_t mb : {64, 64, 128, 256, 512, 1024, 2048, 64, 2, 4096}) {
      const size_t bytes = mb << 20;
      hipMemGenericAllocationHandle_t h;
      hipError_t e = hipMemCreate(&h, bytes, &prop, 0);
      if (e == hipSuccess) e = hipMemMap((char*)base + off, bytes, 0, h, 0);
      if (e == hipSuccess) e = hipMemSetAccess((char*)base + off, bytes, &acc, 1);
      printf("mixed sizes: %5zu MiB at offset %5zu MiB: %s\n", mb, off >> 20, hipGetErrorString(e));
      if (e != hipSuccess) { (void)hipGetLastError(); break; }
      hs.push_back(h); offs.push_back(off); szs.push_back(bytes);
      off += bytes;
      char w[64]; snprintf(w, sizeof w, "  whole range after +%zu MiB", mb);
      if (!fill_check((float*)base, off, (float)hs.size(), w)) break;
    }
    for (int round = 0; round < 3; round++) {
      // unmap + release the last 3 pieces, map fresh ones
      for (int k = 0; k < 3; k++) {
        const size_t i = hs.size() - 1 - k;
        CK(hipMemUnmap((char*)base + offs[i], szs[i]));
        CK(hipMemRelease(hs[i]));
      }
      for (int k = 2; k >= 0; k--) {
        const size_t i = hs.size() - 1 - k;
        CK(hipMemCreate(&hs[i], szs[i], &prop, 0));
        CK(hipMemMap((char*)base + offs[i], szs[i], 0, hs[i], 0));
        CK(hipMemSetAccess((char*)base + offs[i], szs[i], &acc, 1));
      }
      char w[64]; snprintf(w, sizeof w, "  remap round %d", round);
      fill_check((float*)base, off, 100.f + round, w);
    }
  }
  if (test == 2) {
    // reserve / map / free cycles: does a freed-and-reserved-again range work?
    for (int round = 0; round < 6; round++) {
      const size_t va = 1ull << 36, chunk = 512ull << 20;
      hipDeviceptr_t base = nullptr;
      CK(hipMemAddressReserve(&base, va, 0, nullptr, 0));
      hipMemGenericAllocationHandle_t h[4];
      for (int i = 0; i < 4; i++) {
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char*)base + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
      }
      char w[64]; snprintf(w, sizeof w, "  reserve cycle %d at %p", round, base);
      fill_check((float*)base, 4 * chunk, 1.f + round, w);
      for (int i = 0; i < 4; i++) {
        CK(hipMemUnmap((char*)base + i * chunk, chunk));
        CK(hipMemRelease(h[i]));
      }
      CK(hipMemAddressFree(base, va));
    }
  }
  if (test == 3) {
    // what does physical memory cost when the device is nearly full / has been churned:
    // fill 200 GiB with hipMalloc, free half of it, then create + map chunks
    std::vector<void*> blocks;
    for (int i = 0; i < 25; i++) {
      void* p = nullptr;
      if (hipMalloc(&p, 8ull << 30) != hipSuccess) { (void)hipGetLastError(); break; }
      hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)p, (8ull << 30) / 4, 1.f);
      blocks.push_back(p);
    }
    CK(hipDeviceSynchronize());
    printf("filled %zu x 8 GiB\n", blocks.size());
    double f0 = now();
    for (size_t i = 0; i < blocks.size(); i += 2) CK(hipFree(blocks[i]));
    printf("freed every other block: %.1f ms\n", (now() - f0) * 1e3);
    for (size_t gb : {8, 8, 8, 16, 16}) {
      void* p = nullptr;
      double t0 = now();
      hipError_t e = hipMalloc(&p, gb << 30);
      double t1 = now();
      printf("hipMalloc %zu GiB on the churned device: %s %.1f ms\n", gb, hipGetErrorString(e), (t1 - t0) * 1e3);
      if (e != hipSuccess) (void)hipGetLastError();
    }
    const size_t va = 1ull << 40, chunk = 1ull << 30;
    hipDeviceptr_t base = nullptr;
    CK(hipMemAddressReserve(&base, va, 0, nullptr, 0));
    double tc = 0, tm = 0;
    int n = 0;
    for (; n < 40; n++) {
      hipMemGenericAllocationHandle_t h;
      double a = now();
      hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
      double b = now();
      if (e == hipSuccess) e = hipMemMap((char*)base + n * chunk, chunk, 0, h, 0);
      if (e == hipSuccess) e = hipMemSetAccess((char*)base + n * chunk, chunk, &acc, 1);
      double c = now();
      if (e != hipSuccess) { printf("chunk %d: %s\n", n, hipGetErrorString(e)); (void)hipGetLastError(); break; }
      tc += b - a; tm += c - b;
    }
    printf("%d x 1 GiB chunks on the churned device: create %.3f ms, map + access %.3f ms per chunk\n", n,
           tc / n * 1e3, tm / n * 1e3);
    fill_check((float*)base, (size_t)n * chunk, 5.f, "  churned-device range");
  }
  return 0;
}
