// vmm_probe.hip -- what growing a row store costs on this box: hipMalloc / hipFree of a fresh array
// against hipMemCreate + hipMemMap + hipMemSetAccess of an increment inside one reserved address
// range (the dynamic table's growth, det.hip).  hipcc --offload-arch=gfx950 -O2 -o tools/vmm_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
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

int main() {
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
  // reserved range, grown in steps
  const size_t va = 1ull << 40;  // 1 TiB
  hipDeviceptr_t base = nullptr;
  double t0 = now();
  CK(hipMemAddressReserve(&base, va, 0, nullptr, 0));
  printf("reserve 1 TiB: %.2f ms -> %p\n", (now() - t0) * 1e3, base);
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  std::vector<size_t> sizes;
  size_t off = 0;
  double* d_out = nullptr;
  CK(hipMalloc(&d_out, 8));
  for (size_t mb : {2, 64, 256, 1024, 4096, 16384, 32768, 256, 2}) {
    const size_t bytes = mb << 20;
    hipMemGenericAllocationHandle_t h;
    double a = now();
    CK(hipMemCreate(&h, bytes, &prop, 0));
    double b = now();
    CK(hipMemMap((char*)base + off, bytes, 0, h, 0));
    double c = now();
    CK(hipMemSetAccess((char*)base + off, bytes, &acc, 1));
    double d = now();
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (float*)((char*)base + off), bytes / 4, 2.f);
    CK(hipDeviceSynchronize());
    double e = now();
    printf("grow by %6zu MiB at offset %7zu MiB: create %.2f ms, map %.2f ms, set_access %.2f ms, first touch %.2f ms\n",
           mb, off >> 20, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3);
    hs.push_back(h);
    sizes.push_back(bytes);
    off += bytes;
  }
  // the whole mapped range is one flat array for a kernel
  CK(hipMemset(d_out, 0, 8));
  hipLaunchKernelGGL(sum, dim3(1), dim3(1024), 0, 0, (const float*)base, off / 4, (size_t)1 << 18, d_out);
  double hsum = 0;
  CK(hipMemcpy(&hsum, d_out, 8, hipMemcpyDeviceToHost));
  printf("strided sum over %zu MiB mapped: %.1f (expect %.1f)\n", off >> 20, hsum,
         2.0 * (double)((off / 4 + (1 << 18) - 1) >> 18));
  // a kernel running on the mapped range WHILE another piece is mapped behind it
  {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, s, (float*)base, (off / 4), 3.f);
    hipMemGenericAllocationHandle_t h;
    double a = now();
    CK(hipMemCreate(&h, 1ull << 30, &prop, 0));
    CK(hipMemMap((char*)base + off, 1ull << 30, 0, h, 0));
    CK(hipMemSetAccess((char*)base + off, 1ull << 30, &acc, 1));
    double b = now();
    CK(hipStreamSynchronize(s));
    printf("map 1 GiB while a kernel writes the mapped range: %.2f ms (kernel ok)\n", (b - a) * 1e3);
    hs.push_back(h);
    sizes.push_back(1ull << 30);
    off += 1ull << 30;
  }
  // hipMemcpyAsync / hipMemsetAsync on mapped memory
  CK(hipMemsetAsync(base, 0, 1 << 20, 0));
  CK(hipMemcpyAsync((char*)base + (1 << 20), base, 1 << 20, hipMemcpyDeviceToDevice, 0));
  CK(hipDeviceSynchronize());
  printf("memset / memcpy on the mapped range: ok\n");
  double u0 = now();
  size_t o2 = 0;
  for (size_t i = 0; i < hs.size(); i++) {
    CK(hipMemUnmap((char*)base + o2, sizes[i]));
    CK(hipMemRelease(hs[i]));
    o2 += sizes[i];
  }
  CK(hipMemAddressFree(base, va));
  printf("unmap + release %zu MiB + free range: %.1f ms\n", off >> 20, (now() - u0) * 1e3);
  return 0;
}
