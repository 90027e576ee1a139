/*
 * h2d_link.hip -- what the host link of this box delivers (VERDICT r01 item 3 / DESIGN "PCIe-inclusive rate"):
 *   1. hipMemcpyAsync from page-locked memory, 1 / 2 / 4 streams, host->device and device->host;
 *   2. a kernel reading page-locked host memory directly (16-byte loads, N workgroups) into device memory;
 *   3. mailbox round trip: host rings a word in host memory, a resident workgroup sees it and writes a word back.
 *   hipcc --offload-arch=gfx950 -O2 tools/ubench/h2d_link.hip -o h2d_link && ./h2d_link
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void pull_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void pingpong_kernel(volatile uint32_t *bell, volatile uint32_t *done, int rounds, long long max_ticks)
{
  const long long t0 = wall_clock64();
  uint32_t last = 0;
  for (int r = 0; r < rounds; r++) {
    uint32_t d;
    for (;;) {
      d = __hip_atomic_load((uint32_t *)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (d != last)
        break;
      if ((long long)wall_clock64() - t0 > max_ticks)
        return; /* safety: never stays resident */
    }
    last = d;
    __hip_atomic_store((uint32_t *)done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main()
{
  const size_t bytes = 64u << 20;
  uint8_t *h, *d, *d2;
  CK(hipHostMalloc((void **)&h, bytes, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipMalloc((void **)&d, bytes));
  CK(hipMalloc((void **)&d2, bytes));
  memset(h, 1, bytes);
  hipStream_t s[4];
  for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
  for (int dir = 0; dir < 2; dir++)
    for (int ns : {1, 2, 4}) {
      const size_t part = bytes / ns;
      double best = 1e9;
      for (int rep = 0; rep < 6; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < ns; i++) {
          if (dir == 0) CK(hipMemcpyAsync(d + i * part, h + i * part, part, hipMemcpyHostToDevice, s[i]));
          else CK(hipMemcpyAsync(h + i * part, d + i * part, part, hipMemcpyDeviceToHost, s[i]));
        }
        for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(s[i]));
        best = std::min(best, now() - t0);
      }
      printf("hipMemcpyAsync %s pinned 64 MiB, %d stream(s): %.1f GB/s\n", dir == 0 ? "H2D" : "D2H", ns, bytes / best / 1e9);
    }
  for (size_t sz : {(size_t)1 << 20, (size_t)4 << 20, (size_t)27 << 20}) {
    double best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s[0]));
      CK(hipStreamSynchronize(s[0]));
      best = std::min(best, now() - t0);
    }
    printf("hipMemcpyAsync H2D pinned %zu MiB, 1 stream: %.1f GB/s (%.1f us)\n", sz >> 20, sz / best / 1e9, best * 1e6);
  }
  for (int wgs : {64, 256, 1024, 4096}) {
    double best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      hipLaunchKernelGGL(pull_kernel, dim3(wgs), dim3(256), 0, s[0], (const uint4 *)h, (uint4 *)d2, bytes / 16);
      CK(hipStreamSynchronize(s[0]));
      best = std::min(best, now() - t0);
    }
    printf("kernel pull from pinned host memory, %d workgroups x 256: %.1f GB/s\n", wgs, bytes / best / 1e9);
  }
  {
    uint32_t *flags;
    CK(hipHostMalloc((void **)&flags, 256, hipHostMallocCoherent | hipHostMallocMapped));
    memset(flags, 0, 256);
    volatile uint32_t *bell = flags, *done = flags + 32;
    const int rounds = 20000;
    hipLaunchKernelGGL(pingpong_kernel, dim3(1), dim3(64), 0, s[0], bell, done, rounds, 100ll * 1000 * 1000 * 5);
    std::vector<double> lat;
    const double t0 = now();
    for (int r = 1; r <= rounds; r++) {
      const double a = now();
      __atomic_store_n((uint32_t *)bell, (uint32_t)r, __ATOMIC_SEQ_CST);
      while (__atomic_load_n((uint32_t *)done, __ATOMIC_ACQUIRE) != (uint32_t)r)
        if (now() - t0 > 20.0) { printf("pingpong: timeout at round %d\n", r); exit(1); }
      lat.push_back(now() - a);
    }
    CK(hipStreamSynchronize(s[0]));
    std::sort(lat.begin(), lat.end());
    printf("mailbox round trip (host rings -> resident wave sees it -> host sees the answer): median %.2f us, p10 %.2f, p99 %.2f\n",
           lat[lat.size() / 2] * 1e6, lat[lat.size() / 10] * 1e6, lat[lat.size() * 99 / 100] * 1e6);
  }
  return 0;
}
