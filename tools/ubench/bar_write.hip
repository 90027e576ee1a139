/*
 * bar_write.hip -- can the host write straight into device memory (large BAR), and how fast?  The resident server's
 * request path would then need no PCIe READ at all: the host pushes header + LLRs into VRAM with posted writes, the
 * workgroup polls and reads local memory.
 *   hipcc --offload-arch=gfx950 -O2 tools/ubench/bar_write.hip -o bar_write && ./bar_write
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void sum_kernel(const uint32_t *p, size_t n, uint32_t *out)
{
  uint32_t s = 0;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}
__global__ void pingpong_dev(volatile uint32_t *bell_dev, volatile uint32_t *done_host, int rounds, long long max_ticks)
{
  const long long t0 = wall_clock64();
  uint32_t last = 0;
  for (int r = 0; r < rounds; r++) {
    uint32_t d;
    for (;;) {
      d = __hip_atomic_load((uint32_t *)bell_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (d != last) break;
      if ((long long)wall_clock64() - t0 > max_ticks) return;
    }
    last = d;
    __hip_atomic_store((uint32_t *)done_host, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main()
{
  int large_bar = -1;
  CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
  printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
  if (large_bar != 1) { printf("no large BAR: host cannot address device memory\n"); return 0; }
  const size_t bytes = 1 << 20;
  for (int kind = 0; kind < 2; kind++) {
    uint8_t *d = nullptr;
    hipError_t e = kind == 0 ? hipExtMallocWithFlags((void **)&d, bytes, hipDeviceMallocFinegrained) : hipMalloc((void **)&d, bytes);
    printf("%s: %s, pointer %p\n", kind == 0 ? "hipExtMallocWithFlags(finegrained)" : "hipMalloc", hipGetErrorString(e), (void *)d);
    if (e != hipSuccess) continue;
    CK(hipMemset(d, 0, bytes));
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> src(bytes);
    for (size_t i = 0; i < bytes; i++) src[i] = (uint8_t)(i * 7 + 1);
    printf("  host writes into it ...\n"); fflush(stdout);
    for (size_t chunk : {(size_t)64, (size_t)26112, bytes}) {
      double best = 1e9;
      for (int rep = 0; rep < 20; rep++) {
        const double t0 = now();
        memcpy(d, src.data(), chunk);
        __sync_synchronize();
        best = std::min(best, now() - t0);
      }
      printf("  host memcpy of %zu bytes into device memory: %.2f us (%.2f GB/s)\n", chunk, best * 1e6, chunk / best / 1e9);
    }
    uint32_t *out; CK(hipMalloc((void **)&out, 4)); CK(hipMemset(out, 0, 4));
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, out);
    uint32_t got = 0, want = 0; CK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < bytes / 4; i++) want += ((const uint32_t *)src.data())[i];
    printf("  GPU sees the host's writes: %s\n", got == want ? "yes" : "NO");
    /* mailbox: bell in device memory (host writes over the BAR), answer in host memory */
    uint32_t *done; CK(hipHostMalloc((void **)&done, 64, hipHostMallocCoherent | hipHostMallocMapped)); *done = 0;
    volatile uint32_t *bell = (volatile uint32_t *)d; *bell = 0; __sync_synchronize();
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int rounds = 20000;
    hipLaunchKernelGGL(pingpong_dev, dim3(1), dim3(64), 0, s, bell, (volatile uint32_t *)done, rounds, 100ll * 1000 * 1000 * 5);
    std::vector<double> lat; const double t0 = now();
    for (int r = 1; r <= rounds; r++) {
      const double a = now();
      *bell = (uint32_t)r; __sync_synchronize();
      while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (uint32_t)r) if (now() - t0 > 20.0) { printf("  timeout at %d\n", r); return 1; }
      lat.push_back(now() - a);
    }
    CK(hipStreamSynchronize(s));
    std::sort(lat.begin(), lat.end());
    printf("  mailbox round trip, bell in device memory / answer in host memory: median %.2f us, p99 %.2f\n", lat[lat.size() / 2] * 1e6, lat[lat.size() * 99 / 100] * 1e6);
  }
  return 0;
}
