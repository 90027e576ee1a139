// What does one all-to-all exchange between G workgroups (different CUs) cost?  The building block of decoding ONE code
// block on several CUs: every pass each workgroup publishes 20 KB of partial column sums, all meet, each reads the others'.
//   hipcc --offload-arch=gfx950 -O3 -o xcu_exchange.bin xcu_exchange.hip && ./xcu_exchange.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITEMS = 2496; // 26 columns x 96 four-lane groups, 8 bytes each

__global__ void __launch_bounds__(1024) exch(unsigned long long *buf, unsigned int *counter, unsigned long long *out, int G, int iters, int stride8,
                                             long long *ticks)
{
  // group / part of this workgroup: parts of a group either neighbours in blockIdx (spread over the XCDs) or 8 apart (same XCD)
  int grp, part;
  if (stride8) { const int b = blockIdx.x; grp = (b / (8 * G)) * 8 + (b & 7); part = (b / 8) % G; }
  else { grp = blockIdx.x / G; part = blockIdx.x % G; }
  unsigned long long *gb = buf + (size_t)grp * 2 * G * ITEMS;
  unsigned int *cnt = counter + grp * 32;
  unsigned long long acc = 0;
  __shared__ unsigned int seen;
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    unsigned long long *mine = gb + ((it & 1) * G + part) * ITEMS;
    for (int i = threadIdx.x; i < ITEMS; i += blockDim.x)
      __hip_atomic_store(mine + i, (unsigned long long)(it * 131 + i + part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = (unsigned int)G * (unsigned int)(it + 1);
      while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
        __builtin_amdgcn_s_sleep(1);
      seen = target;
    }
    __syncthreads();
    for (int p = 0; p < G; p++) {
      if (p == part)
        continue;
      const unsigned long long *theirs = gb + ((it & 1) * G + p) * ITEMS;
      for (int i = threadIdx.x; i < ITEMS; i += blockDim.x)
        acc += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const long long t1 = wall_clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc + seen;
  if (threadIdx.x == 0)
    ticks[blockIdx.x] = t1 - t0;
}

int main()
{
  const int maxg = 64, iters = 2000;
  unsigned long long *buf, *out; unsigned int *cnt; long long *ticks;
  CK(hipMalloc(&buf, (size_t)maxg * 2 * 8 * ITEMS * 8));
  CK(hipMalloc(&cnt, maxg * 32 * 4));
  CK(hipMalloc(&out, (size_t)maxg * 8 * 1024 * 8));
  CK(hipMalloc(&ticks, maxg * 8 * 8));
  for (int stride8 = 0; stride8 <= 1; stride8++)
    for (int G : {1, 2, 4, 8})
      for (int groups : {1, 8, 32}) {
        if (stride8 && (groups % 8))
          continue;
        CK(hipMemset(cnt, 0, maxg * 32 * 4));
        const int nwg = G * groups;
        hipLaunchKernelGGL(exch, dim3(nwg), dim3(1024), 0, 0, buf, cnt, out, G, iters, stride8, ticks);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(nwg);
        CK(hipMemcpy(h.data(), ticks, nwg * 8, hipMemcpyDeviceToHost));
        long long mx = 0;
        for (long long v : h) mx = v > mx ? v : mx;
        printf("G=%d parts, %2d groups, parts %s: %.2f us per exchange (20 KB published, %d x 20 KB read per part)\n", G, groups,
               stride8 ? "8 apart in blockIdx (same XCD)" : "adjacent in blockIdx (different XCDs)", mx / 100.0 / iters, G - 1);
      }
  return 0;
}
