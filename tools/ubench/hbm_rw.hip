/*
 * hbm_rw.hip -- what HBM delivers to simple streaming kernels on this box (the ceiling the de-matching kernel is held
 * against): write-only (plain / non-temporal), read-only, copy; 256 MiB buffers, 16-byte accesses.
 *   hipcc --offload-arch=gfx950 -O2 tools/ubench/hbm_rw.hip -o hbm_rw && ./hbm_rw
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void wr(u32x4 *d, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (u32x4){1, 2, 3, 4}; }
__global__ void wr_nt(u32x4 *d, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store((u32x4){1, 2, 3, 4}, d + i); }
__global__ void rd(const u32x4 *s, u32x4 *sink, size_t n) { u32x4 a = {0, 0, 0, 0}; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += s[i]; if (a.x == 0x12345678) sink[0] = a; }
__global__ void cp(const u32x4 *s, u32x4 *d, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i]; }
int main()
{
  const size_t bytes = 256u << 20, n = bytes / 16;
  u32x4 *a, *b;
  CK(hipMalloc((void **)&a, bytes)); CK(hipMalloc((void **)&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wgs : {2048, 8192, 32768}) {
    for (int k = 0; k < 4; k++) {
      float best = 1e9;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        if (k == 0) hipLaunchKernelGGL(wr, dim3(wgs), dim3(256), 0, 0, a, n);
        if (k == 1) hipLaunchKernelGGL(wr_nt, dim3(wgs), dim3(256), 0, 0, a, n);
        if (k == 2) hipLaunchKernelGGL(rd, dim3(wgs), dim3(256), 0, 0, a, b, n);
        if (k == 3) hipLaunchKernelGGL(cp, dim3(wgs), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const char *nm[4] = {"write", "write (non-temporal)", "read", "copy (read + write)"};
      printf("%-22s %6d workgroups: %7.1f GB/s\n", nm[k], wgs, (k == 3 ? 2.0 : 1.0) * bytes / best / 1e6);
    }
  }
  return 0;
}
