// Does v_pk_maximum3_f16 order the decoder's magnitude keys the way v_pk_min_u16 does?  Keys are 0x8000 + m, m = 0..255
// (negative f16 denormals: the LARGER f16 is the SMALLER m), the cap 0x807f and the neutral element 0xfbff (-65504).
// Checks every triple of a key set against the integer minimum and times both instructions (issue rate per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 pk_max3_f16.hip -o pk_max3_f16.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void check(const uint16_t *keys, int n, uint32_t *bad)
{
  const int i = blockIdx.x, j = threadIdx.x;
  if (j >= n) return;
  for (int k = 0; k < n; k++) {
    const uint32_t a = keys[i] | ((uint32_t)keys[(i + 3) % n] << 16), b = keys[j] | ((uint32_t)keys[(j + 5) % n] << 16),
                   c = keys[k] | ((uint32_t)keys[(k + 7) % n] << 16);
    uint32_t r3, r2;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r3) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r2) : "v"(a), "v"(b));
    auto mn = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
    const uint32_t e3 = mn(mn(a & 0xffff, b & 0xffff), c & 0xffff) | (mn(mn(a >> 16, b >> 16), c >> 16) << 16);
    const uint32_t e2 = mn(a & 0xffff, b & 0xffff) | (mn(a >> 16, b >> 16) << 16);
    if (r3 != e3) atomicAdd(&bad[0], 1u);
    if (r2 != e2) atomicAdd(&bad[1], 1u);
  }
}
#define N_ACC 16
#define ITERS 1024
#define OPK(NAME, ASMSTR)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t seed)                        \
  {                                                                                                    \
    uint32_t acc[N_ACC];                                                                               \
    for (int i = 0; i < N_ACC; i++) acc[i] = 0x80008000u | ((seed * (i + 1) + threadIdx.x) & 0x00ff00ffu); \
    for (int it = 0; it < ITERS; it++) {                                                               \
      _Pragma("unroll") for (int i = 0; i < N_ACC; i++)                                                \
        asm volatile(ASMSTR : "=v"(acc[i]) : "v"(acc[i]), "v"(acc[(i + 5) % N_ACC]), "v"(acc[(i + 9) % N_ACC])); \
    }                                                                                                  \
    uint32_t r = 0;                                                                                    \
    for (int i = 0; i < N_ACC; i++) r ^= acc[i];                                                       \
    if (r == 0x12345678u) out[0] = r;                                                                  \
  }
OPK(pk_min_u16, "v_pk_min_u16 %0, %1, %2")
OPK(pk_max_f16, "v_pk_max_f16 %0, %1, %2")
OPK(pk_maximum3_f16, "v_pk_maximum3_f16 %0, %1, %2, %3")
OPK(min3_u16, "v_min3_u16 %0, %1, %2, %3")
OPK(xor_b32, "v_xor_b32 %0, %1, %2")
typedef void (*kern_t)(uint32_t *, uint32_t);
static void run(const char *name, kern_t kf, uint32_t *d)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;
  hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, d, 3u);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, d, 3u + r);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double wi_per_simd = (double)grid * 4 / (256.0 * 4) * ITERS * N_ACC;
  printf("%-18s %.3f ms  %.3f ns/wave-instr/SIMD\n", name, ms, ms * 1e6 / wi_per_simd);
}
int main()
{
  std::vector<uint16_t> keys;
  for (int m = 0; m < 256; m += (m < 8 || m > 120) ? 1 : 7) keys.push_back((uint16_t)(0x8000 + m));
  keys.push_back(0x807f); keys.push_back(0xfbff); keys.push_back(0x8100); keys.push_back(0x81ff);
  const int n = (int)keys.size();
  uint16_t *dk; uint32_t *bad;
  (void)hipMalloc(&dk, n * 2); (void)hipMalloc(&bad, 64);
  (void)hipMemcpy(dk, keys.data(), n * 2, hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 64);
  hipLaunchKernelGGL(check, dim3(n), dim3(256), 0, 0, dk, n, bad);
  uint32_t hb[2]; (void)hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
  printf("keys %d: v_pk_maximum3_f16 mismatches vs integer min %u, v_pk_max_f16 mismatches %u (of %d triples)\n", n, hb[0], hb[1], n * n * n);
  run("v_xor_b32", k_xor_b32, bad); run("v_pk_min_u16", k_pk_min_u16, bad); run("v_pk_max_f16", k_pk_max_f16, bad);
  run("v_pk_maximum3_f16", k_pk_maximum3_f16, bad); run("v_min3_u16", k_min3_u16, bad);
  return 0;
}
