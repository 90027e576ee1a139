// Micro-benchmark: issue rate (cycles per wave64 instruction per SIMD) of the VALU opcodes the LDPC decoder
// kernels are built from, each pinned with inline asm so the compiler cannot fold the chain.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_ACC 16
#define ITERS 1024
#define OPK(NAME, ASMSTR)                                                                              \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t *out, uint32_t seed)                        \
  {                                                                                                    \
    uint32_t acc[N_ACC];                                                                               \
    for (int i = 0; i < N_ACC; i++) acc[i] = seed * (i + 1) + threadIdx.x;                             \
    for (int it = 0; it < ITERS; it++) {                                                               \
      _Pragma("unroll") for (int i = 0; i < N_ACC; i++)                                                \
        asm volatile(ASMSTR : "=v"(acc[i]) : "v"(acc[i]), "v"(acc[(i + 5) % N_ACC]));                 \
    }                                                                                                  \
    uint32_t r = 0;                                                                                    \
    for (int i = 0; i < N_ACC; i++) r ^= acc[i];                                                       \
    if (r == 0x12345678u) out[0] = r;                                                                  \
  }
OPK(fma_f32, "v_fma_f32 %0, %1, %2, %2")
OPK(xor_b32, "v_xor_b32 %0, %1, %2")
OPK(add_u32, "v_add_u32 %0, %1, %2")
OPK(sub_u32, "v_sub_u32 %0, %1, %2")
OPK(and_b32, "v_and_b32 %0, %1, %2")
OPK(min_i32, "v_min_i32 %0, %1, %2")
OPK(max_i32, "v_max_i32 %0, %1, %2")
OPK(min_u32, "v_min_u32 %0, %1, %2")
OPK(med3_i32, "v_med3_i32 %0, %1, %2, 63")
OPK(min3_u32, "v_min3_u32 %0, %1, %2, %2")
OPK(lshrrev, "v_lshrrev_b32 %0, 7, %1")
OPK(ashrrev, "v_ashrrev_i32 %0, 7, %1")
OPK(bfe_i32, "v_bfe_i32 %0, %1, 8, 8")
OPK(bfi_b32, "v_bfi_b32 %0, %1, %2, %1")
OPK(and_or, "v_and_or_b32 %0, %1, %2, %1")
OPK(lshl_or, "v_lshl_or_b32 %0, %1, 8, %2")
OPK(cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
OPK(pk_add_u16, "v_pk_add_u16 %0, %1, %2")
OPK(pk_sub_i16, "v_pk_sub_i16 %0, %1, %2")
OPK(pk_min_i16, "v_pk_min_i16 %0, %1, %2")
OPK(pk_max_i16, "v_pk_max_i16 %0, %1, %2")
OPK(pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]")
OPK(pk_lshl_b16, "v_pk_lshlrev_b16 %0, 1, %1 op_sel_hi:[0,1]")
OPK(pk_mul_lo, "v_pk_mul_lo_u16 %0, %1, %2")
OPK(pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %1")
OPK(perm_b32, "v_perm_b32 %0, %1, %2, %2")
OPK(alignbyte, "v_alignbyte_b32 %0, %1, %2, 1")
OPK(alignbit, "v_alignbit_b32 %0, %1, %2, 8")
OPK(sad_u8, "v_sad_u8 %0, %1, %2, 0")
OPK(sub_sdwa, "v_sub_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
OPK(max_i16, "v_max_i16 %0, %1, %2")
OPK(sub_i16, "v_sub_u16 %0, %1, %2")
OPK(mul_u24, "v_mul_u32_u24 %0, %1, %2")
OPK(mul_lo_u32, "v_mul_lo_u32 %0, %1, %2")
OPK(mul_hi_u32, "v_mul_hi_u32 %0, %1, %2")
OPK(dot4_i8, "v_dot4_i32_i8 %0, %1, %2, %1")
OPK(mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
typedef void (*kern_t)(uint32_t *, uint32_t);
static double fma_ns = 0;
void run(const char *name, kern_t kf, uint32_t *d, int blocks_per_cu)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, d, 3u);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, 0, d, 3u + r);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double wi_per_simd = (double)grid * 4 / (256.0 * 4) * ITERS * N_ACC;
  double ns = ms * 1e6 / wi_per_simd;
  if (fma_ns == 0) fma_ns = ns;
  printf("%-14s blocks/CU=%d  %.3f ms  %.3f ns/wave-instr/SIMD  = %.2f x v_fma_f32\n", name, blocks_per_cu, ms, ns, ns / fma_ns);
}
#define RUN(NAME) run(#NAME, k_##NAME, d, bpc)
int main()
{
  uint32_t *d; (void)hipMalloc(&d, 64);
  for (int bpc : {8}) {
    fma_ns = 0;
    RUN(fma_f32); RUN(xor_b32); RUN(add_u32); RUN(sub_u32); RUN(and_b32); RUN(min_i32); RUN(max_i32); RUN(min_u32); RUN(med3_i32); RUN(min3_u32);
    RUN(lshrrev); RUN(ashrrev); RUN(bfe_i32); RUN(bfi_b32); RUN(and_or); RUN(lshl_or); RUN(cndmask);
    RUN(pk_add_u16); RUN(pk_sub_i16); RUN(pk_min_i16); RUN(pk_max_i16); RUN(pk_ashr_i16); RUN(pk_lshl_b16); RUN(pk_mul_lo); RUN(pk_mad_i16);
    RUN(perm_b32); RUN(alignbyte); RUN(alignbit); RUN(sad_u8); RUN(sub_sdwa); RUN(max_i16); RUN(sub_i16); RUN(mul_u24); RUN(mul_lo_u32); RUN(mul_hi_u32); RUN(dot4_i8); RUN(mov_dpp);
  }
  return 0;
}
