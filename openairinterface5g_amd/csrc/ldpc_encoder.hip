/*
 * ldpc_encoder.hip -- NR LDPC encoder kernels for gfx950 (MI355X), one workgroup per code block, the code word
 * resident in LDS: the bit-packed kernel (ldpc_enc_packed_core.h, default) and the byte-per-lane kernel
 * (ldpc_enc_core.h, <= 26 KiB + 1.5 KiB scratch), bit-identical outputs.
 * Replaces LDPCencoder (reference nrLDPC_encoder/ldpc_encoder.c:44-252, ldpc_encoder_optim8segmulti.c:46-213).
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "ldpc_kernels.h"
#include "ldpc_enc_core.h"
#include "ldpc_enc_packed_core.h"

__global__ void __launch_bounds__(1024) ldpc_enc_kernel(const ldpc_enc_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
  typedef const ldpc_enc_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  uint8_t *x = esm;
  uint8_t *lam = esm + ((code->ncols * code->Z + 15) & ~15);
  const uint32_t blk = blockIdx.x;
  const uint8_t *in = a.in + (job ? (size_t)job->in_off : (size_t)blk * a.in_stride);
  uint8_t *out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
  const int Kb = job ? job->Kb : a.Kb;
#pragma unroll
  for (int ph = 0; ph < LDPC_ENC_NUM_PHASES; ph++) {
    ldpc_enc_phase(ph, code, Kb, in, x, lam, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
}

/* Bit-packed variant (ldpc_enc_packed_core.h): 32 lanes per XOR, <= 8 KiB of LDS per block, several blocks per CU. */
__global__ void __launch_bounds__(512) ldpc_enc_packed_kernel(const ldpc_enc_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
  typedef const ldpc_enc_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  ldpc_encp_lds L;
  ldpc_encp_carve(reinterpret_cast<uint32_t *>(esm), code, L);
  const uint32_t blk = blockIdx.x;
  const uint8_t *in = a.in + (job ? (size_t)job->in_off : (size_t)blk * a.in_stride);
  uint8_t *out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
  const int Kb = job ? job->Kb : a.Kb;
  for (int ph = 0; ph <= 3; ph++) {
    ldpc_encp_phase(ph, code, Kb, in, L, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
  if (threadIdx.x < 64) /* phases 4 .. 11: one wave, no workgroup barriers in between */
    ldpc_encp_core_parity_wave(code, L, (int)threadIdx.x);
  __syncthreads();
  for (int ph = 12; ph < LDPC_ENCP_NUM_PHASES; ph++) {
    ldpc_encp_phase(ph, code, Kb, in, L, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
}

/* NRLDPC_HIP_ENC_KERNEL=bytes selects the byte-per-lane kernel (A/B runs, cross-check); default: bit-packed */
static int enc_packed(void)
{
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("NRLDPC_HIP_ENC_KERNEL");
    v = (e && strcmp(e, "bytes") == 0) ? 0 : 1;
  }
  return v;
}

int ldpc_enc_is_packed(void) { return enc_packed(); }

void ldpc_enc_launch_shape(const ldpc_code_desc_t &hc, int *n_threads, int *lds_bytes)
{
  if (enc_packed()) {
    *n_threads = ldpc_encp_threads(hc.nrows, hc.Z);
    *lds_bytes = 4 * ldpc_encp_lds_words(hc.ncols, hc.kb_full, hc.Z, hc.nrows, hc.nedges);
  } else {
    int waves = (hc.Z + 63) / 64 * 2;
    *n_threads = (waves > 16 ? 16 : waves) * 64;
    *lds_bytes = ((hc.ncols * hc.Z + 15) & ~15) + ((4 * hc.Z + 15) & ~15);
  }
}

hipError_t ldpc_launch_enc_jobs(const ldpc_enc_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (enc_packed())
    hipLaunchKernelGGL(ldpc_enc_packed_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  else
    hipLaunchKernelGGL(ldpc_enc_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_enc(const ldpc_enc_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  int n_threads, lds_bytes;
  ldpc_enc_launch_shape(hc, &n_threads, &lds_bytes);
  return ldpc_launch_enc_jobs(a, n_threads, lds_bytes, n_blocks, stream);
}
