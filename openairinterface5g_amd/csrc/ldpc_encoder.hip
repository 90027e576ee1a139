/*
 * ldpc_encoder.hip -- NR LDPC encoder kernel for gfx950 (MI355X), one workgroup per code block; the
 * whole code word stays in LDS (<= 26 KiB + 1.5 KiB scratch).  Phases: ldpc_enc_core.h.
 * Replaces LDPCencoder (reference nrLDPC_encoder/ldpc_encoder.c:44-252, ldpc_encoder_optim8segmulti.c:46-213).
 */
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_enc_core.h"

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

hipError_t ldpc_launch_enc(const ldpc_enc_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  const int lds = ((hc.ncols * hc.Z + 15) & ~15) + ((4 * hc.Z + 15) & ~15);
  int waves = (hc.Z + 63) / 64 * 2;
  if (waves > 16) waves = 16;
  hipLaunchKernelGGL(ldpc_enc_kernel, dim3(n_blocks), dim3(waves * 64), lds, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_enc_jobs(const ldpc_enc_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL(ldpc_enc_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}
