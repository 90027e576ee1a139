/*
 * ldpc_decoder.hip -- generic NR LDPC flooding min-sum decoder kernel for gfx950 (MI355X).
 *
 * Replaces nrLDPC_decoder_core (reference openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_decoder.c:206-880)
 * for a whole batch of code blocks: one workgroup per code block, every message of the block resident
 * in LDS for the entire decode (BG1 Zc=384: 121 344 B of check messages + 9 984 B APP + 26 112 B channel
 * LLR = 153.8 KiB of the CU's 160 KiB), HBM touched only to read the LLRs once and to write the packed
 * hard decisions.  Work mapping: a (lifted row, 64-lane chunk) item per wavefront and step in the
 * check-node phase, a (column, chunk) item in the bit-node phase; the base-graph row/column of an item
 * is wave-uniform, so the edge tables are read with scalar loads and the row degree is a scalar
 * branch.  Per-thread arithmetic lives in ldpc_dec_core.h (shared with the CPU emulation test).
 */
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_generic_block.h"

__global__ void __launch_bounds__(1024) ldpc_dec_generic_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  typedef const ldpc_dec_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code); /* never written while a kernel runs */
  const uint32_t blk = blockIdx.x;
  ldpc_gblock_io io;
  io.llr = a.llr + (job ? (size_t)job->llr_off : (size_t)blk * a.llr_stride);
  io.out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
  io.max_pass = (job ? job->num_max_iter : a.num_max_iter) + 1;
  io.use_crc = a.use_crc;
  io.crcE = job ? job->E : a.E;
  io.crc_pow = job ? a.crc_pow_tbl[job->crc_type & 3] : a.crc_pow;
  io.out_mode = a.out_mode;
  io.tb_abort = (job && a.tb_abort && job->abort_idx >= 0) ? a.tb_abort + job->abort_idx : nullptr;
  io.trace = nullptr;
  io.trace_stride = 0;
  const int n_iter = ldpc_dec_generic_block(smem, code, io);
  if (threadIdx.x == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blk] = n_iter;
}

/* CRC stop with the predicate on the host: all passes, the output of every pass >= 3 kept (block b, pass p at
 * trace + (b * n_trace + p - 3) * trace_stride).  A kernel of its own so that the hot kernels' argument block stays as it is. */
__global__ void __launch_bounds__(1024) ldpc_dec_generic_trace_kernel(const ldpc_dec_args a, int8_t *trace, uint32_t trace_stride, uint32_t n_trace)
{
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)a.code;
  const uint32_t blk = blockIdx.x;
  ldpc_gblock_io io;
  io.llr = a.llr + (size_t)blk * a.llr_stride;
  io.out = nullptr;
  io.max_pass = a.num_max_iter + 1;
  io.use_crc = 1;
  io.crcE = 0;
  io.crc_pow = nullptr;
  io.out_mode = a.out_mode;
  io.tb_abort = nullptr;
  io.trace = trace + (size_t)blk * n_trace * trace_stride;
  io.trace_stride = trace_stride;
  const int n_iter = ldpc_dec_generic_block(smem, code, io);
  if (threadIdx.x == 0)
    a.n_iter[blk] = n_iter;
}

hipError_t ldpc_kernels_init(void)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_dec_generic_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess)
    return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_dec_generic_trace_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t ldpc_launch_dec_generic_trace(const ldpc_dec_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, int8_t *trace,
                                         uint32_t trace_stride, uint32_t n_trace, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL(ldpc_dec_generic_trace_kernel, dim3(n_blocks), dim3(hc.n_threads), hc.lds_total, stream, a, trace, trace_stride, n_trace);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_generic(const ldpc_dec_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks,
                                   hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL(ldpc_dec_generic_kernel, dim3(n_blocks), dim3(hc.n_threads), hc.lds_total, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_generic_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL(ldpc_dec_generic_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}
