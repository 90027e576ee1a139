/*
 * ldpc_decoder_fast.hip -- "fast" NR LDPC flooding min-sum decoder kernel for gfx950 (MI355X).
 *
 * Same contract as the generic kernel (ldpc_decoder.hip; reference nrLDPC_decoder.c:206-880), one
 * workgroup per code block, everything resident in LDS between the LLR load and the bit store.
 * What differs is the work decomposition (ldpc_dec_fast_core.h): 4 lanes per thread, biased-byte
 * messages moved as dwords, packed 16-bit arithmetic, degree-sorted 64-item tasks (ldpc_graph.c
 * build_fast_section) that the waves of the workgroup draw from a queue.  Requires Zc % 4 == 0 and 4-byte
 * aligned LLR rows; other cases are served by the generic kernel.
 */
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_fast_block.h"

template <int MAX_THREADS>
__global__ void __launch_bounds__(MAX_THREADS) ldpc_dec_fast_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  typedef const ldpc_dec_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  const uint32_t blk = blockIdx.x;
  ldpc_block_io io;
  io.src32 = reinterpret_cast<const uint32_t *>(a.llr + (job ? (size_t)job->llr_off : (size_t)blk * a.llr_stride));
  io.out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
  io.max_pass = (job ? job->num_max_iter : a.num_max_iter) + 1;
  io.use_crc = a.use_crc;
  io.crcE = job ? job->E : a.E;
  io.crc_pow = job ? a.crc_pow_tbl[job->crc_type] : a.crc_pow;
  io.out_mode = a.out_mode;
  io.tb_abort = (job && a.tb_abort && job->abort_idx >= 0) ? a.tb_abort + job->abort_idx : nullptr;
  const int n_iter = ldpc_dec_fast_block(fsm, code, io);
  if (threadIdx.x == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blk] = n_iter;
}

hipError_t ldpc_fast_kernel_init(void)
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_dec_fast_kernel<1024>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess)
    return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_dec_fast_kernel<768>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t ldpc_launch_dec_fast(const ldpc_dec_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  /* up to 12 waves: 168 VGPRs per lane available (no spills in the degree-19 rows); 13..16 waves: 128 */
  if (hc.f_n_threads <= 768)
    hipLaunchKernelGGL(ldpc_dec_fast_kernel<768>, dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  else
    hipLaunchKernelGGL(ldpc_dec_fast_kernel<1024>, dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (n_threads <= 768)
    hipLaunchKernelGGL(ldpc_dec_fast_kernel<768>, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  else
    hipLaunchKernelGGL(ldpc_dec_fast_kernel<1024>, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}
