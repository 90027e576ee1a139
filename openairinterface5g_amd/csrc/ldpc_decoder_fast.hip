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

/* a block of a batch launch: addressed by strides (homogeneous batch) or by its job record; everything is read from
 * the kernel arguments / the job record where it is needed (scalar loads), see ldpc_dec_fast_block.h */
typedef const ldpc_dec_job LDPC_CONST_AS *ldpc_job_ptr_t;
template <bool JOBS> struct ldpc_batch_io {
  const ldpc_dec_args &a;
  ldpc_job_ptr_t job; /* nullptr without JOBS: known at compile time, so the selects below fold away */
  __device__ __forceinline__ const uint32_t *src32() const
  {
    return reinterpret_cast<const uint32_t *>(a.llr + (job ? (size_t)job->llr_off : (size_t)blockIdx.x * a.llr_stride));
  }
  __device__ __forceinline__ int8_t *out() const { return a.out + (job ? (size_t)job->out_off : (size_t)blockIdx.x * a.out_stride); }
  __device__ __forceinline__ int max_pass() const { return (job ? job->num_max_iter : a.num_max_iter) + 1; }
  __device__ __forceinline__ int use_crc() const { return a.use_crc; }
  __device__ __forceinline__ int crcE() const { return job ? job->E : a.E; }
  __device__ __forceinline__ const uint32_t *crc_pow() const { return job ? a.crc_pow_tbl[job->crc_type] : a.crc_pow; }
  __device__ __forceinline__ int out_mode() const { return a.out_mode; }
  __device__ __forceinline__ int *tb_abort() const
  {
    return (job && a.tb_abort && job->abort_idx >= 0) ? a.tb_abort + job->abort_idx : nullptr;
  }
  __device__ __forceinline__ uint32_t *stamps() const { return nullptr; }
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  __device__ __forceinline__ bool eager_check() const { return false; }
  __device__ __forceinline__ const uint32_t *src32_prologue() const { return src32(); }
  __device__ __forceinline__ uint32_t *stage_core() const { return nullptr; }
};

/* JOBS = heterogeneous batch (one job record per workgroup, optional transport-block abort flags); the homogeneous
 * variant carries none of that through its loops.  One register budget for every workgroup size -- 128 VGPRs, i.e. 16
 * waves per CU: the throughput shapes put k workgroups of w <= 16 / k waves on a CU and count on all 16 wave slots
 * (a variant compiled for <= 768 threads may take 129+ VGPRs and silently drop the CU to 12 waves: measured 180 -> 262 us
 * on the 1664-segment slot). */
template <bool JOBS>
__global__ void __launch_bounds__(1024) ldpc_dec_fast_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  const ldpc_job_ptr_t job = JOBS ? (ldpc_job_ptr_t)a.jobs + blockIdx.x : (ldpc_job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  const ldpc_batch_io<JOBS> io{a, job};
  const int n_iter = ldpc_dec_fast_block(fsm, code, io);
  if (threadIdx.x == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blockIdx.x] = n_iter;
}

hipError_t ldpc_fast_kernel_init(void)
{
  const void *k[2] = {reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false>), reinterpret_cast<const void *>(ldpc_dec_fast_kernel<true>)};
  for (int i = 0; i < 2; i++) {
    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

hipError_t ldpc_launch_dec_fast(const ldpc_dec_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (a.jobs)
    return ldpc_launch_dec_fast_jobs(a, hc.f_n_threads, hc.f_lds_total, n_blocks, stream);
  hipLaunchKernelGGL((ldpc_dec_fast_kernel<false>), dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  hipLaunchKernelGGL((ldpc_dec_fast_kernel<true>), dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}
