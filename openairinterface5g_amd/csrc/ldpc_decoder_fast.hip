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
#include <cstdlib>
#include "ldpc_kernels.h"
#include "ldpc_dec_fast_block.h"
#include "ldpc_dec_fast_mblock.h"

/* a block of a batch launch: addressed by strides (homogeneous batch) or by its job record; everything is read from
 * the kernel arguments / the job record where it is needed (scalar loads), see ldpc_dec_fast_block.h */
/* CRC = the launch is known to stop on the CRC (the transport-block chain's job launches): the parity of the hard decisions,
 * which only the parity-check stop looks at, is then not computed at all -- an XOR per edge and a dozen ops per item of the
 * check-node bodies fall away as dead code */
template <bool JOBS, bool CRC = false, bool TRACE = false> struct ldpc_batch_io {
  const ldpc_dec_args &a;
  ldpc_job_ptr_t job; /* nullptr without JOBS: known at compile time, so the selects below fold away */
  __device__ __forceinline__ const uint32_t *src32() const
  {
    return reinterpret_cast<const uint32_t *>(a.llr + (job ? (size_t)job->llr_off : (size_t)blockIdx.x * a.llr_stride));
  }
  __device__ __forceinline__ int8_t *out() const { return a.out + (job ? (size_t)job->out_off : (size_t)blockIdx.x * a.out_stride); }
  __device__ __forceinline__ int max_pass() const { return (job ? job->num_max_iter : a.num_max_iter) + 1; }
  /* (the launchers pick the instantiation from a.use_crc: CRC = false never meets a CRC-stop call, and the CRC check is not
   * compiled into it -- its registers were what pushed the parity-check instantiation into scratch memory, and a kernel that
   * uses ANY scratch pays 15-20 us per launch on this stack: profiles/r06/ab_scratch_probe.txt) */
  __device__ __forceinline__ int use_crc() const { return CRC ? 1 : 0; }
  static constexpr bool syndrome = !CRC;
  __device__ __forceinline__ int crcE() const { return job ? job->E : a.E; }
  __device__ __forceinline__ const uint32_t *crc_pow() const { return job ? a.crc_pow_tbl[job->crc_type & 3] : a.crc_pow; }
  __device__ __forceinline__ int out_mode() const { return a.out_mode; }
  __device__ __forceinline__ int *tb_abort() const
  {
    return (job && a.tb_abort && job->abort_idx >= 0) ? a.tb_abort + job->abort_idx : nullptr;
  }
  /* (TRACE: the diagnostic instantiation, NRLDPC_HIP_DEC_TRACE -- the stamps go straight into the workgroup's trace row) */
  __device__ __forceinline__ uint32_t *stamps() const
  {
    return TRACE ? reinterpret_cast<uint32_t *>(a.trace + (size_t)blockIdx.x * 32 + 5) : nullptr;
  }
  static constexpr bool pass_stamps = TRACE;
  __device__ __forceinline__ int fair_turns() const { return a.fair; }
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  /* homogeneous launches: the parity check of a pass right after it when the block is close to converging (a block that
   * stops saves the next pass' check-node phase; one that does not converge never gets close and pays nothing) */
  __device__ __forceinline__ bool eager_check() const { return !JOBS; }
  static constexpr bool mute_items = false; /* (the fused segment kernel's retransmission launches only: tb_rx_fused.hip) */
  __device__ __forceinline__ bool mute_check() const { return false; }
  static constexpr bool bn_tickets = true; /* short bit-node tasks come several to a ticket (ldpc_graph.h f_bn_ticket) */
  static constexpr bool tb_epilogue = false; /* (the chain's fused segment kernel has an IO of its own: tb_rx_fused.hip) */
  __device__ __forceinline__ bool tables_resident() const { return false; }
  __device__ __forceinline__ uint32_t out_tag() const { return 0u; }
  __device__ __forceinline__ uint32_t abort_load() const { return 0u; }
  __device__ __forceinline__ bool abort_is(uint32_t) const { return false; }
  __device__ __forceinline__ bool has_abort() const { return JOBS; }
  __device__ __forceinline__ void put16(uint4 *p, uint32_t x, uint32_t y, uint32_t z, uint32_t t) const { *p = make_uint4(x, y, z, t); }
  __device__ __forceinline__ uint32_t ld_llr(const uint32_t *p) const { return *p; }
  __device__ __forceinline__ const uint32_t *src32_prologue() const { return src32(); }
  __device__ __forceinline__ uint32_t *stage_core() const { return nullptr; }
};

/* JOBS = heterogeneous batch (one job record per workgroup, optional transport-block abort flags); the homogeneous
 * variant carries none of that through its loops.  One register budget for every workgroup size -- 128 VGPRs, i.e. 16
 * waves per CU: the throughput shapes put k workgroups of w <= 16 / k waves on a CU and count on all 16 wave slots
 * (a variant compiled for <= 768 threads may take 129+ VGPRs and silently drop the CU to 12 waves: measured 180 -> 262 us
 * on the 1664-segment slot). */
template <bool JOBS, bool CRC = false, bool TRACE = false, int ZC = 0>
__global__ void __launch_bounds__(1024) ldpc_dec_fast_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  const ldpc_job_ptr_t job = JOBS ? (ldpc_job_ptr_t)a.jobs + blockIdx.x : (ldpc_job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  const ldpc_batch_io<JOBS, CRC, TRACE> io{a, job};
  unsigned long long *tr = TRACE ? a.trace + (size_t)blockIdx.x * 32 : nullptr;
  if (TRACE && threadIdx.x == 0) {
    tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_REG_HW_ID: wave, simd, cu, sh, se ... */
    tr[1] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  /* HW_REG_XCC_ID */
    tr[2] = wall_clock64();
  }
  const int n_iter = ldpc_dec_fast_block<ldpc_batch_io<JOBS, CRC, TRACE>, ZC>(fsm, code, io);
  if (threadIdx.x == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blockIdx.x] = n_iter;
  if (TRACE) {
    __syncthreads(); /* (the output row's stores are on their way) */
    if (threadIdx.x == 0) {
      tr[3] = wall_clock64();
      tr[4] = (unsigned long long)n_iter;
    }
  }
}

/* Host-buffer batches: the block's LLRs sit in page-locked host memory.  The workgroup pulls its row over the link into
 * its row of the device staging buffer (16-byte loads, all of a 1024-thread workgroup's 26 KB in flight at once) and then
 * runs the ordinary block body on that copy, which stays L2-resident for the per-pass re-reads.  With a launch's
 * workgroups at different points of their life (pulling / decoding) the link and the CUs are busy at the same time, with
 * no copy-engine -> kernel dependency anywhere (measured: a chunked hipMemcpyAsync pipeline loses ~40 us per such edge;
 * profiles/r02/README.md). */
template <bool CRC>
__global__ void __launch_bounds__(1024) ldpc_dec_fast_pull_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)a.code;
  const int nb = code->num_llr, tid = (int)threadIdx.x, nt = (int)blockDim.x; /* Zc % 4 == 0: a multiple of 4 bytes */
  const uint8_t *src = reinterpret_cast<const uint8_t *>(a.pull) + (size_t)blockIdx.x * a.pull_stride;
  uint8_t *dst = reinterpret_cast<uint8_t *>(const_cast<int8_t *>(a.llr)) + (size_t)blockIdx.x * a.llr_stride;
  /* The launch's first workgroups all start together and would pull together -- the link busy, no CU decoding, then the
   * reverse.  64 workgroups saturate the link, so the first round's workgroups start in groups of 64, pull_stagger_ticks
   * (10 ns each) apart: the first group decodes while the next one pulls, and the later rounds inherit the stagger. */
  if (a.pull_stagger_ticks && blockIdx.x < a.pull_first_round) {
    const long long until = (long long)wall_clock64() + (long long)(blockIdx.x >> 6) * (long long)a.pull_stagger_ticks;
    while ((long long)wall_clock64() < until)
      __builtin_amdgcn_s_sleep(32);
  }
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const uint4 *s16 = reinterpret_cast<const uint4 *>(src);
    uint4 *d16 = reinterpret_cast<uint4 *>(dst);
    const int n16 = nb >> 4;
    for (int i = tid; i < n16; i += 4 * nt) { /* four loads in flight per thread; out-of-range ones re-read the last unit */
      const int last = n16 - 1;
      const int i1 = i + nt, i2 = i + 2 * nt, i3 = i + 3 * nt;
      const uint4 v0 = s16[i], v1 = s16[i1 < last ? i1 : last], v2 = s16[i2 < last ? i2 : last], v3 = s16[i3 < last ? i3 : last];
      d16[i] = v0;
      if (i1 < n16) d16[i1] = v1;
      if (i2 < n16) d16[i2] = v2;
      if (i3 < n16) d16[i3] = v3;
    }
    const int rest = (nb & 15) >> 2;
    if (tid < rest)
      reinterpret_cast<uint32_t *>(dst)[4 * n16 + tid] = reinterpret_cast<const uint32_t *>(src)[4 * n16 + tid];
  } else {
    const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
    const int n4 = nb >> 2;
    for (int i = tid; i < n4; i += 4 * nt) {
      const int last = n4 - 1;
      const int i1 = i + nt, i2 = i + 2 * nt, i3 = i + 3 * nt;
      const uint32_t v0 = s4[i], v1 = s4[i1 < last ? i1 : last], v2 = s4[i2 < last ? i2 : last], v3 = s4[i3 < last ? i3 : last];
      d4[i] = v0;
      if (i1 < n4) d4[i1] = v1;
      if (i2 < n4) d4[i2] = v2;
      if (i3 < n4) d4[i3] = v3;
    }
  }
  /* The row is read back by other waves of this workgroup only: workgroup scope is enough (the CU's vector cache is
   * write-through and shared by the workgroup's waves).  An agent-scope fence here writes back and invalidates the
   * XCD's whole L2 -- per workgroup: measured 89 us of L2 misses in the prologue that follows
   * (profiles/r02/host_pull_timeline_*.txt). */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const ldpc_batch_io<false, CRC> io{a, (ldpc_job_ptr_t) nullptr};
  const int n_iter = ldpc_dec_fast_block(fsm, code, io);
  if (threadIdx.x == 0)
    a.n_iter[blockIdx.x] = n_iter;
}

/* small lifting sizes: f_mb blocks (SUB = 4: f_mb groups of four byte-interleaved blocks) per workgroup
 * (ldpc_dec_fast_mblock.h); CRC = the launch's stop mode */
template <int SUB, bool CRC>
__global__ void __launch_bounds__(1024) ldpc_dec_fast_multi_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)a.code;
  const uint32_t per_wg = (uint32_t)code->f_mb * SUB;
  const uint32_t first = blockIdx.x * per_wg;
  const uint32_t left = a.n_blocks - first;
  const ldpc_mb_io<false> io{a, first, (ldpc_job_ptr_t) nullptr, a.num_max_iter, a.E, a.crc_pow};
  ldpc_dec_fast_mblock<SUB, false, CRC>(fsm, code, io, left < per_wg ? (int)left : (int)per_wg);
}
/* the same for a group of jobs of the transport-block chain (small segments of one code, cap and CRC) */
template <int SUB, bool CRC>
__global__ void __launch_bounds__(1024) ldpc_dec_fast_multi_jobs_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  typedef const ldpc_dec_mgroup LDPC_CONST_AS *grp_ptr_t;
  const grp_ptr_t gr = (grp_ptr_t)a.mgroups + blockIdx.x;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)gr->code;
  const ldpc_mb_io<true> io{a, gr->first_job, (ldpc_job_ptr_t)a.jobs + gr->first_job, gr->num_max_iter, gr->E, a.crc_pow_tbl[gr->crc_type & 3]};
  ldpc_dec_fast_mblock<SUB, true, CRC>(fsm, code, io, (int)gr->n_valid);
}

bool ldpc_fast_zc_enabled(int zc)
{
  static const bool on = !(getenv("NRLDPC_HIP_ZC") && atoi(getenv("NRLDPC_HIP_ZC")) == 0);
  if (!on)
    return false;
  switch (zc) {
#define X(z) case z:
    LDPC_FAST_ZC_LIST(X)
#undef X
      return LDPC_FAST_ZC != 0;
    default:
      return false;
  }
}

/* launch of the instantiation for lifting size zc (false: there is none) */
template <bool JOBS, bool CRC>
static bool ldpc_launch_zc(int zc, uint32_t n_blocks, int n_threads, int lds_bytes, hipStream_t stream, const ldpc_dec_args &a)
{
  switch (zc) {
#define X(z) \
  case z: \
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<JOBS, CRC, false, z>), dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a); \
    return true;
    LDPC_FAST_ZC_LIST(X)
#undef X
    default:
      return false;
  }
}

hipError_t ldpc_fast_kernel_init(void)
{
  const void *k[] = {reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false, false, true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false>), reinterpret_cast<const void *>(ldpc_dec_fast_kernel<true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<true, true>), reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false, true>),
#define X(z) \
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false, false, false, z>), \
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<false, true, false, z>), \
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<true, true, false, z>), \
                     reinterpret_cast<const void *>(ldpc_dec_fast_kernel<true, false, false, z>),
                     LDPC_FAST_ZC_LIST(X)
#undef X
                     reinterpret_cast<const void *>(ldpc_dec_fast_pull_kernel<false>), reinterpret_cast<const void *>(ldpc_dec_fast_pull_kernel<true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_multi_kernel<1, false>), reinterpret_cast<const void *>(ldpc_dec_fast_multi_kernel<1, true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_multi_kernel<4, false>), reinterpret_cast<const void *>(ldpc_dec_fast_multi_kernel<4, true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_multi_jobs_kernel<1, false>), reinterpret_cast<const void *>(ldpc_dec_fast_multi_jobs_kernel<1, true>),
                     reinterpret_cast<const void *>(ldpc_dec_fast_multi_jobs_kernel<4, false>), reinterpret_cast<const void *>(ldpc_dec_fast_multi_jobs_kernel<4, true>)};
  for (size_t i = 0; i < sizeof(k) / sizeof(k[0]); i++) {
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
  /* a lifting size with instantiations of its own (one block per workgroup: f_rstride = Z + 4, f_astride = 2 Z) */
  if (ldpc_fast_zc_enabled(hc.Z) && hc.f_mb == 1 && hc.f_rstride == hc.Z + 4 && hc.f_astride == 2 * hc.Z && !a.trace) {
    if (a.use_crc ? ldpc_launch_zc<false, true>(hc.Z, n_blocks, hc.f_n_threads, hc.f_lds_total, stream, a)
                  : ldpc_launch_zc<false, false>(hc.Z, n_blocks, hc.f_n_threads, hc.f_lds_total, stream, a))
      return hipGetLastError();
  }
  if (a.use_crc) /* CRC stop: the instantiation without the parity of the hard decisions */
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<false, true>), dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  else if (a.trace)
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<false, false, true>), dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  else
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<false, false>), dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_multi(const ldpc_dec_args &a0, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (a0.jobs || !hc.f_ok || (hc.f_sub != 4 && hc.f_mb < 2))
    return hipErrorInvalidValue;
  ldpc_dec_args a = a0;
  a.n_blocks = n_blocks;
  const uint32_t per_wg = (uint32_t)hc.f_mb * (uint32_t)hc.f_sub;
  const dim3 grid((n_blocks + per_wg - 1) / per_wg), block(hc.f_n_threads);
  if (hc.f_sub == 4) {
    if (a.use_crc)
      hipLaunchKernelGGL((ldpc_dec_fast_multi_kernel<4, true>), grid, block, hc.f_lds_total, stream, a);
    else
      hipLaunchKernelGGL((ldpc_dec_fast_multi_kernel<4, false>), grid, block, hc.f_lds_total, stream, a);
  } else {
    if (a.use_crc)
      hipLaunchKernelGGL((ldpc_dec_fast_multi_kernel<1, true>), grid, block, hc.f_lds_total, stream, a);
    else
      hipLaunchKernelGGL((ldpc_dec_fast_multi_kernel<1, false>), grid, block, hc.f_lds_total, stream, a);
  }
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_multi_jobs(const ldpc_dec_args &a, int sub, int n_threads, int lds_bytes, uint32_t n_groups,
                                           hipStream_t stream)
{
  if (n_groups == 0)
    return hipSuccess;
  if (!a.jobs || !a.mgroups)
    return hipErrorInvalidValue;
  if (sub == 4) {
    if (a.use_crc)
      hipLaunchKernelGGL((ldpc_dec_fast_multi_jobs_kernel<4, true>), dim3(n_groups), dim3(n_threads), lds_bytes, stream, a);
    else
      hipLaunchKernelGGL((ldpc_dec_fast_multi_jobs_kernel<4, false>), dim3(n_groups), dim3(n_threads), lds_bytes, stream, a);
  } else {
    if (a.use_crc)
      hipLaunchKernelGGL((ldpc_dec_fast_multi_jobs_kernel<1, true>), dim3(n_groups), dim3(n_threads), lds_bytes, stream, a);
    else
      hipLaunchKernelGGL((ldpc_dec_fast_multi_jobs_kernel<1, false>), dim3(n_groups), dim3(n_threads), lds_bytes, stream, a);
  }
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_pull(const ldpc_dec_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (a.jobs || !a.pull)
    return hipErrorInvalidValue;
  if (a.use_crc)
    hipLaunchKernelGGL(ldpc_dec_fast_pull_kernel<true>, dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  else
    hipLaunchKernelGGL(ldpc_dec_fast_pull_kernel<false>, dim3(n_blocks), dim3(hc.f_n_threads), hc.f_lds_total, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_dec_fast_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream, int zc)
{
  if (n_blocks == 0)
    return hipSuccess;
  if (ldpc_fast_zc_enabled(zc)) {
    if (a.use_crc ? ldpc_launch_zc<true, true>(zc, n_blocks, n_threads, lds_bytes, stream, a)
                  : ldpc_launch_zc<true, false>(zc, n_blocks, n_threads, lds_bytes, stream, a))
      return hipGetLastError();
  }
  if (a.use_crc) /* (the chain's launches) */
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<true, true>), dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  else
    hipLaunchKernelGGL((ldpc_dec_fast_kernel<true, false>), dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}
