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
#include "ldpc_dec_fast_core.h"

/* next ticket of a task queue (wave-uniform) */
__device__ __forceinline__ int ldpc_draw(int *counter, int lane)
{
  int t = 0;
  if (lane == 0)
    t = atomicAdd(counter, 1);
  return LDPC_UNIFORM(t);
}

template <int MAX_THREADS>
__global__ void __launch_bounds__(MAX_THREADS) ldpc_dec_fast_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  typedef const ldpc_dec_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  const int Z = code->Z, zq = code->f_zq, rstride = code->f_rstride, astride = code->f_astride;
  const uint32_t zq_magic = code->f_zq_magic;
  const uint32_t z_magic = 0xffffffffu / (uint32_t)Z + 1u; /* ceil(2^32 / Z) for Z not a power of two, exact enough
                                                               for b < 2^16 either way (checked on the host) */
  ldpc_fast_lds L;
  L.base = fsm;
  L.r = fsm + code->f_lds_r;
  L.app = fsm + code->f_lds_app;
  L.ext = fsm + code->f_lds_ext;
  uint32_t *etbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_etbl);
  uint32_t *ctbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_ctbl);
  uint32_t *rowtbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_rowtbl);
  uint32_t *coltbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_coltbl);
  L.etbl = etbl; L.ctbl = ctbl; L.rowtbl = rowtbl; L.coltbl = coltbl;
  int *flags = reinterpret_cast<int *>(fsm + code->f_lds_misc);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
#ifdef LDPC_TIMING
  const long long dbg_k0 = clock64(); /* kernel entry */
#endif
  const int wave = LDPC_UNIFORM(tid >> 6);
  const uint32_t blk = blockIdx.x;
  const int ncore = code->ncore, num_llr = code->num_llr, ncz = ncore * Z, nedges = code->nedges;
  const uint32_t *__restrict__ src32 = reinterpret_cast<const uint32_t *>(a.llr + (job ? (size_t)job->llr_off : (size_t)blk * a.llr_stride));

  /* ---- tables and state into LDS -------------------------------------------------------------------- */
  const uint32_t lds0 = ldpc_lds_addr(fsm); /* tables hold absolute LDS addresses from here on */
  const int ext_global = code->f_ext_global;
  L.gllr = reinterpret_cast<const uint8_t *>(src32);
  L.ext_global = ext_global;
  /* The block's LLRs come from HBM: the first four dwords per thread of the core and of the extension columns are
   * requested before anything else and consumed after the table copies and the message initialisation, so that their
   * latency runs in the background (a 1024-thread workgroup needs 3 + 4 such loads per thread for Zc = 384). */
  const int n_app = ncore * zq, n_ext = ext_global ? 0 : (code->ncols - ncore) * zq;
  uint32_t va[4], ve[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int ia = tid + k * nt;
    va[k] = ia < n_app ? src32[ia] : 0u;
    ve[k] = ia < n_ext ? src32[n_app + ia] : 0u;
  }
  for (int i = tid; i < nedges; i += nt)
    etbl[i] = code->f_etbl[i] + ((ext_global && code->e_col[i] >= ncore) ? 0u : lds0);
  for (int i = tid; i < 2 * code->f_n_ctbl; i += nt)
    ctbl[i] = code->f_ctbl[i] + ((i & 1) ? lds0 : 0u);
  for (int i = tid; i < (Z + 4) >> 2; i += nt)
    reinterpret_cast<uint32_t *>(fsm + code->f_lds_zero)[i] = 0u;
  for (int i = tid; i < code->nrows; i += nt)
    rowtbl[i] = code->f_rowtbl[i];
  for (int i = tid; i < ncore; i += nt)
    coltbl[i] = code->f_coltbl[i];
  if (tid < 8)
    flags[tid] = 0; /* [0], [1] syndrome flags of odd / even passes, [2] CRC register, [4], [5] task queues of the two phases */
  {
    const int nr4 = (nedges * rstride) >> 2;
    uint32_t *r32 = reinterpret_cast<uint32_t *>(L.r);
    for (int i = tid; i < nr4; i += nt)
      r32[i] = 0x80808080u;
  }
  /* APP := channel LLR (both copies), so that with r = 0 the first check-node phase sees q = llr */
  uint32_t *e32 = reinterpret_cast<uint32_t *>(L.ext);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i = tid + k * nt;
    if (i < n_app) {
      const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
      const uint32_t w = va[k] ^ 0x80808080u;
      uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride) + j;
      dst[0] = w;
      dst[zq] = w;
    }
    if (i < n_ext)
      e32[i] = ve[k] ^ 0x80808080u;
  }
  for (int i = tid + 4 * nt; i < n_app; i += nt) { /* small workgroups: the rest */
    const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
    const uint32_t w = src32[i] ^ 0x80808080u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride) + j;
    dst[0] = w;
    dst[zq] = w;
  }
  for (int i = tid + 4 * nt; i < n_ext; i += nt)
    e32[i] = src32[n_app + i] ^ 0x80808080u;
  __syncthreads();

  /* ---- passes ------------------------------------------------------------------------------------------ */
  const int max_pass = (job ? job->num_max_iter : a.num_max_iter) + 1;
  const int crcE = job ? job->E : a.E;
  const uint32_t *crc_pow = job ? a.crc_pow_tbl[job->crc_type] : a.crc_pow;
  int n_iter = max_pass;
  const int n_cn_tasks = code->f_n_cn_tasks, n_bn_tasks = code->f_n_bn_tasks, bn_group = code->f_bn_group;
  (void)wave;
#ifdef LDPC_TIMING /* diagnostic build (tools/task_timing.sh): block 0 dumps, for pass 2, {start, end, degree} of every task
                      each wave ran into its output row instead of the decoded bits */
  long long *dbg = reinterpret_cast<long long *>(a.out);
  int dbg_n = 0;
  const long long dbg_t0 = clock64();
#define LDPC_TIMING_BEGIN const long long tt0 = clock64();
#define LDPC_TIMING_END(phase, deg)                                                        \
  if (blk == 0 && lane == 0 && p == 2 && dbg_n < 12) {                                     \
    dbg[(wave * 12 + dbg_n) * 2] = ((tt0 - dbg_t0) << 20) | (long long)((phase) << 8 | (deg)); \
    dbg[(wave * 12 + dbg_n) * 2 + 1] = clock64() - dbg_t0;                                 \
    dbg_n++;                                                                               \
  }
#else
#define LDPC_TIMING_BEGIN
#define LDPC_TIMING_END(phase, deg)
#endif
  for (int p = 1; p <= max_pass; ++p) {
    uint32_t syn = 0;
#ifdef LDPC_ABLATE_CN
    syn = 1;
#else
    /* The phase's tasks are drawn in id order (= most expensive first, ldpc_graph.c) from a queue -- an LDS counter --
     * by whichever wave is free: the SIMD issue arbiter favours a CU's older waves, so static equal shares leave the
     * SIMDs with one or two live waves for the last third of a phase (tools/task_timing.py shows the timeline). */
    for (;;) {
      const int task = ldpc_draw(&flags[4], lane);
      if (task >= n_cn_tasks)
        break;
      LDPC_TIMING_BEGIN
      const int deg = code->f_cn_task[task][0], ext = code->f_cn_task[task][1];
      const int item = code->f_cn_task[task][2] + lane;
      const int gstart = code->f_cn_task[task][3], gend = code->f_cn_task[task][4], srow0 = code->f_cn_task[task][5];
      if (item < gend) {
        const int gi = item - gstart;
        const int rig = (int)ldpc_umulhi((uint32_t)gi, zq_magic), j = gi - rig * zq;
        const uint32_t rowrec = rowtbl[srow0 + rig];
        const int e0 = (int)(rowrec & 0xffffu), valid = (int)(rowrec >> 16) - 4 * j; /* lanes t+i < pc_lo are checked */
        const uint32_t m = ldpc_fast_cn_dispatch(deg, ext, L, e0, j, Z, rstride);
        const uint32_t mask = valid >= 4 ? 0xfu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
        syn |= m & mask;
      }
      LDPC_TIMING_END(0, deg)
    }
#endif
    if (__any(syn != 0) && lane == 0)
      flags[p & 1] = 1;
    if (tid == 0) {
      flags[2] = 0;
      flags[5] = 0; /* nobody draws bit-node tasks now */
    }
    __syncthreads();
    if (!a.use_crc && p >= 3 && flags[p & 1] == 0) {
      n_iter = p - 1;
      break;
    }
#ifndef LDPC_ABLATE_BN
    for (;;) {
      const int ticket = ldpc_draw(&flags[5], lane);
      if (ticket * bn_group >= n_bn_tasks)
        break;
      for (int task = ticket * bn_group; task < (ticket + 1) * bn_group && task < n_bn_tasks; task++) {
        LDPC_TIMING_BEGIN
        const int item = code->f_bn_task[task][0] + lane, end = code->f_bn_task[task][1];
        const int maxdeg = code->f_bn_task[task][2];
        if (item < end) {
          const int sc = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sc * zq;
          const uint32_t colrec = coltbl[sc];
          const uint32_t lw = src32[(int)(colrec & 0xffu) * zq + j];
          ldpc_fast_bn(L, colrec, maxdeg, j, Z, astride, lw);
        }
        LDPC_TIMING_END(1, maxdeg)
      }
    }
#endif
    if (tid == 0) {
      flags[(p + 1) & 1] = 0;
      flags[4] = 0; /* nobody draws check-node tasks now */
    }
    __syncthreads();
    if (a.use_crc && p >= 3) { /* see ldpc_decoder.hip for the CRC argument */
      uint32_t x = 0;
      /* four hard decisions (one APP dword: Zc % 4 == 0 keeps them in one column) and their four table entries per
       * step, the loads unconditional and masked afterwards: independent loads in flight instead of a chain of
       * bit test -> load -> wait (E is a multiple of 8) */
      for (int i = 4 * tid; i < crcE; i += 4 * nt) {
        const int c = (int)ldpc_umulhi((uint32_t)i, z_magic), u = i - c * Z;
        const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u); /* bit 8k+7 set <=> APP of lane k < 0 */
        /* pw[3 - k] belongs to bit i + k; E % 8 == 0 and i % 4 == 0 make the four entries one aligned 16-byte load */
        const uint4 pw = *reinterpret_cast<const uint4 *>(crc_pow + (crcE - 4 - i));
        const uint32_t p3 = pw.w, p2 = pw.z, p1 = pw.y, p0 = pw.x;
        x ^= (p3 & (0u - ((nb >> 7) & 1u))) ^ (p2 & (0u - ((nb >> 15) & 1u))) ^ (p1 & (0u - ((nb >> 23) & 1u))) ^
             (p0 & (0u - (nb >> 31)));
      }
      for (int off = 32; off; off >>= 1)
        x ^= __shfl_xor(x, off);
      if (lane == 0 && x)
        atomicXor(reinterpret_cast<unsigned int *>(&flags[2]), x);
      __syncthreads();
      const int rem = flags[2];
      __syncthreads();
      if (rem == 0) {
        n_iter = p;
        break;
      }
    }
  }

#ifdef LDPC_TIMING
  const long long dbg_t1 = clock64();
#endif
  /* ---- hard decision ------------------------------------------------------------------------------------- */
#ifdef LDPC_TIMING
  if (blk != 0)
#endif
  if (!a.use_crc || n_iter >= 3) {
    if (a.out_mode == 0) {
      uint32_t *o = reinterpret_cast<uint32_t *>(a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride));
      const int nwords = (num_llr + 31) >> 5;
      for (int w = tid; w < nwords; w += nt) {
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int b = 32 * w + 4 * q; /* Z % 4 == 0: the four bits lie in one column */
          if (b < ncz) {
            const int c = (int)ldpc_umulhi((uint32_t)b, z_magic), u = b - c * Z;
            const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u);
            const uint32_t nib = (((nb >> 7) & 1u) << 3) | (((nb >> 15) & 1u) << 2) | (((nb >> 23) & 1u) << 1) | (nb >> 31);
            word |= nib << (8 * (q >> 1) + ((q & 1) ? 0 : 4));
          }
        }
        o[w] = word;
      }
    } else {
      int8_t *o = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
      for (int i = tid; i < num_llr; i += nt)
        o[i] = (i < ncz) ? (int8_t)ldpc_fast_hd(L, i, Z, z_magic, astride) : (int8_t)0;
    }
  }
  if (tid == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blk] = n_iter;
#ifdef LDPC_TIMING
  if (tid == 0 && (blk == 0 || blk == 1000)) { /* phases of one block: prologue, passes (incl. this stamp), whole kernel */
    long long *d2 = reinterpret_cast<long long *>(a.out + (size_t)blk * a.out_stride) + (blk == 0 ? 16 * 12 * 2 : 0);
    d2[0] = dbg_t0 - dbg_k0;
    d2[1] = dbg_t1 - dbg_t0;
    d2[2] = clock64() - dbg_k0;
  }
#endif
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
