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
#include "ldpc_dec_core.h"

__global__ void __launch_bounds__(1024) ldpc_dec_generic_kernel(const ldpc_dec_args a)
{
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];
  /* job records and descriptors are read through the constant address space: uniform address -> scalar loads,
   * so everything derived from them stays in SGPRs */
  typedef const ldpc_dec_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code); /* never written while a kernel runs */
  const int Z = code->Z;
  int8_t *r = smem + code->lds_r;
  int8_t *app = smem + code->lds_app;
  int8_t *llr_s = smem + code->lds_llr;
  int *flags = reinterpret_cast<int *>(smem + code->lds_misc);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const int wave = LDPC_UNIFORM(tid >> 6), nw = nt >> 6;
  const uint32_t blk = blockIdx.x;
  const int num_llr = code->num_llr, ncz = code->ncore * Z;

  /* ---- stage the channel LLRs, clear the messages ------------------------------------------------ */
  const int8_t *__restrict__ src = a.llr + (job ? (size_t)job->llr_off : (size_t)blk * a.llr_stride);
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int n16 = num_llr >> 4;
    for (int i = tid; i < n16; i += nt)
      reinterpret_cast<int4 *>(llr_s)[i] = reinterpret_cast<const int4 *>(src)[i];
    for (int i = (n16 << 4) + tid; i < num_llr; i += nt)
      llr_s[i] = src[i];
  } else {
    for (int i = tid; i < num_llr; i += nt)
      llr_s[i] = src[i];
  }
  const int nr16 = (code->nedges * Z + 15) >> 4;
  for (int i = tid; i < nr16; i += nt)
    reinterpret_cast<int4 *>(r)[i] = make_int4(0, 0, 0, 0);
  if (tid < 4)
    flags[tid] = 0;
  __syncthreads();
  /* with r = 0 and app = llr the first check-node phase sees q = llr, i.e. nrLDPC_llr2CnProcBuf (mPass.h:128-221) */
  for (int i = tid; i < ncz; i += nt)
    app[i] = llr_s[i];
  __syncthreads();

  /* ---- passes (decoder.c:552-558: one unconditional pass + up to numMaxIter more) ------------------ */
  const int max_pass = (job ? job->num_max_iter : a.num_max_iter) + 1;
  const int crcE = job ? job->E : a.E;
  const uint32_t *crc_pow = job ? a.crc_pow_tbl[job->crc_type] : a.crc_pow;
  int n_iter = max_pass;
  /* small lifting sizes: several rows / a run of column bits per 64-lane step (ldpc_graph.h `pack`) */
  const int pack = code->pack;
  const int sub = pack > 1 ? (lane * code->zinv16) >> 16 : 0, tl = lane - sub * Z;
  for (int p = 1; p <= max_pass; ++p) {
    /* check-node phase; its syndrome is that of pass p-1 */
    int par_acc = 0;
    const int ncn = code->n_cn_slots;
    for (int k = wave; k < ncn; k += nw) {
      const int ent = LDPC_UNIFORM(code->cn_order[k]);
      if (pack > 1) {
        if (sub < (ent >> 8)) {
          const int row = code->cn_rows[(ent & 0xff) + sub];
          const int par = ldpc_cn_row(code, row, tl, r, app, llr_s);
          par_acc |= (tl < code->pc_lo[row]) ? par : 0;
        }
        continue;
      }
      const int row = ent >> 4, t = ((ent & 15) << 6) + lane;
      if (t < Z) {
        const int par = ldpc_cn_row(code, row, t, r, app, llr_s);
        par_acc |= (t < code->pc_lo[row]) ? par : 0;
      }
    }
    if (__any(par_acc) && lane == 0)
      flags[p & 1] = 1;
    if (tid == 0)
      flags[2] = 0;
    __syncthreads();
    /* cnProcPc of pass p-1 (decoder.c:842-848); never evaluated after the first pass (decoder.c:548-551) */
    if (!a.use_crc && p >= 3 && flags[p & 1] == 0) {
      n_iter = p - 1;
      break;
    }
    /* bit-node phase */
    const int nbn = code->n_bn_slots;
    for (int k = wave; k < nbn; k += nw) {
      if (pack > 1) {
        const int i = 64 * k + lane;
        if (i < ncz) {
          const int ci = (i * code->zinv16) >> 16;
          ldpc_bn_update(code, code->bn_cols[ci], i - ci * Z, r, app, llr_s);
        }
        continue;
      }
      const int ent = LDPC_UNIFORM(code->bn_order[k]);
      const int c = ent >> 4, u = ((ent & 15) << 6) + lane;
      if (u < Z)
        ldpc_bn_update(code, c, u, r, app, llr_s);
    }
    if (tid == 0)
      flags[(p + 1) & 1] = 0;
    __syncthreads();
    /* CRC stop from the third pass on (decoder.c:849-861).  check_crc(p_out, E, type) (crc_byte.c:314-380)
     * holds iff the E-bit word [data | crc] is divisible by g(x); the remainder is linear in the bits:
     * XOR over the set bits i of x^(E-1-i) mod g. */
    if (a.use_crc && p >= 3) {
      uint32_t x = 0;
      for (int i = tid; i < crcE; i += nt)
        if (app[i] < 0)
          x ^= crc_pow[crcE - 1 - i];
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

  /* ---- hard decision (decoder.c:864-879; in CRC mode p_out is only written from pass 3 on) ----------- */
  if (!a.use_crc || n_iter >= 3) {
    if (a.out_mode == 0) {
      uint32_t *o = reinterpret_cast<uint32_t *>(a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride));
      const int nwords = (num_llr + 31) >> 5;
      for (int w = tid; w < nwords; w += nt)
        o[w] = (32 * w < ncz) ? ldpc_pack_word(app, w, ncz) : 0u;
    } else {
      int8_t *o = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
      for (int i = tid; i < num_llr; i += nt)
        o[i] = (i < ncz) ? (int8_t)(app[i] < 0) : (int8_t)0;
    }
  }
  if (tid == 0)
    a.n_iter[job ? (uint32_t)job->iter_idx : blk] = n_iter;
}

hipError_t ldpc_kernels_init(void)
{
  return hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_dec_generic_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
