/*
 * ldpc_dec_generic_block.h -- one code block through the generic flooding min-sum decoder, executed by one workgroup
 * (device code; any (BG, Zc, R)).  Shared by the batch kernel (ldpc_decoder.hip) and the resident server kernel
 * (ldpc_server.hip).  Contract = nrLDPC_decoder_core (reference nrLDPC_decoder/nrLDPC_decoder.c:206-880); per-thread
 * arithmetic in ldpc_dec_core.h.  Contains workgroup barriers: every thread of the workgroup calls it with the same
 * arguments; any workgroup size that is a multiple of 64 works.
 */
#ifndef LDPC_DEC_GENERIC_BLOCK_H
#define LDPC_DEC_GENERIC_BLOCK_H
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_core.h"

struct ldpc_gblock_io {
  const int8_t *llr;       /* the block's channel LLRs, ncols*Z int8 (read once) */
  int8_t *out;
  int max_pass;            /* numMaxIter + 1 */
  int use_crc, crcE;
  const uint32_t *crc_pow;
  int out_mode;
  int *tb_abort;           /* optional transport-block wide failure flag (decoder.c:190-193, 556-559) */
  /* CRC stop decided by a predicate on the HOST (ldpc_api.cpp dec_host_predicate): with use_crc and trace != NULL no CRC is
   * evaluated here and nothing stops the passes; the hard decisions of every pass p >= 3 -- what the reference hands to
   * check_crc after that pass (decoder.c:849-861) -- are stored at trace + (p - 3) * trace_stride in out_mode's format */
  int8_t *trace;
  uint32_t trace_stride;
};

/* nrLDPC_llrRes2llrOut + llr2bitPacked / llr2bit (decoder.c:851-856, 866-877): columns behind the core report 0 [F5] */
__device__ __forceinline__ void ldpc_gblock_store(int out_mode, int8_t *out, const int8_t *app, int num_llr, int ncz, int tid, int nt)
{
  if (out_mode == 0) {
    uint32_t *o = reinterpret_cast<uint32_t *>(out);
    const int nwords = (num_llr + 31) >> 5;
    for (int w = tid; w < nwords; w += nt)
      o[w] = (32 * w < ncz) ? ldpc_pack_word(app, w, ncz) : 0u;
  } else {
    for (int i = tid; i < num_llr; i += nt)
      out[i] = (i < ncz) ? (int8_t)(app[i] < 0) : (int8_t)0;
  }
}

__device__ __forceinline__ int ldpc_dec_generic_block(int8_t *smem, ldpc_code_ptr_t code, const ldpc_gblock_io &io)
{
  const int Z = code->Z;
  int8_t *r = smem + code->lds_r;
  int8_t *app = smem + code->lds_app;
  int8_t *llr_s = smem + code->lds_llr;
  int *flags = reinterpret_cast<int *>(smem + code->lds_misc);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const int wave = LDPC_UNIFORM(tid >> 6), nw = nt >> 6;
  const int num_llr = code->num_llr, ncz = code->ncore * Z;

  /* ---- stage the channel LLRs, clear the messages ------------------------------------------------ */
  const int8_t *__restrict__ src = io.llr;
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
  const int max_pass = io.max_pass;
  const int crcE = io.crcE;
  const uint32_t *crc_pow = io.crc_pow;
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
    if (tid == 0) {
      flags[2] = 0;
      /* decoder.c:556-559: once a segment of the transport block has failed, its siblings give up at their next pass */
      if (io.tb_abort && p >= 2 && __hip_atomic_load(io.tb_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        flags[3] = 1;
    }
    __syncthreads();
    if (io.tb_abort && flags[3]) {
      n_iter = max_pass + 1;
      break;
    }
    /* cnProcPc of pass p-1 (decoder.c:842-848); never evaluated after the first pass (decoder.c:548-551) */
    if (!io.use_crc && p >= 3 && flags[p & 1] == 0) {
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
    if (io.use_crc && p >= 3 && io.trace) { /* (the next check-node phase only reads app: no barrier needed behind this) */
      ldpc_gblock_store(io.out_mode, io.trace + (size_t)(p - 3) * io.trace_stride, app, num_llr, ncz, tid, nt);
    } else if (io.use_crc && p >= 3) {
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

  if (io.tb_abort && n_iter == max_pass && tid == 0) /* decoder.c:190-193 */
    __hip_atomic_store(io.tb_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  /* ---- hard decision (decoder.c:864-879; in CRC mode p_out is only written from pass 3 on) ----------- */
  if ((!io.use_crc || n_iter >= 3) && n_iter <= max_pass && !(io.use_crc && io.trace))
    ldpc_gblock_store(io.out_mode, io.out, app, num_llr, ncz, tid, nt);
  return n_iter;
}
#endif
