/*
 * ldpc_dec_fast_mblock.h -- SEVERAL code blocks of one small code through the "fast" flooding min-sum decoder in one
 * workgroup (device code; homogeneous batches only).
 *
 * Why: a lifted row of a small code has only Zc/4 four-lane items, and a check-node task is 64 items of ONE degree group
 * (ldpc_graph.c) -- BG1 Zc = 32: 8 items per row, most degree groups hold one to three rows, so a one-block workgroup runs
 * its tasks half empty (Zc = 16: a quarter, Zc = 8: an eighth; profiles/r01/occupancy_sweep.txt: 0.9 / 1.5 / 2.9 ps per
 * edge-lane-pass against 0.5 for the large codes).  With f_mb blocks side by side inside every LDS row (ldpc_graph.h) an
 * item is (row, block, group) and a row of all blocks fills a task.  The per-item arithmetic is the one-block kernel's
 * (ldpc_dec_fast_core.h, with the block's offsets inside a row); what this file adds is per-block state: syndrome flags,
 * CRC registers, pass counts, and an "active" flag -- a block that has stopped (parity or CRC, nrLDPC_decoder.c:842-861)
 * is skipped from then on, so its APP words stay what they were when it stopped, as in the one-block kernel.
 *
 * SUB = 4 (lifting sizes that are not multiples of 4): four blocks are interleaved byte-wise into one block of the virtual
 * code of ldpc_graph.h (f_sub) -- every window an aligned dword, whatever Zc is -- and f_mb counts such groups.  The four
 * blocks of a group share their items, so a stopped block cannot be skipped; instead a block's output is written at the
 * moment it stops (its APP bytes are then still what the stop was decided on) and the group goes on for the others.
 *
 * Contract per block = ldpc_dec_fast_block.h's (nrLDPC_decoder_core, reference nrLDPC_decoder.c:206-880).
 */
#ifndef LDPC_DEC_FAST_MBLOCK_H
#define LDPC_DEC_FAST_MBLOCK_H
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_fast_block.h"

#define LDPC_MB_MAX 64 /* blocks per workgroup: 16 groups x SUB */
/* the workgroup's flag words (ints at f_lds_misc): [0], [1] task queues of the two phases, [2] blocks that have stopped,
 * then arrays of LDPC_MB_MAX: syndrome flags of odd / even passes, CRC registers, active, pass counts; SUB = 4 also: "to
 * be written out now" */
#define LDPC_MB_SYN(par) (16 + LDPC_MB_MAX * (par))
#define LDPC_MB_CRC (16 + 2 * LDPC_MB_MAX)
#define LDPC_MB_ACT (16 + 3 * LDPC_MB_MAX)
#define LDPC_MB_NIT (16 + 4 * LDPC_MB_MAX)
#define LDPC_MB_OUT (16 + 5 * LDPC_MB_MAX) /* needs 16 + 6 * 64 ints = 1600 bytes with SUB = 4 */

/* Where a workgroup's blocks come from and go to.  JOBS = false: blocks first .. of a homogeneous launch, addressed by
 * strides.  JOBS = true: a group of the transport-block chain's jobs (ldpc_kernels.h ldpc_dec_mgroup): buffers, report slot
 * and transport-block abort flag per job; iteration cap, CRC length and CRC polynomial are the group's. */
template <bool JOBS> struct ldpc_mb_io {
  const ldpc_dec_args &a;
  uint32_t first;
  ldpc_job_ptr_t jobs; /* the group's first job (JOBS) */
  int max_iter, crcE;
  const uint32_t *crc_pow;
  __device__ __forceinline__ const int8_t *llr(int b) const
  {
    return a.llr + (JOBS ? (size_t)jobs[b].llr_off : (size_t)(first + (uint32_t)b) * a.llr_stride);
  }
  __device__ __forceinline__ int8_t *out(int b) const
  {
    return a.out + (JOBS ? (size_t)jobs[b].out_off : (size_t)(first + (uint32_t)b) * a.out_stride);
  }
  __device__ __forceinline__ uint32_t iter_slot(int b) const { return JOBS ? (uint32_t)jobs[b].iter_idx : first + (uint32_t)b; }
  __device__ __forceinline__ int *tb_abort(int b) const
  {
    return (JOBS && a.tb_abort && jobs[b].abort_idx >= 0) ? a.tb_abort + jobs[b].abort_idx : nullptr;
  }
};

/* hard decisions of one block -> its output row (bnProc.h:1353-1380 packing).  SUB = 4: the block is byte `sub` of every
 * dword of group `v`. */
template <int SUB>
__device__ __forceinline__ void ldpc_mb_write_out(const ldpc_fast_lds &L, ldpc_code_ptr_t code, const ldpc_dec_args &a, int8_t *orow,
                                                  int v, int sub, int tid, int nt)
{
  const int Zv = code->Z, zr = Zv / SUB, astride = code->f_astride, pa = 2 * Zv;
  const int num_llr = code->num_llr, ncz = code->ncore * zr;
  const uint32_t zr_magic = 0xffffffffu / (uint32_t)zr + 1u;
  if (SUB == 1 && a.out_mode == 0) {
    uint32_t *o = reinterpret_cast<uint32_t *>(orow);
    const int nwords = (num_llr + 31) >> 5;
    for (int w = tid; w < nwords; w += nt) {
      uint32_t word = 0;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int bit = 32 * w + 4 * q; /* Z % 4 == 0: the four bits lie in one column */
        if (bit < ncz) {
          const int c = (int)ldpc_umulhi((uint32_t)bit, zr_magic), u = bit - c * zr;
          const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + v * pa + u);
          const uint32_t nib = (((nb >> 7) & 1u) << 3) | (((nb >> 15) & 1u) << 2) | (((nb >> 23) & 1u) << 1) | (nb >> 31);
          word |= nib << (8 * (q >> 1) + ((q & 1) ? 0 : 4));
        }
      }
      o[w] = word;
    }
    return;
  }
  if (a.out_mode == 0) { /* bit by bit (small codes: at most 68 x 30 bits) */
    uint32_t *o = reinterpret_cast<uint32_t *>(orow);
    const int nwords = (num_llr + 31) >> 5;
    for (int w = tid; w < nwords; w += nt) {
      uint32_t word = 0;
      for (int q = 0; q < 32; q++) {
        const int bit = 32 * w + q;
        if (bit < ncz) {
          const int c = (int)ldpc_umulhi((uint32_t)bit, zr_magic), u = bit - c * zr;
          const uint32_t neg = L.app[c * astride + v * pa + SUB * u + sub] < 128;
          word |= neg << ((q & ~7) + 7 - (q & 7)); /* byte q / 8 of the word, MSB first inside the byte */
        }
      }
      o[w] = word;
    }
  } else {
    for (int i = tid; i < num_llr; i += nt) {
      int8_t val = 0;
      if (i < ncz) {
        const int c = (int)ldpc_umulhi((uint32_t)i, zr_magic), u = i - c * zr;
        val = (int8_t)(L.app[c * astride + v * pa + SUB * u + sub] < 128);
      }
      orow[i] = val;
    }
  }
}

/* blocks first .. first + n_valid - 1 of the launch (n_valid <= f_mb * SUB); results to a.out / a.n_iter */
/* CRC = the launch's stop mode, known to the launcher: the other mode's check is not compiled in (with both, the kernels
 * for SUB = 1 needed scratch memory, which costs every launch 15-20 us on this stack: profiles/r06/README.md) */
template <int SUB, bool JOBS, bool CRC>
__device__ __forceinline__ void ldpc_dec_fast_mblock(uint8_t *fsm, ldpc_code_ptr_t code, const ldpc_mb_io<JOBS> &io, int n_valid)
{
  const ldpc_dec_args &a = io.a;
  const int Z = code->Z, zq = code->f_zq, zqb = code->f_zqb, rstride = code->f_rstride, astride = code->f_astride;
  const int pr = Z + 4, pa = 2 * Z; /* a block's bytes inside a message / extension row, inside an APP row */
  const uint32_t zq_magic = code->f_zq_magic, zqb_magic = code->f_zqb_magic;
  const uint32_t z_magic = 0xffffffffu / (uint32_t)Z + 1u;
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
  L.gllr = nullptr;
  L.ext_global = 0;
  int *flags = reinterpret_cast<int *>(fsm + code->f_lds_misc);
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x, lane = tid & 63;
  const int ncore = code->ncore, num_llr = code->num_llr, ncz = ncore * Z, nedges = code->nedges;
  const int n_app = ncore * zq, n_ext = (code->ncols - ncore) * zq;
  const int zr = Z / SUB; /* the real code's lifting size */
  const int n_groups = (n_valid + SUB - 1) / SUB; /* groups (SUB = 1: blocks) that hold at least one real block */
  uint32_t *llr_lds = reinterpret_cast<uint32_t *>(fsm + code->f_lds_llr); /* SUB = 4: [ncore][f_mb][Z / 4] interleaved channel LLRs */
  (void)llr_lds; (void)ncz; (void)z_magic; (void)zr; (void)num_llr;
  const int max_pass = io.max_iter + 1;

  /* ---- tables and state into LDS ------------------------------------------------------------------------ */
  const uint32_t lds0 = ldpc_lds_addr(fsm);
  for (int i = tid; i < nedges; i += nt)
    etbl[i] = code->f_etbl[i] + lds0;
  for (int i = tid; i < 2 * code->f_n_ctbl; i += nt)
    ctbl[i] = code->f_ctbl[i] + ((i & 1) ? lds0 : 0u);
  for (int i = tid; i < code->nrows; i += nt)
    rowtbl[i] = code->f_rowtbl[i];
  for (int i = tid; i < ncore; i += nt)
    coltbl[i] = code->f_coltbl[i];
  for (int i = tid; i < (code->f_mb * pr) >> 2; i += nt)
    reinterpret_cast<uint32_t *>(fsm + code->f_lds_zero)[i] = 0u;
  if (tid < 16)
    flags[tid] = 0;
  if (tid < LDPC_MB_MAX) {
    flags[LDPC_MB_SYN(0) + tid] = 0;
    flags[LDPC_MB_SYN(1) + tid] = 0;
    flags[LDPC_MB_CRC + tid] = 0;
    flags[LDPC_MB_ACT + tid] = tid < n_valid;
    flags[LDPC_MB_NIT + tid] = max_pass;
    flags[LDPC_MB_OUT + tid] = 0;
  }
  {
    const int nr4 = (nedges * rstride) >> 2;
    uint32_t *r32 = reinterpret_cast<uint32_t *>(L.r);
    for (int i = tid; i < nr4; i += nt)
      r32[i] = 0x80808080u;
  }
  /* APP := channel LLR (both copies) and the extension columns' LLRs, block by block */
  if (SUB == 1) {
    for (int b = 0; b < n_valid; b++) {
      const uint32_t *sb = reinterpret_cast<const uint32_t *>(io.llr(b));
      for (int i = tid; i < n_app; i += nt) {
        const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
        const uint32_t w = sb[i] ^ 0x80808080u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride + b * pa) + j;
        dst[0] = w;
        dst[zq] = w;
      }
      for (int i = tid; i < n_ext; i += nt) {
        const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
        reinterpret_cast<uint32_t *>(L.ext + c * rstride + b * pr)[j] = sb[n_app + i] ^ 0x80808080u;
      }
    }
  } else {
    /* dword (column c, real lane t) of group v = the four blocks' LLR bytes at c * zr + t (zero for a block beyond the
     * batch); n_app / n_ext count exactly these dwords per group (zq = Z / 4 = zr) */
    for (int v = 0; v < n_groups; v++) {
      const int nsub = n_valid - SUB * v; /* real blocks in this group (>= 1) */
      const int8_t *sb[SUB];
#pragma unroll
      for (int q = 0; q < SUB; q++)
        sb[q] = io.llr(SUB * v + (q < nsub ? q : 0));
      for (int i = tid; i < n_app + n_ext; i += nt) {
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < SUB; q++)
          if (q < nsub)
            w |= (uint32_t)(uint8_t)sb[q][i] << (8 * q);
        const uint32_t wb = w ^ 0x80808080u;
        if (i < n_app) {
          const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
          uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride + v * pa) + j;
          dst[0] = wb;
          dst[zq] = wb;
          llr_lds[(c * code->f_mb + v) * zq + j] = w;
        } else {
          const int k = i - n_app;
          const int c = (int)ldpc_umulhi((uint32_t)k, zq_magic), j = k - c * zq;
          reinterpret_cast<uint32_t *>(L.ext + c * rstride + v * pr)[j] = wb;
        }
      }
    }
  }
  __syncthreads();

  /* ---- passes ------------------------------------------------------------------------------------------ */
  const int n_cn_tasks = code->f_n_cn_tasks, n_bn_tasks = code->f_n_bn_tasks, bn_group = code->f_bn_group;
  constexpr int use_crc = CRC ? 1 : 0;
  /* is group / block g still worked on?  SUB = 1: the block's own flag; SUB = 4: any of its four */
  auto group_active = [&](int g) -> bool {
    if (SUB == 1)
      return flags[LDPC_MB_ACT + g] != 0;
    const int4 f = *reinterpret_cast<const int4 *>(&flags[LDPC_MB_ACT + SUB * g]);
    return (f.x | f.y | f.z | f.w) != 0;
  };
  /* SUB = 4: rows of the blocks that have just stopped (flags[LDPC_MB_OUT]) -- called between two barriers while their
   * APP bytes are what the stop was decided on */
  auto flush_stopped = [&]() {
    if (SUB == 1)
      return;
    for (int blk = 0; blk < n_valid; blk++)
      if (flags[LDPC_MB_OUT + blk] && (!use_crc || flags[LDPC_MB_NIT + blk] >= 3))
        ldpc_mb_write_out<SUB>(L, code, a, io.out(blk), blk / SUB, blk % SUB, tid, nt);
  };
  for (int p = 1; p <= max_pass; ++p) {
    for (;;) {
      const int task = ldpc_draw(&flags[0], lane);
      if (task >= n_cn_tasks)
        break;
      const int deg = code->f_cn_task[task][0], ext = code->f_cn_task[task][1];
      const int item = code->f_cn_task[task][2] + lane;
      const int gstart = code->f_cn_task[task][3], gend = code->f_cn_task[task][4], srow0 = code->f_cn_task[task][5];
      if (item < gend) {
        const int gi = item - gstart;
        const int rig = (int)ldpc_umulhi((uint32_t)gi, zqb_magic), jb = gi - rig * zqb;
        const int b = (int)ldpc_umulhi((uint32_t)jb, zq_magic), j = jb - b * zq;
        if (b < n_groups && group_active(b)) {
          const uint32_t rowrec = rowtbl[srow0 + rig];
          const int e0 = (int)(rowrec & 0x1ffu), valid = (int)(rowrec >> 16) - 4 * j;
          const uint32_t m = ldpc_fast_cn_dispatch(deg, ext, L, e0, j, Z, rstride, b * pr, b * pa);
          const uint32_t bad = m & ldpc_fast_valid_lanes(valid); /* byte i = lane 4j + i */
          if (SUB == 1) {
            if (bad)
              flags[LDPC_MB_SYN(p & 1) + b] = 1; /* (same value from every lane that writes) */
          } else { /* bit i = lane 4j + i of the virtual code = lane j of block i */
#pragma unroll
            for (int q = 0; q < SUB; q++)
              if (bad & (1u << (8 * q)))
                flags[LDPC_MB_SYN(p & 1) + SUB * b + q] = 1;
          }
        }
      }
    }
    if (tid == 0)
      flags[1] = 0; /* nobody draws bit-node tasks now */
    __syncthreads();
    /* nrLDPC_decoder.c:842-848: a block whose previous pass satisfied every (looked-at) check stops, from pass 3 on */
    if (tid < n_valid && flags[LDPC_MB_ACT + tid] && !use_crc && p >= 3 && flags[LDPC_MB_SYN(p & 1) + tid] == 0) {
      flags[LDPC_MB_NIT + tid] = p - 1;
      flags[LDPC_MB_ACT + tid] = 0;
      flags[LDPC_MB_OUT + tid] = 1;
      atomicAdd(&flags[2], 1);
    }
    if (JOBS && tid < n_valid && flags[LDPC_MB_ACT + tid] && p >= 2) {
      /* decoder.c:556-559: once a segment of the transport block has failed, its siblings give up at their next pass
       * (pass count max_pass + 1, nothing written) */
      int *ab = io.tb_abort(tid);
      if (ab && __hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        flags[LDPC_MB_NIT + tid] = max_pass + 1;
        flags[LDPC_MB_ACT + tid] = 0;
        atomicAdd(&flags[2], 1);
      }
    }
    __syncthreads();
    flush_stopped();
    if (flags[2] >= n_valid)
      break;
    if (SUB != 1) {
      __syncthreads();
      if (tid < LDPC_MB_MAX)
        flags[LDPC_MB_OUT + tid] = 0;
    }
    for (;;) {
      const int ticket = ldpc_draw(&flags[1], lane);
      if (ticket * bn_group >= n_bn_tasks)
        break;
      for (int task = ticket * bn_group; task < (ticket + 1) * bn_group && task < n_bn_tasks; task++) {
        const int item = code->f_bn_task[task][0] + lane, end = code->f_bn_task[task][1];
        const int maxdeg = code->f_bn_task[task][2];
        if (item < end) {
          const int sc = (int)ldpc_umulhi((uint32_t)item, zqb_magic), jb = item - sc * zqb;
          const int b = (int)ldpc_umulhi((uint32_t)jb, zq_magic), j = jb - b * zq;
          if (b < n_groups && group_active(b)) {
            const uint32_t colrec = coltbl[sc];
            const int c = (int)(colrec & 0xffu);
            const uint32_t lw = SUB == 1 ? reinterpret_cast<const uint32_t *>(io.llr(b))[c * zq + j] : llr_lds[(c * code->f_mb + b) * zq + j];
            ldpc_fast_bn(L, colrec, maxdeg, j, Z, astride, lw, b * pr, b * pa);
          }
        }
      }
    }
    if (tid == 0)
      flags[0] = 0; /* nobody draws check-node tasks now */
    if (tid < LDPC_MB_MAX) {
      flags[LDPC_MB_SYN((p + 1) & 1) + tid] = 0;
      flags[LDPC_MB_CRC + tid] = 0;
    }
    __syncthreads();
    if (use_crc && p >= 3) { /* decoder.c:849-861, per block; see ldpc_dec_generic_block.h for the CRC argument */
      const int crcE = io.crcE;
      const uint32_t *crc_pow = io.crc_pow;
      for (int blk = 0; blk < n_valid; blk++) {
        if (!flags[LDPC_MB_ACT + blk])
          continue;
        uint32_t x = 0;
        if (SUB == 1) {
#pragma unroll 1
          for (int i = 4 * tid; i < crcE; i += 4 * nt) {
            const int c = (int)ldpc_umulhi((uint32_t)i, z_magic), u = i - c * Z;
            const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + blk * pa + u);
            const uint4 pw = *reinterpret_cast<const uint4 *>(crc_pow + (crcE - 4 - i));
            x ^= (pw.w & (0u - ((nb >> 7) & 1u))) ^ (pw.z & (0u - ((nb >> 15) & 1u))) ^ (pw.y & (0u - ((nb >> 23) & 1u))) ^
                 (pw.x & (0u - (nb >> 31)));
          }
        } else {
          const uint32_t zr_magic = 0xffffffffu / (uint32_t)zr + 1u;
          const int v = blk / SUB, sub = blk % SUB;
          for (int i = tid; i < crcE; i += nt) {
            const int c = (int)ldpc_umulhi((uint32_t)i, zr_magic), u = i - c * zr;
            if (L.app[c * astride + v * pa + SUB * u + sub] < 128)
              x ^= crc_pow[crcE - 1 - i];
          }
        }
        for (int off = 32; off; off >>= 1)
          x ^= __shfl_xor(x, off);
        if (lane == 0 && x)
          atomicXor(reinterpret_cast<unsigned int *>(&flags[LDPC_MB_CRC + blk]), x);
      }
      __syncthreads();
      if (tid < n_valid && flags[LDPC_MB_ACT + tid] && flags[LDPC_MB_CRC + tid] == 0) {
        flags[LDPC_MB_NIT + tid] = p;
        flags[LDPC_MB_ACT + tid] = 0;
        flags[LDPC_MB_OUT + tid] = 1;
        atomicAdd(&flags[2], 1);
      }
      __syncthreads();
      flush_stopped();
      if (flags[2] >= n_valid)
        break;
      if (SUB != 1) {
        __syncthreads();
        if (tid < LDPC_MB_MAX)
          flags[LDPC_MB_OUT + tid] = 0;
        __syncthreads();
      }
    }
  }

  /* ---- hard decisions: SUB = 1 every block (a stopped block's APP words were left alone); SUB = 4 the blocks that never
   * stopped ------------------------------------------------------------------------------------------------ */
  for (int blk = 0; blk < n_valid; blk++) {
    const int n_iter = flags[LDPC_MB_NIT + blk];
    if ((use_crc && n_iter < 3) || n_iter > max_pass)
      continue; /* decoder.c:849-861: p_out stays untouched; a block given up with its transport block writes nothing */
    if (SUB != 1 && !flags[LDPC_MB_ACT + blk])
      continue; /* written when it stopped */
    ldpc_mb_write_out<SUB>(L, code, a, io.out(blk), blk / SUB, blk % SUB, tid, nt);
  }
  if (tid < n_valid) {
    const int n_iter = flags[LDPC_MB_NIT + tid];
    a.n_iter[io.iter_slot(tid)] = n_iter;
    if (JOBS && n_iter == max_pass) { /* decoder.c:190-193: a failed segment gives the whole transport block up */
      int *ab = io.tb_abort(tid);
      if (ab)
        __hip_atomic_store(ab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif
