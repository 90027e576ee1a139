/*
 * ldpc_dec_fast_part.h -- ONE code block decoded by SEVERAL workgroups (CUs) of the resident server (device code).
 *
 * Why: a per-segment call (LDPCdecoder, reference nrLDPC_decoder.c:172) is one block, and one CU needs 12 us per pass of
 * BG1 Zc = 384 -- the call's latency is passes x that.  Here the rows of the code are dealt to G parts (ldpc_graph.c,
 * ldpc_build_code_desc_part).  Every part keeps the messages of ITS rows and the APP of ALL columns in its LDS and runs
 *   check-node phase over its rows (ldpc_dec_fast_core.h, unchanged)
 *   bit-node phase, first half: per column the sum of its own edges' messages (ldpc_fast_bn_gather)
 *   EXCHANGE: the partial sums (8 bytes per column and four-lane group: 20 KB for Zc = 384) go to a buffer in device
 *     memory, all parts meet at a counter, every part reads the others' sums
 *   bit-node phase, second half: total + channel LLR -> clamped APP (ldpc_fast_bn_finish), every part for all columns.
 * Sums of integers: the APPs, hence every message, hard decision and pass count, are those of the one-workgroup kernel.
 * The stop criteria need the whole code's view: the parts' unsatisfied-lane counts travel with the partial sums (parity
 * check folded into the check-node phase) resp. through a second, header-only exchange (the sweep right after a pass);
 * the CRC is computed by every part on its own copy of the APPs.  All parts therefore take the same decisions and go
 * through the same number of exchanges, which is what keeps their counters in step.
 *
 * Cost of one exchange between 2 / 4 CUs: 1.8 / 3.3-3.8 us (tools/ubench/xcu_exchange.hip, profiles/r02).
 *
 * IO: ldpc_dec_fast_block.h's accessors plus
 *   int part(), parts()
 *   unsigned long long *exch()       the slot's exchange area: [2 parities][parts][LDPC_PART_STRIDE] words
 *   unsigned int *meet_counter()     the slot's arrival counter (device memory, zero when the kernel starts)
 *   uint32_t *meet_target()          LDS word of this workgroup: arrivals expected so far (persists across calls)
 *   uint32_t *fault()                LDS word: set when a meeting timed out (a part is missing: results are void)
 *   uint32_t *phase_ticks()          optional (LDS, 5 words): clock ticks spent in check-node phase / gather + publish /
 *                                    meeting / loads + finish / sweep after the pass, summed over the passes (diagnostics)
 */
#ifndef LDPC_DEC_FAST_PART_H
#define LDPC_DEC_FAST_PART_H
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_fast_block.h"

#define LDPC_PART_ITEMS_MAX (LDPC_MAX_CORE * (LDPC_MAX_Z / 4)) /* 2496 */
#define LDPC_PART_STRIDE (LDPC_PART_ITEMS_MAX + 8)             /* + header words: [0] unsatisfied lanes of the check-node
                                                                   phase, [1] of the sweep after the pass */
#define LDPC_PART_MAX_ITEMS_PER_THREAD 5                       /* ceil(2496 / 512): workgroups of 512 threads or more */
#define LDPC_PART_MAX_PARTS 4
#define LDPC_PART_MEET_TIMEOUT_TICKS 2000000ll                 /* 20 ms of the 100 MHz clock */

/* The exchange area is device (global) memory; through a generic pointer the compiler emits FLAT instructions, whose
 * completion is counted together with the LDS traffic -- every LDS wait in the gather loop then also waits for the last
 * published store to reach memory.  Typed as global, they are independent of the LDS pipe. */
typedef __attribute__((address_space(1))) unsigned long long ldpc_gu64;
typedef __attribute__((address_space(1))) unsigned int ldpc_gu32;
__device__ __forceinline__ void ldpc_pub64(unsigned long long *p, unsigned long long v)
{
  __hip_atomic_store((ldpc_gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ldpc_get64(const unsigned long long *p)
{
  return __hip_atomic_load((ldpc_gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* all parts of the slot meet: every store this workgroup made before is visible to the others afterwards (they were
 * agent-scope stores, complete before the arrival is counted) */
template <class IO> __device__ __forceinline__ void ldpc_part_meet(const IO &io)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (io.tid() == 0) {
    const uint32_t target = *io.meet_target() + (uint32_t)io.parts();
    *io.meet_target() = target;
    ldpc_gu32 *cnt = (ldpc_gu32 *)io.meet_counter();
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t0 > LDPC_PART_MEET_TIMEOUT_TICKS) { /* never hang the GPU on a protocol error */
        *io.fault() = 1u;
        break;
      }
    }
  }
  __syncthreads();
}

template <class IO>
__device__ __forceinline__ int ldpc_dec_fast_part(uint8_t *fsm, ldpc_code_ptr_t code, const IO &io)
{
  const int Z = code->Z, zq = code->f_zq, rstride = code->f_rstride, astride = code->f_astride;
  const uint32_t zq_magic = code->f_zq_magic;
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
  L.ext_global = 0;
  int *flags = reinterpret_cast<int *>(fsm + code->f_lds_misc); /* [0..15] as in ldpc_dec_fast_block.h; [16 + sc]: full degree
                                                                    of sorted column sc */
  const int tid = io.tid(), nt = blockDim.x, lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
  const int ncore = code->ncore, nedges = code->nedges;
  const int part = io.part(), parts = io.parts();
  const uint32_t *__restrict__ src32 = io.src32();
  const uint32_t *__restrict__ srcp = io.src32_prologue();
  uint32_t *stage = io.stage_core();
  L.gllr = reinterpret_cast<const uint8_t *>(src32);

  /* ---- tables and state into LDS ------------------------------------------------------------------------ */
  const uint32_t lds0 = ldpc_lds_addr(fsm);
  const int n_app = ncore * zq, n_ext = (code->ncols - ncore) * zq;
  if (!io.tables_resident()) {
    for (int i = tid; i < nedges; i += nt)
      etbl[i] = code->f_etbl[i] + lds0;
    for (int i = tid; i < 2 * code->f_n_ctbl; i += nt)
      ctbl[i] = code->f_ctbl[i] + ((i & 1) ? lds0 : 0u);
    for (int i = tid; i < code->nrows; i += nt)
      rowtbl[i] = code->f_rowtbl[i];
    for (int i = tid; i < ncore; i += nt) {
      const uint32_t cr = code->f_coltbl[i];
      coltbl[i] = cr;
      flags[16 + i] = code->col_deg_full[cr & 0xffu];
    }
    for (int i = tid; i < (Z + 4) >> 2; i += nt)
      reinterpret_cast<uint32_t *>(fsm + code->f_lds_zero)[i] = 0u;
  }
  if (tid < 8)
    flags[tid] = 0;
  /* APP := channel LLR of ALL columns (both copies); the extension columns' LLRs; a device copy of the core columns for
   * the per-pass re-reads (this part's own row of the staging area) */
  uint32_t *e32 = reinterpret_cast<uint32_t *>(L.ext);
  /* (the loads of a batch are issued together: each is a trip to memory, and a loop of load-use-load pays it every time) */
  for (int i0 = 0; i0 < n_app + n_ext; i0 += 8 * nt) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = i0 + tid + k * nt;
      v[k] = i < n_app + n_ext ? io.ld_llr(srcp + i) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int i = i0 + tid + k * nt;
      if (i < n_app) {
        const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
        const uint32_t w = v[k] ^ 0x80808080u;
        uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride) + j;
        dst[0] = w;
        dst[zq] = w;
        stage[i] = v[k];
      } else if (i < n_app + n_ext) {
        e32[i - n_app] = v[k] ^ 0x80808080u;
      }
    }
  }
  __syncthreads();
  if (io.stamps() && tid == 0)
    io.stamps()[0] = (uint32_t)wall_clock64();

  /* ---- passes ------------------------------------------------------------------------------------------ */
  const int max_pass = io.max_pass();
  int n_iter = max_pass;
  const int n_cn_tasks = code->f_n_cn_tasks, n_bn_tasks = code->f_n_bn_tasks, n_items = ncore * zq;
  unsigned long long *const ex = io.exch();
  uint32_t *const ph = io.phase_ticks();
  uint32_t t_ph = (uint32_t)wall_clock64();
#define LDPC_PART_PHASE(k) do { if (ph && tid == 0) { const uint32_t t_ = (uint32_t)wall_clock64(); ph[k] += t_ - t_ph; t_ph = t_; } } while (0)
  if (ph && tid < 5)
    ph[tid] = 0;
  for (int p = 1; p <= max_pass; ++p) {
    uint32_t syn = 0;
    for (;;) {
      const int task = ldpc_draw(&flags[4], lane);
      if (task >= n_cn_tasks)
        break;
      const int deg = code->f_cn_task[task][0], ext = code->f_cn_task[task][1];
      const int item = code->f_cn_task[task][2] + lane;
      const int gstart = code->f_cn_task[task][3], gend = code->f_cn_task[task][4], srow0 = code->f_cn_task[task][5];
      if (item < gend) {
        /* (a whole code's descriptor, served here as a single part, may pair its degree-19 items: ldpc_graph.h f_pair19) */
        const bool pair = deg == 19 && code->f_pair19;
        const int gi = pair ? (item - gstart) >> 1 : item - gstart, half = (item - gstart) & 1;
        const int rig = (int)ldpc_umulhi((uint32_t)gi, zq_magic), j = gi - rig * zq;
        const uint32_t rowrec = rowtbl[srow0 + rig];
        const int e0 = (int)(rowrec & 0x1ffu), valid = (int)(rowrec >> 16) - 4 * j;
        uint32_t m;
        (void)half;
#if defined(__HIP_DEVICE_COMPILE__)
        if (pair)
          m = p == 1 ? ldpc_fast_cn19_pair<true>(L, e0, j, Z, rstride, half) : ldpc_fast_cn19_pair<false>(L, e0, j, Z, rstride, half);
        else
#endif
          m = p == 1 ? ldpc_fast_cn_dispatch<true>(deg, ext, L, e0, j, Z, rstride)
                     : ldpc_fast_cn_dispatch<false>(deg, ext, L, e0, j, Z, rstride);
        const uint32_t mask = valid >= 4 ? 0xfu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
        syn |= m & mask;
      }
    }
    {
      const unsigned long long bad_lanes = __ballot(syn != 0);
      if (bad_lanes && lane == 0)
        atomicAdd(&flags[p & 1], (int)__popcll(bad_lanes));
    }
    if (tid == 0)
      flags[2] = 0;
    __syncthreads();
    LDPC_PART_PHASE(0);
    /* bit-node phase, first half: this part's edges.  Items are dealt statically -- task w, w + n_waves, ... to wave w --
     * so that a thread still has its sums when the others' arrive. */
    unsigned long long *mine = ex + ((size_t)(p & 1) * parts + part) * LDPC_PART_STRIDE;
    uint32_t pe[LDPC_PART_MAX_ITEMS_PER_THREAD], po[LDPC_PART_MAX_ITEMS_PER_THREAD];
#pragma unroll
    for (int k = 0; k < LDPC_PART_MAX_ITEMS_PER_THREAD; k++) {
      const int task = wave + k * n_waves;
      pe[k] = po[k] = 0;
      if (task < n_bn_tasks) {
        const int item = task * 64 + lane, maxdeg = code->f_bn_task[task][2];
        if (item < n_items) {
          const int sc = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sc * zq;
          const uint32_t colrec = coltbl[sc];
          ldpc_fast_bn_gather(L, colrec, maxdeg, j, Z, 0, pe[k], po[k]);
          if (parts > 1)
            ldpc_pub64(mine + (int)(colrec & 0xffu) * zq + j, (unsigned long long)pe[k] | ((unsigned long long)po[k] << 32));
        }
      }
    }
    LDPC_PART_PHASE(1);
    /* (parts == 1: the whole code in this workgroup -- small codes served by the same kernel: nothing to exchange) */
    if (parts > 1) {
      if (tid == 0)
        ldpc_pub64(mine + LDPC_PART_ITEMS_MAX, (unsigned long long)(unsigned int)flags[p & 1]);
      ldpc_part_meet(io);
    }
    LDPC_PART_PHASE(2);
    /* what the others published: all loads go out together (each is a trip past the caches), then the arithmetic */
    unsigned long long hv[LDPC_PART_MAX_PARTS], dv[LDPC_PART_MAX_ITEMS_PER_THREAD][LDPC_PART_MAX_PARTS];
    const unsigned long long *const exp = ex + (size_t)(p & 1) * parts * LDPC_PART_STRIDE;
#pragma unroll
    for (int q = 0; q < LDPC_PART_MAX_PARTS; q++)
      hv[q] = (parts > 1 && q < parts)
                  ? ldpc_get64(exp + (size_t)q * LDPC_PART_STRIDE + LDPC_PART_ITEMS_MAX)
                  : 0ull;
    if (parts == 1)
      hv[0] = (unsigned long long)(unsigned int)flags[p & 1];
    int cidx[LDPC_PART_MAX_ITEMS_PER_THREAD];
#pragma unroll
    for (int k = 0; k < LDPC_PART_MAX_ITEMS_PER_THREAD; k++) {
      const int task = wave + k * n_waves, item = task * 64 + lane;
      cidx[k] = -1;
      if (task < n_bn_tasks && item < n_items) {
        const int sc = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sc * zq;
        cidx[k] = (int)(coltbl[sc] & 0xffu) * zq + j;
      }
#pragma unroll
      for (int q = 0; q < LDPC_PART_MAX_PARTS; q++)
        dv[k][q] = (cidx[k] >= 0 && q < parts && q != part)
                       ? ldpc_get64(exp + (size_t)q * LDPC_PART_STRIDE + cidx[k])
                       : 0ull;
    }
    /* everybody's count of unsatisfied lanes after pass p - 1 (after the channel's hard decisions for p = 1) */
    int bad_prev = 0;
#pragma unroll
    for (int q = 0; q < LDPC_PART_MAX_PARTS; q++)
      bad_prev += (int)hv[q];
    bad_prev = LDPC_UNIFORM(bad_prev);
    if (!io.use_crc() && p >= 3 && bad_prev == 0) {
      n_iter = p - 1;
      break;
    }
    /* second half: the others' sums, the channel LLR, clamp (the APPs of all columns, in every part) */
#pragma unroll
    for (int k = 0; k < LDPC_PART_MAX_ITEMS_PER_THREAD; k++) {
      if (cidx[k] >= 0) {
        const int item = (wave + k * n_waves) * 64 + lane;
        const int sc = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sc * zq;
        const int c = (int)(coltbl[sc] & 0xffu);
        uint32_t te = pe[k], to = po[k];
#pragma unroll
        for (int q = 0; q < LDPC_PART_MAX_PARTS; q++) {
          te += (uint32_t)dv[k][q];
          to += (uint32_t)(dv[k][q] >> 32);
        }
        ldpc_fast_bn_finish(L, c, flags[16 + sc], j, Z, astride, src32[cidx[k]], te, to, 0);
      }
    }
    if (tid == 0) {
      flags[(p + 1) & 1] = 0;
      flags[4] = 0; /* nobody draws check-node tasks now */
    }
    __syncthreads();
    LDPC_PART_PHASE(3);
    if (io.eager_check() && !io.use_crc() && p >= 2 && p < max_pass && bad_prev <= LDPC_EAGER_MAX_BAD_LANES) {
      /* the check the next pass would make first thing, right away (ldpc_dec_fast_block.h): every part sweeps its rows,
       * the counts meet in a header-only exchange */
      uint32_t esyn = 0;
      const int n_row_items = code->nrows * zq;
      for (int item = tid; item < n_row_items; item += nt) {
        const int sr = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sr * zq;
        const uint32_t rowrec = rowtbl[sr];
        const int e0 = (int)(rowrec & 0x1ffu), deg = (int)((rowrec >> 9) & 0x1fu), ext = (int)((rowrec >> 14) & 1u);
        const int valid = (int)(rowrec >> 16) - 4 * j;
        const uint32_t m = ldpc_fast_pc(L, deg, ext, e0, j, rstride);
        const uint32_t mask = valid >= 4 ? 0xfu : (valid <= 0 ? 0u : ((1u << valid) - 1u));
        esyn |= m & mask;
      }
      if (__any(esyn != 0) && lane == 0)
        flags[6] = 1;
      __syncthreads();
      const int bad_own = flags[6];
      __syncthreads();
      if (tid == 0) {
        if (parts > 1)
          ldpc_pub64(mine + LDPC_PART_ITEMS_MAX + 1, (unsigned long long)(unsigned int)bad_own);
        flags[6] = 0;
      }
      if (parts > 1)
        ldpc_part_meet(io);
      int bad = parts > 1 ? 0 : bad_own;
#pragma unroll
      for (int q = 0; q < LDPC_PART_MAX_PARTS; q++)
        bad += (parts > 1 && q < parts) ? (int)ldpc_get64(exp + (size_t)q * LDPC_PART_STRIDE + LDPC_PART_ITEMS_MAX + 1) : 0;
      LDPC_PART_PHASE(4);
      if (LDPC_UNIFORM(bad) == 0) {
        n_iter = p;
        break;
      }
    }
    if (io.use_crc() && p >= 3) { /* every part on its own copy of the APPs: same verdict everywhere */
      uint32_t x = 0;
      const int crcE = io.crcE();
      const uint32_t *crc_pow = io.crc_pow();
      for (int i = 4 * tid; i < crcE; i += 4 * nt) {
        const int c = (int)ldpc_umulhi((uint32_t)i, z_magic), u = i - c * Z;
        const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u);
        const uint4 pw = *reinterpret_cast<const uint4 *>(crc_pow + (crcE - 4 - i));
        x ^= (pw.w & (0u - ((nb >> 7) & 1u))) ^ (pw.z & (0u - ((nb >> 15) & 1u))) ^ (pw.y & (0u - ((nb >> 23) & 1u))) ^
             (pw.x & (0u - (nb >> 31)));
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
  if (io.stamps() && tid == 0)
    io.stamps()[1] = (uint32_t)wall_clock64();

  /* ---- hard decision: part 0, in tagged 16-byte units (ldpc_dec_fast_block.h) --------------------------------- */
  if (part == 0 && (!io.use_crc() || n_iter >= 3) && n_iter <= max_pass) {
    const int num_llr = code->num_llr, ncz = ncore * Z;
    const uint32_t tag = io.out_tag();
    const int mode = io.out_mode();
    const int nwords = mode == 0 ? (num_llr + 31) >> 5 : num_llr >> 2;
    uint4 *o16 = reinterpret_cast<uint4 *>(io.out());
    for (int c3 = tid; 3 * c3 < nwords; c3 += nt) {
      uint32_t w3[3];
#pragma unroll
      for (int q3 = 0; q3 < 3; q3++) {
        const int w = 3 * c3 + q3;
        uint32_t word = 0;
        if (w < nwords) {
          if (mode == 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
              const int b = 32 * w + 4 * q;
              if (b < ncz) {
                const int c = (int)ldpc_umulhi((uint32_t)b, z_magic), u = b - c * Z;
                const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u);
                const uint32_t nib = (((nb >> 7) & 1u) << 3) | (((nb >> 15) & 1u) << 2) | (((nb >> 23) & 1u) << 1) | (nb >> 31);
                word |= nib << (8 * (q >> 1) + ((q & 1) ? 0 : 4));
              }
            }
          } else {
            const int b = 4 * w;
            if (b < ncz) {
              const int c = (int)ldpc_umulhi((uint32_t)b, z_magic), u = b - c * Z;
              word = ((~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u)) >> 7) & 0x01010101u;
            }
          }
        }
        w3[q3] = word;
      }
      io.put16(o16 + c3, w3[0], w3[1], w3[2], tag);
    }
  }
  return n_iter;
#undef LDPC_PART_PHASE
}
#endif
