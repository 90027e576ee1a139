/*
 * ldpc_dec_fast_block.h -- one code block through the "fast" flooding min-sum decoder, executed by one workgroup
 * (device code).  Shared by the batch kernel (ldpc_decoder_fast.hip: one workgroup per block of a launch) and by the
 * resident server kernel (ldpc_server.hip: one workgroup per caller slot, blocks arrive through a mailbox).
 *
 * Contract = nrLDPC_decoder_core (reference openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_decoder.c:206-880); per-thread
 * arithmetic in ldpc_dec_fast_core.h.  Everything of the block lives in the workgroup's LDS between the LLR load and
 * the bit store.  The function contains workgroup barriers: every thread of the workgroup must call it with the same
 * (wave-uniform) arguments.  Any workgroup size that is a multiple of 64 works: the loops stride by blockDim.x and the
 * tasks of a phase are drawn from a queue.
 */
#ifndef LDPC_DEC_FAST_BLOCK_H
#define LDPC_DEC_FAST_BLOCK_H
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "ldpc_dec_fast_core.h"

typedef const ldpc_dec_job LDPC_CONST_AS *ldpc_job_ptr_t; /* job records are read through the constant address space */

/* The caller's view of a block is a type IO with these (wave-uniform) accessors, evaluated where the value is needed --
 * not up front -- so that what is only used after the last pass (output row, CRC table) or once per pass (abort flag)
 * is not carried in registers through the check-node loops (the kernel sits at the VGPR limit and already keeps part of
 * its scalar state in VGPR lanes):
 *   const uint32_t *src32()   the block's channel LLRs: ncols*Z int8, 4-byte aligned, device memory (re-read every pass)
 *   const uint32_t *src32_prologue()   where the prologue reads them (the same, or host memory in the server path)
 *   uint32_t *stage_core()    nullptr, or: the prologue also writes the core columns' LLRs there (= src32(), a device
 *                             copy of what it read from host memory), made visible before the first bit-node phase
 *   int8_t *out()             output row (packed bits: 4-byte aligned)
 *   int max_pass()            numMaxIter + 1
 *   int use_crc(), crcE()     CRC stop mode and the bits it covers
 *   const uint32_t *crc_pow() x^j mod g, left aligned
 *   int out_mode()            0 packed bits, else one bit per byte
 *   int *tb_abort()           optional: transport-block wide "a segment failed" flag (decoder.c:190-193, 556-559)
 *   uint32_t abort_load()     resident server: the slot's "caller gave up" word, loaded (past the caches) at the START of a
 *   bool has_abort()          compile-time: does this caller have any abort source (tb_abort / abort_load)?
 *   bool abort_is(word)       pass and looked at after its check-node phase -- the reference's per-iteration check_abort
 *                             (decoder.c:556-559) for a caller whose `ab` is raised by another thread while the call is in
 *                             flight; 0 / false elsewhere
 *   uint32_t *stamps()        optional (LDS): wall_clock64 after the prologue and after the last pass (diagnostics)
 *   static bool pass_stamps   diagnostic instantiation: stamps() is a row of 54 words in memory (zeroed by the launcher) and
 *                             also takes [2p], [2p + 1] the clocks behind pass p's check-node / bit-node barrier, [20] the
 *                             passes that ran the eager parity sweep (bit p), [22 + p] the unsatisfied lanes counted by pass p
 *   int tid()                 threadIdx.x
 *   uint32_t ld_llr(p)        one dword of src32_prologue()'s row: a plain load, or one that bypasses the caches when the
 *                             row was written by the host while this kernel was running (resident server)
 *   uint32_t out_tag()        0: the output row is written as it is.  Otherwise (resident server): the row goes out in
 *                             16-byte units {three output dwords, tag}, one store each, so that the host can tell unit by
 *                             unit that it has arrived and no fence or completion ordering is needed; put16(p, a, b, c,
 *                             tag) is that store (it must leave the caches by itself)
 *   bool tables_resident()    the code's tables are still in this workgroup's LDS from the previous block (resident
 *                             server, same code as last time): the prologue does not copy them again
 *   static bool syndrome      false: the launch stops on the CRC only, the parity of the hard decisions is not computed
 *   static bool mute_items    with syndrome == false: when io.mute_check() says so for this block, extension-row items whose
 *   bool mute_check()         degree-1 bits have channel LLR 0 are not run (ldpc_fast_item_is_mute)
 *   static bool bn_tickets    bit-node queue by tickets (f_bn_ticket: short tasks grouped) or by tasks (f_bn_group)
 *   static bool tb_epilogue   the caller may end a block with the transport-block chain's epilogue (tb_rx_fused.hip):
 *   bool tb_fused()           ... and does so for this block: instead of an output row, io.tb_finish(n_iter, bits_word,
 *                             flags) delivers the segment's payload bytes, its share of the TB CRC and -- from the last
 *                             segment of a transport block to finish -- the block's verdict
 *   int fair_turns()          0, or: the workgroup shares its CU with others of its kind and has this many waves per SIMD -- it
 *                             takes turns with them at the issue priority, pass by pass (see below)
 *   bool eager_check()        latency path: evaluate the parity check of a pass in a sweep of its own right after the
 *                             pass, instead of folding it into the next pass' check-node phase (which costs a whole
 *                             check-node phase when the block has converged); same results, same pass counts */

#ifndef LDPC_MUTE_ITEMS
#define LDPC_MUTE_ITEMS 1 /* A/B: tools/build_variant.sh <name> -DLDPC_MUTE_ITEMS=0 */
#endif
#define LDPC_EAGER_MAX_BAD_LANES 96
#define LDPC_TIMING_SLOTS 28

/* next ticket of a task queue (wave-uniform) */
__device__ __forceinline__ int ldpc_draw(int *counter, int lane)
{
  int t = 0;
  if (lane == 0)
    t = atomicAdd(counter, 1);
  return LDPC_UNIFORM(t);
}

/* the same in two halves: the request (an LDS atomic that returns its value whenever) and its use */
__device__ __forceinline__ int ldpc_draw_issue(int *counter, int lane)
{
  int t = 0;
  if (lane == 0)
    t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return t;
}

/* The code's tables, made absolute for this workgroup's LDS, and the row of zero bytes the padded bit-node lists point at:
 * [f_lds_etbl, f_lds_zero + Z + 4).  Done by the block body's prologue unless its caller says they are there already
 * (io.tables_resident(): the resident server's "same code as last time", the fused segment kernel, which copies them while its
 * de-matching stores drain). */
__device__ __forceinline__ void ldpc_fast_tables_to_lds(uint8_t *fsm, ldpc_code_ptr_t code, int tid, int nt)
{
  const uint32_t lds0 = ldpc_lds_addr(fsm); /* tables hold absolute LDS addresses from here on */
  const int ext_global = code->f_ext_global, ncore = code->ncore, nedges = code->nedges, Z = code->Z;
  uint32_t *etbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_etbl);
  uint32_t *ctbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_ctbl);
  uint32_t *rowtbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_rowtbl);
  uint32_t *coltbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_coltbl);
  for (int i = tid; i < nedges; i += nt)
    etbl[i] = code->f_etbl[i] + ((ext_global && code->e_col[i] >= ncore) ? 0u : lds0);
  for (int i = tid; i < 2 * code->f_n_ctbl; i += nt)
    ctbl[i] = code->f_ctbl[i] + ((i & 1) ? lds0 : 0u);
  for (int i = tid; i < code->nrows; i += nt)
    rowtbl[i] = code->f_rowtbl[i];
  for (int i = tid; i < ncore; i += nt)
    coltbl[i] = code->f_coltbl[i];
  for (int i = tid; i < (Z + 4) >> 2; i += nt)
    reinterpret_cast<uint32_t *>(fsm + code->f_lds_zero)[i] = 0u;
}

/* CRC-stop launches only (IO::syndrome == false).  An extension row's item whose four degree-1 bits have channel LLR 0 -- a
 * position no transmission has reached -- sends zeros to its core neighbours whatever they say (the minimum over the OTHER inputs
 * includes that 0, nrLDPC_cnProc.h:105-114), pass after pass; what it sends to the degree-1 bit itself feeds nothing but the row's
 * own parity check, which a CRC-stop launch does not evaluate.  Such an item writes its zeros in the first pass and is not run
 * again.  (Whole rows of them at the end of a rate mode are cut from the graph beforehand, ldpc_graph.h LDPC_R_COLS; this catches
 * the column a transmission ends in, the holes a retransmission leaves, and callers the plan knows nothing about.) */
__device__ __forceinline__ bool ldpc_fast_item_is_mute(const ldpc_fast_lds &L, int e0, int deg, int j)
{
  const uint32_t info = L.etbl[e0 + deg - 1];
  const uint32_t lw = L.ext_global ? (*reinterpret_cast<const uint32_t *>(L.gllr + info + 4u * (uint32_t)j) ^ 0x80808080u)
                                   : ldpc_lds_ld32(L.base, info + 4u * (uint32_t)j);
  return lw == 0x80808080u;
}
__device__ __forceinline__ void ldpc_fast_item_zero_messages(const ldpc_fast_lds &L, int e0, int deg, int j, int Z, int rstride)
{
  uint8_t *rrow = L.r + e0 * rstride + 4 * j;
  uint8_t *rpad = rrow + (j == 0 ? Z : 0);
  for (int k = 0; k < deg; k++) {
    *reinterpret_cast<uint32_t *>(rrow + k * rstride) = 0x80808080u;
    *reinterpret_cast<uint32_t *>(rpad + k * rstride) = 0x80808080u;
  }
}

/* Returns the pass count as LDPCdecoder reports it (numMaxIter + 2: the transport block was given up, decoder.c:556-559). */
/* ZC != 0: the instantiation for ONE lifting size (the launcher guarantees code->Z == ZC).  With the row strides known at compile
 * time the k-th message of an item is `ds_read_b32 v, base offset:k * (ZC + 4)` -- the general build forms base + k * rstride with
 * a VALU add per edge for the read / first store and another one for the wrap-around store (rstride sits in an SGPR), two of the
 * check-node body's ~26 VALU instructions per edge, and keeps those addresses in a register each. */
template <class IO, int ZC = 0>
__device__ __forceinline__ int ldpc_dec_fast_block(uint8_t *fsm, ldpc_code_ptr_t code, const IO &io)
{
  const int Z = ZC ? ZC : code->Z, zq = ZC ? ZC / 4 : code->f_zq, rstride = ZC ? ZC + 4 : code->f_rstride,
            astride = ZC ? 2 * ZC : code->f_astride;
  const uint32_t zq_magic = ZC ? (uint32_t)((0x100000000ULL + (unsigned long long)(ZC ? ZC / 4 : 1) - 1) / (unsigned long long)(ZC ? ZC / 4 : 1))
                               : code->f_zq_magic;
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
  const int tid = io.tid(), nt = blockDim.x, lane = tid & 63;
  const int ncore = code->ncore, num_llr = code->num_llr, ncz = ncore * Z;
  const uint32_t *__restrict__ src32 = io.src32();
  const uint32_t *__restrict__ srcp = io.src32_prologue();
  uint32_t *stage = io.stage_core();

  /* ---- tables and state into LDS -------------------------------------------------------------------- */
  const int ext_global = code->f_ext_global;
  L.gllr = reinterpret_cast<const uint8_t *>(src32);
  L.ext_global = ext_global;
  /* The block's LLRs: the first four dwords per thread of the core and of the extension columns are requested early and
   * consumed after the message initialisation, so that their latency runs in the background (a 1024-thread workgroup
   * needs 3 + 4 such loads per thread for Zc = 384).  From HBM they go out before anything else; from host memory (the
   * server path, `stage`) only after the table copies -- loads retire in order, and the tables' device loads would
   * otherwise sit behind a trip over the link. */
  const int n_app = ncore * zq, n_ext = ext_global ? 0 : (code->ncols - ncore) * zq;
  uint32_t va[4], ve[4];
  if (!stage) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ia = tid + k * nt;
      va[k] = ia < n_app ? io.ld_llr(srcp + ia) : 0u;
      ve[k] = ia < n_ext ? io.ld_llr(srcp + n_app + ia) : 0u;
    }
  }
  const bool have_tables = io.tables_resident();
  if (!have_tables)
    ldpc_fast_tables_to_lds(fsm, code, tid, nt);
  if (stage) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ia = tid + k * nt;
      va[k] = ia < n_app ? io.ld_llr(srcp + ia) : 0u;
      ve[k] = ia < n_ext ? io.ld_llr(srcp + n_app + ia) : 0u;
    }
  }
  if (tid < 16)
    flags[tid] = 0; /* [0], [1] syndrome flags of odd / even passes, [2] CRC register, [3] TB abort seen, [6] eager check,
                       [8], [9] task queues of even / odd passes */
  /* (the message array is not initialised: the first check-node phase takes r = 0 without reading it and writes every
   * message, wrap-around bytes included)
   * APP := channel LLR (both copies), so that with r = 0 the first check-node phase sees q = llr */
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
      if (stage)
        stage[i] = va[k];
    }
    if (i < n_ext)
      e32[i] = ve[k] ^ 0x80808080u;
  }
  for (int i = tid + 4 * nt; i < n_app; i += nt) { /* small workgroups: the rest */
    const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
    const uint32_t v = io.ld_llr(srcp + i), w = v ^ 0x80808080u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride) + j;
    dst[0] = w;
    dst[zq] = w;
    if (stage)
      stage[i] = v;
  }
  for (int i = tid + 4 * nt; i < n_ext; i += nt)
    e32[i] = io.ld_llr(srcp + n_app + i) ^ 0x80808080u;
  __syncthreads();
  if (io.stamps() && tid == 0)
    io.stamps()[0] = (uint32_t)wall_clock64();

  /* ---- passes ------------------------------------------------------------------------------------------ */
  const int max_pass = io.max_pass();
  int n_iter = max_pass;
  const int n_cn_tasks = code->f_n_cn_tasks, n_bn_tickets = code->f_n_bn_tickets, n_bn_tasks = code->f_n_bn_tasks,
            bn_group = code->f_bn_group;
  int cn_ticket = 0;
  bool mute_check = false; /* (wave-uniform: the job's flag) */
  if constexpr (!IO::syndrome && IO::mute_items)
    mute_check = io.mute_check();
  (void)mute_check;
#ifdef LDPC_TIMING
  /* diagnostic build (tools/task_timing.sh): block 0 logs, for pass 2, every task of every wave into its (oversized)
   * output row as {start << 20 | phase << 8 | degree or loop bound, end} in shader clocks */
  long long *tlog = reinterpret_cast<long long *>(io.out()) + (tid >> 6) * 2 * LDPC_TIMING_SLOTS;
  int tlog_n = 0;
  const long long t_kernel0 = clock64();
#define LDPC_TLOG_BEGIN() const long long tl0_ = clock64()
#define LDPC_TLOG_END(phase, deg) \
  do { \
    if (blockIdx.x == 0 && p == 2 && lane == 0 && tlog_n < LDPC_TIMING_SLOTS) { \
      tlog[2 * tlog_n] = ((tl0_ - t_kernel0) << 20) | ((long long)(phase) << 8) | (long long)(deg); \
      tlog[2 * tlog_n + 1] = clock64() - t_kernel0; \
      tlog_n++; \
    } \
  } while (0)
#define LDPC_TSTAMP(id) \
  do { \
    LDPC_TLOG_BEGIN(); \
    LDPC_TLOG_END(3, id); \
  } while (0)
#else
#define LDPC_TLOG_BEGIN() do { } while (0)
#define LDPC_TLOG_END(phase, deg) do { } while (0)
#define LDPC_TSTAMP(id) do { } while (0)
#endif
  /* Workgroups that share a CU: the SIMDs' issue arbiter serves the OLDEST wave first, so of two co-resident workgroups the
   * one that arrived first runs at the speed it would have alone (8.7 us per pass for BG1 Zc = 384 R = 2/3) and the other one at
   * what is left (14 us) -- until the first one leaves and its successor is the younger one (profiles/r06/README.md).  The CU's
   * throughput is the same either way, but the last workgroup of a CU always ends up finishing alone on a half-used CU.  Taking
   * turns: a wave's slot on its SIMD tells which of the CU's workgroups it belongs to, and a workgroup raises its priority in
   * the passes whose number matches its slot -- two workgroups that run side by side then alternate, and one that has fallen a
   * pass behind holds the high priority until it has caught up. */
  const int fair = io.fair_turns();
  int prio_slot = 0;
  if (fair)
    prio_slot = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 15u) / fair; /* HW_REG_HW_ID.wave_id / waves per SIMD and workgroup */
  for (int p = 1; p <= max_pass; ++p) {
    uint32_t syn = 0;
    if (fair) {
      if ((p + prio_slot) & 1)
        __builtin_amdgcn_s_setprio(1);
      else
        __builtin_amdgcn_s_setprio(0);
    }
    const uint32_t ab_word = (io.has_abort() && tid == 0 && p >= 2) ? io.abort_load() : 0u; /* in flight during the check-node phase */
    /* likewise the transport block's flag (a load that leaves the caches: its latency would otherwise sit between the
     * check-node phase and the barrier, once per pass) */
    int tb_ab = 0;
    if (io.has_abort() && tid == 0 && p >= 2 && io.tb_abort())
      tb_ab = __hip_atomic_load(io.tb_abort(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    /* this pass' task queue: tickets 0 .. n_cn_tasks - 1 are the check-node tasks, the ones behind them the bit-node
     * tickets -- ONE counter, so that the draw that finds the check-node tasks gone is already the draw of a bit-node
     * ticket (a draw is an LDS atomic round trip on a busy CU: 250-300 clocks, profiles/r04/timeline_phase_switch.txt) */
    int *const pq = &flags[8 + (p & 1)];
#ifdef LDPC_ABLATE_CN
    syn = 1;
#else
    /* The phase's tasks are drawn in id order (= most expensive first, ldpc_graph.c) from a queue -- an LDS counter --
     * by whichever wave is free: the SIMD issue arbiter favours a CU's older waves, so static equal shares leave the
     * SIMDs with one or two live waves for the last third of a phase (profiles/r01/task_timeline.txt). */
    /* The first ticket of a phase is drawn BEFORE the barrier in front of it (at the end of the other phase; the queues
     * of odd and even passes are different counters, each reset two barriers ahead of its first draw), so that a wave
     * comes out of the barrier with its task in hand: the draw is an LDS atomic round trip, and right after a barrier
     * nobody has work to hide it behind (profiles/r03/timeline_*.txt: 1.3 k clocks between the phases). */
    if (p == 1)
      cn_ticket = ldpc_draw(pq, lane);
    int task = cn_ticket;
    for (; task < n_cn_tasks; task = ldpc_draw(pq, lane)) {
      LDPC_TLOG_BEGIN();
      const int deg_f = code->f_cn_task[task][0], ext = code->f_cn_task[task][1];
      const int item = code->f_cn_task[task][2] + lane;
      const int gstart = code->f_cn_task[task][3], gend = code->f_cn_task[task][4], srow0 = code->f_cn_task[task][5];
      const int deg = deg_f & 0xff;
      if (deg_f & 0x100) {
        /* a double task (ldpc_graph.h f_cn_task): items lane and lane + 64 of 128 consecutive ones, walked together; a
         * thread without a second item takes its first one twice (the same bytes are stored twice) */
        if (item < gend) {
          const int item_b = item + 64 < gend ? item + 64 : item;
          const int gia = item - gstart, gib = item_b - gstart;
          const int riga = (int)ldpc_umulhi((uint32_t)gia, zq_magic), ja = gia - riga * zq;
          const int rigb = (int)ldpc_umulhi((uint32_t)gib, zq_magic), jb = gib - rigb * zq;
          const uint32_t reca = rowtbl[srow0 + riga], recb = rowtbl[srow0 + rigb];
          const int e0a = (int)(reca & 0x1ffu), e0b = (int)(recb & 0x1ffu);
          const int valida = (int)(reca >> 16) - 4 * ja, validb = (int)(recb >> 16) - 4 * jb;
          uint32_t ma = 0, mb = 0;
          bool mute = false;
          if constexpr (!IO::syndrome && IO::mute_items)
            mute = mute_check && ldpc_fast_item_is_mute(L, e0a, deg, ja) && ldpc_fast_item_is_mute(L, e0b, deg, jb);
          if (mute) {
            if (p == 1) {
              ldpc_fast_item_zero_messages(L, e0a, deg, ja, Z, rstride);
              ldpc_fast_item_zero_messages(L, e0b, deg, jb, Z, rstride);
            }
          } else if (p == 1)
            (void)ldpc_fast_cn2_dispatch<true>(deg, L, e0a, ja, e0b, jb, Z, rstride, mb);
          else
            ma = ldpc_fast_cn2_dispatch<false>(deg, L, e0a, ja, e0b, jb, Z, rstride, mb);
          const uint32_t maska = ldpc_fast_valid_lanes(valida), maskb = ldpc_fast_valid_lanes(validb);
          if constexpr (IO::syndrome)
            syn |= (ma & maska) | (mb & maskb);
          else
            (void)ma, (void)mb, (void)maska, (void)maskb;
        }
      } else if (item < gend) {
        /* degree-19 rows: an item is shared by two neighbouring lanes (ldpc_graph.h f_pair19) */
        const bool pair = deg == 19 && code->f_pair19;
        const int gi = pair ? (item - gstart) >> 1 : item - gstart, half = (item - gstart) & 1;
        const int rig = (int)ldpc_umulhi((uint32_t)gi, zq_magic), j = gi - rig * zq;
        const uint32_t rowrec = rowtbl[srow0 + rig];
        const int e0 = (int)(rowrec & 0x1ffu), valid = (int)(rowrec >> 16) - 4 * j; /* lanes t+i < pc_lo are checked */
        /* pass 1: the messages are all 0 and are not read (nor initialised by the prologue); what its check-node phase
         * returns -- the parity of the CHANNEL's hard decisions -- is looked at by nobody (stops are decided from pass 3
         * on, the eager check from pass 2 on), so it is not accumulated and the bodies drop it as dead code */
        uint32_t m = 0;
        (void)half;
        bool mute = false;
        if constexpr (!IO::syndrome && IO::mute_items)
          mute = mute_check && ext && ldpc_fast_item_is_mute(L, e0, deg, j);
        if (mute) {
          if (p == 1)
            ldpc_fast_item_zero_messages(L, e0, deg, j, Z, rstride);
        } else if (p == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
          if (pair)
            (void)ldpc_fast_cn19_pair<true>(L, e0, j, Z, rstride, half);
          else
#endif
            (void)ldpc_fast_cn_dispatch<true>(deg, ext, L, e0, j, Z, rstride);
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
          if (pair)
            m = ldpc_fast_cn19_pair<false>(L, e0, j, Z, rstride, half);
          else
#endif
            m = ldpc_fast_cn_dispatch<false>(deg, ext, L, e0, j, Z, rstride);
        }
        const uint32_t mask = ldpc_fast_valid_lanes(valid);
        if constexpr (IO::syndrome)
          syn |= m & mask;
        else
          (void)m, (void)mask;
      }
      LDPC_TLOG_END(0, deg);
    }
#endif
    const int bn_ticket = task - n_cn_tasks;
    {
      /* flags[p & 1] = how many lanes saw an unsatisfied check of the previous pass (0 = none: the stop criterion; the
       * count itself only steers the eager check below) */
      if constexpr (IO::syndrome) {
        const unsigned long long bad_lanes = __ballot(syn != 0);
        if (bad_lanes && lane == 0)
          atomicAdd(&flags[p & 1], (int)__popcll(bad_lanes));
      }
    }
    if (tid == 0) {
      flags[2] = 0;
      flags[8 + ((p + 1) & 1)] = 0; /* the next pass' queue: last drawn from before the previous pass' last barrier, first
                                       drawn from behind the barrier below */
      /* decoder.c:556-559: once a segment of the transport block has failed, its siblings give up at their next pass */
      if (tb_ab)
        flags[3] = 1;
      if (p >= 2 && io.abort_is(ab_word))
        flags[3] = 1;
    }
    __syncthreads();
    if constexpr (IO::pass_stamps) {
      if (tid == 0 && p <= 9) {
        io.stamps()[2 * p] = (uint32_t)wall_clock64();
        io.stamps()[22 + p] = (uint32_t)flags[p & 1];
      }
    }
    if (io.has_abort() && flags[3]) { /* (set by one of the two abort sources) */
      n_iter = max_pass + 1;
      break;
    }
    /* (the core columns' device copy, written in the prologue by this workgroup's own waves, is read from here on: the
     * barriers since then order it at workgroup scope, which is all a CU's write-through vector cache needs; an agent-
     * scope fence here would write back and invalidate the XCD's L2) */
    const int bad_prev = flags[p & 1]; /* unsatisfied lanes after pass p - 1 (after the channel's hard decisions for p = 1) */
    if (!io.use_crc() && p >= 3 && flags[p & 1] == 0) {
      n_iter = p - 1;
      break;
    }
#ifndef LDPC_ABLATE_BN
    if constexpr (IO::bn_tickets) {
    /* the next ticket is asked for behind an item's gather and looked at behind its store: the draw's round trip runs
     * under the tail of the task (profiles/r04/decoder_ab25_late_draw.txt) */
    for (int ticket = bn_ticket, nxt = 0; ticket < n_bn_tickets; ticket = LDPC_UNIFORM(nxt) - n_cn_tasks) {
      LDPC_TLOG_BEGIN();
      /* (ticket and task record in one scalar load) */
      const int item0 = code->f_bn_rec[ticket][0] + lane, end = code->f_bn_rec[ticket][1];
      const int maxdeg = code->f_bn_rec[ticket][2], cnt = code->f_bn_rec[ticket][3];
      if (cnt == 1) {
        if (item0 < end) {
          const int sc = (int)ldpc_umulhi((uint32_t)item0, zq_magic), j = item0 - sc * zq;
          const uint32_t colrec = coltbl[sc];
          const uint32_t lw = src32[(int)(colrec & 0xffu) * zq + j];
          uint32_t acc_e = 0, acc_o = 0;
          ldpc_fast_bn_gather(L, colrec, maxdeg, j, Z, 0, acc_e, acc_o);
          nxt = ldpc_draw_issue(pq, lane);
          ldpc_fast_bn_finish(L, (int)(colrec & 0xffu), (int)((colrec >> 8) & 0xffu), j, Z, astride, lw, acc_e, acc_o, 0);
        } else {
          nxt = ldpc_draw_issue(pq, lane);
        }
      } else {
        /* short columns: one item of each of the ticket's tasks per thread, walked together (the tasks' item ranges follow
         * each other, 64 apart) */
        uint32_t rec[LDPC_F_BN_GROUP], lw[LDPC_F_BN_GROUP];
        int jj[LDPC_F_BN_GROUP];
        bool live[LDPC_F_BN_GROUP];
#pragma unroll
        for (int g = 0; g < LDPC_F_BN_GROUP; g++) {
          const int item = item0 + 64 * g;
          live[g] = g < cnt && item < end;
          const int it = live[g] ? item : end - 1; /* filler: the last item, a short column's */
          const int sc = (int)ldpc_umulhi((uint32_t)it, zq_magic);
          jj[g] = it - sc * zq;
          rec[g] = coltbl[sc];
          lw[g] = src32[(int)(rec[g] & 0xffu) * zq + jj[g]];
        }
        ldpc_fast_bn_multi<LDPC_F_BN_GROUP>(L, rec, jj, lw, live, maxdeg, Z, astride);
        nxt = ldpc_draw_issue(pq, lane);
      }
      LDPC_TLOG_END(1, maxdeg);
    }
    } else {
    for (int ticket = bn_ticket; ticket * bn_group < n_bn_tasks; ticket = ldpc_draw(pq, lane) - n_cn_tasks) {
      for (int task = ticket * bn_group; task < (ticket + 1) * bn_group && task < n_bn_tasks; task++) {
        LDPC_TLOG_BEGIN();
        const int item = code->f_bn_task[task][0] + lane, end = code->f_bn_task[task][1];
        const int maxdeg = code->f_bn_task[task][2];
        if (item < end) {
          const int sc = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sc * zq;
          const uint32_t colrec = coltbl[sc];
          const uint32_t lw = src32[(int)(colrec & 0xffu) * zq + j];
          ldpc_fast_bn(L, colrec, maxdeg, j, Z, astride, lw);
        }
        LDPC_TLOG_END(1, maxdeg);
      }
    }
    }
#endif
    cn_ticket = ldpc_draw(&flags[8 + ((p + 1) & 1)], lane);
    if (tid == 0)
      flags[(p + 1) & 1] = 0;
    __syncthreads();
    if constexpr (IO::pass_stamps) {
      if (tid == 0 && p <= 9)
        io.stamps()[2 * p + 1] = (uint32_t)wall_clock64();
    }
    if (io.eager_check() && !io.use_crc() && p >= 2 && p < max_pass && bad_prev <= LDPC_EAGER_MAX_BAD_LANES) {
      /* the check the next pass would make first thing (decoder.c:842-848: cnProcPc on this pass' results; p + 1 >= 3 and
       * p + 1 <= max_pass as there), as a sweep of its own over the same task list -- when the previous pass was already
       * close (few unsatisfied lanes): a block that is far from converging does not pay for sweeps that cannot succeed,
       * and a missed chance only means the stop is noticed one check-node phase later, with the same pass count */
      /* items are walked by index, not by task: row record (degree, first edge) per lane from the LDS table, no queue and
       * no task records from the descriptor -- the sweep is a few loads per edge, and the fetch chain in front of every
       * task was most of its time (6 us -> 2 us for BG1 Zc = 384) */
      if constexpr (IO::pass_stamps) {
        if (tid == 0)
          io.stamps()[20] |= 1u << p;
      }
      uint32_t esyn = 0;
      const int n_items = code->nrows * zq;
      for (int item = tid; item < n_items; item += nt) {
        const int sr = (int)ldpc_umulhi((uint32_t)item, zq_magic), j = item - sr * zq;
        const uint32_t rowrec = rowtbl[sr];
        const int e0 = (int)(rowrec & 0x1ffu), deg = (int)((rowrec >> 9) & 0x1fu), ext = (int)((rowrec >> 14) & 1u);
        const int valid = (int)(rowrec >> 16) - 4 * j;
        const uint32_t m = ldpc_fast_pc(L, deg, ext, e0, j, rstride);
        esyn |= m & ldpc_fast_valid_lanes(valid);
      }
      if (__any(esyn != 0) && lane == 0)
        flags[6] = 1;
      __syncthreads();
      const int bad = flags[6];
      __syncthreads();
      if (tid == 0)
        flags[6] = 0;
      if (!bad) {
        n_iter = p;
        break;
      }
      __syncthreads();
    }
    if (io.use_crc() && p >= 3) { /* see ldpc_dec_generic_block.h for the CRC argument */
      uint32_t x = 0;
      const int crcE = io.crcE();
      const uint32_t *crc_pow = io.crc_pow();
      /* four hard decisions (one APP dword: Zc % 4 == 0 keeps them in one column) and their four table entries per
       * step, the loads unconditional and masked afterwards: independent loads in flight instead of a chain of
       * bit test -> load -> wait (E is a multiple of 8) */
      /* (not unrolled: a handful of iterations per thread, and unrolled it was the kernel's register peak -- 126 instead of
       * 92 VGPRs, scratch memory in the instantiations that carry both stop modes) */
#pragma unroll 1
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
  if (io.stamps() && tid == 0)
    io.stamps()[1] = (uint32_t)wall_clock64();
  if (io.tb_abort() && n_iter == max_pass && tid == 0) /* decoder.c:190-193: a failed segment gives the whole TB up */
    __hip_atomic_store(io.tb_abort(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  /* ---- hard decision ------------------------------------------------------------------------------------- */
#ifdef LDPC_TIMING
  if (blockIdx.x == 0)
    return n_iter; /* (its output row holds the log) */
#endif
  /* output dword w: packed bits 32w .. 32w+31 MSB first (bnProc.h:1353-1380), resp. bits 4w .. 4w+3 one per byte */
  auto bits_word = [&](int w) -> uint32_t {
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
    return word;
  };
  auto bytes_word = [&](int w) -> uint32_t {
    const int b = 4 * w;
    if (b >= ncz)
      return 0u;
    const int c = (int)ldpc_umulhi((uint32_t)b, z_magic), u = b - c * Z;
    const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u);
    return (nb >> 7) & 0x01010101u;
  };
  if constexpr (IO::tb_epilogue) {
    if (io.tb_fused()) {
      io.tb_finish(n_iter, bits_word, flags);
      return n_iter;
    }
  }
  if ((!io.use_crc() || n_iter >= 3) && n_iter <= max_pass) {
    const uint32_t tag = io.out_tag();
    const int nwords = io.out_mode() == 0 ? (num_llr + 31) >> 5 : num_llr >> 2;
    if (tag) {
      uint4 *o16 = reinterpret_cast<uint4 *>(io.out());
      for (int c = tid; 3 * c < nwords; c += nt) {
        uint32_t w3[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const int w = 3 * c + q;
          w3[q] = w < nwords ? (io.out_mode() == 0 ? bits_word(w) : bytes_word(w)) : 0u;
        }
        io.put16(o16 + c, w3[0], w3[1], w3[2], tag);
      }
    } else if (io.out_mode() == 0) {
      uint32_t *o = reinterpret_cast<uint32_t *>(io.out());
      for (int w = tid; w < nwords; w += nt)
        o[w] = bits_word(w);
    } else {
      int8_t *o = io.out();
      for (int i = tid; i < num_llr; i += nt)
        o[i] = (i < ncz) ? (int8_t)ldpc_fast_hd(L, i, Z, z_magic, astride) : (int8_t)0;
    }
  }
  return n_iter;
}
#endif
