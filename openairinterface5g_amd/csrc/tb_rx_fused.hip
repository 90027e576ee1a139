/*
 * tb_rx_fused.hip -- the UL-SCH chain's fused segment kernel for gfx950: one workgroup takes ONE code segment from the
 * received LLRs to its payload bytes, the way the reference's per-segment job does on one CPU thread
 * (openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:122-223 nr_processULSegment: nr_deinterleaving_ldpc ->
 * nr_rate_matching_ldpc_rx -> int8 pack -> LDPCdecoder; SCHED_NR/phy_procedures_nr_gNB.c:271-300 nr_postDecode: copy of
 * the segment's bytes into the transport block, and when the last segment has arrived the TB CRC and the verdict).
 *
 *   prologue  tb_rx_core.h: de-interleave + rate de-match + HARQ combine through an LDS image (it lies in the decoder's
 *             message area, which the first pass does not read), soft buffer updated in HBM, int8 decoder input written
 *             to the segment's scratch row (the decoder re-reads the core columns from it every pass, L2 resident).  The
 *             LLRs may sit in page-locked HOST memory: the workgroup then pulls them over the link itself, and with the
 *             launch's workgroups at different points of their lives the link, HBM and the CUs are busy together -- no
 *             copy engine, no copy -> kernel edge.
 *   body      ldpc_dec_fast_block (CRC stop, transport-block abort flag).
 *   epilogue  instead of an output row: the segment's bytes of the payload (stores that leave the caches), its share of the
 *             TB CRC register (the CRC is linear in the bits) and its pass count in one 8-byte slot tagged with the block's
 *             generation; a per-TB counter tells the last segment to finish that it is the last, and that workgroup
 *             collects the slots (waiting until each shows the generation), delivers ACK / iter_max and zeroes the
 *             payload of a block that failed, as the reassembly kernel does.  No fences: everything another workgroup
 *             looks at is moved by device-scope atomics or write-through stores; an agent-scope release fence would
 *             write back the XCD's whole L2 per segment.
 *
 * Replaces four launches (tb_rx_dematch_kernel, ldpc_dec_fast_kernel<true, true>, tb_rx_assemble_kernel,
 * tb_rx_verdict_kernel) for the segments it serves; those kernels remain for segments of codes the fast decoder does not
 * take (Zc % 4 != 0) and for small segments that share workgroups, and behind NRLDPC_HIP_TB_FUSED=0.
 */
#include <hip/hip_runtime.h>
#include "ldpc_kernels.h"
#include "tb_chain.h"
#include "tb_rx_core.h"
#include "ldpc_dec_fast_block.h"

typedef const tb_rx_seg_job LDPC_CONST_AS *tb_seg_ptr_t;
typedef const tb_rx_tb_job LDPC_CONST_AS *tb_tb_ptr_t;

/* stores / loads that other workgroups (other XCDs: other L2s) or the host may look at while the kernel runs */
template <class T> __device__ __forceinline__ void tb_store_out(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void tb_wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* LROW: the decoder input lives in the workgroup's LDS (tb_rx_fused_args.lrow_off) -- a compile-time property, so that the
 * block body's reads of the channel LLRs are LDS instructions and not flat ones */
template <bool LROW, bool MUTE = false> struct tb_rx_fused_io {
  const ldpc_dec_args &a;
  const tb_rx_fused_args &x;
  ldpc_job_ptr_t job;
  uint8_t *lds;
  bool tables_early = false; /* the kernel has put the code's tables into LDS already (behind the de-matching stores) */
  __device__ __forceinline__ tb_seg_ptr_t seg() const { return (tb_seg_ptr_t)x.segs + job->seg_idx; }
  __device__ __forceinline__ tb_tb_ptr_t tb() const { return (tb_tb_ptr_t)x.tbs + seg()->tb; }
  __device__ __forceinline__ const uint32_t *src32() const
  {
    if constexpr (LROW)
      return reinterpret_cast<const uint32_t *>(lds + x.lrow_off);
    else
      return reinterpret_cast<const uint32_t *>(a.llr + (size_t)job->llr_off);
  }
  __device__ __forceinline__ int8_t *out() const { return a.out + (size_t)job->out_off; }
  __device__ __forceinline__ int max_pass() const { return job->num_max_iter + 1; }
  __device__ __forceinline__ int use_crc() const { return 1; }
  static constexpr bool syndrome = false; /* CRC stop: nobody looks at the parity of the hard decisions */
  __device__ __forceinline__ int crcE() const { return job->E; }
  __device__ __forceinline__ const uint32_t *crc_pow() const { return a.crc_pow_tbl[job->crc_type & 3]; }
  __device__ __forceinline__ int out_mode() const { return 0; }
  __device__ __forceinline__ int *tb_abort() const { return (a.tb_abort && job->abort_idx >= 0) ? a.tb_abort + job->abort_idx : nullptr; }
  __device__ __forceinline__ uint32_t *stamps() const { return nullptr; }
  static constexpr bool pass_stamps = false;
  __device__ __forceinline__ int fair_turns() const { return a.fair; }
  __device__ __forceinline__ int tid() const { return (int)threadIdx.x; }
  __device__ __forceinline__ bool eager_check() const { return false; }
  static constexpr bool mute_items = MUTE; /* launches that hold retransmissions: an instantiation of their own, so that the look
                                              for mute items costs first transmissions nothing (its mere presence was +1 %) */
  __device__ __forceinline__ bool mute_check() const { return job->crc_type & LDPC_JOB_MUTE_CHECK; }
  static constexpr bool bn_tickets = true;
  __device__ __forceinline__ bool tables_resident() const { return tables_early; }
  __device__ __forceinline__ uint32_t out_tag() const { return 0u; }
  __device__ __forceinline__ uint32_t abort_load() const { return 0u; }
  __device__ __forceinline__ bool abort_is(uint32_t) const { return false; }
  __device__ __forceinline__ bool has_abort() const { return true; }
  __device__ __forceinline__ void put16(uint4 *p, uint32_t x0, uint32_t y, uint32_t z, uint32_t t) const { *p = make_uint4(x0, y, z, t); }
  __device__ __forceinline__ uint32_t ld_llr(const uint32_t *p) const { return *p; }
  __device__ __forceinline__ const uint32_t *src32_prologue() const { return src32(); }
  __device__ __forceinline__ uint32_t *stage_core() const { return nullptr; }
  static constexpr bool tb_epilogue = true;
  __device__ __forceinline__ bool tb_fused() const { return job->seg_idx >= 0; }

  /* The segment's part of nr_postDecode (phy_procedures_nr_gNB.c:271-300).  bits_word(w) = the block's hard decisions 32w ..
   * 32w + 31, MSB first inside each byte, as the dword the output row would hold (bnProc.h:1353-1380): byte k of it is byte
   * 4w + k of the segment.  flags = the block body's LDS words ([2], [4], [5] are free here).  Called by every thread. */
  template <class Bits> __device__ __forceinline__ void tb_finish(int n_iter, Bits bits_word, int *flags) const
  {
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u;
    if (x.prio_pro & 2u)
      __builtin_amdgcn_s_setprio(3);
    if (x.trace && tid == 0)
      x.trace[(size_t)blockIdx.x * 16 + 4] = wall_clock64();
    const tb_seg_ptr_t sj = seg();
    const tb_tb_ptr_t tj = tb();
    const uint32_t tbi = sj->tb, C = tj->C, seg_bytes = tj->seg_bytes, bbytes = tj->B >> 3, abytes = tj->A >> 3;
    const uint32_t first = sj->r * seg_bytes;
    uint32_t count = 0; /* bytes of b = payload || TB CRC this segment carries */
    if (first < bbytes)
      count = first + seg_bytes <= bbytes ? seg_bytes : bbytes - first;
    const bool ok = n_iter <= (int)tj->num_max_iter;
    /* the block's generation: written by the last segment of the PREVIOUS call that held this block (a kernel boundary
     * ago: a scalar load sees it), constant while this launch's segments of the block are running */
    const uint32_t gen_prev = ((const uint32_t LDPC_CONST_AS *)x.gen)[tbi];
    uint32_t xr = 0;
    if (ok) {
      uint8_t *dst = x.payload + tj->payload_off + first;
      const uint32_t pay_n = first < abytes ? (count < abytes - first ? count : abytes - first) : 0u; /* the rest is the TB CRC */
      const bool al = (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
      const uint32_t Bt = tj->B - 8 * first; /* bit i of the segment is followed by Bt - 1 - i bits of b */
      const uint32_t *pow = x.pow24a;
      for (uint32_t w = tid; 4 * w < count; w += nt) {
        uint32_t word = bits_word((int)w);
        const uint32_t nb = count - 4 * w;
        if (nb < 4)
          word &= (1u << (8 * nb)) - 1u;
        if (4 * w < pay_n) {
          const uint32_t pb = pay_n - 4 * w;
          if (pb >= 4 && al) {
            tb_store_out(reinterpret_cast<uint32_t *>(dst + 4 * w), word);
          } else {
            for (uint32_t k = 0; k < 4 && k < pb; k++)
              tb_store_out(dst + 4 * w + k, (uint8_t)(word >> (8 * k)));
          }
        }
        if (C > 1) { /* partial TB CRC: bit i contributes x^(Bt - 1 - i) * x^24 mod g (crc_byte.c:148-182 is linear in the bits) */
#pragma unroll
          for (int q = 0; q < 8; q++) {
            const uint32_t i = 32 * w + 4 * (uint32_t)q;
            if (i < 8 * count) { /* four bits and their four table entries (one aligned 16-byte load: Bt, i multiples of 4) */
              const uint4 pw = *reinterpret_cast<const uint4 *>(pow + (Bt - 4 - i));
              const uint32_t nib = (word >> (8 * (q >> 1) + ((q & 1) ? 0 : 4))) & 0xfu; /* bit 3 = bit i of the segment */
              xr ^= (pw.w & (0u - ((nib >> 3) & 1u))) ^ (pw.z & (0u - ((nib >> 2) & 1u))) ^ (pw.y & (0u - ((nib >> 1) & 1u))) ^
                    (pw.x & (0u - (nib & 1u)));
            }
          }
        }
      }
    }
    for (int off = 32; off; off >>= 1)
      xr ^= __shfl_xor(xr, off);
    /* (flags[2], the block body's CRC register, is zero whenever the segment decoded -- which is when xr can be non-zero;
     * flags[4], flags[5] are zero since the block's prologue) */
    if (lane == 0 && xr)
      atomicXor(reinterpret_cast<unsigned int *>(&flags[2]), xr);
    /* the count that decides who collects the block is asked for NOW: its round trip runs under the drain of the payload
     * stores instead of behind it (nothing depends on the order: the collector waits for every slot's generation anyway, and
     * a slot is only written below, behind the drain) */
    int before = 0;
    if (tid == 0)
      before = __hip_atomic_fetch_add(&x.done[tbi], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tb_wait_stores(); /* this thread's payload bytes have left for memory */
    __syncthreads();
    /* What the block's last segment needs of this one travels in ONE 8-byte store -- {CRC share, pass count, the block's
     * generation} -- and nothing orders it against the count below: the last segment reads the slots until every one carries
     * the current generation.  (A slot that shows the generation also says that the segment's payload bytes are in memory:
     * they were waited for above.)  So a segment's epilogue costs one memory round trip behind its stores, not three. */
    const uint32_t gen = gen_prev + 1u;
    if (tid == 0) {
      const unsigned long long slot = (unsigned long long)(ok ? (uint32_t)flags[2] : 0u) |
                                      ((unsigned long long)(((uint32_t)n_iter & 0xffffu) | (gen << 16)) << 32);
      __hip_atomic_store(&x.slots[job->seg_idx], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (int)C - 1)
        flags[4] = 1;
    }
    __syncthreads();
    if (!flags[4])
      return;
    /* the last segment of the transport block to finish */
    if (tid < 64) {
      const int nmi = (int)tj->num_max_iter;
      int imax = 0, bad = 0;
      uint32_t crc = 0;
      for (uint32_t r = lane; r < C; r += 64) {
        unsigned long long v;
        while ((uint32_t)((v = __hip_atomic_load(&x.slots[tj->seg0 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 48) != (gen & 0xffffu))
          __builtin_amdgcn_s_sleep(2);
        const int it = (int)((v >> 32) & 0xffffu);
        imax = it > imax ? it : imax;
        bad |= it > nmi;
        crc ^= (uint32_t)v;
      }
      for (int off = 32; off; off >>= 1) {
        const int o = __shfl_xor(imax, off);
        imax = o > imax ? o : imax;
        bad |= __shfl_xor(bad, off);
        crc ^= __shfl_xor(crc, off);
      }
      if (lane == 0) {
        __hip_atomic_store(&x.done[tbi], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&x.gen[tbi], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.tb_abort)
          __hip_atomic_store(&a.tb_abort[tbi], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        /* single-segment TBs were CRC-checked inside the decoder (phy_procedures_nr_gNB.c:293-299) */
        const int ackv = !bad && (C == 1 || crc == 0);
        tb_store_out(x.ack + tbi, (uint8_t)ackv);
        /* a segment that gave up because a sibling had failed reports numMaxIter + 2 (decoder.c:556-559); which siblings
         * get that far is a matter of timing, so the per-TB figure is capped at "failed" = numMaxIter + 1 */
        tb_store_out(x.iter_max + tbi, imax > nmi + 1 ? nmi + 1 : imax);
        flags[5] = ackv;
      }
    }
    __syncthreads();
    if (!flags[5]) { /* a block that failed delivers zeros, not the bytes of the segments that happened to decode */
      uint8_t *dst = x.payload + tj->payload_off;
      const uint32_t head = (uint32_t)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);
      for (uint32_t k = tid; k < head && k < abytes; k += nt)
        tb_store_out(dst + k, (uint8_t)0);
      if (abytes > head) {
        const uint32_t nw = (abytes - head) >> 2;
        for (uint32_t w = tid; w < nw; w += nt)
          tb_store_out(reinterpret_cast<uint32_t *>(dst + head) + w, 0u);
        for (uint32_t k = head + 4 * nw + tid; k < abytes; k += nt)
          tb_store_out(dst + k, (uint8_t)0);
      }
    }
  }
};

template <bool LROW, bool MUTE, int ZC = 0>
__global__ void __launch_bounds__(1024) tb_rx_fused_kernel(const ldpc_dec_args a, const tb_rx_fused_args x)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  const ldpc_job_ptr_t job = (ldpc_job_ptr_t)a.jobs + blockIdx.x;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)job->code;
  tb_rx_fused_io<LROW, MUTE> io{a, x, job, fsm};
  if (x.stagger_ticks && blockIdx.x >= x.stagger_cus && blockIdx.x < x.stagger_cus * x.stagger_slots) {
    const long long until = (long long)wall_clock64() + (long long)(blockIdx.x / x.stagger_cus) * (long long)x.stagger_ticks;
    while ((long long)wall_clock64() < until)
      __builtin_amdgcn_s_sleep(32);
  }
  unsigned long long *tr = x.trace ? x.trace + (size_t)blockIdx.x * 16 : nullptr;
  if (tr && threadIdx.x == 0) {
    tr[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  /* HW_REG_HW_ID: wave, simd, cu, sh, se ... */
    tr[1] = __builtin_amdgcn_s_getreg((3 << 11) | 20);  /* HW_REG_XCC_ID */
    tr[2] = wall_clock64();
  }
  if (LROW || io.tb_fused()) {
    /* the prologue is a chain of short steps (clear, scatter, stream) with little arithmetic: at the CU's other workgroup's
     * mercy -- an older wave that decodes wins every issue slot -- it took 8 us; with the issue priority raised for its duration
     * it costs the decoder next door a few hundred instructions (x.prio_pro; profiles/r06/ab_prologue_priority.txt) */
    if (x.prio_pro)
      __builtin_amdgcn_s_setprio(3);
    const tb_seg_ptr_t sj = io.seg();
    const tb_rx_geom g = tb_rx_geometry(sj);
    int8_t *l;
    if constexpr (LROW)
      l = reinterpret_cast<int8_t *>(fsm + x.lrow_off);
    else
      l = const_cast<int8_t *>(a.llr) + sj->l_off;
    tb_rx_dematch_block(g, sj->Qm, x.llr + sj->llr_off, x.harq + sj->harq_off, l, reinterpret_cast<int16_t *>(fsm), tr ? tr + 7 : nullptr);
    /* the code's tables go into LDS NOW, their loads in flight beside the de-matching stores -- not behind the barrier, in the
     * decoder's prologue, where nothing hides them.  (Only when the de-matching image, which other waves may still be
     * reading, ends in front of the tables' place: every large code.) */
    if (2u * g.span <= (uint32_t)code->f_lds_etbl) {
      ldpc_fast_tables_to_lds(fsm, code, (int)threadIdx.x, (int)blockDim.x);
      io.tables_early = true;
    }
    if constexpr (LROW) {
      /* the decoder input is in LDS: a barrier behind the LDS writes makes it visible, and nobody waits for the soft-buffer
       * stores -- they drain under the decoding.  (__syncthreads() is a workgroup-scope release over every address space: it
       * waits for the wave's stores to memory too, as the release of the memory-resident row below must.) */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else {
      /* the decoder input is read back by other waves of this workgroup only: workgroup scope (see ldpc_dec_fast_pull_kernel) */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (tr && threadIdx.x == 0)
      tr[10] = wall_clock64();
    if (x.prio_pro)
      __builtin_amdgcn_s_setprio(0);
  }
  if (tr && threadIdx.x == 0)
    tr[3] = wall_clock64();
  const int n_iter = ldpc_dec_fast_block<tb_rx_fused_io<LROW, MUTE>, ZC>(fsm, code, io);
  if (tr && threadIdx.x == 0) {
    tr[5] = wall_clock64();
    tr[6] = (unsigned long long)n_iter;
  }
  if (!io.tb_fused() && threadIdx.x == 0)
    a.n_iter[(uint32_t)job->iter_idx] = n_iter;
}

hipError_t tb_rx_fused_init(void)
{
  const void *k[] = {reinterpret_cast<const void *>(tb_rx_fused_kernel<false, false>), reinterpret_cast<const void *>(tb_rx_fused_kernel<true, false>),
#define X(z) \
                     reinterpret_cast<const void *>(tb_rx_fused_kernel<false, false, z>), reinterpret_cast<const void *>(tb_rx_fused_kernel<true, false, z>), \
                     reinterpret_cast<const void *>(tb_rx_fused_kernel<false, true, z>), reinterpret_cast<const void *>(tb_rx_fused_kernel<true, true, z>),
                     LDPC_FAST_ZC_LIST(X)
#undef X
                     reinterpret_cast<const void *>(tb_rx_fused_kernel<false, true>), reinterpret_cast<const void *>(tb_rx_fused_kernel<true, true>)};
  for (const void *f : k) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess)
      return e;
  }
  return hipSuccess;
}

hipError_t tb_launch_rx_fused(const ldpc_dec_args &a, const tb_rx_fused_args &x, int n_threads, int lds_bytes, uint32_t n_jobs, hipStream_t s)
{
  if (n_jobs == 0)
    return hipSuccess;
  if (!a.jobs || !x.segs || !x.tbs)
    return hipErrorInvalidValue;
  if (ldpc_fast_zc_enabled((int)x.zc)) { /* every job has this lifting size: the instantiations with compile-time row strides */
    switch ((int)x.zc) {
#define X(z) \
  case z: \
    if (x.lrow_off && x.mute) \
      hipLaunchKernelGGL((tb_rx_fused_kernel<true, true, z>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x); \
    else if (x.lrow_off) \
      hipLaunchKernelGGL((tb_rx_fused_kernel<true, false, z>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x); \
    else if (x.mute) \
      hipLaunchKernelGGL((tb_rx_fused_kernel<false, true, z>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x); \
    else \
      hipLaunchKernelGGL((tb_rx_fused_kernel<false, false, z>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x); \
    return hipGetLastError();
      LDPC_FAST_ZC_LIST(X)
#undef X
      default:
        break;
    }
  }
  if (x.lrow_off && x.mute)
    hipLaunchKernelGGL((tb_rx_fused_kernel<true, true>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x);
  else if (x.lrow_off)
    hipLaunchKernelGGL((tb_rx_fused_kernel<true, false>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x);
  else if (x.mute)
    hipLaunchKernelGGL((tb_rx_fused_kernel<false, true>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x);
  else
    hipLaunchKernelGGL((tb_rx_fused_kernel<false, false>), dim3(n_jobs), dim3(n_threads), lds_bytes, s, a, x);
  return hipGetLastError();
}
