/*
 * ldpc_server.hip -- resident server kernel behind the per-segment plugin entry points (protocol: ldpc_server.h).
 *
 * One workgroup of 1024 threads per caller slot, owning a whole CU (160 KiB of LDS).  Wave 0 polls the slot's request
 * line in page-locked host memory (64 bytes: the whole request header arrives with the poll that notices it); the
 * payload is read straight from the slot's host memory by the decoder's own prologue (its latency hides behind the
 * table and message initialisation; the fast decoder drops a device copy of the core columns, which it re-reads every
 * pass), the block runs through the same per-block device functions as the batch kernels (ldpc_dec_fast_block.h /
 * ldpc_dec_generic_block.h / ldpc_enc_packed_core.h) and the results are written straight into the slot's host
 * output area, followed by the completion word.
 *
 * Replaces, for the reference's call pattern (one LDPCdecoder call per code segment from each thread-pool worker,
 * nr_ulsch_decoding.c:435-468; one LDPCencoder call per 8 segments, nr_dlsch_coding.c:386-403), the HIP runtime round
 * trip per call.
 */
#include <hip/hip_runtime.h>
#include "ldpc_server.h"
#include "ldpc_dec_fast_block.h"
#include "ldpc_dec_generic_block.h"
#include "ldpc_enc_packed_core.h"

/* 16 waves (128 VGPRs per lane), like the batch kernel's workgroup for the large codes: a pass is ~9 % faster than with 12
 * waves and 168 VGPRs (measured, profiles/r02/README.md).  Around the inlined decoders the server loop does not fit in
 * 128 VGPRs: ~20 registers spill, all of them per-thread addresses of the request fetch that are stored before the loop
 * and reloaded once per request -- there is no scratch access inside the decoder's phases (checked in the
 * disassembly).  With the encoder's phases compiled into the same kernel the spills reach the hot loops, so LDPCencoder
 * calls are served by a kernel of their own (template parameter ENC: 86 VGPRs, its own slots and stream). */
#ifndef SRV_THREADS
#define SRV_THREADS 1024
#endif
#define SRV_ENC_GROUP 128 /* threads per segment of an encoder call: 8 segments side by side in one workgroup */

__device__ __forceinline__ uint32_t srv_ld_sys(const uint32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void srv_st_sys(uint32_t *p, uint32_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* one 16-byte store that is written through to system memory by itself (sc0 sc1: not left in L2 until somebody's release
 * -- plain stores to page-locked host memory stay there until the kernel ends: measured, every call took the idle
 * timeout) */
typedef uint32_t srv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void srv_st16_sys(void *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
  const srv_u32x4 v = {a, b, c, d};
  /* (the s_nop: a VALU write of the store's data registers needs a wait state after a store of more than 64 bits; the
   * compiler's hazard recogniser does not look inside inline assembly -- without it the units went out with corrupted
   * third / fourth dwords) */
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

/* The server loop keeps next to nothing live across a decode: the kernel arguments are re-read from the kernarg segment
 * (scalar loads) where they are needed, the loop state sits in LDS.  Inlined next to a loop with its own live values
 * the fast decoder -- 127 of the 128 VGPRs a 1024-thread workgroup may use -- would spill; out of line it would need a
 * stack.  Either way scratch memory, which a resident kernel had better not depend on. */
typedef const srv_args LDPC_CONST_AS *srv_args_ptr_t;

/* a wave-uniform pointer the compiler has put (or might put) in VGPRs, back in SGPRs: what is derived from it then is
 * scalar too and does not compete with the decoder for vector registers */
template <typename T> __device__ __forceinline__ T *srv_sgpr(T *p)
{
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<T *>(((uint64_t)(uint32_t)LDPC_UNIFORM((uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)LDPC_UNIFORM((uint32_t)v));
}

/* CRC = the request stops on the CRC (what the reference's per-segment callers ask for, nr_ulsch_decoding.c:219): the block
 * body is instantiated without the parity of the hard decisions (ldpc_dec_fast_block.h `syndrome`) */
template <bool CRC> struct srv_fast_io { /* the request header sits in LDS: read where needed (ldpc_dec_fast_block.h) */
  const srv_req *rq;
  const uint8_t *host_llr;  /* the slot's input area in host memory: read once, by the prologue */
  const uint8_t *staged;    /* device copy of the core columns, written by the prologue, re-read every pass */
  uint8_t *hout;
  srv_args_ptr_t a;
  uint32_t *st;
  int tid_;
  bool resident_;
  uint32_t tag_;
  __device__ __forceinline__ const uint32_t *src32() const { return reinterpret_cast<const uint32_t *>(staged); }
  __device__ __forceinline__ const uint32_t *src32_prologue() const { return reinterpret_cast<const uint32_t *>(host_llr); }
  __device__ __forceinline__ uint32_t *stage_core() const { return reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(staged)); }
  __device__ __forceinline__ int8_t *out() const { return reinterpret_cast<int8_t *>(hout); }
  __device__ __forceinline__ int max_pass() const { return LDPC_UNIFORM((int)rq->max_pass); }
  __device__ __forceinline__ int use_crc() const { return CRC ? 1 : 0; }
  __device__ __forceinline__ int crcE() const { return LDPC_UNIFORM((int)rq->crcE); }
  __device__ __forceinline__ const uint32_t *crc_pow() const { return a->crc_pow_tbl[LDPC_UNIFORM(rq->kind_mode >> 24) & 3u]; }
  __device__ __forceinline__ int out_mode() const { return LDPC_UNIFORM((int)((rq->kind_mode >> 8) & 0xffu)); }
  __device__ __forceinline__ int *tb_abort() const { return nullptr; }
  const uint32_t *abw_;
  /* (typed as global memory: through a generic pointer this would be a FLAT load, which the LDS waits of the check-node
   * phase that follows would wait for as well -- 0.9 us per pass on the critical path) */
  __device__ __forceinline__ uint32_t abort_load() const
  {
    return __hip_atomic_load((const __attribute__((address_space(1))) uint32_t *)abw_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __device__ __forceinline__ bool abort_is(uint32_t w) const { return w == tag_; }
  __device__ __forceinline__ bool has_abort() const { return true; }
  __device__ __forceinline__ uint32_t *stamps() const { return st; }
  static constexpr bool pass_stamps = false;
  __device__ __forceinline__ int fair_turns() const { return 0; }
  __device__ __forceinline__ int tid() const { return tid_; }
  __device__ __forceinline__ bool eager_check() const { return true; }
  /* one block per CU with as many waves as tasks: grouping short bit-node tasks buys nothing here, and the plain loop is
   * 3 % faster on the large codes (profiles/r03/README.md) */
  static constexpr bool mute_items = false; /* (the entry point cuts whole zero columns on the host: ldpc_api.cpp cut_code_for) */
  __device__ __forceinline__ bool mute_check() const { return false; }
  static constexpr bool bn_tickets = false;
  static constexpr bool tb_epilogue = false;
  static constexpr bool syndrome = !CRC;
  __device__ __forceinline__ bool tables_resident() const { return resident_; }
  __device__ __forceinline__ uint32_t out_tag() const { return tag_; }
  __device__ __forceinline__ void put16(uint4 *p, uint32_t x, uint32_t y, uint32_t z, uint32_t t) const { srv_st16_sys(p, x, y, z, t); }
  /* the payload was written by the host (over the BAR, or into page-locked memory) while this kernel runs: system scope */
  __device__ __forceinline__ uint32_t ld_llr(const uint32_t *p) const { return srv_ld_sys(p); }
};

/* ENC: the encoder server -- its own launch with its own slots, serving LDPCencoder calls only: the decoders' kernels do
 * not carry the encoder's phases (with them inlined next to the decoder the spills reached the decoder's loops) */
template <bool ENC = false>
__global__ void __launch_bounds__(SRV_THREADS) ldpc_server_kernel(const srv_args args_by_value)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  /* [0] doorbell / quit, [1] last served, [2], [3] time stamps, [4..19] / [32..47] request header (two copies used in
   * turn: wave 0 may already be fetching the next request while the other waves still read this one's fields on their way
   * out of the decoder -- nothing but the PCIe round trip kept them apart before), [20], [21] code whose tables
   * the LDS holds, [24], [25] the fast decoder's stamps */
  uint32_t *bc = reinterpret_cast<uint32_t *>(fsm + SRV_BC_OFF);
  uint32_t hdr = 4; /* where this iteration's header goes: 4 or 32 (wave-uniform, lives in an SGPR) */
  /* laundered wherever it is used: the loads behind it must not be hoisted out of the loop and kept live */
#define SRV_ARGS() ({ uint64_t p_ = args_u64; asm volatile("" : "+s"(p_)); (srv_args_ptr_t)p_; })
  (void)args_by_value; /* = the kernarg segment, read through the laundered pointer */
  const uint64_t args_u64 = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  srv_args_ptr_t a = SRV_ARGS();
#define SRV_SLOT() (blockIdx.x)
  if (threadIdx.x == 0) {
    srv_slot_ctl *slot = a->ctl + SRV_SLOT();
    bc[20] = bc[21] = 0; /* no code's tables in LDS yet */
    /* a request the previous generation left unserved shows as doorbell != done */
    bc[1] = srv_ld_sys(&slot->done);
    if (blockIdx.x == 0) {
      atomicMax(&a->gctl->last_activity, (long long)wall_clock64());
      srv_st_sys(a->state, 2u * a->gen + 1u);
    }
  }
  for (;;) {
    a = SRV_ARGS();
    if (threadIdx.x < 64) {
      /* wave 0 polls: lanes 0..15 read the 16 words of the slot's request line in ONE load; the request is there when
       * the four chunk tags agree on a number that is not the one served last (ldpc_server.h) */
      const uint32_t *line = reinterpret_cast<const uint32_t *>(a->req + SRV_SLOT());
      const uint32_t last = bc[1], gen = a->gen;
      const int w = blockIdx.x, lane = threadIdx.x;
      uint32_t d, word = 0;
      for (;;) {
        if (lane < 16)
          word = srv_ld_sys(line + lane);
        /* (workgroup 0 also looks at the host's stop word: both host reads are in flight together) */
        const uint32_t hs = (w == 0 && lane == 0) ? srv_ld_sys(a->host_stop) : 0u;
        const uint32_t t0 = __builtin_amdgcn_readlane(word, 0), t1 = __builtin_amdgcn_readlane(word, 4),
                       t2 = __builtin_amdgcn_readlane(word, 8), t3 = __builtin_amdgcn_readlane(word, 12);
        d = t0;
        if (t0 != last && t0 == t1 && t0 == t2 && t0 == t3)
          break;
        if (__hip_atomic_load(&a->gctl->stopping_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
          d = 0xffffffffu;
          break;
        }
        if (w == 0) { /* workgroup 0 decides for everybody: host request, or nobody has called for idle_ticks */
          const long long idle = (long long)wall_clock64() -
                                 __hip_atomic_load(&a->gctl->last_activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)__builtin_amdgcn_readfirstlane(hs) == gen || idle > (long long)a->idle_ticks) {
            __hip_atomic_store(&a->gctl->stopping_gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d = 0xffffffffu;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (lane < 16)
        bc[hdr + lane] = word;
      if (lane == 0) {
        bc[0] = d;
        bc[2] = (uint32_t)wall_clock64();
      }
    }
    __syncthreads();
    const uint32_t d = bc[0];
    if (d == 0xffffffffu)
      break;
    /* the host wrote the payload before the request line; the poll's load has returned, so the payload loads issued from
     * here on see it -- provided they do not hit in a cache: the fast decoder reads the payload with system-scope loads
     * (srv_fast_io::ld_llr), the generic one behind an acquire fence (below).  A system-scope acquire here would drop
     * the XCD's L2 for every request. */
    a = SRV_ARGS();
    if (threadIdx.x == 0)
      bc[3] = (uint32_t)wall_clock64();
    const srv_req *rq = reinterpret_cast<const srv_req *>(bc + hdr);
    hdr ^= 4u ^ 32u;
    const uint32_t kind = LDPC_UNIFORM(rq->kind_mode) & 0xffu;
    /* (readfirstlane returns int: widen as unsigned, or a low dword >= 2^31 smears ones over the high dword) */
    ldpc_code_ptr_t code = (ldpc_code_ptr_t)(((uint64_t)(uint32_t)LDPC_UNIFORM(rq->code_hi) << 32) | (uint64_t)(uint32_t)LDPC_UNIFORM(rq->code_lo));
    const uint8_t *host_in = srv_sgpr(a->in_host + (size_t)SRV_SLOT() * SRV_IN_STRIDE);
    uint8_t *hout = srv_sgpr(a->out_host + (size_t)SRV_SLOT() * SRV_OUT_STRIDE);
    int n_iter = 0;
    if (!ENC && kind == SRV_KIND_DEC_FAST) {
      int tid_l = (int)threadIdx.x;
      const uint8_t *staged = srv_sgpr(a->staging + (size_t)blockIdx.x * SRV_IN_STRIDE);
      /* bc[20], bc[21]: the code whose tables this workgroup's LDS holds (0: none) */
      const bool resident = bc[20] == LDPC_UNIFORM(rq->code_lo) && bc[21] == LDPC_UNIFORM(rq->code_hi);
      const uint32_t *abw = srv_sgpr(a->abort_w + 16u * SRV_SLOT());
      if (LDPC_UNIFORM((int)((rq->kind_mode >> 16) & 0xffu))) {
        const srv_fast_io<true> io{rq, host_in, staged, hout, a, bc + 24, tid_l, resident, d, abw};
        n_iter = ldpc_dec_fast_block(fsm, code, io);
      } else {
        const srv_fast_io<false> io{rq, host_in, staged, hout, a, bc + 24, tid_l, resident, d, abw};
        n_iter = ldpc_dec_fast_block(fsm, code, io);
      }
      if (threadIdx.x == 0) {
        bc[20] = rq->code_lo;
        bc[21] = rq->code_hi;
      }
    } else if (!ENC && kind == SRV_KIND_DEC_GENERIC) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      if (threadIdx.x == 0)
        bc[20] = bc[21] = 0; /* this block overwrites the LDS the fast decoder keeps its tables in */
      ldpc_gblock_io io;
      io.llr = reinterpret_cast<const int8_t *>(host_in); /* read once, straight from the slot's host memory */
      io.out = reinterpret_cast<int8_t *>(hout);
      io.max_pass = LDPC_UNIFORM((int)rq->max_pass);
      io.use_crc = LDPC_UNIFORM((int)((rq->kind_mode >> 16) & 0xffu));
      io.crcE = LDPC_UNIFORM((int)rq->crcE);
      io.crc_pow = a->crc_pow_tbl[LDPC_UNIFORM(rq->kind_mode >> 24) & 3u];
      io.out_mode = LDPC_UNIFORM((int)((rq->kind_mode >> 8) & 0xffu));
      io.tb_abort = nullptr;
      io.trace = nullptr;
      io.trace_stride = 0;
      n_iter = ldpc_dec_generic_block(reinterpret_cast<int8_t *>(fsm), code, io);
    } else if (ENC && kind == SRV_KIND_ENC) {
      /* up to 8 segments of one code side by side, SRV_ENC_GROUP threads each, in lockstep through the phases.  The segment
       * bytes were written by the host while the kernel runs and are read with ordinary loads: acquire first (what the
       * caches hold of the slot's payload area is the previous call's). */
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      const int tid = threadIdx.x, grp = tid / SRV_ENC_GROUP, gt = tid - grp * SRV_ENC_GROUP;
      const int n_seg = LDPC_UNIFORM((int)(rq->kb_nseg >> 16)), Kb = LDPC_UNIFORM((int)(rq->kb_nseg & 0xffffu));
      const int words = ldpc_encp_lds_words(code->ncols, code->kb_full, code->Z, code->nrows, code->nedges);
      ldpc_encp_lds L;
      ldpc_encp_carve(reinterpret_cast<uint32_t *>(fsm) + (size_t)grp * ((words + 3) & ~3), code, L);
      const uint8_t *in = host_in + (size_t)grp * LDPC_UNIFORM(rq->seg_in_stride);
      uint8_t *out = hout + (size_t)grp * LDPC_UNIFORM(rq->seg_out_stride);
      for (int ph = 0; ph < LDPC_ENCP_NUM_PHASES; ph++) {
        if (grp < n_seg)
          ldpc_encp_phase(ph, code, Kb, in, L, out, gt, SRV_ENC_GROUP);
        __syncthreads();
      }
    }
    /* Results -> host.  n_iter and the timing stamps travel in the completion line.
     * Fast decoder: every 16-byte unit of the output area carries the call's sequence number (ldpc_dec_fast_block.h,
     * out_tag) and so does the completion line, one 16-byte store: the host accepts a unit when it shows the number, in
     * whatever order the stores arrive -- no fence, no barrier, nothing to wait for here.
     * Other kinds: plain output bytes; every thread's stores are fenced to system scope before the barrier and only then
     * thread 0 rings, with a system-scope RELEASE store (a plain store was reordered against the result bytes on its way
     * to host memory; a release by the publishing thread alone was overtaken too: about one call in 10^3 came back with
     * stale output bytes). */
    const uint32_t t_decoded = (uint32_t)wall_clock64();
    uint32_t sd = 0, pp = 0;
    if (threadIdx.x == 0) {
      const uint32_t clip = 0xffffu;
      const uint32_t dt_stage = bc[3] - bc[2], dt_dec = t_decoded - bc[3];
      const uint32_t dt_pro = kind == SRV_KIND_DEC_FAST ? bc[24] - bc[3] : 0u, dt_pas = kind == SRV_KIND_DEC_FAST ? bc[25] - bc[24] : 0u;
      sd = (dt_stage < clip ? dt_stage : clip) | ((dt_dec < clip ? dt_dec : clip) << 16);
      pp = (dt_pro < clip ? dt_pro : clip) | ((dt_pas < clip ? dt_pas : clip) << 16);
    }
    if (kind == SRV_KIND_DEC_FAST) {
      if (threadIdx.x == 0) {
        srv_slot_ctl *slot = a->ctl + SRV_SLOT();
        srv_st16_sys(slot, d, (uint32_t)n_iter, sd, pp);
      }
    } else {
      if (threadIdx.x == 0) {
        srv_slot_ctl *slot = a->ctl + SRV_SLOT();
        srv_st_sys(reinterpret_cast<uint32_t *>(&slot->n_iter), (uint32_t)n_iter);
        srv_st_sys(&slot->t_stage_decode, sd);
        srv_st_sys(&slot->t_pro_passes, pp);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      __syncthreads();
      a = SRV_ARGS();
      if (threadIdx.x == 0) {
        srv_slot_ctl *slot = a->ctl + SRV_SLOT();
        __hip_atomic_store(&slot->done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (threadIdx.x == 0) {
      atomicMax(&a->gctl->last_activity, (long long)wall_clock64());
      bc[1] = d;
    }
  }
  /* workgroup 0 publishes "stopped": a caller that sees it relaunches on the same stream, i.e. behind this generation */
  a = SRV_ARGS();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    srv_st_sys(a->state, 2u * a->gen + 2u);
#undef SRV_ARGS
#undef SRV_SLOT
}

int ldpc_server_has_encoder(void) { return 1; }

hipError_t ldpc_server_init(void)
{
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_server_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SRV_LDS_BYTES);
  if (e != hipSuccess)
    return e;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_server_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SRV_LDS_BYTES);
}

hipError_t ldpc_server_launch(const srv_args &a, uint32_t n_slots, hipStream_t stream, int encoder)
{
  if (encoder)
    hipLaunchKernelGGL(ldpc_server_kernel<true>, dim3(n_slots), dim3(SRV_THREADS), SRV_LDS_BYTES, stream, a);
  else
    hipLaunchKernelGGL(ldpc_server_kernel<false>, dim3(n_slots), dim3(SRV_THREADS), SRV_LDS_BYTES, stream, a);
  return hipGetLastError();
}
