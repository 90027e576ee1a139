/*
 * ldpc_server.hip -- resident server kernel behind the per-segment plugin entry points (protocol: ldpc_server.h).
 *
 * One workgroup of 1024 threads per caller slot, owning a whole CU (160 KiB of LDS).  Thread 0 polls the slot's
 * doorbell in page-locked host memory; a request is fetched by all threads straight from host memory ([header |
 * payload], 16 bytes per thread and load, everything in flight at once), staged in device memory (the decoder
 * re-reads the core columns' LLRs every pass), run through the same per-block device functions as the batch kernels
 * (ldpc_dec_fast_block.h / ldpc_dec_generic_block.h / ldpc_enc_packed_core.h) and the results are written straight
 * into the slot's host output area, followed by the completion word.
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
 * disassembly).  With the encoder's phases compiled in as well (SRV_WITH_ENCODER) the spills reach the hot loops, so
 * LDPCencoder calls take the launch path (ldpc_api.cpp) and the encoder job type stays switched off. */
#ifndef SRV_THREADS
#define SRV_THREADS 1024
#endif
#define SRV_ENC_GROUP 128 /* threads per segment of an encoder call: 8 segments side by side in one workgroup */
#define SRV_FETCH ((SRV_IN_STRIDE / 16 + SRV_THREADS - 1) / SRV_THREADS)
#ifndef SRV_WITH_ENCODER
#define SRV_WITH_ENCODER 0
#endif

__device__ __forceinline__ uint32_t srv_ld_sys(const uint32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void srv_st_sys(uint32_t *p, uint32_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

struct srv_fast_io { /* the request header sits in LDS: read where needed (ldpc_dec_fast_block.h) */
  const srv_req *rq;
  const uint8_t *payload;
  uint8_t *hout;
  srv_args_ptr_t a;
  uint32_t *st;
  __device__ __forceinline__ const uint32_t *src32() const { return reinterpret_cast<const uint32_t *>(payload); }
  __device__ __forceinline__ int8_t *out() const { return reinterpret_cast<int8_t *>(hout); }
  __device__ __forceinline__ int max_pass() const { return LDPC_UNIFORM((int)rq->max_pass); }
  __device__ __forceinline__ int use_crc() const { return LDPC_UNIFORM((int)rq->use_crc); }
  __device__ __forceinline__ int crcE() const { return LDPC_UNIFORM((int)rq->crcE); }
  __device__ __forceinline__ const uint32_t *crc_pow() const { return a->crc_pow_tbl[LDPC_UNIFORM(rq->crc_type) & 3u]; }
  __device__ __forceinline__ int out_mode() const { return LDPC_UNIFORM((int)rq->out_mode); }
  __device__ __forceinline__ int *tb_abort() const { return nullptr; }
  __device__ __forceinline__ uint32_t *stamps() const { return st; }
  int tid_;
  __device__ __forceinline__ int tid() const { return tid_; }
  __device__ __forceinline__ bool eager_check() const { return true; }
};

__global__ void __launch_bounds__(SRV_THREADS) ldpc_server_kernel(const srv_args args_by_value)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  uint32_t *bc = reinterpret_cast<uint32_t *>(fsm + SRV_BC_OFF); /* [0] doorbell / quit, [1] last served, [2], [3] time stamps, [4..19] request header */
  /* laundered wherever it is used: the loads behind it must not be hoisted out of the loop and kept live */
#define SRV_ARGS() ({ uint64_t p_ = args_u64; asm volatile("" : "+s"(p_)); (srv_args_ptr_t)p_; })
  (void)args_by_value; /* = the kernarg segment, read through the laundered pointer */
  const uint64_t args_u64 = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
  srv_args_ptr_t a = SRV_ARGS();
  if (threadIdx.x == 0) {
    srv_slot_ctl *slot = a->ctl + blockIdx.x;
    bc[1] = srv_ld_sys(&slot->done); /* a request the previous generation left unserved shows as doorbell != done */
    if (blockIdx.x == 0) {
      atomicMax(&a->gctl->last_activity, (long long)wall_clock64());
      srv_st_sys(a->state, 2u * a->gen + 1u);
    }
  }
  for (;;) {
    a = SRV_ARGS();
    if (threadIdx.x == 0) {
      srv_slot_ctl *slot = a->ctl + blockIdx.x;
      const uint32_t last = bc[1], gen = a->gen;
      const int w = blockIdx.x;
      uint32_t d;
      for (;;) {
        /* both host words are requested before either is looked at: one PCIe round trip per poll */
        d = srv_ld_sys(&slot->doorbell);
        const uint32_t hs = w == 0 ? srv_ld_sys(a->host_stop) : 0u;
        if (d != last)
          break;
        if (__hip_atomic_load(&a->gctl->stopping_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
          d = 0xffffffffu;
          break;
        }
        if (w == 0) { /* workgroup 0 decides for everybody: host request, or nobody has called for idle_ticks */
          const long long idle = (long long)wall_clock64() -
                                 __hip_atomic_load(&a->gctl->last_activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (hs == gen || idle > (long long)a->idle_ticks) {
            __hip_atomic_store(&a->gctl->stopping_gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d = 0xffffffffu;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(4);
      }
      bc[0] = d;
      bc[2] = (uint32_t)wall_clock64();
    }
    __syncthreads();
    const uint32_t d = bc[0];
    if (d == 0xffffffffu)
      break;
    /* the host wrote [header | payload] before the doorbell: order our loads behind the doorbell load */
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    a = SRV_ARGS();
    {
      const uint32_t n16 = d & 0xfffu;
      const uint4 *hin = reinterpret_cast<const uint4 *>(a->in_host + (size_t)blockIdx.x * SRV_IN_STRIDE);
      uint4 *stg = reinterpret_cast<uint4 *>(a->staging + (size_t)blockIdx.x * SRV_IN_STRIDE);
      /* SRV_FETCH loads per thread, all in flight together: SRV_FETCH * SRV_THREADS * 16 bytes >= SRV_IN_STRIDE */
      static_assert((size_t)SRV_FETCH * SRV_THREADS * 16 >= SRV_IN_STRIDE, "a request must fit the fetch");
      uint4 v[SRV_FETCH];
#pragma unroll
      for (int k = 0; k < SRV_FETCH; k++) {
        const uint32_t i = threadIdx.x + (uint32_t)k * SRV_THREADS;
        v[k] = make_uint4(0, 0, 0, 0);
        if (i < n16)
          v[k] = hin[i];
      }
#pragma unroll
      for (int k = 0; k < SRV_FETCH; k++) {
        const uint32_t i = threadIdx.x + (uint32_t)k * SRV_THREADS;
        if (i < n16)
          stg[i] = v[k];
      }
      const uint4 v0 = v[0];
      if (threadIdx.x < 4)
        reinterpret_cast<uint4 *>(bc + 4)[threadIdx.x] = v0;
    }
    __threadfence(); /* the staged payload is re-read by other waves of this workgroup through L1 / L2 */
    __syncthreads();
    a = SRV_ARGS();
    if (threadIdx.x == 0)
      bc[3] = (uint32_t)wall_clock64();
    const srv_req *rq = reinterpret_cast<const srv_req *>(bc + 4);
    const uint32_t kind = LDPC_UNIFORM(rq->kind);
    /* (readfirstlane returns int: widen as unsigned, or a low dword >= 2^31 smears ones over the high dword) */
    ldpc_code_ptr_t code = (ldpc_code_ptr_t)(((uint64_t)(uint32_t)LDPC_UNIFORM((uint32_t)(rq->code >> 32)) << 32) |
                                             (uint64_t)(uint32_t)LDPC_UNIFORM((uint32_t)rq->code));
    const uint8_t *payload = srv_sgpr(a->staging + (size_t)blockIdx.x * SRV_IN_STRIDE + SRV_REQ_BYTES);
    uint8_t *hout = srv_sgpr(a->out_host + (size_t)blockIdx.x * SRV_OUT_STRIDE);
    int n_iter = 0;
    if (kind == SRV_KIND_DEC_FAST) {
      int tid_l = (int)threadIdx.x;
      asm volatile("" : "+v"(tid_l)); /* per iteration: nothing derived from it is hoisted out of the server loop */
      const srv_fast_io io{rq, payload, hout, a, bc + 24, tid_l};
      n_iter = ldpc_dec_fast_block(fsm, code, io);
    } else if (kind == SRV_KIND_DEC_GENERIC) {
      ldpc_gblock_io io;
      io.llr = reinterpret_cast<const int8_t *>(payload);
      io.out = reinterpret_cast<int8_t *>(hout);
      io.max_pass = LDPC_UNIFORM((int)rq->max_pass);
      io.use_crc = LDPC_UNIFORM((int)rq->use_crc);
      io.crcE = LDPC_UNIFORM((int)rq->crcE);
      io.crc_pow = a->crc_pow_tbl[LDPC_UNIFORM(rq->crc_type) & 3u];
      io.out_mode = LDPC_UNIFORM((int)rq->out_mode);
      io.tb_abort = nullptr;
      n_iter = ldpc_dec_generic_block(reinterpret_cast<int8_t *>(fsm), code, io);
    } else if (SRV_WITH_ENCODER && kind == SRV_KIND_ENC) {
      /* up to 8 segments of one code side by side, SRV_ENC_GROUP threads each, in lockstep through the phases */
      const int tid = threadIdx.x, grp = tid / SRV_ENC_GROUP, gt = tid - grp * SRV_ENC_GROUP;
      const int n_seg = LDPC_UNIFORM((int)rq->n_seg), Kb = LDPC_UNIFORM((int)rq->Kb);
      const int words = ldpc_encp_lds_words(code->ncols, code->kb_full, code->Z, code->nrows, code->nedges);
      ldpc_encp_lds L;
      ldpc_encp_carve(reinterpret_cast<uint32_t *>(fsm) + (size_t)grp * ((words + 3) & ~3), code, L);
      const uint8_t *in = payload + (size_t)grp * LDPC_UNIFORM(rq->seg_in_stride);
      uint8_t *out = hout + (size_t)grp * LDPC_UNIFORM(rq->seg_out_stride);
      for (int ph = 0; ph < LDPC_ENCP_NUM_PHASES; ph++) {
        if (grp < n_seg)
          ldpc_encp_phase(ph, code, Kb, in, L, out, gt, SRV_ENC_GROUP);
        __syncthreads();
      }
    }
    /* results -> host, then the completion word: every thread's stores are out before thread 0 rings */
    const uint32_t t_decoded = (uint32_t)wall_clock64();
    __threadfence_system();
    __syncthreads();
    a = SRV_ARGS();
    if (threadIdx.x == 0) {
      srv_slot_ctl *slot = a->ctl + blockIdx.x;
      srv_st_sys(reinterpret_cast<uint32_t *>(&slot->n_iter), (uint32_t)n_iter);
      srv_st_sys(&slot->t_seen, bc[2]);
      srv_st_sys(&slot->t_staged, bc[3]);
      srv_st_sys(&slot->t_decoded, t_decoded);
      srv_st_sys(&slot->t_prologue, bc[24]);
      srv_st_sys(&slot->t_passes, bc[25]);
      __hip_atomic_store(&slot->done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      atomicMax(&a->gctl->last_activity, (long long)wall_clock64());
      bc[1] = d;
    }
  }
  /* workgroup 0 publishes "stopped": a caller that sees it relaunches on the same stream, i.e. behind this generation */
  a = SRV_ARGS();
  if (threadIdx.x == 0 && blockIdx.x == 0)
    srv_st_sys(a->state, 2u * a->gen + 2u);
#undef SRV_ARGS
}

int ldpc_server_has_encoder(void) { return SRV_WITH_ENCODER; }

hipError_t ldpc_server_init(void)
{
  return hipFuncSetAttribute(reinterpret_cast<const void *>(ldpc_server_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             SRV_LDS_BYTES);
}

hipError_t ldpc_server_launch(const srv_args &a, uint32_t n_slots, hipStream_t stream)
{
  hipLaunchKernelGGL(ldpc_server_kernel, dim3(n_slots), dim3(SRV_THREADS), SRV_LDS_BYTES, stream, a);
  return hipGetLastError();
}
