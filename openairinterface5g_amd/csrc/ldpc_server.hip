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

#define SRV_THREADS 1024
#define SRV_ENC_GROUP 128 /* threads per segment of an encoder call: 8 segments side by side in one workgroup */

__device__ __forceinline__ uint32_t srv_ld_sys(const uint32_t *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void srv_st_sys(uint32_t *p, uint32_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* Out of line on purpose: each body gets the register allocation it has in its own batch kernel (the fast decoder sits
 * at 127 of the 128 VGPRs a 1024-thread workgroup may use; inlined next to the server loop's live values it spills). */
/* arguments of a non-kernel function arrive in VGPRs: make the wave-uniform ones scalar again, so that everything
 * derived from the descriptor stays in SGPRs / s_load as in the batch kernels */
template <typename T> __device__ __forceinline__ T *srv_uniform_ptr(T *p)
{
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  return reinterpret_cast<T *>(((uint64_t)LDPC_UNIFORM((uint32_t)(v >> 32)) << 32) | LDPC_UNIFORM((uint32_t)v));
}
__device__ __forceinline__ ldpc_code_ptr_t srv_uniform_code(uint64_t v)
{
  return (ldpc_code_ptr_t)(((uint64_t)LDPC_UNIFORM((uint32_t)(v >> 32)) << 32) | LDPC_UNIFORM((uint32_t)v));
}
__device__ __noinline__ int srv_dec_fast(uint64_t code_addr, const uint32_t *src32, int8_t *out, int max_pass, int use_crc, int crcE,
                                         const uint32_t *crc_pow, int out_mode)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  ldpc_block_io io;
  io.src32 = srv_uniform_ptr(src32);
  io.out = srv_uniform_ptr(out);
  io.max_pass = LDPC_UNIFORM(max_pass);
  io.use_crc = LDPC_UNIFORM(use_crc);
  io.crcE = LDPC_UNIFORM(crcE);
  io.crc_pow = srv_uniform_ptr(crc_pow);
  io.out_mode = LDPC_UNIFORM(out_mode);
  io.tb_abort = nullptr;
  return ldpc_dec_fast_block(fsm, srv_uniform_code(code_addr), io);
}
__device__ __noinline__ int srv_dec_generic(uint64_t code_addr, const int8_t *llr, int8_t *out, int max_pass, int use_crc, int crcE,
                                            const uint32_t *crc_pow, int out_mode)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  ldpc_gblock_io io;
  io.llr = srv_uniform_ptr(llr);
  io.out = srv_uniform_ptr(out);
  io.max_pass = LDPC_UNIFORM(max_pass);
  io.use_crc = LDPC_UNIFORM(use_crc);
  io.crcE = LDPC_UNIFORM(crcE);
  io.crc_pow = srv_uniform_ptr(crc_pow);
  io.out_mode = LDPC_UNIFORM(out_mode);
  io.tb_abort = nullptr;
  return ldpc_dec_generic_block(reinterpret_cast<int8_t *>(fsm), srv_uniform_code(code_addr), io);
}
__device__ __noinline__ void srv_encode(uint64_t code_addr, const uint8_t *payload_, uint8_t *hout_, int n_seg_, int Kb_,
                                        uint32_t seg_in_stride_, uint32_t seg_out_stride_)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  ldpc_code_ptr_t code = srv_uniform_code(code_addr);
  const uint8_t *payload = srv_uniform_ptr(payload_);
  uint8_t *hout = srv_uniform_ptr(hout_);
  const int n_seg = LDPC_UNIFORM(n_seg_), Kb = LDPC_UNIFORM(Kb_);
  const uint32_t seg_in_stride = LDPC_UNIFORM(seg_in_stride_), seg_out_stride = LDPC_UNIFORM(seg_out_stride_);
  /* up to 8 segments of one code side by side, SRV_ENC_GROUP threads each, in lockstep through the phases */
  const int tid = threadIdx.x, grp = tid / SRV_ENC_GROUP, gt = tid - grp * SRV_ENC_GROUP;
  const int words = ldpc_encp_lds_words(code->ncols, code->kb_full, code->Z, code->nrows, code->nedges);
  ldpc_encp_lds L;
  ldpc_encp_carve(reinterpret_cast<uint32_t *>(fsm) + (size_t)grp * ((words + 3) & ~3), code, L);
  const uint8_t *in = payload + (size_t)grp * seg_in_stride;
  uint8_t *out = hout + (size_t)grp * seg_out_stride;
  for (int ph = 0; ph < LDPC_ENCP_NUM_PHASES; ph++) {
    if (grp < n_seg)
      ldpc_encp_phase(ph, code, Kb, in, L, out, gt, SRV_ENC_GROUP);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(SRV_THREADS) ldpc_server_kernel(const srv_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  uint32_t *bc = reinterpret_cast<uint32_t *>(fsm + SRV_BC_OFF); /* [0] doorbell / quit, [4..19] request header */
  const int tid = threadIdx.x, w = blockIdx.x;
  srv_slot_ctl *slot = a.ctl + w;
  const uint4 *hin = reinterpret_cast<const uint4 *>(a.in_host + (size_t)w * SRV_IN_STRIDE);
  uint4 *stg = reinterpret_cast<uint4 *>(a.staging + (size_t)w * SRV_IN_STRIDE);
  uint8_t *hout = a.out_host + (size_t)w * SRV_OUT_STRIDE;
  uint32_t last = 0;
  if (tid == 0) {
    last = srv_ld_sys(&slot->done); /* a request the previous generation left unserved shows as doorbell != done */
    if (w == 0) {
      atomicMax(&a.gctl->last_activity, (long long)wall_clock64());
      srv_st_sys(a.state, 2u * a.gen + 1u);
    }
  }
  for (;;) {
    if (tid == 0) {
      uint32_t d;
      for (;;) {
        /* both host words are requested before either is looked at: one PCIe round trip per poll */
        d = srv_ld_sys(&slot->doorbell);
        const uint32_t hs = w == 0 ? srv_ld_sys(a.host_stop) : 0u;
        if (d != last)
          break;
        if (__hip_atomic_load(&a.gctl->stopping_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.gen) {
          d = 0xffffffffu;
          break;
        }
        if (w == 0) { /* workgroup 0 decides for everybody: host request, or nobody has called for idle_ticks */
          const long long idle = (long long)wall_clock64() -
                                 __hip_atomic_load(&a.gctl->last_activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (hs == a.gen || idle > (long long)a.idle_ticks) {
            __hip_atomic_store(&a.gctl->stopping_gen, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d = 0xffffffffu;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(4);
      }
      bc[0] = d;
    }
    __syncthreads();
    const uint32_t d = bc[0];
    if (d == 0xffffffffu)
      break;
    /* the host wrote [header | payload] before the doorbell: order our loads behind the doorbell load */
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    const uint32_t n16 = d & 0xfffu;
    {
      const uint32_t i0 = (uint32_t)tid, i1 = (uint32_t)tid + SRV_THREADS;
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
      if (i0 < n16)
        v0 = hin[i0];
      if (i1 < n16)
        v1 = hin[i1];
      if (i0 < n16)
        stg[i0] = v0;
      if (i1 < n16)
        stg[i1] = v1;
      if (tid < 4)
        reinterpret_cast<uint4 *>(bc + 4)[tid] = v0;
    }
    __threadfence(); /* the staged payload is re-read by other waves of this workgroup through L1 / L2 */
    __syncthreads();
    const srv_req *rq = reinterpret_cast<const srv_req *>(bc + 4);
    const uint32_t kind = rq->kind;
    const uint64_t code_addr = rq->code;
    const uint8_t *payload = reinterpret_cast<const uint8_t *>(stg) + SRV_REQ_BYTES;
    const uint32_t *crc_pow = a.crc_pow_tbl[rq->crc_type & 3u];
    int n_iter = 0;
    if (kind == SRV_KIND_DEC_FAST)
      n_iter = srv_dec_fast(code_addr, reinterpret_cast<const uint32_t *>(payload), reinterpret_cast<int8_t *>(hout), (int)rq->max_pass,
                            (int)rq->use_crc, (int)rq->crcE, crc_pow, (int)rq->out_mode);
    else if (kind == SRV_KIND_DEC_GENERIC)
      n_iter = srv_dec_generic(code_addr, reinterpret_cast<const int8_t *>(payload), reinterpret_cast<int8_t *>(hout), (int)rq->max_pass,
                               (int)rq->use_crc, (int)rq->crcE, crc_pow, (int)rq->out_mode);
    else if (kind == SRV_KIND_ENC)
      srv_encode(code_addr, payload, hout, (int)rq->n_seg, (int)rq->Kb, rq->seg_in_stride, rq->seg_out_stride);
    /* results -> host, then the completion word: every thread's stores are out before thread 0 rings */
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
      srv_st_sys(reinterpret_cast<uint32_t *>(&slot->n_iter), (uint32_t)n_iter);
      __hip_atomic_store(&slot->done, d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      atomicMax(&a.gctl->last_activity, (long long)wall_clock64());
      last = d;
    }
  }
  /* workgroup 0 publishes "stopped": a caller that sees it relaunches on the same stream, i.e. behind this generation */
  if (tid == 0 && w == 0)
    srv_st_sys(a.state, 2u * a.gen + 2u);
}

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
