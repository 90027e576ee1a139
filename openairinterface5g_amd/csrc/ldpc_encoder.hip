/*
 * ldpc_encoder.hip -- NR LDPC encoder kernels for gfx950 (MI355X), one workgroup per code block, the code word
 * resident in LDS: the bit-packed kernel (ldpc_enc_packed_core.h, default) and the byte-per-lane kernel
 * (ldpc_enc_core.h, <= 26 KiB + 1.5 KiB scratch), bit-identical outputs.
 * Replaces LDPCencoder (reference nrLDPC_encoder/ldpc_encoder.c:44-252, ldpc_encoder_optim8segmulti.c:46-213).
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "ldpc_kernels.h"
#include "ldpc_enc_core.h"
#include "ldpc_enc_packed_core.h"
#include "ldpc_enc_packed32.h"

__global__ void __launch_bounds__(1024) ldpc_enc_kernel(const ldpc_enc_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
  typedef const ldpc_enc_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  uint8_t *x = esm;
  uint8_t *lam = esm + ((code->ncols * code->Z + 15) & ~15);
  const uint32_t blk = blockIdx.x;
  const uint8_t *in = a.in + (job ? (size_t)job->in_off : (size_t)blk * a.in_stride);
  uint8_t *out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
  const int Kb = job ? job->Kb : a.Kb;
#pragma unroll
  for (int ph = 0; ph < LDPC_ENC_NUM_PHASES; ph++) {
    ldpc_enc_phase(ph, code, Kb, in, x, lam, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
}

/* Bit-packed variant (ldpc_enc_packed_core.h): 32 lanes per XOR, <= 8 KiB of LDS per block, several blocks per CU. */
__global__ void __launch_bounds__(512) ldpc_enc_packed_kernel(const ldpc_enc_args a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t esm[];
  typedef const ldpc_enc_job LDPC_CONST_AS *job_ptr_t;
  const job_ptr_t job = a.jobs ? (job_ptr_t)a.jobs + blockIdx.x : (job_ptr_t) nullptr;
  ldpc_code_ptr_t code = (ldpc_code_ptr_t)(job ? job->code : a.code);
  ldpc_encp_lds L;
  ldpc_encp_carve(reinterpret_cast<uint32_t *>(esm), code, L);
  const uint32_t blk = blockIdx.x;
  const uint8_t *in = a.in + (job ? (size_t)job->in_off : (size_t)blk * a.in_stride);
  uint8_t *out = a.out + (job ? (size_t)job->out_off : (size_t)blk * a.out_stride);
#ifdef ENC_PHASE_STOP /* diagnostic build (tools/enc_phase_cost.py): Kb's second byte = the phase after which the kernel returns */
  const int stop_after = ((job ? job->Kb : a.Kb) >> 8) & 15;
  const int Kb = (job ? job->Kb : a.Kb) & 0xff;
#define ENC_STOP(k) do { if (stop_after == (k)) return; } while (0)
  ENC_STOP(9); /* nothing but the launch */
#else
  const int Kb = job ? job->Kb : a.Kb;
#define ENC_STOP(k) do { } while (0)
#endif
  if (ldpc_encp32_applies(code) && ((reinterpret_cast<uintptr_t>(in) & 3) | (reinterpret_cast<uintptr_t>(out) & 15)) == 0) {
    /* Zc % 32 == 0 (ldpc_enc_packed32.h): five barriers, no periodic-extension phases, 16-byte output stores */
    const int tid = threadIdx.x, nt = blockDim.x;
    const ldpc_encp32 g = ldpc_encp32_make(code, Kb);
    const uint32_t nd = (uint32_t)(g.kbf * g.W);
    const uint32_t *__restrict__ in32 = reinterpret_cast<const uint32_t *>(in);
    /* everything that comes from global memory is requested before anything is consumed */
    uint32_t g_in[2], g_et[2], g_rp = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * (uint32_t)nt;
      g_in[k] = i < nd ? in32[i] : 0u;
      g_et[k] = i < (uint32_t)code->nedges ? code->enc_et[i] : 0u;
    }
    if (tid <= code->nrows)
      g_rp = (uint32_t)code->row_ptr[tid];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t i = (uint32_t)tid + (uint32_t)k * (uint32_t)nt;
      if (i < nd)
        ldpc_encp32_info(L, g, i, g_in[k]);
      if (i < (uint32_t)code->nedges)
        L.ET[i] = g_et[k];
    }
    for (uint32_t i = (uint32_t)tid + 2u * (uint32_t)nt; i < nd; i += nt) /* (workgroups of fewer than 192 threads) */
      ldpc_encp32_info(L, g, i, in32[i]);
    for (int e = tid + 2 * nt; e < code->nedges; e += nt)
      L.ET[e] = code->enc_et[e];
    if (tid <= code->nrows)
      L.RP[tid] = g_rp;
    for (int r = tid + nt; r <= code->nrows; r += nt)
      L.RP[r] = (uint32_t)code->row_ptr[r];
    __syncthreads();
    ENC_STOP(1);
    ldpc_encp32_lambda(L, g, tid, nt);
    __syncthreads();
    ENC_STOP(2);
    if (tid < 64)
      ldpc_encp32_core_parity_wave(code, L, g, tid);
    __syncthreads();
    ENC_STOP(3);
    ldpc_encp32_extension(code, L, g, tid, nt);
    __syncthreads();
    ENC_STOP(4);
    /* 1024 x 25 KB at the end of 1024 workgroups that move in step: 3.9 of the kernel's 11.1 us, at ~10 B/clk per CU = 6.7 TB/s
     * chip-wide, the rate the memory system takes stores at.  Sending the information columns out early (from the waves that
     * idle during the core parity steps, behind LDS-only barriers) moved 1.2 us of it forward and the kernel's end not at
     * all: profiles/r05/enc_phase_cost.txt */
    ldpc_encp32_store_bytes(L, g, out, 2, code->ncols, tid, nt);
    return;
  }
  for (int ph = 0; ph <= 3; ph++) {
    ldpc_encp_phase(ph, code, Kb, in, L, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
  if (threadIdx.x < 64) /* phases 4 .. 11: one wave, no workgroup barriers in between */
    ldpc_encp_core_parity_wave(code, L, (int)threadIdx.x);
  __syncthreads();
  for (int ph = 12; ph < LDPC_ENCP_NUM_PHASES; ph++) {
    ldpc_encp_phase(ph, code, Kb, in, L, out, threadIdx.x, blockDim.x);
    __syncthreads();
  }
}

/* NRLDPC_HIP_ENC_KERNEL=bytes selects the byte-per-lane kernel (A/B runs, cross-check); default: bit-packed */
static int enc_packed(void)
{
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("NRLDPC_HIP_ENC_KERNEL");
    v = (e && strcmp(e, "bytes") == 0) ? 0 : 1;
  }
  return v;
}

int ldpc_enc_is_packed(void) { return enc_packed(); }

void ldpc_enc_launch_shape(const ldpc_code_desc_t &hc, int *n_threads, int *lds_bytes)
{
  if (enc_packed()) {
    *n_threads = ldpc_encp_threads(hc.nrows, hc.Z);
    static const int thr_env = [] { const char *e = getenv("NRLDPC_HIP_ENC_THREADS"); return e ? atoi(e) : 0; }(); /* A/B knob */
    if (thr_env >= 64 && thr_env <= 512 && (thr_env & 63) == 0)
      *n_threads = thr_env;
    *lds_bytes = 4 * ldpc_encp_lds_words(hc.ncols, hc.kb_full, hc.Z, hc.nrows, hc.nedges);
  } else {
    int waves = (hc.Z + 63) / 64 * 2;
    *n_threads = (waves > 16 ? 16 : waves) * 64;
    *lds_bytes = ((hc.ncols * hc.Z + 15) & ~15) + ((4 * hc.Z + 15) & ~15);
  }
}

hipError_t ldpc_launch_enc_jobs(const ldpc_enc_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream)
{
  if (n_blocks == 0)
    return hipSuccess;
#ifdef ENC_PHASE_STOP
  if (enc_packed()) {
    ldpc_enc_args b = a;
    const char *e = getenv("NRLDPC_HIP_ENC_STOP");
    if (e)
      b.Kb |= atoi(e) << 8;
    hipLaunchKernelGGL(ldpc_enc_packed_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, b);
    return hipGetLastError();
  }
#endif
  if (enc_packed())
    hipLaunchKernelGGL(ldpc_enc_packed_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  else
    hipLaunchKernelGGL(ldpc_enc_kernel, dim3(n_blocks), dim3(n_threads), lds_bytes, stream, a);
  return hipGetLastError();
}

hipError_t ldpc_launch_enc(const ldpc_enc_args &a, const ldpc_code_desc_t &hc, uint32_t n_blocks, hipStream_t stream)
{
  int n_threads, lds_bytes;
  ldpc_enc_launch_shape(hc, &n_threads, &lds_bytes);
  return ldpc_launch_enc_jobs(a, n_threads, lds_bytes, n_blocks, stream);
}
