/*
 * ldpc_kernels.h -- launch interface between the C-ABI layer (ldpc_api.cpp) and the HIP kernels.
 */
#ifndef LDPC_KERNELS_H
#define LDPC_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ldpc_graph.h"

#define LDPC_CRC_POW_LEN 8448 /* x^j mod g for j < 8448 = largest code block */

/* heterogeneous batches (transport-block chain): one job per workgroup overrides code / buffers / E / crc */
#define LDPC_JOB_MUTE_CHECK 0x100
struct ldpc_dec_job {
  const ldpc_code_desc_t *code; /* device */
  uint64_t llr_off;             /* bytes from ldpc_dec_args.llr */
  uint64_t out_off;             /* bytes from ldpc_dec_args.out */
  int32_t num_max_iter;
  int32_t E;                    /* CRC mode: bits covered */
  int32_t crc_type;             /* CRC mode: index into ldpc_dec_args.crc_pow_tbl (bits 0..7); LDPC_JOB_MUTE_CHECK: extension-row items
                                   whose degree-1 bits have channel LLR 0 are looked for and not run (ldpc_dec_fast_block.h) */
  int32_t iter_idx;             /* where in ldpc_dec_args.n_iter this block reports */
  int32_t abort_idx;            /* index into ldpc_dec_args.tb_abort of the block's transport block, -1: none */
  int32_t seg_idx;              /* fused segment kernel (tb_rx_fused.hip): index of the segment's tb_rx_seg_job -- the workgroup
                                   de-matches the segment first and ends with the chain's epilogue; -1: a plain decoder job */
};
/* several jobs of ONE small code, the same iteration cap and the same CRC in one workgroup (ldpc_dec_fast_mblock.h): the
 * transport-block chain groups a batch's small segments this way; the jobs of a group are consecutive in the job array */
struct ldpc_dec_mgroup {
  const ldpc_code_desc_t *code; /* device: the code's several-blocks-per-workgroup descriptor */
  uint32_t first_job;           /* index into ldpc_dec_args.jobs */
  uint32_t n_valid;             /* jobs in this workgroup, <= f_mb * f_sub */
  int32_t num_max_iter, E, crc_type;
  int32_t pad;
};
struct ldpc_enc_job {
  const ldpc_code_desc_t *code; /* device, full-rate descriptor */
  uint64_t in_off, out_off;     /* bytes from ldpc_enc_args.in / .out */
  int32_t Kb;
  int32_t pad;
};

struct ldpc_dec_args {
  const ldpc_code_desc_t *code; /* device copy of the descriptor */
  const int8_t *llr;
  uint32_t llr_stride;
  int8_t *out;
  uint32_t out_stride;
  int32_t *n_iter;
  int32_t num_max_iter;
  int32_t out_mode; /* 0 packed bits, else one bit per byte */
  int32_t use_crc;  /* 0: parity-check stop, 1: CRC stop */
  int32_t E;        /* bits covered by the CRC check (use_crc) */
  const uint32_t *crc_pow; /* device table, crc_pow[j] = x^j mod g(x), left aligned in 32 bits */
  const ldpc_dec_job *jobs; /* NULL: homogeneous batch addressed by strides */
  const uint32_t *crc_pow_tbl[4]; /* per crc_type, used with jobs */
  int *tb_abort;            /* with jobs: per transport block "a segment failed" flags (zero on entry), or NULL */
  const ldpc_dec_mgroup *mgroups; /* ldpc_launch_dec_fast_multi_jobs: one group per workgroup */
  /* host-buffer batches (ldpc_launch_dec_fast_pull): the workgroup first fetches its row from `pull` -- page-locked host
   * memory, device-mapped address -- into its row of `llr` (device memory, written here), then decodes from there */
  const int8_t *pull;
  uint32_t pull_stride;
  uint32_t n_blocks; /* multi-block launches: blocks in the launch (the last workgroup may hold fewer than f_mb) */
  uint32_t pull_stagger_ticks, pull_first_round; /* pull launches: see ldpc_dec_fast_pull_kernel */
  /* diagnostics (NRLDPC_HIP_DEC_TRACE=<file>, homogeneous fast launches): 32 x uint64 per workgroup -- placement, start / end
   * and the clocks of every phase of every pass (ldpc_dec_fast_kernel<.., .., true>; tools/dec_trace.py) */
  unsigned long long *trace;
  /* > 0: workgroups of this launch share a CU, each with this many waves per SIMD -- they take turns at the issue priority
   * (ldpc_dec_fast_block.h, io.fair_turns()) */
  int32_t fair;
};

struct ldpc_enc_args {
  const ldpc_code_desc_t *code; /* full-rate descriptor (R13 / R15) */
  const uint8_t *in;
  uint32_t in_stride;
  uint8_t *out;
  uint32_t out_stride;
  int32_t Kb;
  const ldpc_enc_job *jobs; /* NULL: homogeneous batch */
};

/* one-time per-process kernel attribute setup (dynamic LDS limit); returns hipSuccess or the error */
hipError_t ldpc_kernels_init(void);
/* generic flooding min-sum decoder: one workgroup per code block, any (BG, Z, R) */
hipError_t ldpc_launch_dec_generic(const ldpc_dec_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks,
                                   hipStream_t stream);
/* the same decoder with the CRC stop left to a predicate on the host: every pass runs, the hard decisions of block b after
 * pass p >= 3 go to trace + (b * n_trace + p - 3) * trace_stride in a.out_mode's format; a.out is not written */
hipError_t ldpc_launch_dec_generic_trace(const ldpc_dec_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks, int8_t *trace,
                                         uint32_t trace_stride, uint32_t n_trace, hipStream_t stream);
/* job-array launches: explicit workgroup size and dynamic LDS (maxima over the jobs) */
hipError_t ldpc_launch_dec_generic_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream);
/* zc: the lifting size EVERY job of the launch has (one block per workgroup: f_rstride = Z + 4), or 0 */
hipError_t ldpc_launch_dec_fast_jobs(const ldpc_dec_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream, int zc = 0);
bool ldpc_fast_zc_enabled(int zc); /* is there an instantiation for this lifting size (and is it switched on: NRLDPC_HIP_ZC) */
/* The lifting sizes whose one-block-per-workgroup decoder kernels (batch, job array, fused segment kernel) have instantiations of
 * their own with compile-time row strides (ldpc_dec_fast_block.h ZC): the sizes the segments of large transport blocks come in
 * (K' between 0.6 and 1.0 of the largest code block) and BASELINE configs[2]'s 208.  LDPC_FAST_ZC=0 (build switch): none. */
#ifndef LDPC_FAST_ZC
#define LDPC_FAST_ZC 1
#endif
#if LDPC_FAST_ZC
#define LDPC_FAST_ZC_LIST(X) X(384) X(352) X(320) X(288) X(256) X(208)
#else
#define LDPC_FAST_ZC_LIST(X)
#endif
hipError_t ldpc_launch_enc_jobs(const ldpc_enc_args &a, int n_threads, int lds_bytes, uint32_t n_blocks, hipStream_t stream);
/* fast decoder (Zc % 4 == 0, 4-byte aligned LLR rows, hc.f_ok): one workgroup per code block */
hipError_t ldpc_fast_kernel_init(void);
hipError_t ldpc_launch_dec_fast(const ldpc_dec_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks,
                                hipStream_t stream);
/* the same with a.pull set: every workgroup pulls its LLR row over the link itself (no copy engine, no staging copy) */
hipError_t ldpc_launch_dec_fast_pull(const ldpc_dec_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks,
                                     hipStream_t stream);
/* small lifting sizes, homogeneous batch: host_code.f_mb blocks per workgroup (ldpc_dec_fast_mblock.h) */
hipError_t ldpc_launch_dec_fast_multi(const ldpc_dec_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks,
                                      hipStream_t stream);
/* the same for groups of jobs (a.jobs, a.mgroups): sub = 1 or 4 = f_sub of the groups' descriptors; workgroup size and LDS =
 * the maxima over the groups */
hipError_t ldpc_launch_dec_fast_multi_jobs(const ldpc_dec_args &a, int sub, int n_threads, int lds_bytes, uint32_t n_groups,
                                           hipStream_t stream);
/* encoder: one workgroup per code block; workgroup size and dynamic LDS of the selected encoder kernel for a code */
int ldpc_enc_is_packed(void); /* 1: bit-packed kernel selected (default), 0: NRLDPC_HIP_ENC_KERNEL=bytes */
void ldpc_enc_launch_shape(const ldpc_code_desc_t &host_code, int *n_threads, int *lds_bytes);
hipError_t ldpc_launch_enc(const ldpc_enc_args &a, const ldpc_code_desc_t &host_code, uint32_t n_blocks,
                           hipStream_t stream);
#endif
