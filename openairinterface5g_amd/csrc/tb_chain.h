/*
 * tb_chain.h -- device jobs of the transport-block chain around the LDPC codec (SURVEY section 8 rows a16-a20,
 * f2, f3): TB CRC attach, code-block segmentation + CB CRC, rate matching + bit interleaving (TX);
 * de-interleaving + rate de-matching with HARQ soft combining + int16->int8 pack, TB reassembly + TB CRC (RX).
 */
#ifndef TB_CHAIN_H
#define TB_CHAIN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tb_jobs.h"

/* TB CRC attach in two steps: per-chunk partial CRCs XOR-ed into acc[tb], then the CRC bytes.  acc[] must be zero on
 * entry and is zero again on exit (uploaded as zeros with the plan; no memset per call) */
/* with_final = 0: only the partial CRCs (the fused TX kernel places the CRC bytes itself and clears acc) */
hipError_t tb_launch_tx_crc(const tb_tx_tb_job *jobs, uint32_t n_tb, const tb_crc_chunk_job *chunks, uint32_t n_chunks,
                            const uint8_t *payload, uint8_t *scratch, uint32_t *acc, const uint32_t *pow24a,
                            const uint32_t *pow16, int with_final, hipStream_t s);
hipError_t tb_launch_tx_segment(const tb_tx_seg_job *jobs, uint32_t n, uint8_t *scratch, const uint32_t *pow24b, hipStream_t s);
hipError_t tb_launch_tx_ratematch(const tb_tx_seg_job *jobs, uint32_t n, const uint8_t *scratch, uint8_t *coded, hipStream_t s);
/* segmentation + CB CRC + encoding + rate matching + interleaving in one kernel (bit-packed encoder); lds_bytes =
 * the encoder's LDS (ldpc_enc_launch_shape) + TB_TX_FUSED_EXTRA_LDS */
#define TB_TX_SEL_SYMS 2048 /* modulation symbols per selection chunk: Qm sub-streams of that many bits are staged in LDS */
#define TB_TX_FUSED_EXTRA_LDS (8 + 1056 + 16 + 8 * (TB_TX_SEL_SYMS / 32 + 1) * 4)
struct ldpc_enc_job;
hipError_t tb_launch_tx_fused(const tb_tx_seg_job *jobs, const struct ldpc_enc_job *ejobs, uint32_t n, int n_threads, int lds_bytes,
                              const uint8_t *scratch, uint8_t *coded, const uint32_t *pow24b, uint32_t *acc, hipStream_t s);
hipError_t tb_launch_rx_dematch(const tb_rx_seg_job *jobs, uint32_t n, uint32_t lds_elems, const int16_t *llr, int16_t *harq,
                                int8_t *scratch, hipStream_t s, int wide = 0);
/* Fused segment kernel (tb_rx_fused.hip): one workgroup takes a code segment from the received LLRs to its payload bytes --
 * de-matching (tb_rx_core.h) as the prologue of the decoder's block body, and instead of an output row the segment's bytes
 * of the payload, its share of the TB CRC and, from the last segment of a transport block to finish, the block's verdict.
 * Jobs = ldpc_dec_job with seg_idx >= 0 (a job with seg_idx < 0 is decoded as by ldpc_launch_dec_fast_jobs).  done[] and
 * the abort flags are zero on entry and on exit. */
struct tb_rx_fused_args {
  const tb_rx_seg_job *segs;
  const tb_rx_tb_job *tbs;
  const int16_t *llr;  /* device memory, or the device address of page-locked host memory (pulled over the link) */
  int16_t *harq;
  uint8_t *payload;
  uint8_t *ack;
  int32_t *iter_max;
  unsigned long long *slots; /* per segment: {CRC share, pass count | generation << 16}: written by the segment, read by
                                the last segment of its transport block to finish */
  uint32_t *gen;       /* per TB: generation, + 1 per call (never reset: the slots of earlier calls stay distinguishable) */
  int *done;           /* per TB: segments finished (zero on entry and on exit) */
  const uint32_t *pow24a;
  /* First-round stagger.  The workgroups that share a CU start together and would stay in step -- all of them in their
   * memory-bound prologue (the CU's VALUs idle, HBM contended by every CU at once), then all of them decoding.  Workgroup
   * b < stagger_cus * stagger_slots waits (b / stagger_cus) * stagger_ticks (10 ns each) before it starts, so that one
   * workgroup's prologue runs under its neighbour's decoding; a workgroup that finishes is replaced at once, so the later
   * rounds inherit the offset.  0 ticks: off. */
  uint32_t stagger_ticks, stagger_cus, stagger_slots;
  /* != 0: the int8 decoder input of a segment never leaves the CU -- it lives this many bytes into the workgroup's LDS (behind
   * the decoder's own arrays: the launch's LDS size includes it) instead of in the segment's scratch row; every job of the
   * launch is a fused one then.  The plan does this when the row fits without costing the CU a workgroup. */
  uint32_t lrow_off;
  /* != 0: some job of the launch carries LDPC_JOB_MUTE_CHECK (retransmissions): the instantiation that looks for mute items */
  uint32_t mute;
  /* != 0: every job of the launch has this lifting size, one block per workgroup (ldpc_dec_fast_block.h ZC) */
  uint32_t zc;
  /* bit 0: the prologue runs at raised issue priority, bit 1: the epilogue too (NRLDPC_HIP_TB_PRIO, default 1: level with 3) */
  uint32_t prio_pro;
  /* diagnostics (NRLDPC_HIP_TB_TRACE=<file>): per workgroup {HW_ID, XCC_ID, wall clock at start, after the prologue, after
   * the last pass, at the end, pass count, after: the LDS image is cleared, the LLRs are scattered, the soft buffer is streamed,
   * the decoder input is visible to the workgroup, 0...} as 16 x uint64; NULL normally */
  unsigned long long *trace;
};
struct ldpc_dec_args;
hipError_t tb_launch_rx_fused(const struct ldpc_dec_args &a, const tb_rx_fused_args &x, int n_threads, int lds_bytes, uint32_t n_jobs,
                              hipStream_t s);
hipError_t tb_rx_fused_init(void);
/* reassembly per segment (payload copy + partial TB CRC into acc[tb], zero on entry and on exit), then per-TB verdict */
hipError_t tb_launch_rx_assemble(const tb_rx_tb_job *jobs, uint32_t n_tb, const tb_rx_seg_job *segs, uint32_t n_seg,
                                 const int32_t *n_iter, uint8_t *scratch, uint8_t *payload, uint8_t *ack, int32_t *iter_max,
                                 uint32_t *acc, int *tb_abort, const uint32_t *pow24a, const uint32_t *pow16, hipStream_t s);
#endif
