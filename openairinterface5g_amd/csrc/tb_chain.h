/*
 * tb_chain.h -- device jobs of the transport-block chain around the LDPC codec (SURVEY section 8 rows a16-a20,
 * f2, f3): TB CRC attach, code-block segmentation + CB CRC, rate matching + bit interleaving (TX);
 * de-interleaving + rate de-matching with HARQ soft combining + int16->int8 pack, TB reassembly + TB CRC (RX).
 */
#ifndef TB_CHAIN_H
#define TB_CHAIN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TB_CRC24A_POW_LEN (1u << 21) /* x^j mod g for j < 2 Mi: covers any NR transport block */

struct tb_tx_tb_job {      /* one per transport block */
  uint64_t payload_off;    /* A/8 bytes in the payload buffer */
  uint64_t b_off;          /* scratch: payload || TB CRC, B/8 bytes */
  uint32_t A, B, crc_type; /* CRC24_A (0) or CRC16 (2) */
  uint32_t pad;
};
struct tb_crc_chunk_job {  /* one per chunk of a transport block: the TB CRC is computed by many workgroups */
  uint32_t tb;             /* index into the per-TB job array */
  uint32_t first_byte;     /* byte range [first_byte, first_byte + chunk) of the TB; chunk = TB_CRC_CHUNK_SMALL when bit 31 is set */
};
/* bytes of a transport block per workgroup of the TB CRC kernel: 8 or 32 per thread.  The byte-table recurrence over a
 * thread's bytes is a chain of dependent look-ups (short pieces = short latency: one transport block 30.6 -> 26.9 us),
 * but every piece costs ~200 instructions and a power-table load to move to the end of the string (long pieces = less
 * work: a 64-block slot's CRC kernels take twice as long with the short ones) -- the plan picks by the call's size. */
#define TB_CRC_CHUNK_SMALL 2048u
#define TB_CRC_CHUNK 8192u
struct tb_tx_seg_job {     /* one per code block */
  uint64_t b_off;          /* the TB's b */
  uint64_t c_off;          /* scratch: packed segment, K/8 bytes (encoder input) */
  uint64_t d_off;          /* scratch: encoder output, one bit per byte */
  uint64_t out_off;        /* coded output: TB offset + sum of the previous segments' E */
  uint32_t r, C, Kprime, L, K; /* segment index, segments, bits incl. CB CRC, CB CRC length, K */
  uint32_t E, Qm, Foffset, Fin, V, rank0;
  uint32_t tb;             /* transport block (index of its CRC accumulator) */
  /* fused kernel, segment that carries the TB CRC (the last one): crc_pos = byte of the segment where the CRC starts,
   * crc_len = 3 (CRC24A) / 2 (CRC16); crc_len = 0: no TB CRC bytes in this segment */
  uint32_t crc_pos, crc_len;
  uint32_t pad;
};
struct tb_rx_seg_job {
  uint64_t llr_off;        /* int16 units: TB offset + sum of the previous segments' E */
  uint64_t harq_off;       /* int16 units: soft buffer d[r] of this segment */
  uint64_t l_off;          /* scratch: decoder input, int8 */
  uint32_t E, Qm, Ncb, Foffset, Fin, V, rank0, clear;
  uint32_t K, F, Z, num_llr; /* num_llr = ncols(R)*Z bytes the decoder reads */
  /* reassembly (tb_rx_assemble_kernel): */
  uint64_t c_off;          /* scratch: this segment's decoded bits */
  uint32_t tb, r;          /* transport block (index into the per-TB jobs) and segment number */
  uint32_t iter_idx, pad;  /* where the decoder reported this segment's pass count */
};
struct tb_rx_tb_job {
  uint64_t payload_off;    /* A/8 bytes out */
  uint64_t b_off;          /* scratch: reassembled b (B/8 bytes) */
  uint64_t c_off0;         /* scratch: first segment's decoded bits; segments are c_stride apart */
  uint32_t c_stride;
  uint32_t seg0, C;        /* index of the first segment in the n_iter array */
  uint32_t A, B, crc_type, num_max_iter;
  uint32_t seg_bytes;      /* payload bytes carried per segment = K/8 - F/8 - (C > 1 ? 3 : 0) */
  uint32_t pad;
};

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
/* lds_elems = the largest tb_rx_lds_elems() over the jobs (int16 slots of LDS a workgroup needs) */
__host__ __device__ static inline uint32_t tb_rx_lds_elems(uint32_t E, uint32_t Fin, uint32_t Ncb)
{
  const uint32_t span = E + Fin;
  return ((span < Ncb ? span : Ncb) + 8u + 7u) & ~7u; /* + 8: the span starts up to 7 slots into its first aligned word */
}
hipError_t tb_launch_rx_dematch(const tb_rx_seg_job *jobs, uint32_t n, uint32_t lds_elems, const int16_t *llr, int16_t *harq,
                                int8_t *scratch, hipStream_t s, int wide = 0);
/* reassembly per segment (payload copy + partial TB CRC into acc[tb], zero on entry and on exit), then per-TB verdict */
hipError_t tb_launch_rx_assemble(const tb_rx_tb_job *jobs, uint32_t n_tb, const tb_rx_seg_job *segs, uint32_t n_seg,
                                 const int32_t *n_iter, uint8_t *scratch, uint8_t *payload, uint8_t *ack, int32_t *iter_max,
                                 uint32_t *acc, int *tb_abort, const uint32_t *pow24a, const uint32_t *pow16, hipStream_t s);
#endif
