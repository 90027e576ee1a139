/*
 * tb_jobs.h -- job records of the transport-block chain kernels (tb_chain.h); plain C++ (no HIP headers) so that the CPU
 * emulation of the kernels' per-thread code (tests/emul) can include them.
 */
#ifndef TB_JOBS_H
#define TB_JOBS_H
#include <stdint.h>
#if defined(__HIPCC__)
#define TB_HD __host__ __device__ static inline
#else
#define TB_HD static inline
#endif

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
  /* ... whose value that segment's workgroup puts together itself: the XOR of the partial registers the CRC kernel's
   * workgroups crc_chunk0 .. crc_chunk0 + crc_nchunks - 1 left (one plain store each: no atomics, nothing to reset) */
  uint32_t crc_chunk0, crc_nchunks;
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
  uint32_t fused;          /* 1: every segment of the block runs through the fused segment kernel, which also delivers the payload,
                              the TB CRC and the verdict (tb_rx_fused.hip); the reassembly kernels skip the block */
};

/* lds_elems = the largest tb_rx_lds_elems() over the jobs (int16 slots of LDS a workgroup needs) */
TB_HD uint32_t tb_rx_lds_elems(uint32_t E, uint32_t Fin, uint32_t Ncb)
{
  const uint32_t span = E + Fin;
  return ((span < Ncb ? span : Ncb) + 8u + 7u) & ~7u; /* + 8: the span starts up to 7 slots into its first aligned word */
}
#endif
