/*
 * ldpc_server.h -- mailbox protocol between the per-segment entry points (LDPCdecoder / LDPCencoder, the plugin ABI
 * the reference's thread-pool workers call: openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:219,435-468,
 * nr_dlsch_coding.c:171,386-403) and the resident server kernel (ldpc_server.hip).
 *
 * Why: a call through the HIP runtime (copy in, launch, copy out, synchronise) costs ~90 us for a ~25 us decode.  The
 * server is ONE kernel launch that stays resident while calls keep coming: one workgroup per caller slot, each polling
 * its slot's doorbell in page-locked host memory.  A call is then: memcpy the LLRs into the slot, ring the doorbell,
 * spin on the completion word -- no runtime call, no lock shared between callers.
 *
 * Where the request lives: with a large PCIe BAR (every Instinct part) the request line and the payload area are DEVICE
 * memory that the host writes directly (write-combining posted writes, ~24 GB/s measured: profiles/r02/bar_write.txt):
 * the workgroup then polls and reads local memory -- no read ever crosses the link, which matters because one CU can
 * keep only ~64 reads in flight (26 KB of LLRs pulled from host memory by one workgroup took 8-10 us).  Without a large
 * BAR both sit in page-locked host memory and the workgroup pulls.
 *
 *   host -> GPU   request line (64 bytes) = the request header in four 16-byte chunks, each led by a tag word holding
 *                 the call's sequence number; the host writes chunk 3, 2, 1, then chunk 0, fields before tag, so that
 *                 whichever way the link splits the GPU's 64-byte read, a chunk whose tag shows the new number is
 *                 complete -- the poll that sees four equal new tags HAS the header (no second trip for it);
 *                 slot input area: the payload (LLRs / segment bytes), read by the decoder straight from host memory
 *   GPU -> host   fast decoder: the slot's output area in 16-byte units {three output dwords, sequence number} and the
 *                 completion line {sequence number, n_iter, two timing words} as one 16-byte store: every unit proves
 *                 its own arrival, nothing is fenced or ordered.  Other kinds: plain output bytes, n_iter and stamps,
 *                 then -- behind a system-scope fence of every thread and a barrier -- done = sequence number served
 *
 * Lifetime: the kernel exits by itself when no slot has seen a request for `idle_ticks` (so a process that stops
 * calling -- or calls hipDeviceSynchronize -- is never stuck behind it) or when the host raises *host_stop.  The
 * decision is taken by workgroup 0 for all workgroups (device flag `stopping`) and published in *state; a caller that
 * finds the server stopped relaunches it on the same stream, which starts only after the old generation has left,
 * so a slot is never served by two workgroups.  A request rung but not served by the old generation is found by the
 * new one (doorbell != done at start-up).
 */
#ifndef LDPC_SERVER_H
#define LDPC_SERVER_H
#include <stdint.h>

#define SRV_MAX_SLOTS 128
#define SRV_IN_STRIDE (28u * 1024u)                /* up to 68*384 LLRs, or 8 segments of 1056 B */
#define SRV_OUT_STRIDE (200u * 1024u)              /* 8 segments x 66*384 coded bytes; decoder: <= 68*384 bytes x 4/3 (tagged units) */
#define SRV_LDS_BYTES (160 * 1024)
#define SRV_BC_OFF (SRV_LDS_BYTES - 256)           /* broadcast area at the end of the workgroup's LDS */
#define SRV_CODE_LDS_MAX SRV_BC_OFF                /* a code is servable when its kernel's LDS fits below */

enum { SRV_KIND_DEC_FAST = 1, SRV_KIND_DEC_GENERIC = 2, SRV_KIND_ENC = 3 };

typedef struct srv_req {   /* ctl line 0, host-written: four chunks of {tag, three words} */
  uint32_t tag0;           /* = sequence number of the call (never 0, never 0xffffffff); written LAST */
  uint32_t kind_mode;      /* kind | out_mode << 8 | use_crc << 16 | crc_type << 24 */
  uint32_t max_pass;       /* decoder: numMaxIter + 1 */
  uint32_t crcE;
  uint32_t tag1;
  uint32_t code_lo, code_hi; /* device address of the ldpc_code_desc_t */
  uint32_t kb_nseg;        /* encoder: information columns | segments in this call (<= 8) << 16 */
  uint32_t tag2;
  uint32_t seg_in_stride, seg_out_stride; /* encoder: bytes between segments in the payload / output area */
  uint32_t payload_bytes;
  uint32_t tag3;
  uint32_t pad[3];
} srv_req;

typedef struct srv_slot_ctl { /* 64 bytes, GPU-written, in page-locked host memory: */
  uint32_t done;           /* the sequence number whose results are complete: written last, system-scope release */
  int32_t n_iter;
  uint32_t t_stage_decode; /* diagnostics, 10 ns ticks: request seen -> decoder entered (low 16 bits), -> block function
                              returned (high 16 bits) */
  uint32_t t_pro_passes;   /* fast decoder only: entered -> state in LDS (low), -> last pass finished (high) */
  uint32_t pad1[12];
} srv_slot_ctl;

typedef struct srv_gctl {  /* device memory, shared by the workgroups of a generation */
  uint32_t stopping_gen;   /* generation that has been told to leave (0: none) */
  uint32_t pad;
  long long last_activity; /* wall_clock64() of the latest request seen by any workgroup */
} srv_gctl;

typedef struct srv_args {
  srv_slot_ctl *ctl;       /* host, [n_slots]: completion lines */
  const srv_req *req;      /* [n_slots] request lines: device memory the host writes over the PCIe BAR (fine-grained, so that
                              the GPU does not keep stale copies in L2), or page-locked host memory without a large BAR */
  const uint8_t *in_host;  /* n_slots x SRV_IN_STRIDE payload areas, same kind of memory as req */
  uint8_t *out_host;       /* host, n_slots x SRV_OUT_STRIDE */
  uint8_t *staging;        /* device, n_slots x SRV_IN_STRIDE: the core columns' LLRs, which the fast decoder re-reads every pass */
  srv_gctl *gctl;          /* device */
  uint32_t *state;         /* host: 2*gen+1 running, 2*gen+2 stopped */
  const uint32_t *host_stop; /* host: == gen asks generation gen to stop */
  uint32_t gen;
  uint32_t idle_ticks;     /* wall_clock64 ticks (100 MHz) */
  const uint32_t *crc_pow_tbl[4];
  const uint32_t *abort_w; /* per slot (16 words apart), same kind of memory as req: == the request's sequence number when its
                              caller has given the transport block up (decode_abort_t raised by another thread) */
} srv_args;

#endif
