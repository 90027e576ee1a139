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
 *   host -> GPU   slot ctl line 0: doorbell = (sequence << 12) | number of 16-byte units of [header | payload]
 *                 slot input area: srv_req header (64 B) followed by the payload (LLRs / segment bytes)
 *   GPU -> host   slot output area (bits / coded bytes), then ctl line 1: n_iter, done = doorbell value served
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
#define SRV_REQ_BYTES 64u
#define SRV_IN_STRIDE (28u * 1024u)                /* header + up to 68*384 LLRs, or 8 segments of 1056 B */
#define SRV_OUT_STRIDE (200u * 1024u)              /* 8 segments x 66*384 coded bytes; decoder: <= 68*384 */
#define SRV_LDS_BYTES (160 * 1024)
#define SRV_BC_OFF (SRV_LDS_BYTES - 256)           /* broadcast area at the end of the workgroup's LDS */
#define SRV_CODE_LDS_MAX SRV_BC_OFF                /* a code is servable when its kernel's LDS fits below */

enum { SRV_KIND_DEC_FAST = 1, SRV_KIND_DEC_GENERIC = 2, SRV_KIND_ENC = 3 };

typedef struct srv_req {   /* 64 bytes, first in the slot's input area */
  uint32_t kind;
  uint32_t max_pass;       /* decoder: numMaxIter + 1 */
  uint32_t use_crc, crcE, crc_type, out_mode;
  uint32_t Kb, n_seg;      /* encoder: information columns, segments in this call (<= 8) */
  uint64_t code;           /* device address of the ldpc_code_desc_t */
  uint32_t seg_in_stride, seg_out_stride; /* encoder: bytes between segments in the payload / output area */
  uint32_t pad[4];
} srv_req;

typedef struct srv_slot_ctl { /* 128 bytes: one cache line per direction */
  uint32_t doorbell;       /* host-written */
  uint32_t pad0[15];
  uint32_t done;           /* GPU-written: the doorbell value whose results are complete */
  int32_t n_iter;
  uint32_t t_seen, t_staged, t_decoded; /* wall_clock64 (100 MHz) stamps of the request just served: doorbell seen,
                                           payload staged, block function returned (diagnostics, nrLDPC_hip_server_stats) */
  uint32_t t_prologue, t_passes; /* fast decoder only: state in LDS (first barrier passed), last pass finished */
  uint32_t pad1[9];
} srv_slot_ctl;

typedef struct srv_gctl {  /* device memory, shared by the workgroups of a generation */
  uint32_t stopping_gen;   /* generation that has been told to leave (0: none) */
  uint32_t pad;
  long long last_activity; /* wall_clock64() of the latest request seen by any workgroup */
} srv_gctl;

typedef struct srv_args {
  srv_slot_ctl *ctl;       /* host, [n_slots] */
  const uint8_t *in_host;  /* host, n_slots x SRV_IN_STRIDE */
  uint8_t *out_host;       /* host, n_slots x SRV_OUT_STRIDE */
  uint8_t *staging;        /* device, n_slots x SRV_IN_STRIDE */
  srv_gctl *gctl;          /* device */
  uint32_t *state;         /* host: 2*gen+1 running, 2*gen+2 stopped */
  const uint32_t *host_stop; /* host: == gen asks generation gen to stop */
  uint32_t gen;
  uint32_t idle_ticks;     /* wall_clock64 ticks (100 MHz) */
  const uint32_t *crc_pow_tbl[4];
} srv_args;

#endif
