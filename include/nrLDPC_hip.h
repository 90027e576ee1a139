/*
 * nrLDPC_hip.h -- C ABI of libldpc_hip.so, the MI355X (gfx950) drop-in for OAI's NR LDPC coding library.
 *
 * Part 1 is the reference's own plugin ABI: the four symbols load_LDPClib() resolves from
 * "libldpc<version>.so" (reference openair1/PHY/CODING/nrLDPC_load.c:45-75, nrLDPC_extern.h:27-45) with
 * the reference's parameter structures restated field for field, so that `ldpctest -v _hip`,
 * `nr_ulsim --loader.ldpc.shlibversion _hip` or the gNB softmodem can load this library unchanged.
 * A translation unit that already includes the reference's nrLDPC_defs.h must define
 * NRLDPC_HIP_NO_REFERENCE_TYPES before including this header.
 *
 * Part 2 adds batched entry points (many code blocks per call, device- or host-resident buffers, caller
 * stream) -- the form in which a GPU is actually fed -- plus the TB-level helpers around the codec.
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions are thread safe.
 */
#ifndef NRLDPC_HIP_H
#define NRLDPC_HIP_H
#include <stdint.h>
#include <stdbool.h>
#include <pthread.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================================================
 * Part 1 -- reference plugin ABI
 * =================================================================================================== */
#ifndef NRLDPC_HIP_NO_REFERENCE_TYPES

/* nrLDPC_decoder/nrLDPC_types.h:75-79 */
typedef enum nrLDPC_outMode {
  nrLDPC_outMode_BIT,     /* 32 bits per uint32_t output, MSB-first inside each byte */
  nrLDPC_outMode_BITINT8, /* 1 bit per int8_t output */
  nrLDPC_outMode_LLRINT8  /* reference quirk: delivers the BITINT8 result at this revision (decoder.c:866-877) */
} e_nrLDPC_outMode;

/* nrLDPC_decoder/nrLDPC_types.h:84-97 */
typedef struct nrLDPC_dec_params {
  uint8_t BG;         /* base graph 1 / 2 */
  uint16_t Z;         /* lifting size */
  uint8_t R;          /* decoder rate mode: 13, 23, 89 (BG1) / 15, 13, 23 (BG2) */
  uint16_t F;         /* filler bits (offload back ends only) */
  uint8_t Qm;         /* modulation (offload back ends only) */
  uint8_t rv;         /* (offload back ends only) */
  uint8_t numMaxIter; /* iteration cap; up to numMaxIter+1 CN/BN passes run (decoder.c:552-558) */
  int E;              /* number of leading output bits covered by the CRC check */
  e_nrLDPC_outMode outMode;
  int crc_type;       /* CRC24_A 0, CRC24_B 1, CRC16 2, CRC8 3 (coding_defs.h:33-36) */
  /* NULL: stop on parity check.  Non-NULL: stop on CRC from pass 3 on, the predicate applied to (p_out, E, crc_type) after
   * every such pass exactly as nrLDPC_decoder.c:849-861 applies it.
   *   - The library's own nrLDPC_hip_check_crc, or the host executable's `check_crc` (looked up once with
   *     dlsym(RTLD_DEFAULT) -- what nr_ulsch_decoding.c:216 / nr_dlsch_decoding.c:253 pass): the same CRC
   *     (crc_byte.c:314-380) is evaluated ON THE GPU from crc_type / E; the pointer is not called.  Needs outMode BIT,
   *     E % 8 == 0, 0 < E <= K; any other combination takes the next path.
   *   - Any other pointer (LDPCdecoder and host-memory LDPCdecoder_batch): it IS called on the host, on p_out, after every
   *     pass >= 3 in order until it returns non-zero -- the decoder runs all its passes without stopping and keeps every
   *     pass' output, so this path costs the full iteration count (slow, exact).
   *   - Device-memory batches, LDPCdecoder_jobs and the transport-block chain evaluate the CRC on the GPU whatever the
   *     pointer (there it is a mode flag) and refuse the combinations the GPU cannot serve. */
  int (*check_crc)(uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type);
  uint8_t setCombIn;
} t_nrLDPC_dec_params;

/* common/utils/time_meas.h:61-74 (oai_cputime_t = long long on x86-64, :39).  The encoder's four optional meters
 * (tinput, tprep, tparity, toutput) and the decoder's `total` are filled the way start_meas()/stop_meas() (:148-176) fill
 * them, gated by the host executable's `opp_enabled` when it exports one. */
typedef struct time_stats {
  long long in;        /* time stamp of the running measurement */
  long long diff;      /* accumulated ticks */
  long long p_time;    /* last duration */
  double diff_square;
  long long max;
  int trials;
  int meas_flag;
  char *meas_name;
  int meas_index;
  int meas_enabled;
  void *tpoolmsg;
  void *tstatptr;
} time_stats_t;
/* nrLDPC_decoder/nrLDPC_types.h:115-127.  Only `total` is written (the reference's per-function split -- cnProc,
 * bnProc, the buffer copies -- has no counterpart in a kernel that keeps a block in LDS from LLR load to bit store). */
typedef struct nrLDPC_time_stats {
  time_stats_t llr2llrProcBuf, llr2CnProcBuf, cnProc, cnProcPc, bnProcPc, bnProc, cn2bnProcBuf, bn2cnProcBuf, llrRes2llrOut,
      llr2bit, total;
} t_nrLDPC_time_stats;

/* openair1/PHY/defs_common.h:998-1027 -- transport-block wide "stop decoding" flag shared by the segments */
typedef struct {
  pthread_mutex_t mutex_failure;
  bool failed;
} decode_abort_t;

/* nrLDPC_defs.h:40-66 */
typedef struct {
  unsigned int n_segments; /* number of segments in input[]/output[] */
  unsigned int macro_num;  /* this call encodes segments 8*macro_num .. min(n_segments, 8*macro_num+8) */
  unsigned char gen_code;
  time_stats_t *tinput;
  time_stats_t *tprep;
  time_stats_t *tparity;
  time_stats_t *toutput;
  int Kr;
  uint32_t Kb;  /* information columns entering the parity (22; 10/9/8/6 for BG2) */
  uint32_t Zc;
  void *harq;
  uint8_t BG;
  unsigned char *output;
  uint32_t K;   /* bits per segment incl. fillers = 22*Zc / 10*Zc */
  uint32_t F;
  uint8_t Qm;
  uint32_t E;
  unsigned int G;
  uint8_t rv;
} encoder_implemparams_t;

#endif /* NRLDPC_HIP_NO_REFERENCE_TYPES */

/* nrLDPC_defs.h:68-87.  LDPCinit: 0 on success (the loader asserts on anything else, nrLDPC_load.c:67);
 * here it selects the GPU (env NRLDPC_HIP_DEVICE, default 0) and fails (-1) when no gfx950-capable HIP
 * device is usable -- there is no CPU fallback. */
int32_t LDPCinit(void);
int32_t LDPCshutdown(void);
/* Optional loader hook (common/utils/load_module_shlib.c:174-185, checkverfunc_t): called by load_module_version_shlib()
 * right after dlopen() with the executable's build string; reports this library's and returns 0.  With
 * NRLDPC_HIP_REQUIRE_BUILD=<text> in the environment it returns -1 -- the loader then refuses the library -- unless the
 * executable's build string contains <text>. */
int32_t ldpc_checkbuildver(char *mainexec_buildversion, char **shlib_buildversion);
int32_t nrLDPC_hip_checkbuildver(char *mainexec_buildversion, char **shlib_buildversion); /* (what the offload-slot library forwards to) */
/* Optional loader hook (common/utils/load_module_shlib.c:186-191, initfunc_t): called by load_module_version_shlib() after the
 * version hook with the `initfunc_arg` of its caller -- NULL from load_LDPClib (nrLDPC_load.c:61).  NULL: nothing to do,
 * returns 0 (the GPU is taken by LDPCinit, which the loader's caller runs next: nrLDPC_load.c:66-67).  A maintainer who passes
 * an argument passes a C string with the GPU list in NRLDPC_HIP_DEVICES' format ("0", "2,3"): it is put into the environment
 * for LDPCinit unless the variable is already set.  The loader ignores the return value. */
int32_t ldpc_autoinit(void *arg);
/* One code block, synchronous, host buffers.  p_llr: int8[ncols(BG,R)*Z] in base-graph column order, the two
 * punctured columns 0 and fillers +127 (callers: nr_ulsch_decoding.c:195-219, ldpctest.c:294-332).
 * Returns the number of passes executed; > numMaxIter means "not decoded" and sets *ab (decoder.c:190-193);
 * numMaxIter+2 when *ab was already set on entry or is raised by another thread while the call is running (looked at
 * once per pass, decoder.c:556-559; p_out is then left as it was).  Never negative, like the reference: an internal error (bad
 * parameters, HIP failure) is reported as numMaxIter+1 with *ab set, so that callers which only test
 * `<= numMaxIter` (nr_ulsch_decoding.c:219-222) NACK; nrLDPC_hip_last_error() tells why.
 * Calls are served by a resident GPU kernel through per-thread mailboxes (no HIP runtime call per segment); see
 * csrc/ldpc_server.h for NRLDPC_HIP_SERVER / _SRV_SLOTS / _SRV_IDLE_US.
 * With check_crc set and whole columns of zeros at the end of p_llr (a high-rate first transmission), the call is decoded
 * without the rows that close on those columns: they send zeros in every pass (nrLDPC_cnProc.h:105-114), so p_out and the
 * return value are those of the whole rate mode (NRLDPC_HIP_CUT=0: off). */
int32_t LDPCdecoder(t_nrLDPC_dec_params *p_decParams, uint8_t harq_pid, uint8_t ulsch_id, uint8_t C, int8_t *p_llr,
                    int8_t *p_out, t_nrLDPC_time_stats *p_profiler, decode_abort_t *ab);
/* Up to 8 segments per call (ldpc_encoder_optim8segmulti.c:46-213): input[j] K/8 bytes MSB first,
 * output[j] one bit per byte, (BG1 ? 66 : 50)*Zc bytes = c[2Zc..K) || parity.  Return value: that of the DEFAULT reference
 * library, ldpc_encoder_optim8segmulti.c:213 -- 0, and -1 on bad parameters (:88-100 there) -- not the output length that
 * ldpc_encoder.c:251 (`libldpc_orig.so`) returns; the reference's callers ignore the value (nr_dlsch_coding.c:171,
 * nr_ulsch_coding.c:167, ldpctest.c:265-282), tests/test_gpu_encoder.py asserts the 0.  Served by a resident kernel of its own, like the decoder
 * (NRLDPC_HIP_ENC_SERVER=0: one launch per call). */
int32_t LDPCencoder(uint8_t **input, uint8_t **output, encoder_implemparams_t *impp);

/* ===================================================================================================
 * Part 2 -- batched / device-resident entry points
 * =================================================================================================== */
#define NRLDPC_HIP_MEM_HOST 0   /* pointers are host memory: staged through pinned buffers, call is synchronous */
#define NRLDPC_HIP_MEM_DEVICE 1 /* pointers are device memory on the library's GPU: call only enqueues on `stream` */
/* nrLDPC_hip_ulsch_decode only, OR-ed into `mem`: where the HARQ soft buffers d[r] live, independently of the other buffers.
 * The reference keeps them per HARQ process for the life of the process (NR_TRANSPORT/nr_ulsch_decoding.c:168,
 * harq_process->d[r]) while the LLRs of a slot arrive in host memory (nr_ulsch_decoding(..., short *ulsch_llr, ...), :320):
 * with NRLDPC_HIP_MEM_HOST | NRLDPC_HIP_MEM_HARQ_DEVICE or ..._HARQ_LIBRARY a call moves the slot's LLRs over the link once
 * and nothing else -- the soft buffers never leave the GPU. */
#define NRLDPC_HIP_MEM_HARQ_DEVICE 2  /* b->harq is device memory (hipMalloc, or hipMallocManaged) on the library's GPU whatever bit 0 says */
#define NRLDPC_HIP_MEM_HARQ_LIBRARY 4 /* b->harq is ignored: the library keeps the soft buffers in GPU memory of its own, one
                                       * set of C x harq_stride int16 per transport block, found by tb[i].harq_off used as an
                                       * opaque 64-bit id chosen by the caller (e.g. ulsch_id << 8 | harq_pid -- the way the T2
                                       * card is addressed, nrLDPC_decoder_offload.c:546-547).  Allocated on first use, kept until
                                       * nrLDPC_hip_harq_release(); with several GPUs the buffers live on the GPU that decodes
                                       * the block (and follow it if a later round is given to another one). */

typedef struct nrLDPC_hip_dec_batch {
  t_nrLDPC_dec_params params; /* shared by every block of the batch (homogeneous batch = one launch) */
  uint32_t n_blocks;
  const int8_t *llr;          /* block b at llr + b*llr_stride, ncols*Z int8 each */
  uint32_t llr_stride;        /* bytes, >= ncols*Z; multiples of 16 give the widest loads */
  int8_t *out;                /* block b at out + b*out_stride; BIT: 4*ceil(ncols*Z/32) bytes, else ncols*Z bytes */
  uint32_t out_stride;        /* bytes, multiple of 4 */
  int32_t *n_iter;            /* [n_blocks] return value of each block (same meaning as LDPCdecoder's) */
  int32_t mem;                /* NRLDPC_HIP_MEM_* for llr/out/n_iter alike */
  void *stream;               /* hipStream_t for DEVICE mem; NULL = HIP's default (null) stream */
  int32_t kernel;             /* 0 = best available for (BG,Z,R); 1 = generic kernel (any code); 2 = fast kernel or error;
                               * 3 / 4 = fast kernel with the throughput / latency workgroup shape forced (0 and 2 pick the
                               * shape from n_blocks: latency shape up to one workgroup round of the GPU; for Zc <= 64 a launch
                               * that fills the GPU packs several blocks into a workgroup); 5 = that multi-block variant forced */
} nrLDPC_hip_dec_batch_t;
/* 0 on success, negative on bad parameters / HIP error.  DEVICE mem: asynchronous w.r.t. the host. */
int32_t LDPCdecoder_batch(const nrLDPC_hip_dec_batch_t *b);

/* A MIXED batch of code blocks in one call: every block with its own code, iteration cap and buffers -- what a slot's
 * segments look like when several UEs with different allocations are decoded together (BASELINE configs[2]: short BG2 blocks
 * of Zc = 64 and Zc = 208 in one batch).  Device memory only; outMode and the stop mode (check_crc NULL / non-NULL) must be
 * the same for all blocks; block i reports into n_iter[i].  The blocks are sorted by workgroup shape into as few launches as
 * the shapes allow (a launch whose last workgroup round is partly empty takes smaller blocks along); the job list derived
 * from an array that repeats byte for byte is reused.  0, negative on bad parameters / HIP error; enqueue only. */
typedef struct nrLDPC_hip_dec_job {
  t_nrLDPC_dec_params params;
  const int8_t *llr; /* ncols(BG, R) * Z int8, 4-byte aligned for the fast kernel */
  int8_t *out;       /* BIT: 4 * ceil(ncols*Z / 32) bytes (4-byte aligned), else ncols*Z bytes */
} nrLDPC_hip_dec_job_t;
int32_t LDPCdecoder_jobs(const nrLDPC_hip_dec_job_t *jobs, uint32_t n_jobs, int32_t *n_iter, int32_t mem, void *stream);

typedef struct nrLDPC_hip_enc_batch {
  uint8_t BG;
  uint16_t Zc;
  uint8_t Kb;            /* information columns entering the parity (see encoder_implemparams_t.Kb) */
  uint32_t n_blocks;
  const uint8_t *in;     /* block b at in + b*in_stride: K/8 bytes, MSB first (K = 22*Zc / 10*Zc) */
  uint32_t in_stride;
  uint8_t *out;          /* block b at out + b*out_stride: (66|50)*Zc bytes, one bit per byte */
  uint32_t out_stride;
  int32_t mem;
  void *stream;
} nrLDPC_hip_enc_batch_t;
int32_t LDPCencoder_batch(const nrLDPC_hip_enc_batch_t *b);

/* ---------------------------------------------------------------------------------------------------
 * Transport-block chain on the GPU (what the reference does on the CPU around the codec):
 *   nrLDPC_hip_dlsch_encode: TB CRC attach -> nr_segmentation (+ CB CRC24B, fillers) -> LDPC encode ->
 *     nr_rate_matching_ldpc -> nr_interleaving_ldpc, the body of nr_dlsch_encoding()/ldpc8blocks()
 *     (openair1/PHY/NR_TRANSPORT/nr_dlsch_coding.c:145-404), for a batch of transport blocks;
 *   nrLDPC_hip_ulsch_decode: nr_deinterleaving_ldpc -> nr_rate_matching_ldpc_rx (HARQ combining) -> int8 pack
 *     -> LDPC decode with CRC early stop -> reassembly + TB CRC, the body of nr_ulsch_decoding()/
 *     nr_processULSegment()/nr_postDecode() (nr_ulsch_decoding.c:122-470, SCHED_NR/phy_procedures_nr_gNB.c:271-300).
 * ------------------------------------------------------------------------------------------------- */
typedef struct nrLDPC_hip_tb {
  uint32_t A;        /* transport block size in bits (multiple of 8) */
  uint32_t G;        /* coded bits of the TB (nr_get_G) */
  uint32_t tbslbrm;  /* Tbslbrm as passed to nr_rate_matching_ldpc (0 = full circular buffer) */
  uint8_t BG;        /* base graph (rel15->maintenance_parms_v3.ldpcBaseGraph) */
  uint8_t Qm;        /* modulation order 2/4/6/8 */
  uint8_t Nl;        /* layers */
  uint8_t rv;        /* redundancy version */
  /* decode only */
  uint8_t numMaxIter;
  uint8_t round;     /* HARQ round; 0 clears the soft buffer first (d_to_be_cleared) */
  int32_t llrLen;    /* in/out: state of nr_get_R_ldpc_decoder across rounds (ulsch_harq->llrLen) */
  /* buffer placement (bytes for payload/coded, int16 elements for llr/harq) */
  uint64_t payload_off; /* A/8 bytes */
  uint64_t coded_off;   /* encode: G bytes, one bit per byte (the reference's `output`); decode: G int16 LLRs */
  uint64_t harq_off;    /* decode: C soft buffers of harq_stride int16 each, kept by the caller across rounds */
} nrLDPC_hip_tb_t;

typedef struct nrLDPC_hip_tb_batch {
  uint32_t n_tb;
  nrLDPC_hip_tb_t *tb;     /* host array [n_tb] (llrLen is updated by the decode call) */
  uint8_t *payload;        /* encode: in, decode: out */
  void *coded;             /* encode: uint8_t* out; decode: const int16_t* in */
  int16_t *harq;           /* decode: soft buffers (device memory when mem = DEVICE or mem & HARQ_DEVICE; unused with HARQ_LIBRARY) */
  uint32_t harq_stride;    /* int16 per code block, >= 66*384 */
  uint8_t *ack;            /* decode out [n_tb]: 1 = every segment decoded and the TB CRC holds */
  int32_t *iter_max;       /* decode out [n_tb]: largest per-segment pass count */
  int32_t mem;             /* NRLDPC_HIP_MEM_HOST or _DEVICE: payload / coded / harq / ack / iter_max alike; decode: optionally
                            * | NRLDPC_HIP_MEM_HARQ_DEVICE or | NRLDPC_HIP_MEM_HARQ_LIBRARY for the soft buffers.  HOST: `coded`
                            * in page-locked memory (nrLDPC_hip_host_alloc / nrLDPC_hip_host_register / hipHostMalloc) is read by
                            * the GPU in place -- the segments' workgroups pull their LLRs over the link while others decode --,
                            * pageable memory goes through a staged copy first */
  void *stream;            /* DEVICE mem: enqueue only (except the small per-call job upload) */
} nrLDPC_hip_tb_batch_t;
int32_t nrLDPC_hip_dlsch_encode(const nrLDPC_hip_tb_batch_t *b);
int32_t nrLDPC_hip_ulsch_decode(const nrLDPC_hip_tb_batch_t *b);
/* Soft buffers kept by the library (NRLDPC_HIP_MEM_HARQ_LIBRARY).  release: forget one transport block's buffers (its HARQ
 * process ended) / all of them; read: copy int16 values [first, first + n) of a block's C x harq_stride soft values to host
 * memory (diagnostics and tests; a device-memory decode call that wrote them must have completed on its stream).  0, or -1
 * (unknown id, range outside the buffers). */
int32_t nrLDPC_hip_harq_release(uint64_t id);
int32_t nrLDPC_hip_harq_release_all(void);
int32_t nrLDPC_hip_harq_read(uint64_t id, int16_t *dst, uint64_t first, uint64_t n);
/* Page-locked host memory for callers that do not link the HIP runtime themselves (a C gNB): LLR arrays placed here are
 * read by the GPU in place.  alloc returns NULL on failure; register / unregister pin an existing allocation (0 / -1). */
void *nrLDPC_hip_host_alloc(uint64_t bytes);
void nrLDPC_hip_host_free(void *p);
int32_t nrLDPC_hip_host_register(void *p, uint64_t bytes);
int32_t nrLDPC_hip_host_unregister(void *p);
/* ---------------------------------------------------------------------------------------------------
 * The reference's OFFLOAD plugin slot (`ldpc_interface_offload`, loaded with the suffix "_t2": nr_init.c:138-139).  Same
 * signatures as LDPCdecoder / LDPCencoder, the semantics of nrLDPC_decoder/nrLDPC_decoder_offload.c:1036-1140: one
 * segment per call, rate (de)matching + (de)interleaving + HARQ combining inside, soft buffers kept on the device per
 * (ulsch_id, segment).  libldpc_hip_t2.so (csrc/ldpc_t2_shim.c) exports them under the plugin names.
 *   decoder: p_llr = E int8 LLRs in transmission order; p_decParams: BG, Z, R, numMaxIter + E, Qm, rv, F, setCombIn
 *     (0: the soft buffer starts afresh, 1: combine); C = segment number r; p_out = ceil(K/8) decoded bytes.  Parity-check
 *     stop.  Returns the passes run (> numMaxIter: not decoded), < 0 on a hard error (nr_ulsch_decoding.c:269).
 *   encoder: input[0] = the segment's K - F bits (CB CRC attached by the caller), output[0] = E rate-matched,
 *     interleaved bits, one per byte; impp: BG, Zc, K, F, Kb, E, Qm, rv.  Returns 0 / -1.
 * ------------------------------------------------------------------------------------------------- */
int32_t nrLDPC_hip_offload_init(void);
int32_t nrLDPC_hip_offload_decoder(t_nrLDPC_dec_params *p_decParams, uint8_t harq_pid, uint8_t ulsch_id, uint8_t C, int8_t *p_llr,
                                   int8_t *p_out, t_nrLDPC_time_stats *p_profiler, decode_abort_t *ab);
int32_t nrLDPC_hip_offload_encoder(uint8_t **input, uint8_t **output, encoder_implemparams_t *impp);
/* helpers with the reference's semantics (nr_segmentation parameter part, nr_get_E, nr_get_R_ldpc_decoder) */
int32_t nrLDPC_hip_segmentation(uint32_t B, uint8_t BG, uint32_t *C, uint32_t *K, uint32_t *Zc, uint32_t *F); /* returns Kb, -1 */
uint32_t nrLDPC_hip_get_E(uint32_t G, uint32_t C, uint32_t Qm, uint32_t Nl, uint32_t r);
int32_t nrLDPC_hip_get_R_ldpc_decoder(int32_t rvidx, int32_t E, int32_t BG, int32_t Z, int32_t *llrLen, int32_t round);
/* Columns of the code graph nrLDPC_hip_ulsch_decode() decodes a segment on; R = what nr_get_R_ldpc_decoder chose (its mode has
 * 68 / 35 / 27 resp. 52 / 32 / 17 columns, nrLDPCdecoder_defs.h:53-57, 80-84).  A first transmission (round 0: the soft buffer is
 * cleared, nr_ulsch_decoding.c:418-422) leaves every position behind the last one it reaches at zero; a check node that closes on
 * an all-zero degree-1 column sends zeros to its other neighbours in every pass (nrLDPC_cnProc.h:105-114: the minimum over the
 * OTHER inputs), and the chain stops on the CRC -- so the rows behind the last column that received anything are not run: same
 * payload, verdict and pass count as on the whole mode.  NRLDPC_HIP_TB_TRUNC=0 turns it off.  -1: invalid parameters. */
int32_t nrLDPC_hip_ulsch_decoder_columns(int32_t BG, uint32_t Zc, uint32_t C, uint32_t F, uint32_t K, uint32_t Tbslbrm, int32_t rv,
                                         uint32_t E, int32_t round, int32_t R);

/* Introspection for tests and benchmarks */
int32_t nrLDPC_hip_num_llr(int BG, int Z, int R);      /* ncols*Z, -1 if invalid */
int32_t nrLDPC_hip_out_bytes(int BG, int Z, int R, int outMode);
int32_t nrLDPC_hip_lds_bytes(int BG, int Z, int R);    /* LDS a decoder workgroup uses for this code */
/* info = {rows, columns, edges of the (BG, R) base graph; 1 if the fast decoder kernel serves the code; its workgroup
 * size; its LDS bytes; check-node and bit-node tasks per pass (fast kernel)}.  0, or -1 for an invalid code. */
int32_t nrLDPC_hip_code_info(int BG, int Z, int R, int32_t info[8]);
/* resident submission path behind LDPCdecoder (csrc/ldpc_server.h): out = {status (-1 not started yet, 0 in use, 1 switched
 * off or unavailable), caller slots, server kernel launches so far, calls served through it, and summed over those calls
 * in ns: GPU doorbell-seen -> payload staged, staged -> decoded, host doorbell -> completion seen, whole host call} */
int32_t nrLDPC_hip_server_stats(int64_t out[8]);
/* HIP events around the stages of the calling thread's nrLDPC_hip_ulsch_decode calls, recorded on the stream the kernels
 * run on (primary device): enable != 0 switches the recording on for the calls that follow; out_us, when not NULL, receives
 * the last recorded call's {de-matching kernel, decoder launches (the fused segment kernel), reassembly + verdict kernels,
 * their sum} in microseconds (waits for that call).  What bench.py's chain_roofline is computed from.  0 / -1. */
int32_t nrLDPC_hip_chain_timing(int32_t enable, float out_us[4]);
/* check_crc() of openair1/PHY/CODING/crc_byte.c:314-380 (same arguments, same result): the predicate to put into
 * t_nrLDPC_dec_params::check_crc by callers that do not carry OAI's own -- it selects the CRC evaluated on the GPU -- and a
 * correct host implementation for whoever calls it. */
int nrLDPC_hip_check_crc(uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type);
const char *nrLDPC_hip_last_error(void);
const char *nrLDPC_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
