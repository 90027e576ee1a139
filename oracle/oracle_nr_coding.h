/*
 * oracle_nr_coding.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar C restatement of OAI's NR LDPC coding hot path (reference revision ~v2.1.0), used as the
 * checker for the HIP library in this repository.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, link or call anything in oracle/.  The product
 * (libldpc_hip.so, openairinterface5g_amd/) never does.
 *
 * PARITY PIN STATUS -- read this before trusting it:
 *   Pinned by reference-COMPILED code (oracle/_ref, recipe oracle/ref_pin/, no SIMDE, no stand-in headers, nothing but the
 *   reference's own sources where they lie; outputs committed as tests/golden/ref_*.npz by tools/make_ref_fixtures.py):
 *     - the encoder: encode_parity_check_part_orig + the generator tables (ldpc_generate_coefficient.c, Gen_shift_value.h)
 *       give this oracle's code words for all 51 x 2 codes, Kb < 10 and bit-sliced 8-segment input included;
 *     - the decoder's code set-up (nrLDPC_init.h: LUT selection, numLLR) and ALL of its data movement (nrLDPC_mPass.h:
 *       circular shifts, CN/BN buffer addressing, the degree-1 columns [D1][D6], output reordering): a decoder made of
 *       those compiled functions + a second, independent restatement of the node arithmetic on the reference's own
 *       buffer layouts (oracle/ref_pin/ref_hybrid_decoder.c) gives this oracle's pass counts and output bytes for every
 *       (BG, Zc, R), every output mode, parity and CRC stop (tests/test_ref_pin.py).
 *     - the node functions' FORMULA AND WIRING on the shipped (generated) code: the reference's own generators for cnProc (AVX-512:
 *       both base graphs; AVX2: BG1), bnProcPc BG1 and bnProc BG2 compile without SIMDE; the headers they write are read statement by statement
 *       (tests/test_ref_pin.py test_generated_*): every check-node output = sign(min_epu8(min over the OTHER inputs of abs_epi8, 127),
 *       sign product) [D2]; every bit-node sum = the N widened messages + the channel LLR, packed with saturation, and NO code
 *       for the 1-check group [D3][F5]; every bit-to-check message = subs_epi8(llrRes, message) [D5]; addresses from the LUTs.
 *   NOT pinnable in this image: the bodies of cnProc / cnProcPc / bnProcPc / bnProc / llr2bit are SIMDE intrinsics
 *   (openair1/PHY/sse_intrin.h:52-60 -> un-vendored SIMDE; the shipped variants are generated at build time by the
 *   reference's CMake).  Their per-element rules [D2][D3][D5][D7][D9] are restated twice (here per (edge, lane), there
 *   per reference buffer) from the same file:line, and agree; CRC, segmentation, rate matching and the RNG
 *   (crc_byte.c, nr_segmentation.c, nr_rate_matching.c, rangen_double.c) all reach SIMDE through tools_defs.h and are
 *   pinned only by catalogue values, tables and the properties listed in DESIGN.md section 2.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root,
 * openair1/PHY/CODING/ unless stated otherwise).
 */
#ifndef ORACLE_NR_CODING_H
#define ORACLE_NR_CODING_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- code graph: nrLDPC_decoder/nrLDPCdecoder_defs.h:38-100, nrLDPC_init.h:58 ------------------ */
typedef struct {
  int BG, Z, R;
  int nrows;          /* base-graph rows processed in this decoder-rate mode */
  int ncols;          /* base-graph columns processed (numLLR = ncols*Z) */
  int ncore;          /* first degree-1 parity column: 26 (BG1) / 14 (BG2) */
  int nedges;         /* edges of the processed rows */
  int row_ptr[47];    /* CSR */
  int col[316];
  int shift[316];     /* already reduced mod Z */
} oracle_graph_t;

int oracle_ldpc_ils(int Z);                                   /* set index iLS of lifting size Z, -1 if invalid */
int oracle_ldpc_graph(int BG, int Z, int R, oracle_graph_t *g); /* 0 ok, -1 bad parameters */

/* ---- decoder: nrLDPC_decoder/nrLDPC_decoder.c:172-880 ----------------------------------------- */
enum { ORACLE_OUT_BIT = 0, ORACLE_OUT_BITINT8 = 1, ORACLE_OUT_LLRINT8 = 2 }; /* nrLDPC_types.h:75-79 */

/* use_crc = 0: parity-check early stop (check_crc == NULL in the reference);
 * use_crc = 1: CRC early stop with oracle_check_crc(p_out, E, crc_type) (decoder.c:849-861).
 * Returns the number of CN/BN passes executed exactly like nrLDPC_decoder_core (decoder.c:880).
 * p_out must hold max(4*ceil(ncols*Z/32), ncols*Z) bytes. It is NOT cleared first (the reference
 * does not clear it either). */
int oracle_ldpc_decode(int BG, int Z, int R, int numMaxIter, int outMode, int use_crc, int E, int crc_type,
                       const int8_t *p_llr, int8_t *p_out);

/* block b decoded by thread b % nthreads (pthreads); PC-stop mode, BIT output; used for the CPU baseline timing */
int oracle_ldpc_decode_mt(int nthreads, int nblocks, int BG, int Z, int R, int numMaxIter, const int8_t *llr,
                          int llr_stride, int8_t *out, int out_stride, int *iters);

/* The same decoder restated for speed (oracle_ldpc_decoder_vec.c: two-minimum check node, lane-contiguous loops that
 * gcc vectorises, AVX-512BW / AVX2 / baseline clones): identical outputs and pass counts, checked against
 * oracle_ldpc_decode for every code in tests/test_oracle.py.  Used where the scalar restatement is too slow: the
 * CPU baseline of bench.py and whole-batch GPU parity tests. */
int oracle_ldpc_decode_vec(int BG, int Z, int R, int numMaxIter, int outMode, int use_crc, int E, int crc_type,
                           const int8_t *p_llr, int8_t *p_out);
int oracle_ldpc_decode_vec_mt(int nthreads, int nblocks, int BG, int Z, int R, int numMaxIter, const int8_t *llr,
                              int llr_stride, int8_t *out, int out_stride, int *iters);

/* ---- encoder: nrLDPC_encoder/ldpc_encoder.c:44-252 (code word), via H instead of the generator lists */
/* in: K/8 bytes MSB-first (K = 22*Zc or 10*Zc); out: one bit per byte, (BG1 ? 66 : 50)*Zc bytes =
 * c[2Zc..K) followed by all parity bits.  Kb = number of information columns that enter the parity
 * (ldpc_encoder.c:232).  Returns the output length, -1 on bad parameters. */
int oracle_ldpc_encode(int BG, int Zc, int Kb, const uint8_t *in, uint8_t *out);
/* H * x == 0 check on a full code word x[ncols*Z] (incl. the two punctured columns), 1 bit/byte. */
int oracle_ldpc_syndrome_weight(int BG, int Z, const uint8_t *x);

/* ---- CRC: crc_byte.c:46-58,148-260,314-380 ---------------------------------------------------- */
enum { ORACLE_CRC24_A = 0, ORACLE_CRC24_B = 1, ORACLE_CRC16 = 2, ORACLE_CRC8 = 3 }; /* coding_defs.h:33-36 */
uint32_t oracle_crc24a(const uint8_t *in, int bitlen);
uint32_t oracle_crc24b(const uint8_t *in, int bitlen);
uint32_t oracle_crc24c(const uint8_t *in, int bitlen);
uint32_t oracle_crc16(const uint8_t *in, int bitlen);
uint32_t oracle_crc8(const uint8_t *in, int bitlen);
int oracle_check_crc(const uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type);

/* ---- segmentation: nr_segmentation.c:32-170 --------------------------------------------------- */
int32_t oracle_nr_segmentation(const uint8_t *input_buffer, uint8_t **output_buffers, unsigned int B,
                               unsigned int *C, unsigned int *K, unsigned int *Zout, unsigned int *F, uint8_t BG);

/* ---- rate matching / interleaving: nr_rate_matching.c ----------------------------------------- */
#define ORACLE_NR_NULL 2 /* coding_defs.h:44 */
void oracle_nr_interleaving_ldpc(uint32_t E, uint8_t Qm, const uint8_t *e, uint8_t *f);          /* :36  */
void oracle_nr_deinterleaving_ldpc(uint32_t E, uint8_t Qm, int16_t *e, const int16_t *f);        /* :310 */
int oracle_nr_get_R_ldpc_decoder(int rvidx, int E, int BG, int Z, int *llrLen, int round);      /* :390 */
int oracle_nr_rate_matching_ldpc(uint32_t Tbslbrm, uint8_t BG, uint16_t Z, const uint8_t *w, uint8_t *e, uint8_t C,
                                 uint32_t F, uint32_t Foffset, uint8_t rvidx, uint32_t E);       /* :424 */
int oracle_nr_rate_matching_ldpc_rx(uint32_t Tbslbrm, uint8_t BG, uint16_t Z, int16_t *w, const int16_t *soft_input,
                                    uint8_t C, uint8_t rvidx, uint8_t clear, uint32_t E, uint32_t F,
                                    uint32_t Foffset);                                           /* :507 */
uint32_t oracle_nr_get_E(uint32_t G, uint8_t C, uint8_t Qm, uint8_t Nl, uint8_t r); /* NR_TRANSPORT/nr_tbs_tools.c:50 */
/* caller pre-pack: openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:195-210 */
void oracle_nr_llr_prepack(const int16_t *d, int8_t *l, int BG, int Z, int K, int F, int ncols_R);

/* ---- channel model used by ldpctest: openair1/SIMULATION/TOOLS/rangen_double.c:53-123,
 *      TESTBENCH/coding_unitary_defs.h:37-49 ---------------------------------------------------- */
typedef struct { unsigned int urseed, iy, ir[98]; int iset; double gset; } oracle_rng_t;
void oracle_randominit(oracle_rng_t *s, unsigned long seed);
double oracle_uniformrandom(oracle_rng_t *s);
double oracle_gaussdouble(oracle_rng_t *s, double mean, double variance);
int8_t oracle_quantize(double D, double x, uint8_t B);
/* ldpctest.c:294-313: BPSK + AWGN + quantize into llr[2Zc .. 2Zc+n) ; llr[0..2Zc) = 0 */
void oracle_ldpctest_channel(oracle_rng_t *s, const uint8_t *coded, int n, int Zc, double sigma, int qbits, int8_t *llr);
/* ulschsim.c:533-552 / dlschsim.c:527-543: channel of the transport-channel sims, int16 LLRs; returns uncoded bit errors */
int oracle_schsim_channel(oracle_rng_t *s, const uint8_t *f, int n, double sigma, int qbits, int16_t *llr);

#ifdef __cplusplus
}
#endif
#endif
