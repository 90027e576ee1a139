/*
 * ref_pin.h -- TEST INFRASTRUCTURE ONLY (like everything under oracle/).
 *
 * C ABI of oracle/_ref/libref_nrldpc.so: the pieces of the reference's LDPC path that compile in this image
 * from the reference's own sources with NOTHING but the reference's own headers (no SIMDE, no stand-ins, no
 * generated code), built by oracle/ref_pin/Makefile from the files where they lie under /root/reference:
 *
 *   reference-COMPILED (ref_wrap.c is a TU that #includes them and exports one-line wrappers):
 *     openair1/PHY/CODING/nrLDPC_encoder/ldpc_generate_coefficient.c   choose_generator_matrix :38-361,
 *                                                                      encode_parity_check_part_orig :363-428
 *                                                                      (+ its tables Gen_shift_value.h)
 *     openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_init.h                 nrLDPC_init :58-1428 (LUT selection, numLLR)
 *                                                                      (+ nrLDPC_lut.h, nrLDPCdecoder_defs.h)
 *     openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_mPass.h                llr2llrProcBuf :98, llr2CnProcBuf_BG1/2 :128,193,
 *                                                                      cn2bnProcBuf_BG2/1 :226,260, bn2cnProcBuf_BG2/1
 *                                                                      :306,344, llrRes2llrOut :394
 *   reference-GENERATED text (ref_gen_main.c links the code generators that compile without SIMDE -- AVX2: cnProc BG1,
 *   bnProcPc BG1, bnProc BG2; AVX-512: cnProc BG1 and BG2 -- and runs them; oracle/_ref/gen/*): the node functions the SHIPPED decoder calls.  They cannot be
 *   compiled here; tests/test_ref_pin.py reads them statement by statement: formula and wiring of [D2], [D3]/[F5], [D5].
 *   RESTATED here because their bodies are SIMDE intrinsics (ref_hybrid_decoder.c, each citing file:line):
 *     nrLDPC_cnProc_BG1/2, nrLDPC_cnProcPc_BG1/2 (nrLDPC_cnProc.h), nrLDPC_bnProcPc, nrLDPC_bnProc,
 *     nrLDPC_llr2bit(Packed) (nrLDPC_bnProc.h), the pass loop of nrLDPC_decoder_core (nrLDPC_decoder.c:206-880),
 *     the bit unpack and output assembly of LDPCencoder (ldpc_encoder.c:86-92,248-251).
 *
 * So: the encoder's parity computation and its generator tables, the decoder's LUT selection and ALL of the
 * decoder's data movement (circular shifts, CN-/BN-buffer addressing, degree-1 handling [D1][D6], output
 * reordering) are executed by reference-compiled code; only the per-element node arithmetic is restated -- on the
 * reference's own buffer layouts, so that it is an independent second formulation next to oracle_ldpc_decoder.c's
 * (edge, lane) one.  Nothing here is shipped, and the product never links or loads it.
 */
#ifndef REF_PIN_H
#define REF_PIN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference-compiled: encoder ---------------------------------------------------------------- */
/* encode_parity_check_part_orig(c, d, BG, Zc, Kb, block_length), ldpc_generate_coefficient.c:363.  c: one bit per
 * byte (or up to 8 bit-sliced segments per byte: the body only XORs bytes), ncols*Zc bytes, ROTATED IN PLACE by the
 * call (net rotation Zc = identity); d: nrows*Zc bytes.  Returns 0, -1 on a bad BG. */
int ref_encode_parity_check_part_orig(unsigned char *c, unsigned char *d, short BG, short Zc, short Kb, short block_length);
/* choose_generator_matrix(BG, Zc) != NULL, :38 */
int ref_has_generator_matrix(short BG, short Zc);

/* ---- reference-compiled: decoder set-up and data movement ------------------------------------------ */
typedef struct ref_dec ref_dec_t; /* t_nrLDPC_dec_params + t_nrLDPC_lut of one (BG, Z, R) */
ref_dec_t *ref_dec_new(int BG, int Z, int R); /* nrLDPC_init(); NULL when it yields numLLR == 0 */
void ref_dec_free(ref_dec_t *);
uint32_t ref_dec_numLLR(const ref_dec_t *);
/* LUT views, for the restated node functions and for table tests */
const uint8_t *ref_dec_numCnInCnGroups(const ref_dec_t *);     /* [9] / [6] */
const uint32_t *ref_dec_startAddrCnGroups(const ref_dec_t *);
const uint8_t *ref_dec_numBnInBnGroups(const ref_dec_t *);     /* [30] */
const uint32_t *ref_dec_startAddrBnGroups(const ref_dec_t *);
const uint16_t *ref_dec_startAddrBnGroupsLlr(const ref_dec_t *);
int ref_dec_numCnGroups(const ref_dec_t *);                    /* NR_LDPC_NUM_CN_GROUPS_BG1 / _BG2 */
int ref_dec_bnInCnGroup(const ref_dec_t *, int g);             /* lut_numBnInCnGroups_BG*_R1x[g] */
int ref_dec_cnInCnGroupFull(const ref_dec_t *, int g);         /* lut_numCnInCnGroups_BG1_R13 / _BG2_R15 [g]: fixes the buffer strides */
void ref_llr2llrProcBuf(ref_dec_t *, int8_t *llr, int8_t *llrProcBuf);
void ref_llr2CnProcBuf(ref_dec_t *, int8_t *llr, int8_t *cnProcBuf);
void ref_cn2bnProcBuf(ref_dec_t *, int8_t *cnProcBufRes, int8_t *bnProcBuf);
void ref_bn2cnProcBuf(ref_dec_t *, int8_t *bnProcBufRes, int8_t *cnProcBuf);
void ref_llrRes2llrOut(ref_dec_t *, int8_t *llrOut, int8_t *llrRes);
/* sizes of the reference's stack buffers (nrLDPCdecoder_defs.h:101-107) */
uint32_t ref_size_cn_proc_buf(void);
uint32_t ref_size_bn_proc_buf(void);
uint32_t ref_max_num_llr(void);

/* ---- hybrid decoder: reference data movement + restated node arithmetic (ref_hybrid_decoder.c) ---- */
/* Follows nrLDPC_decoder_core.  outMode 0 BIT / 1 BITINT8 / 2 LLRINT8 (nrLDPC_types.h:75-79); check_crc may be NULL
 * (parity-check stop) or a predicate called exactly as decoder.c:857 calls it.  deg1_generic = 0: the shipped
 * (generated, UNROLL_*) bnProcPc, which leaves the 1-CN group alone [F5]; 1: the generic bnProcPc (nrLDPC_bnProc.h:
 * 68-110), which also sums the degree-1 columns.  Returns the pass count like decoder.c:880, -1 on bad parameters. */
int ref_hybrid_decode(int BG, int Z, int R, int numMaxIter, int outMode,
                      int (*check_crc)(uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type), int E, int crc_type,
                      int deg1_generic, const int8_t *p_llr, int8_t *p_out);
/* the hybrid's restated parity check and hard decision alone (tests: beside the reference's text read statement by statement) */
uint32_t ref_hybrid_cnProcPc(const ref_dec_t *h, int Z, const int8_t *cnProcBuf, const int8_t *cnProcBufRes);
void ref_hybrid_llr2bit(int8_t *out, const int8_t *llrOut, uint32_t numLLR, int packed);
/* LDPCencoder of ldpc_encoder.c:44-252 with gen_code == 0: unpack, reference-compiled parity part, assemble.
 * in: block_length/8 bytes (MSB first); out: one bit per byte; returns the output length (:251), -1 = no generator. */
int ref_ldpc_encoder_orig(const uint8_t *in, uint8_t *out, int BG, int Zc, int Kb, int block_length);

#ifdef __cplusplus
}
#endif
#endif
