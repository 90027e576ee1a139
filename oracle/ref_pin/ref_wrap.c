/*
 * ref_wrap.c -- TEST INFRASTRUCTURE ONLY.  The one translation unit of oracle/_ref that contains reference code:
 * it #includes reference sources WHERE THEY LIE (the Makefile passes -I paths into /root/reference; nothing is
 * copied) and exports one-line wrappers around their static inline functions.  Everything below the includes is
 * argument plumbing; no arithmetic of the path is (re)written in this file.  See ref_pin.h.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* encoder: choose_generator_matrix, encode_parity_check_part_orig + Gen_shift_value.h */
#include "ldpc_generate_coefficient.c"
/* decoder: types, LUTs, nrLDPC_init, the message-passing copies */
#include "nrLDPC_types.h"
#include "nrLDPC_init.h"
#include "nrLDPC_mPass.h"

#include "ref_pin.h"

struct ref_dec {
  t_nrLDPC_dec_params p;
  t_nrLDPC_lut lut;
  uint32_t numLLR;
};

int ref_encode_parity_check_part_orig(unsigned char *c, unsigned char *d, short BG, short Zc, short Kb, short block_length)
{
  return encode_parity_check_part_orig(c, d, BG, Zc, Kb, block_length);
}
int ref_has_generator_matrix(short BG, short Zc) { return choose_generator_matrix(BG, Zc) != NULL; }

ref_dec_t *ref_dec_new(int BG, int Z, int R)
{
  ref_dec_t *h = calloc(1, sizeof(*h));
  if (!h)
    return NULL;
  h->p.BG = (uint8_t)BG;
  h->p.Z = (uint16_t)Z;
  h->p.R = (uint8_t)R;
  h->numLLR = nrLDPC_init(&h->p, &h->lut);
  if (h->numLLR == 0 || h->lut.circShift[BG == 1 ? 8 : 5].d == NULL) { /* unknown (BG, R) or Z: nothing selected */
    free(h);
    return NULL;
  }
  return h;
}
void ref_dec_free(ref_dec_t *h) { free(h); }
uint32_t ref_dec_numLLR(const ref_dec_t *h) { return h->numLLR; }
const uint8_t *ref_dec_numCnInCnGroups(const ref_dec_t *h) { return h->lut.numCnInCnGroups; }
const uint32_t *ref_dec_startAddrCnGroups(const ref_dec_t *h) { return h->lut.startAddrCnGroups; }
const uint8_t *ref_dec_numBnInBnGroups(const ref_dec_t *h) { return h->lut.numBnInBnGroups; }
const uint32_t *ref_dec_startAddrBnGroups(const ref_dec_t *h) { return h->lut.startAddrBnGroups; }
const uint16_t *ref_dec_startAddrBnGroupsLlr(const ref_dec_t *h) { return h->lut.startAddrBnGroupsLlr; }
int ref_dec_numCnGroups(const ref_dec_t *h) { return h->p.BG == 1 ? NR_LDPC_NUM_CN_GROUPS_BG1 : NR_LDPC_NUM_CN_GROUPS_BG2; }
int ref_dec_bnInCnGroup(const ref_dec_t *h, int g) { return h->p.BG == 1 ? lut_numBnInCnGroups_BG1_R13[g] : lut_numBnInCnGroups_BG2_R15[g]; }
int ref_dec_cnInCnGroupFull(const ref_dec_t *h, int g) { return h->p.BG == 1 ? lut_numCnInCnGroups_BG1_R13[g] : lut_numCnInCnGroups_BG2_R15[g]; }

void ref_llr2llrProcBuf(ref_dec_t *h, int8_t *llr, int8_t *llrProcBuf) { nrLDPC_llr2llrProcBuf(&h->lut, llr, llrProcBuf, h->p.Z, h->p.BG); }
void ref_llr2CnProcBuf(ref_dec_t *h, int8_t *llr, int8_t *cnProcBuf)
{
  if (h->p.BG == 1)
    nrLDPC_llr2CnProcBuf_BG1(&h->lut, llr, cnProcBuf, h->p.Z);
  else
    nrLDPC_llr2CnProcBuf_BG2(&h->lut, llr, cnProcBuf, h->p.Z);
}
void ref_cn2bnProcBuf(ref_dec_t *h, int8_t *cnProcBufRes, int8_t *bnProcBuf)
{
  if (h->p.BG == 1)
    nrLDPC_cn2bnProcBuf_BG1(&h->lut, cnProcBufRes, bnProcBuf, h->p.Z);
  else
    nrLDPC_cn2bnProcBuf_BG2(&h->lut, cnProcBufRes, bnProcBuf, h->p.Z);
}
void ref_bn2cnProcBuf(ref_dec_t *h, int8_t *bnProcBufRes, int8_t *cnProcBuf)
{
  if (h->p.BG == 1)
    nrLDPC_bn2cnProcBuf_BG1(&h->lut, bnProcBufRes, cnProcBuf, h->p.Z);
  else
    nrLDPC_bn2cnProcBuf_BG2(&h->lut, bnProcBufRes, cnProcBuf, h->p.Z);
}
void ref_llrRes2llrOut(ref_dec_t *h, int8_t *llrOut, int8_t *llrRes) { nrLDPC_llrRes2llrOut(&h->lut, llrOut, llrRes, h->p.Z, h->p.BG); }
uint32_t ref_size_cn_proc_buf(void) { return NR_LDPC_SIZE_CN_PROC_BUF; }
uint32_t ref_size_bn_proc_buf(void) { return NR_LDPC_SIZE_BN_PROC_BUF; }
uint32_t ref_max_num_llr(void) { return NR_LDPC_MAX_NUM_LLR; }
