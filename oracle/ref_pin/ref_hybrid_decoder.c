/*
 * ref_hybrid_decoder.c -- TEST INFRASTRUCTURE ONLY.
 *
 * nrLDPC_decoder_core (openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_decoder.c:206-880) with every data-movement step
 * executed by REFERENCE-COMPILED code (ref_wrap.c: nrLDPC_init, nrLDPC_mPass.h) on the reference's own six
 * processing buffers, and the four node functions whose bodies are SIMDE intrinsics RESTATED below, one lane at a
 * time, on those buffers' layouts.  This file includes no reference header: it sees the reference only through
 * ref_pin.h.  All paths below are relative to openair1/PHY/CODING/nrLDPC_decoder/.
 *
 * Buffer layouts (what the reference-compiled copies produce and consume):
 *   cnProcBuf / cnProcBufRes: CN group g (CNs with the same number of BNs, ascending) starts at
 *     startAddrCnGroups[g]; inside it BN j of the group starts at j * bitOffsetInGroup with
 *     bitOffsetInGroup = numCnInCnGroups_<BG>_<lowest rate>[g] * NR_LDPC_ZMAX (nrLDPC_cnProc.h:423: fixed by the
 *     FULL graph, whatever R), and the numCnInCnGroups[g] CNs of this R follow back to back, Z lanes each.
 *   bnProcBuf / bnProcBufRes: the k-th NON-EMPTY BN group (BNs with the same number of CNs, ascending) starts at
 *     startAddrBnGroups[k]; inside it CN c of the group starts at c * cnOffsetInGroup with
 *     cnOffsetInGroup = numBnInBnGroups[deg-1] * NR_LDPC_ZMAX (nrLDPC_bnProc.h:125), BNs back to back, Z lanes each.
 *   llrProcBuf / llrRes: the k-th non-empty BN group starts at startAddrBnGroupsLlr[k].
 */
#include <stdlib.h>
#include <string.h>
#include "ref_pin.h"

#define ZMAX 384 /* NR_LDPC_ZMAX, nrLDPCdecoder_defs.h:39 */

static inline int sat8(int x) { return x > 127 ? 127 : (x < -128 ? -128 : x); }

/* nrLDPC_cnProc.h:47-379 (BG2), :388-877 (BG1), every group alike (e.g. :437-452): for BN j of the group
 *   sgn = sign_epi8(ones, q_k0); min = abs_epi8(q_k0);                       first other BN
 *   min = min_epu8(min, abs_epi8(q_k)); sgn = sign_epi8(sgn, q_k);           the remaining other BNs
 *   min = min_epu8(min, maxLLR = 127);  res = sign_epi8(min, sgn)
 * abs_epi8(-128) = 0x80 which min_epu8 reads as 128; sign_epi8(a, b) = -a, 0, a for b < 0, = 0, > 0. */
static void cnProc(const ref_dec_t *h, int Z, const int8_t *cnProcBuf, int8_t *cnProcBufRes)
{
  const uint8_t *numCn = ref_dec_numCnInCnGroups(h);
  const uint32_t *start = ref_dec_startAddrCnGroups(h);
  const int G = ref_dec_numCnGroups(h);
  for (int g = 0; g < G; g++) {
    if (numCn[g] == 0)
      continue;
    const int nbn = ref_dec_bnInCnGroup(h, g);
    const uint32_t off = (uint32_t)ref_dec_cnInCnGroupFull(h, g) * ZMAX;
    const int M = numCn[g] * Z;
    for (int j = 0; j < nbn; j++)
      for (int l = 0; l < M; l++) {
        int sgn = 1;
        unsigned mn = 255;
        for (int k = 0; k < nbn; k++) {
          if (k == j)
            continue;
          const int v = cnProcBuf[start[g] + k * off + l];
          const unsigned a = (uint8_t)(v < 0 ? -v : v); /* abs_epi8: 128 for -128 */
          if (a < mn)
            mn = a;
          sgn = v < 0 ? -sgn : (v == 0 ? 0 : sgn);
        }
        if (mn > 127)
          mn = 127;
        cnProcBufRes[start[g] + j * off + l] = (int8_t)(sgn < 0 ? -(int)mn : (sgn == 0 ? 0 : (int)mn));
      }
  }
}

/* nrLDPC_cnProc.h:887-1526 (BG1), :1528-1946 (BG2): group after group, 32 lanes at a time,
 *   pcRes ^= movemask_epi8(adds_epi8(cnProcBuf, cnProcBufRes)) over the group's BNs; the chunks before the last one
 *   count in full, the last one masked to its Mrem = M & 31 valid lanes and ONLY `if (Mrem)` (:964-965); a group
 *   that fails returns at once (:968-971) -- only "zero or not" is used by the caller (decoder.c:553). */
static uint32_t cnProcPc(const ref_dec_t *h, int Z, const int8_t *cnProcBuf, const int8_t *cnProcBufRes)
{
  const uint8_t *numCn = ref_dec_numCnInCnGroups(h);
  const uint32_t *start = ref_dec_startAddrCnGroups(h);
  const int G = ref_dec_numCnGroups(h);
  for (int g = 0; g < G; g++) {
    if (numCn[g] == 0)
      continue;
    const int nbn = ref_dec_bnInCnGroup(h, g);
    const uint32_t off = (uint32_t)ref_dec_cnInCnGroupFull(h, g) * ZMAX;
    const int M = numCn[g] * Z, Mrem = M & 31, M32 = (M + 31) >> 5;
    uint32_t pcResSum = 0;
    for (int i = 0; i < M32; i++) {
      const int last = i == M32 - 1;
      if (last && !Mrem)
        continue;
      const int nl = last ? Mrem : 32;
      for (int b = 0; b < nl; b++) {
        const int l = 32 * i + b;
        int par = 0;
        for (int j = 0; j < nbn; j++)
          par ^= sat8(cnProcBuf[start[g] + j * off + l] + cnProcBufRes[start[g] + j * off + l]) < 0;
        pcResSum |= (uint32_t)par << b;
      }
    }
    if (pcResSum > 0)
      return pcResSum;
  }
  return 0;
}

/* nrLDPC_bnProc.h:40-160 bnProcPc: per BN  llrRes = packs_epi16( adds_epi16 over the BN's CNs of cvtepi8_epi16(r) ,
 *   + cvtepi8_epi16(llrProcBuf) ) -- at most 30 + 1 terms of |x| <= 128, the int16 sums never saturate, the pack does.
 * Group 0 (BNs with ONE CN, :68-110): the generic function also copies llrProcBuf to bnProcBufRes and sums; the
 *   generated bnProcPc that the shipped library calls (decoder.c:369-440 under UNROLL_BN_PROC_PC; generator
 *   nrLDPC_tools/generator_bnProc/bnProcPc_gen_BG1_avx2.c:96-141) starts at the 2-CN group, so llrRes of the
 *   degree-1 columns keeps its initial 0 [F5].  deg1_generic selects which of the two is followed. */
static void bnProcPc(const ref_dec_t *h, int Z, int deg1_generic, const int8_t *bnProcBuf, int8_t *bnProcBufRes,
                     const int8_t *llrProcBuf, int8_t *llrRes)
{
  const uint8_t *numBn = ref_dec_numBnInBnGroups(h);
  const uint32_t *start = ref_dec_startAddrBnGroups(h);
  const uint16_t *startLlr = ref_dec_startAddrBnGroupsLlr(h);
  int idx = 0;
  if (deg1_generic) {
    const int M = numBn[0] * Z;
    for (int l = 0; l < M; l++) {
      bnProcBufRes[start[0] + l] = llrProcBuf[startLlr[0] + l];
      llrRes[startLlr[0] + l] = (int8_t)sat8(bnProcBuf[start[0] + l] + llrProcBuf[startLlr[0] + l]);
    }
  }
  for (int cnidx = 1; cnidx < 30; cnidx++) { /* NR_LDPC_NUM_BN_GROUPS_BG1_R13, :113 */
    if (numBn[cnidx] == 0)
      continue;
    idx++;
    const int M = numBn[cnidx] * Z;
    const uint32_t off = (uint32_t)numBn[cnidx] * ZMAX;
    for (int l = 0; l < M; l++) {
      int s = llrProcBuf[startLlr[idx] + l];
      for (int k = 0; k <= cnidx; k++)
        s += bnProcBuf[start[idx] + k * off + l];
      llrRes[startLlr[idx] + l] = (int8_t)sat8(s);
    }
  }
}

/* nrLDPC_bnProc.h:271-1310 bnProc: for the groups with >= 2 CNs, bnProcBufRes = subs_epi8(llrRes, bnProcBuf) (:325) */
static void bnProc(const ref_dec_t *h, int Z, const int8_t *bnProcBuf, int8_t *bnProcBufRes, const int8_t *llrRes)
{
  const uint8_t *numBn = ref_dec_numBnInBnGroups(h);
  const uint32_t *start = ref_dec_startAddrBnGroups(h);
  const uint16_t *startLlr = ref_dec_startAddrBnGroupsLlr(h);
  int idx = 0;
  for (int cnidx = 1; cnidx < 30; cnidx++) {
    if (numBn[cnidx] == 0)
      continue;
    idx++;
    const int M = numBn[cnidx] * Z;
    const uint32_t off = (uint32_t)numBn[cnidx] * ZMAX;
    for (int k = 0; k <= cnidx; k++)
      for (int l = 0; l < M; l++)
        bnProcBufRes[start[idx] + k * off + l] = (int8_t)sat8(llrRes[startLlr[idx] + l] - bnProcBuf[start[idx] + k * off + l]);
  }
}

/* nrLDPC_bnProc.h:1321-1345 llr2bit: out = (llr < 0), one byte each;
 * :1353-1380 llr2bitPacked: byte b = bits 8b .. 8b+7, bit 8b in the MSB (movemask of a byte-reversed load), whole
 * 32-bit words written (ceil(numLLR/32)), the lanes behind numLLR coming from the zero-initialised llrOut. */
static void llr2bit(int8_t *out, const int8_t *llrOut, uint32_t numLLR)
{
  for (uint32_t i = 0; i < numLLR; i++)
    out[i] = llrOut[i] < 0;
}
static void llr2bitPacked(int8_t *out, const int8_t *llrOut, uint32_t numLLR)
{
  const uint32_t nbytes = ((numLLR + 31) >> 5) * 4;
  memset(out, 0, nbytes);
  for (uint32_t i = 0; i < numLLR; i++)
    if (llrOut[i] < 0)
      ((uint8_t *)out)[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
}

/* the three restated functions that have no generated text to be compared with, exported so that tests/test_ref_pin.py can
 * run them beside a statement-by-statement reading of nrLDPC_cnProc.h:887-1946 and nrLDPC_bnProc.h:1321-1380 */
uint32_t ref_hybrid_cnProcPc(const ref_dec_t *h, int Z, const int8_t *cnProcBuf, const int8_t *cnProcBufRes)
{
  return cnProcPc(h, Z, cnProcBuf, cnProcBufRes);
}
void ref_hybrid_llr2bit(int8_t *out, const int8_t *llrOut, uint32_t numLLR, int packed)
{
  if (packed)
    llr2bitPacked(out, llrOut, numLLR);
  else
    llr2bit(out, llrOut, numLLR);
}

int ref_hybrid_decode(int BG, int Z, int R, int numMaxIter, int outMode,
                      int (*check_crc)(uint8_t *, uint32_t, uint8_t), int E, int crc_type, int deg1_generic,
                      const int8_t *p_llr_in, int8_t *p_out)
{
  ref_dec_t *h = ref_dec_new(BG, Z, R);
  if (!h)
    return -1;
  const uint32_t numLLR = ref_dec_numLLR(h);
  const uint32_t ncn = ref_size_cn_proc_buf(), nbn = ref_size_bn_proc_buf(), nllr = ref_max_num_llr();
  /* decoder.c:222-227: six zero-initialised buffers (+ slack for the reference's 32-lane tails) */
  int8_t *mem = calloc(2 * ncn + 2 * nbn + 4 * nllr + 6 * 64, 1);
  if (!mem) {
    ref_dec_free(h);
    return -1;
  }
  int8_t *cnProcBuf = mem, *cnProcBufRes = cnProcBuf + ncn + 64, *bnProcBuf = cnProcBufRes + ncn + 64;
  int8_t *bnProcBufRes = bnProcBuf + nbn + 64, *llrRes = bnProcBufRes + nbn + 64, *llrProcBuf = llrRes + nllr + 64;
  int8_t *llrOut = llrProcBuf + nllr + 64, *p_llr = llrOut + nllr + 64;
  memcpy(p_llr, p_llr_in, numLLR);

  /* decoder.c:233-538: set-up and the first pass (no parity check) */
  ref_llr2llrProcBuf(h, p_llr, llrProcBuf);
  ref_llr2CnProcBuf(h, p_llr, cnProcBuf);
  cnProc(h, Z, cnProcBuf, cnProcBufRes);
  ref_cn2bnProcBuf(h, cnProcBufRes, bnProcBuf);
  bnProcPc(h, Z, deg1_generic, bnProcBuf, bnProcBufRes, llrProcBuf, llrRes);
  bnProc(h, Z, bnProcBuf, bnProcBufRes, llrRes);
  ref_bn2cnProcBuf(h, bnProcBufRes, cnProcBuf);

  uint32_t numIter = 1;
  uint32_t pcRes = 1;
  while (numIter <= (uint32_t)numMaxIter && pcRes != 0) { /* decoder.c:553 */
    numIter++;
    cnProc(h, Z, cnProcBuf, cnProcBufRes);
    ref_cn2bnProcBuf(h, cnProcBufRes, bnProcBuf);
    bnProcPc(h, Z, deg1_generic, bnProcBuf, bnProcBufRes, llrProcBuf, llrRes);
    bnProc(h, Z, bnProcBuf, bnProcBufRes, llrRes);
    ref_bn2cnProcBuf(h, bnProcBufRes, cnProcBuf);
    if (!check_crc) { /* decoder.c:841-848 */
      pcRes = cnProcPc(h, Z, cnProcBuf, cnProcBufRes);
    } else if (numIter > 2) { /* decoder.c:849-861 */
      int8_t *o = outMode == 2 ? p_out : llrOut;
      memset(llrOut, 0, nllr);
      ref_llrRes2llrOut(h, o, llrRes);
      if (outMode == 0)
        llr2bitPacked(p_out, o, numLLR);
      else
        llr2bit(p_out, o, numLLR); /* also for LLRINT8, in place: :855-857 has no mode test */
      if (check_crc((uint8_t *)p_out, (uint32_t)E, (uint8_t)crc_type))
        break;
    }
  }
  if (!check_crc) { /* decoder.c:864-879 */
    int8_t *o = outMode == 2 ? p_out : llrOut;
    memset(llrOut, 0, nllr);
    ref_llrRes2llrOut(h, o, llrRes);
    if (outMode == 0)
      llr2bitPacked(p_out, o, numLLR);
    else
      llr2bit(p_out, o, numLLR);
  }
  free(mem);
  ref_dec_free(h);
  return (int)numIter;
}

/* LDPCencoder, nrLDPC_encoder/ldpc_encoder.c:44-252 with gen_code == 0: :76-92 sizes and the MSB-first unpack,
 * :214-244 = the loop that ldpc_generate_coefficient.c:363-428 holds as encode_parity_check_part_orig (called here,
 * reference-compiled), :248-251 output = c[2Zc .. block_length) || d[0 .. (nrows - punct)*Zc - removed). */
int ref_ldpc_encoder_orig(const uint8_t *in, uint8_t *out, int BG, int Zc, int Kb, int block_length)
{
  if (!ref_has_generator_matrix((short)BG, (short)Zc))
    return -1;
  const int nrows = BG == 1 ? 46 : 42, ncols = BG == 1 ? 22 : 10, rate = BG == 1 ? 3 : 5;
  static __thread uint8_t c[22 * 384], d[68 * 384];
  const int no_punctured_columns = ((nrows - 2) * Zc + block_length - block_length * rate) / Zc;
  const int removed_bit = (nrows - no_punctured_columns - 2) * Zc + block_length - block_length * rate;
  memset(c, 0, (size_t)ncols * Zc);
  memset(d, 0, (size_t)nrows * Zc);
  for (int i = 0; i < block_length; i++)
    c[i] = (uint8_t)((in[i / 8] & (128 >> (i & 7))) >> (7 - (i & 7)));
  if (ref_encode_parity_check_part_orig(c, d, (short)BG, (short)Zc, (short)Kb, (short)block_length) != 0)
    return -1;
  memcpy(out, c + 2 * Zc, (size_t)(block_length - 2 * Zc));
  memcpy(out + block_length - 2 * Zc, d, (size_t)((nrows - no_punctured_columns) * Zc - removed_bit));
  return block_length - 2 * Zc + (nrows - no_punctured_columns) * Zc - removed_bit;
}
