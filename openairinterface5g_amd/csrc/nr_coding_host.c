/*
 * nr_coding_host.c -- see nr_coding_host.h.  Plain C, results identical to the reference functions named there.
 */
#include "nr_coding_host.h"

int nr_hip_segmentation(uint32_t B, int BG, nr_hip_seg_t *s)
{
  /* nr_segmentation.c:44-64 */
  const uint32_t Kcb = BG == 1 ? 8448 : 3840;
  uint32_t Bprime;
  s->B = B;
  if (B <= Kcb) {
    s->L = 0;
    s->C = 1;
    Bprime = B;
  } else {
    s->L = 24;
    s->C = B / (Kcb - s->L);
    if ((Kcb - s->L) * s->C < B)
      s->C++;
    Bprime = B + s->C * s->L;
  }
  s->Kprime = Bprime / s->C; /* :66 */
  /* :68-80 */
  if (BG == 1)
    s->Kb = 22;
  else if (B > 640)
    s->Kb = 10;
  else if (B > 560)
    s->Kb = 9;
  else if (B > 192)
    s->Kb = 8;
  else
    s->Kb = 6;
  const uint32_t Z = (s->Kprime % s->Kb) > 0 ? s->Kprime / s->Kb + 1 : s->Kprime / s->Kb; /* :83-86 */
  uint32_t Zc; /* :90-133: round up to the next lifting size inside the power-of-two band */
  if (Z <= 2) Zc = 2;
  else if (Z <= 16) Zc = Z;
  else if (Z <= 32) { Zc = (Z >> 1) << 1; if (Zc < Z) Zc += 2; }
  else if (Z <= 64) { Zc = (Z >> 2) << 2; if (Zc < Z) Zc += 4; }
  else if (Z <= 128) { Zc = (Z >> 3) << 3; if (Zc < Z) Zc += 8; }
  else if (Z <= 256) { Zc = (Z >> 4) << 4; if (Zc < Z) Zc += 16; }
  else if (Z <= 384) { Zc = (Z >> 5) << 5; if (Zc < Z) Zc += 32; }
  else return -1;
  s->Zc = Zc;
  s->K = BG == 1 ? Zc * 22 : Zc * 10; /* :137-140 */
  s->F = s->K - s->Kprime;
  /* the reference copies whole payload bytes ((Kprime - L) >> 3): a segment payload that is not byte aligned is outside
   * its contract.  K and F themselves need not be multiples of 8 (TBS 24: Zc = 7, K = 70, F = 30). */
  if ((s->Kprime - s->L) & 7)
    return -1;
  return 0;
}

uint32_t nr_hip_get_E(uint32_t G, uint32_t C, uint32_t Qm, uint32_t Nl, uint32_t r)
{ /* nr_tbs_tools.c:50-64 */
  if (r <= C - ((G / (Nl * Qm)) % C) - 1)
    return Nl * Qm * (G / (Nl * Qm * C));
  return Nl * Qm * ((G / (Nl * Qm * C)) + 1);
}

static const uint8_t index_k0[2][4] = {{0, 17, 33, 56}, {0, 13, 25, 43}}; /* nr_rate_matching.c:34 */

int nr_hip_get_R_ldpc_decoder(int rvidx, int E, int BG, int Z, int *llrLen, int round)
{ /* nr_rate_matching.c:390-421 */
  const int Ncb = BG == 1 ? 66 * Z : 50 * Z;
  int infoBits = index_k0[BG - 1][rvidx] * Z + E;
  if (round == 0)
    *llrLen = infoBits;
  if (infoBits > Ncb)
    infoBits = Ncb;
  if (infoBits > *llrLen)
    *llrLen = infoBits;
  const int sysBits = BG == 1 ? 22 * Z : 10 * Z;
  const float decoderR = (float)sysBits / (infoBits + 2 * Z);
  if (BG == 2) {
    if (decoderR < 0.3333)
      return 15;
    else if (decoderR < 0.6667)
      return 13;
    return 23;
  }
  if (decoderR < 0.6667)
    return 13;
  else if (decoderR < 0.8889)
    return 23;
  return 89;
}

int nr_hip_rate_match_geometry(uint32_t Tbslbrm, int BG, uint32_t Zc, uint32_t C, uint32_t F, uint32_t K, int rv,
                               uint32_t E, nr_hip_rm_t *g)
{
  /* nr_rate_matching.c:440-465 (TX) / :523-545 (RX) */
  if (C == 0 || rv < 0 || rv > 3)
    return -1;
  const uint32_t N = BG == 1 ? 66 * Zc : 50 * Zc;
  uint32_t Ncb = N;
  if (Tbslbrm != 0) {
    const uint32_t Nref = 3 * Tbslbrm / (2 * C);
    Ncb = N < Nref ? N : Nref;
  }
  const uint32_t Foffset = K - F - 2 * Zc;
  uint32_t ind = (index_k0[BG - 1][rv] * Ncb / N) * Zc;
  if (Foffset > E || Foffset > Ncb)
    return -1; /* the reference reports "invalid parameters" */
  if (Foffset + F > Ncb)
    return -1; /* the reference's first-lap memcpy lengths go negative here: outside its contract */
  if (ind >= Foffset && ind < F + Foffset)
    ind = F + Foffset;
  g->Ncb = Ncb;
  g->Foffset = Foffset;
  g->Fin = F;
  g->V = Ncb - F;
  g->rank0 = ind < Foffset ? ind : ind - F;
  return g->V > 0 ? 0 : -1;
}

/* Columns of the decoder input that a FIRST transmission can make non-zero: the soft buffer is cleared (nr_ulsch_decoding.c:
 * 418-422) and receives E values at the transmittable positions of rank rank0 .. rank0 + E - 1 (mod V) -- everything behind
 * the last position reached stays 0, the fillers (+127) lie in the information columns.  Returns ceil((2 Zc + reach) / Zc),
 * where reach = Ncb once the lap wraps. */
uint32_t nr_hip_first_tx_columns(const nr_hip_rm_t *g, uint32_t E, uint32_t Zc)
{
  if (E == 0)
    return 2;
  const uint32_t last = g->rank0 + E - 1;
  const uint32_t reach = g->rank0 + E >= g->V ? g->Ncb : (last < g->Foffset ? last : last + g->Fin) + 1;
  return (2 * Zc + reach + Zc - 1) / Zc;
}

/* crc_byte.c:314-380.  Polynomials crc_byte.c:46-54 (left aligned); bit-serial, the host only gets here on the slow path of
 * LDPCdecoder (a caller-supplied predicate is a different function; this one is the library's own, see nrLDPC_hip.h). */
int nr_hip_check_crc(const uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type)
{
  static const uint32_t poly[4] = {0x864CFB00u, 0x80006300u, 0x10210000u, 0x9B000000u};
  static const uint32_t len[4] = {3, 3, 2, 1};
  if (crc_type > 3 || n < 8 * len[crc_type])
    return 0;
  const uint32_t L = len[crc_type], nbits = n - 8 * L;
  uint32_t reg = 0;
  for (uint32_t i = 0; i < nbits; i++) {
    const uint32_t bit = (decoded_bytes[i >> 3] >> (7 - (i & 7))) & 1u;
    const uint32_t top = (reg >> 31) ^ bit;
    reg = (reg << 1) ^ (top ? poly[crc_type] : 0u);
  }
  uint32_t stored = 0;
  for (uint32_t i = 0; i < L; i++)
    stored = (stored << 8) | decoded_bytes[(n >> 3) - L + i];
  return (reg >> (32 - 8 * L)) == stored;
}
