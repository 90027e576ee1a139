/*
 * oracle_crc_seg_rm.c -- TEST INFRASTRUCTURE ONLY (see oracle_nr_coding.h).
 * CRC, code-block segmentation, rate matching and bit interleaving around the LDPC codec,
 * restated bit-serially / element-wise from crc_byte.c, nr_segmentation.c and nr_rate_matching.c.
 */
#include <string.h>
#include "oracle_nr_coding.h"

/* ------------------------------------------------------------------------------------------------
 * CRC.  crc_byte.c:46-58 polynomials (left aligned in 32 bits), :65-84 crcbit() is the defining
 * bit-serial form; the table forms (:148-260) compute the same thing for any bit length, the message
 * being read MSB first and the result left aligned in the uint32 (callers shift by 8/16/24).
 * ---------------------------------------------------------------------------------------------- */
static uint32_t crc_bits(const uint8_t *in, int bitlen, uint32_t poly)
{
  uint32_t crc = 0;
  for (int i = 0; i < bitlen; i++) {
    uint32_t bit = (in[i >> 3] >> (7 - (i & 7))) & 1u;
    if (((crc >> 31) ^ bit) & 1u)
      crc = (crc << 1) ^ poly;
    else
      crc <<= 1;
  }
  return crc;
}
uint32_t oracle_crc24a(const uint8_t *in, int bitlen) { return crc_bits(in, bitlen, 0x864cfb00u); }
uint32_t oracle_crc24b(const uint8_t *in, int bitlen) { return crc_bits(in, bitlen, 0x80006300u); }
uint32_t oracle_crc24c(const uint8_t *in, int bitlen) { return crc_bits(in, bitlen, 0xb2b11700u); }
uint32_t oracle_crc16(const uint8_t *in, int bitlen) { return crc_bits(in, bitlen, 0x10210000u); }
uint32_t oracle_crc8(const uint8_t *in, int bitlen) { return crc_bits(in, bitlen, 0x9B000000u); }

/* crc_byte.c:314-380 */
int oracle_check_crc(const uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type)
{
  uint32_t crc = 0, oldcrc = 0;
  int crc_len = 0;
  switch (crc_type) {
    case ORACLE_CRC24_A:
    case ORACLE_CRC24_B: crc_len = 3; break;
    case ORACLE_CRC16: crc_len = 2; break;
    case ORACLE_CRC8: crc_len = 1; break;
    default: return 0;
  }
  for (int i = 0; i < crc_len; i++)
    oldcrc |= (uint32_t)decoded_bytes[(n >> 3) - crc_len + i] << ((crc_len - 1 - i) << 3);
  switch (crc_type) {
    case ORACLE_CRC24_A: oldcrc &= 0x00ffffff; crc = oracle_crc24a(decoded_bytes, (int)n - 24) >> 8; break;
    case ORACLE_CRC24_B: oldcrc &= 0x00ffffff; crc = oracle_crc24b(decoded_bytes, (int)n - 24) >> 8; break;
    case ORACLE_CRC16: oldcrc &= 0x0000ffff; crc = oracle_crc16(decoded_bytes, (int)n - 16) >> 16; break;
    case ORACLE_CRC8: oldcrc &= 0x000000ff; crc = oracle_crc8(decoded_bytes, (int)n - 8) >> 24; break;
  }
  return crc == oldcrc;
}

/* ------------------------------------------------------------------------------------------------
 * nr_segmentation.c:32-170
 * ---------------------------------------------------------------------------------------------- */
int32_t oracle_nr_segmentation(const uint8_t *input_buffer, uint8_t **output_buffers, unsigned int B,
                               unsigned int *C, unsigned int *K, unsigned int *Zout, unsigned int *F, uint8_t BG)
{
  unsigned int L, Bprime, Z, Kb, Kprime;
  const unsigned int Kcb = BG == 1 ? 8448 : 3840;                       /* :44-47 */
  if (B <= Kcb) { L = 0; *C = 1; Bprime = B; }                          /* :49-52 */
  else {                                                                /* :53-63 */
    L = 24;
    *C = B / (Kcb - L);
    if ((Kcb - L) * (*C) < B) *C = *C + 1;
    Bprime = B + (*C) * L;
  }
  Kprime = Bprime / (*C);                                               /* :66 */
  if (BG == 1) Kb = 22;                                                 /* :68-80 */
  else if (B > 640) Kb = 10;
  else if (B > 560) Kb = 9;
  else if (B > 192) Kb = 8;
  else Kb = 6;
  Z = (Kprime % Kb) > 0 ? Kprime / Kb + 1 : Kprime / Kb;                /* :83-86 */
  /* :90-133 smallest lifting size >= Z, by rounding up inside each power-of-two band */
  if (Z <= 2) *K = 2;
  else if (Z <= 16) *K = Z;
  else if (Z <= 32) { *K = (Z >> 1) << 1; if (*K < Z) *K += 2; }
  else if (Z <= 64) { *K = (Z >> 2) << 2; if (*K < Z) *K += 4; }
  else if (Z <= 128) { *K = (Z >> 3) << 3; if (*K < Z) *K += 8; }
  else if (Z <= 256) { *K = (Z >> 4) << 4; if (*K < Z) *K += 16; }
  else if (Z <= 384) { *K = (Z >> 5) << 5; if (*K < Z) *K += 32; }
  else return -1;
  *Zout = *K;
  *K = BG == 1 ? *K * 22 : *K * 10;                                     /* :137-140 (10 even when Kb < 10) */
  *F = *K - Kprime;
  if (input_buffer && output_buffers) {                                 /* :147-175 */
    unsigned int s = 0;
    for (unsigned int r = 0; r < *C; r++) {
      unsigned int k = 0;
      while (k < ((Kprime - L) >> 3)) output_buffers[r][k++] = input_buffer[s++];
      if (*C > 1) {
        uint32_t crc = oracle_crc24b(output_buffers[r], (int)(Kprime - L)) >> 8;
        output_buffers[r][(Kprime - L) >> 3] = (uint8_t)(crc >> 16);
        output_buffers[r][1 + ((Kprime - L) >> 3)] = (uint8_t)(crc >> 8);
        output_buffers[r][2 + ((Kprime - L) >> 3)] = (uint8_t)crc;
      }
      if (*F > 0)
        for (k = Kprime >> 3; k < (*K) >> 3; k++) output_buffers[r][k] = 0;
    }
  }
  return (int32_t)Kb;
}

/* ------------------------------------------------------------------------------------------------
 * nr_rate_matching.c
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t index_k0[2][4] = {{0, 17, 33, 56}, {0, 13, 25, 43}}; /* :34 */

void oracle_nr_interleaving_ldpc(uint32_t E, uint8_t Qm, const uint8_t *e, uint8_t *f) /* :36, loops :262-268 */
{
  const uint32_t EQm = E / Qm;
  memset(f, 0, E);
  for (uint32_t j = 0; j < EQm; j++)
    for (uint32_t i = 0; i < Qm; i++)
      f[i + j * Qm] = e[i * EQm + j];
}

void oracle_nr_deinterleaving_ldpc(uint32_t E, uint8_t Qm, int16_t *e, const int16_t *f) /* :310-388 */
{
  const uint32_t EQm = E / Qm;
  for (uint32_t j = 0; j < EQm; j++)
    for (uint32_t i = 0; i < Qm; i++)
      e[i * EQm + j] = f[i + j * Qm];
}

int oracle_nr_get_R_ldpc_decoder(int rvidx, int E, int BG, int Z, int *llrLen, int round) /* :390-421 */
{
  int Ncb = BG == 1 ? 66 * Z : 50 * Z;
  int infoBits = index_k0[BG - 1][rvidx] * Z + E;
  if (round == 0) *llrLen = infoBits;
  if (infoBits > Ncb) infoBits = Ncb;
  if (infoBits > *llrLen) *llrLen = infoBits;
  int sysBits = BG == 1 ? 22 * Z : 10 * Z;
  float decoderR = (float)sysBits / (infoBits + 2 * Z);
  if (BG == 2) {
    if (decoderR < 0.3333) return 15;
    else if (decoderR < 0.6667) return 13;
    else return 23;
  } else {
    if (decoderR < 0.6667) return 13;
    else if (decoderR < 0.8889) return 23;
    else return 89;
  }
}

int oracle_nr_rate_matching_ldpc(uint32_t Tbslbrm, uint8_t BG, uint16_t Z, const uint8_t *w, uint8_t *e, uint8_t C,
                                 uint32_t F, uint32_t Foffset, uint8_t rvidx, uint32_t E) /* :424-505 */
{
  uint32_t Ncb, ind, k = 0, Nref, N;
  if (C == 0) return -1;
  N = BG == 1 ? 66u * Z : 50u * Z;
  if (Tbslbrm == 0) Ncb = N;
  else { Nref = 3 * Tbslbrm / (2 * C); Ncb = N < Nref ? N : Nref; }
  ind = (index_k0[BG - 1][rvidx] * Ncb / N) * Z;
  if (Foffset > E) return -1;
  if (Foffset > Ncb) return -1;
  if (ind >= Foffset && ind < F + Foffset) ind = F + Foffset;
  if (ind < Foffset) {
    memcpy(e, w + ind, Foffset - ind);
    if (E + F <= Ncb - ind) { memcpy(e + Foffset - ind, w + Foffset + F, E - Foffset + ind); k = E; }
    else { memcpy(e + Foffset - ind, w + Foffset + F, Ncb - Foffset - F); k = Ncb - F - ind; }
  } else {
    if (E <= Ncb - ind) { memcpy(e, w + ind, E); k = E; }
    else { memcpy(e, w + ind, Ncb - ind); k = Ncb - ind; }
  }
  while (k < E)
    for (ind = 0; ind < Ncb && k < E; ind++)
      if (w[ind] != ORACLE_NR_NULL) e[k++] = w[ind];
  return 0;
}

int oracle_nr_rate_matching_ldpc_rx(uint32_t Tbslbrm, uint8_t BG, uint16_t Z, int16_t *w, const int16_t *soft_input,
                                    uint8_t C, uint8_t rvidx, uint8_t clear, uint32_t E, uint32_t F,
                                    uint32_t Foffset) /* :507-603 */
{
  uint32_t Ncb, ind, k, Nref, N;
  if (C == 0) return -1;
  N = BG == 1 ? 66u * Z : 50u * Z;
  if (Tbslbrm == 0) Ncb = N;
  else { Nref = 3 * Tbslbrm / (2 * C); Ncb = N < Nref ? N : Nref; }
  ind = (index_k0[BG - 1][rvidx] * Ncb / N) * Z;
  if (Foffset > E) return -1;
  if (Foffset > Ncb) return -1;
  if (clear == 1) memset(w, 0, Ncb * sizeof(int16_t));
  k = 0;
  if (ind < Foffset)
    for (; ind < Foffset && k < E; ind++) w[ind] += soft_input[k++];
  if (ind >= Foffset && ind < Foffset + F) ind = Foffset + F;
  for (; ind < Ncb && k < E; ind++) w[ind] += soft_input[k++];
  while (k < E) {
    for (ind = 0; ind < Foffset && k < E; ind++) w[ind] += soft_input[k++];
    for (ind = Foffset + F; ind < Ncb && k < E; ind++) w[ind] += soft_input[k++];
  }
  return 0;
}

/* openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:195-210: d (int16, circular-buffer coordinates, i.e.
 * without the 2Z punctured bits) -> decoder input l (int8): first 2Z zero, fillers +127, saturate. */
void oracle_nr_llr_prepack(const int16_t *d, int8_t *l, int BG, int Z, int K, int F, int ncols_R)
{
  (void)BG;
  const int n = ncols_R * Z, KF = K - F;
  for (int i = 0; i < n; i++) {
    int v;
    if (i < 2 * Z) v = 0;
    else if (i < KF) v = d[i - 2 * Z];
    else if (i < K) v = 127;
    else v = d[i - 2 * Z];
    l[i] = (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
  }
}

/* ------------------------------------------------------------------------------------------------
 * openair1/SIMULATION/TOOLS/rangen_double.c:53-123 and TESTBENCH/coding_unitary_defs.h:37-49
 * ---------------------------------------------------------------------------------------------- */
#include <math.h>
void oracle_randominit(oracle_rng_t *s, unsigned long seed) /* :53-78 (OAI_RNGSEED handling is the caller's) */
{
  const unsigned int a = 1664525u;
  s->urseed = (unsigned int)seed;
  if (s->urseed % 2 == 0) s->urseed += 1;
  for (int i = 1; i <= 97; i++) { s->urseed = a * s->urseed; s->ir[i] = s->urseed; }
  s->iy = 1;
  s->iset = 0;
  s->gset = 0;
}
double oracle_uniformrandom(oracle_rng_t *s) /* :87-96 */
{
  const double mod = 4294967296.0;
  int j = 1 + 97.0 * s->iy / mod;
  s->iy = s->ir[j];
  s->urseed = 1664525u * s->urseed;
  s->ir[j] = s->urseed;
  return (double)s->iy / mod;
}
double oracle_gaussdouble(oracle_rng_t *s, double mean, double variance) /* :102-123 */
{
  double fac, r, v1, v2;
  if (s->iset == 0) {
    do {
      v1 = 2.0 * oracle_uniformrandom(s) - 1.0;
      v2 = 2.0 * oracle_uniformrandom(s) - 1.0;
      r = v1 * v1 + v2 * v2;
    } while (r >= 1.0);
    fac = sqrt(-2.0 * log(r) / r);
    s->gset = v1 * fac;
    s->iset = 1;
    return sqrt(variance) * v2 * fac + mean;
  } else {
    s->iset = 0;
    return sqrt(variance) * s->gset + mean;
  }
}
int8_t oracle_quantize(double D, double x, uint8_t B) /* coding_unitary_defs.h:37-49 */
{
  double qxd = floor(x / D);
  int16_t maxlev = 1 << (B - 1);
  if (qxd <= -maxlev) qxd = -maxlev;
  else if (qxd >= maxlev) qxd = maxlev - 1;
  return (int8_t)qxd;
}
void oracle_ldpctest_channel(oracle_rng_t *s, const uint8_t *coded, int n, int Zc, double sigma, int qbits, int8_t *llr)
{ /* TESTBENCH/ldpctest.c:294-313 */
  memset(llr, 0, (size_t)2 * Zc);
  for (int i = 0; i < n; i++) {
    double mod = coded[i] == 0 ? 1.0 : -1.0;
    llr[2 * Zc + i] = oracle_quantize(sigma / 4.0 / 4.0, mod + sigma * oracle_gaussdouble(s, 0.0, 1.0), (uint8_t)qbits);
  }
}

/* openair1/SIMULATION/NR_PHY/ulschsim.c:533-552 = dlschsim.c:527-543: the transport-channel sims' BPSK + AWGN channel on
 * the rate-matched, interleaved bits f[0..n): +1 for a 0 bit, -1 for a 1 bit, noise sigma * gaussdouble(0, 1), then
 * `(short)quantize(sigma / 4.0 / 4.0, x, qbits)` -- the 8-bit quantiser's char result widened to the int16 LLR array the
 * decoding chains take.  Returns the number of hard-decision errors (ulschsim.c:555-562 errors_bit_uncoded). */
int oracle_schsim_channel(oracle_rng_t *s, const uint8_t *f, int n, double sigma, int qbits, int16_t *llr)
{
  int uncoded_errors = 0;
  for (int i = 0; i < n; i++) {
    double mod = f[i] == 0 ? 1.0 : -1.0;
    llr[i] = (int16_t)oracle_quantize(sigma / 4.0 / 4.0, mod + sigma * oracle_gaussdouble(s, 0.0, 1.0), (uint8_t)qbits);
    uncoded_errors += (llr[i] < 0) != (f[i] != 0);
  }
  return uncoded_errors;
}

/* openair1/PHY/NR_TRANSPORT/nr_tbs_tools.c:50-64 */
uint32_t oracle_nr_get_E(uint32_t G, uint8_t C, uint8_t Qm, uint8_t Nl, uint8_t r)
{
  uint32_t E;
  uint8_t Cprime = C;
  if (r <= Cprime - ((G / (Nl * Qm)) % Cprime) - 1)
    E = Nl * Qm * (G / (Nl * Qm * Cprime));
  else
    E = Nl * Qm * ((G / (Nl * Qm * Cprime)) + 1);
  return E;
}
