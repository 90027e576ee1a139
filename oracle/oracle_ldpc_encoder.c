/*
 * oracle_ldpc_encoder.c -- TEST INFRASTRUCTURE ONLY (see oracle_nr_coding.h).
 *
 * The reference encoder (nrLDPC_encoder/ldpc_encoder.c:44-252) multiplies the information bits by a
 * pre-computed generator matrix stored as shift lists (Gen_shift_value.h, ~100k numbers).  The code
 * is systematic and H = [A | B ; C | D | I] has an invertible parity part, so the code word for given
 * information bits is unique: this restatement solves H x = 0 directly from the 38.212 tables
 * (dual-diagonal core, then the identity extension) and must give the same bits.  The I/O contract
 * restated from ldpc_encoder.c:
 *   :88-92   input unpacked MSB first;
 *   :232     only the first Kb information columns enter the parity sums;
 *   :248-251 output = c[2Zc..K) followed by the parity bits, one bit per byte.
 */
#include <string.h>
#include "oracle_nr_coding.h"
#include "oracle_bg_tables.h"

/* y[t] ^= x[(t + s) mod Z] */
static void xor_rot(uint8_t *y, const uint8_t *x, int s, int Z)
{
  for (int t = 0; t < Z; t++)
    y[t] ^= x[(t + s) % Z];
}

int oracle_ldpc_encode(int BG, int Zc, int Kb, const uint8_t *in, uint8_t *out)
{
  oracle_graph_t g;
  if (oracle_ldpc_graph(BG, Zc, BG == 1 ? 13 : 15, &g) != 0)
    return -1;
  const int Z = Zc, kbf = BG == 1 ? 22 : 10, K = kbf * Z;
  static __thread uint8_t x[68 * 384];
  memset(x, 0, sizeof(x));
  for (int i = 0; i < K; i++) /* ldpc_encoder.c:88-92 */
    x[i] = (in[i / 8] >> (7 - (i & 7))) & 1;

  /* lambda_i = sum over information columns < Kb of the rotated column, core rows 0..3 */
  uint8_t lam[4][384];
  memset(lam, 0, sizeof(lam));
  int pcol_shift[4][4], pcol_has[4][4]; /* core parity sub-matrix: rows 0..3 x columns kbf..kbf+3 */
  memset(pcol_has, 0, sizeof(pcol_has));
  for (int r = 0; r < 4; r++)
    for (int e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) {
      const int c = g.col[e];
      if (c < kbf) {
        if (c < Kb)
          xor_rot(lam[r], &x[c * Z], g.shift[e], Z);
      } else {
        pcol_has[r][c - kbf] = 1;
        pcol_shift[r][c - kbf] = g.shift[e];
      }
    }
  /* first parity column appears in three core rows, two of them with the same shift: summing the
   * four rows cancels everything but one rotated copy of p0 (38.212 5.3.2 structure) */
  uint8_t sum[384];
  memset(sum, 0, sizeof(sum));
  for (int r = 0; r < 4; r++)
    for (int t = 0; t < Z; t++)
      sum[t] ^= lam[r][t];
  int cnt[384];
  memset(cnt, 0, sizeof(cnt));
  for (int r = 0; r < 4; r++)
    if (pcol_has[r][0])
      cnt[pcol_shift[r][0]]++;
  int s0 = -1;
  for (int s = 0; s < Z; s++)
    if (cnt[s] & 1) {
      if (s0 >= 0)
        return -1;
      s0 = s;
    }
  if (s0 < 0)
    return -1;
  uint8_t *p = &x[kbf * Z];                 /* p[j*Z + t], j = 0..3 */
  for (int t = 0; t < Z; t++)               /* sum[t] = p0[(t + s0) mod Z] */
    p[(t + s0) % Z] = sum[t];
  /* remaining three core parity columns: repeatedly take a row with exactly one unknown */
  int known[4] = {1, 0, 0, 0};
  for (int round = 0; round < 3; round++) {
    int done = 0;
    for (int r = 0; r < 4 && !done; r++) {
      int unk = -1, nunk = 0;
      for (int j = 0; j < 4; j++)
        if (pcol_has[r][j] && !known[j]) {
          unk = j;
          nunk++;
        }
      if (nunk != 1)
        continue;
      uint8_t acc[384];
      memcpy(acc, lam[r], Z);
      for (int j = 0; j < 4; j++)
        if (pcol_has[r][j] && known[j])
          xor_rot(acc, &p[j * Z], pcol_shift[r][j], Z);
      for (int t = 0; t < Z; t++) /* acc[t] = p_unk[(t + s) mod Z] */
        p[unk * Z + (t + pcol_shift[r][unk]) % Z] = acc[t];
      known[unk] = 1;
      done = 1;
    }
    if (!done)
      return -1;
  }
  /* extension rows: the degree-1 column (shift 0) closes each row */
  for (int r = 4; r < g.nrows; r++) {
    uint8_t acc[384];
    memset(acc, 0, sizeof(acc));
    int ext = -1;
    for (int e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++) {
      const int c = g.col[e];
      if (c >= g.ncore) {
        ext = c;
        continue;
      }
      if (c < kbf && c >= Kb)
        continue;
      xor_rot(acc, &x[c * Z], g.shift[e], Z);
    }
    memcpy(&x[ext * Z], acc, Z);
  }
  const int N = (g.ncols - 2) * Z;
  memcpy(out, &x[2 * Z], N); /* ldpc_encoder.c:248-251 */
  return N;
}

int oracle_ldpc_syndrome_weight(int BG, int Z, const uint8_t *x)
{
  oracle_graph_t g;
  if (oracle_ldpc_graph(BG, Z, BG == 1 ? 13 : 15, &g) != 0)
    return -1;
  int w = 0;
  for (int r = 0; r < g.nrows; r++)
    for (int t = 0; t < Z; t++) {
      int par = 0;
      for (int e = g.row_ptr[r]; e < g.row_ptr[r + 1]; e++)
        par ^= x[g.col[e] * Z + (t + g.shift[e]) % Z] & 1;
      w += par;
    }
  return w;
}
