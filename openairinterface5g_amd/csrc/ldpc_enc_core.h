/*
 * ldpc_enc_core.h -- per-thread phases of the NR LDPC encoder kernel (HIP device code and, for the CPU
 * emulation test, plain host C++).  A phase has no dependency between threads; the caller separates
 * phases with a workgroup barrier.
 *
 * Output contract = LDPCencoder's (reference openair1/PHY/CODING/nrLDPC_encoder/ldpc_encoder.c:44-252):
 *   :88-92   input bits unpacked MSB first;
 *   :232     only information columns < Kb enter the parity sums;
 *   :248-251 output = c[2Zc..K) || parity, one bit per byte.
 * The reference multiplies by a stored generator matrix (Gen_shift_value.h); here the parity bits are
 * solved from H x = 0 using the 38.212 structure (dual-diagonal core, identity extension).
 */
#ifndef LDPC_ENC_CORE_H
#define LDPC_ENC_CORE_H
#include "ldpc_dec_core.h"

#define LDPC_ENC_NUM_PHASES 8

LDPC_HD int ldpc_wrap(int i, int Z) { return i >= Z ? i - Z : i; }

/* x: code word bits x[c*Z + u] (ncols*Z bytes); lam: 4*Z bytes scratch. */
LDPC_HD void ldpc_enc_phase(int phase, ldpc_code_ptr_t code, int Kb, const uint8_t *__restrict__ in, uint8_t *x,
                            uint8_t *lam, uint8_t *__restrict__ out, int tid, int nt)
{
  const int Z = code->Z, kbf = code->kb_full, K = kbf * Z, ncols = code->ncols;
  if (phase == 0) {
    for (int i = tid; i < K; i += nt)
      x[i] = (in[i >> 3] >> (7 - (i & 7))) & 1;
    for (int i = K + tid; i < ncols * Z; i += nt)
      x[i] = 0;
  } else if (phase == 1) {
    /* lambda_row[t] = XOR over the information edges of core row `row` */
    for (int row = 0; row < 4; row++) {
      const int e0 = code->row_ptr[row], e1 = code->row_ptr[row + 1];
      for (int t = tid; t < Z; t += nt) {
        int acc = 0;
        for (int e = e0; e < e1; e++) {
          const uint32_t info = code->e_info[e];
          if (code->e_col[e] < Kb)
            acc ^= x[(int)(info >> 16) + ldpc_wrap(t + (int)(info & 0xffffu), Z)];
        }
        lam[row * Z + t] = (uint8_t)acc;
      }
    }
  } else if (phase == 2) {
    /* first core parity column from the sum of the four core rows */
    const int s0 = code->enc_p0_shift;
    for (int t = tid; t < Z; t += nt)
      x[kbf * Z + ldpc_wrap(t + s0, Z)] = lam[t] ^ lam[Z + t] ^ lam[2 * Z + t] ^ lam[3 * Z + t];
  } else if (phase <= 5) {
    /* the other three along the dual diagonal */
    const int step = phase - 3;
    const int row = code->enc_row[step], unk = code->enc_unk[step], us = code->enc_ushift[step];
    const int nk = code->enc_nk[step];
    for (int t = tid; t < Z; t += nt) {
      int acc = lam[row * Z + t];
      for (int k = 0; k < nk; k++)
        acc ^= x[(kbf + code->enc_kcol[step][k]) * Z + ldpc_wrap(t + code->enc_kshift[step][k], Z)];
      x[(kbf + unk) * Z + ldpc_wrap(t + us, Z)] = (uint8_t)acc;
    }
  } else if (phase == 6) {
    /* extension parity: row r >= 4 closes on its own degree-1 column (last edge of the row, shift 0) */
    const int zw = code->zw, nslots = (code->nrows - 4) * zw;
    const int lane = tid & 63, wave = LDPC_UNIFORM(tid >> 6), nw = nt >> 6;
    for (int k = wave; k < nslots; k += nw) {
      const int row = 4 + k / zw, t = ((k % zw) << 6) + lane;
      const int e0 = code->row_ptr[row], e1 = code->row_ptr[row + 1] - 1;
      if (t < Z) {
        int acc = 0;
        for (int e = e0; e < e1; e++) {
          const uint32_t info = code->e_info[e];
          const int c = code->e_col[e];
          if (c < Kb || c >= kbf)
            acc ^= x[(int)(info >> 16) + ldpc_wrap(t + (int)(info & 0xffffu), Z)];
        }
        x[(int)(code->e_info[e1] >> 16) + t] = (uint8_t)acc;
      }
    }
  } else {
    const int N = (ncols - 2) * Z;
    const uint8_t *xs = x + 2 * Z;
    if (((reinterpret_cast<uintptr_t>(out) | (uintptr_t)(2 * Z)) & 3) == 0) {
      const int n4 = N >> 2;
      for (int i = tid; i < n4; i += nt)
        reinterpret_cast<uint32_t *>(out)[i] = reinterpret_cast<const uint32_t *>(xs)[i];
      for (int i = (n4 << 2) + tid; i < N; i += nt)
        out[i] = xs[i];
    } else {
      for (int i = tid; i < N; i += nt)
        out[i] = xs[i];
    }
  }
}
#endif
