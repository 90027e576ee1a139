/*
 * ldpc_enc_packed_core.h -- bit-packed NR LDPC encoder: per-thread phases (HIP device code and, for the CPU emulation
 * test, plain host C++).  Same contract and the same solve order as ldpc_enc_core.h (reference
 * openair1/PHY/CODING/nrLDPC_encoder/ldpc_encoder.c:44-252: input unpacked MSB first :88-92, information columns
 * < Kb enter the parity sums :232, output = c[2Zc..K) || parity, one bit per byte :248-251), different data layout:
 *
 *   a column of the lifted code word = Zc bits, 32 per LDS word, bit t of the column in word t >> 5, bit t & 31;
 *   a circular shift by s = one 32-bit window per word at bit offset 32w + s of the column's PERIODIC string
 *   (the column repeated to cw = 2W + 1 words), fetched as two words + v_alignbit_b32;
 *   a work item = (row, word): one XOR per edge per 32 lanes instead of one per lane.
 *
 * LDS (words): B[ncols][W+1]  masked columns (bits >= Zc zero, one zero word behind) -- the result;
 *              X[kbf+4][cw]   periodic strings of the information and core parity columns;
 *              LB[4][W+1], LX[4][cw]  the same for the four core-row partial sums lambda;
 *              ET[nedges] = column << 16 | shift, RP[nrows+1] = first edge of each row (staged from the descriptor:
 *              the per-edge loops are chains of dependent table reads, LDS latency instead of global latency).
 * Phases are separated by workgroup barriers; phases 0 .. LDPC_ENCP_SOLVE_PHASES-1 leave the code word in B, the
 * last phase writes the reference's byte-per-bit output (a fused caller reads B instead).
 */
#ifndef LDPC_ENC_PACKED_CORE_H
#define LDPC_ENC_PACKED_CORE_H
#include "ldpc_dec_core.h"

#define LDPC_ENCP_SOLVE_PHASES 13
#define LDPC_ENCP_NUM_PHASES 14
/* workgroup size cap; measured per 1024 BG1 Zc=384 blocks: 512 -> 0.032 ms, 256 -> 0.024 ms, 128 -> 0.027 ms (more
 * workgroups per CU overlap the latency chains of the short phases) */
#define LDPC_ENCP_MAX_THREADS 256

struct ldpc_encp_lds {
  uint32_t *B, *X, *LB, *LX, *ET, *RP;
};

#if defined(__HIPCC__)
#define LDPC_ENCP_HOSTDEV __host__ __device__ __forceinline__
#else
#define LDPC_ENCP_HOSTDEV static inline
#endif
LDPC_ENCP_HOSTDEV int ldpc_encp_W(int Z) { return (Z + 31) >> 5; }
LDPC_ENCP_HOSTDEV int ldpc_encp_cw(int Z) { return 2 * ldpc_encp_W(Z) + 1; }
LDPC_ENCP_HOSTDEV int ldpc_encp_lds_words(int ncols, int kbf, int Z, int nrows, int nedges)
{
  const int W = ldpc_encp_W(Z), cw = ldpc_encp_cw(Z);
  return ncols * (W + 1) + (kbf + 4) * cw + 4 * (W + 1) + 4 * cw + nedges + nrows + 1;
}
LDPC_HD void ldpc_encp_carve(uint32_t *lds, ldpc_code_ptr_t code, ldpc_encp_lds &L)
{
  const int Z = code->Z, W = ldpc_encp_W(Z), cw = ldpc_encp_cw(Z);
  L.B = lds;
  L.X = L.B + code->ncols * (W + 1);
  L.LB = L.X + (code->kb_full + 4) * cw;
  L.LX = L.LB + 4 * (W + 1);
  L.ET = L.LX + 4 * cw;
  L.RP = L.ET + code->nedges;
}
/* threads that keep every phase busy: the extension rows give (nrows-4)*W items */
LDPC_ENCP_HOSTDEV int ldpc_encp_threads(int nrows, int Z)
{
  int n = ((nrows - 4) * ldpc_encp_W(Z) + 63) & ~63;
  return n < 64 ? 64 : (n > LDPC_ENCP_MAX_THREADS ? LDPC_ENCP_MAX_THREADS : n);
}

/* 32 bits of the bit string s starting at bit offset o */
LDPC_HD uint32_t ldpc_bits_at(const uint32_t *s, uint32_t o)
{
  const uint32_t lo = s[o >> 5], hi = s[(o >> 5) + 1];
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, o & 31u);
#else
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (o & 31u));
#endif
}
LDPC_HD uint32_t ldpc_encp_mask(int Z, int w) { return Z - 32 * w >= 32 ? 0xffffffffu : (1u << (Z - 32 * w)) - 1u; }

/* word wp of the periodic string of a masked column `base` (W words + one zero word) */
LDPC_HD uint32_t ldpc_encp_periodic_word(const uint32_t *base, int Z, int wp)
{
  uint32_t p = (uint32_t)(32 * wp) % (uint32_t)Z, v = 0;
  int filled = 0;
  while (filled < 32) {
    int n = Z - (int)p;
    if (n > 32 - filled)
      n = 32 - filled;
    uint32_t chunk = ldpc_bits_at(base, p);
    if (n < 32)
      chunk &= (1u << n) - 1u;
    v |= chunk << filled;
    filled += n;
    p = 0;
  }
  return v;
}
/* periodic strings X[first .. first+count) from the masked columns B[bfirst .. bfirst+count) */
LDPC_HD void ldpc_encp_extend(const uint32_t *B, uint32_t *X, int count, int Z, int tid, int nt)
{
  const int W = ldpc_encp_W(Z), cw = ldpc_encp_cw(Z);
  for (int i = tid; i < count * cw; i += nt) {
    const int c = i / cw, wp = i - c * cw;
    X[c * cw + wp] = ldpc_encp_periodic_word(B + c * (W + 1), Z, wp);
  }
}

LDPC_HD void ldpc_encp_phase(int phase, ldpc_code_ptr_t code, int Kb, const uint8_t *__restrict__ in, const ldpc_encp_lds &L,
                             uint8_t *__restrict__ out, int tid, int nt)
{
  const int Z = code->Z, kbf = code->kb_full, ncols = code->ncols, W = ldpc_encp_W(Z), cw = ldpc_encp_cw(Z), bs = W + 1;
  if (phase == 0) {
    /* information columns from the MSB-first input bytes; zero word behind every column */
    const int nin = (kbf * Z + 7) >> 3;
    for (int i = tid; i < kbf * W; i += nt) {
      const int c = i / W, w = i - c * W;
      const uint32_t b0 = (uint32_t)(c * Z + 32 * w), j0 = b0 >> 3;
      uint64_t v = 0;
      for (int q = 0; q < 5; q++)
        v = (v << 8) | ((int)j0 + q < nin ? in[j0 + q] : 0u);
      const uint32_t m = (uint32_t)(v >> (8 - (b0 & 7u)));
      L.B[c * bs + w] = __builtin_bitreverse32(m) & ldpc_encp_mask(Z, w);
    }
    for (int c = tid; c < ncols; c += nt)
      L.B[c * bs + W] = 0u;
    if (tid < 4)
      L.LB[tid * bs + W] = 0u;
    for (int e = tid; e < code->nedges; e += nt)
      L.ET[e] = ((uint32_t)code->e_col[e] << 16) | (code->e_info[e] & 0xffffu);
    for (int r = tid; r <= code->nrows; r += nt)
      L.RP[r] = (uint32_t)code->row_ptr[r];
  } else if (phase == 1) {
    ldpc_encp_extend(L.B, L.X, kbf, Z, tid, nt);
  } else if (phase == 2) {
    /* lambda_row = XOR over the information edges of core row `row` */
    for (int i = tid; i < 4 * W; i += nt) {
      const int row = i / W, w = i - row * W;
      const int e0 = (int)L.RP[row], e1 = (int)L.RP[row + 1];
      uint32_t acc = 0;
      for (int e = e0; e < e1; e++) {
        const uint32_t et = L.ET[e];
        const int c = (int)(et >> 16);
        if (c < Kb)
          acc ^= ldpc_bits_at(L.X + c * cw, (uint32_t)(32 * w) + (et & 0xffffu));
      }
      L.LB[row * bs + w] = acc & ldpc_encp_mask(Z, w);
    }
  } else if (phase == 3) {
    ldpc_encp_extend(L.LB, L.LX, 4, Z, tid, nt);
  } else if (phase == 4) {
    /* first core parity column: sum of the four core rows, sum[t] = p0[(t + s0) mod Z] */
    const uint32_t back = (uint32_t)((Z - code->enc_p0_shift % Z) % Z);
    for (int w = tid; w < W; w += nt) {
      uint32_t acc = 0;
      for (int row = 0; row < 4; row++)
        acc ^= ldpc_bits_at(L.LX + row * cw, (uint32_t)(32 * w) + back);
      L.B[kbf * bs + w] = acc & ldpc_encp_mask(Z, w);
    }
  } else if (phase == 5) {
    ldpc_encp_extend(L.B + kbf * bs, L.X + kbf * cw, 1, Z, tid, nt);
  } else if (phase <= 11) {
    /* the other three along the dual diagonal: x_unk[(t + us) mod Z] = lambda_row[t] ^ XOR_k x_k[(t + ks_k) mod Z] */
    const int step = (phase - 6) >> 1;
    const int unk = code->enc_unk[step];
    if (((phase - 6) & 1) == 0) {
      const int row = code->enc_row[step], nk = code->enc_nk[step];
      const uint32_t back = (uint32_t)((Z - code->enc_ushift[step] % Z) % Z);
      for (int w = tid; w < W; w += nt) {
        uint32_t acc = ldpc_bits_at(L.LX + row * cw, (uint32_t)(32 * w) + back);
        for (int k = 0; k < nk; k++)
          acc ^= ldpc_bits_at(L.X + (kbf + code->enc_kcol[step][k]) * cw,
                              (uint32_t)(32 * w) + (back + (uint32_t)code->enc_kshift[step][k]) % (uint32_t)Z);
        L.B[(kbf + unk) * bs + w] = acc & ldpc_encp_mask(Z, w);
      }
    } else {
      ldpc_encp_extend(L.B + (kbf + unk) * bs, L.X + (kbf + unk) * cw, 1, Z, tid, nt);
    }
  } else if (phase == 12) {
    /* extension parity: row r >= 4 closes on its own degree-1 column (last edge of the row, shift 0) */
    const int nitems = (code->nrows - 4) * W;
    for (int i = tid; i < nitems; i += nt) {
      const int rr = i / W, w = i - rr * W, row = 4 + rr;
      const int e0 = (int)L.RP[row], e1 = (int)L.RP[row + 1] - 1;
      uint32_t acc = 0;
      for (int e = e0; e < e1; e++) {
        const uint32_t et = L.ET[e];
        const int c = (int)(et >> 16);
        if (c < Kb || c >= kbf)
          acc ^= ldpc_bits_at(L.X + c * cw, (uint32_t)(32 * w) + (et & 0xffffu));
      }
      L.B[(L.ET[e1] >> 16) * bs + w] = acc & ldpc_encp_mask(Z, w);
    }
  } else {
    /* out[i] = bit i of the code word without its first two columns */
    const int N = (ncols - 2) * Z;
    if ((Z & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
      for (int i4 = tid; i4 < (N >> 2); i4 += nt) {
        const int i = 4 * i4, c = i / Z, t = i - c * Z;
        const uint32_t nib = (L.B[(c + 2) * bs + (t >> 5)] >> (t & 31)) & 0xfu;
        reinterpret_cast<uint32_t *>(out)[i4] = (nib * 0x00204081u) & 0x01010101u;
      }
    } else {
      for (int i = tid; i < N; i += nt) {
        const int c = i / Z, t = i - c * Z;
        out[i] = (uint8_t)((L.B[(c + 2) * bs + (t >> 5)] >> (t & 31)) & 1u);
      }
    }
  }
}
#endif
