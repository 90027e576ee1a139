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

#if defined(__HIP_DEVICE_COMPILE__)
LDPC_HD void ldpc_encp_xor_into(uint32_t *p, uint32_t v) { atomicXor(p, v); }
#else
LDPC_HD void ldpc_encp_xor_into(uint32_t *p, uint32_t v) { *p ^= v; } /* (CPU emulation: one thread at a time) */
#endif
#define LDPC_ENCP_LAMBDA_SLICES 4 /* lanes that share one (core row, word) item of phase 2 */

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
    for (int i = tid; i < 4 * bs; i += nt)
      L.LB[i] = 0u; /* phase 2 accumulates into it */
    for (int e = tid; e < code->nedges; e += nt)
      L.ET[e] = code->enc_et[e];
    for (int r = tid; r <= code->nrows; r += nt)
      L.RP[r] = (uint32_t)code->row_ptr[r];
  } else if (phase == 1) {
    ldpc_encp_extend(L.B, L.X, kbf, Z, tid, nt);
  } else if (phase == 2) {
    /* lambda_row = XOR over the information edges of core row `row`.  A core row has up to 19 edges: an item (row, word)
     * is shared by LDPC_ENCP_LAMBDA_SLICES lanes, each taking every SLICES-th edge (table reads and windows of a lane are
     * independent loads), and the partial sums are XOR-ed into LB, which phase 0 cleared -- 48 lanes walking 19 dependent
     * steps each were the longest phase of the kernel (profiles/r03/tb_tx_phases.txt). */
    for (int i = tid; i < 4 * W * LDPC_ENCP_LAMBDA_SLICES; i += nt) {
      const int sl = i % LDPC_ENCP_LAMBDA_SLICES, rw = i / LDPC_ENCP_LAMBDA_SLICES, row = rw / W, w = rw - row * W;
      const int e0 = (int)L.RP[row], e1 = (int)L.RP[row + 1];
      uint32_t et[5];
#pragma unroll
      for (int k = 0; k < 5; k++) { /* 19 edges / 4 slices */
        const int e = e0 + sl + k * LDPC_ENCP_LAMBDA_SLICES;
        et[k] = L.ET[e < e1 ? e : e1 - 1];
      }
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int e = e0 + sl + k * LDPC_ENCP_LAMBDA_SLICES, c = (int)(et[k] >> 16);
        const uint32_t v = ldpc_bits_at(L.X + (c < kbf ? c : 0) * cw, (uint32_t)(32 * w) + (et[k] & 0xffffu));
        acc ^= (e < e1 && c < Kb) ? v : 0u;
      }
      for (int e = e0 + sl + 5 * LDPC_ENCP_LAMBDA_SLICES; e < e1; e += LDPC_ENCP_LAMBDA_SLICES) { /* (not for NR's base graphs) */
        const uint32_t t = L.ET[e];
        if ((int)(t >> 16) < Kb)
          acc ^= ldpc_bits_at(L.X + (t >> 16) * cw, (uint32_t)(32 * w) + (t & 0xffffu));
      }
      acc &= ldpc_encp_mask(Z, w);
      if (acc)
        ldpc_encp_xor_into(&L.LB[row * bs + w], acc);
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
      /* an extension row has at most 9 edges besides its own column: all table entries first, then all windows (two LDS
       * round trips per item instead of two per edge) */
      uint32_t et[9];
#pragma unroll
      for (int k = 0; k < 9; k++)
        et[k] = L.ET[e0 + k < e1 ? e0 + k : e1];
      const uint32_t own = L.ET[e1];
      uint32_t acc = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int c = (int)(et[k] >> 16);
        const bool use = e0 + k < e1 && (c < Kb || c >= kbf);
        const uint32_t v = ldpc_bits_at(L.X + (c < kbf + 4 ? c : 0) * cw, (uint32_t)(32 * w) + (et[k] & 0xffffu));
        acc ^= use ? v : 0u;
      }
      for (int e = e0 + 9; e < e1; e++) { /* (not for NR's base graphs) */
        const uint32_t t = L.ET[e];
        const int c = (int)(t >> 16);
        if (c < Kb || c >= kbf)
          acc ^= ldpc_bits_at(L.X + c * cw, (uint32_t)(32 * w) + (t & 0xffffu));
      }
      L.B[(own >> 16) * bs + w] = acc & ldpc_encp_mask(Z, w);
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
#if defined(__HIP_DEVICE_COMPILE__)
/* Phases 4 .. 11 (the four core parity columns along the dual diagonal) walked by ONE wavefront without workgroup
 * barriers: at most W <= 12 (solve) resp. cw <= 25 (periodic extension) lanes have work in any of the eight steps, a
 * step is a few dependent LDS reads, and a wave's LDS accesses execute in order -- the fences only stop the compiler from
 * moving them across a step.  All the solve parameters of the descriptor are requested up front (one scalar-load round
 * trip; read where they are used they were a chain of a dozen).  Called by every lane of one wave; the caller's
 * workgroup barrier follows.  Same results as ldpc_encp_phase(4 .. 11) (the CPU emulation runs those). */
__device__ __forceinline__ void ldpc_encp_core_parity_wave(ldpc_code_ptr_t code, const ldpc_encp_lds &L, int lane)
{
  const int Z = code->Z, kbf = code->kb_full, W = ldpc_encp_W(Z), cw = ldpc_encp_cw(Z), bs = W + 1;
  const uint32_t zmagic = 0xffffffffu / (uint32_t)Z + 1u;
  int p0 = code->enc_p0_shift, row[3], unk[3], us[3], nk[3], kc[3][4], ks[3][4];
#pragma unroll
  for (int st = 0; st < 3; st++) {
    row[st] = code->enc_row[st]; unk[st] = code->enc_unk[st]; us[st] = code->enc_ushift[st]; nk[st] = code->enc_nk[st];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      kc[st][k] = code->enc_kcol[st][k];
      ks[st][k] = code->enc_kshift[st][k];
    }
  }
  /* periodic string of the masked column `base` -> X, cw words: lane wp < cw builds word wp */
  auto extend = [&](const uint32_t *base, uint32_t *X) {
    if (lane < cw) {
      const uint32_t o = 32u * (uint32_t)lane;
      uint32_t p = o - __umulhi(o, zmagic) * (uint32_t)Z, v = 0; /* o mod Z */
      int filled = 0;
      while (filled < 32) {
        int n = Z - (int)p;
        n = n > 32 - filled ? 32 - filled : n;
        uint32_t chunk = ldpc_bits_at(base, p);
        if (n < 32)
          chunk &= (1u << n) - 1u;
        v |= chunk << filled;
        filled += n;
        p = 0;
      }
      X[lane] = v;
    }
  };
  auto step_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  { /* first core parity column: sum of the four core rows, sum[t] = p0[(t + s0) mod Z] */
    p0 = p0 >= Z ? p0 - (p0 / Z) * Z : p0;
    const uint32_t back = p0 == 0 ? 0u : (uint32_t)(Z - p0);
    if (lane < W) {
      uint32_t acc = 0;
#pragma unroll
      for (int r = 0; r < 4; r++)
        acc ^= ldpc_bits_at(L.LX + r * cw, 32u * (uint32_t)lane + back);
      L.B[kbf * bs + lane] = acc & ldpc_encp_mask(Z, lane);
    }
    step_sync();
    extend(L.B + kbf * bs, L.X + kbf * cw);
    step_sync();
  }
#pragma unroll
  for (int st = 0; st < 3; st++) {
    int u = us[st];
    u = u >= Z ? u - (u / Z) * Z : u;
    const uint32_t back = u == 0 ? 0u : (uint32_t)(Z - u);
    if (lane < W) {
      uint32_t acc = ldpc_bits_at(L.LX + row[st] * cw, 32u * (uint32_t)lane + back);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t sh = back + (uint32_t)ks[st][k];
        sh = sh >= (uint32_t)Z ? sh - (uint32_t)Z : sh;
        sh = sh >= (uint32_t)Z ? sh - (sh / (uint32_t)Z) * (uint32_t)Z : sh; /* (shifts are stored reduced: never taken) */
        const uint32_t v = ldpc_bits_at(L.X + (kbf + (k < nk[st] ? kc[st][k] : 0)) * cw, 32u * (uint32_t)lane + sh);
        acc ^= k < nk[st] ? v : 0u;
      }
      L.B[(kbf + unk[st]) * bs + lane] = acc & ldpc_encp_mask(Z, lane);
    }
    step_sync();
    extend(L.B + (kbf + unk[st]) * bs, L.X + (kbf + unk[st]) * cw);
    step_sync();
  }
}
#elif defined(__HIPCC__)
/* (host pass of the kernels' translation units: never called) */
__device__ inline void ldpc_encp_core_parity_wave(ldpc_code_ptr_t, const ldpc_encp_lds &, int) {}
#endif
#endif
