/*
 * ldpc_dec_core.h -- per-thread bodies of the generic flooding min-sum decoder kernel.
 *
 * Compiles both as HIP device code (included by ldpc_decoder.hip) and as plain host C++ (included by
 * tests/emul/ldpc_emul.cpp, which walks the threads of one workgroup sequentially, phase by phase, so
 * the indexing can be checked against the oracle on a machine without a GPU).  No phase has an
 * intra-phase dependency between threads, which is what makes both executions equivalent.
 *
 * Decoder semantics (what must be reproduced bit for bit) -- reference
 * openair1/PHY/CODING/nrLDPC_decoder/nrLDPC_decoder.c:206-880, generic forms of the kernels in
 * nrLDPC_cnProc.h / nrLDPC_bnProc.h / nrLDPC_mPass.h:
 *   CN update   r_j = prod_{k!=j} sign(q_k) * min(127, min_{k!=j} |q_k|), sign(0) = 0   (cnProc.h:81-118)
 *   BN update   APP = clamp_s8(llr + sum_k r_k)  for columns with >= 2 checks           (bnProc.h:136-160)
 *               q_k = clamp_s8(APP - r_k); the edge into a degree-1 column keeps q = llr (bnProc.h:325, mPass.h:306-388)
 *   early stop  parity of sign(sat8(q_k + r_k)) over each check, some lanes exempt        (cnProc.h:887-1946)
 *
 * Formulation used here (one message array instead of the reference's four buffers):
 *   state  r[e][t]  int8  check-to-bit message of edge e, lifted-row lane t
 *          app[c][u] int8 clamped a-posteriori LLR of core column c, bit u
 *          llr[c][u] int8 channel LLR
 *   check-node phase of pass p: q = clamp(app - r_old) is formed on the fly (so q is never stored),
 *     r_new overwrites r_old; in the same sweep the syndrome of the PREVIOUS pass' hard decisions is
 *     accumulated (sign(sat8(q_new + r)) == sign(app) for core columns -- see DESIGN.md), which is the
 *     reference's cnProcPc of pass p-1 at no extra memory traffic.
 *   bit-node phase: app = clamp(llr + sum r_new) gathered per column.
 *   q is clamped to [-127,127] instead of [-128,127]: |q| is capped at 127 and only sign(q) is used
 *   besides, so -128 and -127 are indistinguishable downstream (cnProc.h:110-114).
 */
#ifndef LDPC_DEC_CORE_H
#define LDPC_DEC_CORE_H
#include <stdint.h>
#include "ldpc_graph.h"

#if defined(__HIPCC__)
#define LDPC_HD __device__ __forceinline__
/* descriptor tables are read through the constant address space: wave-uniform lookups become s_load */
#define LDPC_CONST_AS __attribute__((address_space(4)))
#define LDPC_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define LDPC_HD static inline
#define LDPC_CONST_AS
#define LDPC_UNIFORM(x) (x)
#endif
typedef const ldpc_code_desc_t LDPC_CONST_AS *ldpc_code_ptr_t;
typedef const uint32_t LDPC_CONST_AS *ldpc_u32c_ptr_t;

LDPC_HD int ldpc_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
LDPC_HD int ldpc_mini(int a, int b) { return a < b ? a : b; }
LDPC_HD int ldpc_maxi(int a, int b) { return a > b ? a : b; }
LDPC_HD int ldpc_absi(int a) { return a < 0 ? -a : a; }

/* One lifted check node (row `row`, lane t) of degree D.  EXT: the row's last edge goes into its
 * degree-1 extension-parity column (true for base-graph rows >= 4, shift 0 by construction of 38.212;
 * checked by tools/gen_bg_tables.py).  Returns the parity (0/1) of the previous pass' hard decisions. */
template <int D, bool EXT>
LDPC_HD int ldpc_cn_update(ldpc_u32c_ptr_t e_info, int e0, int t, int Z, int8_t *__restrict__ r,
                           const int8_t *__restrict__ app, const int8_t *__restrict__ llr)
{
  int q[D];
  int sx = 0, m1 = 255, m2 = 255, par = 0;
#pragma unroll
  for (int k = 0; k < D; k++) {
    const uint32_t info = e_info[e0 + k];
    const int s = (int)(info & 0xffffu), cz = (int)(info >> 16);
    int idx = t + s;
    idx = idx >= Z ? idx - Z : idx;
    const int rv = r[(e0 + k) * Z + t];
    int qq;
    if (EXT && k == D - 1) {
      const int l = llr[cz + idx];
      par ^= (ldpc_clampi(l + rv, -128, 127) < 0);
      qq = ldpc_maxi(l, -127);
    } else {
      const int a = app[cz + idx];
      par ^= (a < 0);
      qq = ldpc_clampi(a - rv, -127, 127);
    }
    q[k] = qq;
    const int mag = ldpc_absi(qq);
    sx ^= qq;
    m2 = ldpc_mini(m2, ldpc_maxi(m1, mag));
    m1 = ldpc_mini(m1, mag);
  }
  const int msum = m1 + m2;
#pragma unroll
  for (int k = 0; k < D; k++) {
    const int mag = ldpc_absi(q[k]);
    /* min over the other edges: m2 if this edge holds the minimum, else m1 */
    const int o = msum - ldpc_mini(mag, m2);
    const int neg = (sx ^ q[k]) < 0;
    r[(e0 + k) * Z + t] = (int8_t)(neg ? -o : o);
  }
  return par;
}

/* dispatch on the (wave-uniform) row degree: core rows 0..3 have degree 19 (BG1) or 8/10 (BG2) and no
 * extension column; rows >= 4 have degree 3..10 (BG1) or 3..6 (BG2) including their extension column */
LDPC_HD int ldpc_cn_row(ldpc_code_ptr_t code, int row, int t, int8_t *__restrict__ r,
                        const int8_t *__restrict__ app, const int8_t *__restrict__ llr)
{
  const int Z = code->Z, e0 = code->row_ptr[row], d = code->row_deg[row];
  ldpc_u32c_ptr_t ei = code->e_info;
  if (row < 4) {
    switch (d) {
      case 19: return ldpc_cn_update<19, false>(ei, e0, t, Z, r, app, llr);
      case 10: return ldpc_cn_update<10, false>(ei, e0, t, Z, r, app, llr);
      default: return ldpc_cn_update<8, false>(ei, e0, t, Z, r, app, llr);
    }
  }
  switch (d) {
    case 3: return ldpc_cn_update<3, true>(ei, e0, t, Z, r, app, llr);
    case 4: return ldpc_cn_update<4, true>(ei, e0, t, Z, r, app, llr);
    case 5: return ldpc_cn_update<5, true>(ei, e0, t, Z, r, app, llr);
    case 6: return ldpc_cn_update<6, true>(ei, e0, t, Z, r, app, llr);
    case 7: return ldpc_cn_update<7, true>(ei, e0, t, Z, r, app, llr);
    case 8: return ldpc_cn_update<8, true>(ei, e0, t, Z, r, app, llr);
    case 9: return ldpc_cn_update<9, true>(ei, e0, t, Z, r, app, llr);
    default: return ldpc_cn_update<10, true>(ei, e0, t, Z, r, app, llr);
  }
}

/* One bit node (core column c, bit u): app = clamp_s8(llr + sum of the incoming check messages). */
LDPC_HD void ldpc_bn_update(ldpc_code_ptr_t code, int c, int u, const int8_t *__restrict__ r,
                            int8_t *__restrict__ app, const int8_t *__restrict__ llr)
{
  const int Z = code->Z;
  int sum = llr[c * Z + u];
  const int j1 = code->col_ptr[c + 1];
  for (int j = code->col_ptr[c]; j < j1; j++) {
    const uint32_t ce = code->col_edge[j];
    const int s = (int)(ce & 0xffffu), e = (int)(ce >> 16);
    int idx = u - s;
    idx = idx < 0 ? idx + Z : idx;
    sum += r[e * Z + idx];
  }
  app[c * Z + u] = (int8_t)ldpc_clampi(sum, -128, 127);
}

/* 32 hard decisions app[32w .. 32w+32) -> one output word, byte b = bits 8b..8b+7 MSB first
 * (nrLDPC_bnProc.h:1353-1380 llr2bitPacked); bits at or beyond n_valid read as 0. */
LDPC_HD uint32_t ldpc_pack_word(const int8_t *__restrict__ app, int w, int n_valid)
{
  uint32_t word = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int b = 32 * w + i;
    const uint32_t bit = (b < n_valid) ? (uint32_t)(app[b] < 0) : 0u;
    word |= bit << ((i & ~7) + 7 - (i & 7));
  }
  return word;
}
#endif
