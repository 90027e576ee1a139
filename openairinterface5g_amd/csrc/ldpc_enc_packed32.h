/*
 * ldpc_enc_packed32.h -- the bit-packed encoder's phases for lifting sizes that are multiples of 32 (Zc = 32 .. 384: the
 * codes of every large transport block), device code only.  Same contract, same solve order and the same LDS carve as
 * ldpc_enc_packed_core.h (reference nrLDPC_encoder/ldpc_encoder.c:44-252), which remains the path of the other lifting
 * sizes and the one the CPU emulation runs; both must give the oracle's code words for every code (tests/test_gpu_encoder.py).
 *
 * What Zc % 32 == 0 buys: a lifted column is W = Zc/32 whole words, so
 *   - nothing is masked, and the periodic string of a column (X[col][2W+1]) is the column's words twice + the first word
 *     again: whoever produces a word stores it in its three places -- the four "extend" phases and their barriers are gone;
 *   - the information columns are the input's dwords, byte order and bit order reversed (one v_bfrev + one v_perm);
 *   - the byte-per-bit output goes out as 16-byte stores, 16 code bits per lane;
 *   - no division anywhere: items are cut by multiplications with reciprocals of W computed once.
 * The solve costs 5 workgroup barriers (general path: 9) and about a third of the general path's VALU instructions:
 * profiles/r05/README.md (the encoder and the fused TX kernel are bound by instruction issue, not by memory).
 */
#ifndef LDPC_ENC_PACKED32_H
#define LDPC_ENC_PACKED32_H
#include "ldpc_enc_packed_core.h"

/* Per-thread phases as host/device functions, like ldpc_enc_packed_core.h: the kernels call them, and tests/emul runs the
 * same source one thread at a time on the CPU (the two cross-lane steps -- the DPP meeting of the lambda slices, the wave-level
 * ordering of the core parity steps -- are the callers' business: device wrappers below, plain loops in the emulation). */
#if defined(__HIP_DEVICE_COMPILE__)
LDPC_HD uint32_t ldpc_encp32_umulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
#else
LDPC_HD uint32_t ldpc_encp32_umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#endif
struct ldpc_encp32 {
  int Z, W, cw, bs, kbf, Kb;
  uint32_t winv; /* ceil(2^32 / W): i / W == umulhi(i, winv) for the item counts used here (i < 2^16) */
};
LDPC_HD ldpc_encp32 ldpc_encp32_make(ldpc_code_ptr_t code, int Kb)
{
  ldpc_encp32 g;
  g.Z = code->Z;
  g.W = g.Z >> 5;
  g.cw = 2 * g.W + 1;
  g.bs = g.W + 1;
  g.kbf = code->kb_full;
  g.Kb = Kb;
  g.winv = 0xffffffffu / (uint32_t)g.W + 1u;
  return g;
}
/* i / W for the item counts used here (W == 1: the reciprocal does not fit 32 bits) */
LDPC_HD uint32_t ldpc_encp32_div(const ldpc_encp32 &g, uint32_t i) { return g.W == 1 ? i : ldpc_encp32_umulhi(i, g.winv); }
LDPC_HD bool ldpc_encp32_applies(ldpc_code_ptr_t code) { return (code->Z & 31) == 0; }

/* word w of column col, produced: the result array and the three places of the column's periodic string */
LDPC_HD void ldpc_encp32_put(const ldpc_encp_lds &L, const ldpc_encp32 &g, int col, int w, uint32_t v)
{
  L.B[col * g.bs + w] = v;
  uint32_t *x = L.X + col * g.cw;
  x[w] = v;
  x[w + g.W] = v;
  if (w == 0)
    x[2 * g.W] = v;
}
/* 32 input bits as they lie in memory (MSB-first bytes, nrLDPC_encoder/ldpc_encoder.c:88-92) -> bit t of the column word */
LDPC_HD uint32_t ldpc_encp32_word(uint32_t in_dword) { return __builtin_bswap32(__builtin_bitreverse32(in_dword)); }

/* information column words from dword i of the block's input (i < kbf * W) */
LDPC_HD void ldpc_encp32_info(const ldpc_encp_lds &L, const ldpc_encp32 &g, uint32_t i, uint32_t in_dword)
{
  const uint32_t c = ldpc_encp32_div(g, i), w = i - c * (uint32_t)g.W;
  ldpc_encp32_put(L, g, (int)c, (int)w, ldpc_encp32_word(in_dword));
}

/* lambda_row = XOR over the information edges of core row `row` (rows 0..3), 32 lanes per item: an item (row, word) is
 * shared by four neighbouring lanes, each taking every fourth edge; their partial sums meet through two DPP exchanges
 * (the general path adds them up with LDS atomics into an array a phase of its own has to clear).
 * Lane i of 16 W: slice i & 3 of item i >> 2.  _partial = the lane's sum; _store = what the lane with slice 0 does with the
 * sum of its four. */
LDPC_HD uint32_t ldpc_encp32_lambda_partial(const ldpc_encp_lds &L, const ldpc_encp32 &g, int i)
{
  const uint32_t rw = (uint32_t)i >> 2, sl = (uint32_t)i & 3u;
  const uint32_t row = ldpc_encp32_div(g, rw), w = rw - row * (uint32_t)g.W;
  const int e0 = (int)L.RP[row], e1 = (int)L.RP[row + 1];
  uint32_t et[5];
#pragma unroll
  for (int k = 0; k < 5; k++) { /* 19 edges / 4 slices */
    const int e = e0 + (int)sl + 4 * k;
    et[k] = L.ET[e < e1 ? e : e1 - 1];
  }
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int e = e0 + (int)sl + 4 * k, c = (int)(et[k] >> 16);
    const uint32_t v = ldpc_bits_at(L.X + (c < g.kbf ? c : 0) * g.cw, 32u * w + (et[k] & 0xffffu));
    acc ^= (e < e1 && c < g.Kb) ? v : 0u;
  }
  for (int e = e0 + (int)sl + 20; e < e1; e += 4) { /* (not for NR's base graphs) */
    const uint32_t t = L.ET[e];
    if ((int)(t >> 16) < g.Kb)
      acc ^= ldpc_bits_at(L.X + (t >> 16) * g.cw, 32u * w + (t & 0xffffu));
  }
  return acc;
}
LDPC_HD void ldpc_encp32_lambda_store(const ldpc_encp_lds &L, const ldpc_encp32 &g, int i, uint32_t acc)
{
  const uint32_t rw = (uint32_t)i >> 2, row = ldpc_encp32_div(g, rw), w = rw - row * (uint32_t)g.W;
  uint32_t *x = L.LX + row * g.cw;
  x[w] = acc;
  x[w + g.W] = acc;
  if (w == 0)
    x[2 * g.W] = acc;
}
#if defined(__HIPCC__)
__device__ __forceinline__ void ldpc_encp32_lambda(const ldpc_encp_lds &L, const ldpc_encp32 &g, int tid, int nt)
{
  const int n = 16 * g.W; /* <= 192: one pass for any workgroup of >= 192 threads, the loop is for the smaller ones */
  for (int i0 = 0; i0 < n; i0 += nt) {
    const int i = i0 + tid;
    const bool live = i < n;
    uint32_t acc = ldpc_encp32_lambda_partial(L, g, live ? i : 0);
    acc ^= __shfl_xor(acc, 1);
    acc ^= __shfl_xor(acc, 2);
    if (live && (i & 3) == 0)
      ldpc_encp32_lambda_store(L, g, i, acc);
  }
}
#endif

/* the four core parity columns, ONE wavefront, two dependent steps: p0 from the four lambdas (lanes < W), then the other
 * three at once from their closed forms over the lambdas and p0 (lanes < 3 W; ldpc_graph.h enc_x_term) -- the dual diagonal
 * walked column by column is three steps where this is one.  A wave's LDS accesses execute in order, the fence only stops
 * the compiler from moving them across the step.  (General path: four steps + a periodic-extension step behind each.) */
LDPC_HD void ldpc_encp32_core_step_p0(ldpc_code_ptr_t code, const ldpc_encp_lds &L, const ldpc_encp32 &g, int lane)
{ /* first core parity column: sum of the four core rows, sum[t] = p0[(t + s0) mod Z] */
  const int Z = g.Z;
  int p0 = code->enc_p0_shift;
  const bool live = lane < g.W;
  const uint32_t o = 32u * (uint32_t)(live ? lane : 0);
  p0 = p0 >= Z ? p0 - (p0 / Z) * Z : p0;
  const uint32_t back = p0 == 0 ? 0u : (uint32_t)(Z - p0);
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < 4; r++)
    acc ^= ldpc_bits_at(L.LX + r * g.cw, o + back);
  if (live)
    ldpc_encp32_put(L, g, g.kbf, lane, acc);
}
LDPC_HD void ldpc_encp32_core_step_rest(ldpc_code_ptr_t code, const ldpc_encp_lds &L, const ldpc_encp32 &g, int lane)
{
  int unk[3];
  uint32_t term[3][8];
#pragma unroll
  for (int st = 0; st < 3; st++) {
    unk[st] = code->enc_unk[st];
#pragma unroll
    for (int m = 0; m < 8; m++)
      term[st][m] = code->enc_x_term[st][m];
  }
  const bool live = lane < 3 * g.W;
  const uint32_t st = ldpc_encp32_div(g, (uint32_t)(live ? lane : 0)), w = (uint32_t)(live ? lane : 0) - st * (uint32_t)g.W;
  uint32_t acc = 0;
#pragma unroll
  for (int m = 0; m < 8; m++) {
    const uint32_t t = st == 0 ? term[0][m] : (st == 1 ? term[1][m] : term[2][m]);
    const uint32_t src = t >> 16;
    const uint32_t *x = src < 4u ? L.LX + src * (uint32_t)g.cw : L.X + g.kbf * g.cw;
    const uint32_t v = ldpc_bits_at(t == 0xffffffffu ? L.LX : x, 32u * w + (t == 0xffffffffu ? 0u : (t & 0xffffu)));
    acc ^= t == 0xffffffffu ? 0u : v;
  }
  if (live)
    ldpc_encp32_put(L, g, g.kbf + (st == 0 ? unk[0] : (st == 1 ? unk[1] : unk[2])), (int)w, acc);
}
#if defined(__HIPCC__)
__device__ __forceinline__ void ldpc_encp32_core_parity_wave(ldpc_code_ptr_t code, const ldpc_encp_lds &L, const ldpc_encp32 &g, int lane)
{
  ldpc_encp32_core_step_p0(code, L, g, lane);
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
  ldpc_encp32_core_step_rest(code, L, g, lane);
}
#endif

/* extension parity: row r >= 4 closes on its own degree-1 column (last edge of the row, shift 0).  A lane takes one row and
 * TWO of its words (w and w + ceil(W/2)): the row's table entries are fetched and taken apart once for both -- the phase is the
 * largest of the encoder and instruction issue is what binds it (BG1 Zc = 384: 42 rows x 6 lanes = 252 of 256 threads, one pass). */
LDPC_HD void ldpc_encp32_extension(ldpc_code_ptr_t code, const ldpc_encp_lds &L, const ldpc_encp32 &g, int tid, int nt)
{
  const uint32_t tpr = (uint32_t)(g.W + 1) >> 1; /* lanes per row */
  const uint32_t tinv = 0xffffffffu / (tpr > 1 ? tpr : 2u) + 1u;
  const int nitems = (code->nrows - 4) * (int)tpr;
  for (int i = tid; i < nitems; i += nt) {
    const uint32_t rr = tpr == 1 ? (uint32_t)i : ldpc_encp32_umulhi((uint32_t)i, tinv), j = (uint32_t)i - rr * tpr;
    const uint32_t w0 = j, w1 = j + tpr;
    const bool two = w1 < (uint32_t)g.W;
    const int row = 4 + (int)rr;
    const int e0 = (int)L.RP[row], e1 = (int)L.RP[row + 1] - 1;
    uint32_t et[9];
#pragma unroll
    for (int k = 0; k < 9; k++) /* an extension row has at most 9 edges besides its own column: all table entries first */
      et[k] = L.ET[e0 + k < e1 ? e0 + k : e1];
    const uint32_t own = L.ET[e1];
    uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const int c = (int)(et[k] >> 16);
      const bool use = e0 + k < e1 && (c < g.Kb || c >= g.kbf);
      const uint32_t *x = L.X + (c < g.kbf + 4 ? c : 0) * g.cw;
      const uint32_t sh = et[k] & 0xffffu;
      const uint32_t v0 = ldpc_bits_at(x, 32u * w0 + sh), v1 = ldpc_bits_at(x, 32u * (two ? w1 : w0) + sh);
      acc0 ^= use ? v0 : 0u;
      acc1 ^= use ? v1 : 0u;
    }
    for (int e = e0 + 9; e < e1; e++) { /* (not for NR's base graphs) */
      const uint32_t t = L.ET[e];
      const int c = (int)(t >> 16);
      if (c < g.Kb || c >= g.kbf) {
        acc0 ^= ldpc_bits_at(L.X + c * g.cw, 32u * w0 + (t & 0xffffu));
        acc1 ^= ldpc_bits_at(L.X + c * g.cw, 32u * (two ? w1 : w0) + (t & 0xffffu));
      }
    }
    uint32_t *bo = L.B + (own >> 16) * g.bs;
    bo[w0] = acc0;
    if (two)
      bo[w1] = acc1;
  }
}

/* the reference's output (one bit per byte, the code word without its first two columns: ldpc_encoder.c:248-251) as
 * 16-byte stores: a lane takes 16 code bits -- half a word of B -- and expands every nibble into a dword with one
 * multiplication (bit j of the nibble -> byte j).  `out` 16-byte aligned. */
typedef uint32_t ldpc_u32x4_t __attribute__((ext_vector_type(4)));
/* columns [col0, col1) of the code word (col0 >= 2), by threads tid of nt */
LDPC_HD void ldpc_encp32_store_bytes(const ldpc_encp_lds &L, const ldpc_encp32 &g, uint8_t *__restrict__ out, int col0,
                                                        int col1, int tid, int nt)
{
  const uint32_t per_col = 2u * (uint32_t)g.W, n = (uint32_t)(col1 - 2) * per_col;
  const uint32_t inv = 0xffffffffu / per_col + 1u;
  ldpc_u32x4_t *__restrict__ o = reinterpret_cast<ldpc_u32x4_t *>(out);
  for (uint32_t i = (uint32_t)(col0 - 2) * per_col + (uint32_t)tid; i < n; i += (uint32_t)nt) {
    const uint32_t c = ldpc_encp32_umulhi(i, inv), h = i - c * per_col;
    const uint32_t bits = L.B[(c + 2u) * (uint32_t)g.bs + (h >> 1)] >> (16u * (h & 1u));
    ldpc_u32x4_t v;
    v.x = ((bits & 0xfu) * 0x00204081u) & 0x01010101u;
    v.y = (((bits >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
    v.z = (((bits >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
    v.w = (((bits >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
    o[i] = v;
  }
}
#endif
