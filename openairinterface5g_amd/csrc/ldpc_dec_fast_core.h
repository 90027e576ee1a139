/*
 * ldpc_dec_fast_core.h -- per-thread bodies of the "fast" flooding min-sum decoder kernel (gfx950).
 *
 * Same decoder semantics as ldpc_dec_core.h (reference nrLDPC_decoder.c:206-880, rules listed there);
 * different data layout and arithmetic:
 *   - a thread owns 4 consecutive lanes of a lifted row (check-node phase) or 4 consecutive bits of a
 *     column (bit-node phase); messages and APPs live in LDS as BIASED bytes b = v + 128 (= v ^ 0x80),
 *     four to a dword, so one ds_read_b32 / ds_write_b32 moves four messages;
 *   - the circularly shifted 4-byte window of a neighbour is fetched as two aligned dwords +
 *     v_alignbyte_b32; APP rows are stored twice back to back (no modulo on t + shift), message rows
 *     carry 4 wrap-around pad bytes;
 *   - arithmetic runs on packed 16-bit pairs (v_pk_*_i16/u16): biased bytes are zero-extended with
 *     v_perm_b32, the bias cancels in a' - r', sums of biased bytes cannot overflow 16 bits
 *     (<= 31 * 255), saturation is done once per result.
 * Bit-exactness argument for the rewritten min-sum (DESIGN.md "Kernel arithmetic"):
 *   q = clamp(app - r, -127, 127) only enters as |q| (capped at 127) and sign(q); |q| = min(|app - r|, 127)
 *   and sign(q) = sign(app - r), so the clamp is applied to the magnitude only.
 *
 * Host build (tests/emul) uses the same code with portable fall-backs for the gfx950 builtins.
 */
#ifndef LDPC_DEC_FAST_CORE_H
#define LDPC_DEC_FAST_CORE_H
#include "ldpc_dec_core.h"

#if !defined(__HIPCC__)
struct uint2 { uint32_t x, y; };
#endif
typedef short ldpc_v2i __attribute__((ext_vector_type(2)));
typedef unsigned short ldpc_v2u __attribute__((ext_vector_type(2)));

#if defined(__HIP_DEVICE_COMPILE__)
LDPC_HD uint32_t ldpc_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
LDPC_HD uint32_t ldpc_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
LDPC_HD uint32_t ldpc_umulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
#else
LDPC_HD uint32_t ldpc_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{ /* v_perm_b32: byte i of the result = byte sel[i] of {s0 (4..7), s1 (0..3)}; 0x0c -> 0x00 */
  const uint64_t src = ((uint64_t)s0 << 32) | s1;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xff;
    const uint32_t b = s <= 7 ? (uint32_t)((src >> (8 * s)) & 0xff) : (s == 0x0c ? 0u : 0xffu);
    r |= b << (8 * i);
  }
  return r;
}
LDPC_HD uint32_t ldpc_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh)
{
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3)));
}
LDPC_HD uint32_t ldpc_umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#endif

LDPC_HD ldpc_v2i ldpc_as_v2i(uint32_t x) { return __builtin_bit_cast(ldpc_v2i, x); }
LDPC_HD ldpc_v2u ldpc_as_v2u(uint32_t x) { return __builtin_bit_cast(ldpc_v2u, x); }
LDPC_HD uint32_t ldpc_u2u32(ldpc_v2u x) { return __builtin_bit_cast(uint32_t, x); }
LDPC_HD ldpc_v2u ldpc_splatu(unsigned v) { return (ldpc_v2u){(unsigned short)v, (unsigned short)v}; }
LDPC_HD ldpc_v2u ldpc_pminu(ldpc_v2u a, ldpc_v2u b) { return __builtin_elementwise_min(a, b); }
LDPC_HD ldpc_v2u ldpc_pmaxu(ldpc_v2u a, ldpc_v2u b) { return __builtin_elementwise_max(a, b); }
/* Minimum of three packed magnitude KEYS in one instruction.  A key is 0x8000 + m (m <= 0x3bff): as an f16 that is the
 * negative number whose magnitude grows with m, so the f16 MAXIMUM of keys is the key with the smallest m -- and gfx950
 * has a packed three-input f16 maximum that issues at the rate of v_pk_min_u16 (tools/ubench/pk_max3_f16.hip: every triple
 * of keys incl. the denormal range 0x8000 .. 0x80ff, the cap and the neutral element checked against the integer minimum;
 * f16 denormals are not flushed in HIP kernels).  LDPC_KEY_NEUTRAL = -65504, the largest finite magnitude. */
#define LDPC_KEY_NEUTRAL 0xfbffu
LDPC_HD ldpc_v2u ldpc_pmin3_keys(ldpc_v2u a, ldpc_v2u b, ldpc_v2u c)
{
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, b)),
      "v"(__builtin_bit_cast(uint32_t, c)));
  return __builtin_bit_cast(ldpc_v2u, r);
#else
  return __builtin_elementwise_min(__builtin_elementwise_min(a, b), c);
#endif
}
LDPC_HD uint32_t ldpc_as_u32(ldpc_v2i x) { return __builtin_bit_cast(uint32_t, x); }
LDPC_HD ldpc_v2i ldpc_splat(int v) { return (ldpc_v2i){(short)v, (short)v}; }
LDPC_HD ldpc_v2i ldpc_pmin(ldpc_v2i a, ldpc_v2i b) { return __builtin_elementwise_min(a, b); }
LDPC_HD ldpc_v2i ldpc_pmax(ldpc_v2i a, ldpc_v2i b) { return __builtin_elementwise_max(a, b); }
/* bytes 0,1 resp. 2,3 of w zero-extended into the two 16-bit halves */
LDPC_HD ldpc_v2i ldpc_unpack_lo(uint32_t w) { return ldpc_as_v2i(ldpc_perm(0u, w, 0x0c010c00u)); }
LDPC_HD ldpc_v2i ldpc_unpack_hi(uint32_t w) { return ldpc_as_v2i(ldpc_perm(0u, w, 0x0c030c02u)); }
/* low bytes of the four 16-bit halves of (lo, hi) -> one dword */
LDPC_HD uint32_t ldpc_pack4(ldpc_v2i lo, ldpc_v2i hi) { return ldpc_perm(ldpc_as_u32(hi), ldpc_as_u32(lo), 0x06040200u); }

/* LDS views used by the fast kernel (byte pointers into the workgroup's LDS) */
struct ldpc_fast_lds {
  uint8_t *base; /* start of the workgroup's LDS; f_etbl entries are byte offsets from here */
  uint8_t *r;    /* [nedges][Z+4]  biased check-to-bit messages (+4 wrap bytes) */
  uint8_t *app;  /* [ncore][2Z]    biased clamped APP, stored twice */
  uint8_t *ext;  /* [ncols-ncore][Z] biased channel LLR of the degree-1 columns */
  const uint32_t *etbl, *ctbl, *rowtbl, *coltbl;
  const uint8_t *gllr; /* the block's channel LLRs in global memory (true int8 bytes) */
  int ext_global;      /* extension-column LLRs are read from gllr instead of `ext` (ldpc_graph.c) */
};

/* LDS offsets held in the LDS-resident tables are ABSOLUTE LDS addresses on the device (the kernel adds the address
 * of its LDS block when it loads the tables), so a neighbour access is `ds_read(entry + lane offset)` with no base add;
 * the host emulation keeps them relative to `base`. */
#if defined(__HIP_DEVICE_COMPILE__)
LDPC_HD uint32_t ldpc_lds_addr(const uint8_t *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)p; }
LDPC_HD uint32_t ldpc_lds_ld32(const uint8_t *, uint32_t a) { return *(const __attribute__((address_space(3))) uint32_t *)a; }
#else
LDPC_HD uint32_t ldpc_lds_addr(const uint8_t *) { return 0u; }
LDPC_HD uint32_t ldpc_lds_ld32(const uint8_t *base, uint32_t a) { return *reinterpret_cast<const uint32_t *>(base + a); }
#endif
/* 4-byte window whose first byte is `sh & 3` bytes into the aligned dword at LDS address `a` */
LDPC_HD uint32_t ldpc_window_al(const uint8_t *base, uint32_t a, uint32_t sh)
{
  return ldpc_alignbyte(ldpc_lds_ld32(base, a + 4u), ldpc_lds_ld32(base, a), sh);
}
/* unaligned 4-byte window starting at LDS address `off` */
LDPC_HD uint32_t ldpc_window(const uint8_t *base, uint32_t off) { return ldpc_window_al(base, off & ~3u, off); }

/* What a check-node item reports: byte i of the result = 1 when the parity of the previous pass' hard decisions over lane
 * t + i's neighbours is odd (the check is not satisfied), 0 otherwise.  parw = xor of the biased APP windows of the row's core
 * columns (bit 7 of a byte = "not negative"); extl / exth = the degree-1 bit's biased sums llr' + r' in 16-bit halves (bit 8 of a
 * half = "not negative", cnProc.h:940) -- the bytes that hold those bits are gathered with one v_perm.  Parity of the hard
 * decisions = parity of the not-negative ones ^ (D & 1).  (Until round 6 the bodies returned a 4-bit mask: seven shifts and ors
 * to pack it and ten for the degree-1 bit, per item -- 4.6 % of the headline kernel's VALU instructions.) */
LDPC_HD uint32_t ldpc_fast_odd_lanes(uint32_t parw, bool ext, uint32_t extl, uint32_t exth, int D)
{
  uint32_t np = parw >> 7;
  if (ext)
    np ^= ldpc_perm(exth, extl, 0x07050301u);
  if (D & 1)
    np = ~np;
  return np & 0x01010101u;
}
/* The lanes of an item whose check counts, as byte flags: valid = pc_lo[row] - 4 j = how many of the item's four lanes lie in
 * front of the row's first excluded lane (ldpc_graph.c [F6]).  With Zc % 4 == 0 -- the only codes these bodies serve -- pc_lo is a
 * multiple of 4 (a class of nr rows drops lanes nr Z - 32 .. nr Z - 1; ldpc_build_code_desc checks it), so an item counts whole
 * or not at all: a compare and a select, where the general mask cost seven instructions per item. */
LDPC_HD uint32_t ldpc_fast_valid_lanes(int valid) { return valid > 0 ? 0x01010101u : 0u; }

/* One check-node item: lifted row with first edge e0, lanes t..t+3 (t = 4j).  D = row degree; EXT = the
 * last edge goes to the row's degree-1 column; KEEP = keep the per-edge magnitudes in registers between
 * the two sweeps (false for the degree-19 rows, which would spill).  Returns ldpc_fast_odd_lanes(): byte i = 1 when
 * the parity of the previous pass' hard decisions of lane t+i is odd.
 *
 * Arithmetic, per 16-bit half (two lanes per register, a' = app + 128, r' = r + 128 as stored):
 *   D1 = (0x8000 | a') - r' = 0x8000 + d          d = app - r in [-255, 255]; bit 15 of D1 = (d >= 0)
 *   D2 = 0x10000 - D1       = 0x8000 - d          (one 32-bit subtract: the low half always borrows)
 *   M  = max_u16(D1, D2)    = 0x8000 + |d|        biased magnitude; running two smallest m1 <= m2
 *   cap m1, m2 at 0x8000 + 127 once per check node (min(127, min_k |q_k|) = min_k min(127, |q_k|))
 *   o_k = m1 + m2 - min(M_k, m2)  (= m2 if edge k holds the minimum, else m1), bias folded into the sum
 *   sign_k = parity of the negative inputs among the other edges = bit 15 of (xor of all D1) ^ D1_k,
 *            flipped when D-1 is odd (bit 15 counts the NON-negative ones)
 * Only the packed min/max/shift/negate run on the half-rate packed-16 pipe; the rest are full-rate
 * 32-bit ALU ops that cannot carry between the halves by construction. */
/* D1 pairs of one edge (see above) from the LDS words; `first` additionally folds the edge into the syndrome
 * accumulators (only wanted once per edge). */
/* P1 = first pass of a block: every message is still 0, `rw` is not looked at (the caller has not loaded it and the
 * message array need not be initialised) */
template <bool IS_EXT, bool P1 = false>
LDPC_HD void ldpc_fast_cn_edge(const ldpc_fast_lds &L, uint32_t info, int t, uint32_t rw, bool first, uint32_t &dl, uint32_t &dh,
                               uint32_t &parw, uint32_t &extl, uint32_t &exth)
{
  uint32_t al, ah, rl, rh;
  if (IS_EXT) {
    const uint32_t lw = L.ext_global ? (*reinterpret_cast<const uint32_t *>(L.gllr + info + (uint32_t)t) ^ 0x80808080u)
                                     : ldpc_lds_ld32(L.base, info + (uint32_t)t);
    al = ldpc_perm(0x80808080u, lw, 0x05010400u);
    ah = ldpc_perm(0x80808080u, lw, 0x05030402u);
    if (first) { /* hard decision of the degree-1 bit: sat8(llr + r) < 0 <=> llr' + r' < 256 (cnProc.h:940) */
      extl = al + (P1 ? 0x00800080u : ldpc_perm(0u, rw, 0x0c010c00u));
      exth = ah + (P1 ? 0x00800080u : ldpc_perm(0u, rw, 0x0c030c02u));
    }
    rl = 0x00800080u; /* this edge's CN input is the channel LLR itself (mPass.h:306-388) */
    rh = 0x00800080u;
  } else {
    const uint32_t aw = ldpc_window(L.base, info + (uint32_t)t);
    if (first)
      parw ^= aw;
    al = ldpc_perm(0x80808080u, aw, 0x05010400u); /* (0x8000 | byte 0), (0x8000 | byte 1) */
    ah = ldpc_perm(0x80808080u, aw, 0x05030402u);
    rl = P1 ? 0x00800080u : ldpc_perm(0u, rw, 0x0c010c00u);
    rh = P1 ? 0x00800080u : ldpc_perm(0u, rw, 0x0c030c02u);
  }
  dl = al - rl;
  dh = ah - rh;
}

/* MODE 0: D1 and the magnitudes of every edge stay in registers between the two sweeps; 1: D1 only (magnitudes
 * recomputed); 2: nothing (the second sweep re-reads LDS and recomputes D1) -- for the degree-19 rows, whose 38+
 * live registers would otherwise spill at 16 waves per workgroup. */
/* boff_r / boff_a (several blocks per workgroup, ldpc_dec_fast_mblock.h): byte offset of the item's block inside a
 * message / extension-LLR row resp. inside an APP row; 0 in the one-block kernels, where they fold away. */
template <int D, bool EXT, int MODE, bool P1 = false>
LDPC_HD uint32_t ldpc_fast_cn(const ldpc_fast_lds &L, int e0, int j, int Z, int rstride, int boff_r = 0, int boff_a = 0)
{
  const int t = 4 * j + boff_r, ta = 4 * j + boff_a; /* t: position in message and extension rows; ta: in APP rows */
  constexpr bool KEEP = MODE == 0;
  uint32_t d_lo[MODE <= 1 ? D : 1], d_hi[MODE <= 1 ? D : 1], g_lo[KEEP ? D : 1], g_hi[KEEP ? D : 1];
  ldpc_v2u m1l = ldpc_splatu(0xffff), m2l = m1l, m1h = m1l, m2h = m1l;
  uint32_t sxl = 0, sxh = 0, parw = 0, extl = 0, exth = 0;
  uint8_t *rrow = L.r + e0 * rstride + t;
  uint8_t *rpad = rrow + (j == 0 ? Z : 0); /* lanes 0..3 are written twice: wrap-around copy behind the row */
#pragma unroll
  for (int k = 0; k < D; k++) {
    const uint32_t info = L.etbl[e0 + k];
    const uint32_t rw = P1 ? 0u : *reinterpret_cast<const uint32_t *>(rrow + k * rstride);
    uint32_t dl, dh;
    if (EXT && k == D - 1)
      ldpc_fast_cn_edge<true, P1>(L, info, t, rw, true, dl, dh, parw, extl, exth);
    else
      ldpc_fast_cn_edge<false, P1>(L, info, ta, rw, true, dl, dh, parw, extl, exth);
    if (MODE <= 1) {
      d_lo[k] = dl;
      d_hi[k] = dh;
    }
    const ldpc_v2u ml = ldpc_pmaxu(ldpc_as_v2u(dl), ldpc_as_v2u(0x00010000u - dl));
    const ldpc_v2u mh = ldpc_pmaxu(ldpc_as_v2u(dh), ldpc_as_v2u(0x00010000u - dh));
    if (KEEP) {
      g_lo[k] = ldpc_u2u32(ml);
      g_hi[k] = ldpc_u2u32(mh);
    }
    sxl ^= dl;
    sxh ^= dh;
    m2l = ldpc_pminu(m2l, ldpc_pmaxu(m1l, ml));
    m1l = ldpc_pminu(m1l, ml);
    m2h = ldpc_pminu(m2h, ldpc_pmaxu(m1h, mh));
    m1h = ldpc_pminu(m1h, mh);
  }
  const ldpc_v2u cap = ldpc_splatu(0x8000 + 127);
  m1l = ldpc_pminu(m1l, cap); m2l = ldpc_pminu(m2l, cap);
  m1h = ldpc_pminu(m1h, cap); m2h = ldpc_pminu(m2h, cap);
  const uint32_t sl = (ldpc_u2u32(m1l) - 0x80008000u) + ldpc_u2u32(m2l), sh = (ldpc_u2u32(m1h) - 0x80008000u) + ldpc_u2u32(m2h);
  if ((D - 1) & 1) {
    sxl ^= 0x80008000u;
    sxh ^= 0x80008000u;
  }
  /* Signs are applied to the four output bytes at once.  Bit 15 of a half of (sx ^ D1_k) says "negative"; the bytes
   * that hold those bits (1 and 3 of each register) are gathered with one v_perm, for the row's xor once and per edge
   * once.  With n = 0 / 1 per byte and a magnitude o <= 127, the biased byte is 128 + o = o ^ 0x80 for n = 0 and
   * 128 - o = (o ^ 0x7f) + 1 for n = 1, i.e. (o ^ (0x80 - n)) + n: three plain 32-bit ops for four lanes, no carry
   * between the bytes (the largest value is 0x80). */
  const uint32_t sx4 = ldpc_perm(sxh, sxl, 0x07050301u);
#pragma unroll
  for (int k = 0; k < D; k++) {
    uint32_t dl, dh;
    if (MODE <= 1) {
      dl = d_lo[k];
      dh = d_hi[k];
    } else {
      const uint32_t info = L.etbl[e0 + k];
      const uint32_t rw = P1 ? 0u : *reinterpret_cast<const uint32_t *>(rrow + k * rstride);
      if (EXT && k == D - 1)
        ldpc_fast_cn_edge<true, P1>(L, info, t, rw, false, dl, dh, parw, extl, exth);
      else
        ldpc_fast_cn_edge<false, P1>(L, info, ta, rw, false, dl, dh, parw, extl, exth);
    }
    const ldpc_v2u ml = KEEP ? ldpc_as_v2u(g_lo[k]) : ldpc_pmaxu(ldpc_as_v2u(dl), ldpc_as_v2u(0x00010000u - dl));
    const ldpc_v2u mh = KEEP ? ldpc_as_v2u(g_hi[k]) : ldpc_pmaxu(ldpc_as_v2u(dh), ldpc_as_v2u(0x00010000u - dh));
    const uint32_t ol = sl - ldpc_u2u32(ldpc_pminu(ml, m2l)), oh = sh - ldpc_u2u32(ldpc_pminu(mh, m2h)); /* magnitudes, 0..127 per half */
    const uint32_t o4 = ldpc_perm(oh, ol, 0x06040200u);
    const uint32_t n4 = ((sx4 ^ ldpc_perm(dh, dl, 0x07050301u)) >> 7) & 0x01010101u;
    const uint32_t w = (o4 ^ (0x80808080u - n4)) + n4;
    *reinterpret_cast<uint32_t *>(rrow + k * rstride) = w;
    *reinterpret_cast<uint32_t *>(rpad + k * rstride) = w;
  }
  /* per lane: number of "not negative" neighbours mod 2, from bit 7 of the biased APP bytes and bit 8 of
   * the extension sums; parity of the hard decisions = that ^ (D & 1) */
  return ldpc_fast_odd_lanes(parw, EXT, extl, exth, D);
}

/* The same check-node item with the "minimum of the OTHER edges" taken from prefixes and suffixes instead of through the
 * two smallest magnitudes, with three-input minima (ldpc_pmin3_keys).  Edges are taken in pairs (2i, 2i+1):
 *   forward   P_i = min(P_{i-1}, M_2i, M_2i+1), P_{-1} = cap (= 127 biased: every output is capped for free)
 *   backward  S_i = min(S_{i+1}, M_2i, M_2i+1)
 *   o_2i = min(P_{i-1}, M_2i+1, S_{i+1}),  o_2i+1 = min(P_{i-1}, M_2i, S_{i+1})
 * i.e. per pair of edges and 16-bit half FOUR packed ops (prefix, suffix, two outputs) + two for the magnitudes, where
 * the two-minima form needs five plus a subtract per EDGE and six more per row -- and the packed ops are the slow ones
 * (tools/ubench/valu_rate.hip).  The sign bytes of an edge are gathered once in the forward sweep (one v_perm, needed
 * anyway) and kept instead of the two D1 words.  Registers per edge: M per half + signs, P per half and pair.
 * Bit-identical outputs: the minimum over the other edges is the same number whichever way it is found. */
template <int D, bool EXT, bool P1 = false>
LDPC_HD uint32_t ldpc_fast_cn_ps(const ldpc_fast_lds &L, int e0, int j, int Z, int rstride, int boff_r = 0, int boff_a = 0)
{
  const int t = 4 * j + boff_r, ta = 4 * j + boff_a;
  constexpr int NP = D / 2; /* whole pairs; an odd D leaves edge D - 1 on its own */
  uint32_t m_lo[D], m_hi[D], p_lo[NP], p_hi[NP], s4[D];
  const ldpc_v2u cap = ldpc_splatu(0x8000 + 127);
  uint32_t sx4 = 0, parw = 0, extl = 0, exth = 0;
  uint8_t *rrow = L.r + e0 * rstride + t;
  uint8_t *rpad = rrow + (j == 0 ? Z : 0); /* lanes 0..3 are written twice: wrap-around copy behind the row */
#pragma unroll
  for (int k = 0; k < D; k++) {
    const uint32_t info = L.etbl[e0 + k];
    const uint32_t rw = P1 ? 0u : *reinterpret_cast<const uint32_t *>(rrow + k * rstride);
    uint32_t dl, dh;
    if (EXT && k == D - 1)
      ldpc_fast_cn_edge<true, P1>(L, info, t, rw, true, dl, dh, parw, extl, exth);
    else
      ldpc_fast_cn_edge<false, P1>(L, info, ta, rw, true, dl, dh, parw, extl, exth);
    const ldpc_v2u ml = ldpc_pmaxu(ldpc_as_v2u(dl), ldpc_as_v2u(0x00010000u - dl));
    const ldpc_v2u mh = ldpc_pmaxu(ldpc_as_v2u(dh), ldpc_as_v2u(0x00010000u - dh));
    m_lo[k] = ldpc_u2u32(ml);
    m_hi[k] = ldpc_u2u32(mh);
    s4[k] = ldpc_perm(dh, dl, 0x07050301u); /* the bytes that carry bit 15 of the four D1 halves */
    sx4 ^= s4[k];
    if ((k & 1) && k < D - 1) { /* P of pair k / 2: wanted by the pairs (and the single edge) behind it */
      const int i = k / 2;
      p_lo[i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_lo[i - 1]) : cap, ldpc_as_v2u(m_lo[k - 1]), ml));
      p_hi[i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_hi[i - 1]) : cap, ldpc_as_v2u(m_hi[k - 1]), mh));
    }
  }
  /* Signs.  The byte of s4 that stands for a lane is the high byte of 0x8000 + d, d in [-255, 255]: 0x80 (d >= 0) or 0x7f
   * (d < 0), nothing else.  The xor over the OTHER edges is therefore 0x80 / 0x7f (odd count) or 0x00 / 0xff (even count:
   * xor 0x80 in) per byte -- 0x7f = "an odd number of them is negative".  With x that byte, the biased output byte
   * 128 + o (x = 0x80) or 128 - o = (o ^ 0x7f) + 1 (x = 0x7f) is (o ^ x) + (x & 1): four plain 32-bit ops per edge for
   * four lanes, no carry between the bytes (the largest value is 0x80 + 0x7f resp. 0x7f + 1). */
  if (!((D - 1) & 1))
    sx4 ^= 0x80808080u;
  auto put = [&](int k, ldpc_v2u ol, ldpc_v2u oh) {
    const uint32_t o4 = ldpc_perm(ldpc_u2u32(oh), ldpc_u2u32(ol), 0x06040200u); /* low bytes: the magnitudes 0..127 */
    const uint32_t x4 = sx4 ^ s4[k];
    const uint32_t w = (o4 ^ x4) + (x4 & 0x01010101u);
    *reinterpret_cast<uint32_t *>(rrow + k * rstride) = w;
    *reinterpret_cast<uint32_t *>(rpad + k * rstride) = w;
  };
  ldpc_v2u sl = cap, sh = cap; /* suffix behind the pair in hand (set before its first use) */
  if (D & 1) {                 /* the single last edge: everything before it */
    put(D - 1, ldpc_as_v2u(p_lo[NP - 1]), ldpc_as_v2u(p_hi[NP - 1]));
    sl = ldpc_as_v2u(m_lo[D - 1]);
    sh = ldpc_as_v2u(m_hi[D - 1]);
  }
#pragma unroll
  for (int i = NP - 1; i >= 0; i--) {
    const int k0 = 2 * i, k1 = 2 * i + 1;
    const ldpc_v2u pvl = i ? ldpc_as_v2u(p_lo[i - 1]) : cap, pvh = i ? ldpc_as_v2u(p_hi[i - 1]) : cap;
    const ldpc_v2u a0l = ldpc_as_v2u(m_lo[k0]), a0h = ldpc_as_v2u(m_hi[k0]), a1l = ldpc_as_v2u(m_lo[k1]), a1h = ldpc_as_v2u(m_hi[k1]);
    const bool last = !(D & 1) && i == NP - 1; /* nothing behind this pair */
    put(k0, last ? ldpc_pminu(pvl, a1l) : ldpc_pmin3_keys(pvl, a1l, sl), last ? ldpc_pminu(pvh, a1h) : ldpc_pmin3_keys(pvh, a1h, sh));
    put(k1, last ? ldpc_pminu(pvl, a0l) : ldpc_pmin3_keys(pvl, a0l, sl), last ? ldpc_pminu(pvh, a0h) : ldpc_pmin3_keys(pvh, a0h, sh));
    if (i > 0) {
      sl = last ? ldpc_pminu(a0l, a1l) : ldpc_pmin3_keys(sl, a0l, a1l);
      sh = last ? ldpc_pminu(a0h, a1h) : ldpc_pmin3_keys(sh, a0h, a1h);
    }
  }
  return ldpc_fast_odd_lanes(parw, EXT, extl, exth, D);
}

/* TWO items of one degree group per thread (a double task, ldpc_graph.h f_cn_task), walked edge by edge TOGETHER: the same
 * arithmetic as ldpc_fast_cn_ps with an item index on everything, so that the two chains of table entry -> window ->
 * arithmetic -> store overlap inside one wave.  e0x / jx = first edge and 4-lane group of item x; returns item A's mask, item
 * B's in mask_b.  (An inactive second item is given item A's coordinates by the caller: it then stores the same bytes
 * twice.) */
template <int D, bool EXT, bool P1 = false>
LDPC_HD uint32_t ldpc_fast_cn_ps2(const ldpc_fast_lds &L, int e0a, int ja, int e0b, int jb, int Z, int rstride, uint32_t &mask_b)
{
  constexpr int N = 2, NP = D / 2;
  const int e0[N] = {e0a, e0b}, t[N] = {4 * ja, 4 * jb};
  uint32_t m_lo[N][D], m_hi[N][D], p_lo[N][NP > 0 ? NP : 1], p_hi[N][NP > 0 ? NP : 1], s4[N][D];
  const ldpc_v2u cap = ldpc_splatu(0x8000 + 127);
  uint32_t sx4[N] = {0, 0}, parw[N] = {0, 0}, extl[N] = {0, 0}, exth[N] = {0, 0};
  uint8_t *rrow[N], *rpad[N];
#pragma unroll
  for (int n = 0; n < N; n++) {
    rrow[n] = L.r + e0[n] * rstride + t[n];
    rpad[n] = rrow[n] + (t[n] == 0 ? Z : 0);
  }
#pragma unroll
  for (int k = 0; k < D; k++) {
    uint32_t info[N], rw[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
      info[n] = L.etbl[e0[n] + k];
      rw[n] = P1 ? 0u : *reinterpret_cast<const uint32_t *>(rrow[n] + k * rstride);
    }
#pragma unroll
    for (int n = 0; n < N; n++) {
      uint32_t dl, dh;
      if (EXT && k == D - 1)
        ldpc_fast_cn_edge<true, P1>(L, info[n], t[n], rw[n], true, dl, dh, parw[n], extl[n], exth[n]);
      else
        ldpc_fast_cn_edge<false, P1>(L, info[n], t[n], rw[n], true, dl, dh, parw[n], extl[n], exth[n]);
      const ldpc_v2u ml = ldpc_pmaxu(ldpc_as_v2u(dl), ldpc_as_v2u(0x00010000u - dl));
      const ldpc_v2u mh = ldpc_pmaxu(ldpc_as_v2u(dh), ldpc_as_v2u(0x00010000u - dh));
      m_lo[n][k] = ldpc_u2u32(ml);
      m_hi[n][k] = ldpc_u2u32(mh);
      s4[n][k] = ldpc_perm(dh, dl, 0x07050301u);
      sx4[n] ^= s4[n][k];
      if ((k & 1) && k < D - 1) {
        const int i = k / 2;
        p_lo[n][i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_lo[n][i - 1]) : cap, ldpc_as_v2u(m_lo[n][k - 1]), ml));
        p_hi[n][i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_hi[n][i - 1]) : cap, ldpc_as_v2u(m_hi[n][k - 1]), mh));
      }
    }
  }
  if (!((D - 1) & 1)) {
    sx4[0] ^= 0x80808080u;
    sx4[1] ^= 0x80808080u;
  }
  auto put = [&](int n, int k, ldpc_v2u ol, ldpc_v2u oh) {
    const uint32_t o4 = ldpc_perm(ldpc_u2u32(oh), ldpc_u2u32(ol), 0x06040200u);
    const uint32_t x4 = sx4[n] ^ s4[n][k];
    const uint32_t w = (o4 ^ x4) + (x4 & 0x01010101u);
    *reinterpret_cast<uint32_t *>(rrow[n] + k * rstride) = w;
    *reinterpret_cast<uint32_t *>(rpad[n] + k * rstride) = w;
  };
  ldpc_v2u sl[N] = {cap, cap}, sh[N] = {cap, cap};
  if (D & 1) {
#pragma unroll
    for (int n = 0; n < N; n++) {
      put(n, D - 1, NP ? ldpc_as_v2u(p_lo[n][NP - 1]) : cap, NP ? ldpc_as_v2u(p_hi[n][NP - 1]) : cap);
      sl[n] = ldpc_as_v2u(m_lo[n][D - 1]);
      sh[n] = ldpc_as_v2u(m_hi[n][D - 1]);
    }
  }
#pragma unroll
  for (int i = NP - 1; i >= 0; i--) {
    const int k0 = 2 * i, k1 = 2 * i + 1;
    const bool last = !(D & 1) && i == NP - 1;
#pragma unroll
    for (int n = 0; n < N; n++) {
      const ldpc_v2u pvl = i ? ldpc_as_v2u(p_lo[n][i - 1]) : cap, pvh = i ? ldpc_as_v2u(p_hi[n][i - 1]) : cap;
      const ldpc_v2u a0l = ldpc_as_v2u(m_lo[n][k0]), a0h = ldpc_as_v2u(m_hi[n][k0]), a1l = ldpc_as_v2u(m_lo[n][k1]), a1h = ldpc_as_v2u(m_hi[n][k1]);
      put(n, k0, last ? ldpc_pminu(pvl, a1l) : ldpc_pmin3_keys(pvl, a1l, sl[n]), last ? ldpc_pminu(pvh, a1h) : ldpc_pmin3_keys(pvh, a1h, sh[n]));
      put(n, k1, last ? ldpc_pminu(pvl, a0l) : ldpc_pmin3_keys(pvl, a0l, sl[n]), last ? ldpc_pminu(pvh, a0h) : ldpc_pmin3_keys(pvh, a0h, sh[n]));
      if (i > 0) {
        sl[n] = last ? ldpc_pminu(a0l, a1l) : ldpc_pmin3_keys(sl[n], a0l, a1l);
        sh[n] = last ? ldpc_pminu(a0h, a1h) : ldpc_pmin3_keys(sh[n], a0h, a1h);
      }
    }
  }
  uint32_t out[N];
#pragma unroll
  for (int n = 0; n < N; n++) {
    out[n] = ldpc_fast_odd_lanes(parw[n], EXT, extl[n], exth[n], D);
  }
  mask_b = out[1];
  return out[0];
}
/* dispatch of a double task (extension rows of degree 3 .. LDPC_F_CN_DOUBLE) */
#if defined(__HIP_DEVICE_COMPILE__) && defined(LDPC_CN2_NOINLINE)
#define LDPC_CN2_ATTR __device__ __attribute__((noinline))
#else
#define LDPC_CN2_ATTR LDPC_HD
#endif
template <bool P1 = false>
LDPC_CN2_ATTR uint32_t ldpc_fast_cn2_dispatch(int deg, const ldpc_fast_lds &L, int e0a, int ja, int e0b, int jb, int Z, int rstride, uint32_t &mask_b)
{
  switch (deg) {
    case 3: return ldpc_fast_cn_ps2<3, true, P1>(L, e0a, ja, e0b, jb, Z, rstride, mask_b);
    case 4: return ldpc_fast_cn_ps2<4, true, P1>(L, e0a, ja, e0b, jb, Z, rstride, mask_b);
    default: return ldpc_fast_cn_ps2<5, true, P1>(L, e0a, ja, e0b, jb, Z, rstride, mask_b);
  }
}

/* A degree-19 row item shared by TWO neighbouring lanes (lane parity = half): half 0 takes the row's edges 0..9, half 1
 * edges 10..18 -- ten edges in registers each (the one-lane version has to re-read LDS in its second sweep, MODE 2, or
 * spill), the partial minima / sign xor / parity word are swapped through a DPP move and merged, then every lane writes
 * its own edges.  The two smallest of a union = {min(m1a, m1b), min(max(m1a, m1b), min(m2a, m2b))}.  Device only (the
 * CPU emulation calls the one-lane body once per pair: same results). */
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t ldpc_swap_pair(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false); }
template <bool P1>
__device__ __forceinline__ uint32_t ldpc_fast_cn19_pair(const ldpc_fast_lds &L, int e0, int j, int Z, int rstride, int half)
{
  /* the pair network of ldpc_fast_cn_ps over this lane's ten slots; the partner lane's minimum over ALL its edges is the
   * suffix behind this lane's last pair, so "the other lane's edges" cost nothing per edge */
  constexpr int N = 10, NP = N / 2;
  const int t = 4 * j;
  const int ebase = e0 + (half ? 10 : 0);
  uint32_t m_lo[N], m_hi[N], p_lo[NP], p_hi[NP], s4[N];
  const ldpc_v2u cap = ldpc_splatu(0x8000 + 127);
  uint32_t sx4 = 0, parw = 0, extl = 0, exth = 0;
  uint8_t *rrow = L.r + ebase * rstride + t;
  uint8_t *rpad = rrow + (j == 0 ? Z : 0);
#pragma unroll
  for (int k = 0; k < N; k++) {
    const bool live = k < N - 1 || !half; /* half 1 has nine edges: its tenth slot is neutral */
    const int kk = live ? k : N - 2;      /* (address of a real edge; what is read there is not used) */
    const uint32_t info = L.etbl[ebase + kk];
    const uint32_t rw = P1 ? 0u : *reinterpret_cast<const uint32_t *>(rrow + kk * rstride);
    uint32_t dl, dh, pw = 0;
    ldpc_fast_cn_edge<false, P1>(L, info, t, rw, true, dl, dh, pw, extl, exth);
    ldpc_v2u ml = ldpc_pmaxu(ldpc_as_v2u(dl), ldpc_as_v2u(0x00010000u - dl));
    ldpc_v2u mh = ldpc_pmaxu(ldpc_as_v2u(dh), ldpc_as_v2u(0x00010000u - dh));
    uint32_t sg = ldpc_perm(dh, dl, 0x07050301u);
    if (!live) {
      ml = mh = ldpc_splatu(LDPC_KEY_NEUTRAL);
      sg = 0u;
      pw = 0u;
    }
    m_lo[k] = ldpc_u2u32(ml); m_hi[k] = ldpc_u2u32(mh);
    s4[k] = sg;
    sx4 ^= sg;
    parw ^= pw;
    if (k & 1) {
      const int i = k / 2;
      p_lo[i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_lo[i - 1]) : cap, ldpc_as_v2u(m_lo[k - 1]), ml));
      p_hi[i] = ldpc_u2u32(ldpc_pmin3_keys(i ? ldpc_as_v2u(p_hi[i - 1]) : cap, ldpc_as_v2u(m_hi[k - 1]), mh));
    }
  }
  /* the partner lane's minimum over all its edges starts this lane's suffix; signs and parity are merged */
  ldpc_v2u sl = ldpc_as_v2u(ldpc_swap_pair(p_lo[NP - 1])), sh = ldpc_as_v2u(ldpc_swap_pair(p_hi[NP - 1]));
  sx4 ^= ldpc_swap_pair(sx4);
  parw ^= ldpc_swap_pair(parw);
  sx4 ^= 0x80808080u; /* 19 - 1 others: an even count (see ldpc_fast_cn_ps; a neutral slot xor-ed zeros into sx4) */
#pragma unroll
  for (int i = NP - 1; i >= 0; i--) {
    const ldpc_v2u pvl = i ? ldpc_as_v2u(p_lo[i - 1]) : cap, pvh = i ? ldpc_as_v2u(p_hi[i - 1]) : cap;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int k = 2 * i + q, ko = 2 * i + 1 - q;
      const bool live = k < N - 1 || !half;
      const ldpc_v2u ol = ldpc_pmin3_keys(pvl, ldpc_as_v2u(m_lo[ko]), sl), oh = ldpc_pmin3_keys(pvh, ldpc_as_v2u(m_hi[ko]), sh);
      const uint32_t o4 = ldpc_perm(ldpc_u2u32(oh), ldpc_u2u32(ol), 0x06040200u);
      const uint32_t x4 = sx4 ^ s4[k];
      const uint32_t w = (o4 ^ x4) + (x4 & 0x01010101u);
      if (live) {
        *reinterpret_cast<uint32_t *>(rrow + k * rstride) = w;
        *reinterpret_cast<uint32_t *>(rpad + k * rstride) = w;
      }
    }
    if (i > 0) {
      sl = ldpc_pmin3_keys(sl, ldpc_as_v2u(m_lo[2 * i]), ldpc_as_v2u(m_lo[2 * i + 1]));
      sh = ldpc_pmin3_keys(sh, ldpc_as_v2u(m_hi[2 * i]), ldpc_as_v2u(m_hi[2 * i + 1]));
    }
  }
  return ldpc_fast_odd_lanes(parw, false, 0u, 0u, 19);
}
#endif

/* Parity check of one check-node item on its own (no message update): the same byte flags ldpc_fast_cn returns, from the
 * APP signs of the row's core columns and, for an extension row, sat8(llr + r) of its degree-1 column (cnProc.h:940).
 * Used by the latency path (ldpc_dec_fast_block.h, eager check): ~6 VALU per edge instead of a check-node update. */
LDPC_HD uint32_t ldpc_fast_pc(const ldpc_fast_lds &L, int D, int ext, int e0, int j, int rstride)
{
  const int t = 4 * j;
  uint32_t parw = 0;
  const int ncore_edges = ext ? D - 1 : D;
  /* the sweep is a chain table entry -> window per edge: four edges in flight */
  int k = 0;
  for (; k + 4 <= ncore_edges; k += 4) {
    uint32_t info[4], lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
      info[i] = L.etbl[e0 + k + i] + (uint32_t)t;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      lo[i] = ldpc_lds_ld32(L.base, info[i] & ~3u);
      hi[i] = ldpc_lds_ld32(L.base, (info[i] & ~3u) + 4u);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      parw ^= ldpc_alignbyte(hi[i], lo[i], info[i]);
  }
  for (; k < ncore_edges; k++)
    parw ^= ldpc_window(L.base, L.etbl[e0 + k] + (uint32_t)t);
  uint32_t extl = 0, exth = 0;
  if (ext) {
    const uint32_t info = L.etbl[e0 + D - 1];
    const uint32_t lw = L.ext_global ? (*reinterpret_cast<const uint32_t *>(L.gllr + info + (uint32_t)t) ^ 0x80808080u)
                                     : ldpc_lds_ld32(L.base, info + (uint32_t)t);
    const uint32_t rw = *reinterpret_cast<const uint32_t *>(L.r + (e0 + D - 1) * rstride + t);
    extl = ldpc_perm(0x80808080u, lw, 0x05010400u) + ldpc_perm(0u, rw, 0x0c010c00u);
    exth = ldpc_perm(0x80808080u, lw, 0x05030402u) + ldpc_perm(0u, rw, 0x0c030c02u);
  }
  return ldpc_fast_odd_lanes(parw, ext != 0, extl, exth, D);
}

#ifndef LDPC_F_MODE_D19
#define LDPC_F_MODE_D19 2
#endif
/* dispatch on the task's (wave-uniform) degree */
template <bool P1 = false>
LDPC_HD uint32_t ldpc_fast_cn_dispatch(int deg, int ext, const ldpc_fast_lds &L, int e0, int j, int Z, int rstride, int boff_r = 0,
                                       int boff_a = 0)
{
  if (!ext) {
    switch (deg) {
      case 19: return ldpc_fast_cn<19, false, LDPC_F_MODE_D19, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
      case 10: return ldpc_fast_cn_ps<10, false, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
      default: return ldpc_fast_cn_ps<8, false, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    }
  }
  switch (deg) {
    case 3: return ldpc_fast_cn_ps<3, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 4: return ldpc_fast_cn_ps<4, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 5: return ldpc_fast_cn_ps<5, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 6: return ldpc_fast_cn_ps<6, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 7: return ldpc_fast_cn_ps<7, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 8: return ldpc_fast_cn_ps<8, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    case 9: return ldpc_fast_cn_ps<9, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
    default: return ldpc_fast_cn_ps<10, true, P1>(L, e0, j, Z, rstride, boff_r, boff_a);
  }
}

/* One bit-node item: core column c, bits u..u+3 (u = 4j): APP = clamp_s8(llr + sum r) (bnProc.h:136-160), in two steps.
 * ldpc_fast_bn_gather: acc_e / acc_o += the biased message bytes of the column's edges, as packed 16-bit sums of lanes
 * (0,2) and (1,3).  colrec = f_coltbl entry of the column; maxdeg = wave-uniform loop bound >= the column's degree.
 * The gather is a chain table entry -> address -> window per edge; four edges are kept in flight.  Table entry =
 * {x = Z - shift, y = LDS address of the message row - (x & 3)}: u and Z are multiples of 4, so the window's byte
 * phase is x & 3 for every item and y + p is the aligned dword that holds its first byte.  Columns with fewer than
 * maxdeg edges are padded with entries that point at a row of zero bytes (contribution 0): no predication. */
/* first = index of the column's first list entry.  The entries of the NEXT four edges are requested before the current four
 * windows are waited for: one exposed LDS round trip per step instead of two (BG1 Zc = 384: -2..4 % on the whole kernel,
 * profiles/r03/decoder_ab*.txt; gathering all entries and then all windows at once, unrolled per loop bound, was slower). */
LDPC_HD void ldpc_fast_bn_gather_from(const ldpc_fast_lds &L, int first, int maxdeg, int j, int Z, int boff_r, uint32_t &acc_e,
                                      uint32_t &acc_o)
{
  const int u = 4 * j;
  const uint2 *tbl = reinterpret_cast<const uint2 *>(L.ctbl) + first;
  int k = 0;
  uint2 ce[4];
  if (maxdeg >= 4) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      ce[i] = tbl[i];
  }
  for (; k + 4 <= maxdeg; k += 4) {
    uint32_t lo[4], hi[4], ph[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t q = (uint32_t)u + ce[i].x;                      /* u + Z - shift in [1, 2Z) */
      const uint32_t p = q < q - (uint32_t)Z ? q : q - (uint32_t)Z;  /* mod Z: q - Z wraps to a huge value when q < Z */
      const uint32_t a = ce[i].y + p + (uint32_t)boff_r;
      lo[i] = ldpc_lds_ld32(L.base, a);
      hi[i] = ldpc_lds_ld32(L.base, a + 4u);
      ph[i] = q;
    }
    if (k + 8 <= maxdeg) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        ce[i] = tbl[k + 4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t w = ldpc_alignbyte(hi[i], lo[i], ph[i]);
      acc_e += w & 0x00ff00ffu;
      acc_o += (w >> 8) & 0x00ff00ffu;
    }
  }
  for (; k < maxdeg; k++) {
    const uint2 c1 = tbl[k];
    const uint32_t q = (uint32_t)u + c1.x;
    const uint32_t p = q < q - (uint32_t)Z ? q : q - (uint32_t)Z;
    const uint32_t v = ldpc_window_al(L.base, c1.y + p + (uint32_t)boff_r, q);
    acc_e += v & 0x00ff00ffu;
    acc_o += (v >> 8) & 0x00ff00ffu;
  }
}
LDPC_HD void ldpc_fast_bn_gather(const ldpc_fast_lds &L, uint32_t colrec, int maxdeg, int j, int Z, int boff_r, uint32_t &acc_e,
                                 uint32_t &acc_o)
{
  ldpc_fast_bn_gather_from(L, (int)(colrec >> 16), maxdeg, j, Z, boff_r, acc_e, acc_o);
}
/* ldpc_fast_bn_finish: the sums of ALL `deg` edges of column c (each byte biased by 128) + the channel LLRs (true int8
 * bytes) -> clamped APP, stored twice */
LDPC_HD void ldpc_fast_bn_finish(const ldpc_fast_lds &L, int c, int deg, int j, int Z, int astride, uint32_t llr_word, uint32_t acc_e,
                                 uint32_t acc_o, int boff_a)
{
  const int u = 4 * j;
  const uint32_t lw = llr_word ^ 0x80808080u;
  acc_e += lw & 0x00ff00ffu;
  acc_o += (lw >> 8) & 0x00ff00ffu;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LDPC_NO_SAT_PK)
  /* biased APP byte = clamp(app, -128, 127) + 128 = clamp(sum - deg * 128, 0, 255): v_sat_pk_u8_i16 clamps both 16-bit
   * halves to 0 .. 255 and packs them into two bytes -- one subtract, one pack per register, one v_perm to interleave */
  const ldpc_v2i bias = ldpc_splat(deg * 128);
  uint32_t se, so;
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(se) : "v"(ldpc_as_u32(ldpc_as_v2i(acc_e) - bias)));
  asm("v_sat_pk_u8_i16 %0, %1" : "=v"(so) : "v"(ldpc_as_u32(ldpc_as_v2i(acc_o) - bias)));
  /* bytes: lane0 = se.b0, lane1 = so.b0, lane2 = se.b1, lane3 = so.b1 */
  const uint32_t w = ldpc_perm(so, se, 0x05010400u);
#else
  const ldpc_v2i bias = ldpc_splat((deg + 1) * 128), lo = ldpc_splat(-128), hi = ldpc_splat(127), b128 = ldpc_splat(128);
  const ldpc_v2i ve = ldpc_pmin(ldpc_pmax(ldpc_as_v2i(acc_e) - bias, lo), hi) + b128;
  const ldpc_v2i vo = ldpc_pmin(ldpc_pmax(ldpc_as_v2i(acc_o) - bias, lo), hi) + b128;
  /* bytes: lane0 = ve.lo, lane1 = vo.lo, lane2 = ve.hi, lane3 = vo.hi */
  const uint32_t w = ldpc_perm(ldpc_as_u32(vo), ldpc_as_u32(ve), 0x06020400u);
#endif
  uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride + u + boff_a);
  dst[0] = w;
  *reinterpret_cast<uint32_t *>(L.app + c * astride + u + boff_a + Z) = w;
}
LDPC_HD void ldpc_fast_bn(const ldpc_fast_lds &L, uint32_t colrec, int maxdeg, int j, int Z, int astride,
                          uint32_t llr_word, int boff_r = 0, int boff_a = 0)
{
  uint32_t acc_e = 0, acc_o = 0;
  ldpc_fast_bn_gather(L, colrec, maxdeg, j, Z, boff_r, acc_e, acc_o);
  ldpc_fast_bn_finish(L, (int)(colrec & 0xffu), (int)((colrec >> 8) & 0xffu), j, Z, astride, llr_word, acc_e, acc_o, boff_a);
}

/* G bit-node items of one thread, all of short columns (ldpc_graph.h f_bn_ticket), walked edge by edge TOGETHER: G table
 * entries, then G windows in flight per step instead of one chain after the other.  md = the ticket's loop bound (every
 * short column's list is padded at least that far); live[g] = 0: slot g is a filler (some valid item, result dropped). */
template <int G>
LDPC_HD void ldpc_fast_bn_multi(const ldpc_fast_lds &L, const uint32_t (&rec)[G], const int (&jj)[G], const uint32_t (&lw)[G],
                                const bool (&live)[G], int md, int Z, int astride)
{
  /* sums: ae = bytes 0 and 2 of the windows (mask, add), at = the windows >> 8 whole (shift, add) = S1 + 2^8 S2 + 2^16 S3 with
   * S_i the sum of byte i over the edges (< 2^13: no overflow); the high half of ae is S2, so the odd lanes' packed sums
   * S1 + 2^16 S3 = at - (S2 << 8), once per item: four ops per edge instead of five (this path is bound by VALU issue) */
  uint32_t ae[G], at[G];
  const uint2 *tbl[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    ae[g] = at[g] = 0u;
    tbl[g] = reinterpret_cast<const uint2 *>(L.ctbl) + (int)(rec[g] >> 16);
  }
  for (int k = 0; k < md; k++) {
    uint2 ce[G];
    uint32_t lo[G], hi[G], ph[G];
#pragma unroll
    for (int g = 0; g < G; g++)
      ce[g] = tbl[g][k];
#pragma unroll
    for (int g = 0; g < G; g++) {
      const uint32_t q = 4u * (uint32_t)jj[g] + ce[g].x;
      const uint32_t p = q < q - (uint32_t)Z ? q : q - (uint32_t)Z;
      const uint32_t a = ce[g].y + p;
      lo[g] = ldpc_lds_ld32(L.base, a);
      hi[g] = ldpc_lds_ld32(L.base, a + 4u);
      ph[g] = q;
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      const uint32_t w = ldpc_alignbyte(hi[g], lo[g], ph[g]);
      ae[g] += w & 0x00ff00ffu;
      at[g] += w >> 8;
    }
  }
#pragma unroll
  for (int g = 0; g < G; g++)
    if (live[g])
      ldpc_fast_bn_finish(L, (int)(rec[g] & 0xffu), (int)((rec[g] >> 8) & 0xffu), jj[g], Z, astride, lw[g], ae[g],
                          at[g] - ((ae[g] >> 16) << 8), 0);
}

/* hard decision of code bit `b` (< ncore*Z) from the biased APP store */
LDPC_HD int ldpc_fast_hd(const ldpc_fast_lds &L, int b, int Z, uint32_t zmagic, int astride)
{
  const int c = (int)ldpc_umulhi((uint32_t)b, zmagic), u = b - c * Z;
  return L.app[c * astride + u] < 128;
}
#endif
