/*
 * tb_rx_core.h -- per-thread bodies of the UL-SCH front step of one code segment: de-interleaving + rate de-matching with
 * HARQ soft combining + the int16 -> int8 pack in front of the decoder (reference openair1/PHY/CODING/nr_rate_matching.c:
 * 310-388 nr_deinterleaving_ldpc, :507-603 nr_rate_matching_ldpc_rx; NR_TRANSPORT/nr_ulsch_decoding.c:153-210).
 *
 * One workgroup per segment, the segment's received contributions transposed through an LDS image of the part of the
 * circular buffer this transmission touches, so that both sides move whole cache lines:
 *   phase Z  request every thread's first symbols (their latency runs under the rest of this phase), zero the image and
 *            the decoder input's punctured columns;
 *   phase A  a thread per modulation symbol jj reads the symbol's Qm LLRs f[jj*Qm .. +Qm) in ONE load -- from device
 *            memory, or over the link from the caller's page-locked array -- and drops each at its soft-buffer position
 *            in LDS (e_lds[pos(i*E/Qm + jj)]); one lap of the circular buffer at a time, so that every position receives
 *            at most one value per lap (plain read-modify-write, no atomics, no division);
 *   phase B  a thread per 8 consecutive soft-buffer positions: w (16-byte load / store) + the lap sums from LDS, then the
 *            saturated int8 decoder input (8-byte store) with the punctured zeros and +127 fillers.
 * HBM traffic = E int16 in, Ncb int16 in (unless first round) and out, num_llr int8 out: the compulsory bytes.
 *
 * Used by tb_rx_dematch_kernel (tb_chain.hip: one launch for all segments of a call) and as the prologue of the fused
 * segment kernel (tb_rx_fused.hip: de-matching, decoding and reassembly of a segment by one workgroup).  Compiles as HIP
 * device code and as plain host C++ (tests/emul walks the threads of a workgroup phase by phase); no phase has a
 * dependency between threads except through the barriers between the phases.
 */
#ifndef TB_RX_CORE_H
#define TB_RX_CORE_H
#include <stdint.h>
#include "tb_jobs.h"

#if defined(__HIPCC__)
#define TB_RX_HD __device__ __forceinline__
#else
#define TB_RX_HD static inline
#endif

typedef uint32_t tb_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t tb_u32x2 __attribute__((ext_vector_type(2)));

template <int QM> struct tb_sym; /* the Qm LLRs of one modulation symbol as one load (4-byte aligned source) */
template <> struct tb_sym<2> { uint32_t w[1]; };
template <> struct tb_sym<4> { uint32_t w[2]; };
template <> struct tb_sym<6> { uint32_t w[3]; };
template <> struct tb_sym<8> { uint32_t w[4]; };

/* what the phases need of a segment's job (wave-uniform: built from scalar loads of the job record) */
struct tb_rx_geom {
  uint32_t E, Ncb, Foffset, Fin, V, rank0, clear;
  uint32_t twoZ, num_llr, Klo, Khi;
  uint32_t np;      /* soft-buffer positions the decoder reads */
  uint32_t n;       /* positions this call looks at: max(Ncb, np) */
  uint32_t p_align; /* first position of the LDS image (multiple of 8) */
  uint32_t span;    /* LDS slots in use, a multiple of 8 */
};
template <class JobPtr> TB_RX_HD tb_rx_geom tb_rx_geometry(JobPtr j)
{
  tb_rx_geom g;
  g.E = j->E; g.Ncb = j->Ncb; g.Foffset = j->Foffset; g.Fin = j->Fin; g.V = j->V; g.rank0 = j->rank0; g.clear = j->clear;
  g.twoZ = 2 * j->Z; g.num_llr = j->num_llr; g.Klo = j->K - j->F; g.Khi = j->K;
  g.np = g.num_llr > g.twoZ ? g.num_llr - g.twoZ : 0;
  g.n = g.Ncb > g.np ? g.Ncb : g.np;
  const uint32_t p_base = g.rank0 < g.Foffset ? g.rank0 : g.rank0 + g.Fin;
  g.p_align = p_base & ~7u;
  g.span = tb_rx_lds_elems(g.E, g.Fin, g.Ncb);
  return g;
}

/* LDS slot of soft-buffer position p.  Positions are visited in circular order starting at p_base = pos(rank0); the slot
 * is the circular distance from p_align = p_base rounded down to a multiple of 8, so that 8 consecutive positions that
 * start at a multiple of 8 sit in one aligned 16-byte LDS word (as long as Ncb % 8 == 0 across the wrap).  A segment
 * touches min(Ncb, E + Fin) + 8 slots at most (tb_rx_lds_elems); the slots are zeroed first, so a slot that receives
 * nothing (filler positions, the tail of the last lap) simply contributes 0 -- no coverage logic on the way out. */
TB_RX_HD uint32_t tb_rx_slot(uint32_t p, uint32_t p_align, uint32_t Ncb) { return p >= p_align ? p - p_align : p + Ncb - p_align; }

/* ---- phase Z ---------------------------------------------------------------------------------------------------------- */
TB_RX_HD void tb_rx_phase_zero(const tb_rx_geom &g, int16_t *e_lds, int8_t *__restrict__ l, uint32_t tid, uint32_t nt)
{
  for (uint32_t i = tid; i < g.span / 8; i += nt)
    reinterpret_cast<tb_u32x4 *>(e_lds)[i] = (tb_u32x4){0u, 0u, 0u, 0u};
  for (uint32_t i = tid; i < g.twoZ && i < g.num_llr; i += nt)
    l[i] = 0;                                                       /* punctured columns (nr_ulsch_decoding.c:198) */
}

/* ---- phase A, one lap ------------------------------------------------------------------------------------------------- */
TB_RX_HD uint32_t tb_rx_laps(const tb_rx_geom &g) { return (g.E + g.V - 1) / g.V; }
#ifndef TB_RX_U
#define TB_RX_U 4
#endif
/* The first TB_RX_U symbols of every thread (jj = tid + u * nt), requested BEFORE the LDS image is cleared: their trip to
 * HBM -- or over the link to the caller's page-locked array -- runs under the clearing and its barrier, and all of them are
 * in flight together (unconditional loads from a clamped index: a load inside `if (jj < EQ)` is not moved across the
 * branch by the compiler).  A 512-thread workgroup covers a 64QAM segment's 1575 symbols with them; what is left is walked
 * by the loop of tb_rx_phase_scatter_lap. */
struct tb_rx_ahead { uint32_t w[TB_RX_U > 0 ? TB_RX_U : 1][4]; }; /* up to Qm = 8 values per symbol; the same storage whatever Qm,
                                                                    so that only the loads and the scatter are compiled per Qm */
template <int QM>
TB_RX_HD void tb_rx_phase_load_first(const tb_rx_geom &g, const int16_t *__restrict__ f, uint32_t tid, uint32_t nt, tb_rx_ahead &first)
{
  const uint32_t EQ = g.E / QM;
  const bool vec = (reinterpret_cast<uintptr_t>(f) & 3) == 0;
#pragma unroll
  for (int u = 0; u < TB_RX_U; u++) {
    uint32_t jj = tid + (uint32_t)u * nt;
    jj = jj < EQ ? jj : (EQ ? EQ - 1 : 0);
    if (vec) {
      const tb_sym<QM> sy = *reinterpret_cast<const tb_sym<QM> *>(f + (size_t)jj * QM);
#pragma unroll
      for (int i = 0; i < QM / 2; i++)
        first.w[u][i] = sy.w[i];
    } else {
#pragma unroll
      for (int i = 0; i < QM; i += 2)
        first.w[u][i >> 1] = (uint32_t)(uint16_t)f[(size_t)jj * QM + i] | ((uint32_t)(uint16_t)f[(size_t)jj * QM + i + 1] << 16);
    }
  }
}
/* one symbol's Qm values to their slots (lap `lap` of `nlaps`: only the values whose k falls into this lap) */
/* ONE: the transmission is a single lap of the circular buffer (E <= V: everything but repetition) -- every value is a plain
 * store, and the code is straight-line: the general form below is a branch per value (is k in this lap? first lap or add?), and
 * with two dozen values per thread the branches, not the arithmetic, were the 3.9 us this phase took (profiles/r06) */
template <int QM, bool ONE = false>
TB_RX_HD void tb_rx_scatter_symbol(const tb_rx_geom &g, const tb_sym<QM> &sy, uint32_t jj, int16_t *e_lds, uint32_t lap, uint32_t nlaps)
{
  const uint32_t V = g.V, rank0 = g.rank0, Foffset = g.Foffset, Fin = g.Fin, p_align = g.p_align, Ncb = g.Ncb, EQ = g.E / QM;
  if (ONE) {
#pragma unroll
    for (int i = 0; i < QM; i++) {
      const int16_t v = (int16_t)(sy.w[i >> 1] >> (16 * (i & 1)));
      uint32_t r = rank0 + (uint32_t)i * EQ + jj;
      r = r >= V ? r - V : r;
      e_lds[tb_rx_slot(r < Foffset ? r : r + Fin, p_align, Ncb)] = v;
    }
    return;
  }
  const uint32_t k_lo = lap * V, k_hi = k_lo + V; /* this lap's k range; (rank0 + k - k_lo) < 2V: one conditional subtract */
#pragma unroll
  for (int i = 0; i < QM; i++) {
    const int16_t v = (int16_t)(sy.w[i >> 1] >> (16 * (i & 1)));
    const uint32_t k = (uint32_t)i * EQ + jj;
    if (nlaps == 1 || (k >= k_lo && k < k_hi)) {
      uint32_t r = rank0 + (k - k_lo);
      r = r >= V ? r - V : r;
      const uint32_t q = tb_rx_slot(r < Foffset ? r : r + Fin, p_align, Ncb);
      e_lds[q] = lap == 0 ? v : (int16_t)(e_lds[q] + v);
    }
  }
}
template <int QM, bool ONE = false>
TB_RX_HD void tb_rx_phase_scatter_lap(const tb_rx_geom &g, const int16_t *__restrict__ f, int16_t *e_lds, uint32_t lap, uint32_t nlaps,
                                      uint32_t tid, uint32_t nt, const tb_rx_ahead &first)
{
  const uint32_t EQ = g.E / QM;
  const bool vec = (reinterpret_cast<uintptr_t>(f) & 3) == 0;
#pragma unroll
  for (int u = 0; u < TB_RX_U; u++) { /* the symbols loaded ahead */
    const uint32_t jj = tid + (uint32_t)u * nt;
    if (jj < EQ) {
      tb_sym<QM> sy;
#pragma unroll
      for (int i = 0; i < QM / 2; i++)
        sy.w[i] = first.w[u][i];
      tb_rx_scatter_symbol<QM, ONE>(g, sy, jj, e_lds, lap, nlaps);
    }
  }
  for (uint32_t jj0 = tid + TB_RX_U * nt; jj0 < EQ; jj0 += 2 * nt) { /* the rest, two symbols per step */
    tb_sym<QM> sy[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      uint32_t jj = jj0 + (uint32_t)u * nt;
      jj = jj < EQ ? jj : EQ - 1;
      if (vec) {
        sy[u] = *reinterpret_cast<const tb_sym<QM> *>(f + (size_t)jj * QM);
      } else {
#pragma unroll
        for (int i = 0; i < QM; i += 2)
          sy[u].w[i >> 1] = (uint32_t)(uint16_t)f[(size_t)jj * QM + i] | ((uint32_t)(uint16_t)f[(size_t)jj * QM + i + 1] << 16);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const uint32_t jj = jj0 + (uint32_t)u * nt;
      if (jj < EQ)
        tb_rx_scatter_symbol<QM, ONE>(g, sy[u], jj, e_lds, lap, nlaps);
    }
  }
}

/* ---- phase B ----------------------------------------------------------------------------------------------------------- */
/* 8 positions per thread and step, lanes side by side: one 16-byte load / store of w, one aligned 16-byte LDS read, one
 * 8-byte store of the decoder input.  w[p] = (first round ? 0 : w[p]) + received (nr_rate_matching.c:554-603; beyond
 * Ncb the reference's buffer is calloc'ed and never written: 0 on a first round); a position that received nothing
 * keeps its value, so the store is skipped unless something changes. */
TB_RX_HD void tb_rx_phase_stream(const tb_rx_geom &g, const int16_t *e_lds, int16_t *__restrict__ w, int8_t *__restrict__ l, uint32_t tid,
                                 uint32_t nt)
{
  const uint32_t Ncb = g.Ncb, p_align = g.p_align, span = g.span, twoZ = g.twoZ, Klo = g.Klo, Khi = g.Khi, np = g.np, n = g.n,
                 clear = g.clear;
  /* what position p received in this call (all laps), 0 if nothing */
  auto received = [&](uint32_t p) -> int16_t {
    const uint32_t q = tb_rx_slot(p, p_align, Ncb);
    return (p < Ncb && q < span) ? e_lds[q] : (int16_t)0;
  };
  auto pack = [&](uint32_t p, int16_t acc) -> int8_t {             /* nr_ulsch_decoding.c:200-210 */
    const uint32_t i = p + twoZ;
    const int v = (i >= Klo && i < Khi) ? 127 : (int)acc;
    return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((reinterpret_cast<uintptr_t>(l + twoZ) & 7) == 0);
  const uint32_t n8 = vec ? (n & ~7u) : 0;
  const uint32_t fill_lo = Klo > twoZ ? Klo - twoZ : 0, fill_hi = Khi > twoZ ? Khi - twoZ : 0; /* decoder-input fillers in p */
  typedef union { tb_u32x4 q; int16_t h[8]; uint32_t u[4]; } chunk_t;
  auto finish = [&](uint32_t p0, const chunk_t &old) { /* the chunk at p0, its old soft values already loaded */
    chunk_t e, acc;
    const uint32_t q0 = tb_rx_slot(p0, p_align, Ncb);
    if (p0 + 8 <= Ncb && (p0 >= p_align || (Ncb & 7u) == 0)) {       /* the chunk is one aligned LDS word (or outside the span) */
      e.q = q0 < span ? *reinterpret_cast<const tb_u32x4 *>(e_lds + q0) : (tb_u32x4){0u, 0u, 0u, 0u};
    } else {
#pragma unroll
      for (int t = 0; t < 8; t++)
        e.h[t] = received(p0 + t);
    }
    bool any = clear != 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {                                    /* int16 wrapping add, two lanes per op */
      acc.u[t] = ((old.u[t] & 0x7fff7fffu) + (e.u[t] & 0x7fff7fffu)) ^ ((old.u[t] ^ e.u[t]) & 0x80008000u);
      any |= e.u[t] != 0;
    }
    if (any)
      *reinterpret_cast<tb_u32x4 *>(w + p0) = acc.q;
    if (p0 < np) {
      union { tb_u32x2 q; int8_t b[8]; } lo;
      if (p0 + 8 <= fill_lo || p0 >= fill_hi) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const int v = acc.h[t];
          lo.b[t] = (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; t++)
          lo.b[t] = pack(p0 + t, acc.h[t]);
      }
      if (p0 + 8 <= np)
        *reinterpret_cast<tb_u32x2 *>(l + twoZ + p0) = lo.q;
      else
        for (int t = 0; t < 8; t++)
          if (p0 + t < np)
            l[twoZ + p0 + t] = lo.b[t];
    }
  };
  /* gfx9 counts loads AND stores in one in-order counter (vmcnt): waiting for a load waits for every store issued before
   * it.  The loop over a thread's chunks therefore must not wait for a load that was issued behind the previous chunk's
   * stores -- each iteration would then take a store's round trip to HBM, twelve times per thread (that was the kernel's
   * critical path: profiles/r04/ab_dematch_store_waits.txt).  First transmissions load nothing, and their loop has no
   * load in it (the compiler cannot drop the wait from a loop in which the load is conditional); retransmissions request
   * the NEXT chunk's soft values before the current chunk's stores go out. */
  const chunk_t zero = {(tb_u32x4){0u, 0u, 0u, 0u}};
  if (clear) {
    for (uint32_t p0 = 8 * tid; p0 < n8; p0 += 8 * nt) {
      finish(p0, zero);
#if defined(__HIP_DEVICE_COMPILE__) && defined(TB_RX_STORE_WINDOW)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TB_RX_STORE_WINDOW) : "memory"); /* tuning knob: stores a thread keeps in flight */
#endif
    }
  } else {
    uint32_t p0 = 8 * tid;
    chunk_t cur = zero;
    if (p0 < n8)
      cur.q = *reinterpret_cast<const tb_u32x4 *>(w + p0);
    for (; p0 < n8; p0 += 8 * nt) {
      const uint32_t pn = p0 + 8 * nt;
      chunk_t nxt = zero;
      if (pn < n8)
        nxt.q = *reinterpret_cast<const tb_u32x4 *>(w + pn);
      finish(p0, cur);
      cur = nxt;
    }
  }
  for (uint32_t p = n8 + tid; p < n; p += nt) {
    const int16_t ev = received(p);
    const int16_t acc = (int16_t)((clear ? 0 : w[p]) + ev);
    if (clear || ev != 0)
      w[p] = acc;
    if (p < np)
      l[p + twoZ] = pack(p, acc);
  }
}

#if defined(__HIPCC__)
/* the phases with their barriers, executed by every thread of the workgroup (wave-uniform arguments).  Only the loads and
 * the scatter are compiled per modulation order; the clearing and the streaming phase exist once. */
__device__ __forceinline__ void tb_rx_dematch_block(const tb_rx_geom &g, uint32_t Qm, const int16_t *__restrict__ f, int16_t *__restrict__ w,
                                                    int8_t *__restrict__ l, int16_t *e_lds, unsigned long long *stamps = nullptr)
{
  const uint32_t tid = threadIdx.x, nt = blockDim.x;
  tb_rx_ahead first;
  switch (Qm) {
    case 2: tb_rx_phase_load_first<2>(g, f, tid, nt, first); break;
    case 4: tb_rx_phase_load_first<4>(g, f, tid, nt, first); break;
    case 6: tb_rx_phase_load_first<6>(g, f, tid, nt, first); break;
    default: tb_rx_phase_load_first<8>(g, f, tid, nt, first); break;
  }
  tb_rx_phase_zero(g, e_lds, l, tid, nt);
  __syncthreads();
  if (stamps && tid == 0)
    stamps[0] = wall_clock64();
  const uint32_t nlaps = tb_rx_laps(g);
  if (nlaps == 1) {
    switch (Qm) {
      case 2: tb_rx_phase_scatter_lap<2, true>(g, f, e_lds, 0, 1, tid, nt, first); break;
      case 4: tb_rx_phase_scatter_lap<4, true>(g, f, e_lds, 0, 1, tid, nt, first); break;
      case 6: tb_rx_phase_scatter_lap<6, true>(g, f, e_lds, 0, 1, tid, nt, first); break;
      default: tb_rx_phase_scatter_lap<8, true>(g, f, e_lds, 0, 1, tid, nt, first); break;
    }
    __syncthreads();
  } else
  switch (Qm) {
    case 2:
      for (uint32_t lap = 0; lap < nlaps; lap++) { tb_rx_phase_scatter_lap<2>(g, f, e_lds, lap, nlaps, tid, nt, first); __syncthreads(); }
      break;
    case 4:
      for (uint32_t lap = 0; lap < nlaps; lap++) { tb_rx_phase_scatter_lap<4>(g, f, e_lds, lap, nlaps, tid, nt, first); __syncthreads(); }
      break;
    case 6:
      for (uint32_t lap = 0; lap < nlaps; lap++) { tb_rx_phase_scatter_lap<6>(g, f, e_lds, lap, nlaps, tid, nt, first); __syncthreads(); }
      break;
    default:
      for (uint32_t lap = 0; lap < nlaps; lap++) { tb_rx_phase_scatter_lap<8>(g, f, e_lds, lap, nlaps, tid, nt, first); __syncthreads(); }
      break;
  }
  if (stamps && tid == 0)
    stamps[1] = wall_clock64();
  tb_rx_phase_stream(g, e_lds, w, l, tid, nt);
  if (stamps && tid == 0)
    stamps[2] = wall_clock64();
}
#endif
#endif
