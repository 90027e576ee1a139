/*
 * ldpc_emul.cpp -- CPU emulation of ONE workgroup of the HIP kernels (test infrastructure).
 *
 * Includes the very per-thread phase bodies the GPU kernels are built from
 * (openairinterface5g_amd/csrc/ldpc_dec_core.h, ldpc_enc_core.h) and walks the threads of a workgroup
 * sequentially, phase by phase, following the block-level control flow of ldpc_decoder.hip /
 * ldpc_encoder.hip.  It exists so that the table builder, the schedules and the index arithmetic can be
 * checked against the oracle in the GPU-less development container; it is never part of the product and
 * proves nothing about barriers, LDS sizes or wave intrinsics -- the `-m gpu` tests do that.
 */
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../openairinterface5g_amd/csrc/ldpc_enc_core.h"

static void crc_pow_table(uint32_t poly, std::vector<uint32_t> &t, int n)
{
  t.resize(n);
  uint32_t r = poly;
  for (int j = 0; j < n; j++) {
    t[j] = r;
    r = (r & 0x80000000u) ? ((r << 1) ^ poly) : (r << 1);
  }
}

extern "C" int ldpc_emul_decode(int BG, int Z, int R, int numMaxIter, int outMode, int use_crc, int E, int crc_type,
                                const int8_t *llr_in, int8_t *out)
{
  ldpc_code_desc_t code_s;
  if (ldpc_build_code_desc(BG, Z, R, &code_s) != 0)
    return -1;
  const ldpc_code_desc_t *code = &code_s;
  std::vector<int8_t> smem(code->lds_total, 0x5a); /* poison: the kernel must initialise what it reads */
  int8_t *r = smem.data() + code->lds_r, *app = smem.data() + code->lds_app, *llr_s = smem.data() + code->lds_llr;
  int flags[4] = {0, 0, 0, 0};
  const int nt = code->n_threads, nw = nt >> 6, num_llr = code->num_llr, ncz = code->ncore * Z;
  static const uint32_t polys[4] = {0x864cfb00u, 0x80006300u, 0x10210000u, 0x9B000000u};
  std::vector<uint32_t> crc_pow;
  if (use_crc)
    crc_pow_table(polys[crc_type], crc_pow, 8448);

  memcpy(llr_s, llr_in, num_llr);
  memset(r, 0, (size_t)((code->nedges * Z + 15) >> 4) << 4);
  memcpy(app, llr_s, ncz);

  const int max_pass = numMaxIter + 1;
  int n_iter = max_pass;
  for (int p = 1; p <= max_pass; ++p) {
    for (int tid = 0; tid < nt; tid++) {
      const int lane = tid & 63, wave = tid >> 6;
      int par_acc = 0;
      const int pack = code->pack, sub = pack > 1 ? (lane * code->zinv16) >> 16 : 0, tl = lane - sub * Z;
      for (int k = wave; k < code->n_cn_slots; k += nw) {
        const int ent = code->cn_order[k];
        if (pack > 1) {
          if (sub < (ent >> 8)) {
            const int row = code->cn_rows[(ent & 0xff) + sub];
            const int par = ldpc_cn_row(code, row, tl, r, app, llr_s);
            par_acc |= (tl < code->pc_lo[row]) ? par : 0;
          }
          continue;
        }
        const int row = ent >> 4, t = ((ent & 15) << 6) + lane;
        if (t < Z) {
          const int par = ldpc_cn_row(code, row, t, r, app, llr_s);
          par_acc |= (t < code->pc_lo[row]) ? par : 0;
        }
      }
      if (par_acc)
        flags[p & 1] = 1;
    }
    flags[2] = 0;
    if (!use_crc && p >= 3 && flags[p & 1] == 0) {
      n_iter = p - 1;
      break;
    }
    for (int tid = 0; tid < nt; tid++) {
      const int lane = tid & 63, wave = tid >> 6;
      for (int k = wave; k < code->n_bn_slots; k += nw) {
        if (code->pack > 1) {
          const int i = 64 * k + lane;
          if (i < ncz) {
            const int ci = (i * code->zinv16) >> 16;
            ldpc_bn_update(code, code->bn_cols[ci], i - ci * Z, r, app, llr_s);
          }
          continue;
        }
        const int ent = code->bn_order[k];
        const int c = ent >> 4, u = ((ent & 15) << 6) + lane;
        if (u < Z)
          ldpc_bn_update(code, c, u, r, app, llr_s);
      }
    }
    flags[(p + 1) & 1] = 0;
    if (use_crc && p >= 3) {
      uint32_t x = 0;
      for (int i = 0; i < E; i++)
        if (app[i] < 0)
          x ^= crc_pow[E - 1 - i];
      if (x == 0) {
        n_iter = p;
        break;
      }
    }
  }
  if (!use_crc || n_iter >= 3) {
    if (outMode == 0) {
      const int nwords = (num_llr + 31) >> 5;
      for (int w = 0; w < nwords; w++) {
        const uint32_t v = (32 * w < ncz) ? ldpc_pack_word(app, w, ncz) : 0u;
        memcpy(out + 4 * w, &v, 4);
      }
    } else {
      for (int i = 0; i < num_llr; i++)
        out[i] = (i < ncz) ? (int8_t)(app[i] < 0) : (int8_t)0;
    }
  }
  return n_iter;
}

extern "C" int ldpc_emul_encode(int BG, int Zc, int Kb, const uint8_t *in, uint8_t *out)
{
  ldpc_code_desc_t code_s;
  if (ldpc_build_code_desc(BG, Zc, BG == 1 ? 13 : 15, &code_s) != 0)
    return -1;
  const ldpc_code_desc_t *code = &code_s;
  std::vector<uint8_t> x(((code->ncols * Zc + 15) & ~15), 0x5a), lam(4 * Zc + 16, 0x5a);
  int waves = (Zc + 63) / 64 * 2;
  if (waves > 16) waves = 16;
  const int nt = waves * 64;
  for (int ph = 0; ph < LDPC_ENC_NUM_PHASES; ph++)
    for (int tid = 0; tid < nt; tid++)
      ldpc_enc_phase(ph, code, Kb, in, x.data(), lam.data(), out, tid, nt);
  return (code->ncols - 2) * Zc;
}

/* ---- bit-packed encoder kernel (ldpc_enc_packed_kernel) ------------------------------------------------------- */
#include "../../openairinterface5g_amd/csrc/ldpc_enc_packed_core.h"

extern "C" int ldpc_emul_encode_packed(int BG, int Zc, int Kb, const uint8_t *in, uint8_t *out)
{
  ldpc_code_desc_t code_s;
  if (ldpc_build_code_desc(BG, Zc, BG == 1 ? 13 : 15, &code_s) != 0)
    return -1;
  const ldpc_code_desc_t *code = &code_s;
  std::vector<uint32_t> lds(ldpc_encp_lds_words(code->ncols, code->kb_full, Zc, code->nrows, code->nedges), 0x5a5a5a5au);
  ldpc_encp_lds L;
  ldpc_encp_carve(lds.data(), code, L);
  const int nt = ldpc_encp_threads(code->nrows, Zc);
  for (int ph = 0; ph < LDPC_ENCP_NUM_PHASES; ph++)
    for (int tid = 0; tid < nt; tid++)
      ldpc_encp_phase(ph, code, Kb, in, L, out, tid, nt);
  return (code->ncols - 2) * Zc;
}

/* ---- the same kernel's path for Zc % 32 == 0 (ldpc_enc_packed32.h): the per-lane phases as the kernel calls them, one
 * thread at a time; the DPP meeting of a lambda item's four slices and the wave-level order of the two core parity steps
 * are loops here.  -2: the code does not take this path. */
#include "../../openairinterface5g_amd/csrc/ldpc_enc_packed32.h"

extern "C" int ldpc_emul_encode_packed32(int BG, int Zc, int Kb, const uint8_t *in, uint8_t *out, int nt)
{
  ldpc_code_desc_t code_s;
  if (ldpc_build_code_desc(BG, Zc, BG == 1 ? 13 : 15, &code_s) != 0)
    return -1;
  const ldpc_code_desc_t *code = &code_s;
  if (!ldpc_encp32_applies(code))
    return -2;
  std::vector<uint32_t> lds(ldpc_encp_lds_words(code->ncols, code->kb_full, Zc, code->nrows, code->nedges), 0x5a5a5a5au);
  ldpc_encp_lds L;
  ldpc_encp_carve(lds.data(), code, L);
  if (nt <= 0)
    nt = ldpc_encp_threads(code->nrows, Zc);
  const ldpc_encp32 g = ldpc_encp32_make(code, Kb);
  const uint32_t nd = (uint32_t)(g.kbf * g.W);
  std::vector<uint32_t> in32(nd);
  memcpy(in32.data(), in, (size_t)nd * 4);
  /* loads -> LDS (ldpc_enc_packed_kernel's first stage) */
  for (uint32_t i = 0; i < nd; i++)
    ldpc_encp32_info(L, g, i, in32[i]);
  for (int e = 0; e < code->nedges; e++)
    L.ET[e] = code->enc_et[e];
  for (int r = 0; r <= code->nrows; r++)
    L.RP[r] = (uint32_t)code->row_ptr[r];
  /* lambda: items of four lanes */
  for (int i = 0; i < 16 * g.W; i += 4) {
    uint32_t acc = 0;
    for (int sl = 0; sl < 4; sl++)
      acc ^= ldpc_encp32_lambda_partial(L, g, i + sl);
    ldpc_encp32_lambda_store(L, g, i, acc);
  }
  /* core parity: one wave, two steps */
  for (int lane = 0; lane < 64; lane++)
    ldpc_encp32_core_step_p0(code, L, g, lane);
  for (int lane = 0; lane < 64; lane++)
    ldpc_encp32_core_step_rest(code, L, g, lane);
  for (int tid = 0; tid < nt; tid++)
    ldpc_encp32_extension(code, L, g, tid, nt);
  std::vector<uint8_t> o2((size_t)(code->ncols - 2) * Zc + 32); /* (the stores are 16 bytes wide and want alignment) */
  uint8_t *oa = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(o2.data()) + 15) & ~(uintptr_t)15);
  for (int tid = 0; tid < nt; tid++)
    ldpc_encp32_store_bytes(L, g, oa, 2, code->ncols, tid, nt);
  memcpy(out, oa, (size_t)(code->ncols - 2) * Zc);
  return (code->ncols - 2) * Zc;
}

extern "C" int ldpc_emul_desc(int BG, int Z, int R, ldpc_code_desc_t *d) { return ldpc_build_code_desc(BG, Z, R, d); }

/* ---- fast kernel (ldpc_decoder_fast.hip) ---------------------------------------------------------------- */
#include "../../openairinterface5g_amd/csrc/ldpc_dec_fast_core.h"

static int g_fast_shape = LDPC_SHAPE_THROUGHPUT;
extern "C" void ldpc_emul_set_fast_shape(int shape) { g_fast_shape = shape; }

extern "C" int ldpc_emul_decode_fast(int BG, int Z, int R, int numMaxIter, int outMode, int use_crc, int E, int crc_type,
                                     const int8_t *llr_in, int8_t *out)
{
  ldpc_code_desc_t code_s;
  if (ldpc_build_code_desc_shape(BG, Z, R, g_fast_shape, &code_s) != 0)
    return -1;
  const ldpc_code_desc_t *code = &code_s;
  if (!code->f_ok)
    return -2;
  std::vector<uint32_t> smem32((code->f_lds_total + 3) / 4 + 4, 0x5a5a5a5au);
  uint8_t *fsm = reinterpret_cast<uint8_t *>(smem32.data());
  const int zq = code->f_zq, rstride = code->f_rstride, astride = code->f_astride;
  const uint32_t zq_magic = code->f_zq_magic, z_magic = 0xffffffffu / (uint32_t)Z + 1u;
  ldpc_fast_lds L;
  L.base = fsm;
  L.r = fsm + code->f_lds_r;
  L.app = fsm + code->f_lds_app;
  L.ext = fsm + code->f_lds_ext;
  uint32_t *etbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_etbl);
  uint32_t *ctbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_ctbl);
  uint32_t *rowtbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_rowtbl);
  uint32_t *coltbl = reinterpret_cast<uint32_t *>(fsm + code->f_lds_coltbl);
  L.etbl = etbl; L.ctbl = ctbl; L.rowtbl = rowtbl; L.coltbl = coltbl;
  int flags[4] = {0, 0, 0, 0};
  const int nt = code->f_n_threads, ncore = code->ncore, num_llr = code->num_llr, ncz = ncore * Z, nedges = code->nedges;
  std::vector<uint32_t> src_copy((num_llr + 3) / 4 + 1);
  memcpy(src_copy.data(), llr_in, num_llr);
  const uint32_t *src32 = src_copy.data();
  static const uint32_t polys[4] = {0x864cfb00u, 0x80006300u, 0x10210000u, 0x9B000000u};
  std::vector<uint32_t> crc_pow;
  if (use_crc)
    crc_pow_table(polys[crc_type], crc_pow, 8448);

  L.gllr = reinterpret_cast<const uint8_t *>(src32);
  L.ext_global = code->f_ext_global;
  for (int i = 0; i < nedges; i++) etbl[i] = code->f_etbl[i];
  for (int i = 0; i < 2 * code->f_n_ctbl; i++) ctbl[i] = code->f_ctbl[i];
  for (int i = 0; i < (Z + 4) >> 2; i++) reinterpret_cast<uint32_t *>(fsm + code->f_lds_zero)[i] = 0u;
  for (int i = 0; i < code->nrows; i++) rowtbl[i] = code->f_rowtbl[i];
  for (int i = 0; i < ncore; i++) coltbl[i] = code->f_coltbl[i];
  for (int i = 0; i < ncore * zq; i++) {
    const int c = (int)ldpc_umulhi((uint32_t)i, zq_magic), j = i - c * zq;
    const uint32_t w = src32[i] ^ 0x80808080u;
    uint32_t *dst = reinterpret_cast<uint32_t *>(L.app + c * astride) + j;
    dst[0] = w;
    dst[zq] = w;
  }
  {
    const int next4 = code->f_ext_global ? 0 : (code->ncols - ncore) * zq;
    uint32_t *e32 = reinterpret_cast<uint32_t *>(L.ext);
    for (int i = 0; i < next4; i++) e32[i] = src32[ncore * zq + i] ^ 0x80808080u;
    /* (the message array keeps the 0x5a poison: the first pass must not read it -- as in the kernel) */
    (void)rstride;
  }
  const int max_pass = numMaxIter + 1;
  int n_iter = max_pass;
  for (int p = 1; p <= max_pass; ++p) {
    (void)nt;
    /* the kernel's waves draw the tasks of a phase from a queue; the tasks are independent, so any order will do */
    for (int task = 0; task < code->f_n_cn_tasks; task++) {
      /* (a double task -- degree | 0x100 -- is 128 items: the kernel's threads take items lane and lane + 64 together) */
      for (int lane = 0; lane < ((code->f_cn_task[task][0] & 0x100) ? 128 : 64); lane++) {
        uint32_t syn = 0;
        const int deg = code->f_cn_task[task][0] & 0xff, ext = code->f_cn_task[task][1];
        const int item = code->f_cn_task[task][2] + lane;
        const int gstart = code->f_cn_task[task][3], gend = code->f_cn_task[task][4], srow0 = code->f_cn_task[task][5];
        if (item < gend) {
          /* degree-19 rows come as pairs of items (two lanes of the kernel share a row item): one call per pair here */
          const bool pair = deg == 19 && code->f_pair19;
          if (pair && ((item - gstart) & 1))
            continue;
          const int gi = pair ? (item - gstart) >> 1 : item - gstart;
          const int rig = (int)ldpc_umulhi((uint32_t)gi, zq_magic), j = gi - rig * zq;
          const uint32_t rowrec = rowtbl[srow0 + rig];
          const int e0 = (int)(rowrec & 0x1ffu), valid = (int)(rowrec >> 16) - 4 * j;
          const uint32_t m = p == 1 ? ldpc_fast_cn_dispatch<true>(deg, ext, L, e0, j, Z, rstride)
                                    : ldpc_fast_cn_dispatch<false>(deg, ext, L, e0, j, Z, rstride);
          syn |= m & ldpc_fast_valid_lanes(valid);
        }
        if (syn)
          flags[p & 1] = 1;
      }
    }
    flags[2] = 0;
    if (!use_crc && p >= 3 && flags[p & 1] == 0) {
      n_iter = p - 1;
      break;
    }
    /* the kernel's bit-node queue: tickets (ldpc_graph.h f_bn_ticket) -- a long task on its own, short ones several to a
     * ticket and walked together, one item of each per thread (ldpc_fast_bn_multi) */
    for (int ticket = 0; ticket < code->f_n_bn_tickets; ticket++) {
      const int task = code->f_bn_ticket[ticket][0], cnt = code->f_bn_ticket[ticket][1];
      for (int lane = 0; lane < 64; lane++) {
        const int item0 = code->f_bn_task[task][0] + lane, end = code->f_bn_task[task][1];
        const int maxdeg = code->f_bn_task[task][2];
        if (cnt == 1) {
          if (item0 < end) {
            const int sc = (int)ldpc_umulhi((uint32_t)item0, zq_magic), j = item0 - sc * zq;
            const uint32_t colrec = coltbl[sc];
            const uint32_t lw = src32[(int)(colrec & 0xffu) * zq + j];
            ldpc_fast_bn(L, colrec, maxdeg, j, Z, astride, lw);
          }
        } else {
          uint32_t rec[LDPC_F_BN_GROUP], lw[LDPC_F_BN_GROUP];
          int jj[LDPC_F_BN_GROUP];
          bool live[LDPC_F_BN_GROUP];
          for (int g = 0; g < LDPC_F_BN_GROUP; g++) {
            const int item = item0 + 64 * g;
            live[g] = g < cnt && item < end;
            const int it = live[g] ? item : end - 1;
            const int sc = (int)ldpc_umulhi((uint32_t)it, zq_magic);
            jj[g] = it - sc * zq;
            rec[g] = coltbl[sc];
            lw[g] = src32[(int)(rec[g] & 0xffu) * zq + jj[g]];
          }
          ldpc_fast_bn_multi<LDPC_F_BN_GROUP>(L, rec, jj, lw, live, maxdeg, Z, astride);
        }
      }
    }
    flags[(p + 1) & 1] = 0;
    if (use_crc && p >= 3) {
      uint32_t x = 0;
      for (int i = 0; i < E; i++)
        if (ldpc_fast_hd(L, i, Z, z_magic, astride))
          x ^= crc_pow[E - 1 - i];
      if (x == 0) {
        n_iter = p;
        break;
      }
    }
  }
  if (!use_crc || n_iter >= 3) {
    if (outMode == 0) {
      const int nwords = (num_llr + 31) >> 5;
      for (int w = 0; w < nwords; w++) {
        uint32_t word = 0;
        for (int q = 0; q < 8; q++) {
          const int b = 32 * w + 4 * q;
          if (b < ncz) {
            const int c = (int)ldpc_umulhi((uint32_t)b, z_magic), u = b - c * Z;
            const uint32_t nb = ~*reinterpret_cast<const uint32_t *>(L.app + c * astride + u);
            const uint32_t nib = (((nb >> 7) & 1u) << 3) | (((nb >> 15) & 1u) << 2) | (((nb >> 23) & 1u) << 1) | (nb >> 31);
            word |= nib << (8 * (q >> 1) + ((q & 1) ? 0 : 4));
          }
        }
        memcpy(out + 4 * w, &word, 4);
      }
    } else {
      for (int i = 0; i < num_llr; i++)
        out[i] = (i < ncz) ? (int8_t)ldpc_fast_hd(L, i, Z, z_magic, astride) : (int8_t)0;
    }
  }
  return n_iter;
}

/* product descriptor -> flat arrays, for tests/test_tables.py */
extern "C" int ldpc_emul_desc_edges(int BG, int Z, int R, int *dims /*nrows,ncols,ncore,nedges,f_ok,lds_total,f_lds_total*/,
                                    int *row_ptr, int *col, int *shift, int *pc_lo)
{
  ldpc_code_desc_t d;
  if (ldpc_build_code_desc(BG, Z, R, &d) != 0)
    return -1;
  dims[0] = d.nrows; dims[1] = d.ncols; dims[2] = d.ncore; dims[3] = d.nedges; dims[4] = d.f_ok;
  dims[5] = d.lds_total; dims[6] = d.f_lds_total;
  for (int r = 0; r <= d.nrows; r++) row_ptr[r] = d.row_ptr[r];
  for (int r = 0; r < d.nrows; r++) pc_lo[r] = d.pc_lo[r];
  for (int e = 0; e < d.nedges; e++) {
    col[e] = d.e_col[e];
    shift[e] = (int)(d.e_info[e] & 0xffffu);
    if ((int)(d.e_info[e] >> 16) != d.e_col[e] * Z)
      return -2;
  }
  return 0;
}

/* ---- UL-SCH front step of one segment (tb_rx_core.h: the de-matching kernel and the fused segment kernel's prologue) ------
 * The workgroup's phases, thread after thread; the LDS image is a heap buffer of exactly the size the kernels reserve. */
#include "../../openairinterface5g_amd/csrc/tb_rx_core.h"
#include "../../openairinterface5g_amd/csrc/nr_coding_host.h"

template <int QM> static void emul_scatter(const tb_rx_geom &g, const int16_t *f, int16_t *e_lds, int8_t *l, int nt)
{
  std::vector<tb_rx_ahead> first(nt); /* every thread's registers across the barrier */
  for (int tid = 0; tid < nt; tid++) {
    tb_rx_phase_load_first<QM>(g, f, (uint32_t)tid, (uint32_t)nt, first[tid]);
    tb_rx_phase_zero(g, e_lds, l, (uint32_t)tid, (uint32_t)nt);
  }
  const uint32_t nlaps = tb_rx_laps(g);
  if (nlaps == 1) { /* the straight-line form of a single lap, as tb_rx_dematch_block picks it */
    for (int tid = 0; tid < nt; tid++)
      tb_rx_phase_scatter_lap<QM, true>(g, f, e_lds, 0, 1, (uint32_t)tid, (uint32_t)nt, first[tid]);
    return;
  }
  for (uint32_t lap = 0; lap < nlaps; lap++)
    for (int tid = 0; tid < nt; tid++)
      tb_rx_phase_scatter_lap<QM>(g, f, e_lds, lap, nlaps, (uint32_t)tid, (uint32_t)nt, first[tid]);
}

extern "C" int tb_emul_rx_dematch(uint32_t Tbslbrm, int BG, uint32_t Zc, uint32_t C, uint32_t F, uint32_t K, int rv, uint32_t E, uint32_t Qm,
                                  uint32_t num_llr, int clear, int nt, const int16_t *f, int16_t *w, int8_t *l)
{
  nr_hip_rm_t rm;
  if (nr_hip_rate_match_geometry(Tbslbrm, BG, Zc, C, F, K, rv, E, &rm) != 0)
    return -1;
  tb_rx_seg_job j;
  memset(&j, 0, sizeof(j));
  j.E = E; j.Qm = Qm; j.Ncb = rm.Ncb; j.Foffset = rm.Foffset; j.Fin = rm.Fin; j.V = rm.V; j.rank0 = rm.rank0;
  j.clear = clear ? 1u : 0u;
  j.K = K; j.F = F; j.Z = Zc; j.num_llr = num_llr;
  const tb_rx_geom g = tb_rx_geometry(&j);
  std::vector<tb_u32x4> lds(g.span / 8 + 1);
  memset(lds.data(), 0x5a, lds.size() * sizeof(tb_u32x4)); /* poison: phase Z must initialise what is read */
  int16_t *e_lds = reinterpret_cast<int16_t *>(lds.data());
  switch (Qm) {
    case 2: emul_scatter<2>(g, f, e_lds, l, nt); break;
    case 4: emul_scatter<4>(g, f, e_lds, l, nt); break;
    case 6: emul_scatter<6>(g, f, e_lds, l, nt); break;
    default: emul_scatter<8>(g, f, e_lds, l, nt); break;
  }
  for (int tid = 0; tid < nt; tid++)
    tb_rx_phase_stream(g, e_lds, w, l, (uint32_t)tid, (uint32_t)nt);
  return (int)g.span;
}
