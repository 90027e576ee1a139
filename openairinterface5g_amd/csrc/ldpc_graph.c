/*
 * ldpc_graph.c -- host-side builder of the lifted-code descriptor (see ldpc_graph.h).
 * Plain C, no GPU dependency: unit-tested on CPU against the reference's LUT data.
 */
#include <string.h>
#include <stdlib.h>
#include "ldpc_graph.h"
#include "nr_ldpc_bg_tables.h"

int ldpc_lifting_set_index(int Z)
{
  /* 38.212 Table 5.3.2-1: Z = a * 2^j with a in {2,3,5,7,9,11,13,15}; set index = position of a */
  static const int a[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  int ok = 0;
  for (int i = 0; i < NR_LDPC_NUM_LIFT; i++)
    ok |= nr_ldpc_lift_sizes[i] == Z;
  if (!ok)
    return -1;
  int odd = Z;
  while ((odd & 1) == 0)
    odd >>= 1;
  if (odd == 1)
    return 0; /* pure power of two belongs to the a = 2 family */
  for (int i = 1; i < 8; i++)
    if (a[i] == odd)
      return i;
  return -1;
}

typedef struct { int key, id; } sort_item_t;
static int by_key_desc(const void *pa, const void *pb)
{
  const sort_item_t *a = (const sort_item_t *)pa, *b = (const sort_item_t *)pb;
  if (a->key != b->key)
    return b->key - a->key;
  return a->id - b->id;
}

static int align16(int x) { return (x + 15) & ~15; }

/* does row r end in its degree-1 (extension) column?  (rows >= 4 of the whole code; a part descriptor renumbers its rows) */
static int row_has_ext(const ldpc_code_desc_t *d, int r) { return d->e_col[d->row_ptr[r] + d->row_deg[r] - 1] >= d->ncore; }

/* schedules and tables of the "fast" decoder kernel (see ldpc_graph.h) */
static void build_fast_section(ldpc_code_desc_t *d, int shape, int mb, int pair19)
{
  const int Z = d->Z;
  d->f_ok = 0;
  if ((Z & 3) || Z < 8)
    return;
  const int zq1 = Z / 4;
  const int zq = zq1 * mb; /* items per lifted row / column: all blocks of the workgroup (mb = 1: the one block) */
  d->f_mb = mb;
  d->f_pair19 = 0;
  if (d->f_sub != 4)
    d->f_sub = 1;
  d->f_zq = zq1;
  d->f_zq_magic = (uint32_t)((0x100000000ULL + (uint64_t)zq1 - 1) / (uint64_t)zq1);
  d->f_zqb = zq;
  d->f_zqb_magic = (uint32_t)((0x100000000ULL + (uint64_t)zq - 1) / (uint64_t)zq);
  d->f_rstride = mb * (Z + 4);
  d->f_astride = mb * 2 * Z;

  /* rows sorted by degree (descending, stable) */
  sort_item_t rows[LDPC_MAX_ROWS];
  for (int r = 0; r < d->nrows; r++) {
    rows[r].key = d->row_deg[r];
    rows[r].id = r;
  }
  qsort(rows, d->nrows, sizeof(rows[0]), by_key_desc);
  for (int i = 0; i < d->nrows; i++)
    d->f_rowtbl[i] = (uint32_t)d->row_ptr[rows[i].id] | ((uint32_t)rows[i].key << 9) | ((uint32_t)row_has_ext(d, rows[i].id) << 14) |
                     ((uint32_t)d->pc_lo[rows[i].id] << 16);
  /* CN tasks: one degree group after the other, 64 items per task */
  int nt = 0, item = 0, cost[LDPC_F_MAX_CN_TASKS];
  for (int i = 0; i < d->nrows;) {
    int j = i;
    while (j < d->nrows && rows[j].key == rows[i].key)
      j++;
    /* degree-19 rows (the core rows, no extension column): two lanes per item when pair19 */
    const int paired = pair19 && rows[i].key == 19 && !row_has_ext(d, rows[i].id);
    if (paired)
      d->f_pair19 = 1;
    const int gstart = item, gend = item + (j - i) * zq * (paired ? 2 : 1);
    int dbl_max = LDPC_F_CN_DOUBLE;
    {
      const char *e = getenv("NRLDPC_HIP_CN_DOUBLE"); /* tuning knob: largest row degree that gets double tasks (0: none) */
      if (e)
        dbl_max = atoi(e);
      if (dbl_max > LDPC_F_CN_DOUBLE)
        dbl_max = LDPC_F_CN_DOUBLE; /* (the kernel has two-item bodies for degrees 3 .. LDPC_F_CN_DOUBLE) */
    }
    /* (groups of at least three rows: a lone row's 96 items make one three-quarters-full double task where two single ones
     * were enough -- BG1 R = 8/9 lost 3 % that way, profiles/r06/ab_double_tasks.txt) */
    const int dbl = mb == 1 && d->f_sub != 4 && !paired && rows[i].key >= 3 && rows[i].key <= dbl_max && row_has_ext(d, rows[i].id) && zq >= 64 &&
                    j - i >= 3;
    for (int b = gstart; b < gend; b += dbl ? 128 : 64) {
      if (nt >= LDPC_F_MAX_CN_TASKS)
        return;
      d->f_cn_task[nt][0] = rows[i].key | (dbl ? 0x100 : 0);
      d->f_cn_task[nt][1] = row_has_ext(d, rows[i].id); /* rows of one degree group are all core or all extension rows? checked below */
      d->f_cn_task[nt][2] = b;
      d->f_cn_task[nt][3] = gstart;
      d->f_cn_task[nt][4] = gend;
      d->f_cn_task[nt][5] = i;
      /* instruction-count model of a task: ~36 VALU per edge (47 for the degree-19 rows, which re-read LDS in their
       * second sweep) + ~40 of prologue/epilogue */
      cost[nt] = paired ? 10 * 36 + 60 : rows[i].key * (rows[i].key >= 16 ? 47 : 36) + 40;
      if (dbl)
        cost[nt] = 2 * rows[i].key * 36 + 60;
      nt++;
    }
    /* a degree group must not mix core rows (no extension column) with extension rows */
    for (int k = i; k < j; k++)
      if (row_has_ext(d, rows[k].id) != row_has_ext(d, rows[i].id))
        return;
    item = gend;
    i = j;
  }
  d->f_n_cn_tasks = nt;
  /* the queue hands tasks out in id order: most expensive first (insertion sort, stable; double tasks move up) */
  for (int a = 1; a < nt; a++) {
    int32_t rec[6];
    const int c = cost[a];
    memcpy(rec, d->f_cn_task[a], sizeof(rec));
    int k = a - 1;
    while (k >= 0 && cost[k] < c) {
      memcpy(d->f_cn_task[k + 1], d->f_cn_task[k], sizeof(rec));
      cost[k + 1] = cost[k];
      k--;
    }
    memcpy(d->f_cn_task[k + 1], rec, sizeof(rec));
    cost[k + 1] = c;
  }
  /* columns sorted by degree (descending), their adjacency, BN tasks */
  sort_item_t cols[LDPC_MAX_CORE];
  for (int c = 0; c < d->ncore; c++) {
    cols[c].key = d->col_ptr[c + 1] - d->col_ptr[c];
    cols[c].id = c;
  }
  qsort(cols, d->ncore, sizeof(cols[0]), by_key_desc);
  /* adjacency lists in sorted-column order, padded (see ldpc_graph.h); the row offsets are filled in below once the
   * LDS layout is known */
  const int span = (63 + zq - 1) / zq; /* how many earlier columns can share a 64-item task with a column */
  int bn_short = LDPC_F_BN_SHORT, short_max = 0; /* degree of the largest short column (ldpc_graph.h f_bn_ticket) */
  {
    const char *e = getenv("NRLDPC_HIP_BN_SHORT"); /* tuning knob: 0 = every bit-node task is a ticket of its own */
    if (e)
      bn_short = atoi(e);
    if (mb > 1 || d->f_sub == 4)
      bn_short = 0; /* (the several-blocks kernels keep f_bn_group) */
    for (int i = 0; i < d->ncore; i++)
      if (cols[i].key <= bn_short && cols[i].key > short_max)
        short_max = cols[i].key;
  }
  int n = 0;
  for (int i = 0; i < d->ncore; i++) {
    const int c = cols[i].id;
    int padded = cols[i - span > 0 ? i - span : 0].key;
    if (cols[i].key <= bn_short && padded < short_max)
      padded = short_max;
    if (n + padded > LDPC_F_MAX_CTBL)
      return;
    d->f_coltbl[i] = (uint32_t)c | ((uint32_t)cols[i].key << 8) | ((uint32_t)n << 16);
    int k = 0;
    for (int e = d->col_ptr[c]; e < d->col_ptr[c + 1]; e++, k++) {
      const uint32_t ce = d->col_edge[e];
      d->f_ctbl[2 * (n + k)] = (uint32_t)Z - (ce & 0xffffu);
      d->f_ctbl[2 * (n + k) + 1] = (ce >> 16) * (uint32_t)d->f_rstride; /* + f_lds_r below */
    }
    for (; k < padded; k++) {
      d->f_ctbl[2 * (n + k)] = (uint32_t)Z;
      d->f_ctbl[2 * (n + k) + 1] = 0xffffffffu; /* -> f_lds_zero below */
    }
    n += padded;
  }
  d->f_n_ctbl = n;
  const int nitems = d->ncore * zq;
  int nb = 0, bcost[LDPC_F_MAX_CN_TASKS] = {0};
  for (int b = 0; b < nitems; b += 64) {
    if (nb >= LDPC_F_MAX_BN_TASKS)
      return;
    d->f_bn_task[nb][0] = b;
    d->f_bn_task[nb][1] = nitems;
    d->f_bn_task[nb][2] = cols[b / zq].key;
    bcost[nb] = cols[b / zq].key * 11 + 45; /* ~11 VALU per gathered edge + finalisation */
    nb++;
  }
  d->f_n_bn_tasks = nb;

  /* the magic division must be exact for every item index that occurs */
  const int max_item = (d->nrows + (d->f_pair19 ? 4 : 0)) * zq > nitems ? (d->nrows + (d->f_pair19 ? 4 : 0)) * zq : nitems;
  for (int i = 0; i < max_item + 64; i++)
    if ((int)(((uint64_t)i * d->f_zqb_magic) >> 32) != i / zq || (int)(((uint64_t)i * d->f_zq_magic) >> 32) != i / zq1)
      return;

  /* LDS layout.  The channel LLRs of the degree-1 columns are read once per pass by one check-node edge each; when
   * leaving them in global memory (L2) lets one more workgroup fit on a CU, they are not staged (f_ext_global). */
  const int ext_bytes = (mb > 1 || d->f_sub == 4) ? align16((d->ncols - d->ncore) * d->f_rstride) : align16((d->ncols - d->ncore) * Z);
  const int misc_bytes = (d->f_sub == 4 || mb > 1) ? 1664 : 256; /* ldpc_dec_fast_mblock.h: per-block flag arrays; one block:
                                                                    16 flag words */
  const int llr_bytes = d->f_sub == 4 ? align16(d->ncore * d->f_astride / 2) : 0; /* [ncore][mb][Z] */
  const int fixed = align16(d->nedges * d->f_rstride) + align16(d->ncore * d->f_astride) + align16(d->nedges * 4) +
                    align16(d->f_n_ctbl * 8) + align16(d->nrows * 4) + align16(d->ncore * 4) + align16(mb * (Z + 4)) + misc_bytes + llr_bytes;
  /* Workgroup shape.  A CU holds 16 waves of this kernel (<= 128 VGPRs), so the waves per workgroup w and the
   * workgroups per CU k are chosen together: maximise the resident waves k*w subject to k workgroups fitting in the
   * 160 KiB of LDS (with or without the staged extension LLRs) and w <= check-node tasks; w a multiple of 4 so that the
   * waves spread evenly over the four SIMDs.  Ties: more workgroups per CU (their barriers overlap), then LDS staging.
   * Measured on MI355X (profiles/r01/occupancy_sweep.txt): BG1 Zc=192 10 waves x 1 -> 8 x 2: 35 -> 48 Gb/s.
   * That is the THROUGHPUT shape (launches that fill the GPU more than once).  The LATENCY shape, for launches of at
   * most one workgroup round (the per-segment entry point above all), takes as many waves as there are tasks:
   * BG1 Zc=384 R=8/9, one block per CU: 2 waves 210 us, 8 waves 80 us. */
  int waves = 1;
  const int lds_cu = 160 * 1024;
  if (shape == LDPC_SHAPE_LATENCY) {
    waves = nt < LDPC_F_MAX_WAVES ? nt : LDPC_F_MAX_WAVES;
    d->f_ext_global = 0;
  } else {
    int best_score = -1, best_k = 0;
    d->f_ext_global = 0;
    for (int w = LDPC_F_MAX_WAVES; w >= 1; w = (w > 4 ? w - 4 : w - 1)) {
      if (w > nt)
        continue;
      for (int eg = 0; eg <= ((mb > 1 || d->f_sub == 4) ? 0 : 1); eg++) {
        int k = lds_cu / (fixed + (eg ? 0 : ext_bytes));
        if (k > 16 / w) k = 16 / w;
        if (k > 8) k = 8;
        if (k < 1)
          continue;
        const int score = k * w;
        if (score > best_score || (score == best_score && k > best_k)) {
          best_score = score; best_k = k; waves = w; d->f_ext_global = eg;
        }
      }
    }
    const char *env = getenv("NRLDPC_HIP_FAST_WAVES"); /* tuning knobs: force waves per workgroup / LLR staging */
    if (env && atoi(env) >= 1 && atoi(env) <= LDPC_F_MAX_WAVES)
      waves = atoi(env) < nt ? atoi(env) : nt;
    const char *eg = getenv("NRLDPC_HIP_EXT_GLOBAL");
    if (eg && (eg[0] == '0' || eg[0] == '1') && mb == 1 && d->f_sub != 4)
      d->f_ext_global = eg[0] == '1';
  }
  {
    int k = lds_cu / (fixed + (d->f_ext_global ? 0 : ext_bytes));
    if (k > 16 / waves) k = 16 / waves;
    d->f_wg_per_cu = k < 1 ? 1 : k;
  }
  d->f_n_threads = waves * 64;
  /* bit-node tasks of low-degree columns (the high-rate modes) are short next to the fetch chain in front of every
   * task: a queue ticket then stands for f_bn_group consecutive tasks, as long as every wave still gets two tickets */
  {
    int sum = 0;
    for (int i = 0; i < nb; i++)
      sum += bcost[i];
    const int avg = nb ? sum / nb : 1;
    int m = avg >= 120 ? 1 : (240 + avg - 1) / avg;
    if (m > 4) m = 4;
    if (m > nb / (2 * waves)) m = nb / (2 * waves);
    d->f_bn_group = m < 1 ? 1 : m;
  }
  /* tickets: long tasks alone, short ones grouped so that every wave gets about one group */
  {
    int n_short = 0;
    for (int i = 0; i < nb; i++)
      n_short += d->f_bn_task[i][2] <= bn_short;
    int g = waves > 0 ? (n_short + waves - 1) / waves : 1;
    if (g > LDPC_F_BN_GROUP) g = LDPC_F_BN_GROUP;
    if (g < 1) g = 1;
    const char *e = getenv("NRLDPC_HIP_BN_GROUP");
    if (e && atoi(e) >= 1 && atoi(e) <= LDPC_F_BN_GROUP)
      g = atoi(e);
    int nk = 0;
    for (int i = 0; i < nb;) {
      const int cnt = d->f_bn_task[i][2] <= bn_short ? (nb - i < g ? nb - i : g) : 1;
      d->f_bn_ticket[nk][0] = i;
      d->f_bn_ticket[nk][1] = cnt;
      nk++;
      i += cnt;
    }
    d->f_n_bn_tickets = nk;
  }
  /* the kernel's waves draw tasks 0, 1, 2, ... from a queue: task ids must already be in descending cost order
   * (rows and columns are sorted by degree, so they are) */
  for (int i = 1; i < nt; i++)
    if (cost[i] > cost[i - 1])
      return;
  for (int i = 1; i < nb; i++)
    if (bcost[i] > bcost[i - 1])
      return;
  d->f_lds_r = 0;
  d->f_lds_app = align16(d->nedges * d->f_rstride);
  d->f_lds_ext = d->f_lds_app + align16(d->ncore * d->f_astride);
  d->f_lds_etbl = d->f_lds_ext + (d->f_ext_global ? 0 : ext_bytes);
  d->f_lds_ctbl = d->f_lds_etbl + align16(d->nedges * 4);
  d->f_lds_rowtbl = d->f_lds_ctbl + align16(d->f_n_ctbl * 8);
  d->f_lds_coltbl = d->f_lds_rowtbl + align16(d->nrows * 4);
  d->f_lds_zero = d->f_lds_coltbl + align16(d->ncore * 4);
  d->f_lds_llr = d->f_lds_zero + align16(mb * (Z + 4));
  d->f_lds_misc = d->f_lds_llr + llr_bytes;
  d->f_lds_total = d->f_lds_misc + misc_bytes;
  if (d->f_lds_total > 160 * 1024)
    return;
  for (int i = 0; i < d->f_n_ctbl; i++) /* minus the window's byte phase (ldpc_fast_bn); may wrap below 0, mod 2^32 */
    d->f_ctbl[2 * i + 1] = (d->f_ctbl[2 * i + 1] == 0xffffffffu ? (uint32_t)d->f_lds_zero : d->f_ctbl[2 * i + 1] + (uint32_t)d->f_lds_r) -
                           (d->f_ctbl[2 * i] & 3u);
  /* edge table: absolute LDS byte offset of the neighbour's row start + shift */
  for (int e = 0; e < d->nedges; e++) {
    const int c = d->e_col[e], s = (int)(d->e_info[e] & 0xffffu);
    d->f_etbl[e] = c < d->ncore ? (uint32_t)(d->f_lds_app + c * d->f_astride + s)
                                : (d->f_ext_global ? (uint32_t)(c * Z)
                                                   : (uint32_t)(d->f_lds_ext + (c - d->ncore) * ((mb > 1 || d->f_sub == 4) ? d->f_rstride : Z)));
  }
  for (int k = 0; k < d->f_n_bn_tickets; k++) {
    const int task = d->f_bn_ticket[k][0];
    d->f_bn_rec[k][0] = d->f_bn_task[task][0];
    d->f_bn_rec[k][1] = d->f_bn_task[task][1];
    d->f_bn_rec[k][2] = d->f_bn_task[task][2];
    d->f_bn_rec[k][3] = d->f_bn_ticket[k][1];
  }
  /* the kernels treat an item's four lanes as checked or excluded TOGETHER (ldpc_fast_valid_lanes): the first excluded lane of a
   * row is a multiple of 4 -- nr Z - 32 - k Z with Z % 4 == 0, resp. 4 times the real code's for the interleaved form */
  for (int r = 0; r < d->nrows; r++)
    if (d->pc_lo[r] & 3)
      return; /* (never for NR's lifting sizes; f_ok stays 0: the generic kernel serves the code) */
  d->f_ok = 1;
}

int ldpc_build_code_desc(int BG, int Z, int R, ldpc_code_desc_t *d)
{
  return ldpc_build_code_desc_shape(BG, Z, R, LDPC_SHAPE_THROUGHPUT, d);
}

int ldpc_build_code_desc_shape(int BG, int Z, int R, int shape, ldpc_code_desc_t *d)
{
  const int ils = ldpc_lifting_set_index(Z);
  if (ils < 0)
    return -1;
  memset(d, 0, sizeof(*d));
  const uint8_t *deg, *col;
  const uint16_t *sh;
  if (BG == 1) {
    deg = nr_ldpc_bg1_row_deg; col = nr_ldpc_bg1_col; sh = nr_ldpc_bg1_shift[ils];
    d->ncore = 26; d->kb_full = 22;
    /* nrLDPCdecoder_defs.h:53-57: columns kept per decoder rate mode */
    d->ncols = R == 13 ? 68 : R == 23 ? 35 : R == 89 ? 27 : -1;
    if (R >= LDPC_R_COLS && R - LDPC_R_COLS > d->ncore && R - LDPC_R_COLS <= 68)
      d->ncols = R - LDPC_R_COLS; /* a rate mode cut to its first columns (ldpc_graph.h) */
  } else if (BG == 2) {
    deg = nr_ldpc_bg2_row_deg; col = nr_ldpc_bg2_col; sh = nr_ldpc_bg2_shift[ils];
    d->ncore = 14; d->kb_full = 10;
    /* nrLDPCdecoder_defs.h:80-84 */
    d->ncols = R == 15 ? 52 : R == 13 ? 32 : R == 23 ? 17 : -1;
    if (R >= LDPC_R_COLS && R - LDPC_R_COLS > d->ncore && R - LDPC_R_COLS <= 52)
      d->ncols = R - LDPC_R_COLS;
  } else
    return -1;
  if (d->ncols < 0)
    return -1;
  d->BG = BG; d->Z = Z; d->R = R; d->ils = ils;
  d->nrows = d->ncols - d->kb_full;
  d->zw = (Z + 63) / 64;
  d->num_llr = d->ncols * Z;

  int e = 0;
  for (int r = 0; r < d->nrows; r++) {
    d->row_ptr[r] = e;
    d->row_deg[r] = deg[r];
    for (int k = 0; k < deg[r]; k++, e++) {
      d->e_col[e] = col[e];
      d->e_info[e] = ((uint32_t)(col[e] * Z) << 16) | (uint32_t)(sh[e] % Z);
      d->enc_et[e] = ((uint32_t)col[e] << 16) | (uint32_t)(sh[e] % Z);
    }
  }
  d->row_ptr[d->nrows] = e;
  d->nedges = e;

  /* bit-node adjacency of the core columns */
  int n = 0;
  for (int c = 0; c < d->ncore; c++) {
    d->col_ptr[c] = n;
    for (int k = 0; k < d->nedges; k++)
      if (d->e_col[k] == c)
        d->col_edge[n++] = ((uint32_t)k << 16) | (d->e_info[k] & 0xffffu);
  }
  d->col_ptr[d->ncore] = n;

  /* [F6] parity-check lane exclusion.  The reference walks the CNs of one degree class back to back
   * (class = CN group, ascending degree; CNs in base-graph row order, Z lanes each) in 32-lane chunks
   * and ignores the last chunk when the class holds a multiple of 32 lanes
   * (nrLDPC_cnProc.h:937-965, same pattern for every group of BG1 and BG2). */
  for (int r = 0; r < d->nrows; r++)
    d->pc_lo[r] = Z;
  for (int dg = 1; dg <= 19; dg++) {
    int rows[LDPC_MAX_ROWS], nr = 0;
    for (int r = 0; r < d->nrows; r++)
      if (d->row_deg[r] == dg)
        rows[nr++] = r;
    if (!nr)
      continue;
    const int M = nr * Z;
    if (M & 31)
      continue;
    const int first_dropped = M - 32; /* flattened lane index m = k*Z + t */
    for (int k = 0; k < nr; k++) {
      int lo = first_dropped - k * Z;
      if (lo < 0)
        lo = 0;
      if (lo < Z)
        d->pc_lo[rows[k]] = lo;
    }
  }

  /* schedules (longest work first, then round-robin over the waves of the workgroup) */
  sort_item_t items[LDPC_MAX_ROWS * LDPC_MAX_ZW];
  d->pack = Z <= 32 ? 64 / Z : 1;
  d->zinv16 = (65536 + Z - 1) / Z;
  if (d->pack > 1) {
    for (int x = 0; x < d->ncore * Z + 64; x++) /* the reciprocal must be exact for every lane and bit index */
      if (((x * d->zinv16) >> 16) != x / Z)
        d->pack = 1;
  }
  if (d->pack > 1) {
    /* rows by (core rows first, degree descending); packs of up to `pack` rows of one kind and degree */
    for (int r = 0; r < d->nrows; r++) {
      items[r].id = r;
      items[r].key = (r < 4 ? 1000 : 0) + d->row_deg[r];
    }
    qsort(items, d->nrows, sizeof(items[0]), by_key_desc);
    for (int r = 0; r < d->nrows; r++)
      d->cn_rows[r] = items[r].id;
    int np = 0;
    for (int i = 0; i < d->nrows;) {
      int j = i;
      while (j < d->nrows && j - i < d->pack && items[j].key == items[i].key)
        j++;
      d->cn_order[np++] = i | ((j - i) << 8);
      i = j;
    }
    d->n_cn_slots = np;
    for (int c = 0; c < d->ncore; c++) {
      items[c].id = c;
      items[c].key = d->col_ptr[c + 1] - d->col_ptr[c];
    }
    qsort(items, d->ncore, sizeof(items[0]), by_key_desc);
    for (int c = 0; c < d->ncore; c++)
      d->bn_cols[c] = items[c].id;
    d->n_bn_slots = (d->ncore * Z + 63) / 64;
  } else {
  d->n_cn_slots = d->nrows * d->zw;
  for (int s = 0; s < d->n_cn_slots; s++) {
    items[s].id = s;
    items[s].key = d->row_deg[s / d->zw];
  }
  qsort(items, d->n_cn_slots, sizeof(items[0]), by_key_desc);
  for (int s = 0; s < d->n_cn_slots; s++)
    d->cn_order[s] = ((items[s].id / d->zw) << 4) | (items[s].id % d->zw);
  d->n_bn_slots = d->ncore * d->zw;
  for (int s = 0; s < d->n_bn_slots; s++) {
    const int c = s / d->zw;
    items[s].id = s;
    items[s].key = d->col_ptr[c + 1] - d->col_ptr[c];
  }
  qsort(items, d->n_bn_slots, sizeof(items[0]), by_key_desc);
  for (int s = 0; s < d->n_bn_slots; s++)
    d->bn_order[s] = ((items[s].id / d->zw) << 4) | (items[s].id % d->zw);
  }

  /* encoder core-parity solve order (only meaningful for the full-rate descriptors R13 / R15, but the
   * four core rows are present in every mode) */
  {
    int has[4][4], shf[4][4], cnt[LDPC_MAX_Z];
    memset(has, 0, sizeof(has));
    memset(shf, 0, sizeof(shf));
    memset(cnt, 0, sizeof(cnt));
    for (int r = 0; r < 4; r++)
      for (int k = d->row_ptr[r]; k < d->row_ptr[r + 1]; k++) {
        const int c = d->e_col[k] - d->kb_full;
        if (c >= 0 && c < 4) {
          has[r][c] = 1;
          shf[r][c] = (int)(d->e_info[k] & 0xffffu);
        }
      }
    for (int r = 0; r < 4; r++)
      if (has[r][0])
        cnt[shf[r][0]]++;
    d->enc_p0_shift = -1;
    for (int s = 0; s < Z; s++)
      if (cnt[s] & 1) {
        if (d->enc_p0_shift >= 0)
          return -1;
        d->enc_p0_shift = s;
      }
    if (d->enc_p0_shift < 0)
      return -1;
    int known[4] = {1, 0, 0, 0};
    for (int step = 0; step < 3; step++) {
      int found = 0;
      for (int r = 0; r < 4 && !found; r++) {
        int unk = -1, nunk = 0;
        for (int j = 0; j < 4; j++)
          if (has[r][j] && !known[j]) {
            unk = j;
            nunk++;
          }
        if (nunk != 1)
          continue;
        d->enc_row[step] = r;
        d->enc_unk[step] = unk;
        d->enc_ushift[step] = shf[r][unk];
        int nk = 0;
        for (int j = 0; j < 4; j++)
          if (has[r][j] && known[j]) {
            d->enc_kcol[step][nk] = j;
            d->enc_kshift[step][nk] = shf[r][j];
            nk++;
          }
        d->enc_nk[step] = nk;
        known[unk] = 1;
        found = 1;
      }
      if (!found)
        return -1;
    }
    /* the same three steps with every known column replaced by its own expression: sums of (source, shift) terms over
     * the lambdas (sources 0..3) and p0 (source 4) only; a term met twice cancels */
    {
      int n[4] = {1, 0, 0, 0}, src[4][32], sh[4][32];
      src[0][0] = 4;
      sh[0][0] = 0;
      for (int step = 0; step < 3; step++) {
        const int u = d->enc_unk[step], us = d->enc_ushift[step] % Z;
        int tn = 0, tsrc[32], tsh[32];
        tsrc[tn] = d->enc_row[step];
        tsh[tn++] = (Z - us) % Z;
        for (int k = 0; k < d->enc_nk[step]; k++) {
          const int kc = d->enc_kcol[step][k], ks = d->enc_kshift[step][k] % Z;
          for (int m = 0; m < n[kc]; m++) {
            if (tn >= 32)
              return -1;
            tsrc[tn] = src[kc][m];
            tsh[tn++] = (sh[kc][m] + ks + Z - us) % Z;
          }
        }
        n[u] = 0;
        for (int a = 0; a < tn; a++) { /* keep the terms that occur an odd number of times */
          int cnt2 = 0, first = 1;
          for (int b = 0; b < tn; b++)
            if (tsrc[b] == tsrc[a] && tsh[b] == tsh[a]) {
              cnt2++;
              if (b < a)
                first = 0;
            }
          if (first && (cnt2 & 1)) {
            src[u][n[u]] = tsrc[a];
            sh[u][n[u]++] = tsh[a];
          }
        }
        if (n[u] > 8)
          return -1;
        for (int m = 0; m < 8; m++)
          d->enc_x_term[step][m] = m < n[u] ? ((uint32_t)src[u][m] << 16) | (uint32_t)sh[u][m] : 0xffffffffu;
      }
    }
  }

  /* LDS carve-up of the generic kernel: messages, APP of the core columns, channel LLRs, flags */
  d->lds_r = 0;
  d->lds_app = align16(d->nedges * Z);
  d->lds_llr = d->lds_app + align16(d->ncore * Z);
  d->lds_misc = d->lds_llr + align16(d->num_llr);
  d->lds_total = d->lds_misc + 64;
  int waves = (d->n_cn_slots + 3) / 4; /* aim at >= 4 slots per wave and phase */
  if (waves < 1) waves = 1;
  if (waves > 16) waves = 16;
  d->n_threads = waves * 64;
  {
    const char *e = getenv("NRLDPC_HIP_PAIR19"); /* tuning knob: 0 = one lane per degree-19 row item */
    build_fast_section(d, shape, 1, !(e && e[0] == '0'));
  }
  return 0;
}

int ldpc_multi_blocks_for(int Z)
{
  if ((Z & 3) || Z < 8 || Z > 64)
    return 1;
  int mb = 64 / (Z / 4); /* one lifted row of all blocks ~ one 64-item task */
  if (mb > 16)
    mb = 16;
  return mb < 2 ? 1 : mb;
}

int ldpc_build_code_desc_multi(int BG, int Z, int R, int mb, ldpc_code_desc_t *d)
{
  if (ldpc_build_code_desc_shape(BG, Z, R, LDPC_SHAPE_THROUGHPUT, d) != 0)
    return -1;
  if (mb < 2 || mb > 16) {
    d->f_ok = 0;
    return 0;
  }
  build_fast_section(d, LDPC_SHAPE_THROUGHPUT, mb, 0);
  return 0;
}

int ldpc_build_code_desc_interleaved(int BG, int Z, int R, int mb, ldpc_code_desc_t *d)
{
  if (ldpc_build_code_desc_shape(BG, Z, R, LDPC_SHAPE_THROUGHPUT, d) != 0)
    return -1;
  d->f_ok = 0;
  if (mb < 1 || mb > 16 || 4 * Z > LDPC_MAX_Z)
    return 0;
  /* the virtual code: lane 4t + i = lane t of block i; a shift by s lanes of every block = a shift by 4s virtual lanes */
  for (int e = 0; e < d->nedges; e++)
    d->e_info[e] = (d->e_info[e] & 0xffff0000u) | (4u * (d->e_info[e] & 0xffffu));
  for (int k = 0; k < d->col_ptr[d->ncore]; k++)
    d->col_edge[k] = (d->col_edge[k] & 0xffff0000u) | (4u * (d->col_edge[k] & 0xffffu));
  for (int r = 0; r < d->nrows; r++)
    d->pc_lo[r] *= 4;
  d->Z = 4 * Z; /* (num_llr, ncols, zw and the generic kernel's fields keep describing the real code) */
  d->f_sub = 4;
  build_fast_section(d, LDPC_SHAPE_THROUGHPUT, mb, 0);
  return 0;
}
