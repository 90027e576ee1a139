/*
 * ldpc_graph.h -- lifted-code descriptor shared by the host table builder and the HIP kernels.
 *
 * One descriptor per (BG, Zc, decoder-rate mode R): the rows/columns the reference decoder processes
 * in that mode (reference: nrLDPC_decoder/nrLDPC_init.h:58-160, nrLDPCdecoder_defs.h:53-57,80-84),
 * the edge list in CSR form with the 38.212 shifts already reduced mod Zc, the column (bit-node)
 * adjacency, and the work schedules the kernels walk.  It replaces the reference's per-(BG,Z,R)
 * LUT set (nrLDPC_lut.h: circShift_*, startAddrBnProcBuf_*, bnPosBnProcBuf_*, posBnInCnProcBuf_*):
 * those tables describe copies between a CN-ordered and a BN-ordered message buffer; this design has
 * a single message array addressed by (edge, lane) and does the circular shift as index arithmetic.
 */
#ifndef LDPC_GRAPH_H
#define LDPC_GRAPH_H
#include <stdint.h>

#define LDPC_MAX_ROWS 46
#define LDPC_MAX_COLS 68
#define LDPC_MAX_CORE 26
#define LDPC_MAX_EDGES 316
#define LDPC_MAX_Z 384
#define LDPC_MAX_ZW 6 /* ceil(384/64) wavefront-wide chunks per lifted row */
#define LDPC_F_MAX_WAVES 16
#define LDPC_F_DEFAULT_WAVES 16 /* waves per workgroup the fast kernel uses for large codes */
#define LDPC_F_MAX_CN_TASKS 96
#define LDPC_F_MAX_BN_TASKS 48
#define LDPC_F_BN_SHORT 6 /* columns up to this degree count as short */
#define LDPC_F_CN_DOUBLE 5 /* extension rows up to this degree come as tasks of 128 items, two per thread (f_cn_task) */
#define LDPC_F_BN_GROUP 5 /* short tasks per ticket at most */
#define LDPC_F_MAX_CTBL 800 /* <= 26 columns x degree 30 when every list is padded to the maximum */

typedef struct ldpc_code_desc {
  int32_t BG, Z, R, ils;
  int32_t nrows;  /* base-graph rows processed in this rate mode */
  int32_t ncols;  /* base-graph columns processed; numLLR = ncols*Z */
  int32_t ncore;  /* 26 / 14: columns >= ncore are the degree-1 (extension parity) columns */
  int32_t kb_full; /* 22 / 10 */
  int32_t nedges;
  int32_t zw;     /* ceil(Z/64) */
  int32_t num_llr; /* ncols*Z */
  int32_t n_cn_slots; /* nrows*zw  (row, 64-lane chunk) work items of the check-node phase */
  int32_t n_bn_slots; /* ncore*zw  (column, chunk) work items of the bit-node phase */
  /* LDS carve-up (bytes, 16-aligned) used by the generic decoder kernel */
  int32_t lds_r, lds_app, lds_llr, lds_misc, lds_total;
  int32_t n_threads; /* workgroup size the generic decoder kernel is launched with for this code */
  /* Small lifting sizes (Z <= 32): a 64-lane step of the generic kernel takes pack = 64 / Z lifted rows (of equal degree)
   * resp. 64 consecutive bits of the degree-sorted core columns instead of one row / column, lane l -> (l / Z, l % Z).
   * pack = 1: one (row | column, 64-lane chunk) per step.  zinv16 = ceil(2^16 / Z): (x * zinv16) >> 16 = x / Z for the
   * indices that occur (checked on the host). */
  int32_t pack, zinv16;
  int32_t cn_rows[LDPC_MAX_ROWS + 2];  /* pack > 1: rows sorted by (core/extension, degree); cn_order[k] = first | count << 8 */
  int32_t bn_cols[LDPC_MAX_CORE + 2];  /* pack > 1: core columns sorted by degree, descending; slot k = bits 64k .. 64k+63 */

  /* All tables are 32-bit so that wave-uniform lookups compile to scalar (s_load_dword) loads. */
  int32_t row_ptr[LDPC_MAX_ROWS + 2];
  int32_t row_deg[LDPC_MAX_ROWS + 2];
  /* [F6] reference parity-check quirk (nrLDPC_cnProc.h:964-965): lanes t >= pc_lo[row] of a lifted row
   * are not looked at by the early-stop parity check (Z when every lane is checked). */
  int32_t pc_lo[LDPC_MAX_ROWS + 2];
  /* per edge (row-major): lo16 = shift mod Z, hi16 = col*Z (byte offset of the column in a [col][Z] array) */
  uint32_t e_info[LDPC_MAX_EDGES + 4];
  int32_t e_col[LDPC_MAX_EDGES + 4];
  uint32_t enc_et[LDPC_MAX_EDGES + 4]; /* per edge: column << 16 | shift mod Z (the bit-packed encoder's edge table, one load per edge) */
  /* core columns (c < ncore): edges touching the column, as (edge << 16) | shift */
  int32_t col_ptr[LDPC_MAX_CORE + 2];
  uint32_t col_edge[LDPC_MAX_EDGES + 4];
  /* schedules: (row << 4) | chunk resp. (col << 4) | chunk, sorted by work (degree) descending so that
   * round-robin over the waves of a workgroup balances */
  int32_t cn_order[LDPC_MAX_ROWS * LDPC_MAX_ZW + 4];
  int32_t bn_order[LDPC_MAX_CORE * LDPC_MAX_ZW + 4];
  /* encoder: solve order of the 4 core parity columns kb_full..kb_full+3 (38.212 5.3.2 dual diagonal).
   * p0 follows from the sum of the four core rows: sum[t] = p0[(t + enc_p0_shift) mod Z]; then three
   * steps, each closing one core row that has exactly one unknown parity column left:
   *   acc[t] = lambda_row[t] ^ XOR_k p_{kcol[k]}[(t + kshift[k]) mod Z];  p_unk[(t + ushift) mod Z] = acc[t] */
  int32_t enc_p0_shift;
  int32_t enc_row[3], enc_unk[3], enc_ushift[3], enc_nk[3], enc_kcol[3][4], enc_kshift[3][4];

  /* ---- "fast" decoder kernel: 4 consecutive lanes per thread, messages as biased bytes (v + 128) packed
   * four to a dword, arithmetic on packed 16-bit pairs.  Usable when f_ok (Z % 4 == 0, Z >= 8, fits LDS).
   * Work item = (lifted row, 4-lane group) resp. (core column, 4-lane group); Z/4 items per row/column.
   * Rows are sorted by degree and their items flattened per degree group, columns likewise; a task is 64
   * consecutive items (one wavefront, uniform degree / loop bound); tasks are spread over the waves of
   * the workgroup by longest-processing-time-first. */
  int32_t f_ok;
  int32_t f_zq;        /* Z/4 */
  uint32_t f_zq_magic; /* ceil(2^32 / zq): i / zq == umulhi(i, magic) for every item index i that occurs */
  int32_t f_rstride;   /* Z + 4: bytes per message row, the first 4 bytes repeated at the end (wrap-around window) */
  int32_t f_astride;   /* 2Z: bytes per APP row, stored twice back to back so that (t + shift) needs no modulo */
  int32_t f_lds_r, f_lds_app, f_lds_ext, f_lds_etbl, f_lds_ctbl, f_lds_rowtbl, f_lds_coltbl, f_lds_misc, f_lds_total;
  int32_t f_n_threads;
  int32_t f_n_cn_tasks, f_n_bn_tasks;
  /* The kernel's waves draw the tasks of a phase in id order (= most expensive first) from a queue, an LDS counter:
   * static per-wave shares leave the SIMDs' oldest waves idle early, because the issue arbiter favours them and equal
   * shares do not finish together. */
  /* CN task: {degree (| 0x100: a DOUBLE task), has_ext, first item, group first item, group end item, sorted-row index of
   * the group's first row}.  A double task is 128 consecutive items of a low-degree group and a thread takes items lane and
   * lane + 64 TOGETHER (ldpc_fast_cn_ps2): the chain of round trips in front of a task -- ticket, record, row record, edge
   * table, windows -- is longer than the arithmetic of a degree-3..5 row, and two items share one chain. */
  int32_t f_cn_task[LDPC_F_MAX_CN_TASKS][6];
  /* BN task: {first item, end item (all columns), loop bound = degree of the first item's column} */
  int32_t f_bn_task[LDPC_F_MAX_BN_TASKS][3];
  /* Bit-node queue tickets (one-block kernels): {first task, tasks}.  A task of high-degree columns is a ticket of its own;
   * tasks of columns with at most LDPC_F_BN_SHORT edges -- where the fetch chain in front of a task (ticket, task record,
   * column record, table entry, window) outweighs the few edges -- come up to LDPC_F_BN_GROUP to a ticket, and a thread
   * walks one item of each of them edge by edge together (ldpc_fast_bn_multi): that many chains in flight instead of one.
   * The lists of ALL short columns are padded to the degree of the largest short column, so the walk needs no per-item bound. */
  int32_t f_bn_ticket[LDPC_F_MAX_BN_TASKS][2];
  int32_t f_n_bn_tickets;
  /* tables the kernel copies into LDS (read per lane): */
  uint32_t f_rowtbl[LDPC_MAX_ROWS + 2];  /* per sorted row: first edge (9 bits) | degree << 9 | has extension column << 14 | pc_lo << 16 */
  uint32_t f_etbl[LDPC_MAX_EDGES + 4];   /* per edge: LDS byte offset of the neighbour's data: core column: f_lds_app + col*astride + shift;
                                            extension column: f_lds_ext + (col-ncore)*Z, or col*Z = byte offset in the
                                            block's LLR input when f_ext_global */
  uint32_t f_coltbl[LDPC_MAX_CORE + 2];  /* per sorted column: col | degree << 8 | first entry in f_ctbl << 16 */
  /* per (sorted column, k): two dwords {x = Z - shift, LDS byte offset of the edge's message row - (x & 3)}.  A column's list is
   * padded up to the degree of the earliest column it can share a task with; padding entries {Z, f_lds_zero} point
   * at a row of zero bytes, so short columns need no predication in the gather loop. */
  uint32_t f_ctbl[2 * LDPC_F_MAX_CTBL];
  int32_t f_n_ctbl;   /* entries (pairs) used */
  int32_t f_wg_per_cu;  /* workgroups of this shape resident on one CU (16 wave slots, 160 KiB LDS) */
  int32_t f_bn_group;   /* bit-node tasks per queue ticket (>= 1) */
  int32_t f_ext_global; /* 1: the degree-1 columns' LLRs are read from the input buffer, not staged in LDS */
  int32_t f_lds_zero; /* Z + 4 zero bytes */
  /* Several code blocks per workgroup (small lifting sizes: a lifted row has only Z/4 items, so one block's degree groups
   * leave most of a 64-item task empty).  f_mb > 1: the workgroup's LDS holds f_mb blocks side by side INSIDE every row --
   * message / extension-LLR row of an edge = f_mb x (Z + 4) bytes, APP row of a column = f_mb x 2Z bytes -- and an item
   * is (row | column, block, 4-lane group): f_zqb = f_mb * Z/4 items per row.  f_rstride / f_astride / f_etbl / f_ctbl
   * and the task lists are built for that layout; the per-item bodies get the block's offsets inside a row (ldpc_dec_fast_core.h,
   * boff_r / boff_a).  Kernel: ldpc_dec_fast_mblock.h. */
  int32_t f_mb, f_zqb;
  uint32_t f_zqb_magic;
  /* f_sub = 4: lifting sizes that are not multiples of 4.  FOUR code blocks are interleaved byte-wise -- byte i of the dword
   * at (row, lane t) belongs to block i -- which is the same as ONE block of a code with lifting size 4 Zc whose shifts are
   * 4 s: every window is an aligned dword, whatever Zc is.  The descriptor then describes that virtual code (Z = 4 Zc,
   * shifts and pc_lo times 4; num_llr, ncols stay the real code's) and f_mb counts groups of four.  f_lds_llr: the core
   * columns' channel LLRs in the interleaved layout (the bit-node phase reads them every pass).  f_sub = 1 otherwise. */
  int32_t f_sub, f_lds_llr;
  /* 1: the items of the degree-19 rows come in PAIRS of neighbouring lanes, each lane taking half of the row's edges
   * (ldpc_fast_cn19_pair): the group holds 2 x rows x Z/4 items, item = 2 x (row item) + half */
  int32_t f_pair19;
  /* per bit-node ticket {first item, end item, loop bound, tasks}: ticket and task record in one scalar load.  (Carrying
   * the column records of a ticket -- or the row records of a check-node task -- in its record, instead of looking them
   * up in LDS, was slower: profiles/r04/decoder_ab26_record_carried_lookups.txt) */
  int32_t f_bn_rec[LDPC_F_MAX_BN_TASKS][4];
  /* encoder, the three solve steps above in CLOSED form: core parity column enc_unk[st] as a sum of shifted copies of the
   * four lambdas and of p0 alone -- p_unk[u] = XOR_m src_m[(u + shift_m) mod Z], enc_x_term[st][m] = src << 16 | shift with
   * src 0..3 = lambda of core row src, 4 = p0; unused entries 0xffffffff.  With these the three columns no longer wait for
   * each other: one step after p0 instead of three (ldpc_enc_packed32.h). */
  uint32_t enc_x_term[3][8];
} ldpc_code_desc_t;

#ifdef __cplusplus
extern "C" {
#endif
/* Fill *d for (BG, Z, R).  Returns 0, or -1 if (BG, Z, R) is not a valid NR LDPC configuration.
 * R = a decoder rate mode of the reference (13 / 23 / 89 resp. 15 / 13 / 23: nrLDPCdecoder_defs.h:53-57, 80-84), or -- internal,
 * never accepted from a caller -- LDPC_R_COLS + n: the code cut to its first n columns (ncore < n <= all of them), i.e. the
 * first n - kb_full rows.  The transport-block chain decodes a segment on such a graph when every row it drops closes on a
 * degree-1 column whose channel LLRs are ALL ZERO (never transmitted): the check node's input from that column has magnitude 0,
 * so every message the row sends to the core columns is 0 (cnProc.h:105-114: min over the OTHER inputs), in every pass -- the
 * core columns' sums, the hard decisions of the information bits and the CRC verdict of every pass are those of the rate mode
 * the reference would have run.  (Not so the parity check of such a row: only launches that stop on the CRC may use it.) */
#define LDPC_R_COLS 1000
int ldpc_build_code_desc(int BG, int Z, int R, ldpc_code_desc_t *d); /* = throughput shape */
/* The fast kernel's workgroup shape (waves per workgroup, LLR staging; everything else is identical):
 * THROUGHPUT fills a CU's 16 wave slots with as many workgroups as its LDS admits -- for launches of more than one
 * workgroup round; LATENCY gives one block as many waves as it has tasks -- for small launches. */
#define LDPC_SHAPE_THROUGHPUT 0
#define LDPC_SHAPE_LATENCY 1
int ldpc_build_code_desc_shape(int BG, int Z, int R, int shape, ldpc_code_desc_t *d);
/* throughput shape with `mb` blocks per workgroup (f_ok = 0 when the code does not qualify or does not fit) */
int ldpc_build_code_desc_multi(int BG, int Z, int R, int mb, ldpc_code_desc_t *d);
/* any Z: four blocks interleaved byte-wise (f_sub = 4), `mb` such groups per workgroup */
int ldpc_build_code_desc_interleaved(int BG, int Z, int R, int mb, ldpc_code_desc_t *d);
/* blocks per workgroup that fill a 64-item task row by row for this lifting size; 1: not worth it */
int ldpc_multi_blocks_for(int Z);
/* set index iLS of lifting size Z (38.212 Table 5.3.2-1), -1 if Z is not a lifting size */
int ldpc_lifting_set_index(int Z);
#ifdef __cplusplus
}
#endif
#endif
