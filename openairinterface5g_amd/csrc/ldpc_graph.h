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

  /* All tables are 32-bit so that wave-uniform lookups compile to scalar (s_load_dword) loads. */
  int32_t row_ptr[LDPC_MAX_ROWS + 2];
  int32_t row_deg[LDPC_MAX_ROWS + 2];
  /* [F6] reference parity-check quirk (nrLDPC_cnProc.h:964-965): lanes t >= pc_lo[row] of a lifted row
   * are not looked at by the early-stop parity check (Z when every lane is checked). */
  int32_t pc_lo[LDPC_MAX_ROWS + 2];
  /* per edge (row-major): lo16 = shift mod Z, hi16 = col*Z (byte offset of the column in a [col][Z] array) */
  uint32_t e_info[LDPC_MAX_EDGES + 4];
  int32_t e_col[LDPC_MAX_EDGES + 4];
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
} ldpc_code_desc_t;

#ifdef __cplusplus
extern "C" {
#endif
/* Fill *d for (BG, Z, R).  Returns 0, or -1 if (BG, Z, R) is not a valid NR LDPC configuration. */
int ldpc_build_code_desc(int BG, int Z, int R, ldpc_code_desc_t *d);
/* set index iLS of lifting size Z (38.212 Table 5.3.2-1), -1 if Z is not a lifting size */
int ldpc_lifting_set_index(int Z);
#ifdef __cplusplus
}
#endif
#endif
