/*
 * oracle_ldpc_decoder_vec.c -- TEST INFRASTRUCTURE ONLY (see oracle_nr_coding.h for the pin status).
 *
 * The same decoder as oracle_ldpc_decoder.c (reference nrLDPC_decoder/nrLDPC_decoder.c:206-880, rules D1-D9, F5,
 * F6 listed there), restated a second time so that it runs at a useful speed on a CPU:
 *   - the check node keeps the two smallest magnitudes and the sign parity per lane instead of recomputing the
 *     minimum over "all the others" for every edge (O(d) instead of O(d^2); identical results: the minimum over the
 *     others is m2 for an edge that holds the minimum, m1 otherwise, and a zero input makes the magnitude 0, which
 *     is where the reference's sign_epi8(.,0) = 0 lands as well);
 *   - every inner loop runs over the Zc lanes of one edge on contiguous int8 data (the circular shift becomes two
 *     contiguous pieces), which gcc vectorises; the hot functions are cloned for AVX-512BW / AVX2 / baseline x86-64
 *     and dispatched at load time (target_clones), so one .so serves the build container and the GPU box.
 * It exists for two users: bench.py's cpu_baseline leg, and GPU parity tests that compare whole 1024-block batches.
 * tests/test_oracle.py checks it bit for bit (outputs and pass counts) against the scalar restatement for every code.
 */
#define _GNU_SOURCE
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "oracle_nr_coding.h"

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define ORACLE_CLONES __attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))
#else
#define ORACLE_CLONES
#endif

typedef struct {
  int n;         /* degree groups */
  int deg[46];   /* ascending */
  int nrow[46];  /* rows per group */
  int rows[46][46];
} pc_groups_t;

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static void build_groups(const oracle_graph_t *g, pc_groups_t *pg)
{
  pg->n = 0;
  for (int row = 0; row < g->nrows; row++) {
    const int d = g->row_ptr[row + 1] - g->row_ptr[row];
    int seen = 0;
    for (int i = 0; i < pg->n; i++)
      seen |= pg->deg[i] == d;
    if (!seen)
      pg->deg[pg->n++] = d;
  }
  qsort(pg->deg, pg->n, sizeof(int), cmp_int);
  for (int gi = 0; gi < pg->n; gi++) {
    pg->nrow[gi] = 0;
    for (int row = 0; row < g->nrows; row++)
      if (g->row_ptr[row + 1] - g->row_ptr[row] == pg->deg[gi])
        pg->rows[gi][pg->nrow[gi]++] = row;
  }
}

/* [D2] */
ORACLE_CLONES static void cn_pass_vec(const oracle_graph_t *g, const int8_t *restrict q, int8_t *restrict r)
{
  const int Z = g->Z;
  uint8_t m1[384], m2[384], neg[384];
  for (int row = 0; row < g->nrows; row++) {
    const int e0 = g->row_ptr[row], e1 = g->row_ptr[row + 1];
    for (int t = 0; t < Z; t++) {
      m1[t] = 255;
      m2[t] = 255;
      neg[t] = 0;
    }
    for (int k = e0; k < e1; k++) {
      const int8_t *qk = q + k * Z;
      for (int t = 0; t < Z; t++) {
        const int8_t v = qk[t];
        const uint8_t a = (uint8_t)(v < 0 ? -v : v); /* 128 for -128 */
        const uint8_t hi = a > m1[t] ? a : m1[t];
        m2[t] = hi < m2[t] ? hi : m2[t];
        m1[t] = a < m1[t] ? a : m1[t];
        neg[t] ^= (uint8_t)(v < 0);
      }
    }
    for (int j = e0; j < e1; j++) {
      const int8_t *qj = q + j * Z;
      int8_t *rj = r + j * Z;
      for (int t = 0; t < Z; t++) {
        const int8_t v = qj[t];
        const uint8_t a = (uint8_t)(v < 0 ? -v : v);
        uint8_t mag = a == m1[t] ? m2[t] : m1[t];
        mag = mag > 127 ? 127 : mag;
        const uint8_t s = neg[t] ^ (uint8_t)(v < 0);
        rj[t] = (int8_t)(s ? -(int8_t)mag : (int8_t)mag);
      }
    }
  }
}

/* [D3]-[D6] */
ORACLE_CLONES static void bn_pass_vec(const oracle_graph_t *g, const int8_t *restrict llr, const int8_t *restrict r,
                                      int8_t *restrict app, int8_t *restrict q, int16_t *restrict acc)
{
  const int Z = g->Z, n = g->ncore * Z;
  for (int i = 0; i < n; i++)
    acc[i] = llr[i];
  for (int e = 0; e < g->nedges; e++) {
    const int c = g->col[e], s = g->shift[e];
    if (c >= g->ncore)
      continue;
    const int8_t *re = r + e * Z;
    int16_t *a = acc + c * Z;
    for (int t = 0; t < Z - s; t++)
      a[s + t] += re[t];
    for (int t = Z - s; t < Z; t++)
      a[t - (Z - s)] += re[t];
  }
  for (int i = 0; i < n; i++) {
    const int16_t v = acc[i];
    app[i] = (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
  }
  for (int e = 0; e < g->nedges; e++) {
    const int c = g->col[e], s = g->shift[e];
    if (c >= g->ncore)
      continue;
    const int8_t *re = r + e * Z, *ap = app + c * Z;
    int8_t *qe = q + e * Z;
    for (int t = 0; t < Z - s; t++) {
      const int16_t v = (int16_t)(ap[s + t] - re[t]);
      qe[t] = (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
    }
    for (int t = Z - s; t < Z; t++) {
      const int16_t v = (int16_t)(ap[t - (Z - s)] - re[t]);
      qe[t] = (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v));
    }
  }
}

/* [D7] + [F6] */
ORACLE_CLONES static uint32_t parity_check_vec(const oracle_graph_t *g, const pc_groups_t *pg, const int8_t *restrict q,
                                               const int8_t *restrict r)
{
  const int Z = g->Z;
  uint8_t par[384];
  uint32_t res = 0;
  for (int gi = 0; gi < pg->n; gi++) {
    const int M = pg->nrow[gi] * Z;
    const int checked = (M & 31) ? M : M - 32; /* [F6] the last 32-lane chunk is dropped when M % 32 == 0 */
    for (int ri = 0; ri < pg->nrow[gi]; ri++) {
      const int row = pg->rows[gi][ri];
      memset(par, 0, (size_t)Z);
      for (int e = g->row_ptr[row]; e < g->row_ptr[row + 1]; e++) {
        const int8_t *qe = q + e * Z, *re = r + e * Z;
        for (int t = 0; t < Z; t++) {
          const int16_t v = (int16_t)(qe[t] + re[t]);
          par[t] ^= (uint8_t)(v < 0); /* sign of the saturated sum = sign of the sum */
        }
      }
      int hi = checked - ri * Z;
      hi = hi > Z ? Z : hi;
      uint8_t any = 0;
      for (int t = 0; t < hi; t++)
        any |= par[t];
      res |= any;
    }
  }
  return res;
}

/* [D9] (same as the scalar file's write_output) */
static void write_output_vec(const oracle_graph_t *g, int outMode, const int8_t *app, int8_t *p_out)
{
  const int Z = g->Z, numLLR = g->ncols * Z, ncz = g->ncore * Z;
  if (outMode != ORACLE_OUT_BIT) {
    for (int i = 0; i < numLLR; i++)
      p_out[i] = i < ncz ? app[i] < 0 : 0; /* parity columns keep 0 [F5]; LLRINT8 == BITINT8 (see the scalar file) */
  } else {
    const int nwords = (numLLR + 31) >> 5;
    uint8_t *o = (uint8_t *)p_out;
    memset(o, 0, (size_t)nwords * 4);
    for (int i = 0; i < ncz; i++)
      if (app[i] < 0)
        o[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
  }
}

int oracle_ldpc_decode_vec(int BG, int Z, int R, int numMaxIter, int outMode, int use_crc, int E, int crc_type,
                           const int8_t *p_llr, int8_t *p_out)
{
  oracle_graph_t g;
  if (oracle_ldpc_graph(BG, Z, R, &g) != 0)
    return -1;
  pc_groups_t pg;
  build_groups(&g, &pg);
  int8_t q[316 * 384], r[316 * 384], app[26 * 384];
  int16_t acc[26 * 384];
  memset(r, 0, (size_t)g.nedges * Z);
  memset(app, 0, (size_t)g.ncore * Z);
  for (int e = 0; e < g.nedges; e++) { /* [D1] */
    const int s = g.shift[e];
    const int8_t *src = p_llr + g.col[e] * Z;
    memcpy(q + e * Z, src + s, (size_t)(Z - s));
    memcpy(q + e * Z + (Z - s), src, (size_t)s);
  }
  cn_pass_vec(&g, q, r);
  bn_pass_vec(&g, p_llr, r, app, q, acc);
  uint32_t numIter = 1, pcRes = 1;
  while (numIter <= (uint32_t)numMaxIter && pcRes != 0) {
    numIter++;
    cn_pass_vec(&g, q, r);
    bn_pass_vec(&g, p_llr, r, app, q, acc);
    if (!use_crc) {
      pcRes = parity_check_vec(&g, &pg, q, r);
    } else if (numIter > 2) { /* [D8] */
      write_output_vec(&g, outMode, app, p_out);
      if (oracle_check_crc((uint8_t *)p_out, (uint32_t)E, (uint8_t)crc_type))
        break;
    }
  }
  if (!use_crc)
    write_output_vec(&g, outMode, app, p_out);
  return (int)numIter;
}

/* block b decoded by thread b % nthreads; PC-stop mode, BIT output (cf. oracle_ldpc_decode_mt) */
typedef struct {
  int tid, nthreads, nblocks, BG, Z, R, numMaxIter;
  const int8_t *llr;
  int llr_stride;
  int8_t *out;
  int out_stride;
  int *iters;
  int cpu; /* CPU to pin to, -1: none */
} mtv_arg_t;
static void *mtv_worker(void *p)
{
  mtv_arg_t *a = (mtv_arg_t *)p;
  if (a->cpu >= 0) { /* one worker per allowed CPU: short-lived threads are otherwise not always spread by the scheduler */
    cpu_set_t one;
    CPU_ZERO(&one);
    CPU_SET(a->cpu, &one);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
  }
  for (int b = a->tid; b < a->nblocks; b += a->nthreads)
    a->iters[b] = oracle_ldpc_decode_vec(a->BG, a->Z, a->R, a->numMaxIter, ORACLE_OUT_BIT, 0, 0, 0,
                                         a->llr + (size_t)b * a->llr_stride, a->out + (size_t)b * a->out_stride);
  return NULL;
}
int oracle_ldpc_decode_vec_mt(int nthreads, int nblocks, int BG, int Z, int R, int numMaxIter, const int8_t *llr,
                              int llr_stride, int8_t *out, int out_stride, int *iters)
{
  if (nthreads < 1 || nthreads > 1024)
    return -1;
  pthread_t th[1024];
  mtv_arg_t args[1024];
  cpu_set_t allowed;
  int cpus[1024], ncpu = 0;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
    for (int c = 0; c < CPU_SETSIZE && ncpu < 1024; c++)
      if (CPU_ISSET(c, &allowed))
        cpus[ncpu++] = c;
  for (int t = 0; t < nthreads; t++) {
    args[t] = (mtv_arg_t){t, nthreads, nblocks, BG, Z, R, numMaxIter, llr, llr_stride, out, out_stride, iters,
                          ncpu > 0 ? cpus[t % ncpu] : -1};
    if (pthread_create(&th[t], NULL, mtv_worker, &args[t]) != 0)
      return -1;
  }
  for (int t = 0; t < nthreads; t++)
    pthread_join(th[t], NULL);
  return 0;
}
