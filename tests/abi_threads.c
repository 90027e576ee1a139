/*
 * abi_threads.c -- N pthreads hammering the reference's per-segment entry point LDPCdecoder() of libldpc_hip.so, the
 * way the reference's thread-pool workers do (openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:435-468), each call with
 * its own code / iteration cap / stop mode.  Checks that every concurrent call returns exactly what the same call
 * returned single-threaded (outputs and pass counts), and reports calls per second.  Test infrastructure.
 *
 *   gcc -O2 -I include tests/abi_threads.c -o abi_threads -ldl -lpthread
 *   ./abi_threads <path/libldpc_hip.so> <threads> <calls per thread> [case]
 * With a case index every call uses that one case (1 = BG1 Zc=384 R=1/3 at high SNR, 2 passes: the per-call latency
 * figure that corresponds to `ldpctest -l 8448 -s10`'s "Decoding time mean").
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <time.h>
#include "nrLDPC_hip.h"

typedef int32_t (*init_t)(void);
typedef int32_t (*stats_t)(int64_t *);
typedef int32_t (*batch_t)(const nrLDPC_hip_dec_batch_t *);
typedef int32_t (*dec_t)(t_nrLDPC_dec_params *, uint8_t, uint8_t, uint8_t, int8_t *, int8_t *, t_nrLDPC_time_stats *, decode_abort_t *);

#define NCASE 12
static const int cases[NCASE][4] = { /* BG, Z, R, numMaxIter */
  {1, 384, 13, 8}, {1, 384, 13, 8}, {1, 384, 23, 8}, {1, 352, 89, 5}, {2, 64, 15, 8}, {2, 208, 13, 8},
  {1, 96, 13, 8},  {1, 384, 13, 2}, {2, 6, 15, 8},   {1, 18, 13, 8},  {2, 384, 23, 8}, {1, 384, 13, 8}};
static int ncols(int BG, int R) { return BG == 1 ? (R == 13 ? 68 : R == 23 ? 35 : 27) : (R == 15 ? 52 : R == 13 ? 32 : 17); }
typedef int (*crc_t)(uint8_t *, uint32_t, uint8_t);
static crc_t crc_cb; /* the library's nrLDPC_hip_check_crc: the CRC stop that is evaluated on the GPU (a caller's own predicate would be called on the host) */

static dec_t dec;
static int8_t *llr[NCASE];
static uint8_t *expect[NCASE];
static int expect_iter[NCASE], out_len[NCASE];
static t_nrLDPC_dec_params prm[NCASE];
static int calls_per_thread, only_case = -1;
static volatile int failures, nacked;
static int allow_nack; /* ABI_ALLOW_NACK: a call may come back as "not decoded" (numMaxIter + 1, p_out untouched): fault-injection runs */
static int progress;
static struct timespec t_start;

static void *worker(void *arg)
{
  const int tid = (int)(long)arg;
  uint8_t *out = malloc(68 * 384);
  for (int i = 0; i < calls_per_thread; i++) {
    const int c = only_case >= 0 ? only_case : (tid * 7 + i) % NCASE;
    t_nrLDPC_dec_params p = prm[c];
    memset(out, 0xA5, out_len[c]);
    if (progress && i % 100 == 0) {
      struct timespec t;
      clock_gettime(CLOCK_MONOTONIC, &t);
      fprintf(stderr, "thread %d call %d at %.3f ms\n", tid, i, (t.tv_sec - t_start.tv_sec) * 1e3 + (t.tv_nsec - t_start.tv_nsec) / 1e6);
    }
    const int n = dec(&p, 0, 0, 0, llr[c], (int8_t *)out, NULL, NULL);
    if (allow_nack && n == prm[c].numMaxIter + 1) { /* (also for a case that fails to decode anyway: a real failure writes p_out) */
      int touched = 0;
      for (int k = 0; k < out_len[c]; k++) touched |= out[k] != 0xA5;
      if (!touched) { __sync_fetch_and_add(&nacked, 1); continue; }
    }
    if (n != expect_iter[c] || memcmp(out, expect[c], out_len[c]) != 0) {
      __sync_fetch_and_add(&failures, 1);
      fprintf(stderr, "thread %d call %d case %d: n %d (expected %d)\n", tid, i, c, n, expect_iter[c]);
    }
  }
  free(out);
  return NULL;
}

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s lib threads calls\n", argv[0]); return 2; }
  void *h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  init_t init = (init_t)dlsym(h, "LDPCinit");
  dec = (dec_t)dlsym(h, "LDPCdecoder");
  crc_cb = (crc_t)dlsym(h, "nrLDPC_hip_check_crc");
  if (!init || !dec || !crc_cb || init() != 0) { fprintf(stderr, "LDPCinit failed\n"); return 2; }
  progress = getenv("ABI_PROGRESS") != NULL;
  allow_nack = getenv("ABI_ALLOW_NACK") != NULL;
  clock_gettime(CLOCK_MONOTONIC, &t_start);
  const int T = atoi(argv[2]);
  calls_per_thread = atoi(argv[3]);
  if (argc > 4) only_case = atoi(argv[4]) % NCASE;
  unsigned s = 12345;
  if (getenv("ABI_WARM")) { /* build every code's descriptor before the first per-segment call */
    batch_t batch = (batch_t)dlsym(h, "LDPCdecoder_batch");
    for (int c = 0; c < NCASE; c++) {
      nrLDPC_hip_dec_batch_t b;
      int8_t dummy[16];
      int32_t it;
      memset(&b, 0, sizeof(b));
      b.params.BG = cases[c][0]; b.params.Z = cases[c][1]; b.params.R = cases[c][2]; b.params.numMaxIter = 1;
      b.llr = dummy; b.out = dummy; b.n_iter = &it; b.llr_stride = 68 * 384; b.out_stride = 68 * 384;
      batch(&b);
    }
  }
  for (int c = 0; c < NCASE; c++) {
    const int BG = cases[c][0], Z = cases[c][1], R = cases[c][2], n = ncols(BG, R) * Z;
    llr[c] = aligned_alloc(64, (n + 63) / 64 * 64);
    for (int i = 0; i < n; i++) {
      s = s * 1664525u + 1013904223u;
      const int r = (int)((s >> 16) % 41) - 20;            /* -20..20 */
      llr[c][i] = (int8_t)(i < 2 * Z ? 0 : (c & 1) ? 14 + r / 2 : r); /* odd cases: noisy all-zero code word, even: noise */
    }
    memset(&prm[c], 0, sizeof(prm[c]));
    prm[c].BG = BG; prm[c].Z = Z; prm[c].R = R; prm[c].numMaxIter = cases[c][3]; prm[c].outMode = nrLDPC_outMode_BIT;
    if (c == 11) { prm[c].check_crc = crc_cb; prm[c].E = 22 * 384; prm[c].crc_type = 1; } /* CRC-stop mode in the mix */
    out_len[c] = (n + 31) / 32 * 4;
    expect[c] = malloc(out_len[c]);
    memset(expect[c], 0xA5, out_len[c]);
    t_nrLDPC_dec_params p = prm[c];
    expect_iter[c] = dec(&p, 0, 0, 0, llr[c], (int8_t *)expect[c], NULL, NULL);
    if (expect_iter[c] < 0) { fprintf(stderr, "case %d failed single-threaded\n", c); return 1; }
  }
  if (getenv("ABI_FAIL_LAUNCHES_FROM_NOW")) /* fault injection: the resident server cannot be (re)launched any more */
    setenv("NRLDPC_HIP_SRV_TEST_FAIL_LAUNCH", "1", 1);
  pthread_t th[256];
  struct timespec a, b;
  struct rusage ru0, ru1;
  getrusage(RUSAGE_SELF, &ru0);
  clock_gettime(CLOCK_MONOTONIC, &a);
  for (long t = 0; t < T; t++) pthread_create(&th[t], NULL, worker, (void *)t);
  for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &b);
  getrusage(RUSAGE_SELF, &ru1);
  const double dt = (b.tv_sec - a.tv_sec) + (b.tv_nsec - a.tv_nsec) / 1e9;
  /* host CPU burnt by the callers while they wait (user + system, all threads) */
  const double cpu_s = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) / 1e6 +
                       (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) / 1e6;
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* resident submission path: status, slots, kernel launches, calls served, ns sums */
  stats_t stats = (stats_t)dlsym(h, "nrLDPC_hip_server_stats");
  if (stats) stats(st);
  printf("{\"cpu_seconds\": %.4f, \"cpu_seconds_per_1e5_calls\": %.3f, \"nacked\": %d, ", cpu_s, cpu_s * 1e5 / (T * (double)calls_per_thread), nacked);
  printf("\"threads\": %d, \"calls\": %d, \"seconds\": %.4f, \"calls_per_s\": %.0f, \"failures\": %d, \"served\": %lld, "
         "\"server_launches\": %lld, \"slots\": %lld, \"us_per_call_per_thread\": %.2f, \"srv_us\": {\"gpu_stage\": %.2f, \"gpu_decode\": %.2f, "
         "\"host_wait\": %.2f, \"host_call\": %.2f}, \"iters\": [", T, T * calls_per_thread, dt,
         T * calls_per_thread / dt, failures, (long long)st[3], (long long)st[2], (long long)st[1], dt / calls_per_thread * 1e6,
         st[3] ? st[4] / 1e3 / st[3] : 0.0, st[3] ? st[5] / 1e3 / st[3] : 0.0, st[3] ? st[6] / 1e3 / st[3] : 0.0,
         st[3] ? st[7] / 1e3 / st[3] : 0.0);
  for (int c = 0; c < NCASE; c++) printf("%d%s", expect_iter[c], c + 1 < NCASE ? ", " : "]}\n");
  return failures ? 1 : 0;
}
