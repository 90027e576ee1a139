/* A caller's own CRC predicate behind LDPCdecoder(), from plain C through dlopen/dlsym like the reference's loader
 * (nrLDPC_load.c:55-65).  The reference calls p_decParams->check_crc((uint8_t *)p_out, E, crc_type) after every pass >= 3
 * until it returns non-zero (nrLDPC_decoder.c:849-861); so must this library when the pointer is not the CRC it can evaluate
 * on the GPU.  Checked here without any oracle:
 *   - arguments: the caller's p_out pointer itself, E and crc_type as given, also for E % 8 != 0 and outMode BITINT8;
 *   - a predicate that accepts on its k-th call ends the decode with return value k + 2 after exactly k calls, and p_out then
 *     holds the hard decisions of pass k + 2 = what a parity-stop decode capped at numMaxIter = k + 1 delivers for an input
 *     that never converges;
 *   - a predicate that never accepts: numMaxIter - 1 calls, return numMaxIter + 1, p_out = the last pass;
 *   - numMaxIter < 2: never called, p_out untouched.
 * usage: abi_check_crc libldpc_hip.so      exit 0 = all good */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nrLDPC_hip.h"

typedef int32_t (*init_t)(void);
typedef int32_t (*dec_t)(t_nrLDPC_dec_params *, uint8_t, uint8_t, uint8_t, int8_t *, int8_t *, t_nrLDPC_time_stats *, decode_abort_t *);

static int calls, accept_at, bad_args;
static uint8_t *want_ptr;
static uint32_t want_n;
static uint8_t want_type;
static uint8_t seen[16][68 * 384];
static int seen_len;

static int predicate(uint8_t *bytes, uint32_t n, uint8_t type)
{
  if (bytes != want_ptr || n != want_n || type != want_type)
    bad_args++;
  if (calls < 16)
    memcpy(seen[calls], bytes, seen_len);
  calls++;
  return calls == accept_at;
}

static int fails;
#define CHECK(c, ...) do { if (!(c)) { fails++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "usage: %s lib\n", argv[0]); return 2; }
  void *h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  init_t init = (init_t)dlsym(h, "LDPCinit");
  dec_t dec = (dec_t)dlsym(h, "LDPCdecoder");
  if (!init || !dec || init() != 0) { fprintf(stderr, "LDPCinit failed\n"); return 2; }
  static const int codes[4][4] = {{1, 384, 13, 68}, {2, 64, 15, 52}, {1, 22, 23, 35}, {2, 15, 13, 32}}; /* BG, Z, R, ncols */
  unsigned s = 99;
  for (int ci = 0; ci < 4; ci++) {
    const int BG = codes[ci][0], Z = codes[ci][1], R = codes[ci][2], n = codes[ci][3] * Z;
    int8_t *llr = aligned_alloc(64, (n + 63) / 64 * 64);
    for (int i = 0; i < n; i++) { /* pure noise: never converges, never a code word */
      s = s * 1664525u + 1013904223u;
      llr[i] = (int8_t)(i < 2 * Z ? 0 : (int)((s >> 16) % 61) - 30);
    }
    for (int mode = 0; mode < 2; mode++) { /* BIT, BITINT8 */
      const int ob = mode == 0 ? (n + 31) / 32 * 4 : n;
      uint8_t *out = malloc(ob + 64), *ref = malloc(ob + 64);
      seen_len = ob;
      for (int E_odd = 0; E_odd < 2; E_odd++) {
        t_nrLDPC_dec_params p;
        memset(&p, 0, sizeof(p));
        p.BG = BG; p.Z = Z; p.R = R; p.numMaxIter = 8; p.outMode = mode == 0 ? nrLDPC_outMode_BIT : nrLDPC_outMode_BITINT8;
        p.check_crc = predicate; p.E = (BG == 1 ? 22 : 10) * Z - (E_odd ? 3 : 0); p.crc_type = 1 + E_odd;
        want_ptr = out; want_n = (uint32_t)p.E; want_type = (uint8_t)p.crc_type;
        for (int k = 1; k <= 8; k += (k < 3 ? 1 : 2)) { /* accept on the k-th call; 8: never (only 7 calls happen) */
          calls = bad_args = 0;
          accept_at = k;
          memset(out, 0xC3, ob);
          t_nrLDPC_dec_params q = p;
          const int n_it = dec(&q, 0, 0, 0, llr, (int8_t *)out, NULL, NULL);
          const int want_calls = k <= 7 ? k : 7, want_ret = k <= 7 ? k + 2 : 9;
          CHECK(n_it == want_ret && calls == want_calls && bad_args == 0, "code %d mode %d E %d accept %d: ret %d calls %d bad args %d",
                ci, mode, p.E, k, n_it, calls, bad_args);
          /* p_out = the pass that was checked last = a parity-stop decode capped at that pass (the input never converges) */
          t_nrLDPC_dec_params pc = p;
          pc.check_crc = NULL; pc.numMaxIter = (uint8_t)(want_ret - 1);
          memset(ref, 0x3C, ob);
          const int n_pc = dec(&pc, 0, 0, 0, llr, (int8_t *)ref, NULL, NULL);
          CHECK(n_pc == want_ret, "parity-stop reference run converged?! %d", n_pc);
          CHECK(memcmp(out, ref, ob) == 0, "code %d mode %d accept %d: p_out differs from pass %d", ci, mode, k, want_ret);
          CHECK(memcmp(seen[want_calls - 1], ref, ob) == 0, "the predicate saw something else than pass %d on its last call", want_ret);
        }
        /* iteration caps below 2: never checked, p_out untouched, every pass runs */
        for (int cap = 0; cap < 2; cap++) {
          t_nrLDPC_dec_params q = p;
          q.numMaxIter = (uint8_t)cap;
          calls = 0; accept_at = 1;
          memset(out, 0xC3, ob);
          const int n_it = dec(&q, 0, 0, 0, llr, (int8_t *)out, NULL, NULL);
          int touched = 0;
          for (int i = 0; i < ob; i++) touched |= out[i] != 0xC3;
          CHECK(n_it == cap + 1 && calls == 0 && !touched, "cap %d: ret %d calls %d touched %d", cap, n_it, calls, touched);
        }
      }
      free(out); free(ref);
    }
    free(llr);
  }
  printf("abi_check_crc: %s\n", fails ? "FAILED" : "OK");
  return fails != 0;
}
