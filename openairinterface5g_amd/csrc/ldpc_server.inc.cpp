/*
 * ldpc_server.inc.cpp -- host side of the resident submission path (protocol and rationale: ldpc_server.h; kernel:
 * ldpc_server.hip).  Included by ldpc_api.cpp (shares its library state).
 *
 * The reference's callers invoke LDPCdecoder / LDPCencoder from N thread-pool workers at once, one code segment (resp.
 * one group of 8 segments) per call (openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:219,435-468, nr_dlsch_coding.c:171,
 * 386-403).  Each calling thread owns a slot (its mailbox + a workgroup of the server kernel); a call touches no lock
 * shared with other callers and makes no HIP runtime call while the server is up.
 *
 * Environment: NRLDPC_HIP_SERVER=0 falls back to one launch per call (stream + pinned staging per thread);
 * NRLDPC_HIP_SRV_BAR=0 keeps requests in host memory (pulled by the GPU) even where the host could push them into device
 * memory;
 * NRLDPC_HIP_SRV_SLOTS=<n> caller slots = workgroups = CUs the server occupies while it is up (default 64, more
 * threads than slots share them); NRLDPC_HIP_SRV_IDLE_US=<n> the server leaves the GPU after this long without a
 * call (default 20000: ldpctest-style callers spend about a millisecond generating noise between two calls);
 * NRLDPC_HIP_ENC_SERVER=0 / NRLDPC_HIP_ENC_SLOTS=<n>: LDPCencoder's own resident kernel off / its slots (default 16);
 * NRLDPC_HIP_SRV_WAIT=spin|yield|sleep how a caller waits for its completion line: `spin` never leaves the core (lowest
 * latency, one core per caller burnt), `yield` (default) spins briefly and then sched_yield()s between looks, `sleep`
 * sleeps through most of the expected service time (a running average per slot) before it starts looking -- for thread
 * pools wider than the cores they may burn (profiles/r03/abi_wait_modes.txt has calls/s and CPU-seconds for each).
 */
#include <emmintrin.h>
#include <xmmintrin.h>
#include <atomic>
#include <sched.h>
#include <time.h>
#include "ldpc_server.h"

hipError_t ldpc_server_init(void);
hipError_t ldpc_server_launch(const srv_args &a, uint32_t n_slots, hipStream_t stream, int encoder);
int ldpc_server_has_encoder(void);

namespace {

void meter_start(time_stats_t *ts);
void meter_stop(time_stats_t *ts);

struct alignas(64) SrvSlotHost {
  std::atomic<uint32_t> busy{0};
  uint32_t seq = 0;
  uint64_t calls = 0; /* written by the slot's holder only */
  uint64_t ticks_stage = 0, ticks_decode = 0; /* GPU-side: doorbell seen -> payload staged -> block function returned */
  uint64_t ticks_prologue = 0, ticks_passes = 0; /* fast decoder: staged -> state in LDS -> last pass done */
  double host_wait_s = 0, host_total_s = 0;   /* host-side: doorbell rung -> completion seen; whole call */
  double avg_wait_s = 0;                      /* running average of the wait (NRLDPC_HIP_SRV_WAIT=sleep) */
};

struct Server {
  int role = 0;              /* 0: decoder server (LDPCdecoder), 1: encoder server (LDPCencoder): its own kernel, slots and stream */
  std::mutex mu;            /* init / launch / stop only -- never taken by a call that finds the server running */
  std::atomic<int> status{-1}; /* -1 not initialised, 0 usable, 1 disabled or failed */
  int n_slots = 0;
  srv_args args;            /* template of the kernel arguments (gen filled in per launch) */
  srv_slot_ctl *ctl = nullptr;
  srv_req *req = nullptr;   /* request lines and payload areas: device memory written over the BAR, or host memory */
  uint8_t *in_h = nullptr, *out_h = nullptr;
  uint32_t *abort_h = nullptr; /* per slot, 16 words apart: "the caller of request <seq> has given up" (same memory as req) */
  bool over_bar = false;
  uint32_t *state = nullptr, *host_stop = nullptr;
  hipStream_t stream = nullptr;
  std::atomic<uint32_t> gen{0};
  std::atomic<uint32_t> next_slot{0};
  SrvSlotHost *slots = nullptr;
} srv, srv_e;
const int srv_e_role_set = (srv_e.role = 1); /* (namespace-scope initialiser: runs before anything can call into the library) */

void srv_stop_at_exit();

int srv_init_locked(Server &S)
{
  const char *e = getenv("NRLDPC_HIP_SERVER");
  if (e && atoi(e) == 0) {
    S.status = 1;
    return 1;
  }
  if (S.role == 1 && (!ldpc_server_has_encoder() || ((e = getenv("NRLDPC_HIP_ENC_SERVER")) && atoi(e) == 0))) {
    S.status = 1;
    return 1;
  }
  /* the encoder server: LDPCencoder calls carry up to 8 segments each and there are far fewer of them than decoder calls */
  int n = S.role == 1 ? 16 : 64;
  if ((e = getenv(S.role == 1 ? "NRLDPC_HIP_ENC_SLOTS" : "NRLDPC_HIP_SRV_SLOTS")) && atoi(e) >= 1)
    n = atoi(e);
  n = std::min(n, std::min((int)SRV_MAX_SLOTS, std::max(1, g.dev[0].n_cus / 2)));
  int idle_us = 20000;
  if ((e = getenv("NRLDPC_HIP_SRV_IDLE_US")) && atoi(e) >= 1)
    idle_us = std::min(atoi(e), 40000000); /* (idle_ticks is 32 bits of a 100 MHz clock: 42 s) */
  S.status = 1; /* until everything below has worked */
  UseDevice use(g.dev[0]); /* the server lives on the primary device */
  HIP_TRY(ldpc_server_init());
  const unsigned flags = hipHostMallocCoherent | hipHostMallocMapped;
  uint8_t *small = nullptr;
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.ctl), (size_t)n * sizeof(srv_slot_ctl), flags));
  /* request lines + payload areas: in device memory when the host can write there (large BAR), see ldpc_server.h */
  int large_bar = 0;
  if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, g.dev[0].id) != hipSuccess)
    large_bar = 0;
  const char *eb = getenv("NRLDPC_HIP_SRV_BAR");
  S.over_bar = large_bar == 1 && !(eb && atoi(eb) == 0);
  if (S.over_bar) {
    uint8_t *blk = nullptr;
    if (hipExtMallocWithFlags(reinterpret_cast<void **>(&blk), (size_t)n * (sizeof(srv_req) + SRV_IN_STRIDE + 64), hipDeviceMallocFinegrained) == hipSuccess) {
      HIP_TRY(hipMemset(blk, 0, (size_t)n * (sizeof(srv_req) + SRV_IN_STRIDE + 64)));
      HIP_TRY(hipDeviceSynchronize());
      S.req = reinterpret_cast<srv_req *>(blk);
      S.in_h = blk + (size_t)n * sizeof(srv_req);
      S.abort_h = reinterpret_cast<uint32_t *>(blk + (size_t)n * (sizeof(srv_req) + SRV_IN_STRIDE));
    } else {
      (void)hipGetLastError();
      S.over_bar = false;
    }
  }
  if (!S.over_bar) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.req), (size_t)n * sizeof(srv_req), flags));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.in_h), (size_t)n * SRV_IN_STRIDE, flags));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.abort_h), (size_t)n * 64, flags));
    memset(S.req, 0, (size_t)n * sizeof(srv_req));
    memset(S.abort_h, 0, (size_t)n * 64);
  }
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&S.out_h), (size_t)n * SRV_OUT_STRIDE, flags));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&small), 256, flags));
  memset(S.ctl, 0, (size_t)n * sizeof(srv_slot_ctl));
  memset(small, 0, 256);
  S.state = reinterpret_cast<uint32_t *>(small);
  S.host_stop = reinterpret_cast<uint32_t *>(small + 128);
  srv_args &a = S.args;
  memset(&a, 0, sizeof(a));
  void *dp = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dp, S.ctl, 0));
  a.ctl = static_cast<srv_slot_ctl *>(dp);
  if (S.over_bar) {
    a.req = S.req;
    a.in_host = S.in_h;
    a.abort_w = S.abort_h;
  } else {
    HIP_TRY(hipHostGetDevicePointer(&dp, S.abort_h, 0));
    a.abort_w = static_cast<const uint32_t *>(dp);
    HIP_TRY(hipHostGetDevicePointer(&dp, S.req, 0));
    a.req = static_cast<const srv_req *>(dp);
    HIP_TRY(hipHostGetDevicePointer(&dp, S.in_h, 0));
    a.in_host = static_cast<const uint8_t *>(dp);
  }
  HIP_TRY(hipHostGetDevicePointer(&dp, S.out_h, 0));
  a.out_host = static_cast<uint8_t *>(dp);
  HIP_TRY(hipHostGetDevicePointer(&dp, S.state, 0));
  a.state = static_cast<uint32_t *>(dp);
  HIP_TRY(hipHostGetDevicePointer(&dp, S.host_stop, 0));
  a.host_stop = static_cast<const uint32_t *>(dp);
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.staging), (size_t)n * SRV_IN_STRIDE)); /* one row per workgroup */
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&a.gctl), sizeof(srv_gctl)));
  HIP_TRY(hipMemset(a.gctl, 0, sizeof(srv_gctl)));
  a.idle_ticks = (uint32_t)idle_us * 100u; /* wall_clock64: 100 MHz */
  for (int i = 0; i < 4; i++)
    a.crc_pow_tbl[i] = g.dev[0].crc_pow[i];
  /* its own hardware queue: a kernel that stays resident must not sit in front of other streams' launches */
  int lo = 0, hi = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, hi));
  if (getenv("NRLDPC_HIP_SRV_DEBUG"))
    fprintf(stderr, "[libldpc_hip] %s server: %d slots, requests %s, ctl %p in %p out %p state %p | staging %p (%zu B) gctl %p crc_pow %p %p %p %p\n", S.role == 1 ? "encoder" : "decoder", n,
            S.over_bar ? "pushed into device memory over the BAR" : "pulled from host memory",
            (void *)a.ctl, (const void *)a.in_host, (void *)a.out_host, (void *)a.state, (void *)a.staging, (size_t)n * SRV_IN_STRIDE,
            (void *)a.gctl, (const void *)a.crc_pow_tbl[0], (const void *)a.crc_pow_tbl[1], (const void *)a.crc_pow_tbl[2],
            (const void *)a.crc_pow_tbl[3]);
  S.slots = new SrvSlotHost[n];
  S.n_slots = n;
  static bool at_exit_set = false;
  if (!at_exit_set) {
    atexit(srv_stop_at_exit); /* registered after the HIP runtime's own handlers, hence run before them */
    at_exit_set = true;
  }
  S.status = 0;
  return 0;
}

/* 0: usable */
int srv_ready(Server &S)
{
  int st = S.status.load(std::memory_order_acquire);
  if (st >= 0)
    return st;
  if (ensure_ready() != 0)
    return 1;
  /* Both servers are set up together, before either kernel is launched: the set-up allocates, clears and synchronises,
   * and every one of those calls waits for a resident kernel to leave (its idle time-out: 20 ms). */
  static std::mutex init_mu;
  std::lock_guard<std::mutex> lk(init_mu);
  for (Server *T : {&srv, &srv_e}) {
    std::lock_guard<std::mutex> lk2(T->mu);
    if (T->status.load() < 0 && srv_init_locked(*T) != 0 && T->status.load() < 0)
      T->status = 1;
  }
  return S.status.load();
}

/* Called by a waiting caller: (re)launch the server if no generation is running or on its way. */
int srv_ensure_running(Server &S)
{
  uint32_t gcur = S.gen.load(std::memory_order_acquire);
  uint32_t st = __atomic_load_n(S.state, __ATOMIC_ACQUIRE);
  if (gcur && st <= 2 * gcur + 1)
    return 0; /* running (== 2g+1) or launched and not started yet (< 2g+1) */
  std::lock_guard<std::mutex> lk(S.mu);
  gcur = S.gen.load();
  st = __atomic_load_n(S.state, __ATOMIC_ACQUIRE);
  if (gcur && st <= 2 * gcur + 1)
    return 0;
  srv_args a = S.args;
  a.gen = gcur + 1;
  /* fault injection for tests/test_gpu_decoder.py: from the n-th launch of the decoder server on, launching fails the way
   * a HIP error would (NRLDPC_HIP_SRV_TEST_FAIL_LAUNCH=<n>) */
  const char *fe = getenv("NRLDPC_HIP_SRV_TEST_FAIL_LAUNCH"); /* (read per launch: a test arms it after its own set-up) */
  const int fail_from = fe ? atoi(fe) : 0;
  if (fail_from > 0 && S.role == 0 && (int)a.gen >= fail_from)
    return set_error("resident server: launch failed (injected by NRLDPC_HIP_SRV_TEST_FAIL_LAUNCH)");
  UseDevice use(g.dev[0]);
  HIP_TRY(ldpc_server_launch(a, (uint32_t)S.n_slots, S.stream, S.role));
  S.gen.store(gcur + 1, std::memory_order_release);
  return 0;
}

/* Ask the running generation to leave and wait for it (requests already rung are served first or picked up by the next
 * generation).  Used at exit / LDPCshutdown; nothing on the call path needs it. */
void srv_stop(Server &S)
{
  if (S.status.load() != 0)
    return;
  std::lock_guard<std::mutex> lk(S.mu);
  const uint32_t gcur = S.gen.load();
  if (!gcur)
    return;
  __atomic_store_n(S.host_stop, gcur, __ATOMIC_RELEASE);
  UseDevice use(g.dev[0]);
  (void)hipStreamSynchronize(S.stream);
}
void srv_stop_at_exit()
{
  srv_stop(srv);
  srv_stop(srv_e);
}

thread_local int tls_srv_slot[2] = {-1, -1};

enum { SRV_WAIT_SPIN = 0, SRV_WAIT_YIELD = 1, SRV_WAIT_SLEEP = 2 };
int srv_wait_mode()
{
  static const int v = [] {
    const char *e = getenv("NRLDPC_HIP_SRV_WAIT");
    if (e && !strcmp(e, "spin"))
      return (int)SRV_WAIT_SPIN;
    if (e && !strcmp(e, "sleep"))
      return (int)SRV_WAIT_SLEEP;
    return (int)SRV_WAIT_YIELD;
  }();
  return v;
}

/* The server cannot be (re)launched any more (HIP error): every later call takes the launch-per-call path, which reports
 * its own errors; the slot of the call that found out is retired with its request still rung -- a generation that did
 * start would serve it into the slot's areas while another caller owns them. */
void srv_give_up(Server &S, int slot)
{
  S.status.store(1, std::memory_order_release);
  tls_srv_slot[S.role] = -1;
  (void)slot; /* its busy flag stays set: nobody is handed this slot again */
}

inline double srv_now()
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct SrvCall {
  int slot;
  uint8_t *in, *out; /* in: write-only for the host when it is device memory (reads over the BAR crawl) */
  srv_slot_ctl *ctl;
  srv_req *req;
};

SrvCall srv_acquire(Server &S)
{
  if (tls_srv_slot[S.role] < 0)
    tls_srv_slot[S.role] = (int)(S.next_slot.fetch_add(1) % (uint32_t)S.n_slots);
  const int s = tls_srv_slot[S.role];
  SrvSlotHost &h = S.slots[s];
  for (int spins = 0;; spins++) { /* uncontended unless there are more caller threads than slots */
    uint32_t z = 0;
    if (h.busy.compare_exchange_weak(z, 1, std::memory_order_acquire))
      break;
    if (spins > 64)
      sched_yield();
  }
  return SrvCall{s, S.in_h + (size_t)s * SRV_IN_STRIDE, S.out_h + (size_t)s * SRV_OUT_STRIDE, S.ctl + s, S.req + s};
}
void srv_release(Server &S, const SrvCall &c) { S.slots[c.slot].busy.store(0, std::memory_order_release); }

/* publish the request header `rq` (tags still unset) in the slot's ctl line and wait for the completion word; returns
 * n_iter via *n_iter.  The payload must already be in the slot's input area. */
int srv_submit(Server &S, const SrvCall &c, srv_req &rq, int32_t *n_iter, decode_abort_t *ab = nullptr)
{
  SrvSlotHost &h = S.slots[c.slot];
  h.seq = h.seq + 1 >= 0xfffffff0u ? 1u : h.seq + 1;
  h.calls++;
  const uint32_t seq = h.seq;
  const double t_ring = srv_now();
  /* Fields first, then the four chunk tags (ldpc_server.h).  Host memory: x86 keeps the stores in order.  Device memory
   * over the BAR is write-combining: the fence drains the payload and the fields before any tag goes out; the tags
   * themselves may arrive in any order -- the poller waits for all four. */
  volatile srv_req *q = c.req;
  q->seg_in_stride = rq.seg_in_stride; q->seg_out_stride = rq.seg_out_stride; q->payload_bytes = rq.payload_bytes;
  q->code_lo = rq.code_lo; q->code_hi = rq.code_hi; q->kb_nseg = rq.kb_nseg;
  q->kind_mode = rq.kind_mode; q->max_pass = rq.max_pass; q->crcE = rq.crcE;
  __builtin_ia32_sfence();
  __atomic_store_n(&c.req->tag3, seq, __ATOMIC_RELEASE);
  __atomic_store_n(&c.req->tag2, seq, __ATOMIC_RELEASE);
  __atomic_store_n(&c.req->tag1, seq, __ATOMIC_RELEASE);
  __atomic_store_n(&c.req->tag0, seq, __ATOMIC_RELEASE); /* (plain stores only: a locked instruction on BAR memory is a bus lock) */
  __builtin_ia32_sfence(); /* push the line out now */
  bool told = false;
  const int wait_mode = srv_wait_mode();
  if (wait_mode == SRV_WAIT_SLEEP && h.avg_wait_s > 30e-6) {
    /* sleep through ~70 % of what the last calls of this slot took (minus the timer's own slack), then look */
    struct timespec ts = {0, (long)((h.avg_wait_s * 0.7 - 10e-6) * 1e9)};
    if (ts.tv_nsec > 0)
      nanosleep(&ts, nullptr);
  }
  for (uint32_t spins = 0;; spins++) {
    if (__atomic_load_n(&c.ctl->done, __ATOMIC_ACQUIRE) == seq)
      break;
    /* decoder.c:556-559: the decoder looks at the transport block's abort flag every iteration; another worker may raise
     * it while this call is in flight -- passed on to the GPU, which looks at the slot's word once per pass */
    if (ab && !told && __atomic_load_n(reinterpret_cast<const volatile unsigned char *>(&ab->failed), __ATOMIC_RELAXED)) {
      __atomic_store_n(S.abort_h + 16 * c.slot, seq, __ATOMIC_RELEASE);
      __builtin_ia32_sfence();
      told = true;
    }
    if ((spins & 7) == 0 && srv_ensure_running(S) != 0)
      return -2; /* the slot is lost with its request rung: the caller retires it */
    if (spins < 32 || wait_mode == SRV_WAIT_SPIN)
      __builtin_ia32_pause();
    else
      sched_yield(); /* callers outnumber cores on a loaded box: give the others the CPU while the GPU works */
  }
  const double waited = srv_now() - t_ring;
  h.host_wait_s += waited;
  h.avg_wait_s = h.avg_wait_s == 0 ? waited : 0.875 * h.avg_wait_s + 0.125 * waited;
  const int32_t n_rep = __atomic_load_n(&c.ctl->n_iter, __ATOMIC_RELAXED);
  if (n_iter)
    *n_iter = n_rep;
  const uint32_t sd = c.ctl->t_stage_decode, pp = c.ctl->t_pro_passes;
  h.ticks_stage += sd & 0xffffu;
  h.ticks_decode += sd >> 16;
  h.ticks_prologue += pp & 0xffffu;
  h.ticks_passes += pp >> 16;
  return 0;
}

/* 0: decoded through the server, 1: this code cannot be served (caller uses the launch path), -1: error */
/* ob_full: the output bytes of the rate mode the caller asked for when `ce` is that mode cut to its first columns (ldpc_api.cpp
 * LDPCdecoder): what lies behind the cut code's output is zeros -- the hard decisions of degree-1 columns, nrLDPC_bnProc.h has no
 * code for them -- and is written here; 0: `ce` is the mode itself */
int srv_decode(const t_nrLDPC_dec_params *p, const CodeEntry *ce, const int8_t *llr, int8_t *out, int32_t *n_iter, decode_abort_t *ab,
               int ob_full = 0)
{
  const ldpc_code_desc_t &hl = ce->host_lat;
  uint32_t kind;
  if (hl.f_ok && hl.f_lds_total <= SRV_CODE_LDS_MAX)
    kind = SRV_KIND_DEC_FAST;
  else if (hl.lds_total <= SRV_CODE_LDS_MAX)
    kind = SRV_KIND_DEC_GENERIC;
  else
    return 1;
  ldpc_dec_args a;
  if (fill_dec_args(*p, ce, a) != 0)
    return -1;
  const int out_mode = a.out_mode, ob = out_bytes_of(hl, out_mode);
  if ((size_t)hl.num_llr > SRV_IN_STRIDE || (size_t)ob * 4 / 3 + 32 > SRV_OUT_STRIDE)
    return 1;
  const double t_call = srv_now();
  Server &S = srv;
  const SrvCall c = srv_acquire(S);
  srv_req rq;
  memset(&rq, 0, sizeof(rq));
  rq.kind_mode = kind | ((uint32_t)out_mode << 8) | ((uint32_t)(a.use_crc != 0) << 16) | ((a.use_crc ? (uint32_t)p->crc_type : 0u) << 24);
  rq.max_pass = (uint32_t)p->numMaxIter + 1u;
  rq.crcE = (uint32_t)a.E;
  const uint64_t code = reinterpret_cast<uint64_t>(ce->dev_lat);
  rq.code_lo = (uint32_t)code;
  rq.code_hi = (uint32_t)(code >> 32);
  rq.payload_bytes = (uint32_t)hl.num_llr;
  memcpy(c.in, llr, (size_t)hl.num_llr);
  int32_t n = 0;
  int rc = srv_submit(S, c, rq, &n, ab);
  if (rc == -2) {
    srv_give_up(S, c.slot);
    return -1;
  }
  if (rc == 0) {
    *n_iter = n;
    if ((!a.use_crc || n >= 3) && n <= (int32_t)p->numMaxIter + 1) { /* (numMaxIter + 2: given up on the way, nothing was written) */ /* the reference leaves p_out untouched otherwise (decoder.c:849-861) */
      if (kind == SRV_KIND_DEC_FAST) {
        /* 16-byte units {three output dwords, sequence number} (ldpc_server.h): a unit is there when it shows this call's
         * number; one aligned 16-byte load per look, so a unit is never seen half written */
        const uint32_t seq = srv.slots[c.slot].seq;
        const int n_units = (ob / 4 + 2) / 3;
        /* the units' cache lines were invalidated by the GPU's writes: ask for all of them at once instead of taking one
         * miss after the other in the loop below */
        for (int off = 0; off < 16 * n_units; off += 64)
          _mm_prefetch(reinterpret_cast<const char *>(c.out) + off, _MM_HINT_T0);
        /* two steps: every unit is waited for and taken into a local copy first, the caller's p_out is written only when
         * all of them are there -- a call that fails half way leaves p_out as it was */
        static thread_local std::vector<__m128i> got_v; /* (up to 2 177 units for a one-byte-per-bit output row of 68 x 384) */
        if (got_v.size() < (size_t)n_units)
          got_v.resize((size_t)n_units);
        __m128i *got = got_v.data();
        for (int u = 0; u < n_units; u++) {
          const __m128i *src = reinterpret_cast<const __m128i *>(c.out) + u;
          __m128i v = _mm_load_si128(src);
          double t_unit = 0;
          for (uint32_t spins = 0; (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi32(v, 0xff)) != seq; spins++) {
            __builtin_ia32_pause();
            if ((spins & 0xfff) == 0xfff) {
              /* The completion line has arrived, so the unit's store left the GPU before it: microseconds at most.  A unit
               * that stays away means the stores were lost with their generation (device reset, process being torn
               * down): the call FAILS -- never a stale unit copied out under rc 0. */
              const double now = srv_now();
              if (t_unit == 0)
                t_unit = now;
              if (srv_ensure_running(S) != 0 || now - t_unit > 0.5) {
                rc = -1;
                break;
              }
            }
            v = _mm_load_si128(src);
          }
          if (rc != 0)
            break;
          got[u] = v;
        }
        if (rc != 0) {
          /* late stores of this sequence number may still land in the slot's output area: the slot is retired, not handed
           * to the next caller (as srv_give_up does for a call whose completion never came) */
          set_error("resident server: the output of a completed call never arrived");
          srv.slots[c.slot].host_total_s += srv_now() - t_call;
          srv_give_up(S, c.slot);
          return -1;
        }
        for (int u = 0; u < n_units; u++) {
          const int left = ob - 12 * u;
          if (left >= 12) {
            memcpy(out + 12 * u, &got[u], 12);
          } else if (left > 0) {
            memcpy(out + 12 * u, &got[u], (size_t)left);
          }
        }
      } else {
        memcpy(out, c.out, (size_t)ob);
      }
      if (ob_full > ob)
        memset(out + ob, 0, (size_t)(ob_full - ob));
    }
  }
  srv.slots[c.slot].host_total_s += srv_now() - t_call;
  srv_release(S, c);
  return rc;
}

/* segments first .. first+n-1 of an LDPCencoder call; 0 / 1 / -1 as above */
int srv_encode(const CodeEntry *ce, int Kb, uint8_t **input, uint8_t **output, unsigned first, unsigned n, time_stats_t *tinput,
               time_stats_t *tprep, time_stats_t *tparity, time_stats_t *toutput)
{
  const ldpc_code_desc_t &hc = ce->host;
  const int K = hc.kb_full * hc.Z, in_bytes = (K + 7) / 8, N = (hc.ncols - 2) * hc.Z;
  const size_t in_stride = align_up((size_t)in_bytes + 8, 16), out_stride = align_up((size_t)N, 16);
  const size_t lds = (size_t)4 * ((ldpc_encp_lds_words(hc.ncols, hc.kb_full, hc.Z, hc.nrows, hc.nedges) + 3) & ~3) * 8;
  if (!ldpc_server_has_encoder() || n > 8 || in_stride * n > SRV_IN_STRIDE || out_stride * n > SRV_OUT_STRIDE || lds > SRV_CODE_LDS_MAX)
    return 1;
  Server &S = srv_e;
  const SrvCall c = srv_acquire(S);
  srv_req rq;
  memset(&rq, 0, sizeof(rq));
  rq.kind_mode = SRV_KIND_ENC;
  rq.kb_nseg = (uint32_t)Kb | (n << 16);
  const uint64_t code = reinterpret_cast<uint64_t>(ce->dev);
  rq.code_lo = (uint32_t)code;
  rq.code_hi = (uint32_t)(code >> 32);
  rq.seg_in_stride = (uint32_t)in_stride;
  rq.seg_out_stride = (uint32_t)out_stride;
  rq.payload_bytes = (uint32_t)(in_stride * n);
  for (unsigned j = 0; j < n; j++)
    memcpy(c.in + j * in_stride, input[first + j], (size_t)in_bytes);
  meter_stop(tinput); /* started by the caller */
  meter_start(tprep);
  meter_stop(tprep);
  meter_start(tparity);
  const int rc = srv_submit(S, c, rq, nullptr);
  meter_stop(tparity);
  if (rc == -2) {
    srv_give_up(S, c.slot);
    return -1;
  }
  meter_start(toutput);
  if (rc == 0)
    for (unsigned j = 0; j < n; j++)
      memcpy(output[first + j], c.out + j * out_stride, (size_t)N);
  meter_stop(toutput);
  srv_release(S, c);
  return rc;
}

} // namespace
