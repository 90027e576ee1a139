/*
 * ldpc_api.cpp -- C ABI of libldpc_hip.so (see include/nrLDPC_hip.h).
 *
 * Host side of the drop-in: the four plugin symbols of the reference
 * (openair1/PHY/CODING/nrLDPC_extern.h:27-45, resolved by nrLDPC_load.c:45-75) plus the batched entry
 * points.  Everything numeric runs in the HIP kernels; there is no CPU decode/encode path in this
 * library -- without a usable GPU LDPCinit() fails and every other entry point returns an error.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nrLDPC_hip.h"
#include "ldpc_graph.h"
#include "ldpc_kernels.h"
#include "nr_coding_host.h"
#include "tb_chain.h"
#include "ldpc_enc_packed_core.h"

namespace {

thread_local std::string tls_error;
int set_error(const char *what, hipError_t e = hipSuccess)
{
  tls_error = what;
  if (e != hipSuccess) {
    tls_error += ": ";
    tls_error += hipGetErrorString(e);
  }
  return -1;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess)                               \
      return set_error(#expr, e_);                      \
  } while (0)

struct CodeEntry {
  ldpc_code_desc_t host;            /* throughput shape (ldpc_graph.h); everything shape independent is read from here */
  ldpc_code_desc_t *dev = nullptr;
  ldpc_code_desc_t host_lat;        /* latency shape, for launches of at most one workgroup round */
  ldpc_code_desc_t *dev_lat = nullptr;
  ldpc_code_desc_t host_multi;      /* small lifting sizes: several blocks per workgroup -- side by side (Zc % 4 == 0) or four
                                       interleaved byte-wise (any Zc, f_sub = 4); f_ok = 0: not for this code */
  ldpc_code_desc_t *dev_multi = nullptr;
  /* homogeneous launches that fill the GPU with multi-block workgroups at least once */
  bool use_multi(uint32_t n_blocks, int n_cus) const
  {
    return dev_multi && n_blocks >= (uint32_t)(n_cus * host_multi.f_wg_per_cu * host_multi.f_mb * host_multi.f_sub);
  }
  /* the variant for a decoder launch of n_blocks workgroups; force: 0 = by size, 1 = throughput, 2 = latency */
  bool use_latency(uint32_t n_blocks, int n_cus, int force = 0) const
  {
    if (force)
      return force == 2;
    return host.f_ok && n_blocks <= (uint32_t)(n_cus * host_lat.f_wg_per_cu);
  }
};

#define NRLDPC_HIP_MAX_DEVICES 16

/* One per logical device: everything that lives in a GPU's memory.  NRLDPC_HIP_DEVICES=0,1,... lists the GPUs the
 * host-buffer entry points shard their batches over (whole code blocks / whole transport blocks per GPU, no data-path
 * exchange: SURVEY 8e); without it there is one, NRLDPC_HIP_DEVICE (default 0).  Device-buffer calls run on the GPU that
 * owns the buffers.  The same ordinal may be listed twice (two independent contexts on one GPU: used by the tests). */
struct Device {
  int id = 0, n_cus = 256; /* HIP ordinal */
  bool ready = false;
  std::map<uint32_t, CodeEntry *> codes;      /* under Library::mu: the builder's view */
  std::atomic<CodeEntry *> code_tbl[2][385][3]; /* published entries, [BG-1][Z][rate index]: read without a lock */
  uint32_t *crc_pow[4] = {nullptr, nullptr, nullptr, nullptr}; /* CRC24_A, CRC24_B, CRC16, CRC8: x^j mod g, j < 8448 */
  uint32_t *crc_pow_24a_long = nullptr;                        /* CRC24_A up to a whole transport block */
};

struct Library {
  std::mutex mu;
  std::atomic<bool> ready{false};
  int n_dev = 0;     /* logical devices in use */
  int n_shard = 1;   /* the first n_shard of them take part in sharding host-buffer batches */
  Device dev[NRLDPC_HIP_MAX_DEVICES];
  const int *opp_enabled = nullptr;           /* the host executable's meter switch (common/utils/time_meas.h), if it has one */
  /* the host executable's own check_crc (crc_byte.c:314), if it exports one: the pointer the reference's callers put into
   * t_nrLDPC_dec_params (nr_ulsch_decoding.c:216).  That predicate -- and the library's nrLDPC_hip_check_crc -- is the one
   * the GPU evaluates; any other pointer is called on the host (dec_host_predicate) */
  int (*host_check_crc)(uint8_t *, uint32_t, uint8_t) = nullptr;
} g;

/* the device the calling thread is working on (set by the entry points through UseDevice) */
thread_local Device *tls_dev = nullptr;
inline Device &G() { return tls_dev ? *tls_dev : g.dev[0]; }
struct UseDevice { /* selects a logical device for this thread and leaves the caller's HIP device as it found it */
  Device *prev;
  int prev_id = -1;
  bool ok;
  explicit UseDevice(Device &d) : prev(tls_dev)
  {
    (void)hipGetDevice(&prev_id);
    tls_dev = &d;
    ok = hipSetDevice(d.id) == hipSuccess;
  }
  ~UseDevice()
  {
    tls_dev = prev;
    if (prev_id >= 0)
      (void)hipSetDevice(prev_id);
  }
};

/* x^j mod g(x), left aligned in 32 bits, for the polynomials of crc_byte.c:46-58 */
void fill_crc_pow(uint32_t poly, std::vector<uint32_t> &t, size_t len = LDPC_CRC_POW_LEN)
{
  /* A message bit at distance j from the end of an E-bit word contributes x^j (mod g) to word(x) mod g.
   * With the left-aligned register, word(x)*x^deg mod g is what crcbit() (crc_byte.c:65-84) computes; divisibility is
   * the same question, so tabulate r_j = (x^j * x^deg) mod g: r_0 = poly (one 1 bit clocked in), r_{j+1} = r_j * x mod g */
  t.resize(len);
  uint32_t r = poly;
  for (size_t j = 0; j < len; j++) {
    t[j] = r;
    r = (r & 0x80000000u) ? ((r << 1) ^ poly) : (r << 1);
  }
}

int device_init_locked(Device &d, int ordinal)
{
  d.id = ordinal;
  int prev = -1;
  (void)hipGetDevice(&prev);
  HIP_TRY(hipSetDevice(ordinal));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
  d.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  static const uint32_t polys[4] = {0x864cfb00u, 0x80006300u, 0x10210000u, 0x9B000000u};
  for (int i = 0; i < 4; i++) {
    std::vector<uint32_t> t;
    fill_crc_pow(polys[i], t);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.crc_pow[i]), t.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d.crc_pow[i], t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  {
    std::vector<uint32_t> t;
    fill_crc_pow(polys[0], t, TB_CRC24A_POW_LEN);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d.crc_pow_24a_long), t.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d.crc_pow_24a_long, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  d.ready = true;
  if (prev >= 0)
    (void)hipSetDevice(prev);
  return 0;
}

int ensure_ready_locked()
{
  if (g.ready)
    return 0;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_error("no HIP device available (libldpc_hip.so has no CPU fallback)", e);
  int list[NRLDPC_HIP_MAX_DEVICES], n = 0;
  if (const char *env = getenv("NRLDPC_HIP_DEVICES")) {
    for (const char *p = env; *p && n < NRLDPC_HIP_MAX_DEVICES;) {
      char *end = nullptr;
      const long v = strtol(p, &end, 10);
      if (end == p)
        break;
      list[n++] = (int)v;
      p = *end == ',' ? end + 1 : end;
    }
  }
  if (n == 0) {
    const char *env = getenv("NRLDPC_HIP_DEVICE");
    list[n++] = env ? atoi(env) : 0;
  }
  for (int i = 0; i < n; i++)
    if (list[i] < 0 || list[i] >= ndev)
      return set_error("NRLDPC_HIP_DEVICE(S) out of range");
  HIP_TRY(ldpc_kernels_init());
  HIP_TRY(ldpc_fast_kernel_init());
  HIP_TRY(tb_rx_fused_init());
  for (int i = 0; i < n; i++)
    if (device_init_locked(g.dev[i], list[i]) != 0)
      return -1;
  g.n_dev = g.n_shard = n;
  g.opp_enabled = static_cast<const int *>(dlsym(RTLD_DEFAULT, "opp_enabled"));
  g.host_check_crc = reinterpret_cast<int (*)(uint8_t *, uint32_t, uint8_t)>(dlsym(RTLD_DEFAULT, "check_crc"));
  g.ready = true;
  return 0;
}

int ensure_ready()
{
  std::lock_guard<std::mutex> lk(g.mu);
  return ensure_ready_locked();
}

/* the logical device for buffers that live on HIP device `ordinal` (a device outside the configured list is set up on
 * first use); nullptr + error when that fails */
Device *device_for_ordinal(int ordinal)
{
  std::lock_guard<std::mutex> lk(g.mu);
  if (ensure_ready_locked() != 0)
    return nullptr;
  for (int i = 0; i < g.n_dev; i++)
    if (g.dev[i].id == ordinal)
      return &g.dev[i];
  if (g.n_dev >= NRLDPC_HIP_MAX_DEVICES) {
    set_error("too many devices");
    return nullptr;
  }
  if (device_init_locked(g.dev[g.n_dev], ordinal) != 0)
    return nullptr;
  return &g.dev[g.n_dev++];
}
/* the logical device that owns device pointer p (the primary one if the runtime does not know the pointer) */
Device *device_of_pointer(const void *p)
{
  hipPointerAttribute_t at;
  if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice)
    return device_for_ordinal(at.device);
  (void)hipGetLastError();
  return ensure_ready() == 0 ? &g.dev[0] : nullptr;
}

int rate_index(int BG, int R)
{
  if (BG == 1)
    return R == 13 ? 0 : R == 23 ? 1 : R == 89 ? 2 : -1;
  return R == 15 ? 0 : R == 13 ? 1 : R == 23 ? 2 : -1;
}

/* descriptor cache of the current device: built on first use of a (BG, Z, R), uploaded once, never modified afterwards.
 * The per-segment entry points come through here on every call: a published entry is found without taking the library
 * mutex. */
static const CodeEntry *get_code_impl(int BG, int Z, int R, int ri);
const CodeEntry *get_code(int BG, int Z, int R)
{
  const int ri = (BG == 1 || BG == 2) ? rate_index(BG, R) : -1;
  if (ri < 0 || Z < 2 || Z > 384) {
    set_error("invalid (BG, Z, R)");
    return nullptr;
  }
  return get_code_impl(BG, Z, R, ri);
}
/* the code cut to its first `ncols` columns (ldpc_graph.h LDPC_R_COLS; the transport-block chain's plans only) */
const CodeEntry *get_code_cols(int BG, int Z, int ncols)
{
  if ((BG != 1 && BG != 2) || Z < 2 || Z > 384 || ncols <= (BG == 1 ? 26 : 14) || ncols > (BG == 1 ? 68 : 52)) {
    set_error("invalid (BG, Z, columns)");
    return nullptr;
  }
  return get_code_impl(BG, Z, LDPC_R_COLS + ncols, -1);
}
static const CodeEntry *get_code_impl(int BG, int Z, int R, int ri)
{
  if (g.ready && ri >= 0) /* (read without the lock: set once, after everything it guards) */
    if (CodeEntry *hit = G().code_tbl[BG - 1][Z][ri].load(std::memory_order_acquire))
      return hit;
  const uint32_t key = ((uint32_t)BG << 28) | ((uint32_t)Z << 16) | (uint32_t)R;
  std::lock_guard<std::mutex> lk(g.mu);
  if (ensure_ready_locked() != 0)
    return nullptr;
  Device &d = G();
  auto it = d.codes.find(key);
  if (it != d.codes.end())
    return it->second;
  CodeEntry *ce = new CodeEntry();
  if (ldpc_build_code_desc_shape(BG, Z, R, LDPC_SHAPE_THROUGHPUT, &ce->host) != 0 ||
      ldpc_build_code_desc_shape(BG, Z, R, LDPC_SHAPE_LATENCY, &ce->host_lat) != 0) {
    delete ce;
    set_error("invalid (BG, Z, R)");
    return nullptr;
  }
  static const int multi_env = [] { const char *v = getenv("NRLDPC_HIP_MULTI"); return v ? atoi(v) : -1; }(); /* 0: off, n: force n per workgroup */
  ce->host_multi.f_ok = 0;
  if (multi_env != 0) {
    if (ce->host.f_ok) {
      const int mb = multi_env > 0 ? multi_env : ldpc_multi_blocks_for(Z);
      if (mb > 1 && ldpc_build_code_desc_multi(BG, Z, R, mb, &ce->host_multi) != 0)
        ce->host_multi.f_ok = 0;
    } else if (Z <= 30) { /* the lifting sizes that are not multiples of 4 (and 2 .. 7): four blocks interleaved */
      int mb = multi_env > 0 ? multi_env : 64 / Z;
      mb = mb < 1 ? 1 : (mb > 16 ? 16 : mb);
      if (ldpc_build_code_desc_interleaved(BG, Z, R, mb, &ce->host_multi) != 0)
        ce->host_multi.f_ok = 0;
    }
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(d.id);
  if (e == hipSuccess)
    e = hipMalloc(reinterpret_cast<void **>(&ce->dev), 3 * sizeof(ldpc_code_desc_t));
  if (e == hipSuccess)
    e = hipMemcpy(ce->dev, &ce->host, sizeof(ldpc_code_desc_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ce->dev_lat = ce->dev + 1;
    e = hipMemcpy(ce->dev_lat, &ce->host_lat, sizeof(ldpc_code_desc_t), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && ce->host_multi.f_ok) {
    ce->dev_multi = ce->dev + 2;
    e = hipMemcpy(ce->dev_multi, &ce->host_multi, sizeof(ldpc_code_desc_t), hipMemcpyHostToDevice);
  }
  if (prev >= 0)
    (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    set_error("descriptor upload", e);
    delete ce;
    return nullptr;
  }
  if (getenv("NRLDPC_HIP_SRV_DEBUG"))
    fprintf(stderr, "[libldpc_hip] code BG%d Z%d R%d on device %d: descriptors at %p (2 x %zu B)\n", BG, Z, R, d.id, (void *)ce->dev,
            sizeof(ldpc_code_desc_t));
  d.codes[key] = ce;
  if (ri >= 0)
    d.code_tbl[BG - 1][Z][ri].store(ce, std::memory_order_release);
  return ce;
}

int out_bytes_of(const ldpc_code_desc_t &c, int outMode)
{
  return outMode == 0 ? ((c.num_llr + 31) / 32) * 4 : c.num_llr;
}

/* Buffers that are outgrown are parked, not freed: hipFree / hipHostFree wait for every stream of the device, and the
 * resident server kernel (ldpc_server.inc.cpp) may be running on one.  Growth is geometric, so the parked total stays
 * below the live total. */
std::mutex retired_mu;
std::vector<std::pair<void *, bool>> retired; /* (pointer, is host) */
void retire(void *p, bool host)
{
  if (!p)
    return;
  std::lock_guard<std::mutex> lk(retired_mu);
  retired.emplace_back(p, host);
}

/* per-thread staging for the synchronous host-buffer entry points (callers are thread-pool workers:
 * reference nr_ulsch_decoding.c:435-468, nr_dlsch_coding.c:389-403).  A context outlives its thread: it goes back to a
 * pool and the next new thread takes it over, so threads that come and go neither leak nor free. */
struct ThreadCtx {
  hipStream_t stream = nullptr, stream2 = nullptr; /* stream2: second lane of the chunked host-buffer pipeline */
  uint8_t *h_in = nullptr, *h_out = nullptr; /* pinned */
  uint8_t *d_in = nullptr, *d_out = nullptr;
  size_t cap_in = 0, cap_out = 0;
  /* chunks of the host-buffer decode in flight: [first block, blocks) and the event behind the chunk's last copy */
  struct Chunk { uint32_t k0, n; };
  std::vector<Chunk> chunks;
  bool out_direct = false, iter_direct = false; /* the kernels of the call in flight write the caller's arrays themselves */
  std::vector<hipEvent_t> events;
  int event(size_t i, hipEvent_t *e)
  {
    while (events.size() <= i) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      events.push_back(ev);
    }
    *e = events[i];
    return 0;
  }
  int ensure(size_t in_bytes, size_t out_bytes) /* on the current device (UseDevice) */
  {
    if (!stream) {
      HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    }
    if (in_bytes > cap_in) {
      retire(h_in, true); retire(d_in, false);
      h_in = d_in = nullptr; cap_in = 0;
      const size_t cap = in_bytes + in_bytes / 2 + 4096;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_in), cap, hipHostMallocDefault));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_in), cap));
      cap_in = cap;
    }
    if (out_bytes > cap_out) {
      retire(h_out, true); retire(d_out, false);
      h_out = d_out = nullptr; cap_out = 0;
      const size_t cap = out_bytes + out_bytes / 2 + 4096;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_out), cap, hipHostMallocDefault));
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_out), cap));
      cap_out = cap;
    }
    return 0;
  }
  /* an entry point that fails half way leaves nothing in flight on this thread's streams */
  void drain()
  {
    if (stream) {
      (void)hipStreamSynchronize(stream);
      (void)hipStreamSynchronize(stream2);
    }
  }
};
template <typename T> struct CtxPool {
  std::mutex mu;
  std::vector<T *> idle;
  T *take()
  {
    std::lock_guard<std::mutex> lk(mu);
    if (idle.empty())
      return new T();
    T *c = idle.back();
    idle.pop_back();
    return c;
  }
  void give(T *c)
  {
    std::lock_guard<std::mutex> lk(mu);
    idle.push_back(c);
  }
};
template <typename T> struct CtxHolder { /* thread_local, one per logical device: takes a context of that device on first
                                              use, hands it back when the thread ends */
  static CtxPool<T> &pool(int dev)
  {
    static CtxPool<T> *p = new CtxPool<T>[NRLDPC_HIP_MAX_DEVICES]; /* never destroyed: threads may end after the static destructors ran */
    return p[dev];
  }
  T *c = nullptr;
  int dev = 0;
  T &get(int dev_index)
  {
    if (!c) {
      dev = dev_index;
      c = pool(dev).take();
    }
    return *c;
  }
  ~CtxHolder()
  {
    if (c) {
      UseDevice use(g.dev[dev]);
      c->drain();
      pool(dev).give(c);
    }
  }
};
inline int cur_dev_index() { return (int)(&G() - g.dev); }
thread_local CtxHolder<ThreadCtx> tls_ctx_holder[NRLDPC_HIP_MAX_DEVICES];
#define tls_ctx (tls_ctx_holder[cur_dev_index()].get(cur_dev_index()))

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* is [p, p + bytes) page-locked host memory the GPU can address?  Both ends are asked about: a caller may have registered
 * only part of an array, and a range that was registered once and given back to the allocator must not be taken for
 * page-locked because its first byte lies in somebody else's registration now. */
bool host_ptr_is_pinned(const void *p, size_t bytes = 1)
{
  hipPointerAttribute_t at;
  if (!p || hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost) {
    (void)hipGetLastError(); /* plain malloc memory: not an error for us */
    return false;
  }
  if (bytes > 1) {
    hipPointerAttribute_t at2;
    if (hipPointerGetAttributes(&at2, static_cast<const uint8_t *>(p) + bytes - 1) != hipSuccess || at2.type != hipMemoryTypeHost) {
      (void)hipGetLastError();
      return false;
    }
  }
  return true;
}

/* The CPU side of the pageable-array policy (tb_api.inc.cpp TbCtx): large bounce copies are cut over a few helper threads --
 * one memcpy stream moves 8-10 GB/s, the link 50+, and a slot with host soft buffers bounces 84-170 MB per call.  The pool is
 * created on first use (NRLDPC_HIP_BOUNCE_THREADS helpers, default min(8, hardware threads / 2); 0 / 1: the caller copies
 * alone), shared by all calling threads (one job at a time) and torn down at exit. */
struct BouncePool {
  struct Job { uint8_t *dst; const uint8_t *src; size_t width, rows, dpitch, spitch; };
  std::mutex use, mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> workers;
  Job job{};
  size_t next_row = 0, chunk_rows = 1;
  int pending = 0;
  uint64_t generation = 0;
  bool quit = false;
  std::atomic<int> n_threads{-1}; /* -1: pool not set up yet (set once, under `use`) */
  static void copy_rows(const Job &j, size_t r0, size_t r1)
  {
    if (j.width == j.dpitch && j.width == j.spitch) {
      memcpy(j.dst + r0 * j.dpitch, j.src + r0 * j.spitch, (r1 - r0) * j.width);
      return;
    }
    for (size_t r = r0; r < r1; r++)
      memcpy(j.dst + r * j.dpitch, j.src + r * j.spitch, j.width);
  }
  void work()
  {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_go.wait(lk, [&] { return quit || generation != seen; });
      if (quit)
        return;
      seen = generation;
      while (next_row < job.rows) {
        const size_t r0 = next_row, r1 = std::min(job.rows, r0 + chunk_rows);
        next_row = r1;
        lk.unlock();
        copy_rows(job, r0, r1);
        lk.lock();
      }
      if (--pending == 0)
        cv_done.notify_all();
    }
  }
  void run(uint8_t *dst, size_t dpitch, const uint8_t *src, size_t spitch, size_t width, size_t rows)
  {
    Job j{dst, src, width, rows, dpitch, spitch};
    const size_t bytes = width * rows;
    if (n_threads.load(std::memory_order_acquire) < 0) {
      std::lock_guard<std::mutex> lk(use);
      if (n_threads.load(std::memory_order_relaxed) < 0) {
        const char *e = getenv("NRLDPC_HIP_BOUNCE_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        int n = e ? atoi(e) : (int)std::min<unsigned>(8, hw / 2);
        n = n < 2 ? 0 : std::min(n, 32);
        for (int i = 0; i < n; i++)
          workers.emplace_back([this] { work(); });
        n_threads.store(n, std::memory_order_release);
      }
    }
    const int nthr = n_threads.load(std::memory_order_relaxed);
    if (nthr == 0 || bytes < ((size_t)2 << 20)) {
      copy_rows(j, 0, rows);
      return;
    }
    /* a contiguous block is cut into pieces of >= 256 KB, a strided one into runs of rows of about 512 KB */
    size_t tail = 0, per_draw = 1;
    if (rows == 1 || (width == dpitch && width == spitch)) { /* (one row: the pitches mean nothing -- ADVICE r05) */
      const size_t piece = std::max<size_t>((size_t)256 << 10, (bytes / (4 * (size_t)(nthr + 1))) & ~(size_t)63);
      j = Job{dst, src, piece, bytes / piece, piece, piece};
      tail = bytes - j.rows * piece;
    } else {
      per_draw = std::max<size_t>(1, ((size_t)512 << 10) / width);
    }
    {
      std::lock_guard<std::mutex> one(use);
      std::unique_lock<std::mutex> lk(mu);
      job = j;
      next_row = 0;
      chunk_rows = per_draw;
      pending = nthr;
      generation++;
      cv_go.notify_all();
      while (next_row < job.rows) { /* the calling thread works too */
        const size_t r0 = next_row, r1 = std::min(job.rows, r0 + chunk_rows);
        next_row = r1;
        lk.unlock();
        copy_rows(job, r0, r1);
        lk.lock();
      }
      cv_done.wait(lk, [&] { return pending == 0; });
    }
    if (tail)
      memcpy(dst + (bytes - tail), src + (bytes - tail), tail);
  }
  ~BouncePool()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      cv_go.notify_all();
    }
    for (std::thread &t : workers)
      t.join();
  }
};
/* One pool per PROCESS: a child of fork() inherits the pool's bookkeeping but none of its threads -- its first large bounce
 * would wait for helpers that do not exist.  The child therefore forgets the parent's pool (leaked, never
 * destroyed: its mutexes may have been held at the fork) and builds its own on first use. */
std::atomic<BouncePool *> g_bounce_pool{nullptr};
std::atomic<int> g_bounce_lock{0};
void bounce_pool_forget() /* pthread_atfork child handler: async-signal-safe stores only */
{
  g_bounce_pool.store(nullptr, std::memory_order_relaxed);
  g_bounce_lock.store(0, std::memory_order_relaxed);
}
BouncePool &bounce_pool()
{
  BouncePool *p = g_bounce_pool.load(std::memory_order_acquire);
  if (p)
    return *p;
  while (g_bounce_lock.exchange(1, std::memory_order_acquire)) /* (a spin lock: nothing a fork could leave held by a mutex) */
    std::this_thread::yield();
  p = g_bounce_pool.load(std::memory_order_relaxed);
  if (!p) {
    static bool hooked = false;
    if (!hooked) {
      (void)pthread_atfork(nullptr, nullptr, bounce_pool_forget);
      hooked = true;
    }
    p = new BouncePool;
    g_bounce_pool.store(p, std::memory_order_release);
  }
  g_bounce_lock.store(0, std::memory_order_release);
  return *p;
}
struct BouncePoolAtExit { /* (the helpers are joined when the library is unloaded or the process exits normally) */
  ~BouncePoolAtExit()
  {
    if (BouncePool *p = g_bounce_pool.exchange(nullptr))
      delete p;
  }
} g_bounce_pool_at_exit;
/* dst / src: `rows` rows of `width` bytes, dpitch / spitch apart (rows == 1: one block) */
void bounce_copy(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows)
{
  if (width && rows)
    bounce_pool().run(static_cast<uint8_t *>(dst), dpitch, static_cast<const uint8_t *>(src), spitch, width, rows);
}
/* NRLDPC_HIP_PAGEABLE_DIRECT=1 -- CONTROL ARM of the soak in profiles/r05 only: pageable arrays are handed to hipMemcpy*Async
 * again, as before 6dd81de (the arrangement that took a GPU memory fault in one test-suite run in ten) */
bool pageable_direct()
{
  static const bool v = [] { const char *e = getenv("NRLDPC_HIP_PAGEABLE_DIRECT"); return e && atoi(e) != 0; }();
  return v;
}
/* does the transport-block chain have to bounce this host array through page-locked memory? */
bool needs_bounce(const void *p, size_t bytes);

/* [p, p + bytes) is page-locked at one end only: the HIP runtime refuses copies out of such a range (it takes the array for
 * the registered piece), so the entry points that take host arrays say so instead of failing somewhere inside */
bool host_range_is_partly_pinned(const void *p, size_t bytes)
{
  if (!p || bytes < 2)
    return false;
  return host_ptr_is_pinned(p) != host_ptr_is_pinned(static_cast<const uint8_t *>(p) + bytes - 1);
}

bool needs_bounce(const void *p, size_t bytes) { return !pageable_direct() && !host_ptr_is_pinned(p, bytes); }

/* kernel choice: 0 = best available, 1 = generic, 2 = fast (error when the code / buffers do not allow it), with the
 * workgroup shape picked from the launch size; 3 / 4 = fast kernel, throughput / latency shape forced (tests, tuning) */
int launch_decoder(int kernel, ldpc_dec_args a, const CodeEntry *ce, uint32_t n_blocks, hipStream_t s, uint32_t batch_blocks = 0)
{
  if (batch_blocks < n_blocks)
    batch_blocks = n_blocks; /* the launch is one chunk of a larger batch: the whole batch decides the shape */
  const ldpc_code_desc_t &hc = ce->host;
  const bool fast_ok = hc.f_ok && ((reinterpret_cast<uintptr_t>(a.llr) | a.llr_stride) & 3) == 0;
  /* kernel 5: several blocks per workgroup forced (tests); 0 / 2: when the launch fills the GPU with such workgroups */
  const bool multi_ok = !a.jobs && ce->dev_multi && (ce->host_multi.f_sub == 4 || fast_ok);
  if (multi_ok && (kernel == 5 || ((kernel == 0 || kernel == 2) && ce->use_multi(batch_blocks, G().n_cus)))) {
    a.code = ce->dev_multi;
    HIP_TRY(ldpc_launch_dec_fast_multi(a, ce->host_multi, n_blocks, s));
    return 0;
  }
  if (kernel == 5)
    return set_error("no multi-block variant for this code");
  if (kernel >= 2 && !fast_ok)
    return set_error("fast kernel not applicable (needs Zc % 4 == 0, Zc >= 8, 4-byte aligned LLR rows)");
  if (kernel != 1 && fast_ok) {
    const bool lat = ce->use_latency(batch_blocks, G().n_cus, kernel == 3 ? 1 : (kernel == 4 ? 2 : 0));
    a.code = lat ? ce->dev_lat : ce->dev;
    /* diagnostics: per-workgroup clocks of this launch (the file holds the process' last traced launch) */
    static const char *trace_file = getenv("NRLDPC_HIP_DEC_TRACE");
    void *trace_d = nullptr;
    if (trace_file && !a.jobs && !a.use_crc) {
      HIP_TRY(hipMalloc(&trace_d, (size_t)n_blocks * 256));
      HIP_TRY(hipMemsetAsync(trace_d, 0, (size_t)n_blocks * 256, s));
      a.trace = static_cast<unsigned long long *>(trace_d);
    }
    const ldpc_code_desc_t &hsel = lat ? ce->host_lat : ce->host;
    /* workgroups that share a CU take turns at the issue priority (ldpc_dec_fast_block.h; NRLDPC_HIP_FAIR=0: off) */
    static const int fair_env = [] { const char *e = getenv("NRLDPC_HIP_FAIR"); return e ? atoi(e) : 1; }();
    if (fair_env && hsel.f_wg_per_cu >= 2 && n_blocks > (uint32_t)G().n_cus && hsel.f_n_threads >= 256)
      a.fair = hsel.f_n_threads / 256; /* waves per SIMD of one workgroup */
    HIP_TRY(ldpc_launch_dec_fast(a, hsel, n_blocks, s));
    if (trace_d) {
      std::vector<unsigned long long> h((size_t)n_blocks * 32);
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(hipMemcpy(h.data(), trace_d, h.size() * 8, hipMemcpyDeviceToHost));
      HIP_TRY(hipFree(trace_d));
      if (FILE *f = fopen(trace_file, "wb")) {
        fwrite(h.data(), 8, h.size(), f);
        fclose(f);
      }
    }
  } else {
    a.code = ce->dev;
    HIP_TRY(ldpc_launch_dec_generic(a, hc, n_blocks, s));
  }
  return 0;
}

/* CRC stop: can the GPU evaluate this call's predicate (see t_nrLDPC_dec_params::check_crc in nrLDPC_hip.h)?  false = the
 * predicate has to be called on the host, on p_out, as nrLDPC_decoder.c:857 calls it. */
bool crc_on_device(const t_nrLDPC_dec_params &p, const CodeEntry *ce)
{
  /* NRLDPC_HIP_CRC_TRUST_POINTER=1: ANY non-NULL predicate counts as OAI's check_crc (round 4's behaviour) -- for hosts whose
   * own check_crc cannot be found by name: executables linked without -rdynamic (the reference links with it,
   * CMakeLists.txt:165), static or hidden-visibility builds, a check_crc that lives in an RTLD_LOCAL library */
  static const bool trust = [] { const char *e = getenv("NRLDPC_HIP_CRC_TRUST_POINTER"); return e && atoi(e) != 0; }();
  const bool known = trust || p.check_crc == &nrLDPC_hip_check_crc || (g.host_check_crc && p.check_crc == g.host_check_crc);
  if (!known && p.check_crc) { /* say once what the slow path is and why it was taken */
    static std::atomic<bool> told{false};
    if (!told.exchange(true))
      fprintf(stderr, "[libldpc_hip] check_crc = %p is neither the library's nrLDPC_hip_check_crc nor the host's check_crc (%s): "
                      "the predicate is called on the host after every pass from the third on -- every pass runs on the GPU, one output "
                      "row per pass crosses the link, no early stop on the device.  If this pointer IS OAI's check_crc "
                      "(crc_byte.c:314), set NRLDPC_HIP_CRC_TRUST_POINTER=1 or link the executable with -rdynamic.\n",
              reinterpret_cast<void *>(p.check_crc), g.host_check_crc ? "found by dlsym, another address" : "dlsym(RTLD_DEFAULT, \"check_crc\") found none");
  }
  return known && p.outMode == nrLDPC_outMode_BIT && p.E > 0 && (p.E & 7) == 0 && p.E <= ce->host.kb_full * ce->host.Z &&
         p.E <= LDPC_CRC_POW_LEN && p.crc_type >= 0 && p.crc_type <= 3;
}

int fill_dec_args(const t_nrLDPC_dec_params &p, const CodeEntry *ce, ldpc_dec_args &a)
{
  a.code = ce->dev;
  a.num_max_iter = p.numMaxIter;
  a.out_mode = p.outMode == nrLDPC_outMode_BIT ? 0 : 1;
  a.use_crc = p.check_crc != nullptr;
  a.E = 0;
  a.crc_pow = nullptr;
  a.jobs = nullptr;
  a.tb_abort = nullptr;
  a.pull = nullptr;
  a.pull_stride = 0;
  a.n_blocks = 0;
  a.pull_stagger_ticks = a.pull_first_round = 0;
  a.trace = nullptr;
  a.fair = 0;
  for (int i = 0; i < 4; i++)
    a.crc_pow_tbl[i] = G().crc_pow[i];
  if (a.use_crc) {
    if (p.crc_type < 0 || p.crc_type > 3)
      return set_error("invalid crc_type");
    if (p.E <= 0 || (p.E & 7) || p.E > ce->host.kb_full * ce->host.Z || p.E > LDPC_CRC_POW_LEN)
      return set_error("CRC mode needs E > 0, E % 8 == 0, E <= K");
    if (p.outMode != nrLDPC_outMode_BIT)
      return set_error("CRC mode needs outMode BIT (the reference checks the packed bytes)");
    a.E = p.E;
    a.crc_pow = G().crc_pow[p.crc_type];
  }
  return 0;
}

} // namespace

#include "ldpc_server.inc.cpp"

namespace {
/* start_meas / stop_meas of the reference (common/utils/time_meas.h:148-176), which are inline functions of the host
 * executable gated by its global `opp_enabled`; a host without that symbol (tests) counts as enabled */
inline bool meters_on() { return !g.opp_enabled || *g.opp_enabled; }
inline long long meter_clock() { return (long long)__builtin_ia32_rdtsc(); }
void meter_start(time_stats_t *ts)
{
  if (!ts || !meters_on())
    return;
  if (ts->meas_flag == 0) {
    ts->trials++;
    ts->meas_flag = 1;
  }
  ts->in = meter_clock();
  if ((ts->trials & 16383) < 10)
    ts->max = 0;
}
void meter_stop(time_stats_t *ts)
{
  if (!ts || !meters_on() || !ts->in)
    return;
  const long long d = meter_clock() - ts->in;
  ts->diff += d;
  ts->p_time = d;
  ts->diff_square += (double)d * (double)d;
  if (d > ts->max)
    ts->max = d;
  ts->meas_flag = 0;
}
} // namespace

namespace {
/* LDPCinit's body, also behind the offload slot's init (tb_offload.inc.cpp): internal, so that neither entry point depends
 * on which LDPCinit the dynamic linker binds */
int32_t lib_init()
{
  if (ensure_ready() != 0)
    return -1;
  /* the resident servers' mailboxes and buffers (tens of milliseconds of allocations) are set up here, not inside the
   * first LDPCdecoder / LDPCencoder call; the kernels themselves start with the first call and leave when idle */
  (void)srv_ready(srv);
  return 0;
}
} // namespace

extern "C" {

int nrLDPC_hip_check_crc(uint8_t *decoded_bytes, uint32_t n, uint8_t crc_type) { return nr_hip_check_crc(decoded_bytes, n, crc_type); }
const char *nrLDPC_hip_last_error(void) { return tls_error.c_str(); }
const char *nrLDPC_hip_version(void) { return "libldpc_hip 0.5 (gfx950)"; }

/* Optional hook of the reference's module loader (common/utils/load_module_shlib.c:174-185: "<modname>_checkbuildver",
 * modname = "ldpc" whatever the version suffix of the file name, nrLDPC_load.c:48,62): called right after
 * dlopen() with the executable's build string; a negative return makes the loader refuse the library.  The reference's
 * in-tree modules demand identical build strings (3gpplte_sse.c:361-372); an out-of-tree library cannot, so the operator
 * pins what this build was validated against: with NRLDPC_HIP_REQUIRE_BUILD=<text> the library refuses any executable
 * whose build string does not contain <text> (structure layouts are checked against the reference headers in
 * tests/test_abi.py -- an executable of another revision may have moved them). */
int32_t ldpc_checkbuildver(char *mainexec_buildversion, char **shlib_buildversion)
{
  return nrLDPC_hip_checkbuildver(mainexec_buildversion, shlib_buildversion);
}
/* load_module_shlib.c:186-191: the loader's second optional hook (include/nrLDPC_hip.h) */
int32_t ldpc_autoinit(void *arg)
{
  if (arg) {
    const char *list = static_cast<const char *>(arg);
    for (const char *c = list; *c; c++)
      if (!((*c >= '0' && *c <= '9') || *c == ',') || c - list > 64) {
        fprintf(stderr, "[libldpc_hip] ldpc_autoinit: argument is not a GPU list (\"0\", \"2,3\"), ignored\n");
        return -1;
      }
    if (*list)
      setenv("NRLDPC_HIP_DEVICES", list, 0);
  }
  return 0;
}
/* the same under a name that does not clash with the loader's hook: libldpc_hip_t2.so forwards its own ldpc_checkbuildver here */
int32_t nrLDPC_hip_checkbuildver(char *mainexec_buildversion, char **shlib_buildversion)
{
  static char version[] = "libldpc_hip 0.5 (gfx950; plugin ABI of openairinterface5g v2.1.0: nrLDPC_defs.h:40-87, nrLDPC_types.h:75-127)";
  if (shlib_buildversion)
    *shlib_buildversion = version;
  const char *need = getenv("NRLDPC_HIP_REQUIRE_BUILD");
  if (need && *need && (!mainexec_buildversion || !strstr(mainexec_buildversion, need))) {
    fprintf(stderr, "[libldpc_hip] refusing to load into \"%s\": NRLDPC_HIP_REQUIRE_BUILD=\"%s\"\n",
            mainexec_buildversion ? mainexec_buildversion : "(null)", need);
    return -1;
  }
  return 0;
}

int32_t nrLDPC_hip_num_llr(int BG, int Z, int R)
{
  ldpc_code_desc_t d;
  return ldpc_build_code_desc(BG, Z, R, &d) == 0 ? d.num_llr : -1;
}
int32_t nrLDPC_hip_out_bytes(int BG, int Z, int R, int outMode)
{
  ldpc_code_desc_t d;
  return ldpc_build_code_desc(BG, Z, R, &d) == 0 ? out_bytes_of(d, outMode) : -1;
}
int32_t nrLDPC_hip_lds_bytes(int BG, int Z, int R)
{
  ldpc_code_desc_t d;
  return ldpc_build_code_desc(BG, Z, R, &d) == 0 ? d.lds_total : -1;
}

int32_t nrLDPC_hip_code_info(int BG, int Z, int R, int32_t info[8])
{
  static thread_local ldpc_code_desc_t d;
  if (!info || ldpc_build_code_desc(BG, Z, R, &d) != 0)
    return -1;
  info[0] = d.nrows; info[1] = d.ncols; info[2] = d.nedges; info[3] = d.f_ok;
  info[4] = d.f_ok ? d.f_n_threads : d.n_threads;
  info[5] = d.f_ok ? d.f_lds_total : d.lds_total;
  info[6] = d.f_ok ? d.f_n_cn_tasks : 0;
  info[7] = d.f_ok ? d.f_n_bn_tasks : 0;
  return 0;
}

int32_t nrLDPC_hip_server_stats(int64_t out[8])
{
  if (!out)
    return -1;
  out[4] = out[5] = out[6] = out[7] = 0;
  if (getenv("NRLDPC_HIP_SRV_DEBUG")) {
    uint64_t pro = 0, pas = 0, n = 0;
    for (int i = 0; i < srv.n_slots; i++) {
      pro += srv.slots[i].ticks_prologue; pas += srv.slots[i].ticks_passes; n += srv.slots[i].calls;
    }
    if (n)
      fprintf(stderr, "[libldpc_hip] server, fast decoder per call (only meaningful when every call used it): prologue %.2f us, passes %.2f us\n",
              pro / 100.0 / n, pas / 100.0 / n);
  }
  for (int i = 0; i < srv.n_slots; i++) {
    out[4] += (int64_t)srv.slots[i].ticks_stage * 10;  /* ns: doorbell seen -> payload staged (GPU clock) */
    out[5] += (int64_t)srv.slots[i].ticks_decode * 10; /* ns: payload staged -> block decoded */
    out[6] += (int64_t)(srv.slots[i].host_wait_s * 1e9);  /* ns: doorbell rung -> completion seen (host clock) */
    out[7] += (int64_t)(srv.slots[i].host_total_s * 1e9); /* ns: whole srv_decode call */
  }
  out[0] = srv.status.load();
  out[1] = srv.n_slots;
  out[2] = srv.gen.load();
  int64_t calls = 0;
  for (int i = 0; i < srv.n_slots; i++)
    calls += (int64_t)srv.slots[i].calls; /* racy snapshot, exact when no call is in flight */
  out[3] = calls;
  return 0;
}

int32_t LDPCinit(void) { return lib_init(); }

int32_t LDPCshutdown(void)
{
  /* The reference's LDPCshutdown is a no-op (nrLDPC_decoder.c:167).  Device objects are kept: the loader
   * maps the library RTLD_NODELETE (common/utils/load_module_shlib.c:160) and other threads may still be
   * inside a call.  The resident server kernel is asked to leave (a later call simply starts it again). */
  srv_stop(srv);
  srv_stop(srv_e);
  return 0;
}

} /* extern "C" */

namespace {

/* contiguous, balanced [lo, hi) of n items for part k of parts */
inline void shard_range(uint32_t n, int k, int parts, uint32_t &lo, uint32_t &hi)
{
  const uint32_t base = n / (uint32_t)parts, rem = n % (uint32_t)parts;
  lo = (uint32_t)k * base + std::min<uint32_t>((uint32_t)k, rem);
  hi = lo + base + ((uint32_t)k < rem ? 1u : 0u);
}

/* Host-buffer decode of blocks [i0, i0+n) of b on the CURRENT device: everything is enqueued on this thread's two
 * streams of that device; dec_host_finish() waits and hands the results over.  Large batches go in chunks alternating
 * between the two streams, so that the CPU copy of chunk k+1 into pinned memory (pageable source only), the PCIe
 * transfer of chunk k and the kernel of chunk k-1 overlap; every chunk has its own staging region (no intermediate
 * waits).  LLRs that already live in page-locked memory (hipHostMalloc / hipHostRegister) are fetched by the copy
 * engine in place, rows and padding alike, in one linear copy per chunk. */
int dec_host_enqueue(const nrLDPC_hip_dec_batch_t *b, uint32_t i0, uint32_t n_part, bool pinned)
{
  const t_nrLDPC_dec_params &p = b->params;
  const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  ldpc_dec_args a;
  if (fill_dec_args(p, ce, a) != 0)
    return -1;
  const int ob = out_bytes_of(hc, a.out_mode);
  ThreadCtx &c = tls_ctx;
  /* page-locked source with a sane stride: the device copy keeps the caller's row pitch */
  const bool direct = pinned && n_part >= 16 && b->llr_stride <= 2u * (uint32_t)hc.num_llr;
  const size_t in_stride = direct ? b->llr_stride : align_up(hc.num_llr, 16), out_stride = align_up(ob, 16);
  /* the pass counts travel behind the output rows in the same buffers: one device->host copy for a one-chunk call */
  const size_t iter_off = out_stride * n_part;
  if (c.ensure(in_stride * n_part, iter_off + sizeof(int32_t) * n_part) != 0)
    return -1;
  int32_t *d_iter = reinterpret_cast<int32_t *>(c.d_out + iter_off);
  /* page-locked source: a chunk = one full round of decoder workgroups on this GPU (a chunk that leaves a quarter round
   * over pays a whole round for it), at least ~4 MiB so that the link runs near its streaming rate */
  const size_t chunk_bytes = direct ? ((size_t)4 << 20) : ((size_t)3 << 20);
  uint32_t chunk = (uint32_t)std::max<size_t>(1, chunk_bytes / in_stride);
  if (hc.f_ok) { /* pageable source: half rounds (the CPU staging copy is the slower stage; measured in profiles/r02/host_path_sweep.txt) */
    const uint32_t round = (uint32_t)(G().n_cus * std::max(1, hc.f_wg_per_cu)) / (direct ? 1u : 2u);
    chunk = (chunk + round - 1) / round * round;
  }
  static const int chunk_env = [] { const char *e = getenv("NRLDPC_HIP_HOST_CHUNK"); return e ? atoi(e) : 0; }(); /* tuning knob: blocks per chunk */
  if (chunk_env > 0)
    chunk = (uint32_t)chunk_env;
  int lane = 0;
  c.chunks.clear();
  /* Fast-kernel codes: the decoder's workgroups pull their rows over the link themselves (ldpc_dec_fast_pull_kernel) --
   * from the caller's page-locked buffer in place, or from this thread's page-locked staging area, which the calling
   * thread fills chunk by chunk while the previous chunks are being pulled and decoded -- and write bits and pass counts
   * straight into page-locked memory.  No copy engine on the path.  NRLDPC_HIP_HOST_PULL=0: copy-engine path. */
  static const int pull_env = [] { const char *e = getenv("NRLDPC_HIP_HOST_PULL"); return e ? atoi(e) : 1; }();
  bool pull = pull_env != 0 && hc.f_ok && b->kernel != 1 && n_part >= 16;
  const int8_t *pull_src = nullptr;
  uint8_t *stage_out = nullptr;
  c.out_direct = c.iter_direct = false;
  int8_t *out_dev = nullptr;
  int32_t *iter_dev = nullptr;
  if (pull) {
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, c.h_out, 0) != hipSuccess) {
      (void)hipGetLastError();
      pull = false;
    }
    stage_out = static_cast<uint8_t *>(dp);
    const void *hsrc = direct ? static_cast<const void *>(b->llr + (size_t)i0 * b->llr_stride) : static_cast<const void *>(c.h_in);
    if (pull && (((reinterpret_cast<uintptr_t>(hsrc) | in_stride) & 3) != 0 ||
                 hipHostGetDevicePointer(&dp, const_cast<void *>(hsrc), 0) != hipSuccess)) {
      (void)hipGetLastError();
      pull = false;
    }
    pull_src = static_cast<const int8_t *>(dp);
  }
  if (pull) {
    /* results: straight into the caller's arrays when those are page-locked too (nothing left to hand over but the
     * event), otherwise into this thread's page-locked staging rows, copied out chunk by chunk in dec_host_finish */
    void *dp = nullptr;
    int8_t *out0 = b->out + (size_t)i0 * b->out_stride;
    if (((reinterpret_cast<uintptr_t>(out0) | b->out_stride) & 3) == 0 && host_ptr_is_pinned(out0, (size_t)(n_part - 1) * b->out_stride + 1) &&
        hipHostGetDevicePointer(&dp, out0, 0) == hipSuccess) {
      c.out_direct = true;
      out_dev = static_cast<int8_t *>(dp);
    }
    if (host_ptr_is_pinned(b->n_iter + i0, (size_t)n_part * sizeof(int32_t)) && hipHostGetDevicePointer(&dp, b->n_iter + i0, 0) == hipSuccess) {
      c.iter_direct = true;
      iter_dev = static_cast<int32_t *>(dp);
    }
    (void)hipGetLastError();
    if (chunk_env <= 0) {
      /* page-locked source: two workgroup rounds per launch when results are handed over by this thread (the copy-out of
       * one chunk overlaps the next chunk's kernel), one launch otherwise; pageable: one round per launch, the CPU copy sets the pace */
      const uint32_t round = (uint32_t)(G().n_cus * std::max(1, hc.f_wg_per_cu));
      chunk = direct ? (c.out_direct ? n_part : 2 * round) : round;
    }
    const size_t stage_stride = align_up(hc.num_llr, 16);
    for (uint32_t k0 = 0; k0 < n_part; k0 += chunk, lane ^= 1) {
      const uint32_t n = std::min(chunk, n_part - k0);
      hipStream_t s = lane ? c.stream2 : c.stream;
      if (!direct) {
        const int8_t *src = b->llr + (size_t)(i0 + k0) * b->llr_stride;
        for (uint32_t i = 0; i < n; i++)
          memcpy(c.h_in + (k0 + i) * in_stride, src + (size_t)i * b->llr_stride, hc.num_llr);
      }
      a.pull = pull_src + (size_t)k0 * in_stride; a.pull_stride = (uint32_t)in_stride;
      a.llr = reinterpret_cast<const int8_t *>(c.d_in + k0 * stage_stride); a.llr_stride = (uint32_t)stage_stride;
      if (c.out_direct) {
        a.out = out_dev + (size_t)k0 * b->out_stride; a.out_stride = b->out_stride;
      } else {
        a.out = reinterpret_cast<int8_t *>(stage_out + k0 * out_stride); a.out_stride = (uint32_t)out_stride;
      }
      a.n_iter = c.iter_direct ? iter_dev + k0 : reinterpret_cast<int32_t *>(stage_out + iter_off) + k0;
      const bool lat = ce->use_latency(n_part, G().n_cus, b->kernel == 3 ? 1 : (b->kernel == 4 ? 2 : 0));
      a.code = lat ? ce->dev_lat : ce->dev;
      /* first-round stagger of the pull kernel (ticks of 10 ns between groups of 64 workgroups; only the first launch of a
       * call, and only when it fills the GPU): NRLDPC_HIP_PULL_STAGGER_US, default 25 */
      static const int stagger_us = [] { const char *e = getenv("NRLDPC_HIP_PULL_STAGGER_US"); return e ? atoi(e) : 25; }();
      const ldpc_code_desc_t &hsel = lat ? ce->host_lat : ce->host;
      a.pull_first_round = (uint32_t)(G().n_cus * std::max(1, hsel.f_wg_per_cu));
      a.pull_stagger_ticks = (k0 == 0 && n >= a.pull_first_round) ? (uint32_t)stagger_us * 100u : 0u;
      HIP_TRY(ldpc_launch_dec_fast_pull(a, hsel, n, s));
      hipEvent_t ev;
      if (c.event(c.chunks.size(), &ev) != 0)
        return -1;
      HIP_TRY(hipEventRecord(ev, s));
      c.chunks.push_back(ThreadCtx::Chunk{k0, n});
    }
    return 0;
  }
  for (uint32_t k0 = 0; k0 < n_part; k0 += chunk, lane ^= 1) {
    const uint32_t n = std::min(chunk, n_part - k0);
    hipStream_t s = lane ? c.stream2 : c.stream;
    const int8_t *src = b->llr + (size_t)(i0 + k0) * b->llr_stride;
    if (direct) {
      const size_t bytes = (size_t)(n - 1) * in_stride + (size_t)hc.num_llr; /* not past the last row's LLRs */
      HIP_TRY(hipMemcpyAsync(c.d_in + k0 * in_stride, src, bytes, hipMemcpyHostToDevice, s));
    } else {
      for (uint32_t i = 0; i < n; i++)
        memcpy(c.h_in + (k0 + i) * in_stride, src + (size_t)i * b->llr_stride, hc.num_llr);
      HIP_TRY(hipMemcpyAsync(c.d_in + k0 * in_stride, c.h_in + k0 * in_stride, in_stride * n, hipMemcpyHostToDevice, s));
    }
    a.llr = reinterpret_cast<const int8_t *>(c.d_in + k0 * in_stride); a.llr_stride = (uint32_t)in_stride;
    a.out = reinterpret_cast<int8_t *>(c.d_out + k0 * out_stride); a.out_stride = (uint32_t)out_stride;
    a.n_iter = d_iter + k0;
    if (launch_decoder(b->kernel, a, ce, n, s, n_part) != 0)
      return -1;
    if (n == n_part) {
      HIP_TRY(hipMemcpyAsync(c.h_out, c.d_out, iter_off + sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    } else {
      HIP_TRY(hipMemcpyAsync(c.h_out + k0 * out_stride, c.d_out + k0 * out_stride, out_stride * n, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(c.h_out + iter_off + sizeof(int32_t) * k0, c.d_out + iter_off + sizeof(int32_t) * k0,
                             sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    }
    hipEvent_t ev;
    if (c.event(c.chunks.size(), &ev) != 0)
      return -1;
    HIP_TRY(hipEventRecord(ev, s));
    c.chunks.push_back(ThreadCtx::Chunk{k0, n});
  }
  return 0;
}

int dec_host_finish(const nrLDPC_hip_dec_batch_t *b, uint32_t i0, uint32_t n_part)
{
  const t_nrLDPC_dec_params &p = b->params;
  const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
  if (!ce)
    return -1;
  const int ob = out_bytes_of(ce->host, p.outMode == nrLDPC_outMode_BIT ? 0 : 1);
  const size_t out_stride = align_up(ob, 16), iter_off = out_stride * n_part;
  ThreadCtx &c = tls_ctx;
  const int32_t *h_iter = reinterpret_cast<const int32_t *>(c.h_out + iter_off);
  const bool use_crc = p.check_crc != nullptr;
  /* chunk by chunk: a chunk's results are handed over while the later chunks are still on the link / in the decoder */
  for (size_t q = 0; q < c.chunks.size(); q++) {
    HIP_TRY(hipEventSynchronize(c.events[q]));
    if (c.out_direct && c.iter_direct)
      continue;
    for (uint32_t i = c.chunks[q].k0; i < c.chunks[q].k0 + c.chunks[q].n; i++) {
      const int32_t n = c.iter_direct ? b->n_iter[i0 + i] : h_iter[i];
      b->n_iter[i0 + i] = n;
      if (!c.out_direct && (!use_crc || n >= 3)) /* the reference leaves p_out untouched otherwise (decoder.c:849-861) */
        memcpy(b->out + (size_t)(i0 + i) * b->out_stride, c.h_out + i * out_stride, ob);
    }
  }
  c.chunks.clear();
  c.out_direct = c.iter_direct = false;
  return 0;
}

/* CRC stop with a predicate the GPU cannot evaluate (a caller's own function; or E % 8 != 0, an output mode other than BIT:
 * see t_nrLDPC_dec_params::check_crc).  The decoder is deterministic and the predicate only decides WHERE it stops, so the
 * blocks run all their passes on the GPU, the hard decisions of every pass >= 3 come back, and the predicate is applied on
 * the host exactly as nrLDPC_decoder.c:849-861 applies it: to p_out, after every such pass in order, until it holds.  p_out
 * ends up holding the last pass that was checked, the return value is that pass.  Host buffers, primary device; slow (the
 * full iteration count + one output row per pass over the link) and exact. */
int dec_host_predicate(const nrLDPC_hip_dec_batch_t *b)
{
  const t_nrLDPC_dec_params &p = b->params;
  UseDevice use(g.dev[0]);
  const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  const int mode = p.outMode == nrLDPC_outMode_BIT ? 0 : 1;
  const int ob = out_bytes_of(hc, mode), max_pass = p.numMaxIter + 1;
  if (max_pass < 3) { /* decoder.c:849: never checked, p_out never written, every pass runs */
    for (uint32_t i = 0; i < b->n_blocks; i++)
      b->n_iter[i] = max_pass;
    return 0;
  }
  const uint32_t n_trace = (uint32_t)max_pass - 2, tstride = (uint32_t)align_up(ob, 16);
  const size_t in_stride = align_up(hc.num_llr, 16), per_block = (size_t)n_trace * tstride;
  const uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)64 << 20) / per_block);
  ThreadCtx &c = tls_ctx;
  ldpc_dec_args a;
  memset(&a, 0, sizeof(a));
  a.code = ce->dev;
  a.num_max_iter = p.numMaxIter;
  a.out_mode = mode;
  a.use_crc = 1;
  for (uint32_t k0 = 0; k0 < b->n_blocks; k0 += chunk) {
    const uint32_t n = std::min(chunk, b->n_blocks - k0);
    if (c.ensure(in_stride * n, per_block * n + sizeof(int32_t) * n) != 0)
      return -1;
    for (uint32_t i = 0; i < n; i++) /* through the page-locked staging area (pageable arrays never meet the copy functions) */
      memcpy(c.h_in + i * in_stride, b->llr + (size_t)(k0 + i) * b->llr_stride, hc.num_llr);
    HIP_TRY(hipMemcpyAsync(c.d_in, c.h_in, in_stride * n, hipMemcpyHostToDevice, c.stream));
    a.llr = reinterpret_cast<const int8_t *>(c.d_in); a.llr_stride = (uint32_t)in_stride;
    a.n_iter = reinterpret_cast<int32_t *>(c.d_out + per_block * n);
    HIP_TRY(ldpc_launch_dec_generic_trace(a, hc, n, reinterpret_cast<int8_t *>(c.d_out), tstride, n_trace, c.stream));
    HIP_TRY(hipMemcpyAsync(c.h_out, c.d_out, per_block * n + sizeof(int32_t) * n, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    for (uint32_t i = 0; i < n; i++) {
      int8_t *out = b->out + (size_t)(k0 + i) * b->out_stride;
      int32_t n_iter = max_pass;
      for (int pass = 3; pass <= max_pass; pass++) {
        memcpy(out, c.h_out + (size_t)i * per_block + (size_t)(pass - 3) * tstride, ob);
        if (p.check_crc(reinterpret_cast<uint8_t *>(out), (uint32_t)p.E, (uint8_t)p.crc_type)) {
          n_iter = pass;
          break;
        }
      }
      b->n_iter[k0 + i] = n_iter;
    }
  }
  return 0;
}

} // namespace

extern "C" {

int32_t LDPCdecoder_batch(const nrLDPC_hip_dec_batch_t *b)
{
  if (!b || !b->llr || !b->out || !b->n_iter)
    return set_error("null argument");
  if (b->mem == NRLDPC_HIP_MEM_DEVICE) {
    /* the buffers' own GPU does the work, whatever device is current in the calling thread (and stays current) */
    Device *d = device_of_pointer(b->llr);
    if (!d)
      return -1;
    UseDevice use(*d);
    const t_nrLDPC_dec_params &p = b->params;
    const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
    if (!ce)
      return -1;
    const ldpc_code_desc_t &hc = ce->host;
    const int ob = out_bytes_of(hc, p.outMode == nrLDPC_outMode_BIT ? 0 : 1);
    if (b->llr_stride < (uint32_t)hc.num_llr || b->out_stride < (uint32_t)ob || (b->out_stride & 3))
      return set_error("bad stride");
    ldpc_dec_args a;
    if (fill_dec_args(p, ce, a) != 0)
      return -1;
    if (b->n_blocks == 0)
      return 0;
    if ((reinterpret_cast<uintptr_t>(b->out) & 3))
      return set_error("out must be 4-byte aligned");
    a.llr = b->llr; a.llr_stride = b->llr_stride;
    a.out = b->out; a.out_stride = b->out_stride;
    a.n_iter = b->n_iter;
    hipStream_t s = static_cast<hipStream_t>(b->stream); /* NULL = the legacy default stream */
    return launch_decoder(b->kernel, a, ce, b->n_blocks, s);
  }
  if (ensure_ready() != 0)
    return -1;
  bool host_predicate = false;
  {
    /* parameter checks once, against the primary device's descriptor */
    UseDevice use(g.dev[0]);
    const t_nrLDPC_dec_params &p = b->params;
    const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
    if (!ce)
      return -1;
    const int ob = out_bytes_of(ce->host, p.outMode == nrLDPC_outMode_BIT ? 0 : 1);
    if (b->llr_stride < (uint32_t)ce->host.num_llr || b->out_stride < (uint32_t)ob || (b->out_stride & 3))
      return set_error("bad stride");
    host_predicate = p.check_crc && !crc_on_device(p, ce);
    ldpc_dec_args a;
    if (!host_predicate && fill_dec_args(p, ce, a) != 0)
      return -1;
  }
  if (host_predicate)
    return b->n_blocks ? dec_host_predicate(b) : 0;
  if (b->n_blocks == 0)
    return 0;
  /* host buffers: contiguous block ranges over the sharding devices (code blocks are independent, nothing is
   * exchanged), every device fed over its own link; a handful of blocks stays on the primary device */
  const int parts = b->n_blocks >= 64u * (uint32_t)g.n_shard ? g.n_shard : 1;
  const bool pinned = b->n_blocks >= 16 && host_ptr_is_pinned(b->llr, (size_t)(b->n_blocks - 1) * b->llr_stride + 1);
  int rc = 0;
  for (int k = 0; k < parts && rc == 0; k++) {
    uint32_t lo, hi;
    shard_range(b->n_blocks, k, parts, lo, hi);
    UseDevice use(g.dev[k]);
    rc = dec_host_enqueue(b, lo, hi - lo, pinned);
  }
  for (int k = 0; k < parts; k++) { /* also after an error: nothing stays in flight */
    uint32_t lo, hi;
    shard_range(b->n_blocks, k, parts, lo, hi);
    UseDevice use(g.dev[k]);
    if (rc == 0)
      rc = dec_host_finish(b, lo, hi - lo);
    else
      tls_ctx.drain();
  }
  return rc;
}

namespace {
/* NRLDPC_HIP_CUT=0: per-segment calls always run the whole rate mode */
bool cut_enabled()
{
  static const int v = [] { const char *e = getenv("NRLDPC_HIP_CUT"); return e ? atoi(e) : 1; }();
  return v != 0;
}
/* the rate mode of `ce` cut behind the last column of `llr` that holds a non-zero value, or nullptr when that is the mode
 * itself.  The scan runs backwards over at most the extension columns (a few KB of zeros when it pays, one word when not). */
const CodeEntry *cut_code_for(const CodeEntry *ce, const int8_t *llr)
{
  if (!cut_enabled())
    return nullptr;
  const ldpc_code_desc_t &hc = ce->host;
  const int Z = hc.Z, floor_cols = hc.ncore + 1;
  int i = hc.num_llr; /* bytes [i, num_llr) are zero */
  const int stop = floor_cols * Z;
  while (i - 8 >= stop) {
    uint64_t w;
    memcpy(&w, llr + i - 8, 8);
    if (w)
      break;
    i -= 8;
  }
  while (i > stop && llr[i - 1] == 0)
    i--;
  const int need = std::max((i + Z - 1) / Z, floor_cols);
  if (need >= hc.ncols)
    return nullptr;
  /* per-thread memo: the published table of get_code() is indexed by rate mode; cut codes live in the locked map */
  struct Memo { const CodeEntry *full; int need; const CodeEntry *cut; };
  static thread_local Memo memo[4] = {};
  for (const Memo &m : memo)
    if (m.full == ce && m.need == need)
      return m.cut;
  const CodeEntry *c = get_code_cols(hc.BG, Z, need);
  if (!c)
    return nullptr;
  static thread_local unsigned next = 0;
  memo[next++ & 3] = Memo{ce, need, c};
  return c;
}
} // namespace

int32_t LDPCdecoder(t_nrLDPC_dec_params *p_decParams, uint8_t harq_pid, uint8_t ulsch_id, uint8_t C, int8_t *p_llr,
                    int8_t *p_out, t_nrLDPC_time_stats *p_profiler, decode_abort_t *ab)
{
  (void)harq_pid; (void)ulsch_id; (void)C;
  if (!p_decParams)
    return set_error("null argument");
  /* decoder.c:556-559: a segment of an already failed transport block is not worked on */
  if (ab) {
    pthread_mutex_lock(&ab->mutex_failure);
    const bool failed = ab->failed;
    pthread_mutex_unlock(&ab->mutex_failure);
    if (failed)
      return p_decParams->numMaxIter + 2;
  }
  if (p_profiler)
    meter_start(&p_profiler->total);
  int32_t n_iter = 0;
  int rc = -1;
  const CodeEntry *ce = (p_llr && p_out) ? get_code(p_decParams->BG, p_decParams->Z, p_decParams->R) : nullptr;
  if (!p_llr || !p_out)
    set_error("null argument");
  if (ce) {
    rc = 1;
    /* resident submission path: no runtime call, no shared lock (ldpc_server.inc.cpp); a CRC predicate that has to run on
     * the host goes through the batch entry point (dec_host_predicate) */
    if (!(p_decParams->check_crc && !crc_on_device(*p_decParams, ce)) && srv_ready(srv) == 0) {
      /* CRC stop with the columns at the end of the rate mode all zero -- what nr_ulsch_decoding.c hands over for a high-rate
       * first transmission: the rows that close on them send zeros in every pass (ldpc_graph.h LDPC_R_COLS), so the call is
       * served on the mode cut behind its last column that holds anything; fewer bytes over the link both ways */
      const CodeEntry *cut = p_decParams->check_crc ? cut_code_for(ce, p_llr) : nullptr;
      if (cut)
        rc = srv_decode(p_decParams, cut, p_llr, p_out, &n_iter, ab,
                        out_bytes_of(ce->host, p_decParams->outMode == nrLDPC_outMode_BIT ? 0 : 1));
      if (!cut || rc == 1)
        rc = srv_decode(p_decParams, ce, p_llr, p_out, &n_iter, ab);
    }
    if (rc == 1) { /* server switched off, or a code it cannot hold: one launch per call on this thread's stream */
      const int ob = out_bytes_of(ce->host, p_decParams->outMode == nrLDPC_outMode_BIT ? 0 : 1);
      nrLDPC_hip_dec_batch_t b;
      memset(&b, 0, sizeof(b));
      b.params = *p_decParams;
      b.n_blocks = 1;
      b.llr = p_llr; b.llr_stride = (uint32_t)ce->host.num_llr;
      b.out = p_out; b.out_stride = (uint32_t)align_up(ob, 4);
      b.n_iter = &n_iter;
      b.mem = NRLDPC_HIP_MEM_HOST;
      rc = LDPCdecoder_batch(&b);
    }
  }
  if (rc != 0) {
    /* The reference's LDPCdecoder cannot fail and its callers test only `decodeIterations <= numMaxIter`
     * (nr_ulsch_decoding.c:219-222, nr_dlsch_decoding.c:259): report an internal error as "not decoded" so that they
     * NACK instead of taking an untouched p_out for a code word.  nrLDPC_hip_last_error() has the reason. */
    fprintf(stderr, "[libldpc_hip] LDPCdecoder failed: %s\n", tls_error.c_str());
    n_iter = p_decParams->numMaxIter + 1;
  }
  if (p_profiler)
    meter_stop(&p_profiler->total);
  if (n_iter > p_decParams->numMaxIter && ab) { /* decoder.c:190-193 */
    pthread_mutex_lock(&ab->mutex_failure);
    ab->failed = true;
    pthread_mutex_unlock(&ab->mutex_failure);
  }
  return n_iter;
}

int32_t LDPCencoder_batch(const nrLDPC_hip_enc_batch_t *b)
{
  if (!b || !b->in || !b->out)
    return set_error("null argument");
  /* device buffers: their own GPU; host buffers: the primary device (the call is bound by the 8x larger, one byte per
   * bit, output crossing the link -- use nrLDPC_hip_dlsch_encode to shard transport blocks over GPUs) */
  Device *dv = b->mem == NRLDPC_HIP_MEM_DEVICE ? device_of_pointer(b->in) : (ensure_ready() == 0 ? &g.dev[0] : nullptr);
  if (!dv)
    return -1;
  UseDevice use(*dv);
  const CodeEntry *ce = get_code(b->BG, b->Zc, b->BG == 1 ? 13 : 15);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  const int K = hc.kb_full * hc.Z, in_bytes = (K + 7) / 8, N = (hc.ncols - 2) * hc.Z;
  if (b->Kb < 1 || b->Kb > hc.kb_full)
    return set_error("bad Kb");
  if (b->in_stride < (uint32_t)in_bytes || b->out_stride < (uint32_t)N)
    return set_error("bad stride");
  if (b->n_blocks == 0)
    return 0;
  ldpc_enc_args a;
  a.jobs = nullptr;
  a.code = ce->dev;
  a.Kb = b->Kb;
  if (b->mem == NRLDPC_HIP_MEM_DEVICE) {
    a.in = b->in; a.in_stride = b->in_stride;
    a.out = b->out; a.out_stride = b->out_stride;
    hipStream_t s = static_cast<hipStream_t>(b->stream); /* NULL = the legacy default stream */
    HIP_TRY(ldpc_launch_enc(a, hc, b->n_blocks, s));
    return 0;
  }
  ThreadCtx &c = tls_ctx;
  const size_t in_stride = align_up(in_bytes, 16), out_stride = align_up(N, 16);
  if (c.ensure(in_stride * b->n_blocks, out_stride * b->n_blocks) != 0)
    return -1;
  for (uint32_t i = 0; i < b->n_blocks; i++)
    memcpy(c.h_in + i * in_stride, b->in + (size_t)i * b->in_stride, in_bytes);
  HIP_TRY(hipMemcpyAsync(c.d_in, c.h_in, in_stride * b->n_blocks, hipMemcpyHostToDevice, c.stream));
  a.in = c.d_in; a.in_stride = (uint32_t)in_stride;
  a.out = c.d_out; a.out_stride = (uint32_t)out_stride;
  HIP_TRY(ldpc_launch_enc(a, hc, b->n_blocks, c.stream));
  HIP_TRY(hipMemcpyAsync(c.h_out, c.d_out, out_stride * b->n_blocks, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  for (uint32_t i = 0; i < b->n_blocks; i++)
    memcpy(b->out + (size_t)i * b->out_stride, c.h_out + i * out_stride, N);
  return 0;
}

int32_t LDPCencoder(uint8_t **input, uint8_t **output, encoder_implemparams_t *impp)
{
  /* Returns 0, or -1 on bad parameters / an internal error -- which the reference's callers ignore
   * (nr_dlsch_coding.c:171, nr_ulsch_coding.c:167 discard the return value), as they do for the reference library. */
  if (!input || !output || !impp)
    return set_error("null argument");
  /* ldpc_encoder_optim8segmulti.c:64-65: this call covers segments 8*macro_num .. min(n_segments, +8) */
  const unsigned first = 8 * impp->macro_num;
  const unsigned last = impp->n_segments > 8 * (impp->macro_num + 1) ? 8 * (impp->macro_num + 1) : impp->n_segments;
  if (last <= first)
    return 0;
  const CodeEntry *ce = get_code(impp->BG, (int)impp->Zc, impp->BG == 1 ? 13 : 15);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  const int K = hc.kb_full * hc.Z, in_bytes = (K + 7) / 8, N = (hc.ncols - 2) * hc.Z;
  /* impp->K = bits the caller supplies per segment.  Every caller in the stack passes the full K = 22*Zc / 10*Zc
   * (nr_dlsch_coding.c:366); ldpctest passes its -l block length, which may be shorter: the missing bits are zeros and
   * the code word is cut to the mother rate, out = c[2Zc..K') || d[0 .. rate*K' - K' + 2Zc)  (ldpc_encoder.c:82-92,
   * 248-251; ldpc_encoder_optim8segmulti.c:112-118,175-208). */
  const int Kp = (int)impp->K, rate = impp->BG == 1 ? 3 : 5;
  if (Kp > K || Kp <= 2 * hc.Z)
    return set_error("K must be in (2*Zc, 22*Zc] (BG1) or (2*Zc, 10*Zc] (BG2)");
  const bool shortened = Kp != K;
  const int n_info = Kp - 2 * hc.Z, n_par = rate * Kp - n_info; /* bytes of the two output parts */
  if ((int)impp->Kb < 1 || (int)impp->Kb > hc.kb_full)
    return set_error("bad Kb");
  const unsigned n = last - first;
  /* the reference's four meters (ldpc_encoder_optim8segmulti.c:120-210), with this library's phases: tinput = staging
   * the segments, tprep = nothing, tparity = the encode on the GPU, toutput = handing the code words back */
  if (!shortened && srv_ready(srv_e) == 0) {
    meter_start(impp->tinput);
    const int rc = srv_encode(ce, (int)impp->Kb, input, output, first, n, impp->tinput, impp->tprep, impp->tparity, impp->toutput);
    if (rc <= 0)
      return rc;
  }
  UseDevice use(g.dev[0]);
  ThreadCtx &c = tls_ctx;
  const size_t in_stride = align_up(in_bytes, 16), out_stride = align_up(N, 16);
  if (c.ensure(in_stride * n, out_stride * n) != 0)
    return -1;
  meter_start(impp->tinput);
  for (unsigned j = 0; j < n; j++) {
    uint8_t *dst = c.h_in + j * in_stride;
    const int nb = (Kp + 7) / 8;
    memcpy(dst, input[first + j], nb);
    if (shortened) {
      if (Kp & 7)
        dst[nb - 1] &= (uint8_t)(0xff << (8 - (Kp & 7)));
      memset(dst + nb, 0, in_bytes - nb);
    }
  }
  meter_stop(impp->tinput);
  meter_start(impp->tprep);
  meter_stop(impp->tprep);
  meter_start(impp->tparity);
  ldpc_enc_args a;
  a.jobs = nullptr;
  a.code = ce->dev;
  a.Kb = (int)impp->Kb;
  a.in = c.d_in; a.in_stride = (uint32_t)in_stride;
  a.out = c.d_out; a.out_stride = (uint32_t)out_stride;
  hipError_t e = hipMemcpyAsync(c.d_in, c.h_in, in_stride * n, hipMemcpyHostToDevice, c.stream);
  if (e == hipSuccess)
    e = ldpc_launch_enc(a, hc, n, c.stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(c.h_out, c.d_out, out_stride * n, hipMemcpyDeviceToHost, c.stream);
  const hipError_t e2 = hipStreamSynchronize(c.stream); /* also on the error path: nothing stays in flight */
  meter_stop(impp->tparity);
  if (e != hipSuccess || e2 != hipSuccess)
    return set_error("LDPCencoder", e != hipSuccess ? e : e2);
  meter_start(impp->toutput);
  for (unsigned j = 0; j < n; j++) {
    const uint8_t *src = c.h_out + j * out_stride;
    if (!shortened) {
      memcpy(output[first + j], src, N);
    } else {
      memcpy(output[first + j], src, n_info);
      memcpy(output[first + j] + n_info, src + (K - 2 * hc.Z), n_par);
    }
  }
  meter_stop(impp->toutput);
  return 0;
}

} /* extern "C" */

#include "tb_api.inc.cpp"
#include "dec_jobs.inc.cpp"
#include "tb_offload.inc.cpp"
