/*
 * tb_api.inc.cpp -- transport-block chain entry points (included at the end of ldpc_api.cpp; shares its
 * library state).  Host part = the parameter arithmetic of nr_dlsch_encoding()/nr_ulsch_decoding()
 * (segmentation, E per segment, decoder rate mode, rate-matching geometry: nr_coding_host.c) and the job lists;
 * every byte of payload/LLR data is touched on the GPU only (tb_chain.hip + the codec kernels).
 */

namespace {

struct DevBuf { /* growable device buffer */
  uint8_t *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n)
  {
    if (n <= cap)
      return 0;
    retire(p, false); /* parked, not freed: see retire() */
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 65536;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), want));
    cap = want;
    return 0;
  }
};
struct PinBuf { /* growable pinned host buffer */
  uint8_t *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n)
  {
    if (n <= cap)
      return 0;
    retire(p, true);
    p = nullptr;
    cap = 0;
    const size_t want = n + n / 4 + 65536;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocDefault));
    cap = want;
    return 0;
  }
};

/* What a call derives from its descriptor array alone (job lists on the device, launch shapes, buffer extents): kept
 * per thread and per direction, reused as long as the next call's descriptors are byte-identical -- a scheduler that
 * repeats an allocation slot after slot then skips the parameter arithmetic and the job upload. */
struct TbPlan {
  std::vector<uint8_t> key;
  DevBuf jobs_d;
  bool valid = false;
  size_t n_seg = 0, n_aux = 0, scratch_top = 0;
  size_t ext[6] = {0, 0, 0, 0, 0, 0}; /* [lo, hi) of the payload, coded and harq ranges the blocks touch */
  uint32_t rx_lds_elems = 8; /* LDS the de-matching kernel needs per workgroup (int16 slots) */
  bool out_dense = true; /* the blocks' outputs tile their range: one copy back; else one per block (nothing between them is touched) */
  size_t off[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int threads[4] = {64, 64, 64, 64}, lds[4] = {0, 0, 0, 0};
  size_t n_fast = 0, n_gen = 0; /* encode */
  /* decode: the decoder launches of the batch.  A job-array launch has ONE workgroup shape (threads, LDS) = the maxima over
   * its jobs, so a batch that mixes code sizes is cut into launches by how many workgroups of a job's shape a CU holds
   * (1, 2, 4, 8, 16+): a Zc = 8 segment does not occupy the LDS of a Zc = 384 one.  kind 0: fast kernel, 1: generic
   * kernel, 2 / 3: several small segments per workgroup (f_sub = 1 / 4; grp_off = their ldpc_dec_mgroup array) */
  struct DecLaunch { int kind; size_t jobs_off, grp_off; uint32_t n; int threads, lds; bool fused; bool all_fused = false; int max_llr = 0; uint32_t lrow = 0; bool mute = false;
                     int zc = 0; /* kind 0: the lifting size every job of the launch has (one block per workgroup), else 0 -- selects the
                                    kernels' instantiation for that size (ldpc_dec_fast_block.h ZC) */ };
  std::vector<DecLaunch> dec;
  std::vector<int32_t> llr_len; /* decode: the llrLen every TB leaves with */
  /* decode: segments / transport blocks that go through the separate de-matching, reassembly and verdict kernels (the others
   * are served from LLRs to payload by the fused segment kernel, tb_rx_fused.hip); legacy_off = their copy of the segment jobs */
  size_t n_legacy_seg = 0, n_legacy_tb = 0, legacy_off = 0;
  bool any_fused = false;
  /* decode, host-resident soft buffers: runs of rows of the caller's array (first int16, rows, int16 per row that the kernels
   * look at, uploaded before the call?) -- a first transmission is cleared on the device and never uploaded; what lies behind
   * a row's Ncb values in the caller's array is never touched */
  struct HarqRun { size_t first; uint32_t rows, width; bool upload; };
  std::vector<HarqRun> harq_runs;
  /* staged outputs that do not tile their range (the caller aligned every block's payload, say): runs of equally sized,
   * equally spaced blocks -- one strided copy each instead of one copy per block (64 copies of a slot's 64 blocks cost
   * 0.9 ms of enqueueing; nothing between the blocks is touched either way) */
  struct OutRun { size_t first, width, pitch; uint32_t rows; };
  std::vector<OutRun> out_runs;
  void build_out_runs(const uint64_t *off, const size_t *len, uint32_t n)
  {
    out_runs.clear();
    for (uint32_t i = 0; i < n; i++) {
      if (!out_runs.empty()) {
        OutRun &r = out_runs.back();
        const size_t last = r.first + (size_t)(r.rows - 1) * r.pitch;
        if (len[i] == r.width && off[i] > last && (r.rows == 1 ? off[i] - last >= r.width : off[i] - last == r.pitch)) {
          if (r.rows == 1)
            r.pitch = (size_t)off[i] - last;
          r.rows++;
          continue;
        }
      }
      out_runs.push_back(OutRun{(size_t)off[i], len[i], len[i], 1u});
    }
  }
  uint64_t stamp = 0; /* LRU */
  /* the descriptors tb[0 .. n_tb) of a call (one device's share of the batch); salt = whatever else the plan depends on */
  bool matches(const nrLDPC_hip_tb_t *tb, uint32_t n_tb, const uint64_t salt[3]) const
  {
    const size_t n = (size_t)n_tb * sizeof(nrLDPC_hip_tb_t);
    return valid && key.size() == n + 32 && memcmp(key.data(), &n_tb, 4) == 0 && memcmp(key.data() + 8, salt, 24) == 0 &&
           memcmp(key.data() + 32, tb, n) == 0;
  }
  /* any descriptor array as bytes (LDPCdecoder_jobs) */
  bool matches_raw(const void *d, size_t n, const uint64_t salt[3]) const
  {
    return valid && key.size() == n + 32 && memcmp(key.data() + 8, salt, 24) == 0 && memcmp(key.data() + 32, d, n) == 0;
  }
  void remember_raw(const void *d, size_t n, const uint64_t salt[3])
  {
    key.assign(n + 32, 0);
    memcpy(key.data() + 8, salt, 24);
    memcpy(key.data() + 32, d, n);
    valid = true;
  }
  void remember(const void *tb_bytes, uint32_t n_tb, const uint64_t salt[3])
  {
    const size_t n = (size_t)n_tb * sizeof(nrLDPC_hip_tb_t);
    key.assign(n + 32, 0);
    memcpy(key.data(), &n_tb, 4);
    memcpy(key.data() + 8, salt, 24);
    memcpy(key.data() + 32, tb_bytes, n);
    valid = true;
  }
};

/* A thread's plans of one direction: a handful, least recently used one replaced -- a caller that alternates between a few
 * allocations (parallel.ShardedUlsch sends a slot as three chunks per rank; a large host-buffer call is worked on in up to
 * eight pieces, first transmissions and retransmissions alternating; a scheduler's DL / UL patterns) finds each of them
 * again instead of rebuilding the one plan every call. */
#define TB_PLAN_SLOTS 24
struct TbPlanCache {
  TbPlan slot[TB_PLAN_SLOTS];
  uint64_t clock = 0;
  TbPlan *find(const nrLDPC_hip_tb_t *tb, uint32_t n_tb, const uint64_t salt[3])
  {
    for (TbPlan &p : slot)
      if (p.matches(tb, n_tb, salt)) {
        p.stamp = ++clock;
        return &p;
      }
    return nullptr;
  }
  TbPlan *find_raw(const void *d, size_t n, const uint64_t salt[3])
  {
    for (TbPlan &p : slot)
      if (p.matches_raw(d, n, salt)) {
        p.stamp = ++clock;
        return &p;
      }
    return nullptr;
  }
  TbPlan &victim() /* an unused slot, else the least recently used one; the caller rebuilds it */
  {
    TbPlan *v = &slot[0];
    for (TbPlan &p : slot) {
      if (!p.valid) { v = &p; break; }
      if (p.stamp < v->stamp) v = &p;
    }
    v->valid = false;
    v->stamp = ++clock;
    return *v;
  }
};

#define TB_SIDE_STREAMS 3
struct TbCtx {
  TbPlanCache tx, rx, cb; /* cb: mixed code-block batches (dec_jobs.inc.cpp) */
  DevBuf scratch, jobs_d, io_payload, io_coded, io_harq, io_small, trace_d;
  PinBuf jobs_h, small_h, payload_h, coded_h, harq_h;
  /* Pageable arrays of the caller never meet the runtime's copy functions: they are bounced through this thread's
   * page-locked areas by the CPU (in: before the copy is enqueued; out: in the finish call, after the stream has drained).
   * Handing pageable memory to hipMemcpy*Async makes the runtime page-lock the caller's pages behind the scenes, and a
   * process that allocates and frees such arrays call after call eventually took a GPU memory fault inside a soft-buffer
   * copy -- once in ten runs of the test suite, all kernels of the call already finished (profiles/r04/README.md). */
  /* ev: the event behind the device -> mirror copy this hand-over waits for (nullptr: the caller has drained the stream).  The
   * soft-buffer rows of a large call come down in runs of ~8 MB with an event behind each, so that the CPU hands run k over
   * -- bounce_copy(): a few threads -- while the link brings run k + 1 (profiles/r05/slot_chain_host.json). */
  struct FinCopy { uint8_t *dst; const uint8_t *src; size_t width, rows, dpitch, spitch; hipEvent_t ev; };
  std::vector<FinCopy> fin_copies;
  std::vector<hipEvent_t> fin_ev; /* pool of events for the runs; fin_ev_used of them belong to the call in flight */
  size_t fin_ev_used = 0;
  size_t harq_h_used = 0; /* bytes of harq_h the pieces of the call in flight have taken (a call may come in several pieces) */
  hipEvent_t next_fin_event()
  {
    if (fin_ev_used == fin_ev.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
        return nullptr;
      fin_ev.push_back(e);
    }
    return fin_ev[fin_ev_used++];
  }
  void finish_copies(bool drained) /* drained: the caller has waited for the stream already */
  {
    harq_h_used = 0;
    hipEvent_t waited = nullptr;
    for (const FinCopy &f : fin_copies) {
      if (f.ev && f.ev != waited && !drained) {
        (void)hipEventSynchronize(f.ev);
        waited = f.ev;
      } else if (!f.ev && !drained) {
        (void)hipStreamSynchronize(own);
        drained = true;
      }
      bounce_copy(f.dst, f.dpitch, f.src, f.spitch, f.width, f.rows);
    }
    fin_copies.clear();
    fin_ev_used = 0;
  }
  /* the host-buffer decode in flight: what tb_rx_finish has to hand over from payload_h (0 bytes: the kernels wrote the
   * caller's page-locked array themselves) */
  size_t fin_pay_lo = 0, fin_pay_n = 0;
  hipStream_t own = nullptr, last = nullptr;
  hipStream_t aux = nullptr;          /* copy lane of the chunked host-buffer decode (tb_rx_enqueue_host) */
  hipStream_t side[TB_SIDE_STREAMS] = {nullptr, nullptr, nullptr}; /* decoder launches side by side (NRLDPC_HIP_TB_OVERLAP) */
  hipEvent_t side_ev[TB_SIDE_STREAMS + 1] = {nullptr, nullptr, nullptr, nullptr};
  /* nrLDPC_hip_chain_timing: HIP events around the stages of this thread's UL-SCH calls, on the stream the kernels run on */
  bool timing = false, timed = false;
  hipEvent_t tev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> chunk_ev;
  hipEvent_t uploaded = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr; /* device-resident batches cut over several GPUs (tb_run_sharded) */
  bool pending = false;
  void drain() /* nothing of this thread's stays in flight (error paths, thread exit) */
  {
    if (aux)
      (void)hipStreamSynchronize(aux);
    for (hipStream_t q : side)
      if (q)
        (void)hipStreamSynchronize(q);
    if (own)
      (void)hipStreamSynchronize(own);
    pending = false;
    fin_copies.clear();
    fin_ev_used = 0;
    harq_h_used = 0;
  }
};
thread_local CtxHolder<TbCtx> tls_tb_holder[NRLDPC_HIP_MAX_DEVICES]; /* pooled like ThreadCtx, one per logical device */
#define tls_tb (tls_tb_holder[cur_dev_index()].get(cur_dev_index()))

struct Arena { /* bump allocator over the scratch buffer, 16-byte granules */
  size_t top = 0;
  size_t take(size_t n)
  {
    const size_t o = top;
    top += align_up(n, 16);
    return o;
  }
};

/* on the current device (UseDevice) */
/* staged: the part works on copies of the caller's buffers, on this thread's own stream; else in place on s_direct */
int tb_begin(hipStream_t &s, hipStream_t s_direct, bool staged)
{
  TbCtx &c = tls_tb;
  if (!c.own) {
    HIP_TRY(hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c.uploaded, hipEventDisableTiming));
  }
  s = staged ? c.own : s_direct;
  if (c.last && c.last != s) { /* scratch is reused: calls on different streams are serialised */
    HIP_TRY(hipStreamSynchronize(c.last));
  }
  c.last = s;
  return 0;
}

/* before the pinned job staging buffer is rewritten: the previous upload must have left it.  (Calls that reuse a cached
 * plan upload nothing and never get here, which keeps them free of host synchronisation -- capturable in a HIP graph.) */
int tb_wait_upload(TbCtx &c)
{
  if (c.pending) {
    HIP_TRY(hipEventSynchronize(c.uploaded));
    c.pending = false;
  }
  return 0;
}

/* NRLDPC_HIP_TB_ABORT=0: every segment of a lost transport block is decoded to the end (the reference's behaviour when its
 * workers never overlap); default: siblings of a failed segment give up at their next pass (decoder.c:190-193, 556-559) */
/* NRLDPC_HIP_TB_MULTI=0: every segment of the chain gets a workgroup of its own.  Default: when a batch holds enough
 * segments to fill the GPU anyway, small segments of the same code, iteration cap and CRC share workgroups
 * (ldpc_dec_fast_mblock.h: a lifted row of a small code fills only a fraction of a 64-item task). */
/* NRLDPC_HIP_TB_CLASSES=0: one decoder launch per kernel for the whole batch (workgroup shape = the largest segment's) */
bool tb_classes_enabled() /* (read when a plan is built, part of its key: the tests switch it between calls) */
{
  const char *e = getenv("NRLDPC_HIP_TB_CLASSES");
  return !(e && atoi(e) == 0);
}
#define TB_MULTI_MIN_PER_CU 64 /* shared workgroups from this many small segments per CU on (see profiles/r03/README.md) */
/* read when a plan is built (not cached: the tests switch it between calls) */
int tb_multi_mode()
{
  const char *e = getenv("NRLDPC_HIP_TB_MULTI");
  return e ? atoi(e) : 1;
}

/* NRLDPC_HIP_TB_FUSED=0: de-matching, decoding, reassembly and verdict as four launches (the round-3 path, kept as the
 * cross-check); default: one fused segment kernel wherever the fast decoder serves the segment (tb_rx_fused.hip) */
/* NRLDPC_HIP_TB_PULL: page-locked host LLRs of nrLDPC_hip_ulsch_decode -- 0: always copied to the device first (in chunks
 * that overlap with the decoding of the chunks before them, as pageable LLRs are); 1 (default) / 2: read in place by the
 * segments' workgroups, 1 only for calls below NRLDPC_HIP_TB_PULL_MAX_MB when that is set.  Measured on the 64-block slot
 * (31.4 MB of LLRs, profiles/r04/slot_chain_host.json): in place 0.70 ms -- the launch's workgroups keep 48 GB/s on the
 * link while others decode, one launch, no copy -> kernel edge -- against 0.75 ms for the chunked copy-engine pipeline and
 * 0.56 ms for the bare transfer. */
int tb_pull_mode()
{
  const char *e = getenv("NRLDPC_HIP_TB_PULL");
  return e ? atoi(e) : 1;
}

/* NRLDPC_HIP_TB_FILL=0: every workgroup-shape class of a mixed call gets launches of its own (the round-3 plan); default:
 * a launch's last workgroup round is filled with jobs of the smaller classes.  Read when a plan is built, part of its key. */
int tb_fill_mode()
{
  const char *e = getenv("NRLDPC_HIP_TB_FILL");
  return e ? atoi(e) : 1;
}

int tb_fused_mode()
{
  const char *e = getenv("NRLDPC_HIP_TB_FUSED");
  return e ? atoi(e) : 1;
}

/* NRLDPC_HIP_TB_TRUNC=0: first transmissions are decoded on the whole graph of the reference's rate mode (cross-check) */
bool tb_trunc_enabled()
{
  static const int v = [] {
    const char *e = getenv("NRLDPC_HIP_TB_TRUNC");
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return v != 0;
}

bool tb_lrow_enabled()
{
  static const int v = [] { const char *e = getenv("NRLDPC_HIP_TB_LROW"); return e ? atoi(e) : 1; }();
  return v != 0;
}

bool tb_abort_enabled()
{
  static const int v = [] {
    const char *e = getenv("NRLDPC_HIP_TB_ABORT");
    return (e && atoi(e) == 0) ? 0 : 1;
  }();
  return v != 0;
}

int tb_validate(const nrLDPC_hip_tb_t &t)
{
  if (t.A == 0 || (t.A & 7) || (t.BG != 1 && t.BG != 2) || t.rv > 3 || t.Nl == 0 ||
      !(t.Qm == 2 || t.Qm == 4 || t.Qm == 6 || t.Qm == 8) || t.G == 0 || t.G % (t.Nl * t.Qm))
    return set_error("invalid transport block parameters");
  if (t.A + 24 > TB_CRC24A_POW_LEN)
    return set_error("transport block too large");
  return 0;
}

/* upload `n` bytes of jobs staged at c.jobs_h.p to `dst` */
int tb_upload_jobs(TbCtx &c, uint8_t *dst, size_t n, hipStream_t s)
{
  HIP_TRY(hipMemcpyAsync(dst, c.jobs_h.p, n, hipMemcpyHostToDevice, s));
  HIP_TRY(hipEventRecord(c.uploaded, s));
  c.pending = true;
  return 0;
}

} // namespace

extern "C" {

int32_t nrLDPC_hip_segmentation(uint32_t B, uint8_t BG, uint32_t *C, uint32_t *K, uint32_t *Zc, uint32_t *F)
{
  nr_hip_seg_t s;
  if ((BG != 1 && BG != 2) || nr_hip_segmentation(B, BG, &s) != 0)
    return -1;
  if (C) *C = s.C;
  if (K) *K = s.K;
  if (Zc) *Zc = s.Zc;
  if (F) *F = s.F;
  return (int32_t)s.Kb;
}
uint32_t nrLDPC_hip_get_E(uint32_t G, uint32_t C, uint32_t Qm, uint32_t Nl, uint32_t r) { return nr_hip_get_E(G, C, Qm, Nl, r); }
/* columns of the code the chain decodes a segment on (nrLDPC_hip.h) */
int32_t nrLDPC_hip_ulsch_decoder_columns(int32_t BG, uint32_t Zc, uint32_t C, uint32_t F, uint32_t K, uint32_t Tbslbrm, int32_t rv,
                                         uint32_t E, int32_t round, int32_t R)
{
  ldpc_code_desc_t *d = new ldpc_code_desc_t;
  const int rc = ldpc_build_code_desc(BG, (int)Zc, R, d);
  const int ncols = d->ncols, ncore = d->ncore;
  delete d;
  nr_hip_rm_t rm;
  if (rc != 0 || nr_hip_rate_match_geometry(Tbslbrm, BG, Zc, C, F, K, rv, E, &rm) != 0)
    return -1;
  if (!tb_trunc_enabled() || round != 0)
    return ncols;
  return std::min(ncols, std::max((int)nr_hip_first_tx_columns(&rm, E, Zc), ncore + 1));
}

int32_t nrLDPC_hip_get_R_ldpc_decoder(int32_t rvidx, int32_t E, int32_t BG, int32_t Z, int32_t *llrLen, int32_t round)
{
  return nr_hip_get_R_ldpc_decoder(rvidx, E, BG, Z, llrLen, round);
}

} /* extern "C" */

namespace {

/* [lo, hi) ranges of the caller's buffers that the transport blocks tb[0 .. n) touch (bytes / int16 elements) */
struct TbExtent {
  size_t pay_lo = SIZE_MAX, pay_hi = 0, cod_lo = SIZE_MAX, cod_hi = 0, harq_lo = SIZE_MAX, harq_hi = 0;
  size_t pay_sum = 0, cod_sum = 0, harq_sum = 0; /* bytes / elements the blocks own: == hi - lo when they tile their range */
  void add(size_t &lo, size_t &hi, size_t a, size_t b)
  {
    lo = std::min(lo, a);
    hi = std::max(hi, b);
  }
};

/* ---- TX: transport blocks [tb0, tb0+ntb) of b on the current device -------------------------------------------------
 * staged (host buffers, or another GPU's memory -- a device-resident batch cut over several GPUs): the device works on copies
 * of exactly the byte ranges its blocks touch (job offsets stay the caller's: the device pointers are biased by the range
 * start); enqueue only, tb_tx_finish() waits for the copies back to the host */
/* diagnostics: NRLDPC_HIP_DEBUG_SYNC=1 waits after every stage of a receive call and names the stage that failed */
#define TB_DEBUG_STAGE(label)                                                                \
  do {                                                                                       \
    static const int dbg_ = [] { const char *e = getenv("NRLDPC_HIP_DEBUG_SYNC"); return e ? atoi(e) : 0; }(); \
    hipStreamCaptureStatus cs_ = hipStreamCaptureStatusNone;                                 \
    if (dbg_ && hipStreamIsCapturing(s, &cs_) == hipSuccess && cs_ == hipStreamCaptureStatusNone) { \
      const hipError_t e_ = hipStreamSynchronize(s);                                        \
      if (e_ != hipSuccess)                                                                  \
        return set_error("stage failed: " label, e_);                                       \
      fprintf(stderr, "[tb_rx dbg] ok: %s (tb0 %u ntb %u staged %d)\n", label, tb0, ntb, (int)staged);                \
    }                                                                                        \
  } while (0)

int tb_tx_enqueue(const nrLDPC_hip_tb_batch_t *b, uint32_t tb0, uint32_t ntb, bool staged, hipStream_t s_direct)
{
  hipStream_t s;
  if (tb_begin(s, s_direct, staged) != 0)
    return -1;
  if (ntb == 0)
    return 0;
  const nrLDPC_hip_tb_t *tbs = b->tb + tb0;
  TbCtx &c = tls_tb;
  const bool fused = ldpc_enc_is_packed() != 0;
  const uint64_t salt[3] = {(fused ? 1u : 0u) | (tb_trunc_enabled() ? 2u : 0u), 0, 0};
  TbPlan *hit = c.tx.find(tbs, ntb, salt);
  TbPlan &pl = hit ? *hit : c.tx.victim();
  if (!hit) {
    std::vector<tb_tx_tb_job> tbj(ntb);
    std::vector<tb_tx_seg_job> sj;
    std::vector<ldpc_enc_job> ej;
    std::vector<tb_crc_chunk_job> cj;
    Arena ar;
    int enc_threads = 64, enc_lds = 0;
    TbExtent ex;
    size_t total_payload = 0;
    for (uint32_t i = 0; i < ntb; i++)
      total_payload += tbs[i].A / 8;
    /* tb_jobs.h; NRLDPC_HIP_TB_CRC_CHUNK = 1 / 2 forces short / long pieces (A/B knob, read when a plan is built) */
    static const int chunk_env = [] { const char *e = getenv("NRLDPC_HIP_TB_CRC_CHUNK"); return e ? atoi(e) : 0; }();
    const uint32_t crc_chunk = chunk_env == 1 ? TB_CRC_CHUNK_SMALL : chunk_env == 2 ? TB_CRC_CHUNK
                               : total_payload <= 256u * 1024u ? TB_CRC_CHUNK_SMALL : TB_CRC_CHUNK;
    for (uint32_t i = 0; i < ntb; i++) {
      const nrLDPC_hip_tb_t &t = tbs[i];
      if (tb_validate(t) != 0)
        return -1;
      /* nr_dlsch_coding.c:300-331 */
      const uint32_t B = t.A + (t.A > NR_HIP_MAX_PDSCH_TBS ? 24 : 16);
      nr_hip_seg_t sg;
      if (nr_hip_segmentation(B, t.BG, &sg) != 0)
        return set_error("nr_segmentation: unsupported block size");
      const CodeEntry *ce = get_code(t.BG, (int)sg.Zc, t.BG == 1 ? 13 : 15);
      if (!ce)
        return -1;
      tbj[i].payload_off = t.payload_off;
      tbj[i].b_off = ar.take(B / 8 + 4);
      tbj[i].A = t.A;
      tbj[i].B = B;
      tbj[i].crc_type = t.A > NR_HIP_MAX_PDSCH_TBS ? NR_HIP_CRC24_A : NR_HIP_CRC16;
      ex.add(ex.pay_lo, ex.pay_hi, (size_t)t.payload_off, (size_t)t.payload_off + t.A / 8);
      ex.add(ex.cod_lo, ex.cod_hi, (size_t)t.coded_off, (size_t)t.coded_off + t.G);
      ex.cod_sum += t.G;
      const uint32_t chunk0 = (uint32_t)cj.size();
      for (uint32_t fb = 0; fb < t.A / 8; fb += crc_chunk)
        cj.push_back(tb_crc_chunk_job{i, fb | (crc_chunk == TB_CRC_CHUNK_SMALL ? 0x80000000u : 0u)});
      const ldpc_code_desc_t &hc = ce->host;
      const int N = (hc.ncols - 2) * hc.Z;
      int nthr, nlds;
      ldpc_enc_launch_shape(hc, &nthr, &nlds);
      enc_threads = std::max(enc_threads, nthr);
      enc_lds = std::max(enc_lds, nlds);
      uint32_t r_offset = 0;
      const CodeEntry *cut_ce = nullptr;
      int cut_need = 0;
      for (uint32_t r = 0; r < sg.C; r++) {
        tb_tx_seg_job j;
        memset(&j, 0, sizeof(j));
        j.b_off = fused ? t.payload_off : tbj[i].b_off; /* fused: the segment's bytes come straight from the payload array */
        if (!fused) { /* the fused kernel keeps c and d in LDS */
          j.c_off = ar.take(sg.K / 8 + 4);
          j.d_off = ar.take(N);
        }
        j.out_off = t.coded_off + r_offset;
        j.r = r; j.C = sg.C; j.Kprime = sg.Kprime; j.L = sg.L; j.K = sg.K;
        j.E = nr_hip_get_E(t.G, sg.C, t.Qm, t.Nl, r);
        j.Qm = t.Qm;
        nr_hip_rm_t rm;
        if (nr_hip_rate_match_geometry(t.tbslbrm, t.BG, sg.Zc, sg.C, sg.F, sg.K, t.rv, j.E, &rm) != 0)
          return set_error("nr_rate_matching: invalid parameters");
        j.Foffset = rm.Foffset; j.Fin = rm.Fin; j.V = rm.V; j.rank0 = rm.rank0;
        j.tb = i;
        if (r + 1 == sg.C) { /* the TB CRC bytes sit at the end of the last segment's share of b */
          j.crc_len = (B - t.A) / 8;
          j.crc_pos = t.A / 8 - r * ((sg.Kprime - sg.L) >> 3);
          j.crc_chunk0 = chunk0;
          j.crc_nchunks = (uint32_t)cj.size() - chunk0;
        }
        r_offset += j.E;
        sj.push_back(j);
        /* the fused kernel selects the E transmitted bits straight from the code word in LDS: parity columns behind the last
         * position the selection reaches are never looked at, so their rows are not computed -- the segment is encoded on the
         * code cut to the columns it sends (at MCS 27: 2 of BG1's 42 extension rows).  Same output bits. */
        const CodeEntry *ce_seg = ce;
        if (fused && tb_trunc_enabled()) {
          const int need = std::max((int)nr_hip_first_tx_columns(&rm, j.E, sg.Zc), hc.ncore + 1);
          if (need < hc.ncols) {
            if (!(cut_ce && cut_need == need)) {
              cut_ce = get_code_cols(t.BG, (int)sg.Zc, need);
              cut_need = need;
            }
            if (!(ce_seg = cut_ce))
              return -1;
          }
        }
        ldpc_enc_job e;
        e.code = ce_seg->dev; e.in_off = j.c_off; e.out_off = j.d_off; e.Kb = (int32_t)sg.Kb; e.pad = 0;
        ej.push_back(e);
      }
    }
    const size_t n_seg = sj.size();
    const size_t o_tb = 0, o_seg = align_up(tbj.size() * sizeof(tb_tx_tb_job), 16),
                 o_enc = o_seg + align_up(n_seg * sizeof(tb_tx_seg_job), 16),
                 o_chk = o_enc + align_up(n_seg * sizeof(ldpc_enc_job), 16),
                 o_acc = o_chk + align_up(cj.size() * sizeof(tb_crc_chunk_job), 16),
                 /* CRC accumulators, uploaded as zeros: one per block (unfused path, atomics) / one per chunk (fused) */
                 jobs_bytes = o_acc + align_up(std::max<size_t>(ntb, cj.size()) * sizeof(uint32_t), 16);
    if (tb_wait_upload(c) != 0 || c.jobs_h.ensure(jobs_bytes) != 0 || pl.jobs_d.ensure(jobs_bytes) != 0)
      return -1;
    memset(c.jobs_h.p + o_acc, 0, jobs_bytes - o_acc);
    memcpy(c.jobs_h.p + o_chk, cj.data(), cj.size() * sizeof(tb_crc_chunk_job));
    memcpy(c.jobs_h.p + o_tb, tbj.data(), tbj.size() * sizeof(tb_tx_tb_job));
    memcpy(c.jobs_h.p + o_seg, sj.data(), n_seg * sizeof(tb_tx_seg_job));
    memcpy(c.jobs_h.p + o_enc, ej.data(), n_seg * sizeof(ldpc_enc_job));
    if (tb_upload_jobs(c, pl.jobs_d.p, jobs_bytes, s) != 0)
      return -1;
    pl.n_seg = n_seg; pl.n_aux = cj.size(); pl.scratch_top = ar.top;
    pl.ext[0] = ex.pay_lo; pl.ext[1] = ex.pay_hi; pl.ext[2] = ex.cod_lo; pl.ext[3] = ex.cod_hi;
    pl.out_dense = ex.cod_sum == ex.cod_hi - ex.cod_lo;
    {
      std::vector<uint64_t> off(ntb);
      std::vector<size_t> len(ntb);
      for (uint32_t i = 0; i < ntb; i++) {
        off[i] = tbs[i].coded_off;
        len[i] = tbs[i].G;
      }
      pl.build_out_runs(off.data(), len.data(), ntb);
    }
    pl.off[0] = o_tb; pl.off[1] = o_seg; pl.off[2] = o_enc; pl.off[3] = o_chk; pl.off[4] = o_acc;
    pl.threads[0] = enc_threads; pl.lds[0] = enc_lds;
    pl.remember(tbs, ntb, salt);
  }
  if (c.scratch.ensure(pl.scratch_top + 16) != 0) /* (+16: the fused kernel reads whole dwords around a segment's bytes) */
    return -1;
  const size_t n_seg = pl.n_seg;
  const size_t o_tb = pl.off[0], o_seg = pl.off[1], o_enc = pl.off[2], o_chk = pl.off[3], o_acc = pl.off[4];
  const int enc_threads = pl.threads[0], enc_lds = pl.lds[0];
  const uint8_t *payload = b->payload;
  uint8_t *coded = static_cast<uint8_t *>(b->coded);
  if (staged) {
    const size_t pay_lo = pl.ext[0], pay_n = pl.ext[1] - pl.ext[0], cod_lo = pl.ext[2], cod_n = pl.ext[3] - pl.ext[2];
    if (c.io_payload.ensure(pay_n) != 0 || c.io_coded.ensure(cod_n) != 0)
      return -1;
    const uint8_t *pay_src = b->payload + pay_lo;
    if (!(b->mem & NRLDPC_HIP_MEM_DEVICE) && needs_bounce(pay_src, pay_n)) { /* pageable: bounced (TbCtx::fin_copies) */
      if (c.payload_h.ensure(pay_n) != 0)
        return -1;
      bounce_copy(c.payload_h.p, pay_n, pay_src, pay_n, pay_n, 1);
      pay_src = c.payload_h.p;
    }
    HIP_TRY(hipMemcpyAsync(c.io_payload.p, pay_src, pay_n, hipMemcpyDefault, s));
    payload = c.io_payload.p - pay_lo;
    coded = c.io_coded.p - cod_lo;
  }
  const tb_tx_tb_job *d_tb = reinterpret_cast<const tb_tx_tb_job *>(pl.jobs_d.p + o_tb);
  const tb_tx_seg_job *d_seg = reinterpret_cast<const tb_tx_seg_job *>(pl.jobs_d.p + o_seg);
  uint32_t *d_acc = reinterpret_cast<uint32_t *>(pl.jobs_d.p + o_acc);
  TB_DEBUG_STAGE("tx: entry + copies in");
  HIP_TRY(tb_launch_tx_crc(d_tb, ntb, reinterpret_cast<const tb_crc_chunk_job *>(pl.jobs_d.p + o_chk), (uint32_t)pl.n_aux,
                           payload, c.scratch.p, d_acc, G().crc_pow_24a_long, G().crc_pow[NR_HIP_CRC16], fused ? 0 : 1, s));
  const ldpc_enc_job *d_enc = reinterpret_cast<const ldpc_enc_job *>(pl.jobs_d.p + o_enc);
  if (fused) {
    /* workgroup size: 256 threads let every CU hold eight segments (a whole 1664-segment slot is resident at once);
     * a launch that does not even fill the GPU four deep takes 512 and halves the rounds of its long stages */
    const int fused_threads = n_seg <= (size_t)4 * (size_t)G().n_cus ? 512 : enc_threads;
    TB_DEBUG_STAGE("tx: TB CRC launch");
    HIP_TRY(tb_launch_tx_fused(d_seg, d_enc, (uint32_t)n_seg, fused_threads, enc_lds + TB_TX_FUSED_EXTRA_LDS, payload, coded,
                               G().crc_pow[NR_HIP_CRC24_B], d_acc, s));
    TB_DEBUG_STAGE("tx: fused segment launch");
  } else {
    HIP_TRY(tb_launch_tx_segment(d_seg, (uint32_t)n_seg, c.scratch.p, G().crc_pow[NR_HIP_CRC24_B], s));
    ldpc_enc_args ea;
    memset(&ea, 0, sizeof(ea));
    ea.in = c.scratch.p;
    ea.out = c.scratch.p;
    ea.jobs = d_enc;
    HIP_TRY(ldpc_launch_enc_jobs(ea, enc_threads, enc_lds, (uint32_t)n_seg, s));
    HIP_TRY(tb_launch_tx_ratematch(d_seg, (uint32_t)n_seg, c.scratch.p, coded, s));
    TB_DEBUG_STAGE("tx: segment + rate matching launches");
  }
  if (staged) {
    uint8_t *hc = static_cast<uint8_t *>(b->coded);
    const size_t cod_lo = pl.ext[2], cod_n = pl.ext[3] - pl.ext[2];
    uint8_t *out = hc; /* where the copies go: the caller's array, or (pageable array) this thread's page-locked mirror of it */
    if (!(b->mem & NRLDPC_HIP_MEM_DEVICE) && needs_bounce(hc + cod_lo, cod_n)) {
      if (c.coded_h.ensure(cod_n) != 0)
        return -1;
      out = c.coded_h.p - cod_lo;
      if (pl.out_dense) {
        c.fin_copies.push_back(TbCtx::FinCopy{hc + cod_lo, c.coded_h.p, cod_n, 1, cod_n, cod_n, nullptr}); /* (one contiguous block: cut over the helpers) */
      } else {
        for (const TbPlan::OutRun &r : pl.out_runs)
          c.fin_copies.push_back(TbCtx::FinCopy{hc + r.first, out + r.first, r.width, r.rows, r.pitch, r.pitch, nullptr});
      }
    }
    if (pl.out_dense) {
      HIP_TRY(hipMemcpyAsync(out + cod_lo, c.io_coded.p, cod_n, hipMemcpyDefault, s));
    } else {
      for (const TbPlan::OutRun &r : pl.out_runs)
        HIP_TRY(hipMemcpy2DAsync(out + r.first, r.pitch, c.io_coded.p + (r.first - cod_lo), r.pitch, r.width, r.rows, hipMemcpyDefault, s));
    }
  }
  return 0;
}

int tb_tx_finish(const nrLDPC_hip_tb_batch_t *b, uint32_t ntb)
{
  if ((b->mem & NRLDPC_HIP_MEM_DEVICE) || ntb == 0)
    return 0;
  TbCtx &c = tls_tb;
  HIP_TRY(hipStreamSynchronize(c.own));
  c.finish_copies(true);
  return 0;
}

/* ---- soft buffers kept by the library (NRLDPC_HIP_MEM_HARQ_LIBRARY) -------------------------------------------------------
 * id -> {logical device, device pointer, int16 count}.  Buffers that are released go to a per-device pool and are handed out
 * again (hipFree would wait for every stream of the device, the resident server kernels' included).  `gen` changes whenever
 * an id's address does: the plans that hold absolute addresses are keyed on it. */
struct HarqEntry { int dev; int16_t *p; size_t n; };
struct HarqTable {
  std::mutex mu;
  std::map<uint64_t, HarqEntry> m;
  std::vector<HarqEntry> pool;
  std::atomic<uint64_t> gen{1};
} harq_tbl;

/* the soft buffers of block `id` on the current device, n int16; `fresh`: the call clears them before use (first
 * transmission).  A block that arrives with round > 0 and no history combines with zeros; one whose buffers live on another
 * GPU (the partition moved it) takes its history along.  Stream-ordered on s. */
int16_t *harq_lookup(uint64_t id, size_t n, bool fresh, hipStream_t s)
{
  const int di = cur_dev_index();
  std::lock_guard<std::mutex> lk(harq_tbl.mu);
  auto it = harq_tbl.m.find(id);
  if (it != harq_tbl.m.end() && it->second.dev == di && it->second.n >= n)
    return it->second.p;
  HarqEntry e{di, nullptr, n};
  for (size_t k = 0; k < harq_tbl.pool.size(); k++)
    if (harq_tbl.pool[k].dev == di && harq_tbl.pool[k].n >= n && harq_tbl.pool[k].n <= 2 * n) {
      e = harq_tbl.pool[k];
      harq_tbl.pool.erase(harq_tbl.pool.begin() + (long)k);
      break;
    }
  if (!e.p && hipMalloc(reinterpret_cast<void **>(&e.p), n * sizeof(int16_t)) != hipSuccess) {
    set_error("soft buffer allocation");
    return nullptr;
  }
  hipError_t err = hipSuccess;
  if (it != harq_tbl.m.end()) {
    if (!fresh) {
      const size_t keep = std::min(n, it->second.n);
      err = hipMemcpyAsync(e.p, it->second.p, keep * sizeof(int16_t), hipMemcpyDefault, s);
      if (err == hipSuccess && keep < n)
        err = hipMemsetAsync(e.p + keep, 0, (n - keep) * sizeof(int16_t), s);
      if (err == hipSuccess)
        err = hipStreamSynchronize(s); /* the old buffers go back to the pool below */
    }
    harq_tbl.pool.push_back(it->second);
  } else if (!fresh) {
    err = hipMemsetAsync(e.p, 0, n * sizeof(int16_t), s);
  }
  if (err != hipSuccess) {
    harq_tbl.pool.push_back(e);
    set_error("soft buffer set-up", err);
    return nullptr;
  }
  harq_tbl.m[id] = e;
  harq_tbl.gen.fetch_add(1);
  return e.p;
}

/* ---- RX: transport blocks [tb0, tb0+ntb) of b on the current device ----------------------------------------------------
 * staged = the caller's buffers are not this device's memory (host buffers; or, for a device-resident batch cut over
 * several GPUs, the owning GPU's memory): the device works on copies of exactly the ranges its blocks touch.  Two things are
 * never staged: LLRs in page-locked host memory (the segments' workgroups read them over the link in place) and soft
 * buffers that already live here (NRLDPC_HIP_MEM_HARQ_DEVICE on this GPU, NRLDPC_HIP_MEM_HARQ_LIBRARY). */
/* host-buffer calls that come in several pieces (tb_rx_enqueue_host): where this piece's LLRs already are, and where its
 * verdicts go in the call's page-locked result area */
struct RxHostStage {
  const int16_t *llr_dev; /* device copy of the caller's LLR array, biased like it (index = coded_off); NULL: none yet */
  bool no_pull;           /* do not read page-locked LLRs in place: a copy is (being) made */
  uint32_t call_tb0, call_ntb;
  size_t pay_lo, pay_hi;  /* payload bytes the whole call's transport blocks touch */
};

int tb_rx_enqueue(const nrLDPC_hip_tb_batch_t *b, uint32_t tb0, uint32_t ntb, bool staged, hipStream_t s_direct,
                  const RxHostStage *st = nullptr)
{
  hipStream_t s;
  if (tb_begin(s, s_direct, staged) != 0)
    return -1;
  if (ntb == 0)
    return 0;
  nrLDPC_hip_tb_t *tbs = b->tb + tb0;
  TbCtx &c = tls_tb;
  const bool harq_lib = (b->mem & NRLDPC_HIP_MEM_HARQ_LIBRARY) != 0;
  /* soft buffers usable in place: the caller's device memory on THIS GPU */
  bool harq_here = false;
  if (!harq_lib && (!staged || (b->mem & (NRLDPC_HIP_MEM_HARQ_DEVICE | NRLDPC_HIP_MEM_DEVICE)))) {
    hipPointerAttribute_t at;
    bool harq_is_dev = false;
    /* (hipMalloc memory, or hipMallocManaged memory -- the kernels and the peer copies reach both: ADVICE r05) */
    if (hipPointerGetAttributes(&at, b->harq) == hipSuccess && (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged)) {
      harq_is_dev = true;
      harq_here = at.device == G().id;
    } else {
      (void)hipGetLastError();
    }
    if (!staged && !harq_here)
      return set_error("soft buffers must be memory of the GPU that holds the LLRs");
    /* whatever the number of GPUs the call is cut over: a host pointer under this flag would be handed to peer copies */
    if ((b->mem & NRLDPC_HIP_MEM_HARQ_DEVICE) && !harq_is_dev)
      return set_error("NRLDPC_HIP_MEM_HARQ_DEVICE: harq is not device memory");
    /* test hook: logical devices that alias one GPU (the GPU box has one) treat the owner's soft buffers as a peer's */
    static const bool stage_env = [] { const char *e = getenv("NRLDPC_HIP_TEST_STAGE_HARQ"); return e && atoi(e) != 0; }();
    if (staged && stage_env)
      harq_here = false;
    if ((b->mem & NRLDPC_HIP_MEM_HARQ_DEVICE) && !(b->mem & NRLDPC_HIP_MEM_DEVICE) && !harq_here && g.n_shard == 1)
      return set_error("NRLDPC_HIP_MEM_HARQ_DEVICE: harq is not device memory of the library's GPU");
  }
  const bool harq_staged = !harq_lib && !harq_here;
  const int fused_mode = tb_fused_mode();
  uint64_t salt[3] = {(uint64_t)b->harq_stride | ((uint64_t)(harq_lib ? 2 : (harq_staged ? 1 : 0)) << 32),
                            (uint64_t)(fused_mode & 0xff) | ((uint64_t)(tb_multi_mode() & 0xff) << 8) | ((uint64_t)tb_classes_enabled() << 16) |
                                ((uint64_t)(tb_fill_mode() & 0xff) << 24) | ((uint64_t)tb_trunc_enabled() << 32) | ((uint64_t)tb_lrow_enabled() << 33),
                            harq_lib ? harq_tbl.gen.load() : 0};
  TbPlan *hit = c.rx.find(tbs, ntb, salt);
  if (hit) {
    for (uint32_t i = 0; i < ntb; i++) /* nr_get_R_ldpc_decoder's state leaves the call as it did the first time */
      tbs[i].llrLen = hit->llr_len[i];
  }
  TbPlan &pl = hit ? *hit : c.rx.victim();
  if (!hit) {
    std::vector<uint8_t> key_tb((const uint8_t *)tbs, (const uint8_t *)tbs + (size_t)ntb * sizeof(nrLDPC_hip_tb_t));
    std::vector<tb_rx_tb_job> tbj(ntb);
    std::vector<tb_rx_seg_job> sj, sj_legacy;
    struct ShapedJob { ldpc_dec_job dj; int kind, threads, lds; double cost; int num_llr; int zc; };
    std::vector<ShapedJob> single; /* segments that get a workgroup of their own */
    Arena ar;
    TbExtent ex;
    std::vector<TbPlan::HarqRun> runs;
    uint32_t rx_lds_elems = 8;
    /* decoder workgroup shape (ldpc_graph.h): a batch that does not even give every CU one segment wants the latency shape */
    uint32_t n_seg_total = 0;
    for (uint32_t i = 0; i < ntb; i++) {
      nr_hip_seg_t sg;
      if (tbs[i].A && (tbs[i].BG == 1 || tbs[i].BG == 2) &&
          nr_hip_segmentation((uint32_t)nr_hip_len_with_crc(1, (int)tbs[i].A), tbs[i].BG, &sg) == 0)
        n_seg_total += sg.C;
    }
    const bool lat_shape = n_seg_total <= (uint32_t)G().n_cus;
    const int multi_mode = tb_multi_mode();
    const bool multi_ok = multi_mode != 0 && n_seg_total >= 2u * (uint32_t)G().n_cus;
    const bool classes = tb_classes_enabled() && !lat_shape;
    auto add_single = [&](const CodeEntry *ce, const ldpc_dec_job &dj, uint32_t fused_lds) {
      const ldpc_code_desc_t &hc = ce->host, &shape = lat_shape ? ce->host_lat : ce->host;
      const double cost = (double)hc.num_llr * dj.num_max_iter;
      if (hc.f_ok)
        single.push_back(ShapedJob{dj, 0, shape.f_n_threads, std::max(shape.f_lds_total, (int)fused_lds), cost, hc.num_llr,
                                   (shape.f_mb == 1 && shape.f_rstride == shape.Z + 4 && shape.f_astride == 2 * shape.Z) ? shape.Z : 0});
      else
        single.push_back(ShapedJob{dj, 1, hc.n_threads, hc.lds_total, cost, hc.num_llr, 0});
    };
    struct MultiCand { const CodeEntry *ce; ldpc_dec_job dj; };
    std::vector<MultiCand> cands;
    std::vector<ldpc_dec_job> mjobs[2];
    std::vector<ldpc_dec_mgroup> mgrp[2];
    int m_threads[2] = {64, 64}, m_lds[2] = {0, 0};
    size_t n_legacy_tb = 0;
    bool any_fused = false;
    for (uint32_t i = 0; i < ntb; i++) {
      nrLDPC_hip_tb_t &t = tbs[i];
      if (tb_validate(t) != 0)
        return -1;
      /* nr_ulsch_decoding.c:386-395: segmentation parameters from lenWithCrc(1, A) */
      const uint32_t B = (uint32_t)nr_hip_len_with_crc(1, (int)t.A);
      nr_hip_seg_t sg;
      if (nr_hip_segmentation(B, t.BG, &sg) != 0)
        return set_error("nr_segmentation: unsupported block size");
      const CodeEntry *full = get_code(t.BG, (int)sg.Zc, t.BG == 1 ? 13 : 15);
      if (!full)
        return -1;
      /* the block's soft buffers: the caller's array (offset in int16), or the library's (absolute address / 2: the kernels
       * then get a null base) */
      uint64_t harq_base = t.harq_off;
      if (harq_lib) {
        const int16_t *hp = harq_lookup(t.harq_off, (size_t)sg.C * b->harq_stride, t.round == 0, s);
        if (!hp)
          return -1;
        harq_base = (uint64_t)(reinterpret_cast<uintptr_t>(hp) / sizeof(int16_t));
      }
      /* every segment of the block through the fused segment kernel?  (the fast decoder serves the code, and the block's
       * segments are not candidates for shared workgroups) */
      bool fused_tb = fused_mode != 0 && full->host.f_ok && !(multi_ok && full->dev_multi);
      const uint32_t cstride = (uint32_t)align_up(out_bytes_of(full->host, 0), 16);
      tb_rx_tb_job &tj = tbj[i];
      memset(&tj, 0, sizeof(tj));
      tj.payload_off = t.payload_off;
      if (!fused_tb) {
        tj.b_off = ar.take(B / 8 + 4);
        tj.c_off0 = ar.take((size_t)cstride * sg.C);
      }
      tj.c_stride = cstride;
      tj.seg0 = (uint32_t)sj.size();
      tj.C = sg.C;
      tj.A = t.A;
      tj.B = B;
      tj.crc_type = (uint32_t)nr_hip_crc_type(1, (int)t.A);
      tj.num_max_iter = t.numMaxIter;
      tj.seg_bytes = sg.K / 8 - sg.F / 8 - (sg.C > 1 ? 3 : 0); /* phy_procedures_nr_gNB.c:287 */
      tj.fused = fused_tb ? 1u : 0u;
      any_fused |= fused_tb;
      n_legacy_tb += fused_tb ? 0 : 1;
      ex.add(ex.pay_lo, ex.pay_hi, (size_t)t.payload_off, (size_t)t.payload_off + t.A / 8);
      ex.add(ex.cod_lo, ex.cod_hi, (size_t)t.coded_off, (size_t)t.coded_off + t.G);
      if (!harq_lib)
        ex.add(ex.harq_lo, ex.harq_hi, (size_t)t.harq_off, (size_t)t.harq_off + (size_t)sg.C * b->harq_stride);
      ex.pay_sum += t.A / 8;
      uint32_t r_offset = 0;
      int llrLen = t.llrLen;
      const CodeEntry *cut_ce = nullptr;
      int cut_need = 0;
      for (uint32_t r = 0; r < sg.C; r++) {
        const uint32_t E = nr_hip_get_E(t.G, sg.C, t.Qm, t.Nl, r);
        /* nr_ulsch_decoding.c:439-444: decoder rate mode, stateful in llrLen */
        const int R = nr_hip_get_R_ldpc_decoder(t.rv, (int)E, t.BG, (int)sg.Zc, &llrLen, t.round);
        const CodeEntry *ce = get_code(t.BG, (int)sg.Zc, R);
        if (!ce)
          return -1;
        nr_hip_rm_t rm;
        if (nr_hip_rate_match_geometry(t.tbslbrm, t.BG, sg.Zc, sg.C, sg.F, sg.K, t.rv, E, &rm) != 0)
          return set_error("nr_rate_matching_rx: invalid parameters");
        if (tb_trunc_enabled() && t.round == 0) {
          /* A first transmission leaves every soft-buffer position behind the last one it reaches at 0 (the buffer is cleared,
           * nr_ulsch_decoding.c:418-422), and with them whole degree-1 columns at the end of the rate mode's graph -- at MCS 27 eight
           * of the nine extension columns of BG1's R = 2/3 mode.  A row that closes on such a column sends zeros to the core columns
           * in every pass (ldpc_graph.h LDPC_R_COLS) and the chain stops on the CRC: the segment is decoded on the rate mode's graph
           * cut behind its last column that received anything -- the same payload, verdict and pass count from 87 edges instead of 144. */
          const int need = std::max((int)nr_hip_first_tx_columns(&rm, E, sg.Zc), ce->host.ncore + 1);
          if (need < ce->host.ncols) {
            if (!(cut_ce && cut_need == need)) { /* (the segments of a block agree but for one symbol: one locked look-up per block) */
              cut_ce = get_code_cols(t.BG, (int)sg.Zc, need);
              cut_need = need;
            }
            ce = cut_ce;
            if (!ce)
              return -1;
          }
        }
        const ldpc_code_desc_t &hc = ce->host;
        tb_rx_seg_job j;
        memset(&j, 0, sizeof(j));
        j.llr_off = t.coded_off + r_offset;
        j.harq_off = harq_base + (uint64_t)r * b->harq_stride;
        j.l_off = ar.take(hc.num_llr);
        j.E = E; j.Qm = t.Qm; j.Ncb = rm.Ncb; j.Foffset = rm.Foffset; j.Fin = rm.Fin; j.V = rm.V; j.rank0 = rm.rank0;
        j.clear = t.round == 0; /* harq_to_be_cleared -> d_to_be_cleared[r] (nr_ulsch_decoding.c:418-422) */
        j.K = sg.K; j.F = sg.F; j.Z = sg.Zc; j.num_llr = (uint32_t)hc.num_llr;
        j.c_off = tj.c_off0 + (uint64_t)r * cstride;
        const uint32_t lds_elems = tb_rx_lds_elems(E, rm.Fin, rm.Ncb);
        if (!fused_tb)
          rx_lds_elems = std::max(rx_lds_elems, lds_elems);
        j.tb = i; j.r = r; j.iter_idx = (uint32_t)sj.size();
        if (harq_staged) { /* rows of the caller's array that travel: what the kernels look at of this segment's row */
          const uint32_t np = (uint32_t)hc.num_llr > 2 * sg.Zc ? (uint32_t)hc.num_llr - 2 * sg.Zc : 0u;
          const uint32_t width = std::min<uint32_t>(std::max(rm.Ncb, np), b->harq_stride);
          const size_t first = (size_t)t.harq_off + (size_t)r * b->harq_stride;
          if (!runs.empty() && runs.back().width == width && runs.back().upload == (t.round != 0) &&
              runs.back().first + (size_t)runs.back().rows * b->harq_stride == first)
            runs.back().rows++;
          else
            runs.push_back(TbPlan::HarqRun{first, 1u, width, t.round != 0});
        }
        ldpc_dec_job dj;
        dj.code = (hc.f_ok && lat_shape) ? ce->dev_lat : ce->dev;
        dj.llr_off = j.l_off;
        dj.out_off = tj.c_off0 + (uint64_t)r * cstride;
        dj.num_max_iter = t.numMaxIter;
        dj.E = nr_hip_len_with_crc((int)sg.C, (int)t.A); /* nr_ulsch_decoding.c:190 */
        dj.crc_type = nr_hip_crc_type((int)sg.C, (int)t.A);
        /* a retransmission is decoded on the whole rate mode (what earlier rounds left in the soft buffer is the caller's
         * business) -- but the mode it lands in is usually a larger one, with columns between and behind the transmissions that
         * nothing has reached: their rows' items are looked for in the kernel and not run (ldpc_dec_fast_block.h; -6 % on the
         * rv 2 retransmission of the MCS 27 slot).  Not for first transmissions: the cut graph has taken the rows already, and the
         * look costs two dependent loads per extension item (+1.7 % on that slot). */
        if (t.round != 0 && tb_trunc_enabled())
          dj.crc_type |= LDPC_JOB_MUTE_CHECK;
        dj.iter_idx = (int32_t)sj.size();
        dj.abort_idx = tb_abort_enabled() ? (int32_t)i : -1;
        dj.seg_idx = fused_tb ? (int32_t)sj.size() : -1;
        if (dj.E > hc.kb_full * hc.Z || (dj.E & 7))
          return set_error("CRC length outside the code block");
        if (fused_tb && !hc.f_ok)
          return set_error("internal: fused block with a segment outside the fast decoder");
        if (!fused_tb && multi_ok && ce->dev_multi) /* candidates for a shared workgroup; sorted into groups below */
          cands.push_back(MultiCand{ce, dj});
        else
          add_single(ce, dj, fused_tb ? lds_elems * (uint32_t)sizeof(int16_t) : 0u);
        sj.push_back(j);
        if (!fused_tb)
          sj_legacy.push_back(j);
        r_offset += E;
      }
      t.llrLen = llrLen;
    }
    /* the candidates, grouped by (code, iteration cap, CRC length and type): a group of at least four shares workgroups of
     * f_mb * f_sub segments each; smaller groups go back to one workgroup per segment */
    if (multi_mode != 2 && cands.size() < (size_t)TB_MULTI_MIN_PER_CU * (size_t)G().n_cus) { /* they all fit side by side as they are */
      for (const MultiCand &q : cands)
        add_single(q.ce, q.dj, 0u);
      cands.clear();
    }
    std::stable_sort(cands.begin(), cands.end(), [](const MultiCand &x, const MultiCand &y) {
      if (x.ce != y.ce) return x.ce < y.ce;
      if (x.dj.num_max_iter != y.dj.num_max_iter) return x.dj.num_max_iter < y.dj.num_max_iter;
      if (x.dj.E != y.dj.E) return x.dj.E < y.dj.E;
      return (x.dj.crc_type & 3) < (y.dj.crc_type & 3);
    });
    for (size_t i0 = 0; i0 < cands.size();) {
      size_t i1 = i0;
      while (i1 < cands.size() && cands[i1].ce == cands[i0].ce && cands[i1].dj.num_max_iter == cands[i0].dj.num_max_iter &&
             cands[i1].dj.E == cands[i0].dj.E && (cands[i1].dj.crc_type & 3) == (cands[i0].dj.crc_type & 3))
        i1++;
      const CodeEntry *ce = cands[i0].ce;
      const ldpc_code_desc_t &hm = ce->host_multi;
      if (i1 - i0 >= 4) {
        const int cls = hm.f_sub == 4 ? 1 : 0;
        const size_t per_wg = (size_t)hm.f_mb * (size_t)hm.f_sub;
        for (size_t k = i0; k < i1; k += per_wg) {
          ldpc_dec_mgroup gq;
          memset(&gq, 0, sizeof(gq));
          gq.code = ce->dev_multi;
          gq.first_job = (uint32_t)mjobs[cls].size();
          gq.n_valid = (uint32_t)std::min(per_wg, i1 - k);
          gq.num_max_iter = cands[i0].dj.num_max_iter; gq.E = cands[i0].dj.E; gq.crc_type = cands[i0].dj.crc_type & 3;
          for (size_t q = k; q < k + gq.n_valid; q++)
            mjobs[cls].push_back(cands[q].dj);
          mgrp[cls].push_back(gq);
        }
        m_threads[cls] = std::max(m_threads[cls], hm.f_n_threads);
        m_lds[cls] = std::max(m_lds[cls], hm.f_lds_total);
      } else {
        for (size_t q = i0; q < i1; q++)
          add_single(ce, cands[q].dj, 0u);
      }
      i0 = i1;
    }
    /* Decoder launches.  A job-array launch has ONE workgroup shape (threads, LDS = the maxima over its jobs), so the jobs
     * are sorted by kernel and by how many workgroups of their shape a CU holds (1, 2, 4, 8, 16+; the heaviest first inside
     * a class) and cut into launches class by class -- and a launch whose last workgroup round would be partly empty takes
     * jobs of the following, smaller classes along until that round is full: they run in slots that would otherwise idle
     * (a smaller job fits any larger shape), and the smaller classes' own launches get shorter.  Round 3 gave every class
     * its own launch: 9 workgroup rounds for the 93 small TBS x 64 where 4.3 rounds of work exist.
     * NRLDPC_HIP_TB_FILL=0: no filling. */
    const int fill_env = tb_fill_mode();
    auto per_cu_of = [&](int kind, int threads, int lds) {
      const int waves = kind == 0 ? 16 : 32; /* the fast kernel's 128 VGPRs allow 16 waves per CU */
      return std::max(1, std::min(waves * 64 / std::max(threads, 64), (160 * 1024) / std::max(lds, 1024)));
    };
    auto wg_class = [&](const ShapedJob &j) {
      if (!classes)
        return 0;
      const int per_cu = per_cu_of(j.kind, j.threads, j.lds);
      return per_cu >= 16 ? 4 : per_cu >= 8 ? 3 : per_cu >= 4 ? 2 : per_cu >= 2 ? 1 : 0;
    };
    std::stable_sort(single.begin(), single.end(), [&](const ShapedJob &x, const ShapedJob &y) {
      const int cx = x.kind * 8 + wg_class(x), cy = y.kind * 8 + wg_class(y);
      return cx != cy ? cx < cy : (classes && x.cost > y.cost);
    });
    std::vector<ldpc_dec_job> single_jobs(single.size());
    for (size_t q = 0; q < single.size(); q++)
      single_jobs[q] = single[q].dj;
    std::vector<TbPlan::DecLaunch> dec;
    for (size_t q = 0; q < single.size();) {
      const int kind = single[q].kind, cls = wg_class(single[q]);
      size_t e = q;
      int threads = 64, lds = 0;
      while (e < single.size() && single[e].kind == kind && wg_class(single[e]) == cls) {
        threads = std::max(threads, single[e].threads);
        lds = std::max(lds, single[e].lds);
        e++;
      }
      if (classes && fill_env) {
        const size_t slots = (size_t)G().n_cus * (size_t)per_cu_of(kind, threads, lds);
        const size_t rem = (e - q) % slots;
        size_t room = rem ? slots - rem : 0;
        /* (jobs of the classes behind: they fit when neither their threads nor their LDS exceed the launch's) */
        while (room && e < single.size() && single[e].kind == kind && single[e].threads <= threads && single[e].lds <= lds) {
          e++;
          room--;
        }
      }
      TbPlan::DecLaunch dl{kind, q * sizeof(ldpc_dec_job), 0, (uint32_t)(e - q), threads, lds, false};
      dl.all_fused = true;
      dl.zc = single[q].zc;
      for (size_t i = q; i < e; i++) {
        if (single[i].zc != dl.zc)
          dl.zc = 0;
        dl.fused |= single[i].dj.seg_idx >= 0;
        dl.all_fused &= single[i].dj.seg_idx >= 0;
        dl.mute |= (single[i].dj.crc_type & LDPC_JOB_MUTE_CHECK) != 0;
        dl.max_llr = std::max(dl.max_llr, single[i].num_llr);
      }
      dec.push_back(dl);
      q = e;
    }
    const size_t n_seg = sj.size();
    const size_t o_tb = 0, o_seg = align_up(tbj.size() * sizeof(tb_rx_tb_job), 16),
                 o_leg = o_seg + align_up(n_seg * sizeof(tb_rx_seg_job), 16),
                 o_single = o_leg + align_up(sj_legacy.size() * sizeof(tb_rx_seg_job), 16),
                 o_mj0 = o_single + align_up(single_jobs.size() * sizeof(ldpc_dec_job), 16),
                 o_mg0 = o_mj0 + align_up(mjobs[0].size() * sizeof(ldpc_dec_job), 16),
                 o_mj1 = o_mg0 + align_up(mgrp[0].size() * sizeof(ldpc_dec_mgroup), 16),
                 o_mg1 = o_mj1 + align_up(mjobs[1].size() * sizeof(ldpc_dec_job), 16),
                 o_acc = o_mg1 + align_up(mgrp[1].size() * sizeof(ldpc_dec_mgroup), 16), /* per TB: CRC accumulators, abort
                                                                                            flags, finished-segment counters:
                                                                                            uploaded as zeros, left zero */
                 o_slots = o_acc + align_up((size_t)ntb * 4 * sizeof(uint32_t), 16), /* (+ per TB: generation) */
                 jobs_bytes = o_slots + align_up(n_seg * sizeof(uint64_t), 16), /* per segment: the fused kernel's slots */
                 o_iter = jobs_bytes; /* n_iter lives behind the uploaded part in the same device buffer */
    if (tb_wait_upload(c) != 0 || c.jobs_h.ensure(jobs_bytes) != 0 ||
        pl.jobs_d.ensure(o_iter + n_seg * sizeof(int32_t)) != 0)
      return -1;
    memset(c.jobs_h.p + o_acc, 0, jobs_bytes - o_acc);
    memcpy(c.jobs_h.p + o_tb, tbj.data(), tbj.size() * sizeof(tb_rx_tb_job));
    memcpy(c.jobs_h.p + o_seg, sj.data(), n_seg * sizeof(tb_rx_seg_job));
    memcpy(c.jobs_h.p + o_leg, sj_legacy.data(), sj_legacy.size() * sizeof(tb_rx_seg_job));
    memcpy(c.jobs_h.p + o_single, single_jobs.data(), single_jobs.size() * sizeof(ldpc_dec_job));
    memcpy(c.jobs_h.p + o_mj0, mjobs[0].data(), mjobs[0].size() * sizeof(ldpc_dec_job));
    memcpy(c.jobs_h.p + o_mg0, mgrp[0].data(), mgrp[0].size() * sizeof(ldpc_dec_mgroup));
    memcpy(c.jobs_h.p + o_mj1, mjobs[1].data(), mjobs[1].size() * sizeof(ldpc_dec_job));
    memcpy(c.jobs_h.p + o_mg1, mgrp[1].data(), mgrp[1].size() * sizeof(ldpc_dec_mgroup));
    if (tb_upload_jobs(c, pl.jobs_d.p, jobs_bytes, s) != 0)
      return -1;
    /* a launch too short to be worth its own tail (one segment's latency at the end of every launch) joins the launch of
     * the next larger shape before it: the jobs are consecutive */
    for (size_t k = 1; k < dec.size();) {
      if (dec[k].kind == dec[k - 1].kind && dec[k].n < (uint32_t)G().n_cus / 2) {
        dec[k - 1].n += dec[k].n;
        dec[k - 1].threads = std::max(dec[k - 1].threads, dec[k].threads);
        dec[k - 1].lds = std::max(dec[k - 1].lds, dec[k].lds);
        dec[k - 1].fused |= dec[k].fused;
        dec[k - 1].all_fused &= dec[k].all_fused;
        dec[k - 1].mute |= dec[k].mute;
        if (dec[k - 1].zc != dec[k].zc)
          dec[k - 1].zc = 0;
        dec[k - 1].max_llr = std::max(dec[k - 1].max_llr, dec[k].max_llr);
        dec.erase(dec.begin() + (long)k);
      } else {
        k++;
      }
    }
    for (TbPlan::DecLaunch &dl : dec)
      dl.jobs_off += o_single;
    /* a launch of fused segments keeps the int8 decoder input in LDS when that costs the CU no workgroup (tb_chain.h lrow_off):
     * no row through memory, no wait for the soft-buffer stores in front of the decoder.  NRLDPC_HIP_TB_LROW=0: off */
    for (TbPlan::DecLaunch &dl : dec)
      if (tb_lrow_enabled() && dl.kind == 0 && dl.fused && dl.all_fused) {
        const int base = (int)align_up((size_t)dl.lds, 16), with_row = base + (int)align_up((size_t)dl.max_llr, 16);
        if (with_row <= 160 * 1024 && per_cu_of(0, dl.threads, with_row) == per_cu_of(0, dl.threads, dl.lds)) {
          dl.lrow = (uint32_t)base;
          dl.lds = with_row;
        }
      }
    if (!mgrp[0].empty())
      dec.push_back(TbPlan::DecLaunch{2, o_mj0, o_mg0, (uint32_t)mgrp[0].size(), m_threads[0], m_lds[0], false});
    if (!mgrp[1].empty())
      dec.push_back(TbPlan::DecLaunch{3, o_mj1, o_mg1, (uint32_t)mgrp[1].size(), m_threads[1], m_lds[1], false});
    pl.dec.swap(dec);
    pl.n_seg = n_seg; pl.scratch_top = ar.top;
    pl.ext[0] = ex.pay_lo; pl.ext[1] = ex.pay_hi; pl.ext[2] = ex.cod_lo; pl.ext[3] = ex.cod_hi;
    pl.ext[4] = harq_lib ? 0 : ex.harq_lo; pl.ext[5] = harq_lib ? 0 : ex.harq_hi;
    pl.out_dense = ex.pay_sum == ex.pay_hi - ex.pay_lo;
    {
      std::vector<uint64_t> off(ntb);
      std::vector<size_t> len(ntb);
      for (uint32_t i = 0; i < ntb; i++) {
        off[i] = tbs[i].payload_off;
        len[i] = tbs[i].A / 8;
      }
      pl.build_out_runs(off.data(), len.data(), ntb);
    }
    pl.harq_runs.swap(runs);
    pl.off[0] = o_tb; pl.off[1] = o_seg; pl.off[4] = o_iter; pl.off[5] = o_acc; pl.off[6] = o_slots;
    pl.legacy_off = o_leg; pl.n_legacy_seg = sj_legacy.size(); pl.n_legacy_tb = n_legacy_tb; pl.any_fused = any_fused;
    pl.rx_lds_elems = rx_lds_elems;
    pl.llr_len.resize(ntb);
    for (uint32_t i = 0; i < ntb; i++)
      pl.llr_len[i] = tbs[i].llrLen;
    /* the key is the descriptor array as it ARRIVED (llrLen is updated by the loop above); the library's soft buffers as
     * they are now, after this build's allocations */
    salt[2] = harq_lib ? harq_tbl.gen.load() : 0;
    pl.remember(key_tb.data(), ntb, salt);
  }
  if (c.scratch.ensure(pl.scratch_top) != 0)
    return -1;
  const size_t n_seg = pl.n_seg;
  const size_t o_tb = pl.off[0], o_seg = pl.off[1], o_iter = pl.off[4], o_acc = pl.off[5];
  int32_t *d_iter = reinterpret_cast<int32_t *>(pl.jobs_d.p + o_iter);
  uint8_t *payload = b->payload;
  const int16_t *llr = static_cast<const int16_t *>(b->coded);
  int16_t *harq = harq_lib ? nullptr : b->harq;
  uint8_t *ack = b->ack + tb0;
  int32_t *iter_max = b->iter_max + tb0;
  const size_t stride2 = (size_t)b->harq_stride * sizeof(int16_t);
  const bool to_host = staged && !(b->mem & NRLDPC_HIP_MEM_DEVICE);
  const bool payload_direct = false;
  bool harq_bounce = false;
  uint8_t *harq_mirror = nullptr;
  if (staged) {
    const size_t pay_lo = pl.ext[0], pay_n = pl.ext[1] - pl.ext[0], cod_lo = pl.ext[2], cod_n = pl.ext[3] - pl.ext[2];
    /* LLRs in page-locked host memory are read in place (the device address of the caller's array); anything else is copied */
    const int16_t *pulled = nullptr;
    if (st && st->llr_dev) {
      llr = st->llr_dev;
    } else {
      if (to_host && host_range_is_partly_pinned(static_cast<const int16_t *>(b->coded) + cod_lo, cod_n * sizeof(int16_t)))
        return set_error("coded: the array is page-locked in part only (register the whole range the call reads, or none of it)");
      if (to_host && !(st && st->no_pull) && tb_pull_mode() != 0 &&
          host_ptr_is_pinned(static_cast<const int16_t *>(b->coded) + cod_lo, cod_n * sizeof(int16_t))) {
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, b->coded, 0) == hipSuccess)
          pulled = static_cast<const int16_t *>(dp);
        else
          (void)hipGetLastError();
      }
      if (pulled) {
        llr = pulled;
      } else {
        if (c.io_coded.ensure(cod_n * 2) != 0)
          return -1;
        const void *cod_src = static_cast<const int16_t *>(b->coded) + cod_lo;
        if (to_host && needs_bounce(cod_src, cod_n * 2)) { /* pageable: bounced (TbCtx::fin_copies) */
          if (c.coded_h.ensure(cod_n * 2) != 0)
            return -1;
          bounce_copy(c.coded_h.p, cod_n * 2, cod_src, cod_n * 2, cod_n * 2, 1);
          cod_src = c.coded_h.p;
        }
        HIP_TRY(hipMemcpyAsync(c.io_coded.p, cod_src, cod_n * 2, hipMemcpyDefault, s));
        llr = reinterpret_cast<const int16_t *>(c.io_coded.p) - cod_lo;
      }
    }
    if (harq_staged) {
      const size_t harq_lo = pl.ext[4], harq_n = pl.ext[5] - pl.ext[4];
      if (c.io_harq.ensure(harq_n * 2) != 0)
        return -1;
      /* a pageable array of the caller's is mirrored in page-locked memory: rows that travel up are copied into the mirror
       * by the CPU first, rows that come back are handed over by tb_rx_finish (TbCtx::fin_copies) */
      /* ... and only a HOST array is: device soft buffers of another GPU (a call cut over several GPUs, this part not the
       * owner's) take the strided copies directly -- peer copies --, the CPU must never touch them */
      const bool harq_is_host = !(b->mem & (NRLDPC_HIP_MEM_HARQ_DEVICE | NRLDPC_HIP_MEM_DEVICE));
      harq_bounce = to_host && harq_is_host && needs_bounce(b->harq + harq_lo, harq_n * 2);
      if (harq_bounce) { /* this piece's share of the mirror (growing it parks the old area: earlier pieces keep theirs) */
        const size_t at = align_up(c.harq_h_used, 64);
        if (c.harq_h.ensure(at + harq_n * 2) != 0)
          return -1;
        harq_mirror = c.harq_h.p + at;
        c.harq_h_used = at + harq_n * 2;
      }
      const int16_t *harq_src = harq_bounce ? reinterpret_cast<const int16_t *>(harq_mirror) - harq_lo : b->harq;
      for (const TbPlan::HarqRun &r : pl.harq_runs)
        if (r.upload) {
          /* a bounced run goes up in pieces of ~8 MB: the CPU fills the mirror for piece k + 1 while the link takes piece k */
          const uint32_t step = harq_bounce ? std::max<uint32_t>(1, (uint32_t)(((size_t)8 << 20) / ((size_t)r.width * 2))) : r.rows;
          for (uint32_t q0 = 0; q0 < r.rows; q0 += step) {
            const uint32_t nq = std::min(step, r.rows - q0);
            const size_t first = r.first + (size_t)q0 * b->harq_stride;
            if (harq_bounce)
              bounce_copy(harq_mirror + (first - harq_lo) * 2, stride2, b->harq + first, stride2, (size_t)r.width * 2, nq);
            HIP_TRY(hipMemcpy2DAsync(c.io_harq.p + (first - harq_lo) * 2, stride2, harq_src + first, stride2, (size_t)r.width * 2, nq,
                                     hipMemcpyDefault, s));
          }
        }
      harq = reinterpret_cast<int16_t *>(c.io_harq.p) - harq_lo;
    }
    if (to_host) {
      /* verdicts: straight into this thread's page-locked result area (device-mapped; tb_rx_finish hands them over) -- no
       * copy engine for a few bytes; payload: into the caller's array when that is page-locked too, else via a device copy */
      const uint32_t call_tb0 = st ? st->call_tb0 : tb0, call_ntb = st ? st->call_ntb : ntb;
      if (c.small_h.ensure((size_t)call_ntb * 8 + 64) != 0)
        return -1;
      void *dp = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dp, c.small_h.p, 0));
      iter_max = static_cast<int32_t *>(dp) + (tb0 - call_tb0);
      ack = static_cast<uint8_t *>(dp) + (size_t)call_ntb * 4 + (tb0 - call_tb0);
      /* payload: into device memory, then one copy per piece into PAGE-LOCKED host memory -- the caller's array when it is
       * page-locked, else this thread's staging area, handed over by tb_rx_finish.  (A device -> host copy into pageable
       * memory makes the host wait for the piece before it can enqueue the next one; and letting the kernels store the
       * payload bytes over the link themselves was measured at 48 ms per slot: 4-byte writes, each one waited for.) */
      const size_t lo = st ? st->pay_lo : pay_lo, n = st ? st->pay_hi - st->pay_lo : pay_n;
      if (!needs_bounce(b->payload + lo, n)) {
        c.fin_pay_n = 0;
      } else {
        (void)hipGetLastError();
        if (c.payload_h.ensure(n) != 0)
          return -1;
        c.fin_pay_lo = lo;
        c.fin_pay_n = n;
      }
    } else {
      if (c.io_small.ensure((size_t)ntb * 8 + 64) != 0)
        return -1;
      iter_max = reinterpret_cast<int32_t *>(c.io_small.p);
      ack = c.io_small.p + (size_t)ntb * 4;
    }
    if (!payload_direct) {
      if (c.io_payload.ensure(pay_n) != 0)
        return -1;
      payload = c.io_payload.p - pay_lo;
    }
  }
  const tb_rx_seg_job *d_seg = reinterpret_cast<const tb_rx_seg_job *>(pl.jobs_d.p + o_seg);
  const tb_rx_seg_job *d_leg = reinterpret_cast<const tb_rx_seg_job *>(pl.jobs_d.p + pl.legacy_off);
  const tb_rx_tb_job *d_tb = reinterpret_cast<const tb_rx_tb_job *>(pl.jobs_d.p + o_tb);
  if (c.timing) {
    for (hipEvent_t &e : c.tev)
      if (!e)
        HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(c.tev[0], s));
  }
  TB_DEBUG_STAGE("entry + copies in (job upload, LLRs, soft buffers)");
  HIP_TRY(tb_launch_rx_dematch(d_leg, (uint32_t)pl.n_legacy_seg, pl.rx_lds_elems, llr, harq, reinterpret_cast<int8_t *>(c.scratch.p), s,
                               n_seg <= (size_t)G().n_cus));
  TB_DEBUG_STAGE("de-matching launch (segments outside the fused kernel)");
  if (c.timing)
    HIP_TRY(hipEventRecord(c.tev[1], s));
  ldpc_dec_args da;
  memset(&da, 0, sizeof(da));
  da.llr = reinterpret_cast<const int8_t *>(c.scratch.p);
  da.out = reinterpret_cast<int8_t *>(c.scratch.p);
  da.n_iter = d_iter;
  da.out_mode = 0;
  da.use_crc = 1;
  for (int k = 0; k < 4; k++)
    da.crc_pow_tbl[k] = G().crc_pow[k];
  da.crc_pow_tbl[NR_HIP_CRC24_A] = G().crc_pow_24a_long;
  uint32_t *d_acc = reinterpret_cast<uint32_t *>(pl.jobs_d.p + o_acc);
  int *d_abort = reinterpret_cast<int *>(d_acc) + ntb; /* zero on entry, left zero by the verdict */
  da.tb_abort = d_abort;
  tb_rx_fused_args fx;
  fx.segs = d_seg; fx.tbs = d_tb; fx.llr = llr; fx.harq = harq; fx.payload = payload; fx.ack = ack; fx.iter_max = iter_max;
  fx.done = d_abort + ntb; fx.gen = reinterpret_cast<uint32_t *>(d_abort + 2 * ntb);
  fx.slots = reinterpret_cast<unsigned long long *>(pl.jobs_d.p + pl.off[6]); fx.pow24a = G().crc_pow_24a_long;
  fx.stagger_ticks = fx.stagger_cus = fx.stagger_slots = 0;
  fx.lrow_off = 0;
  fx.mute = 0;
  fx.zc = 0;
  fx.prio_pro = 0;
  fx.trace = nullptr;
  /* A call that mixes code sizes has several decoder launches (TbPlan::DecLaunch); nothing orders them among themselves --
   * disjoint jobs, scratch rows, per-block state.  NRLDPC_HIP_TB_OVERLAP=1 sends them out on side streams, forked from and
   * joined to the call's stream, so that the CUs one launch leaves free could take the next one's workgroups.  Default 0:
   * with several hardware queues active every kernel of the call runs 2 - 4 x longer on this stack (the 96 large workgroups
   * of the 93-small-TBS mix: 38 -> 162 us; the whole call 0.35 -> 0.53 ms; profiles/r04/small_tbs_overlap.txt) -- round 3
   * had seen the same with the unfused kernels.  What replaces it is in the plan: launches that fill their last workgroup
   * round with jobs of the smaller classes. */
  static const int overlap_env = [] { const char *e = getenv("NRLDPC_HIP_TB_OVERLAP"); return e ? atoi(e) : 0; }();
  const size_t n_side = (overlap_env && pl.dec.size() >= 2 && !c.timing) ? std::min<size_t>(pl.dec.size() - 1, TB_SIDE_STREAMS) : 0;
  if (n_side) {
    for (size_t j = 0; j < TB_SIDE_STREAMS + 1; j++)
      if (!c.side_ev[j])
        HIP_TRY(hipEventCreateWithFlags(&c.side_ev[j], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c.side_ev[TB_SIDE_STREAMS], s)); /* fork: behind the uploads, the copies in and the de-matching launch */
    for (size_t j = 0; j < n_side; j++) {
      if (!c.side[j])
        HIP_TRY(hipStreamCreateWithFlags(&c.side[j], hipStreamNonBlocking));
      HIP_TRY(hipStreamWaitEvent(c.side[j], c.side_ev[TB_SIDE_STREAMS], 0));
    }
  }
  hipStream_t s_call = s;
  for (size_t k = 0; k < pl.dec.size(); k++) {
    const TbPlan::DecLaunch &dl = pl.dec[k];
    const size_t lane = n_side ? k % (n_side + 1) : 0;
    hipStream_t s = lane ? c.side[lane - 1] : s_call; /* (shadows the call's stream inside the loop) */
    da.jobs = reinterpret_cast<const ldpc_dec_job *>(pl.jobs_d.p + dl.jobs_off);
    da.mgroups = reinterpret_cast<const ldpc_dec_mgroup *>(pl.jobs_d.p + dl.grp_off);
    if (dl.kind == 0 && dl.fused) {
      /* more than one workgroup per CU in the first round: staggered start (tb_chain.h).  NRLDPC_HIP_TB_STAGGER_US = the
       * offset between the two workgroups of a CU (k workgroups: 2 / k of it each).  Default 0 = off: measured level to
       * slower at every offset (profiles/r04/fused_stagger_sweep.txt) -- a CU's workgroups only share their prologue in the
       * first round, drift apart by themselves afterwards, and a workgroup that decodes alone is no faster than one that
       * shares its CU (profiles/r04/wg_trace_*.txt) */
      static const int stagger_us = [] { const char *e = getenv("NRLDPC_HIP_TB_STAGGER_US"); return e ? atoi(e) : 0; }();
      const int per_cu = std::min(16 / std::max(dl.threads / 64, 1), (160 * 1024) / std::max(dl.lds, 1024));
      fx.stagger_ticks = 0;
      /* workgroups that share a CU take turns at the issue priority, pass by pass (ldpc_dec_fast_block.h fair_turns;
       * NRLDPC_HIP_TB_FAIR=0: off) */
      static const int fair_env = [] { const char *e = getenv("NRLDPC_HIP_TB_FAIR"); return e ? atoi(e) : 1; }();
      da.fair = (fair_env && per_cu >= 2 && dl.n > (uint32_t)G().n_cus && dl.threads >= 256) ? dl.threads / 256 : 0;
      if (stagger_us > 0 && per_cu >= 2 && dl.n > (uint32_t)G().n_cus) {
        fx.stagger_ticks = (uint32_t)(stagger_us * 100 * 2 / per_cu);
        fx.stagger_cus = (uint32_t)G().n_cus;
        fx.stagger_slots = (uint32_t)per_cu;
      }
      static const char *trace_file = getenv("NRLDPC_HIP_TB_TRACE"); /* diagnostics: per-workgroup clocks of this launch */
      if (trace_file && c.trace_d.ensure((size_t)dl.n * 128) == 0) {
        HIP_TRY(hipMemsetAsync(c.trace_d.p, 0, (size_t)dl.n * 128, s));
        fx.trace = reinterpret_cast<unsigned long long *>(c.trace_d.p);
      }
      fx.lrow_off = dl.lrow;
      fx.mute = dl.mute ? 1u : 0u;
      fx.zc = (uint32_t)dl.zc;
      {
        static const int prio_env = [] { const char *e = getenv("NRLDPC_HIP_TB_PRIO"); return e ? atoi(e) : 1; }();
        fx.prio_pro = per_cu >= 2 ? (uint32_t)prio_env : 0u;
      }
      HIP_TRY(tb_launch_rx_fused(da, fx, dl.threads, dl.lds, dl.n, s));
      TB_DEBUG_STAGE("fused segment kernel");
      if (fx.trace) {
        std::vector<unsigned long long> h((size_t)dl.n * 16);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(h.data(), c.trace_d.p, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_file, "wb")) {
          fwrite(h.data(), 8, h.size(), f);
          fclose(f);
        }
        fx.trace = nullptr;
      }
    }
    else if (dl.kind == 0)
    {
      HIP_TRY(ldpc_launch_dec_fast_jobs(da, dl.threads, dl.lds, dl.n, s, dl.zc));
      TB_DEBUG_STAGE("decoder launch (fast kernel, job array)");
    }
    else if (dl.kind == 1)
      HIP_TRY(ldpc_launch_dec_generic_jobs(da, dl.threads, dl.lds, dl.n, s));
    else
      HIP_TRY(ldpc_launch_dec_fast_multi_jobs(da, dl.kind == 3 ? 4 : 1, dl.threads, dl.lds, dl.n, s));
  }
  for (size_t j = 0; j < n_side; j++) { /* join */
    HIP_TRY(hipEventRecord(c.side_ev[j], c.side[j]));
    HIP_TRY(hipStreamWaitEvent(s, c.side_ev[j], 0));
  }
  if (c.timing)
    HIP_TRY(hipEventRecord(c.tev[2], s));
  if (pl.n_legacy_tb)
    HIP_TRY(tb_launch_rx_assemble(d_tb, ntb, d_leg, (uint32_t)pl.n_legacy_seg, d_iter, c.scratch.p, payload, ack, iter_max, d_acc, d_abort,
                                  G().crc_pow_24a_long, G().crc_pow[NR_HIP_CRC16], s));
  if (c.timing) {
    HIP_TRY(hipEventRecord(c.tev[3], s));
    c.timed = true;
  }
  TB_DEBUG_STAGE("remaining decoder launches + reassembly / verdict");
  if (staged) {
    /* payload back: to the caller's array, or (host call, pageable array) to this thread's page-locked staging area */
    uint8_t *pay_dst = (to_host && c.fin_pay_n) ? c.payload_h.p - c.fin_pay_lo : b->payload;
    /* one copy of the whole range when the piece's blocks fill it, or when the destination is the staging area of a call that
     * comes in ONE piece (the area is ours, what lies between the blocks is never handed over).  A call in several pieces
     * shares the staging area: with payload offsets that do not grow with the block index a later piece's range can span
     * blocks an earlier piece has already delivered, so there only the blocks themselves are copied, run by run */
    const bool several_pieces = st && st->call_ntb != ntb;
    if (pl.out_dense || (pay_dst != b->payload && !several_pieces)) {
      HIP_TRY(hipMemcpyAsync(pay_dst + pl.ext[0], c.io_payload.p, pl.ext[1] - pl.ext[0], hipMemcpyDefault, s));
    } else {
      for (const TbPlan::OutRun &r : pl.out_runs)
        HIP_TRY(hipMemcpy2DAsync(pay_dst + r.first, r.pitch, c.io_payload.p + (r.first - pl.ext[0]), r.pitch, r.width, r.rows,
                                 hipMemcpyDefault, s));
    }
    if (harq_staged)
      for (const TbPlan::HarqRun &r : pl.harq_runs) {
        /* a bounced run comes down in pieces of ~8 MB with an event behind each: tb_rx_finish hands piece k over to the
         * caller's array while piece k + 1 is on the link (TbCtx::finish_copies) */
        const uint32_t step = harq_bounce ? std::max<uint32_t>(1, (uint32_t)(((size_t)8 << 20) / ((size_t)r.width * 2))) : r.rows;
        for (uint32_t q0 = 0; q0 < r.rows; q0 += step) {
          const uint32_t nq = std::min(step, r.rows - q0);
          const size_t first = r.first + (size_t)q0 * b->harq_stride;
          uint8_t *dst = harq_bounce ? harq_mirror + (first - pl.ext[4]) * 2 : reinterpret_cast<uint8_t *>(b->harq + first);
          HIP_TRY(hipMemcpy2DAsync(dst, stride2, c.io_harq.p + (first - pl.ext[4]) * 2, stride2, (size_t)r.width * 2, nq, hipMemcpyDefault, s));
          if (harq_bounce) {
            hipEvent_t ev = c.next_fin_event();
            if (ev)
              HIP_TRY(hipEventRecord(ev, s));
            c.fin_copies.push_back(TbCtx::FinCopy{reinterpret_cast<uint8_t *>(b->harq + first), dst, (size_t)r.width * 2, nq, stride2, stride2, ev});
          }
        }
      }
    if (!to_host) { /* a peer GPU's share of a device-resident batch: the verdicts go to the owner's arrays */
      HIP_TRY(hipMemcpyAsync(b->iter_max + tb0, c.io_small.p, (size_t)ntb * 4, hipMemcpyDefault, s));
      HIP_TRY(hipMemcpyAsync(b->ack + tb0, c.io_small.p + (size_t)ntb * 4, ntb, hipMemcpyDefault, s));
    }
  }
  return 0;
}

int tb_rx_finish(const nrLDPC_hip_tb_batch_t *b, uint32_t tb0, uint32_t ntb)
{
  if ((b->mem & NRLDPC_HIP_MEM_DEVICE) || ntb == 0)
    return 0;
  TbCtx &c = tls_tb;
  c.finish_copies(false); /* first: it waits run by run and copies while the later runs are still on the link */
  HIP_TRY(hipStreamSynchronize(c.own));
  memcpy(b->iter_max + tb0, c.small_h.p, (size_t)ntb * 4); /* (written by the kernels: the area is device-mapped) */
  memcpy(b->ack + tb0, c.small_h.p + (size_t)ntb * 4, ntb);
  if (c.fin_pay_n) { /* block by block: nothing between the blocks' payloads is touched */
    for (uint32_t i = tb0; i < tb0 + ntb; i++)
      memcpy(b->payload + b->tb[i].payload_off, c.payload_h.p + (b->tb[i].payload_off - c.fin_pay_lo), b->tb[i].A / 8);
    c.fin_pay_n = 0;
  }
  return 0;
}

/* Whole transport blocks per device, contiguous index ranges, balanced by decoder work ~ segments x edges x Zc (SURVEY
 * 8e: a TB stays on one GPU so that its CRC, its abort flag and its HARQ buffers stay local).  cut[k] .. cut[k+1] = the
 * share of part k. */
void tb_partition(const nrLDPC_hip_tb_batch_t *b, int parts, uint32_t *cut, uint32_t first = 0, uint32_t count = UINT32_MAX)
{
  if (count == UINT32_MAX)
    count = b->n_tb - first;
  std::vector<double> cost(count, 1.0);
  double total = 0;
  for (uint32_t i = 0; i < count; i++) {
    const nrLDPC_hip_tb_t &t = b->tb[first + i];
    nr_hip_seg_t sg;
    if (t.A && (t.BG == 1 || t.BG == 2) && nr_hip_segmentation((uint32_t)nr_hip_len_with_crc(1, (int)t.A), t.BG, &sg) == 0)
      cost[i] = (double)sg.C * sg.Zc * (t.BG == 1 ? 316.0 : 197.0);
    total += cost[i];
  }
  cut[0] = first;
  double acc = 0;
  uint32_t i = 0;
  for (int k = 1; k < parts; k++) {
    const double target = total * k / parts;
    while (i < count && acc + cost[i] * 0.5 <= target)
      acc += cost[i++];
    cut[k] = first + i;
  }
  cut[parts] = first + count;
}

/* Host buffers, one device's share [tb0, tb0 + ntb) of a call.  Small calls: one piece (page-locked LLRs read in place).
 * Large calls: the LLRs cross the link through the copy engine in K chunks of whole transport blocks on a stream of their
 * own, back to back at the link's streaming rate; chunk k is de-matched and decoded -- on the compute stream, behind an
 * event -- while chunks k+1.. are still on the link, so that what a call costs beyond its transfer is one chunk's decoding
 * and one copy -> kernel edge (the reference overlaps the same way: segments are decoded by the pool while the next
 * symbols are still being demodulated). */
int tb_rx_enqueue_host(const nrLDPC_hip_tb_batch_t *b, uint32_t tb0, uint32_t ntb)
{
  RxHostStage st{nullptr, false, tb0, ntb, 0, 0};
  if (ntb == 0)
    return tb_rx_enqueue(b, tb0, ntb, true, nullptr, &st);
  size_t lo = SIZE_MAX, hi = 0;
  st.pay_lo = SIZE_MAX;
  for (uint32_t i = tb0; i < tb0 + ntb; i++) {
    lo = std::min(lo, (size_t)b->tb[i].coded_off);
    hi = std::max(hi, (size_t)b->tb[i].coded_off + b->tb[i].G);
    st.pay_lo = std::min(st.pay_lo, (size_t)b->tb[i].payload_off);
    st.pay_hi = std::max(st.pay_hi, (size_t)b->tb[i].payload_off + b->tb[i].A / 8);
  }
  const size_t bytes = (hi - lo) * sizeof(int16_t);
  static const size_t pull_max = [] { const char *e = getenv("NRLDPC_HIP_TB_PULL_MAX_MB"); return e ? (size_t)atoi(e) << 20 : SIZE_MAX; }();
  static const int chunks_env = [] { const char *e = getenv("NRLDPC_HIP_TB_HOST_CHUNKS"); return e ? atoi(e) : 0; }(); /* tuning knob */
  const int mode = tb_pull_mode();
  int K = chunks_env > 0 ? chunks_env : (int)std::min<size_t>(8, bytes / ((size_t)5 << 20));
  K = std::min<int>(std::min<int>(K, 8), (int)ntb / 2); /* cut[] below holds 8 chunks; the knob is not trusted */
  const bool pinned = host_ptr_is_pinned(static_cast<const int16_t *>(b->coded) + lo, bytes);
  if (K < 2 || (pinned && (mode == 2 || (mode == 1 && bytes < pull_max))))
    return tb_rx_enqueue(b, tb0, ntb, true, nullptr, &st);
  hipStream_t s;
  if (tb_begin(s, nullptr, true) != 0)
    return -1;
  TbCtx &c = tls_tb;
  if (!c.aux)
    HIP_TRY(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
  while (c.chunk_ev.size() < (size_t)K + 1) {
    hipEvent_t ev;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c.chunk_ev.push_back(ev);
  }
  if (c.io_coded.ensure(bytes) != 0)
    return -1;
  if (!pinned && c.coded_h.ensure(bytes) != 0) /* pageable: every chunk is bounced through the page-locked mirror by the CPU */
    return -1;
  /* the staging buffer may still be read by the previous call's kernels on the compute stream */
  HIP_TRY(hipEventRecord(c.chunk_ev[K], s));
  HIP_TRY(hipStreamWaitEvent(c.aux, c.chunk_ev[K], 0));
  uint32_t cut[9];
  tb_partition(b, K, cut, tb0, ntb);
  const int16_t *src = static_cast<const int16_t *>(b->coded);
  st.llr_dev = reinterpret_cast<const int16_t *>(c.io_coded.p) - lo;
  st.no_pull = true;
  auto copy_chunk = [&](int k) -> int {
    size_t a = SIZE_MAX, e = 0;
    for (uint32_t i = cut[k]; i < cut[k + 1]; i++) {
      a = std::min(a, (size_t)b->tb[i].coded_off);
      e = std::max(e, (size_t)b->tb[i].coded_off + b->tb[i].G);
    }
    if (e > a) {
      const void *from = src + a;
      if (!pinned && !pageable_direct()) {
        bounce_copy(c.coded_h.p + (a - lo) * 2, (e - a) * 2, from, (e - a) * 2, (e - a) * 2, 1);
        from = c.coded_h.p + (a - lo) * 2;
      }
      HIP_TRY(hipMemcpyAsync(c.io_coded.p + (a - lo) * 2, from, (e - a) * 2, hipMemcpyHostToDevice, c.aux));
    }
    HIP_TRY(hipEventRecord(c.chunk_ev[k], c.aux));
    return 0;
  };
  if (copy_chunk(0) != 0)
    return -1;
  for (int k = 0; k < K; k++) { /* the copy lane always holds the next chunk's copy while this one's kernels are enqueued */
    if (k + 1 < K && copy_chunk(k + 1) != 0)
      return -1;
    HIP_TRY(hipStreamWaitEvent(s, c.chunk_ev[k], 0));
    if (tb_rx_enqueue(b, cut[k], cut[k + 1] - cut[k], true, nullptr, &st) != 0)
      return -1;
  }
  return 0;
}

/* Run a batch over the library's GPUs.  enq(tb0, n, staged, stream) / fin(tb0, n) work on the current device.
 *   host buffers: contiguous whole-TB ranges, one per sharding device, every device fed over its own link; enqueue on all of
 *     them, then finish in order (so that they work concurrently);
 *   device buffers: the GPU that owns them takes its range in place on the caller's stream; with NRLDPC_HIP_DEVICES listing
 *     peers the other ranges are copied GPU to GPU (xGMI) into the peers' staging buffers, worked on there and the results
 *     copied back -- ordered behind what the caller's stream held on entry, and the caller's stream waits for them on exit, so
 *     the call stays "enqueue only" (whole transport blocks per GPU, one exchange step each way, no all-reduce). */
template <typename Enq, typename Fin> int tb_run_sharded(const nrLDPC_hip_tb_batch_t *b, Enq enq, Fin fin)
{
  if (b->mem & NRLDPC_HIP_MEM_DEVICE) {
    Device *d = device_of_pointer(b->coded);
    if (!d)
      return -1;
    hipStream_t cs = static_cast<hipStream_t>(b->stream);
    int owner = -1;
    for (int k = 0; k < g.n_shard; k++)
      if (&g.dev[k] == d)
        owner = k;
    const int parts = (owner >= 0 && b->n_tb >= 2u * (uint32_t)g.n_shard) ? g.n_shard : 1;
    if (parts == 1) {
      UseDevice use(*d);
      return enq(0u, b->n_tb, false, cs) != 0 ? -1 : fin(0u, b->n_tb);
    }
    uint32_t cut[NRLDPC_HIP_MAX_DEVICES + 1];
    tb_partition(b, parts, cut);
    int rc = 0;
    hipEvent_t ev_in = nullptr;
    {
      UseDevice use(*d);
      TbCtx &c = tls_tb;
      if (!c.ev_in && hipEventCreateWithFlags(&c.ev_in, hipEventDisableTiming) != hipSuccess)
        return set_error("event");
      ev_in = c.ev_in;
      HIP_TRY(hipEventRecord(ev_in, cs)); /* what the caller's stream has produced so far: the peers read it */
    }
    hipEvent_t ev_out[NRLDPC_HIP_MAX_DEVICES];
    int n_out = 0;
    for (int k = 0; k < parts && rc == 0; k++) {
      const uint32_t n = cut[k + 1] - cut[k];
      if (n == 0)
        continue;
      UseDevice use(g.dev[k]);
      if (k == owner) {
        rc = enq(cut[k], n, false, cs);
        continue;
      }
      TbCtx &c = tls_tb;
      hipStream_t s;
      if (tb_begin(s, nullptr, true) != 0) { rc = -1; break; }
      if (!c.ev_out && hipEventCreateWithFlags(&c.ev_out, hipEventDisableTiming) != hipSuccess) { rc = set_error("event"); break; }
      if (hipStreamWaitEvent(s, ev_in, 0) != hipSuccess) { rc = set_error("hipStreamWaitEvent"); break; }
      rc = enq(cut[k], n, true, nullptr);
      if (rc == 0 && hipEventRecord(c.ev_out, s) != hipSuccess)
        rc = set_error("hipEventRecord");
      if (rc == 0)
        ev_out[n_out++] = c.ev_out;
    }
    {
      UseDevice use(*d);
      for (int q = 0; q < n_out; q++)
        if (hipStreamWaitEvent(cs, ev_out[q], 0) != hipSuccess)
          rc = set_error("hipStreamWaitEvent");
    }
    return rc;
  }
  if (ensure_ready() != 0)
    return -1;
  const int parts = (b->n_tb >= 2u * (uint32_t)g.n_shard) ? g.n_shard : 1;
  uint32_t cut[NRLDPC_HIP_MAX_DEVICES + 1];
  tb_partition(b, parts, cut);
  int rc = 0;
  for (int k = 0; k < parts && rc == 0; k++) {
    UseDevice use(g.dev[k]);
    rc = enq(cut[k], cut[k + 1] - cut[k], true, nullptr);
  }
  for (int k = 0; k < parts; k++) { /* also after an error: nothing stays in flight */
    UseDevice use(g.dev[k]);
    if (rc == 0)
      rc = fin(cut[k], cut[k + 1] - cut[k]);
    else
      tls_tb.drain();
  }
  return rc;
}

} // namespace

extern "C" {

int32_t nrLDPC_hip_dlsch_encode(const nrLDPC_hip_tb_batch_t *b)
{
  if (!b || !b->tb || !b->payload || !b->coded)
    return set_error("null argument");
  if (b->mem & ~NRLDPC_HIP_MEM_DEVICE)
    return set_error("encode: mem must be NRLDPC_HIP_MEM_HOST or NRLDPC_HIP_MEM_DEVICE");
  if (b->n_tb == 0)
    return ensure_ready();
  return tb_run_sharded(
      b, [&](uint32_t tb0, uint32_t n, bool staged, hipStream_t s) { return tb_tx_enqueue(b, tb0, n, staged, s); },
      [&](uint32_t, uint32_t n) { return tb_tx_finish(b, n); });
}

int32_t nrLDPC_hip_ulsch_decode(const nrLDPC_hip_tb_batch_t *b)
{
  if (!b || !b->tb || !b->payload || !b->coded)
    return set_error("null argument");
  const bool harq_lib = (b->mem & NRLDPC_HIP_MEM_HARQ_LIBRARY) != 0;
  if ((b->mem & ~(NRLDPC_HIP_MEM_DEVICE | NRLDPC_HIP_MEM_HARQ_DEVICE | NRLDPC_HIP_MEM_HARQ_LIBRARY)) ||
      (harq_lib && (b->mem & NRLDPC_HIP_MEM_HARQ_DEVICE)))
    return set_error("decode: invalid mem flags");
  if ((!b->harq && !harq_lib) || !b->ack || !b->iter_max || b->harq_stride < 66 * 384)
    return set_error("decode needs harq (stride >= 66*384), ack and iter_max buffers");
  if (b->n_tb == 0)
    return ensure_ready();
  return tb_run_sharded(
      b,
      [&](uint32_t tb0, uint32_t n, bool staged, hipStream_t s) {
        return (staged && !(b->mem & NRLDPC_HIP_MEM_DEVICE)) ? tb_rx_enqueue_host(b, tb0, n) : tb_rx_enqueue(b, tb0, n, staged, s);
      },
      [&](uint32_t tb0, uint32_t n) { return tb_rx_finish(b, tb0, n); });
}

int32_t nrLDPC_hip_harq_release(uint64_t id)
{
  std::lock_guard<std::mutex> lk(harq_tbl.mu);
  auto it = harq_tbl.m.find(id);
  if (it == harq_tbl.m.end())
    return set_error("unknown soft-buffer id");
  harq_tbl.pool.push_back(it->second);
  harq_tbl.m.erase(it);
  harq_tbl.gen.fetch_add(1);
  return 0;
}

int32_t nrLDPC_hip_harq_release_all(void)
{
  std::lock_guard<std::mutex> lk(harq_tbl.mu);
  for (auto &kv : harq_tbl.m)
    harq_tbl.pool.push_back(kv.second);
  harq_tbl.m.clear();
  harq_tbl.gen.fetch_add(1);
  return 0;
}

int32_t nrLDPC_hip_harq_read(uint64_t id, int16_t *dst, uint64_t first, uint64_t n)
{
  HarqEntry e;
  {
    std::lock_guard<std::mutex> lk(harq_tbl.mu);
    auto it = harq_tbl.m.find(id);
    if (it == harq_tbl.m.end())
      return set_error("unknown soft-buffer id");
    e = it->second;
  }
  if (!dst || first > e.n || n > e.n - first) /* (written so that huge arguments cannot wrap) */
    return set_error("range outside the soft buffers");
  /* (no device-wide wait: the resident server kernels of the per-segment entry points may be running for as long as
   * requests keep coming.  A host-memory decode call has finished when it returns; after a device-memory call the caller
   * waits for its own stream before it looks.) */
  UseDevice use(g.dev[e.dev]);
  HIP_TRY(hipMemcpy(dst, e.p + first, n * sizeof(int16_t), hipMemcpyDeviceToHost));
  return 0;
}

int32_t nrLDPC_hip_chain_timing(int32_t enable, float out_us[4])
{
  if (ensure_ready() != 0)
    return -1;
  UseDevice use(g.dev[0]);
  TbCtx &c = tls_tb;
  int rc = 0;
  if (out_us) {
    if (!c.timed)
      return set_error("no timed UL-SCH call on this thread yet");
    HIP_TRY(hipEventSynchronize(c.tev[3]));
    float ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
      HIP_TRY(hipEventElapsedTime(&ms[k], c.tev[k], c.tev[k + 1]));
    out_us[0] = ms[0] * 1e3f; out_us[1] = ms[1] * 1e3f; out_us[2] = ms[2] * 1e3f; out_us[3] = (ms[0] + ms[1] + ms[2]) * 1e3f;
  }
  c.timing = enable != 0;
  return rc;
}

void *nrLDPC_hip_host_alloc(uint64_t bytes)
{
  if (ensure_ready() != 0)
    return nullptr;
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc");
    return nullptr;
  }
  return p;
}
void nrLDPC_hip_host_free(void *p)
{
  if (p)
    (void)hipHostFree(p);
}
int32_t nrLDPC_hip_host_register(void *p, uint64_t bytes)
{
  if (ensure_ready() != 0)
    return -1;
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return 0;
}
int32_t nrLDPC_hip_host_unregister(void *p)
{
  HIP_TRY(hipHostUnregister(p));
  return 0;
}

} /* extern "C" */
