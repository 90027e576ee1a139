/*
 * ldpc_aggregator.inc.cpp -- call aggregation behind the per-segment entry point.
 * Included by ldpc_api.cpp (shares its library state).
 *
 * The reference's callers invoke LDPCdecoder from N thread-pool workers at once, one code segment per call
 * (openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:435-468, nr_dlsch_decoding.c).  A GPU wants them in one launch.
 * Combining protocol ("natural batching", no timers by default): a caller queues its request.  At most
 * NRLDPC_HIP_AGG_INFLIGHT (default 2) batches execute at a time; while they do, new arrivals pile up in the queue.
 * Whenever a slot is free, one queued caller becomes the leader, takes the WHOLE queue, stages all inputs through its
 * pinned buffers, launches ONE heterogeneous job-array decode (requests may differ in BG/Zc/R/numMaxIter/E), copies
 * the results back and wakes the followers.  A lone caller (ldpctest) therefore never waits for anybody, and under
 * load the batch size grows by itself to whatever arrived during one GPU round trip.  Every request keeps its own
 * return value, output buffer and abort flag, so the call semantics of LDPCdecoder are unchanged.
 *
 * Environment: NRLDPC_HIP_AGGREGATE=1 enables it (default: each call launches on its own per-thread stream, which
 * measured as fast or faster up to 32 caller threads -- profiles/r01/abi_threads.txt);
 * NRLDPC_HIP_AGG_INFLIGHT=<n> batches in flight; NRLDPC_HIP_AGG_US=<n> optional extra collection window in
 * microseconds (default 0), ended early by NRLDPC_HIP_AGG_MAX=<n> queued requests (default 64).
 */
#include <chrono>
#include <condition_variable>

namespace {

struct AggReq {
  const t_nrLDPC_dec_params *p;
  const CodeEntry *ce;
  const int8_t *llr;
  int8_t *out;
  int32_t n_iter = 0;
  int rc = 0;
  bool taken = false; /* a leader has moved it from the queue into its batch */
  bool done = false;
};

struct Aggregator {
  std::mutex mu;
  std::condition_variable cv_more;  /* leader: more requests arrived */
  std::condition_variable cv_done;  /* followers: a batch completed */
  std::vector<AggReq *> queue;
  bool collecting = false;
  int running = 0; /* batches executing right now */
  int enabled = -1, window_us = 0, max_batch = 64, max_running = 2;
} agg;

void agg_config_locked()
{
  if (agg.enabled >= 0)
    return;
  const char *e = getenv("NRLDPC_HIP_AGGREGATE");
  agg.enabled = (e && atoi(e) != 0) ? 1 : 0; /* opt-in: see the measurements in DESIGN.md section 4.5 */
  if ((e = getenv("NRLDPC_HIP_AGG_US")) && atoi(e) >= 0)
    agg.window_us = atoi(e);
  if ((e = getenv("NRLDPC_HIP_AGG_MAX")) && atoi(e) >= 1)
    agg.max_batch = atoi(e);
  if ((e = getenv("NRLDPC_HIP_AGG_INFLIGHT")) && atoi(e) >= 1)
    agg.max_running = atoi(e);
}

/* decode one group of requests that share (use_crc, out_mode) with a single job-array launch per kernel type */
int agg_run_group(std::vector<AggReq *> &reqs)
{
  ThreadCtx &c = tls_ctx;
  const size_t n = reqs.size();
  std::vector<size_t> in_off(n), out_off(n);
  size_t in_bytes = 0, out_bytes = 0;
  for (size_t i = 0; i < n; i++) {
    const ldpc_code_desc_t &hc = reqs[i]->ce->host;
    in_off[i] = in_bytes;
    in_bytes += align_up(hc.num_llr, 16);
    out_off[i] = out_bytes;
    out_bytes += align_up(out_bytes_of(hc, reqs[i]->p->outMode == nrLDPC_outMode_BIT ? 0 : 1), 16);
  }
  const size_t jobs_bytes = align_up(n * sizeof(ldpc_dec_job), 16);
  /* the job records travel in front of the LLRs in the same staging buffer */
  if (c.ensure(jobs_bytes + in_bytes, out_bytes, n) != 0)
    return -1;
  ldpc_dec_job *hj = reinterpret_cast<ldpc_dec_job *>(c.h_in);
  std::vector<uint32_t> fast_idx, gen_idx;
  int fast_threads = 64, fast_lds = 0, gen_threads = 64, gen_lds = 0;
  ldpc_dec_args a;
  if (fill_dec_args(*reqs[0]->p, reqs[0]->ce, a) != 0)
    return -1;
  for (size_t i = 0; i < n; i++) {
    const ldpc_code_desc_t &hc = reqs[i]->ce->host;
    ldpc_dec_args ai;
    if (fill_dec_args(*reqs[i]->p, reqs[i]->ce, ai) != 0) {
      reqs[i]->rc = -1; /* bad parameters: this request only; it still occupies a (skipped) slot */
      continue;
    }
    memcpy(c.h_in + jobs_bytes + in_off[i], reqs[i]->llr, hc.num_llr);
    (hc.f_ok ? fast_idx : gen_idx).push_back((uint32_t)i);
    if (hc.f_ok) { /* a handful of blocks per launch: latency shape */
      fast_threads = std::max(fast_threads, reqs[i]->ce->host_lat.f_n_threads);
      fast_lds = std::max(fast_lds, reqs[i]->ce->host_lat.f_lds_total);
    } else {
      gen_threads = std::max(gen_threads, hc.n_threads);
      gen_lds = std::max(gen_lds, hc.lds_total);
    }
  }
  /* job records: fast ones first, then generic ones */
  size_t k = 0;
  for (int pass = 0; pass < 2; pass++)
    for (uint32_t i : (pass == 0 ? fast_idx : gen_idx)) {
      ldpc_dec_job &j = hj[k++];
      j.code = pass == 0 ? reqs[i]->ce->dev_lat : reqs[i]->ce->dev;
      j.llr_off = jobs_bytes + in_off[i];
      j.out_off = out_off[i];
      j.num_max_iter = reqs[i]->p->numMaxIter;
      j.E = reqs[i]->p->check_crc ? reqs[i]->p->E : 0;
      j.crc_type = reqs[i]->p->check_crc ? reqs[i]->p->crc_type : 0;
      j.iter_idx = (int32_t)i;
    }
  HIP_TRY(hipMemcpyAsync(c.d_in, c.h_in, jobs_bytes + in_bytes, hipMemcpyHostToDevice, c.stream));
  a.llr = reinterpret_cast<const int8_t *>(c.d_in);
  a.out = reinterpret_cast<int8_t *>(c.d_out);
  a.n_iter = c.d_iter;
  const ldpc_dec_job *dj = reinterpret_cast<const ldpc_dec_job *>(c.d_in);
  if (!fast_idx.empty()) {
    a.jobs = dj;
    HIP_TRY(ldpc_launch_dec_fast_jobs(a, fast_threads, fast_lds, (uint32_t)fast_idx.size(), c.stream));
  }
  if (!gen_idx.empty()) {
    a.jobs = dj + fast_idx.size();
    HIP_TRY(ldpc_launch_dec_generic_jobs(a, gen_threads, gen_lds, (uint32_t)gen_idx.size(), c.stream));
  }
  HIP_TRY(hipMemcpyAsync(c.h_out, c.d_out, out_bytes, hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipMemcpyAsync(c.h_iter, c.d_iter, n * sizeof(int32_t), hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  for (size_t i = 0; i < n; i++) {
    if (reqs[i]->rc != 0)
      continue;
    const int32_t it = c.h_iter[i];
    reqs[i]->n_iter = it;
    if (!reqs[i]->p->check_crc || it >= 3) /* the reference leaves p_out untouched otherwise (decoder.c:849-861) */
      memcpy(reqs[i]->out, c.h_out + out_off[i],
             out_bytes_of(reqs[i]->ce->host, reqs[i]->p->outMode == nrLDPC_outMode_BIT ? 0 : 1));
  }
  return 0;
}

/* returns 0 and fills *n_iter, or -1 */
int agg_decode(const t_nrLDPC_dec_params *p, const CodeEntry *ce, const int8_t *llr, int8_t *out, int32_t *n_iter)
{
  AggReq req;
  req.p = p; req.ce = ce; req.llr = llr; req.out = out;
  std::unique_lock<std::mutex> lk(agg.mu);
  agg_config_locked();
  agg.queue.push_back(&req);
  if (agg.collecting)
    agg.cv_more.notify_one(); /* the collecting leader may have reached its batch size */
  for (;;) {
    if (req.done)
      return req.rc == 0 ? (*n_iter = req.n_iter, 0) : set_error("aggregated decode failed");
    if (req.taken || agg.collecting || agg.running >= agg.max_running) {
      /* in flight with some leader, or queued while a leader collects / every execution slot is busy: sleep until
       * done, or until the request is still queued and a slot and the leadership are free (then this thread leads) */
      agg.cv_done.wait(lk, [&] { return req.done || (!req.taken && !agg.collecting && agg.running < agg.max_running); });
      continue;
    }
    agg.collecting = true;
    if (agg.window_us > 0)
      agg.cv_more.wait_for(lk, std::chrono::microseconds(agg.window_us), [&] { return (int)agg.queue.size() >= agg.max_batch; });
    std::vector<AggReq *> batch;
    batch.swap(agg.queue);
    for (AggReq *r : batch)
      r->taken = true;
    agg.running++;
    agg.collecting = false;
    agg.cv_done.notify_all(); /* a queued follower may become the next leader while this batch runs */
    lk.unlock();
    /* one launch set per (stop mode, output mode) group -- normally a single group */
    int rc_all = 0;
    std::vector<bool> taken(batch.size(), false);
    for (size_t i = 0; i < batch.size(); i++) {
      if (taken[i])
        continue;
      std::vector<AggReq *> grp;
      for (size_t k2 = i; k2 < batch.size(); k2++)
        if (!taken[k2] && (batch[k2]->p->check_crc != nullptr) == (batch[i]->p->check_crc != nullptr) &&
            (batch[k2]->p->outMode == nrLDPC_outMode_BIT) == (batch[i]->p->outMode == nrLDPC_outMode_BIT)) {
          taken[k2] = true;
          grp.push_back(batch[k2]);
        }
      if (agg_run_group(grp) != 0) {
        rc_all = -1;
        for (AggReq *r : grp)
          r->rc = -1;
      }
    }
    (void)rc_all;
    lk.lock();
    agg.running--;
    for (AggReq *r : batch)
      r->done = true;
    agg.cv_done.notify_all();
    /* loop: this thread's own request is in `batch`, so req.done is true now */
  }
}

} // namespace
