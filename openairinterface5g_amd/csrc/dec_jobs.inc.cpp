/*
 * dec_jobs.inc.cpp -- LDPCdecoder_jobs: a mixed batch of code blocks (every block with its own code and buffers) in one
 * call (included at the end of ldpc_api.cpp, after tb_api.inc.cpp whose per-thread context and plan cache it shares).
 * The reference decodes such a mix one LDPCdecoder() call per segment on its worker pool (nr_ulsch_decoding.c:435-468);
 * here the segments of all codes share job-array launches of the decoder kernels, cut by workgroup shape.
 */
namespace {

int dec_jobs_enqueue(const nrLDPC_hip_dec_job_t *jobs, uint32_t n, int32_t *n_iter, hipStream_t s)
{
  hipStream_t s0;
  if (tb_begin(s0, s, false) != 0)
    return -1;
  TbCtx &c = tls_tb;
  const uint64_t salt[3] = {0x6a6f6273ull /* "jobs" */, (uint64_t)tb_classes_enabled() | ((uint64_t)(tb_fill_mode() & 0xff) << 8),
                            (uint64_t)reinterpret_cast<uintptr_t>(n_iter)};
  /* the plan cache compares descriptor BYTES: a job array is just another kind of descriptor */
  static_assert(sizeof(nrLDPC_hip_dec_job_t) % 8 == 0, "job records are compared as bytes");
  const size_t key_bytes = (size_t)n * sizeof(nrLDPC_hip_dec_job_t);
  TbPlan *hit = c.cb.find_raw(jobs, key_bytes, salt);
  TbPlan &pl = hit ? *hit : c.cb.victim();
  const t_nrLDPC_dec_params &p0 = jobs[0].params;
  const int out_mode = p0.outMode == nrLDPC_outMode_BIT ? 0 : 1, use_crc = p0.check_crc != nullptr;
  if (!hit) {
    struct ShapedJob { ldpc_dec_job dj; int kind, threads, lds; double cost; int zc; };
    std::vector<ShapedJob> v(n);
    const bool lat_shape = n <= (uint32_t)G().n_cus;
    const bool classes = tb_classes_enabled() && !lat_shape;
    for (uint32_t i = 0; i < n; i++) {
      const t_nrLDPC_dec_params &p = jobs[i].params;
      if (!jobs[i].llr || !jobs[i].out)
        return set_error("null buffer in job");
      if ((p.outMode == nrLDPC_outMode_BIT ? 0 : 1) != out_mode || (p.check_crc != nullptr) != (use_crc != 0))
        return set_error("LDPCdecoder_jobs: outMode and stop mode must be the same for all blocks");
      const CodeEntry *ce = get_code(p.BG, p.Z, p.R);
      if (!ce)
        return -1;
      ldpc_dec_args tmp;
      if (fill_dec_args(p, ce, tmp) != 0) /* (CRC parameters of the block) */
        return -1;
      if (reinterpret_cast<uintptr_t>(jobs[i].out) & 3)
        return set_error("out must be 4-byte aligned");
      const ldpc_code_desc_t &hc = ce->host, &shape = lat_shape ? ce->host_lat : ce->host;
      const bool fast = hc.f_ok && (reinterpret_cast<uintptr_t>(jobs[i].llr) & 3) == 0;
      ldpc_dec_job &dj = v[i].dj;
      dj.code = fast ? (lat_shape ? ce->dev_lat : ce->dev) : ce->dev;
      dj.llr_off = reinterpret_cast<uintptr_t>(jobs[i].llr); /* absolute: the launches get a null base */
      dj.out_off = reinterpret_cast<uintptr_t>(jobs[i].out);
      dj.num_max_iter = p.numMaxIter;
      dj.E = use_crc ? p.E : 0;
      dj.crc_type = use_crc ? p.crc_type : 0;
      dj.iter_idx = (int32_t)i;
      dj.abort_idx = -1;
      dj.seg_idx = -1;
      v[i].kind = fast ? 0 : 1;
      v[i].threads = fast ? shape.f_n_threads : hc.n_threads;
      v[i].lds = fast ? shape.f_lds_total : hc.lds_total;
      v[i].cost = (double)hc.num_llr * p.numMaxIter;
      v[i].zc = (fast && shape.f_mb == 1 && shape.f_rstride == shape.Z + 4 && shape.f_astride == 2 * shape.Z) ? shape.Z : 0;
    }
    auto per_cu_of = [&](int kind, int threads, int lds) {
      const int waves = kind == 0 ? 16 : 32;
      return std::max(1, std::min(waves * 64 / std::max(threads, 64), (160 * 1024) / std::max(lds, 1024)));
    };
    auto wg_class = [&](const ShapedJob &j) {
      if (!classes)
        return 0;
      const int per_cu = per_cu_of(j.kind, j.threads, j.lds);
      return per_cu >= 16 ? 4 : per_cu >= 8 ? 3 : per_cu >= 4 ? 2 : per_cu >= 2 ? 1 : 0;
    };
    std::stable_sort(v.begin(), v.end(), [&](const ShapedJob &x, const ShapedJob &y) {
      const int cx = x.kind * 8 + wg_class(x), cy = y.kind * 8 + wg_class(y);
      return cx != cy ? cx < cy : (classes && x.cost > y.cost);
    });
    std::vector<TbPlan::DecLaunch> dec;
    for (size_t q = 0; q < v.size();) { /* as in the transport-block chain's plan (tb_api.inc.cpp): class by class, last rounds filled */
      const int kind = v[q].kind, cls = wg_class(v[q]);
      size_t e = q;
      int threads = 64, lds = 0;
      while (e < v.size() && v[e].kind == kind && wg_class(v[e]) == cls) {
        threads = std::max(threads, v[e].threads);
        lds = std::max(lds, v[e].lds);
        e++;
      }
      if (classes && tb_fill_mode()) {
        const size_t slots = (size_t)G().n_cus * (size_t)per_cu_of(kind, threads, lds);
        const size_t rem = (e - q) % slots;
        size_t room = rem ? slots - rem : 0;
        while (room && e < v.size() && v[e].kind == kind && v[e].threads <= threads && v[e].lds <= lds) {
          e++;
          room--;
        }
      }
      TbPlan::DecLaunch dl{kind, q * sizeof(ldpc_dec_job), 0, (uint32_t)(e - q), threads, lds, false};
      dl.zc = v[q].zc;
      for (size_t i = q; i < e; i++)
        if (v[i].zc != dl.zc)
          dl.zc = 0;
      dec.push_back(dl);
      q = e;
    }
    const size_t bytes = align_up((size_t)n * sizeof(ldpc_dec_job), 16);
    if (tb_wait_upload(c) != 0 || c.jobs_h.ensure(bytes) != 0 || pl.jobs_d.ensure(bytes) != 0)
      return -1;
    ldpc_dec_job *h = reinterpret_cast<ldpc_dec_job *>(c.jobs_h.p);
    for (uint32_t i = 0; i < n; i++)
      h[i] = v[i].dj;
    if (tb_upload_jobs(c, pl.jobs_d.p, bytes, s) != 0)
      return -1;
    pl.dec.swap(dec);
    pl.remember_raw(jobs, key_bytes, salt);
  }
  ldpc_dec_args da;
  memset(&da, 0, sizeof(da));
  da.n_iter = n_iter;
  da.out_mode = out_mode;
  da.use_crc = use_crc;
  for (int k = 0; k < 4; k++)
    da.crc_pow_tbl[k] = G().crc_pow[k];
  for (const TbPlan::DecLaunch &dl : pl.dec) {
    da.jobs = reinterpret_cast<const ldpc_dec_job *>(pl.jobs_d.p + dl.jobs_off);
    if (dl.kind == 0)
      HIP_TRY(ldpc_launch_dec_fast_jobs(da, dl.threads, dl.lds, dl.n, s, dl.zc));
    else
      HIP_TRY(ldpc_launch_dec_generic_jobs(da, dl.threads, dl.lds, dl.n, s));
  }
  return 0;
}

} // namespace

extern "C" int32_t LDPCdecoder_jobs(const nrLDPC_hip_dec_job_t *jobs, uint32_t n_jobs, int32_t *n_iter, int32_t mem, void *stream)
{
  if (!jobs || !n_iter)
    return set_error("null argument");
  if (mem != NRLDPC_HIP_MEM_DEVICE)
    return set_error("LDPCdecoder_jobs takes device memory (host batches: LDPCdecoder_batch per code, or the transport-block chain)");
  if (n_jobs == 0)
    return ensure_ready();
  Device *d = device_of_pointer(jobs[0].llr);
  if (!d)
    return -1;
  UseDevice use(*d);
  return dec_jobs_enqueue(jobs, n_jobs, n_iter, static_cast<hipStream_t>(stream));
}
