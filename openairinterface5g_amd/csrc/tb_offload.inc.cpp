/*
 * tb_offload.inc.cpp -- the per-segment entry points of the reference's SECOND plugin slot, `ldpc_interface_offload`
 * (loaded as libldpc_t2.so when --ldpc-offload-enable is given: openair1/PHY/INIT/nr_init.c:138-139).  That slot has the
 * same four symbols as the CPU library but different semantics (nrLDPC_decoder/nrLDPC_decoder_offload.c:1036-1140): the
 * back end does rate (de)matching, (de)interleaving and HARQ combining itself and keeps the soft buffers in ITS memory.
 *
 *   decoder (caller nr_ulsch_decoding.c:225-268): p_llr = the segment's E received LLRs as int8, in transmission order;
 *     p_decParams carries BG, Z, R, numMaxIter and the offload-only fields E, Qm, rv, F, setCombIn; the soft buffer of
 *     segment r of ULSCH ulsch_id lives on the device (offload.c:546-547 addresses it by ulsch_id and r, not by harq_pid);
 *     setCombIn = 0 starts it afresh, 1 combines into it.  p_out = the K/8 decoded bytes.  Returns the passes run, < 0
 *     on a hard error (the one slot whose callers test for that, nr_ulsch_decoding.c:269).
 *   encoder (caller nr_dlsch_coding.c:366-383): input[0] = the segment (K - F bits, the fillers are added here),
 *     output[0] = E rate-matched and interleaved bits, one per byte (offload.c:608-623 unpacks them that way).
 *
 * What the T2 card computes inside is not in the reference (DPDK bbdev PMD + hardware).  This implementation gives the
 * offload slot the arithmetic of the reference's own CPU chain, so that both slots of a gNB produce the same bits:
 * de-interleaving (nr_rate_matching.c:310-388), de-matching with int16 accumulation over the full circular buffer
 * n_cb = 66 Zc / 50 Zc (:507-603; offload.c:1049), int16 -> int8 saturation with fillers at +127
 * (nr_ulsch_decoding.c:195-210), the flooding min-sum decoder in parity-check stop mode (the card's ITERATION_STOP,
 * offload.c:537; the CRC length is not available here, decParams->E holds the rate-matching E).  Kernels: the chain's
 * (tb_chain.hip) and the decoder's latency shape, one workgroup each.
 */
namespace {

struct OffloadHarq { /* per logical device: ulsch_id -> 64 soft buffers of OFFLOAD_HARQ_STRIDE int16 (device memory) */
  std::mutex mu;
  int16_t *buf[256] = {nullptr};
};
constexpr size_t OFFLOAD_HARQ_STRIDE = 66 * 384; /* LDPC_MAX_CB_SIZE of the card's layout, in soft values */
constexpr int OFFLOAD_MAX_SEG = 64;              /* offload.c:546: ulsch_id * 64 * LDPC_MAX_CB_SIZE + r * LDPC_MAX_CB_SIZE */

OffloadHarq &offload_harq()
{
  static OffloadHarq *h = new OffloadHarq[NRLDPC_HIP_MAX_DEVICES]; /* kept like every other device object (LDPCshutdown) */
  return h[cur_dev_index()];
}

/* `s`: the calling thread's stream.  The zero fill of a new ULSCH's buffers runs there and is waited for under the lock (no
 * null-stream call: that would wait for every blocking stream of the process, the resident servers' included) */
int16_t *offload_soft_buffer(uint8_t ulsch_id, uint8_t r, hipStream_t s)
{
  OffloadHarq &h = offload_harq();
  std::lock_guard<std::mutex> lk(h.mu);
  if (!h.buf[ulsch_id]) {
    void *p = nullptr;
    const size_t bytes = (size_t)OFFLOAD_MAX_SEG * OFFLOAD_HARQ_STRIDE * sizeof(int16_t);
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemsetAsync(p, 0, bytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      set_error("offload: soft buffer allocation failed");
      return nullptr;
    }
    h.buf[ulsch_id] = static_cast<int16_t *>(p);
  }
  return h.buf[ulsch_id] + (size_t)r * OFFLOAD_HARQ_STRIDE;
}

int offload_begin(TbCtx &c, hipStream_t &s)
{
  if (!c.own) {
    HIP_TRY(hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c.uploaded, hipEventDisableTiming));
  }
  if (c.last && c.last != c.own)
    HIP_TRY(hipStreamSynchronize(c.last));
  c.last = s = c.own;
  return tb_wait_upload(c);
}

int offload_decode(const t_nrLDPC_dec_params *p, uint8_t ulsch_id, uint8_t r, const int8_t *p_llr, int8_t *p_out)
{
  if (r >= OFFLOAD_MAX_SEG)
    return set_error("offload decoder: segment number above 63");
  if (!(p->Qm == 2 || p->Qm == 4 || p->Qm == 6 || p->Qm == 8) || p->E <= 0 || p->E % p->Qm || p->rv > 3)
    return set_error("offload decoder: invalid E / Qm / rv");
  const CodeEntry *ce = get_code(p->BG, p->Z, p->R);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  const uint32_t Z = (uint32_t)hc.Z, K = (uint32_t)hc.kb_full * Z, E = (uint32_t)p->E;
  if (p->F >= K)
    return set_error("offload decoder: invalid filler count");
  nr_hip_rm_t rm;
  if (nr_hip_rate_match_geometry(0, p->BG, Z, 1, p->F, K, p->rv, E, &rm) != 0) /* n_cb = N: no LBRM on this slot (offload.c:1049) */
    return set_error("offload decoder: invalid rate-matching parameters");
  TbCtx &c = tls_tb;
  hipStream_t s;
  if (offload_begin(c, s) != 0)
    return -1;
  int16_t *w = offload_soft_buffer(ulsch_id, r, s);
  if (!w)
    return -1;
  /* device scratch: [decoder input | decoded bits]; staging: [jobs | E int16 in] and [pass count | bits] back */
  /* (the pass count sits right in front of the decoded bits: one copy brings both back) */
  const size_t o_l = 0, o_iter = align_up((size_t)hc.num_llr, 16), o_c = o_iter + 16,
               scratch_top = o_c + align_up((size_t)out_bytes_of(hc, 0), 16);
  const size_t o_seg = 0, o_dec = align_up(sizeof(tb_rx_seg_job), 16), jobs_bytes = o_dec + align_up(sizeof(ldpc_dec_job), 16),
               o_in = jobs_bytes, up_bytes = o_in + (size_t)E * 2;
  const size_t kbytes = (K + 7) / 8;
  if (c.scratch.ensure(scratch_top) != 0 || c.jobs_h.ensure(up_bytes) != 0 || c.jobs_d.ensure(up_bytes) != 0 ||
      c.small_h.ensure(16 + kbytes) != 0)
    return -1;
  tb_rx_seg_job j;
  memset(&j, 0, sizeof(j));
  j.llr_off = 0; j.harq_off = 0; j.l_off = o_l;
  j.E = E; j.Qm = p->Qm; j.Ncb = rm.Ncb; j.Foffset = rm.Foffset; j.Fin = rm.Fin; j.V = rm.V; j.rank0 = rm.rank0;
  j.clear = p->setCombIn ? 0 : 1;
  j.K = K; j.F = p->F; j.Z = Z; j.num_llr = (uint32_t)hc.num_llr;
  j.c_off = o_c;
  ldpc_dec_job dj;
  memset(&dj, 0, sizeof(dj));
  dj.code = hc.f_ok ? ce->dev_lat : ce->dev;
  dj.llr_off = o_l; dj.out_off = o_c;
  dj.num_max_iter = p->numMaxIter;
  dj.iter_idx = 0; dj.abort_idx = -1; dj.seg_idx = -1;
  memcpy(c.jobs_h.p + o_seg, &j, sizeof(j));
  memcpy(c.jobs_h.p + o_dec, &dj, sizeof(dj));
  int16_t *in16 = reinterpret_cast<int16_t *>(c.jobs_h.p + o_in);
  for (uint32_t i = 0; i < E; i++)
    in16[i] = p_llr[i];
  if (tb_upload_jobs(c, c.jobs_d.p, up_bytes, s) != 0)
    return -1;
  HIP_TRY(tb_launch_rx_dematch(reinterpret_cast<const tb_rx_seg_job *>(c.jobs_d.p + o_seg), 1, tb_rx_lds_elems(E, rm.Fin, rm.Ncb),
                               reinterpret_cast<const int16_t *>(c.jobs_d.p + o_in), w, reinterpret_cast<int8_t *>(c.scratch.p), s, 1));
  ldpc_dec_args da;
  memset(&da, 0, sizeof(da));
  da.llr = reinterpret_cast<const int8_t *>(c.scratch.p);
  da.out = reinterpret_cast<int8_t *>(c.scratch.p);
  da.n_iter = reinterpret_cast<int32_t *>(c.scratch.p + o_iter);
  da.out_mode = 0;
  da.use_crc = 0;
  da.jobs = reinterpret_cast<const ldpc_dec_job *>(c.jobs_d.p + o_dec);
  if (hc.f_ok)
    HIP_TRY(ldpc_launch_dec_fast_jobs(da, ce->host_lat.f_n_threads, ce->host_lat.f_lds_total, 1, s));
  else
    HIP_TRY(ldpc_launch_dec_generic_jobs(da, hc.n_threads, hc.lds_total, 1, s));
  HIP_TRY(hipMemcpyAsync(c.small_h.p, c.scratch.p + o_iter, 16 + kbytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  c.pending = false;
  memcpy(p_out, c.small_h.p + 16, kbytes);
  int32_t n_iter;
  memcpy(&n_iter, c.small_h.p, 4);
  return n_iter;
}

int offload_encode(const uint8_t *in, uint8_t *out, const encoder_implemparams_t *ip)
{
  if (!ldpc_enc_is_packed())
    return set_error("offload encoder: needs the bit-packed encoder kernel (NRLDPC_HIP_ENC_KERNEL unset)");
  if (!(ip->Qm == 2 || ip->Qm == 4 || ip->Qm == 6 || ip->Qm == 8) || ip->E == 0 || ip->E % ip->Qm || ip->rv > 3)
    return set_error("offload encoder: invalid E / Qm / rv");
  const CodeEntry *ce = get_code(ip->BG, (int)ip->Zc, ip->BG == 1 ? 13 : 15);
  if (!ce)
    return -1;
  const ldpc_code_desc_t &hc = ce->host;
  const uint32_t Z = (uint32_t)hc.Z, K = (uint32_t)hc.kb_full * Z;
  if (ip->K != K || ip->F >= K || ((K - ip->F) & 7))
    return set_error("offload encoder: K / F do not fit the code (K - F must be whole bytes)");
  /* information columns that enter the parity: the caller's Kb when it gives one (nr_segmentation's return value,
   * nr_dlsch_coding.c:333), all of them otherwise */
  const uint32_t Kb = (ip->Kb >= 6 && ip->Kb <= (uint32_t)hc.kb_full) ? ip->Kb : (uint32_t)hc.kb_full;
  nr_hip_rm_t rm;
  if (nr_hip_rate_match_geometry(0, ip->BG, Z, 1, ip->F, K, ip->rv, ip->E, &rm) != 0)
    return set_error("offload encoder: invalid rate-matching parameters");
  TbCtx &c = tls_tb;
  hipStream_t s;
  if (offload_begin(c, s) != 0)
    return -1;
  const uint32_t kprime = K - ip->F, in_bytes = kprime / 8;
  const size_t o_seg = 0, o_enc = align_up(sizeof(tb_tx_seg_job), 16), o_acc = o_enc + align_up(sizeof(ldpc_enc_job), 16),
               jobs_bytes = o_acc + 16, o_in = jobs_bytes, up_bytes = o_in + align_up((size_t)in_bytes + 16, 16);
  if (c.jobs_h.ensure(up_bytes) != 0 || c.jobs_d.ensure(up_bytes) != 0 || c.io_coded.ensure(ip->E) != 0 || c.scratch.ensure(64) != 0)
    return -1;
  tb_tx_seg_job j;
  memset(&j, 0, sizeof(j));
  j.b_off = 0;                      /* relative to the `scratch` argument below = the uploaded segment */
  j.out_off = 0;
  j.r = 0; j.C = 1; j.Kprime = kprime; j.L = 0; j.K = K; /* the caller has attached the CB CRC already */
  j.E = ip->E; j.Qm = ip->Qm;
  j.Foffset = rm.Foffset; j.Fin = rm.Fin; j.V = rm.V; j.rank0 = rm.rank0;
  j.tb = 0; j.crc_len = 0;
  ldpc_enc_job e;
  memset(&e, 0, sizeof(e));
  e.code = ce->dev; e.Kb = (int32_t)Kb;
  memset(c.jobs_h.p, 0, up_bytes);
  memcpy(c.jobs_h.p + o_seg, &j, sizeof(j));
  memcpy(c.jobs_h.p + o_enc, &e, sizeof(e));
  memcpy(c.jobs_h.p + o_in, in, in_bytes);
  if (tb_upload_jobs(c, c.jobs_d.p, up_bytes, s) != 0)
    return -1;
  int nthr, nlds;
  ldpc_enc_launch_shape(hc, &nthr, &nlds);
  HIP_TRY(tb_launch_tx_fused(reinterpret_cast<const tb_tx_seg_job *>(c.jobs_d.p + o_seg),
                             reinterpret_cast<const ldpc_enc_job *>(c.jobs_d.p + o_enc), 1, 512, nlds + TB_TX_FUSED_EXTRA_LDS,
                             c.jobs_d.p + o_in, c.io_coded.p, G().crc_pow[NR_HIP_CRC24_B],
                             reinterpret_cast<uint32_t *>(c.jobs_d.p + o_acc), s));
  /* (the caller's array never meets the runtime's copy functions: TbCtx::fin_copies) */
  if (c.coded_h.ensure(ip->E) != 0)
    return -1;
  HIP_TRY(hipMemcpyAsync(c.coded_h.p, c.io_coded.p, ip->E, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(out, c.coded_h.p, ip->E);
  c.pending = false;
  return 0;
}

} // namespace

extern "C" {

/* (not through the exported name: libldpc_hip_t2.so exports an LDPCinit of its own, and plugins are loaded RTLD_GLOBAL) */
int32_t nrLDPC_hip_offload_init(void) { return lib_init(); }

int32_t nrLDPC_hip_offload_decoder(t_nrLDPC_dec_params *p_decParams, uint8_t harq_pid, uint8_t ulsch_id, uint8_t C, int8_t *p_llr,
                                   int8_t *p_out, t_nrLDPC_time_stats *p_profiler, decode_abort_t *ab)
{
  (void)harq_pid; /* the card's soft-buffer address does not depend on it either (offload.c:546-547) */
  (void)ab;       /* NULL at the one call site (nr_ulsch_decoding.c:268) */
  if (!p_decParams || !p_llr || !p_out) {
    set_error("null argument");
    return -1;
  }
  if (ensure_ready() != 0)
    return -1;
  UseDevice use(g.dev[0]); /* one card: the soft buffers live where the first call put them */
  meter_start(p_profiler ? &p_profiler->total : nullptr);
  const int rc = offload_decode(p_decParams, ulsch_id, C, p_llr, p_out);
  meter_stop(p_profiler ? &p_profiler->total : nullptr);
  return rc;
}

int32_t nrLDPC_hip_offload_encoder(uint8_t **input, uint8_t **output, encoder_implemparams_t *impp)
{
  if (!input || !output || !impp || !input[0] || !output[0]) {
    set_error("null argument");
    return -1;
  }
  if (ensure_ready() != 0)
    return -1;
  UseDevice use(g.dev[0]);
  return offload_encode(input[0], output[0], impp);
}

} /* extern "C" */
