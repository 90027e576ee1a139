"""-m gpu: transport-block chain on the GPU (TB CRC, segmentation, encode, rate matching, interleaving; and back:
de-interleaving, rate de-matching with HARQ combining, pack, decode with CRC stop, reassembly, TB CRC) vs the same
chain composed from the oracle pieces the way the reference composes its functions."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def valid_tbs(target_bits, BG):
    """Smallest A >= target whose segmentation is byte aligned, as every 38.214 TBS is (the reference copies bytes)."""
    A = (target_bits + 7) // 8 * 8
    while True:
        B = O.len_with_crc(1, A)
        s = O.segmentation(None, B, BG)
        L = 24 if s["C"] > 1 else 0
        if (s["K"] - s["F"] - L) % 8 == 0 and s["F"] % 8 == 0 and (B + L * s["C"]) % s["C"] == 0:
            return A
        A += 8


def make_tbs():
    mk = lambda bits, G, BG, Qm, Nl, rv=0, lbrm=0: dict(A=valid_tbs(bits, BG), G=G, BG=BG, Qm=Qm, Nl=Nl, rv=rv, tbslbrm=lbrm)
    return [
        mk(9600, 2 * 12 * 100 * 14, 1, 4, 1),            # 2 segments, 16QAM
        mk(32000, 60000, 1, 6, 2),                       # 4 segments Zc=384, 64QAM, 2 layers
        mk(5000, 14400, 2, 2, 1),                        # BG2, 2 segments
        mk(800, 2400, 2, 2, 1),                          # single BG2 segment, CRC16
        mk(3824, 12000, 1, 4, 1),                        # largest CRC16 block
        mk(3832, 9600, 1, 8, 1),                         # smallest CRC24A block, 256QAM
        mk(100000, 8 * 4 * 9000, 1, 8, 4, rv=0),         # 12 segments, 4 layers
        mk(20000, 26400, 1, 2, 1, rv=2),                 # rv 2 start
        mk(20000, 80000, 1, 4, 1, rv=3),                 # repetition (E > Ncb)
        mk(30000, 54000, 1, 6, 1, rv=1, lbrm=24000),            # limited-buffer rate matching (Ncb = 9000 < N)
        mk(264, 1200, 2, 2, 1),                          # Kb = 8 (B in (192, 560])
        mk(24, 240, 2, 2, 1),                            # Kb = 6
        mk(600, 1800, 2, 2, 1),                          # Kb = 9 / 10 boundary
    ]


def test_dlsch_encode_matches_reference_chain(hip):
    rng = np.random.default_rng(1)
    tbs = make_tbs()
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    coded = hip.ldpc.dlsch_encode_host(tbs, pays)
    for t, p, f in zip(tbs, pays, coded):
        ref = O.dlsch_encode(t, p)
        assert f.size == t["G"] == ref.size
        assert np.array_equal(f, ref), t


def test_reference_compiled_code_words_through_both_chains(hip):
    """tests/golden/ref_encoder.npz, transport blocks: every segment's code word comes from the reference-COMPILED encoder
    (oracle/_ref).  TX: the DL-SCH chain's output (rv 0, QPSK, every transmittable bit once) equals the one derived from
    those code words.  RX: that output as noiseless LLRs comes back through the UL-SCH chain -- fused segment kernel and the
    four-launch path's kernels alike -- as the payload, ACK, within three passes."""
    from common import load_ref_transport_blocks
    cases = list(load_ref_transport_blocks())
    tbs = [c["tb"] for c in cases]
    coded = hip.ldpc.dlsch_encode_host(tbs, [c["payload"] for c in cases])
    for c, f in zip(cases, coded):
        assert f.size == c["coded"].size and np.array_equal(f, c["coded"]), c["tb"]
    stride = hip.ldpc.HARQ_STRIDE
    llrs = [(10 * (1 - 2 * c["coded"].astype(np.int16))).astype(np.int16) for c in cases]
    for fused in ("1", "0"):
        os.environ["NRLDPC_HIP_TB_FUSED"] = fused
        try:
            harq = np.zeros((sum(c["C"] for c in cases), stride), np.int16)
            for t in tbs:
                t.pop("llrLen", None)
                t["round"] = 0
            out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, llrs, harq, numMaxIter=8)
        finally:
            os.environ.pop("NRLDPC_HIP_TB_FUSED", None)
        for c, o, a, it in zip(cases, out, ack, itm):
            assert a and it <= 3 and np.array_equal(o, c["payload"]), (fused, c["tb"], int(it))


def test_dlsch_encode_random_sweep(hip):
    """Seeded random transport blocks (size, base graph, modulation, layers, rv, LBRM, code rate from 0.15 to 0.95 incl.
    repetition) in ONE heterogeneous call: every output bit equals the oracle chain's."""
    rng = np.random.default_rng(20260927)
    tbs = []
    for _ in range(48):
        bits = int(np.exp(rng.uniform(np.log(24), np.log(120000))))
        BG = 2 if bits <= 292 else (int(rng.integers(1, 3)) if bits <= 30000 else 1)
        Qm, Nl = int(rng.choice([2, 4, 6, 8])), int(rng.integers(1, 5))
        A = valid_tbs(bits, BG)
        rate = rng.uniform(0.15, 0.95)
        G = max(1, int(A / rate) // (Qm * Nl)) * Qm * Nl
        C = O.segmentation(None, O.len_with_crc(1, A), BG)["C"]
        G = max(G, C * Qm * Nl * 4)
        lbrm = int(rng.choice([0, 0, 2 * A, 3 * A]))          # Nref >= K - 2Zc (the reference's own contract)
        tbs.append(dict(A=A, G=G, BG=BG, Qm=Qm, Nl=Nl, rv=int(rng.integers(0, 4)), tbslbrm=lbrm))
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    coded = hip.ldpc.dlsch_encode_host(tbs, pays)
    for t, p, f in zip(tbs, pays, coded):
        ref = O.dlsch_encode(t, p)
        assert f.size == t["G"] == ref.size and np.array_equal(f, ref), t


def test_ulsch_decode_matches_reference_chain_with_harq(hip):
    """Two HARQ rounds (rv 0 then rv 2) at an SNR where the first round fails for some blocks: payload, ACK, pass
    counts, soft buffers and the llrLen state must equal the oracle chain's after each round."""
    rng = np.random.default_rng(2)
    tbs = [t for t in make_tbs() if t["rv"] == 0 and t["tbslbrm"] == 0][:7]
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    harq_gpu = np.zeros((sum(segs), stride), np.int16)
    harq_ref = [[np.zeros(stride, np.int16) for _ in range(c)] for c in segs]
    state_ref = [0] * len(tbs)
    for rnd, rv, sigma in ((0, 0, 9.0), (1, 2, 6.0)):
        llrs = []
        for t, p in zip(tbs, pays):
            t["rv"], t["round"] = rv, rnd
            f = O.dlsch_encode(t, p)
            llrs.append(np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16))
        out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, llrs, harq_gpu, numMaxIter=8)
        row = 0
        for i, t in enumerate(tbs):
            p_ref, ack_ref, its, state_ref[i] = O.ulsch_decode(t, llrs[i], harq_ref[i], 8, rnd, state_ref[i])
            assert bool(ack[i]) == ack_ref and itm[i] == max(its), (rnd, t, its, int(itm[i]))
            assert t["llrLen"] == state_ref[i]
            if ack_ref:
                assert np.array_equal(out[i], p_ref) and np.array_equal(out[i], pays[i])
            for r in range(segs[i]):
                assert np.array_equal(harq_gpu[row + r], harq_ref[i][r]), (rnd, i, r)
            row += segs[i]
    assert ack.all()                                          # after combining every block decodes


def test_harq_rounds_on_dirty_soft_buffers_with_lbrm_and_repetition(hip):
    """Soft buffers that are NOT zero when a round starts (the caller reuses HARQ memory): limited-buffer rate matching,
    repetition (several laps over the circular buffer) and plain blocks.  Round 0 clears exactly Ncb entries before it
    accumulates (nr_rate_matching.c:554-555) -- garbage below Ncb must not leak; rounds 1 and 2 add to whatever the buffer
    holds, stale values included, also beyond Ncb.  Every soft-buffer value, ACK and pass count equals the oracle
    chain's.  Finally a fresh all-garbage buffer entered at round 1."""
    rng = np.random.default_rng(31)
    mk = lambda bits, G, BG, Qm, Nl, rv=0, lbrm=0: dict(A=valid_tbs(bits, BG), G=G, BG=BG, Qm=Qm, Nl=Nl, rv=rv, tbslbrm=lbrm)
    tbs = [mk(30000, 54000, 1, 6, 1, rv=0, lbrm=24000), mk(20000, 80000, 1, 4, 1), mk(9600, 33600, 1, 8, 1, lbrm=9000),
           mk(5000, 14400, 2, 2, 1, lbrm=6000), mk(3000, 30000, 2, 4, 2), mk(20000, 26400, 1, 2, 1)]
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE

    def ncb_of(t, c):
        s = O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])
        N = (66 if t["BG"] == 1 else 50) * s["Z"]
        return N if not t["tbslbrm"] else min(N, 3 * t["tbslbrm"] // (2 * c))

    def run_rounds(rounds, harq0):
        harq_gpu = np.concatenate([np.stack(h) for h in harq0]).copy()
        harq_ref = [[x.copy() for x in h] for h in harq0]
        state = [0] * len(tbs)
        for t in tbs:
            t.pop("llrLen", None)
        for rnd, rv, sigma in rounds:
            llrs = []
            for t, p in zip(tbs, pays):
                t["rv"], t["round"] = rv, rnd
                f = O.dlsch_encode(t, p)
                llrs.append(np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16))
            out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, llrs, harq_gpu, numMaxIter=8)
            row = 0
            for i, t in enumerate(tbs):
                p_ref, ack_ref, its, state[i] = O.ulsch_decode(t, llrs[i], harq_ref[i], 8, rnd, state[i], vec=True)
                assert bool(ack[i]) == ack_ref and itm[i] == min(max(its), 9), (rnd, i, its, int(itm[i]))
                if ack_ref:
                    assert np.array_equal(out[i], p_ref)
                for r in range(segs[i]):
                    assert np.array_equal(harq_gpu[row + r], harq_ref[i][r]), (rnd, i, r)
                row += segs[i]
        return ack

    # garbage below Ncb only (what a first round must wipe); zeros above, as the reference's calloc'ed buffers have
    dirty = []
    for t, c in zip(tbs, segs):
        n = ncb_of(t, c)
        dirty.append([np.concatenate([rng.integers(-3000, 3000, n).astype(np.int16), np.zeros(stride - n, np.int16)]) for _ in range(c)])
    ack = run_rounds(((0, 0, 10.0), (1, 2, 8.0), (2, 3, 5.0)), dirty)
    assert ack.sum() >= 4                                     # (parity with the oracle chain is asserted inside, per round)
    # a buffer that is garbage everywhere, entered at round 1: nothing is cleared, stale values beyond Ncb reach the decoder
    junk = [[rng.integers(-40, 40, stride).astype(np.int16) for _ in range(c)] for c in segs]
    run_rounds(((1, 0, 5.0), (2, 1, 5.0)), junk)


def test_ulsch_decode_random_sweep(hip):
    """Seeded random transport blocks (both base graphs, all modulations, rv 0..3, LBRM, some too noisy to decode) in one
    heterogeneous call: ACK, pass counts, payload, soft buffers and llrLen equal the oracle chain's."""
    rng = np.random.default_rng(777)
    tbs = []
    for _ in range(12):
        bits = int(np.exp(rng.uniform(np.log(24), np.log(20000))))
        BG = 2 if bits <= 292 else int(rng.integers(1, 3))
        Qm, Nl = int(rng.choice([2, 4, 6, 8])), int(rng.integers(1, 3))
        A = valid_tbs(bits, BG)
        C = O.segmentation(None, O.len_with_crc(1, A), BG)["C"]
        G = max(int(A / rng.uniform(0.2, 0.8)) // (Qm * Nl), C * 4) * Qm * Nl
        tbs.append(dict(A=A, G=G, BG=BG, Qm=Qm, Nl=Nl, rv=int(rng.choice([0, 0, 0, 1, 2, 3])), round=0,
                        tbslbrm=int(rng.choice([0, 0, 3 * A]))))
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    harq_gpu = np.zeros((sum(segs), stride), np.int16)
    llrs = []
    for t, p in zip(tbs, pays):
        f = O.dlsch_encode(t, p)
        sigma = rng.choice([3.0, 5.0, 9.0])
        llrs.append(np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16))
    out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, llrs, harq_gpu, numMaxIter=6)
    row, n_ack = 0, 0
    for i, t in enumerate(tbs):
        harq_ref = [np.zeros(stride, np.int16) for _ in range(segs[i])]
        p_ref, ack_ref, its, state = O.ulsch_decode(t, llrs[i], harq_ref, 6, 0, 0)
        assert bool(ack[i]) == ack_ref and itm[i] == max(its), (t, its, int(itm[i]))
        assert t["llrLen"] == state
        if ack_ref:
            # (a rv != 0 transmission that carries no systematic bits can "decode" to the all-zero word, whose CRC is
            # zero: the reference ACKs it, so must we -- hence the comparison is with the oracle, not with the payload)
            assert np.array_equal(out[i], p_ref), t
            n_ack += int(np.array_equal(p_ref, pays[i]))
        for r in range(segs[i]):
            assert np.array_equal(harq_gpu[row + r], harq_ref[r]), (i, r)
        row += segs[i]
    assert n_ack >= 4


def test_full_size_slot_64_transport_blocks(hip):
    """BASELINE configs[3] and [4] at full size: 64 transport blocks of ~213 kbit (273 PRB, 64QAM, 26 segments each,
    1664 code blocks) through the DL-SCH chain and back through the UL-SCH chain on device-resident buffers; EVERY
    coded bit, payload byte, ACK, pass count, llrLen and soft-buffer value is compared with the oracle chain."""
    import torch
    m = hip.ldpc
    rng = np.random.default_rng(64)
    A = valid_tbs(213176, 1)
    G = (12 * 13 - 6) * 273 * 6
    n_tb = 64
    tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n_tb)]
    po, co, ho, segs = m.tb_layout(tbs)
    pay_h = np.zeros(int(po[-1]) + 16, np.uint8)
    for i in range(n_tb):
        pay_h[po[i]:po[i] + A // 8] = rng.integers(0, 256, A // 8, dtype=np.uint8)
    payload = torch.from_numpy(pay_h).cuda()
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    m.dlsch_encode_device(tbs, payload, coded)
    coded_h = coded.cpu().numpy()
    refs = [O.dlsch_encode(tbs[i], pay_h[po[i]:po[i] + A // 8]) for i in range(n_tb)]
    for i in range(n_tb):
        assert np.array_equal(coded_h[co[i]:co[i] + G], refs[i]), i
    # a channel where a few blocks need more than the minimum three passes and TB 7 is lost
    llr_h = np.zeros(int(co[-1]) + 16, np.int16)
    for i in range(n_tb):
        sigma = 40.0 if i == 7 else 2.9
        llr_h[co[i]:co[i] + G] = np.clip(np.round((1 - 2 * refs[i].astype(np.float64)) * 8 + sigma * rng.standard_normal(G)),
                                         -200, 200).astype(np.int16)
    llr = torch.from_numpy(llr_h).cuda()
    harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
    pay_out = torch.zeros_like(payload)
    ack = torch.zeros(n_tb, dtype=torch.uint8, device="cuda")
    itm = torch.zeros(n_tb, dtype=torch.int32, device="cuda")
    m.ulsch_decode_device(tbs, llr, harq, pay_out, ack, itm)
    torch.cuda.synchronize()
    ack_h, itm_h, out_h, harq_h = ack.cpu().numpy(), itm.cpu().numpy(), pay_out.cpu().numpy(), harq.cpu().numpy()
    stride = m.HARQ_STRIDE
    n_ack = 0
    for i in range(n_tb):
        harq_ref = [np.zeros(stride, np.int16) for _ in range(segs[i])]
        p_ref, ack_ref, its, state = O.ulsch_decode(tbs[i], llr_h[co[i]:co[i] + G], harq_ref, 8, 0, 0, vec=True)
        assert bool(ack_h[i]) == ack_ref and itm_h[i] == max(its), (i, its, int(itm_h[i]))
        assert tbs[i]["llrLen"] == state
        if ack_ref:
            n_ack += 1
            assert np.array_equal(out_h[po[i]:po[i] + A // 8], p_ref) and np.array_equal(p_ref, pay_h[po[i]:po[i] + A // 8])
        for r in range(segs[i]):
            assert np.array_equal(harq_h[ho[i] + r * stride:ho[i] + (r + 1) * stride], harq_ref[r]), (i, r)
    assert n_ack == n_tb - 1 and not ack_h[7] and itm_h[7] == 9


def test_repeated_descriptors_reuse_the_plan(hip):
    """The library keeps the job lists of the last call and reuses them when the descriptors repeat (a scheduler that
    keeps an allocation): calls 2..4 with the same descriptors but new data must still match the oracle chain."""
    rng = np.random.default_rng(11)
    tbs = make_tbs()[:6]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    for call in range(4):
        pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
        tx = [dict(t) for t in tbs]                                  # identical descriptors every call
        coded = hip.ldpc.dlsch_encode_host(tx, pays)
        refs = [O.dlsch_encode(t, p) for t, p in zip(tbs, pays)]
        for f, ref in zip(coded, refs):
            assert np.array_equal(f, ref), call
        llrs = [np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + 4.0 * rng.standard_normal(f.size)), -200, 200).astype(np.int16)
                for f in refs]
        rx = [dict(t, round=0, llrLen=0) for t in tbs]               # identical on arrival every call
        harq_gpu = np.zeros((sum(segs), stride), np.int16)
        out, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq_gpu, numMaxIter=8)
        row = 0
        for i, t in enumerate(tbs):
            harq_ref = [np.zeros(stride, np.int16) for _ in range(segs[i])]
            p_ref, ack_ref, its, state = O.ulsch_decode(dict(t), llrs[i], harq_ref, 8, 0, 0, vec=True)
            assert bool(ack[i]) == ack_ref and itm[i] == max(its) and rx[i]["llrLen"] == state, (call, i)
            if ack_ref:
                assert np.array_equal(out[i], p_ref), (call, i)
            for r in range(segs[i]):
                assert np.array_equal(harq_gpu[row + r], harq_ref[r]), (call, i, r)
            row += segs[i]


def test_small_transport_blocks_share_workgroups(hip):
    """A batch with enough segments to fill the GPU: the small single-segment transport blocks of the same code, cap and
    CRC length are decoded several to a workgroup (tb_api.inc.cpp groups them, ldpc_dec_fast_mblock.h decodes a group --
    side by side for Zc % 4 == 0, four byte-interleaved for the other lifting sizes), the rest one per workgroup as
    before.  All 93 sizes of 38.214 Table 5.1.3.2-1 at two rates, six copies each with their own payload and noise level
    (clean, marginal, hopeless): ACK, pass count, payload, soft buffer and llrLen of every block equal the oracle
    chain's."""
    tbs_table = [24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 208, 224,
                 240, 256, 272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576, 608, 640, 672, 704,
                 736, 768, 808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480,
                 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216, 2280, 2408, 2472, 2536, 2600, 2664, 2728,
                 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752, 3824]
    rng = np.random.default_rng(214)
    tbs, copies = [], 6
    for A in tbs_table:
        for rate in (0.3, 0.75):
            BG = 2 if (A <= 292 or rate <= 0.25 or (A <= 3824 and rate <= 0.67)) else 1
            Qm = int(rng.choice([2, 4, 6]))
            G = max(int(A / rate) // Qm, 4) * Qm
            for _ in range(copies):
                tbs.append(dict(A=A, G=G, BG=BG, Qm=Qm, Nl=1, rv=0, tbslbrm=0))
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    coded = hip.ldpc.dlsch_encode_host(tbs, pays)
    llrs = []
    for k, f in enumerate(coded):
        sigma = (1.5, 3.0, 4.5, 2.2, 6.0, 12.0)[k % copies]
        llrs.append(np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16))
    rx = [dict(t, round=0, llrLen=0) for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    harq_gpu = np.zeros((len(tbs), stride), np.int16)
    os.environ["NRLDPC_HIP_TB_MULTI"] = "2" # share whatever the batch size (the default waits for 64 small segments per CU)
    try:
        out, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq_gpu, numMaxIter=8)
    finally:
        del os.environ["NRLDPC_HIP_TB_MULTI"]
    # the same call under the other launch plans -- the default (one workgroup per segment, launches cut by workgroup shape,
    # last rounds filled with segments of the smaller shapes), without the filling, without the classes, unfused: the plan
    # decides where a segment runs, never what comes out
    for env in ({}, {"NRLDPC_HIP_TB_FILL": "0"}, {"NRLDPC_HIP_TB_CLASSES": "0"}, {"NRLDPC_HIP_TB_FUSED": "0"},
                {"NRLDPC_HIP_TB_FUSED": "0", "NRLDPC_HIP_TB_FILL": "0"}):
        os.environ.update(env)
        try:
            rx2 = [dict(t, round=0, llrLen=0) for t in tbs]
            harq2 = np.zeros((len(tbs), stride), np.int16)
            out2, ack2, itm2 = hip.ldpc.ulsch_decode_host(rx2, llrs, harq2, numMaxIter=8)
        finally:
            for k in env:
                del os.environ[k]
        assert np.array_equal(ack2, ack) and np.array_equal(itm2, itm) and np.array_equal(harq2, harq_gpu), env
        assert all(np.array_equal(a, b) for a, b in zip(out2, out)), env
        assert [t["llrLen"] for t in rx2] == [t["llrLen"] for t in rx], env
    n_ack = 0
    for i, t in enumerate(tbs):
        harq_ref = [np.zeros(stride, np.int16)]
        p_ref, ack_ref, its, state = O.ulsch_decode(dict(t), llrs[i], harq_ref, 8, 0, 0, vec=True)
        assert bool(ack[i]) == ack_ref and itm[i] == min(max(its), 9) and rx[i]["llrLen"] == state, (i, t, its, int(itm[i]), bool(ack[i]))
        if ack_ref:
            assert np.array_equal(out[i], p_ref), t
            n_ack += 1
        else:
            assert not out[i].any(), t               # a lost block delivers zeros
        assert np.array_equal(harq_gpu[i], harq_ref[0]), t
    assert 0.3 * len(tbs) < n_ack < 0.9 * len(tbs), n_ack


def test_every_small_tbs_of_38214(hip):
    """All 93 transport block sizes of 38.214 Table 5.1.3.2-1 (TBS <= 3824, the CRC16 / single-segment regime where
    Kb, Zc and the filler count change from entry to entry), base graph by the 38.212 7.2.2 rule for two code rates:
    the coded bits of the whole heterogeneous batch equal the oracle chain's."""
    tbs_table = [24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 208, 224,
                 240, 256, 272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576, 608, 640, 672, 704,
                 736, 768, 808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480,
                 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216, 2280, 2408, 2472, 2536, 2600, 2664, 2728,
                 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752, 3824]
    assert len(tbs_table) == 93
    rng = np.random.default_rng(38214)
    tbs = []
    for A in tbs_table:
        for rate in (0.3, 0.75):
            BG = 2 if (A <= 292 or rate <= 0.25 or (A <= 3824 and rate <= 0.67)) else 1
            Qm = int(rng.choice([2, 4, 6, 8]))
            G = max(int(A / rate) // Qm, 4) * Qm
            tbs.append(dict(A=A, G=G, BG=BG, Qm=Qm, Nl=1, rv=int(rng.integers(0, 4)), tbslbrm=0))
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    coded = hip.ldpc.dlsch_encode_host(tbs, pays)
    for t, p, f in zip(tbs, pays, coded):
        assert np.array_equal(f, O.dlsch_encode(t, p)), t
    # and back: mild noise, one call for all 186 blocks; everything against the oracle chain
    llrs = [np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + 2.5 * rng.standard_normal(f.size)), -200, 200).astype(np.int16)
            for f in coded]
    rx = [dict(t, round=0, llrLen=0) for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    harq_gpu = np.zeros((len(tbs), stride), np.int16)
    out, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq_gpu, numMaxIter=8)
    n_good = 0
    for i, t in enumerate(tbs):
        harq_ref = [np.zeros(stride, np.int16)]
        p_ref, ack_ref, its, state = O.ulsch_decode(dict(t), llrs[i], harq_ref, 8, 0, 0, vec=True)
        assert bool(ack[i]) == ack_ref and itm[i] == max(its) and rx[i]["llrLen"] == state, (t, its, int(itm[i]))
        if ack_ref:
            assert np.array_equal(out[i], p_ref), t
            n_good += int(np.array_equal(p_ref, pays[i]))
        assert np.array_equal(harq_gpu[i], harq_ref[0]), t
    assert n_good > 90


def test_chain_calls_are_capturable_in_a_hip_graph(hip):
    """With device-resident buffers and repeated descriptors the two chain calls only enqueue kernels and memsets:
    capture them once in a HIP graph, replay with new payloads / LLRs, compare with the oracle chain."""
    import torch
    m = hip.ldpc
    rng = np.random.default_rng(5)
    tbs = [dict(t, round=0, llrLen=0) for t in make_tbs()[:5]]
    po, co, ho, segs = m.tb_layout(tbs)
    payload = torch.zeros(int(po[-1]) + 16, dtype=torch.uint8, device="cuda")
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    llr = torch.zeros(int(co[-1]) + 16, dtype=torch.int16, device="cuda")
    harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
    pay_out = torch.zeros_like(payload)
    ack = torch.zeros(len(tbs), dtype=torch.uint8, device="cuda")
    itm = torch.zeros(len(tbs), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        enc = m.PreparedTbBatch(tbs, payload, coded)
        dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
        for _ in range(3):                   # from the third call on both plans are cached: nothing is uploaded any more
            enc.encode()
            dec.decode()
    torch.cuda.synchronize()
    g_enc, g_dec = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_enc, stream=side):
        enc.encode()
    with torch.cuda.graph(g_dec, stream=side):
        dec.decode()
    for rep in range(2):
        pay_h = np.zeros(payload.numel(), np.uint8)
        for i, t in enumerate(tbs):
            pay_h[po[i]:po[i] + t["A"] // 8] = rng.integers(0, 256, t["A"] // 8, dtype=np.uint8)
        payload.copy_(torch.from_numpy(pay_h))
        g_enc.replay()
        torch.cuda.synchronize()
        coded_h = coded.cpu().numpy()
        for i, t in enumerate(tbs):
            assert np.array_equal(coded_h[co[i]:co[i] + t["G"]], O.dlsch_encode(t, pay_h[po[i]:po[i] + t["A"] // 8])), (rep, i)
        llr.copy_(((1 - 2 * coded.to(torch.int16)) * 20).to(torch.int16))
        harq.zero_()
        g_dec.replay()
        torch.cuda.synchronize()
        out_h = pay_out.cpu().numpy()
        if not ack.cpu().numpy().all():     # diagnostics: the same call issued eagerly
            a_g, i_g = ack.cpu().numpy().copy(), itm.cpu().numpy().copy()
            harq.zero_()
            with torch.cuda.stream(side):
                dec.decode()
            torch.cuda.synchronize()
            raise AssertionError(("graph", a_g, i_g, "eager", ack.cpu().numpy(), itm.cpu().numpy(), rep))
        for i, t in enumerate(tbs):
            assert np.array_equal(out_h[po[i]:po[i] + t["A"] // 8], pay_h[po[i]:po[i] + t["A"] // 8]), (rep, i)


def test_chain_calls_from_concurrent_threads(hip):
    """Four host threads, each with its own batches, call both chain entry points at the same time (every calling thread
    has its own stream, scratch and plan cache): all results equal the oracle chain's."""
    import threading
    all_tbs = make_tbs()
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(100 + k)
            tbs = [dict(t) for t in all_tbs[k::4]]
            for _ in range(3):
                pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
                coded = hip.ldpc.dlsch_encode_host(tbs, pays)
                for t, p, f in zip(tbs, pays, coded):
                    assert np.array_equal(f, O.dlsch_encode(t, p)), (k, t)
                rx = [dict(t, rv=0, tbslbrm=0, round=0, llrLen=0) for t in tbs]
                coded0 = [O.dlsch_encode(t, p) for t, p in zip(rx, pays)]
                llrs = [((1 - 2 * f.astype(np.int16)) * 20).astype(np.int16) for f in coded0]
                segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in rx]
                harq = np.zeros((sum(segs), hip.ldpc.HARQ_STRIDE), np.int16)
                out, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq)
                assert ack.all() and all(np.array_equal(o, p) for o, p in zip(out, pays)), k
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_encode_then_decode_round_trip_rv_and_lbrm(hip):
    rng = np.random.default_rng(3)
    tbs = make_tbs()
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    coded = hip.ldpc.dlsch_encode_host(tbs, pays)
    llrs = [((1 - 2 * f.astype(np.int16)) * 24 + rng.integers(-10, 11, f.size)).astype(np.int16) for f in coded]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    harq = np.zeros((sum(segs), hip.ldpc.HARQ_STRIDE), np.int16)
    for t in tbs:
        t["round"] = 0
    out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, llrs, harq)
    for i, t in enumerate(tbs):
        if t["rv"] in (0, 3) or t["G"] > 2 * t["A"]:           # self-decodable transmissions
            assert ack[i] and np.array_equal(out[i], pays[i]), (t, int(itm[i]))
    # a corrupted block is NACKed
    bad = [l.copy() for l in llrs]
    bad[1][:] = rng.integers(-5, 6, bad[1].size)
    harq[:] = 0
    out, ack, itm = hip.ldpc.ulsch_decode_host(tbs, bad, harq)
    assert not ack[1] and itm[1] == 9 and ack[0]


def test_largest_transport_block(hip):
    """The largest NR transport block (38.214: 1 277 992 bits = 152 segments of BG1 Zc = 384, 256QAM, 4 layers) next to
    the smallest (24 bits) in one call: coded bits, payloads, ACKs, pass counts and soft buffers against the oracle chain,
    first transmission at rv 0 and a retransmission at rv 3."""
    rng = np.random.default_rng(1277992)
    big = dict(A=1277992, G=8 * 4 * 48000, BG=1, Qm=8, Nl=4, rv=0, tbslbrm=0)
    tiny = dict(A=24, G=2 * 40, BG=2, Qm=2, Nl=1, rv=0, tbslbrm=0)
    s = O.segmentation(None, O.len_with_crc(1, big["A"]), 1)
    assert s["C"] == 152 and s["Z"] == 384
    tbs = [tiny, big]
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    stride = hip.ldpc.HARQ_STRIDE
    segs = [1, 152]
    harq_gpu = np.zeros((sum(segs), stride), np.int16)
    harq_ref = [[np.zeros(stride, np.int16) for _ in range(c)] for c in segs]
    llrlen = [0, 0]
    for rnd, rv in enumerate((0, 3)):
        cur = [dict(t, rv=rv) for t in tbs]
        coded = hip.ldpc.dlsch_encode_host(cur, pays)
        for t, p, f in zip(cur, pays, coded):
            assert np.array_equal(f, O.dlsch_encode(t, p)), (t["A"], rv)
        llrs = [np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + 5.0 * rng.standard_normal(f.size)), -200, 200).astype(np.int16)
                for f in coded]
        rx = [dict(t, round=rnd, llrLen=llrlen[i]) for i, t in enumerate(cur)]
        out, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq_gpu, numMaxIter=8)
        off = 0
        for i, t in enumerate(cur):
            p_ref, ack_ref, its, state = O.ulsch_decode(dict(t), llrs[i], harq_ref[i], 8, rnd, llrlen[i], vec=True)
            assert bool(ack[i]) == ack_ref and itm[i] == min(max(its), 9) and rx[i]["llrLen"] == state, (t["A"], rnd, its[:4], int(itm[i]))
            if ack_ref:
                assert np.array_equal(out[i], p_ref) and np.array_equal(p_ref, pays[i])
            for r in range(segs[i]):
                assert np.array_equal(harq_gpu[off + r], harq_ref[i][r]), (t["A"], rnd, r)
            off += segs[i]
            llrlen[i] = state
    assert ack[1]  # the retransmission brings the big block home


def test_invalid_parameters(hip):
    with pytest.raises(RuntimeError):
        hip.ldpc.dlsch_encode_host([dict(A=1001, G=4000, BG=1, Qm=2, Nl=1)], [np.zeros(200, np.uint8)])     # A % 8
    with pytest.raises(RuntimeError):
        hip.ldpc.dlsch_encode_host([dict(A=1000, G=4001, BG=1, Qm=2, Nl=1)], [np.zeros(200, np.uint8)])     # G % (Nl*Qm)
    with pytest.raises(RuntimeError):
        hip.ldpc.dlsch_encode_host([dict(A=1000, G=4000, BG=3, Qm=2, Nl=1)], [np.zeros(200, np.uint8)])


def test_unfused_tx_path_too():
    """Default TX path = fused segment/encode/rate-match kernel; re-run the DL-SCH tests with the three-kernel path."""
    import os
    import subprocess
    import sys
    if os.environ.get("NRLDPC_HIP_ENC_KERNEL") == "bytes":
        pytest.skip("already the three-kernel run")
    env = dict(os.environ, NRLDPC_HIP_ENC_KERNEL="bytes")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_tb_chain.py"), "-m", "gpu", "-q", "-x",
                        "-k", "dlsch"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_host_batches_sharded_over_logical_devices(hip, tmp_path):
    """NRLDPC_HIP_DEVICES=0,0,0: the host-buffer entry points (LDPCdecoder_batch, nrLDPC_hip_dlsch_encode,
    nrLDPC_hip_ulsch_decode over two HARQ rounds) split their batch over three device contexts -- contiguous block ranges
    resp. whole transport blocks per device, each with its own streams, staging and plan cache (here all on GPU 0: the
    box has one) -- and every output byte, pass count, ACK, soft buffer and llrLen equals the single-device run's."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "multidev_script.py"
    outs = []
    for devs in (None, "0,0,0"):
        env = dict(os.environ)
        env.pop("NRLDPC_HIP_DEVICES", None)
        if devs:
            env["NRLDPC_HIP_DEVICES"] = devs
        f = tmp_path / f"out_{devs or 'single'}.npz"
        r = subprocess.run([sys.executable, str(script), str(f)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(f))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 13
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert not a["rx0_ack"].all() and a["rx1_ack"].sum() > a["rx0_ack"].sum()   # round 0 loses blocks, combining recovers them
    for o in outs:      # scattered, gapped, reverse-ordered soft buffers: same results, the gaps untouched
        assert o["rx0_scattered_equal"].all() and o["rx1_scattered_equal"].all()


def test_abort_stops_the_siblings_of_a_failed_segment(hip):
    """Transport-block-wide abort on the device (decoder.c:190-193, nr_ulsch_decoding.c:235-245: the first segment that
    fails raises the flag, its siblings stop at their next pass).  60 lost transport blocks (26 segments each, pure noise,
    numMaxIter = 20) next to 4 clean ones: verdicts, pass counts, payloads and soft buffers are the same with and without
    the flag -- a lost TB reports numMaxIter + 1, NACK and a zeroed payload either way.  What the flag saves depends on how
    a block's segments fall over the GPU's workgroup rounds: siblings that run side by side fail together and save
    nothing, segments that start after a sibling has failed leave at once (here the call gets ~15 % shorter; it must not
    get longer)."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "abort_script.py"
    res = {}
    for flag in ("1", "0"):
        env = dict(os.environ, NRLDPC_HIP_TB_ABORT=flag)
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[flag] = json.loads(r.stdout.strip().splitlines()[-1])
    on, off = res["1"], res["0"]
    for r in (on, off):
        assert r["ack"] == [0] * 60 + [1] * 4 and r["itm"][:60] == [21] * 60 and max(r["itm"][60:]) <= 3
        assert r["clean_payload_ok"] and r["lost_payload_zero"]
    assert on["harq"] == off["harq"]                # the soft buffers are written before the decoder runs
    print(f"abort on {on['ms']:.3f} ms, off {off['ms']:.3f} ms")
    assert on["ms"] < 1.05 * off["ms"], (on["ms"], off["ms"])


def test_first_transmissions_on_the_cut_graph_equal_the_whole_rate_mode(hip):
    """High-rate first transmissions (the MCS 27 regime: E reaches column 27-30 of BG1's 35-column R = 2/3 mode, or a few columns
    of BG2's) are decoded on the rate mode's graph cut behind the last column that received anything
    (nrLDPC_hip_ulsch_decoder_columns).  The oracle runs the WHOLE mode, as the reference does: payload, ACK and the largest
    pass count of every block must be equal at noise levels on both sides of the waterfall (pass counts 3 .. cap, lost blocks)."""
    m = hip.ldpc
    rng = np.random.default_rng(4242)
    tbs, n_cut = [], 0
    for bits, rate, BG, Qm in [(40000, 0.86, 1, 6), (40000, 0.80, 1, 6), (25000, 0.72, 1, 4), (8424 * 3, 0.90, 1, 8), (3000, 0.62, 2, 2),
                               (1500, 0.55, 2, 4), (40000, 0.86, 1, 6), (60000, 0.78, 1, 6)]:
        A = valid_tbs(bits, BG)
        sg = O.segmentation(None, O.len_with_crc(1, A), BG)
        G = int(A / rate) // (Qm * sg["C"]) * Qm * sg["C"]
        t = dict(A=A, G=G, BG=BG, Qm=Qm, Nl=1, rv=0, round=0, tbslbrm=0)
        E = O.get_E(G, sg["C"], Qm, 1, 0)
        R = O.get_R(0, E, BG, sg["Z"], 0, 0)[0]
        cols = m.ulsch_decoder_columns(BG, sg["Z"], sg["C"], sg["F"], sg["K"], 0, 0, E, 0, R)
        n_cut += cols < m.NCOLS[(BG, R)]
        tbs.append(t)
    assert n_cut >= 6, n_cut                                     # the cut is what this test exercises
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    stride = m.HARQ_STRIDE
    seen = set()
    for sigma in (1.5, 2.6, 3.2, 4.0):
        harq_gpu = np.zeros((sum(segs), stride), np.int16)
        llrs = []
        for t, p in zip(tbs, pays):
            f = O.dlsch_encode(t, p)
            llrs.append(np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16))
        out, ack, itm = m.ulsch_decode_host(tbs, llrs, harq_gpu, numMaxIter=8)
        row = 0
        for i, t in enumerate(tbs):
            harq_ref = [np.zeros(stride, np.int16) for _ in range(segs[i])]
            p_ref, ack_ref, its, _ = O.ulsch_decode(t, llrs[i], harq_ref, 8, 0, 0)
            assert bool(ack[i]) == ack_ref and itm[i] == max(its), (sigma, t, its, int(itm[i]))
            if ack_ref:
                assert np.array_equal(out[i], p_ref) and np.array_equal(out[i], pays[i])
            for r in range(segs[i]):
                assert np.array_equal(harq_gpu[row + r], harq_ref[r]), (sigma, i, r)
            row += segs[i]
            seen.add((bool(ack_ref), max(its)))
    assert any(a for a, _ in seen) and any(not a for a, _ in seen) and len({n for a, n in seen if a}) >= 3, seen
