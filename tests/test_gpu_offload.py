"""The reference's OFFLOAD plugin slot on the GPU (libldpc_hip_t2.so -> csrc/tb_offload.inc.cpp): one segment per call,
rate (de)matching + (de)interleaving + HARQ combining inside the library, soft buffers on the device.

What the T2 card computes internally is not in the reference (DPDK PMD + hardware); the oracle for this slot is the
reference's own CPU chain on the same inputs -- nr_deinterleaving_ldpc + nr_rate_matching_ldpc_rx + the int8 pack of
nr_ulsch_decoding.c:195-210 + the decoder in parity-check mode; LDPCencoder + nr_rate_matching_ldpc + nr_interleaving_ldpc
-- restated in oracle/ (bit exact)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

CASES = [  # BG, Z, F, Qm, code rate of the first transmission
    (1, 384, 0, 6, 0.6), (1, 384, 88, 4, 0.4), (1, 176, 32, 2, 0.5), (1, 96, 16, 8, 0.85), (2, 208, 40, 4, 0.3),
    (2, 64, 8, 2, 0.25), (2, 64, 0, 6, 0.6), (2, 384, 296, 4, 0.45), (1, 44, 8, 4, 0.5), (2, 12, 8, 2, 0.4),
]


def _segment(rng, BG, Z, F):
    K = (22 if BG == 1 else 10) * Z
    info = rng.integers(0, 256, K // 8, dtype=np.uint8)
    info[(K - F) // 8:] = 0                                   # filler bits are zeros (nr_segmentation.c:169-173)
    return K, info


def _oracle_tx(BG, Z, info, F, E, Qm, rv):
    K = (22 if BG == 1 else 10) * Z
    d = O.encode(BG, Z, info).copy()                          # c[2Zc..K) || parity, one bit per byte
    if F:
        d[K - F - 2 * Z:K - 2 * Z] = 2                        # NR_NULL (nr_dlsch_coding.c:177-180)
    rc, e = O.rate_match(0, BG, Z, d, 1, F, K - F - 2 * Z, rv, E)
    assert rc == 0
    return O.interleave(E, Qm, e)


@pytest.mark.parametrize("BG,Z,F,Qm,rate", CASES)
def test_offload_encoder_equals_the_cpu_chain(hip, BG, Z, F, Qm, rate):
    rng = np.random.default_rng(BG * 1000 + Z + Qm)
    K, info = _segment(rng, BG, Z, F)
    E = max(int((K - F) / rate) // Qm, 8) * Qm
    for rv in range(4):
        f_ref = _oracle_tx(BG, Z, info, F, E, Qm, rv)
        f = hip.ldpc.offload_encoder(BG, Z, info, F, E, Qm, rv)
        assert np.array_equal(f, f_ref), (BG, Z, F, Qm, rv)


@pytest.mark.parametrize("BG,Z,F,Qm,rate", CASES)
def test_offload_decoder_equals_the_cpu_chain_over_harq_rounds(hip, BG, Z, F, Qm, rate):
    """Four transmissions rv 0, 2, 3, 1 of one segment at a noise level where the first does not decode: after every
    round the pass count and the decoded bytes equal deinterleave -> de-match (int16 accumulation) -> pack -> decode of
    the CPU chain; the device soft buffer is per (ulsch_id, segment): a second segment interleaved on another r and one on
    another ulsch_id do not disturb it, and setCombIn = 0 starts afresh."""
    rng = np.random.default_rng(BG * 7919 + Z * 13 + Qm)
    K, info = _segment(rng, BG, Z, F)
    E = max(int((K - F) / rate) // Qm, 8) * Qm
    ncols = {1: {13: 68, 23: 35, 89: 27}, 2: {15: 52, 13: 32, 23: 17}}[BG]
    ulsch, r = int(rng.integers(0, 256)), int(rng.integers(0, 64))
    w_ref = np.zeros(66 * 384, np.int16)
    sigma = 7.0 if rate < 0.5 else 4.5
    llrLen = 0
    decoded_round = None
    for rnd, rv in enumerate((0, 2, 3, 1)):
        f = _oracle_tx(BG, Z, info, F, E, Qm, rv)
        y = np.clip(np.round((1 - 2 * f.astype(np.float64)) * 6 + sigma * rng.standard_normal(E)), -128, 127).astype(np.int8)
        R, llrLen = O.get_R(rv, E, BG, Z, llrLen, rnd)
        # CPU chain
        e = O.deinterleave(E, Qm, y.astype(np.int16))
        rc, w_ref = O.rate_match_rx(0, BG, Z, w_ref, e, 1, rv, 1 if rnd == 0 else 0, E, F, K - F - 2 * Z)
        assert rc == 0
        l = O.llr_prepack(w_ref, BG, Z, K, F, ncols[R])
        it_ref, out_ref = O.decode(BG, Z, R, l, max_iter=8)
        # a stranger on the neighbouring segment slot and on another ULSCH, between the rounds
        other = rng.integers(-60, 60, E).astype(np.int8)
        hip.ldpc.offload_decoder(BG, Z, R, other, Qm, rv, F, setCombIn=rnd > 0, ulsch_id=ulsch, r=(r + 1) % 64, numMaxIter=2)
        hip.ldpc.offload_decoder(BG, Z, R, other, Qm, rv, F, setCombIn=rnd > 0, ulsch_id=(ulsch + 1) % 256, r=r, numMaxIter=2)
        it, out = hip.ldpc.offload_decoder(BG, Z, R, y, Qm, rv, F, setCombIn=rnd > 0, ulsch_id=ulsch, r=r, numMaxIter=8)
        assert it == it_ref, (rnd, rv, it, it_ref)
        assert np.array_equal(out, out_ref[:(K + 7) // 8]), (rnd, rv)
        if it <= 8 and decoded_round is None:
            decoded_round = rnd
            assert np.array_equal(out[:(K - F) // 8], info[:(K - F) // 8])
    assert decoded_round is not None, "never decoded: the case does not exercise combining"
    # a first transmission again: the buffer starts afresh (same result as round 0 of a new process)
    f = _oracle_tx(BG, Z, info, F, E, Qm, 0)
    y = np.clip((1 - 2 * f.astype(np.int16)) * 20, -128, 127).astype(np.int8)
    R, _ = O.get_R(0, E, BG, Z, 0, 0)
    rc, w0 = O.rate_match_rx(0, BG, Z, np.zeros_like(w_ref), O.deinterleave(E, Qm, y.astype(np.int16)), 1, 0, 1, E, F, K - F - 2 * Z)
    it_ref, out_ref = O.decode(BG, Z, R, O.llr_prepack(w0, BG, Z, K, F, ncols[R]), max_iter=8)
    it, out = hip.ldpc.offload_decoder(BG, Z, R, y, Qm, 0, F, setCombIn=False, ulsch_id=ulsch, r=r, numMaxIter=8)
    assert it == it_ref and np.array_equal(out, out_ref[:(K + 7) // 8])


def test_offload_slot_rejects_bad_parameters(hip):
    y = np.zeros(600, np.int8)
    for kw in (dict(Qm=3), dict(rv=4), dict(F=10 * 64), dict(r=64)):
        args = dict(BG=2, Z=64, R=15, llr=y, Qm=2, rv=0, F=0, setCombIn=False)
        args.update(kw)
        with pytest.raises(RuntimeError):
            hip.ldpc.offload_decoder(**args)
    with pytest.raises(RuntimeError):
        hip.ldpc.offload_decoder(2, 64, 15, np.zeros(601, np.int8), 2, 0, 0, False)     # E not a multiple of Qm
    with pytest.raises(RuntimeError):
        hip.ldpc.offload_encoder(2, 64, np.zeros(80, np.uint8), 4, 600, 2, 0)            # K - F not whole bytes


def test_offload_slot_serves_threads_side_by_side(hip):
    """The T2 library serialises its callers with a global mutex (offload.c:1044, 1097); this one does not need to: four
    threads, each with its own ULSCH (= its own soft buffers), two HARQ rounds each, results equal to the CPU chain's."""
    import threading
    BG, Z, F, Qm, E = 1, 96, 16, 4, 3000
    K = 22 * Z
    ncols = {13: 68, 23: 35, 89: 27}
    cases = []
    for t in range(4):
        rng = np.random.default_rng(900 + t)
        _, info = _segment(rng, BG, Z, F)
        w = np.zeros(66 * 384, np.int16)
        llrLen, rounds = 0, []
        for rnd, rv in enumerate((0, 2)):
            f = _oracle_tx(BG, Z, info, F, E, Qm, rv)
            y = np.clip(np.round((1 - 2 * f.astype(np.float64)) * 6 + 5.5 * rng.standard_normal(E)), -128, 127).astype(np.int8)
            R, llrLen = O.get_R(rv, E, BG, Z, llrLen, rnd)
            rc, w = O.rate_match_rx(0, BG, Z, w, O.deinterleave(E, Qm, y.astype(np.int16)), 1, rv, 1 if rnd == 0 else 0, E, F, K - F - 2 * Z)
            it_ref, out_ref = O.decode(BG, Z, R, O.llr_prepack(w, BG, Z, K, F, ncols[R]), max_iter=8)
            rounds.append((rv, R, y, it_ref, out_ref[:(K + 7) // 8].copy()))
        cases.append(rounds)
    bad = []

    def worker(t):
        for rep in range(20):
            for rnd, (rv, R, y, it_ref, out_ref) in enumerate(cases[t]):
                it, out = hip.ldpc.offload_decoder(BG, Z, R, y, Qm, rv, F, setCombIn=rnd > 0, ulsch_id=40 + t, r=t, numMaxIter=8)
                if it != it_ref or not np.array_equal(out, out_ref):
                    bad.append((t, rep, rnd, it, it_ref))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad[:5]
