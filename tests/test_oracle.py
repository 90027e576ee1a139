"""The oracle itself (CPU only): what pins the C restatement in oracle/ -- see DESIGN.md "Oracle"."""
import numpy as np
import pytest

import oracle_lib as O
from common import ALL_RATES, GOLDEN, kbits, load_survey_decoder_vectors, make_llr, random_info


def test_crc_known_answers():
    """Catalogue check values for the message "123456789" (CRC-24/LTE-A, CRC-24/LTE-B, CRC-16/XMODEM, CRC-8/LTE):
    the polynomials of crc_byte.c:46-58 with zero init, MSB first."""
    msg = np.frombuffer(b"123456789", dtype=np.uint8)
    assert O.crc("crc24a", msg, 72) >> 8 == 0xCDE703
    assert O.crc("crc24b", msg, 72) >> 8 == 0x23EF52
    assert O.crc("crc16", msg, 72) >> 16 == 0x31C3
    assert O.crc("crc8", msg, 72) >> 24 == 0xEA
    # appending the CRC makes check_crc true, flipping any bit makes it false
    for name, ct, nb in (("crc24a", 0, 3), ("crc24b", 1, 3), ("crc16", 2, 2), ("crc8", 3, 1)):
        c = O.crc(name, msg, 72) >> (32 - 8 * nb)
        word = np.concatenate([msg, np.array([(c >> (8 * (nb - 1 - i))) & 255 for i in range(nb)], np.uint8)])
        assert O.check_crc(word, 8 * word.size, ct) == 1
        word[3] ^= 0x10
        assert O.check_crc(word, 8 * word.size, ct) == 0
    # non byte-aligned lengths follow the bit-serial definition (crc_byte.c:65-84)
    assert O.crc("crc24c", msg, 13) == O.crc("crc24c", np.array([msg[0], msg[1] & 0xF8], np.uint8), 13)


def test_encoder_is_pinned_by_the_parity_check_matrix():
    """Systematic + H c = 0 for every lifting size: the code word is unique, so this pins the encoder."""
    rng = np.random.default_rng(1)
    for BG in (1, 2):
        for Z in O.LIFT_SIZES:
            info = random_info(rng, BG, Z)
            K = kbits(BG, Z)
            cw = O.encode(BG, Z, info)
            bits = np.unpackbits(info)[:K]
            assert np.array_equal(cw[:K - 2 * Z], bits[2 * Z:])
            assert O.syndrome_weight(BG, Z, np.concatenate([bits[:2 * Z], cw])) == 0
            cw[7 % cw.size] ^= 1
            assert O.syndrome_weight(BG, Z, np.concatenate([bits[:2 * Z], cw])) > 0


def test_encoder_linearity_and_kb():
    rng = np.random.default_rng(2)
    for (BG, Z) in [(1, 384), (2, 64), (2, 208), (1, 20)]:
        a, b = random_info(rng, BG, Z), random_info(rng, BG, Z)
        assert np.array_equal(O.encode(BG, Z, a) ^ O.encode(BG, Z, b), O.encode(BG, Z, a ^ b))
    # Kb < 10 (BG2 short blocks): columns >= Kb are fillers; with zero fillers the result equals Kb = 10
    for Kb in (6, 8, 9):
        bits = rng.integers(0, 2, 10 * 40, dtype=np.uint8)
        bits[Kb * 40:] = 0
        info = np.packbits(bits)
        assert np.array_equal(O.encode(2, 40, info, Kb), O.encode(2, 40, info, 10))


def test_decoder_round_trip_and_reference_acceptance():
    """ldpctest's acceptance criterion (cmake_targets/autotests/test_case_list.xml:68-94: BLER 0 at -s10, BG1 R=1/3)
    with ldpctest's own channel (OAI RNG, quantiser) -- a few blocks per length."""
    for length, Z in ((3872, 176), (8448, 384)):
        rng = O.OaiRng(1234)
        snr_lin = 10 ** (10 / 10.0) * 1 / 3                    # ldpctest.c:519-522
        sigma = 1.0 / np.sqrt(2 * snr_lin)
        nprng = np.random.default_rng(length)
        for _ in range(4):
            info = random_info(nprng, 1, Z)
            cw = O.encode(1, Z, info)
            llr = rng.ldpctest_channel(cw, Z, sigma)
            n, out = O.decode(1, Z, 13, llr, 5)
            assert n <= 5 and np.array_equal(out[:length // 8], info)


def test_decoder_iteration_control():
    rng = np.random.default_rng(5)
    BG, Z, R = 2, 64, 15
    noise = make_llr(rng, BG, Z, R, "rand")
    for it in (0, 1, 2, 8, 20):
        n, _ = O.decode(BG, Z, R, noise, it)
        assert n == it + 1                                      # decoder.c:552-558
    clean = make_llr(rng, BG, Z, R, 8.0)
    assert O.decode(BG, Z, R, clean, 8)[0] in (2, 3)            # earliest possible stop: parity check after pass 2
    # CRC mode: never before pass 3, p_out untouched when fewer than 3 passes run (decoder.c:849-861)
    info = random_info(rng, BG, Z, with_crc24b=True)
    clean = make_llr(rng, BG, Z, R, 8.0, info)
    n, out = O.decode(BG, Z, R, clean, 8, use_crc=True, E=640, crc_type=1, out_init=0x77)
    assert n == 3 and np.array_equal(out[:80], info)
    n, out = O.decode(BG, Z, R, clean, 1, use_crc=True, E=640, crc_type=1, out_init=0x77)
    assert n == 2 and (out == 0x77).all()
    # parity columns are reported as 0, LLRINT8 == BITINT8 (decoder.c:866-877)
    n, bits = O.decode(BG, Z, R, clean, 8, out_mode=O.OUT_BITINT8)
    assert not bits[14 * Z:].any()
    assert np.array_equal(bits, O.decode(BG, Z, R, clean, 8, out_mode=O.OUT_LLRINT8)[1])


def test_vectorisable_restatement_equals_the_scalar_one():
    """oracle_ldpc_decoder_vec.c (two-minimum check node, lane-contiguous loops; used for the CPU baseline and for
    whole-batch GPU parity tests) against oracle_ldpc_decoder.c: same outputs and pass counts for every (BG, Zc, R),
    clean / noisy / random / saturated inputs, every output mode, both stop modes, several iteration caps -- and on
    the survey-stage vectors."""
    rng = np.random.default_rng(77)
    n = 0
    for BG in (1, 2):
        for Z in O.LIFT_SIZES:
            for R in ALL_RATES[BG]:
                K = kbits(BG, Z)
                info = random_info(rng, BG, Z, with_crc24b=True)
                for kind in (-1.0, 1.0, "rand", "sat"):
                    llr = make_llr(rng, BG, Z, R, kind, info)
                    for it, mode, crc in ((8, 0, False), (1, 1, False), (3, 2, False), (8, 0, True), (2, 0, True)):
                        if crc and (K % 8 or K < 48):
                            continue
                        if Z > 64 and (it, mode) not in ((8, 0),):   # keep the scalar side's run time in check
                            continue
                        a = O.decode(BG, Z, R, llr, it, mode, crc, K, 1, out_init=0x3c)
                        b = O.decode(BG, Z, R, llr, it, mode, crc, K, 1, out_init=0x3c, vec=True)
                        assert a[0] == b[0] and np.array_equal(a[1], b[1]), (BG, Z, R, kind, it, mode, crc)
                        n += 1
    assert n > 2000
    for v in load_survey_decoder_vectors():
        it, out = O.decode(v["BG"], v["Z"], v["R"], v["llr"], v["numMaxIter"], v["outMode"], v["use_crc"], v["E"],
                           v["crc_type"], out_init=0x55, vec=True)
        assert it == v["n_iter"] and np.array_equal(out, v["out"]), v


def test_survey_stage_vectors():
    """Supplementary vectors recorded from the survey-stage reference build (tools/dev_make_survey_vectors.py;
    NOT the parity pin -- see its docstring)."""
    n = 0
    for v in load_survey_decoder_vectors():
        it, out = O.decode(v["BG"], v["Z"], v["R"], v["llr"], v["numMaxIter"], v["outMode"], v["use_crc"], v["E"],
                           v["crc_type"], out_init=0x55)
        assert it == v["n_iter"] and np.array_equal(out, v["out"]), v
        n += 1
    assert n == 456
    z = np.load(GOLDEN / "survey_ref_encoder.npz")
    for i, (BG, Z, Kb, nbits) in enumerate(z["meta"]):
        cw = O.encode(int(BG), int(Z), z[f"info_{i}"], int(Kb))
        assert np.array_equal(np.packbits(cw[:nbits]), z[f"coded_{i}"])


def test_segmentation():
    # nr_segmentation.c:44-64: C, K', Zc, F; lifting sizes only
    for BG, B in [(1, 8448), (1, 8449), (1, 100000), (1, 24), (2, 3840), (2, 3841), (2, 640), (2, 641), (2, 560),
                  (2, 192), (2, 40), (1, 152000)]:
        s = O.segmentation(None, B, BG)
        Kcb = 8448 if BG == 1 else 3840
        C_ = 1 if B <= Kcb else -(-B // (Kcb - 24))
        assert s["C"] == C_
        assert s["Z"] in O.LIFT_SIZES and s["K"] == (22 if BG == 1 else 10) * s["Z"]
        Bp = B if C_ == 1 else B + 24 * C_
        assert s["F"] == s["K"] - Bp // C_ and s["Kb"] * s["Z"] >= Bp // C_
        if s["Z"] > 2:
            assert s["Kb"] * O.LIFT_SIZES[O.LIFT_SIZES.index(s["Z"]) - 1] < Bp // C_   # smallest that fits
    rng = np.random.default_rng(3)
    tb = rng.integers(0, 256, 3000, dtype=np.uint8)
    s = O.segmentation(tb, 24000, 1)
    assert s["C"] == 3
    kp = (24000 + 72) // 3
    for r, seg in enumerate(s["segs"]):
        assert np.array_equal(seg[:(kp - 24) // 8], tb[r * (kp - 24) // 8:(r + 1) * (kp - 24) // 8])
        assert O.check_crc(seg, kp, O.CRC24_B) == 1 and not seg[kp // 8:].any()


def test_rate_matching_and_interleaving_round_trip():
    rng = np.random.default_rng(4)
    BG, Z = 1, 96
    K, F = 22 * Z, 40
    N = 66 * Z
    Foff = K - F - 2 * Z
    for Tbs, rv, E, Qm in [(0, 0, 3000, 2), (0, 1, 5000, 4), (0, 2, 7002, 6), (0, 3, 12000, 8), (40000, 0, 9000, 2),
                           (40000, 3, 2000, 4), (0, 0, 2 * N, 2)]:
        w = rng.integers(0, 2, N, dtype=np.uint8)
        w[Foff:Foff + F] = 2                                  # NR_NULL (coding_defs.h:44)
        rc, e = O.rate_match(Tbs, BG, Z, w, 2, F, Foff, rv, E)
        assert rc == 0 and e.max() <= 1                       # filler bits are never transmitted
        f = O.interleave(E, Qm, e)
        assert np.array_equal(f.reshape(E // Qm, Qm).T.reshape(-1), e)
        soft = (1 - 2 * f.astype(np.int16)) * 10
        e_rx = O.deinterleave(E, Qm, soft)
        assert np.array_equal(e_rx, (1 - 2 * e.astype(np.int16)) * 10)
        rc, d = O.rate_match_rx(Tbs, BG, Z, np.zeros(N, np.int16), e_rx, 2, rv, 1, E, F, Foff)
        assert rc == 0
        nz = d != 0
        assert not nz[Foff:Foff + F].any()
        assert np.array_equal(np.sign(d[nz]), 1 - 2 * w[nz].astype(np.int16))   # every combined LLR has the bit's sign
        if E >= 2 * N:
            assert (np.abs(d[:Foff]) >= 20).all()             # repetition: each position hit at least twice
    # HARQ combining accumulates, `clear` resets (nr_rate_matching.c:554-555)
    rc, d1 = O.rate_match_rx(0, BG, Z, np.zeros(N, np.int16), np.full(1000, 3, np.int16), 2, 0, 1, 1000, F, Foff)
    rc, d2 = O.rate_match_rx(0, BG, Z, d1.copy(), np.full(1000, 3, np.int16), 2, 0, 0, 1000, F, Foff)
    assert np.array_equal(d2, 2 * d1)
    # decoder rate mode thresholds (nr_rate_matching.c:405-421)
    assert O.get_R(0, 66 * 384, 1, 384)[0] == 13 and O.get_R(0, 22 * 384 + 100, 1, 384)[0] == 89
    assert O.get_R(0, 30 * 384, 1, 384)[0] == 23 and O.get_R(0, 50 * 64, 2, 64)[0] == 15
    assert O.get_R(0, 20 * 64, 2, 64)[0] == 13 and O.get_R(0, 11 * 64, 2, 64)[0] == 23
    l = O.llr_prepack(np.arange(-300, -300 + 66 * 8, dtype=np.int16), 1, 8, 22 * 8, 16, 68)
    assert not l[:16].any() and (l[22 * 8 - 16:22 * 8] == 127).all() and l[16] == -128 and l.size == 68 * 8


def test_oai_rng_is_deterministic_and_gaussian():
    a, b = O.OaiRng(42), O.OaiRng(42)
    xs = np.array([a.gauss() for _ in range(20000)])
    assert xs[:50].tolist() == [b.gauss() for _ in range(50)]
    assert abs(xs.mean()) < 0.03 and abs(xs.std() - 1.0) < 0.03
    assert O.OaiRng(2).uniform() == O.OaiRng(3).uniform()       # even seeds are bumped to odd (rangen_double.c:67-68)
    assert O.lib().oracle_quantize(0.5, 1000.0, 8) == 127 and O.lib().oracle_quantize(0.5, -1000.0, 8) == -128
    assert O.lib().oracle_quantize(0.5, -0.1, 8) == -1            # floor, not round


def test_rows_that_close_on_all_zero_columns_say_nothing_in_crc_stop_mode():
    """The claim behind the product's cut graphs (ldpc_graph.h LDPC_R_COLS), checked on the ORACLE alone -- the
    restatement of the reference's arithmetic, no GPU, no product code: a decoder input whose last columns are all zero, decoded
    in CRC-stop mode (a) on the whole rate mode, as the reference would, and (b) on the base graph cut behind the last non-zero
    column.  Pass counts and every output byte in front of the cut must be equal, what lies behind the cut in (a) must be
    zeros (degree-1 columns get no hard decision: F5) -- for converging and lost blocks, every iteration cap, cuts inside a
    column, at a column edge and right behind the core.  In PARITY-CHECK mode the claim does NOT hold (the dropped rows' own
    checks count), and the test shows that too."""
    rng = np.random.default_rng(606)
    n_cases, pc_differs = 0, 0
    for (BG, Z, R) in [(1, 384, 23), (1, 96, 13), (1, 176, 23), (2, 208, 13), (2, 64, 15), (1, 32, 89), (2, 16, 23)]:
        K, ncols, ncore = kbits(BG, Z), O.NCOLS[(BG, R)], 26 if BG == 1 else 14
        if K % 8:
            continue
        for keep in (ncore + 0.4, ncore + 1.0, ncore + 2.7, (ncore + ncols) / 2):
            if keep >= ncols - 1:
                continue
            for kind in (0.5, 3.0, 8.0):
                info = random_info(rng, BG, Z, with_crc24b=True)
                llr = make_llr(rng, BG, Z, R, kind, info)
                llr[int(keep * Z):] = 0
                need = max(-(-int(keep * Z) // Z), ncore + 1)
                for it in (1, 3, 8):
                    n_a, out_a = O.decode(BG, Z, R, llr, it, 0, True, K, 1, out_init=0x5a)
                    n_b, out_b = O.decode(BG, Z, 1000 + need, llr[:need * Z], it, 0, True, K, 1, out_init=0x5a)
                    assert n_a == n_b, (BG, Z, R, keep, kind, it, n_a, n_b)
                    if n_a >= 3 and n_a <= it + 1:      # (the reference leaves p_out alone before the first CRC look)
                        assert np.array_equal(out_a[:out_b.size], out_b) and not out_a[out_b.size:].any(), (BG, Z, R, keep, kind, it)
                    n_cases += 1
                n_pa, _ = O.decode(BG, Z, R, llr, 8)
                n_pb, _ = O.decode(BG, Z, 1000 + need, llr[:need * Z], 8)
                pc_differs += n_pa != n_pb
    assert n_cases > 150 and pc_differs > 0, (n_cases, pc_differs)
