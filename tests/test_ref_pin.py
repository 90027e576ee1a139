"""oracle/_ref (reference-COMPILED: encoder parity part + generator tables, nrLDPC_init LUT selection, all decoder data
movement of nrLDPC_mPass.h) against the oracle's restatement.  Runs wherever oracle/_ref can be built or was shipped
(the development container; the GPU box when the snapshot carried the .so); the committed fixtures
tests/golden/ref_*.npz carry the same outputs everywhere else (test_ref_fixtures.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as RL

pytestmark = pytest.mark.skipif(not RL.available(), reason="oracle/_ref not built (needs /root/reference)")

MODES = [(1, 13), (1, 23), (1, 89), (2, 15), (2, 13), (2, 23)]


def test_ref_library_needs_nothing_but_libc():
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", str(RL.LIB_PATH)], capture_output=True, text=True).stdout
    names = {l.split()[-1].split("@")[0] for l in und.splitlines() if l.strip()}
    names -= {n for n in names if n.startswith("_ITM") or n.startswith("__gmon") or n.startswith("__cxa")}
    assert names <= {"memcpy", "memmove", "memset", "calloc", "free", "puts", "printf", "__memset_chk", "__memcpy_chk",
                     "__memmove_chk", "__stack_chk_fail", "__tls_get_addr", "__printf_chk"}, names


@pytest.mark.parametrize("BG", [1, 2])
def test_reference_encoder_equals_oracle_every_lifting_size(BG):
    """a14/a21: ldpc_encoder.c's code word, parity part by reference-compiled encode_parity_check_part_orig with the
    reference's generator tables, = the oracle's (which solves H x = 0 from the BG*_I* shift tables)."""
    rng = np.random.default_rng(100 + BG)
    kb = 22 if BG == 1 else 10
    for Z in O.LIFT_SIZES:
        assert RL.lib().ref_has_generator_matrix(BG, Z)
        for trial in range(2):
            K = kb * Z
            bits = rng.integers(0, 2, K, dtype=np.uint8)
            info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
            ref = RL.encode(BG, Z, info)
            assert np.array_equal(ref, O.encode(BG, Z, info)), (BG, Z)
            full = np.concatenate([bits[:2 * Z], ref])
            assert O.syndrome_weight(BG, Z, full) == 0


def test_reference_encoder_kb_below_ten_and_shortened_lengths():
    """BG2 with Kb = 6, 8, 9 (nr_segmentation.c:69-82) and ldpctest's shortened block lengths."""
    rng = np.random.default_rng(7)
    for Z, Kb in ((8, 6), (36, 8), (64, 9), (208, 10), (384, 9)):
        K = 10 * Z
        bits = rng.integers(0, 2, K, dtype=np.uint8)
        bits[Kb * Z:] = 0                      # columns Kb..9 are fillers = 0 when Kb < 10
        info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
        assert np.array_equal(RL.encode(2, Z, info, Kb=Kb), O.encode(2, Z, info, Kb=Kb)), (Z, Kb)


def test_reference_parity_part_bit_sliced_eight_segments():
    """a15: ldpc_encoder_optim8segmulti.c:132-208 feeds the SAME function with 8 segments sliced into the bits of a
    byte; the result de-sliced = 8 independent code words."""
    rng = np.random.default_rng(8)
    for BG, Z in ((1, 384), (1, 22), (2, 64), (2, 15)):
        kb, nrows = (22, 46) if BG == 1 else (10, 42)
        bits = rng.integers(0, 2, (8, kb * Z), dtype=np.uint8)
        c = np.zeros(kb * Z, np.uint8)
        for s in range(8):
            c |= bits[s] << s
        d = RL.parity_part(BG, Z, c)
        for s in range(8):
            info = np.packbits(np.concatenate([bits[s], np.zeros((-kb * Z) % 8, np.uint8)]))
            cw = O.encode(BG, Z, info)
            assert np.array_equal((d >> s) & 1, cw[(kb - 2) * Z:]), (BG, Z, s)


@pytest.mark.parametrize("BG,R", MODES)
def test_reference_init_numllr_and_group_luts(BG, R):
    """a3: nrLDPC_init selects a LUT set for every lifting size; numLLR, CN-group populations and BN-group
    populations agree with the oracle's graph."""
    for Z in O.LIFT_SIZES:
        h = RL.lib().ref_dec_new(BG, Z, R)
        assert h, (BG, Z, R)
        g = O.graph(BG, Z, R)
        assert RL.lib().ref_dec_numLLR(h) == g.ncols * Z == O.NCOLS[(BG, R)] * Z
        G = RL.lib().ref_dec_numCnGroups(h)
        ncn = RL.lib().ref_dec_numCnInCnGroups(h)
        degs = [g.row_ptr[r + 1] - g.row_ptr[r] for r in range(g.nrows)]
        for k in range(G):
            assert ncn[k] == degs.count(RL.lib().ref_dec_bnInCnGroup(h, k)), (BG, Z, R, k)
        assert sum(ncn[k] for k in range(G)) == g.nrows
        nbn = RL.lib().ref_dec_numBnInBnGroups(h)
        cols = [g.col[e] for e in range(g.nedges)]
        hist = [0] * 31
        for c in range(g.ncols):
            hist[cols.count(c)] += 1
        assert [nbn[d - 1] for d in range(1, 31)] == hist[1:31], (BG, Z, R)
        RL.lib().ref_dec_free(h)
    assert not RL.lib().ref_dec_new(BG, 7 * 64 + 1, R)   # not a lifting size


def _llr_cases(rng, BG, Z, R, n):
    """Inputs that exercise both outcomes: noisy code words over a range of SNRs (converging and not), uniform
    garbage, and the saturation corners."""
    kb = 22 if BG == 1 else 10
    ncols = O.NCOLS[(BG, R)]
    for i in range(n):
        info = rng.integers(0, 256, (kb * Z + 7) // 8, dtype=np.uint8)
        if (kb * Z) % 8:
            info[-1] &= 0xFF << (8 - (kb * Z) % 8) & 0xFF
        cw = O.encode(BG, Z, info)
        rate = kb / (ncols - 2)
        snr = rng.choice([-3.0, 0.0, 1.0, 2.0, 4.0, 8.0]) + 10 * np.log10(rate * 3)
        yield O.awgn_llr(rng, cw, Z, snr)[:ncols * Z].copy()
    yield rng.integers(-128, 128, ncols * Z, dtype=np.int8)
    yield rng.choice(np.array([-128, -127, 127, 0], np.int8), ncols * Z)
    yield np.full(ncols * Z, -128, np.int8)
    yield np.zeros(ncols * Z, np.int8)


@pytest.mark.parametrize("BG,R", MODES)
def test_hybrid_reference_decoder_equals_oracle_every_lifting_size(BG, R):
    """a2-a12: the decoder whose set-up and data movement are reference-compiled (nrLDPC_init, nrLDPC_mPass.h) and
    whose node arithmetic is restated on the reference's buffer layouts gives the oracle's pass counts and bits for
    every lifting size -- i.e. the oracle's (edge, lane) formulation, its graph and its parity-check group order
    [F6] are the reference's."""
    rng = np.random.default_rng(BG * 100 + R)
    for Z in O.LIFT_SIZES:
        n = 1 if Z > 100 else 2
        for llr in _llr_cases(rng, BG, Z, R, n):
            for max_iter in (8,) if Z > 100 else (1, 8):
                n_ref, out_ref = RL.decode(BG, Z, R, llr, max_iter, 0, out_init=0xA5)
                n_or, out_or = O.decode(BG, Z, R, llr, max_iter, O.OUT_BIT, out_init=0xA5)
                assert n_ref == n_or, (BG, Z, R, max_iter, n_ref, n_or)
                assert np.array_equal(out_ref, out_or), (BG, Z, R, max_iter)


@pytest.mark.parametrize("BG,Z,R", [(1, 384, 13), (1, 176, 23), (1, 36, 89), (2, 64, 15), (2, 208, 13), (2, 15, 23)])
def test_hybrid_reference_decoder_output_modes_and_crc_stop(BG, Z, R):
    """a10/a11: BITINT8 / LLRINT8 outputs and the CRC stop, the predicate called as decoder.c:857 calls it."""
    rng = np.random.default_rng(Z)
    kb = 22 if BG == 1 else 10
    K = kb * Z
    for snr in (6.0, 1.0, -6.0):
        nbytes = K // 8
        payload = rng.integers(0, 256, nbytes - 3, dtype=np.uint8)
        crc = O.crc("crc24b", np.concatenate([payload, np.zeros(4, np.uint8)]), (nbytes - 3) * 8) >> 8
        info = np.concatenate([payload, np.array([(crc >> 16) & 255, (crc >> 8) & 255, crc & 255], np.uint8)])
        if K % 8:
            info = np.concatenate([info, np.zeros(1, np.uint8)])
        cw = O.encode(BG, Z, info)
        llr = O.awgn_llr(rng, cw, Z, snr + 10 * np.log10(kb / (O.NCOLS[(BG, R)] - 2) * 3))[:O.NCOLS[(BG, R)] * Z]
        E = nbytes * 8
        calls = []

        def pred(p, n, t):
            calls.append((n, t))
            buf = np.ctypeslib.as_array(p, shape=(n // 8 + 4,))
            return O.check_crc(buf, n, t)
        for mode in (0, 1, 2):
            calls.clear()
            n_ref, out_ref = RL.decode(BG, Z, R, llr, 8, mode, check_crc=pred, E=E, crc_type=O.CRC24_B, out_init=0x5A)
            n_or, out_or = O.decode(BG, Z, R, llr, 8, mode, use_crc=True, E=E, crc_type=O.CRC24_B, out_init=0x5A)
            assert n_ref == n_or and np.array_equal(out_ref, out_or), (BG, Z, R, snr, mode)
            assert calls and all(c == (E, O.CRC24_B) for c in calls) and len(calls) == max(0, n_ref - 2)
            n_ref, out_ref = RL.decode(BG, Z, R, llr, 8, mode, out_init=0x5A)
            n_or, out_or = O.decode(BG, Z, R, llr, 8, mode, out_init=0x5A)
            assert n_ref == n_or and np.array_equal(out_ref, out_or)


def test_generic_bnprocpc_differs_only_in_parity_columns():
    """[F5]: following the generic bnProcPc (degree-1 group summed) instead of the generated one changes the output
    behind the core columns only -- and never the pass count."""
    rng = np.random.default_rng(55)
    for BG, Z, R in ((1, 96, 13), (2, 52, 15), (1, 384, 23)):
        ncore = 26 if BG == 1 else 14
        for llr in _llr_cases(rng, BG, Z, R, 2):
            n0, o0 = RL.decode(BG, Z, R, llr, 8, 1)
            n1, o1 = RL.decode(BG, Z, R, llr, 8, 1, deg1_generic=True)
            assert n0 == n1 and np.array_equal(o0[:ncore * Z], o1[:ncore * Z])
            assert not o0[ncore * Z:].any()
