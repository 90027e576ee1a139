"""CPU emulation of the HIP kernels' per-thread code (tests/emul/ldpc_emul.cpp) vs the oracle.

Checks the product's table builder, schedules and index/packed arithmetic without a GPU; says nothing about
barriers, LDS capacity or wave intrinsics (the -m gpu tests cover those)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O
from common import ALL_RATES, kbits, load_survey_decoder_vectors, make_llr, random_info


@pytest.fixture(scope="module")
def emul(built):
    L = C.CDLL(str(Path(__file__).resolve().parent / "emul" / "libldpc_emul.so"))
    for f in (L.ldpc_emul_decode, L.ldpc_emul_decode_fast):
        f.argtypes = [C.c_int] * 8 + [C.c_void_p, C.c_void_p]
    L.ldpc_emul_encode.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
    L.ldpc_emul_encode_packed.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_void_p]
    L.ldpc_emul_encode_packed32.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_int]
    return L


def run(emul, fast, BG, Z, R, llr, it, mode=0, use_crc=False, E=0, ct=1, init=0):
    out = np.full(68 * 384 + 64, init, np.uint8)
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    f = emul.ldpc_emul_decode_fast if fast else emul.ldpc_emul_decode
    if fast:
        emul.ldpc_emul_set_fast_shape(1 if fast == "latency" else 0)   # ldpc_graph.h LDPC_SHAPE_*
    n = f(BG, Z, R, it, mode, int(use_crc), E, ct, llr.ctypes.data, out.ctypes.data)
    return n, out[:O.out_bytes(BG, Z, R, mode)]


def variants(Z):
    """generic kernel; fast kernel in its throughput and latency workgroup shapes where it applies"""
    return (False, "throughput", "latency") if (Z % 4 == 0 and Z >= 8) else (False,)


@pytest.mark.parametrize("BG", [1, 2])
def test_decoder_every_code(emul, BG):
    rng = np.random.default_rng(BG)
    for Z in O.LIFT_SIZES:
        for R in ALL_RATES[BG]:
            K = kbits(BG, Z)
            info = random_info(rng, BG, Z, with_crc24b=True)
            for kind in (-1.0, 1.0, "rand", "sat"):
                llr = make_llr(rng, BG, Z, R, kind, info)
                for it, mode, crc in ((8, 0, False), (1, 0, False), (8, 1, False), (8, 0, True), (2, 0, True)):
                    if crc and (K % 8 or K < 48):
                        continue
                    ref = O.decode(BG, Z, R, llr, it, mode, crc, K, 1, out_init=0x3c)
                    for fast in variants(Z):
                        n, out = run(emul, fast, BG, Z, R, llr, it, mode, crc, K, 1, init=0x3c)
                        assert n == ref[0] and np.array_equal(out, ref[1]), (fast, BG, Z, R, kind, it, mode, crc)


def test_decoder_survey_vectors(emul):
    for v in load_survey_decoder_vectors():
        for fast in variants(v["Z"]):
            n, out = run(emul, fast, v["BG"], v["Z"], v["R"], v["llr"], v["numMaxIter"], v["outMode"], v["use_crc"],
                         v["E"], v["crc_type"], init=0x55)
            assert n == v["n_iter"] and np.array_equal(out, v["out"]), (fast, v["BG"], v["Z"], v["R"])


def test_encoder_every_code(emul):
    rng = np.random.default_rng(9)
    for BG in (1, 2):
        for Z in O.LIFT_SIZES:
            for Kb in ([22] if BG == 1 else [10, 9, 8, 6]):
                bits = rng.integers(0, 2, kbits(BG, Z), dtype=np.uint8)
                if Kb < 10:
                    bits[Kb * Z:] = 0
                info = np.packbits(np.concatenate([bits, np.zeros((-bits.size) % 8, np.uint8)]))
                out = np.full(68 * 384 + 8, 7, np.uint8)
                ref = O.encode(BG, Z, info, Kb)
                for fn in (emul.ldpc_emul_encode, emul.ldpc_emul_encode_packed):  # byte-per-lane and bit-packed kernels
                    out[:] = 7
                    n = fn(BG, Z, Kb, info.ctypes.data, out.ctypes.data)
                    assert n == ref.size and np.array_equal(out[:n], ref), (BG, Z, Kb, fn.__name__)


def test_encoder_word_aligned_path_on_the_cpu(emul):
    """ldpc_enc_packed32.h -- what ldpc_enc_packed_kernel and the fused TX kernel run for Zc % 32 == 0 -- one thread at a time on
    the CPU: every such code, Kb < 10, several workgroup sizes (the extension phase's two-words-per-lane split and the loops'
    strides depend on it), against the oracle AND against the code words of the reference-compiled encoder
    (tests/golden/ref_encoder.npz)."""
    from common import load_ref_code_words
    rng = np.random.default_rng(32)
    sizes = [Z for Z in O.LIFT_SIZES if Z % 32 == 0]
    assert len(sizes) == 12
    out = np.full(68 * 384 + 64, 7, np.uint8)
    for BG in (1, 2):
        for Z in sizes:
            for Kb in ([22] if BG == 1 else [10, 9, 8, 6]):
                bits = rng.integers(0, 2, kbits(BG, Z), dtype=np.uint8)
                bits[Kb * Z:] = 0
                info = np.packbits(bits)
                ref = O.encode(BG, Z, info, Kb)
                for nt in (0, 64, 128, 512):
                    out[:] = 7
                    n = emul.ldpc_emul_encode_packed32(BG, Z, Kb, info.ctypes.data, out.ctypes.data, nt)
                    assert n == ref.size and np.array_equal(out[:n], ref), (BG, Z, Kb, nt)
                    assert (out[n:] == 7).all()
    assert emul.ldpc_emul_encode_packed32(1, 36, 22, info.ctypes.data, out.ctypes.data, 0) == -2   # not a word-aligned code
    n_ref = 0
    for v in load_ref_code_words():
        if v["Z"] % 32:
            continue
        n = emul.ldpc_emul_encode_packed32(v["BG"], v["Z"], v["Kb"], np.ascontiguousarray(v["info"]).ctypes.data, out.ctypes.data, 0)
        assert n == v["coded"].size and np.array_equal(out[:n], v["coded"]), (v["BG"], v["Z"], v["Kb"])
        n_ref += 1
    assert n_ref >= 48


def test_rx_dematch_phases_against_the_oracle(emul):
    """tb_rx_core.h (the de-matching kernel = the fused segment kernel's prologue), a workgroup's threads walked phase by
    phase: soft buffer and int8 decoder input equal nr_deinterleaving_ldpc -> nr_rate_matching_ldpc_rx -> the caller's
    pack as the oracle restates them -- first transmission and a combining round on a dirty buffer, rv 0-3, LBRM,
    repetition (several laps), fillers, every Qm, unaligned soft buffers (scalar tail path)."""
    emul.tb_emul_rx_dematch.argtypes = [C.c_uint32, C.c_int] + [C.c_uint32] * 4 + [C.c_int] + [C.c_uint32] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 3
    rng = np.random.default_rng(77)
    cases = 0
    for BG, A, lbrm in ((1, 30000, 0), (1, 30000, 24000), (2, 3000, 0), (1, 9000, 0), (2, 640, 0), (1, 100000, 150000)):
        s = O.segmentation(None, O.len_with_crc(1, A), BG)
        Z, K, F, Cn = s["Z"], s["K"], s["F"], s["C"]
        N = (66 if BG == 1 else 50) * Z
        for Qm in (2, 4, 6, 8):
            for rv in range(4):
                for rate in (0.25, 0.6, 0.92, 0.08):                # 0.08: E > Ncb, several laps
                    E = max(Qm * 4, int((K - F) / rate) // Qm * Qm)
                    R, _ = O.get_R(rv, E, BG, Z, 0, 0)
                    ncols = O.NCOLS[(BG, R)]
                    f = rng.integers(-300, 300, E).astype(np.int16)
                    for clear, misalign in ((1, 0), (0, 0), (0, 1)):
                        w0 = rng.integers(-2000, 2000, 66 * 384 + 16).astype(np.int16)
                        Ncb = N if not lbrm else min(N, (3 * lbrm // (2 * Cn)))
                        w0[misalign + Ncb:misalign + N] = 0      # the reference's d[r] is calloc'ed and never written behind Ncb
                        # oracle: the reference's three steps
                        e = O.deinterleave(E, Qm, f)
                        d_ref = w0[misalign:misalign + N].copy()
                        rc, d_ref = O.rate_match_rx(lbrm, BG, Z, d_ref, e, Cn, rv, clear, E, F, K - F - 2 * Z)
                        assert rc == 0
                        l_ref = O.llr_prepack(d_ref, BG, Z, K, F, ncols)
                        # emulated workgroup
                        w = w0.copy()
                        l = np.full(ncols * Z + 8, 0x11, np.int8)
                        span = emul.tb_emul_rx_dematch(lbrm, BG, Z, Cn, F, K, rv, E, Qm, ncols * Z, clear, 256, f.ctypes.data,
                                                       w[misalign:].ctypes.data, l.ctypes.data)
                        assert span > 0
                        got = w[misalign:misalign + N]
                        assert np.array_equal(got, d_ref), (BG, A, Qm, rv, rate, clear, misalign)
                        assert np.array_equal(w[misalign + N:], w0[misalign + N:])      # nothing behind the row is touched
                        assert np.array_equal(l[:ncols * Z], l_ref), (BG, A, Qm, rv, rate, clear, misalign)
                        assert (l[ncols * Z:] == 0x11).all()
                        cases += 1
    assert cases > 1000
