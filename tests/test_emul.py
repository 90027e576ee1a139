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
