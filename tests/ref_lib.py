"""ctypes binding of oracle/_ref/libref_nrldpc.so -- reference-COMPILED pieces of the path (TEST INFRASTRUCTURE ONLY).

oracle/ref_pin/ holds the recipe; it compiles reference sources where they lie under /root/reference, so the
library can only be BUILT in the development container.  Once built it travels to the GPU box with the snapshot
(oracle/_ref/ is git-ignored, not gpurun-ignored); tests that want it call available() and skip otherwise --
the committed fixtures tests/golden/ref_*.npz (tools/make_ref_fixtures.py) carry its outputs everywhere.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
PIN_DIR = ROOT / "oracle" / "ref_pin"
LIB_PATH = ROOT / "oracle" / "_ref" / "libref_nrldpc.so"
REFERENCE = Path("/root/reference")
NCOLS = {(1, 13): 68, (1, 23): 35, (1, 89): 27, (2, 15): 52, (2, 13): 32, (2, 23): 17}
CHECK_CRC_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint8)

_lib = None


def build():
    """(Re)build oracle/_ref when the reference tree is present; no-op otherwise."""
    if (REFERENCE / "openair1/PHY/CODING/nrLDPC_encoder/ldpc_generate_coefficient.c").exists():
        subprocess.run(["make", "-C", str(PIN_DIR)], check=True, capture_output=True)
    return LIB_PATH


def available():
    try:
        build()
    except subprocess.CalledProcessError:
        return False
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        L.ref_encode_parity_check_part_orig.argtypes = [C.c_void_p, C.c_void_p] + [C.c_short] * 4
        L.ref_has_generator_matrix.argtypes = [C.c_short, C.c_short]
        L.ref_dec_new.argtypes = [C.c_int] * 3
        L.ref_dec_new.restype = C.c_void_p
        L.ref_dec_free.argtypes = [C.c_void_p]
        L.ref_dec_numLLR.argtypes = [C.c_void_p]
        L.ref_dec_numLLR.restype = C.c_uint32
        for n, t in (("numCnInCnGroups", C.c_uint8), ("startAddrCnGroups", C.c_uint32), ("numBnInBnGroups", C.c_uint8),
                     ("startAddrBnGroups", C.c_uint32), ("startAddrBnGroupsLlr", C.c_uint16)):
            f = getattr(L, "ref_dec_" + n)
            f.argtypes = [C.c_void_p]
            f.restype = C.POINTER(t)
        for n in ("numCnGroups",):
            getattr(L, "ref_dec_" + n).argtypes = [C.c_void_p]
        for n in ("bnInCnGroup", "cnInCnGroupFull"):
            getattr(L, "ref_dec_" + n).argtypes = [C.c_void_p, C.c_int]
        for n in ("llr2llrProcBuf", "llr2CnProcBuf", "cn2bnProcBuf", "bn2cnProcBuf", "llrRes2llrOut"):
            getattr(L, "ref_" + n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(L, "ref_" + n).restype = None
        for n in ("size_cn_proc_buf", "size_bn_proc_buf", "max_num_llr"):
            getattr(L, "ref_" + n).restype = C.c_uint32
        L.ref_hybrid_cnProcPc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_hybrid_cnProcPc.restype = C.c_uint32
        L.ref_hybrid_llr2bit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.ref_hybrid_llr2bit.restype = None
        L.ref_hybrid_decode.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_ldpc_encoder_orig.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def encode(BG, Z, info_bytes, Kb=None, block_length=None):
    """LDPCencoder (ldpc_encoder.c) around the reference-compiled parity part. Returns one bit per byte."""
    kbf = 22 if BG == 1 else 10
    K = kbf * Z if block_length is None else block_length
    info = np.concatenate([np.ascontiguousarray(info_bytes, dtype=np.uint8), np.zeros(8, np.uint8)])
    out = np.zeros(68 * 384, dtype=np.uint8)
    n = lib().ref_ldpc_encoder_orig(_p(info), _p(out), BG, Z, kbf if Kb is None else Kb, K)
    assert n > 0, (BG, Z, n)
    return out[:n].copy()


def parity_part(BG, Z, c, Kb=None):
    """encode_parity_check_part_orig on a caller-built c (uint8[ncols*Z], may be bit-sliced). Returns d uint8[nrows*Z]."""
    kbf, nrows = (22, 46) if BG == 1 else (10, 42)
    c = np.ascontiguousarray(c, dtype=np.uint8).copy()
    d = np.zeros(nrows * Z, dtype=np.uint8)
    rc = lib().ref_encode_parity_check_part_orig(_p(c), _p(d), BG, Z, kbf if Kb is None else Kb, kbf * Z)
    assert rc == 0
    return d


def out_bytes(BG, Z, R, out_mode):
    n = NCOLS[(BG, R)] * Z
    return ((n + 31) // 32) * 4 if out_mode == 0 else n


def decode(BG, Z, R, llr, max_iter=8, out_mode=0, check_crc=None, E=0, crc_type=1, deg1_generic=False, out_init=0):
    """Hybrid decoder: reference-compiled set-up and data movement + restated node arithmetic.
    check_crc: None or a Python callable (bytes_ptr, n, crc_type) -> int."""
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    out = np.full(max(out_bytes(BG, Z, R, 0), out_bytes(BG, Z, R, 1)) + 64, out_init, dtype=np.uint8)
    cb = CHECK_CRC_T(check_crc) if check_crc is not None else None
    n = lib().ref_hybrid_decode(BG, Z, R, max_iter, out_mode, C.cast(cb, C.c_void_p) if cb else None, E, crc_type,
                                int(deg1_generic), _p(llr), _p(out))
    return n, out[:out_bytes(BG, Z, R, out_mode)]
