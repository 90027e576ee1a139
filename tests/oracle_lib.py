"""ctypes binding of oracle/liboracle_nr_coding.so -- the CPU checker (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB_PATH = ORACLE_DIR / "liboracle_nr_coding.so"

LIFT_SIZES = sorted(a * (1 << j) for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * (1 << j) <= 384)
class _Ncols(dict):
    def __missing__(self, key):        # (BG, 1000 + n): the graph cut to n columns (oracle_ldpc_graph; test infrastructure only)
        if key[1] > 1000:
            return key[1] - 1000
        raise KeyError(key)


NCOLS = _Ncols({(1, 13): 68, (1, 23): 35, (1, 89): 27, (2, 15): 52, (2, 13): 32, (2, 23): 17})
OUT_BIT, OUT_BITINT8, OUT_LLRINT8 = 0, 1, 2
CRC24_A, CRC24_B, CRC16, CRC8 = 0, 1, 2, 3


def build(force=False):
    srcs = list(ORACLE_DIR.glob("*.c")) + list(ORACLE_DIR.glob("*.h")) + [
        ROOT / "openairinterface5g_amd/csrc/nr_ldpc_bg_tables.h"]
    if force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
    return LIB_PATH


class Graph(C.Structure):
    _fields_ = [("BG", C.c_int), ("Z", C.c_int), ("R", C.c_int), ("nrows", C.c_int), ("ncols", C.c_int),
                ("ncore", C.c_int), ("nedges", C.c_int), ("row_ptr", C.c_int * 47), ("col", C.c_int * 316),
                ("shift", C.c_int * 316)]


class Rng(C.Structure):
    _fields_ = [("urseed", C.c_uint), ("iy", C.c_uint), ("ir", C.c_uint * 98), ("iset", C.c_int), ("gset", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        i8p, u8p, i16p = C.POINTER(C.c_int8), C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
        L.oracle_ldpc_decode.argtypes = [C.c_int] * 8 + [C.c_void_p, C.c_void_p]
        L.oracle_ldpc_decode.restype = C.c_int
        L.oracle_ldpc_decode_mt.argtypes = [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_ldpc_decode_mt.restype = C.c_int
        L.oracle_ldpc_decode_vec.argtypes = L.oracle_ldpc_decode.argtypes
        L.oracle_ldpc_decode_vec.restype = C.c_int
        L.oracle_ldpc_decode_vec_mt.argtypes = L.oracle_ldpc_decode_mt.argtypes
        L.oracle_ldpc_decode_vec_mt.restype = C.c_int
        L.oracle_ldpc_encode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_ldpc_encode.restype = C.c_int
        L.oracle_ldpc_syndrome_weight.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.oracle_ldpc_graph.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Graph)]
        for n in ("crc24a", "crc24b", "crc24c", "crc16", "crc8"):
            f = getattr(L, "oracle_" + n)
            f.argtypes = [C.c_void_p, C.c_int]
            f.restype = C.c_uint32
        L.oracle_check_crc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8]
        L.oracle_nr_segmentation.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint, C.POINTER(C.c_uint),
                                             C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_uint8]
        L.oracle_nr_interleaving_ldpc.argtypes = [C.c_uint32, C.c_uint8, C.c_void_p, C.c_void_p]
        L.oracle_nr_deinterleaving_ldpc.argtypes = [C.c_uint32, C.c_uint8, C.c_void_p, C.c_void_p]
        L.oracle_nr_get_R_ldpc_decoder.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int), C.c_int]
        L.oracle_nr_rate_matching_ldpc.argtypes = [C.c_uint32, C.c_uint8, C.c_uint16, C.c_void_p, C.c_void_p, C.c_uint8,
                                                   C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint32]
        L.oracle_nr_rate_matching_ldpc_rx.argtypes = [C.c_uint32, C.c_uint8, C.c_uint16, C.c_void_p, C.c_void_p,
                                                      C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_nr_get_E.argtypes = [C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8]
        L.oracle_nr_get_E.restype = C.c_uint32
        L.oracle_nr_llr_prepack.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5
        L.oracle_randominit.argtypes = [C.POINTER(Rng), C.c_ulong]
        L.oracle_uniformrandom.argtypes = [C.POINTER(Rng)]
        L.oracle_uniformrandom.restype = C.c_double
        L.oracle_gaussdouble.argtypes = [C.POINTER(Rng), C.c_double, C.c_double]
        L.oracle_gaussdouble.restype = C.c_double
        L.oracle_quantize.argtypes = [C.c_double, C.c_double, C.c_uint8]
        L.oracle_quantize.restype = C.c_int8
        L.oracle_ldpctest_channel.argtypes = [C.POINTER(Rng), C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.oracle_schsim_channel.argtypes = [C.POINTER(Rng), C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.oracle_schsim_channel.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def graph(BG, Z, R):
    g = Graph()
    rc = lib().oracle_ldpc_graph(BG, Z, R, C.byref(g))
    assert rc == 0, (BG, Z, R)
    return g


def out_bytes(BG, Z, R, out_mode):
    n = NCOLS[(BG, R)] * Z
    return ((n + 31) // 32) * 4 if out_mode == OUT_BIT else n


def decode(BG, Z, R, llr, max_iter=8, out_mode=OUT_BIT, use_crc=False, E=0, crc_type=CRC24_B, out_init=0, vec=False):
    """One code block. llr: int8[ncols*Z]. Returns (n_iter, out uint8[out_bytes]).  vec: the vectorisable restatement
    (oracle_ldpc_decoder_vec.c) instead of the scalar one."""
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    assert llr.size >= NCOLS[(BG, R)] * Z
    out = np.full(max(out_bytes(BG, Z, R, OUT_BIT), out_bytes(BG, Z, R, OUT_BITINT8)), out_init, dtype=np.uint8)
    f = lib().oracle_ldpc_decode_vec if vec else lib().oracle_ldpc_decode
    n = f(BG, Z, R, max_iter, out_mode, int(use_crc), E, crc_type, _p(llr), _p(out))
    return n, out[:out_bytes(BG, Z, R, out_mode)]


def decode_mt(nthreads, BG, Z, R, llr, max_iter=8, vec=False):
    """Many blocks on `nthreads` pthreads (PC stop, packed bits). llr: int8[n, >= ncols*Z]. Returns (n_iter, out)."""
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    n = llr.shape[0]
    ob = out_bytes(BG, Z, R, OUT_BIT)
    out = np.zeros((n, ob), dtype=np.uint8)
    it = np.zeros(n, dtype=np.int32)
    f = lib().oracle_ldpc_decode_vec_mt if vec else lib().oracle_ldpc_decode_mt
    rc = f(nthreads, n, BG, Z, R, max_iter, _p(llr), llr.shape[1], _p(out), ob, _p(it))
    assert rc == 0
    return it, out


def encode(BG, Z, info_bytes, Kb=None):
    """info_bytes: uint8[K/8] MSB-first. Returns uint8[(66|50)*Z], one bit per byte."""
    kbf = 22 if BG == 1 else 10
    Kb = kbf if Kb is None else Kb
    info_bytes = np.ascontiguousarray(info_bytes, dtype=np.uint8)
    out = np.zeros((68 if BG == 1 else 52) * Z, dtype=np.uint8)
    n = lib().oracle_ldpc_encode(BG, Z, Kb, _p(info_bytes), _p(out))
    assert n == ((66 if BG == 1 else 50) * Z), n
    return out[:n]


def syndrome_weight(BG, Z, x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    return lib().oracle_ldpc_syndrome_weight(BG, Z, _p(x))


def crc(name, data, bitlen):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return getattr(lib(), "oracle_" + name)(_p(data), bitlen)


def check_crc(data, n, crc_type):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return lib().oracle_check_crc(_p(data), n, crc_type)


def segmentation(tb_bytes, B, BG):
    """Returns dict(C,K,Z,F,Kb, segs=list of uint8[K/8])."""
    Cn, K, Z, F = C.c_uint(), C.c_uint(), C.c_uint(), C.c_uint()
    kb = lib().oracle_nr_segmentation(None, None, B, C.byref(Cn), C.byref(K), C.byref(Z), C.byref(F), BG)
    res = dict(C=Cn.value, K=K.value, Z=Z.value, F=F.value, Kb=kb, segs=[])
    if tb_bytes is not None and kb > 0:
        tb = np.ascontiguousarray(tb_bytes, dtype=np.uint8)
        # 0xAA where nr_segmentation must write every byte; zeros behind K>>3, like the reference's calloc'ed harq->c[r]:
        # when K is not a multiple of 8 (Zc = 15: K = 330) the reference never touches the last partial byte, so the
        # filler bits in it are whatever the buffer held -- zero on first use, which is the case modelled here
        segs = [np.concatenate([np.full(K.value // 8, 0xAA, np.uint8), np.zeros(4, np.uint8)]) for _ in range(Cn.value)]
        ptrs = (C.c_void_p * Cn.value)(*[s.ctypes.data for s in segs])
        lib().oracle_nr_segmentation(_p(tb), ptrs, B, C.byref(Cn), C.byref(K), C.byref(Z), C.byref(F), BG)
        res["segs"] = [s[:K.value // 8] for s in segs]
    return res


def rate_match(Tbslbrm, BG, Z, w, C_, F, Foffset, rv, E):
    w = np.ascontiguousarray(w, dtype=np.uint8)
    e = np.zeros(E, dtype=np.uint8)
    rc = lib().oracle_nr_rate_matching_ldpc(Tbslbrm, BG, Z, _p(w), _p(e), C_, F, Foffset, rv, E)
    return rc, e


def rate_match_rx(Tbslbrm, BG, Z, w, soft, C_, rv, clear, E, F, Foffset):
    w = np.ascontiguousarray(w, dtype=np.int16)
    soft = np.ascontiguousarray(soft, dtype=np.int16)
    rc = lib().oracle_nr_rate_matching_ldpc_rx(Tbslbrm, BG, Z, _p(w), _p(soft), C_, rv, clear, E, F, Foffset)
    return rc, w


def interleave(E, Qm, e):
    e = np.ascontiguousarray(e, dtype=np.uint8)
    f = np.zeros(E, dtype=np.uint8)
    lib().oracle_nr_interleaving_ldpc(E, Qm, _p(e), _p(f))
    return f


def deinterleave(E, Qm, f):
    f = np.ascontiguousarray(f, dtype=np.int16)
    e = np.zeros(E, dtype=np.int16)
    lib().oracle_nr_deinterleaving_ldpc(E, Qm, _p(e), _p(f))
    return e


def get_R(rv, E, BG, Z, llrLen=0, rnd=0):
    ll = C.c_int(llrLen)
    r = lib().oracle_nr_get_R_ldpc_decoder(rv, E, BG, Z, C.byref(ll), rnd)
    return r, ll.value


def llr_prepack(d, BG, Z, K, F, ncols_R):
    d = np.ascontiguousarray(d, dtype=np.int16)
    l = np.zeros(ncols_R * Z, dtype=np.int8)
    lib().oracle_nr_llr_prepack(_p(d), _p(l), BG, Z, K, F, ncols_R)
    return l


def get_E(G, C_, Qm, Nl, r):
    return lib().oracle_nr_get_E(G, C_, Qm, Nl, r)


NR_MAX_PDSCH_TBS = 3824  # openair1/PHY/defs_nr_common.h:84


def len_with_crc(nseg, length):  # openair1/PHY/defs_gNB.h:224-229
    if nseg > 1:
        return (length + 24 + 24 * nseg) // nseg
    return length + (24 if length > NR_MAX_PDSCH_TBS else 16)


def crc_type(nseg, length):      # openair1/PHY/defs_gNB.h:230-235
    if nseg > 1:
        return CRC24_B
    return CRC24_A if length > NR_MAX_PDSCH_TBS else CRC16


def dlsch_encode(tb, payload):
    """The reference's TX chain for one transport block, composed from the oracle pieces exactly as
    nr_dlsch_encoding()/ldpc8blocks() compose theirs (openair1/PHY/NR_TRANSPORT/nr_dlsch_coding.c:145-404).
    tb: dict(A, G, BG, Qm, Nl, rv, tbslbrm). Returns uint8[G], one bit per byte."""
    A, BG = tb["A"], tb["BG"]
    a = np.concatenate([np.asarray(payload, np.uint8)[:A // 8], np.zeros(4, np.uint8)])
    if A > NR_MAX_PDSCH_TBS:
        c = crc("crc24a", a, A) >> 8
        a[A // 8:A // 8 + 3] = [(c >> 16) & 255, (c >> 8) & 255, c & 255]
        B = A + 24
    else:
        c = crc("crc16", a, A) >> 16
        a[A // 8:A // 8 + 2] = [(c >> 8) & 255, c & 255]
        B = A + 16
    s = segmentation(a, B, BG)
    Z, K, F, Cn = s["Z"], s["K"], s["F"], s["C"]
    out = []
    for r in range(Cn):
        d = encode(BG, Z, s["segs"][r], s["Kb"]).copy()
        if F:
            d[K - F - 2 * Z:K - 2 * Z] = 2                       # NR_NULL (nr_dlsch_coding.c:177-180)
        E = get_E(tb["G"], Cn, tb["Qm"], tb["Nl"], r)
        rc, e = rate_match(tb["tbslbrm"], BG, Z, d, Cn, F, K - F - 2 * Z, tb["rv"], E)
        assert rc == 0
        out.append(interleave(E, tb["Qm"], e))
    return np.concatenate(out)


def ulsch_decode(tb, llr, harq_d, max_iter=8, rnd=0, llrLen=0, vec=False):
    """The reference's RX chain for one transport block (nr_ulsch_decoding.c:122-470 + nr_postDecode).
    llr: int16[G]; harq_d: list of C int16 arrays (soft buffers, updated in place).
    Returns (payload uint8[A/8], ack, per-segment pass counts, llrLen)."""
    A, BG = tb["A"], tb["BG"]
    B = len_with_crc(1, A)
    s = segmentation(None, B, BG)
    Z, K, F, Cn = s["Z"], s["K"], s["F"], s["C"]
    b = np.zeros(B // 8 + 4, np.uint8)
    offset = r_off = 0
    iters, all_ok = [], True
    for r in range(Cn):
        E = get_E(tb["G"], Cn, tb["Qm"], tb["Nl"], r)
        R, llrLen = get_R(tb["rv"], E, BG, Z, llrLen, rnd)
        e = deinterleave(E, tb["Qm"], llr[r_off:r_off + E])
        rc, d = rate_match_rx(tb["tbslbrm"], BG, Z, harq_d[r], e, Cn, tb["rv"], 1 if rnd == 0 else 0, E, F, K - F - 2 * Z)
        assert rc == 0
        harq_d[r][:] = d
        l = llr_prepack(d, BG, Z, K, F, NCOLS[(BG, R)])
        n, out = decode(BG, Z, R, l, max_iter, OUT_BIT, True, len_with_crc(Cn, A), crc_type(Cn, A), vec=vec)
        iters.append(n)
        nb = K // 8 - F // 8 - (3 if Cn > 1 else 0)
        if n <= max_iter:
            b[offset:offset + nb] = out[:nb]
        else:
            all_ok = False
        offset += nb
        r_off += E
    crc_ok = True
    if Cn > 1:
        crc_ok = bool(check_crc(b, len_with_crc(1, A), crc_type(1, A)))
    return b[:A // 8], bool(all_ok and crc_ok), iters, llrLen


class OaiRng:
    """OAI's uniformrandom/gaussdouble stream (rangen_double.c), seedable like OAI_RNGSEED."""

    def __init__(self, seed):
        self.s = Rng()
        lib().oracle_randominit(C.byref(self.s), seed)

    def uniform(self):
        return lib().oracle_uniformrandom(C.byref(self.s))

    def gauss(self, mean=0.0, var=1.0):
        return lib().oracle_gaussdouble(C.byref(self.s), mean, var)

    def schsim_channel(self, f, sigma, qbits=8):
        """ulschsim.c:533-552 / dlschsim.c:527-543: int16 LLRs of the rate-matched bits f; also the uncoded error count"""
        f = np.ascontiguousarray(f, dtype=np.uint8)
        llr = np.zeros(f.size, dtype=np.int16)
        n_err = lib().oracle_schsim_channel(C.byref(self.s), _p(f), f.size, sigma, qbits, _p(llr))
        return llr, n_err

    def ldpctest_channel(self, coded, Zc, sigma, qbits=8, ncols=None):
        coded = np.ascontiguousarray(coded, dtype=np.uint8)
        llr = np.zeros(2 * Zc + coded.size, dtype=np.int8)
        lib().oracle_ldpctest_channel(C.byref(self.s), _p(coded), coded.size, Zc, sigma, qbits, _p(llr))
        return llr


def awgn_llr(rng, coded, Z, snr_db, n_cols_tx=None):
    """numpy-RNG version of ldpctest's channel (BPSK, sigma = 1/sqrt(2*SNR), quantize(sigma/16, y, 8))."""
    sigma = 1.0 / np.sqrt(2.0 * 10.0 ** (snr_db / 10.0))
    y = (1.0 - 2.0 * coded.astype(np.float64)) + sigma * rng.standard_normal(coded.size)
    q = np.clip(np.floor(y / (sigma / 16.0)), -128, 127).astype(np.int8)
    return np.concatenate([np.zeros(2 * Z, dtype=np.int8), q])
