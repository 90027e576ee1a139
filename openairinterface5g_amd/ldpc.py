"""Python host side of libldpc_hip.so -- mirrors the reference's LDPC plugin interface.

The reference exposes the codec as ``ldpc_interface_t {LDPCinit, LDPCshutdown, LDPCdecoder, LDPCencoder}``
(openair1/PHY/CODING/nrLDPC_extern.h:27-33) taking ``t_nrLDPC_dec_params`` / ``encoder_implemparams_t``
(nrLDPC_types.h:84-97, nrLDPC_defs.h:40-66).  This module binds the same four symbols plus the batched
entry points of include/nrLDPC_hip.h with ctypes; argument names and meaning follow the reference.

There is no fallback: if the shared library is missing or no GPU is usable, calls raise.
torch is used only as the owner of device memory and streams for the ``*_device`` calls.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libldpc_hip.so"

# e_nrLDPC_outMode (nrLDPC_types.h:75-79)
nrLDPC_outMode_BIT, nrLDPC_outMode_BITINT8, nrLDPC_outMode_LLRINT8 = 0, 1, 2
# coding_defs.h:33-36
CRC24_A, CRC24_B, CRC16, CRC8 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
MEM_HARQ_DEVICE, MEM_HARQ_LIBRARY = 2, 4   # OR-ed into mem for nrLDPC_hip_ulsch_decode: where the soft buffers live

NCOLS = {(1, 13): 68, (1, 23): 35, (1, 89): 27, (2, 15): 52, (2, 13): 32, (2, 23): 17}
LIFT_SIZES = sorted(a * (1 << j) for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * (1 << j) <= 384)

CHECK_CRC_T = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint8)


class t_nrLDPC_dec_params(C.Structure):
    """nrLDPC_types.h:84-97"""
    _fields_ = [("BG", C.c_uint8), ("Z", C.c_uint16), ("R", C.c_uint8), ("F", C.c_uint16), ("Qm", C.c_uint8),
                ("rv", C.c_uint8), ("numMaxIter", C.c_uint8), ("E", C.c_int), ("outMode", C.c_int),
                ("crc_type", C.c_int), ("check_crc", C.c_void_p), ("setCombIn", C.c_uint8)]


class encoder_implemparams_t(C.Structure):
    """nrLDPC_defs.h:40-66"""
    _fields_ = [("n_segments", C.c_uint), ("macro_num", C.c_uint), ("gen_code", C.c_ubyte),
                ("tinput", C.c_void_p), ("tprep", C.c_void_p), ("tparity", C.c_void_p), ("toutput", C.c_void_p),
                ("Kr", C.c_int), ("Kb", C.c_uint32), ("Zc", C.c_uint32), ("harq", C.c_void_p), ("BG", C.c_uint8),
                ("output", C.c_void_p), ("K", C.c_uint32), ("F", C.c_uint32), ("Qm", C.c_uint8), ("E", C.c_uint32),
                ("G", C.c_uint), ("rv", C.c_uint8)]


class time_stats_t(C.Structure):
    """common/utils/time_meas.h:61-74 (x86-64)"""
    _fields_ = [("in_", C.c_longlong), ("diff", C.c_longlong), ("p_time", C.c_longlong), ("diff_square", C.c_double),
                ("max", C.c_longlong), ("trials", C.c_int), ("meas_flag", C.c_int), ("meas_name", C.c_char_p),
                ("meas_index", C.c_int), ("meas_enabled", C.c_int), ("tpoolmsg", C.c_void_p), ("tstatptr", C.c_void_p)]


class t_nrLDPC_time_stats(C.Structure):
    """nrLDPC_types.h:115-127"""
    _fields_ = [(n, time_stats_t) for n in ("llr2llrProcBuf", "llr2CnProcBuf", "cnProc", "cnProcPc", "bnProcPc", "bnProc",
                                            "cn2bnProcBuf", "bn2cnProcBuf", "llrRes2llrOut", "llr2bit", "total")]


class decode_abort_t(C.Structure):
    """openair1/PHY/defs_common.h:998-1001 (pthread_mutex_t is 40 bytes on x86-64 glibc; all-zero = initialised)"""
    _fields_ = [("mutex_failure", C.c_uint64 * 5), ("failed", C.c_bool)]


class nrLDPC_hip_dec_batch_t(C.Structure):
    _fields_ = [("params", t_nrLDPC_dec_params), ("n_blocks", C.c_uint32), ("llr", C.c_void_p),
                ("llr_stride", C.c_uint32), ("out", C.c_void_p), ("out_stride", C.c_uint32), ("n_iter", C.c_void_p),
                ("mem", C.c_int32), ("stream", C.c_void_p), ("kernel", C.c_int32)]


class nrLDPC_hip_dec_job_t(C.Structure):
    _fields_ = [("params", t_nrLDPC_dec_params), ("llr", C.c_void_p), ("out", C.c_void_p)]


class nrLDPC_hip_enc_batch_t(C.Structure):
    _fields_ = [("BG", C.c_uint8), ("Zc", C.c_uint16), ("Kb", C.c_uint8), ("n_blocks", C.c_uint32),
                ("in_", C.c_void_p), ("in_stride", C.c_uint32), ("out", C.c_void_p), ("out_stride", C.c_uint32),
                ("mem", C.c_int32), ("stream", C.c_void_p)]


_user_predicates = []   # CFUNCTYPE objects of caller-supplied predicates must outlive the parameter blocks that hold them


def device_crc_pointer():
    """Address of the library's nrLDPC_hip_check_crc: in t_nrLDPC_dec_params.check_crc it selects the CRC stop that the GPU
    evaluates (as the host executable's own `check_crc` does); any other pointer is called on the host (nrLDPC_hip.h)."""
    return C.cast(load_library().nrLDPC_hip_check_crc, C.c_void_p).value

EXPORTS = ["LDPCinit", "LDPCshutdown", "LDPCdecoder", "LDPCencoder", "ldpc_checkbuildver", "ldpc_autoinit", "LDPCdecoder_batch", "LDPCencoder_batch",
           "LDPCdecoder_jobs", "nrLDPC_hip_checkbuildver", "nrLDPC_hip_check_crc",
           "nrLDPC_hip_num_llr", "nrLDPC_hip_out_bytes", "nrLDPC_hip_lds_bytes", "nrLDPC_hip_code_info", "nrLDPC_hip_last_error",
           "nrLDPC_hip_version", "nrLDPC_hip_server_stats"]

_lib = None


def load_library(path=None):
    """dlopen libldpc_hip.so (the analogue of load_LDPClib, nrLDPC_load.c:45-75). Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else Path(os.environ.get("NRLDPC_HIP_LIB", LIB_PATH))   # (env: A/B builds of the library)
    if not p.exists():
        raise RuntimeError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C {_PKG / 'csrc'}` -- there is no CPU fallback")
    L = C.CDLL(str(p), mode=os.RTLD_NOW | os.RTLD_GLOBAL)
    L.LDPCinit.restype = C.c_int32
    L.LDPCshutdown.restype = C.c_int32
    L.LDPCdecoder.restype = C.c_int32
    L.LDPCdecoder.argtypes = [C.POINTER(t_nrLDPC_dec_params), C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p]
    L.LDPCencoder.restype = C.c_int32
    L.LDPCencoder.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(encoder_implemparams_t)]
    L.LDPCdecoder_batch.restype = C.c_int32
    L.LDPCdecoder_batch.argtypes = [C.POINTER(nrLDPC_hip_dec_batch_t)]
    L.LDPCencoder_batch.restype = C.c_int32
    L.LDPCencoder_batch.argtypes = [C.POINTER(nrLDPC_hip_enc_batch_t)]
    for n in ("nrLDPC_hip_num_llr", "nrLDPC_hip_lds_bytes"):
        getattr(L, n).argtypes = [C.c_int] * 3
        getattr(L, n).restype = C.c_int32
    L.nrLDPC_hip_out_bytes.argtypes = [C.c_int] * 4
    L.nrLDPC_hip_out_bytes.restype = C.c_int32
    L.nrLDPC_hip_last_error.restype = C.c_char_p
    L.nrLDPC_hip_version.restype = C.c_char_p
    if path is None:
        _lib = L
    return L


def last_error():
    return load_library().nrLDPC_hip_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")


def LDPCinit():
    """nrLDPC_decoder.c:162.  Raises when no MI355X/HIP device is usable."""
    _check(load_library().LDPCinit(), "LDPCinit")
    return 0


def LDPCshutdown():
    return load_library().LDPCshutdown()


def server_stats():
    """State of the resident submission path behind LDPCdecoder / LDPCencoder (nrLDPC_hip_server_stats)."""
    L = load_library()
    a = (C.c_int64 * 8)()
    L.nrLDPC_hip_server_stats.argtypes = [C.POINTER(C.c_int64)]
    L.nrLDPC_hip_server_stats(a)
    return dict(status=int(a[0]), slots=int(a[1]), launches=int(a[2]), calls=int(a[3]), gpu_stage_ns=int(a[4]),
                gpu_decode_ns=int(a[5]), host_wait_ns=int(a[6]), host_call_ns=int(a[7]))


def num_llr(BG, Z, R):
    return NCOLS[(BG, R)] * Z


def code_info(BG, Z, R):
    """Shape of a code and of the decoder launch that serves it (nrLDPC_hip_code_info)."""
    L = load_library()
    a = (C.c_int32 * 8)()
    L.nrLDPC_hip_code_info.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    if L.nrLDPC_hip_code_info(BG, Z, R, a) != 0:
        raise ValueError("invalid code")
    return dict(nrows=a[0], ncols=a[1], nedges=a[2], kernel="fast" if a[3] else "generic", threads=a[4],
                lds_kib=round(a[5] / 1024, 1), cn_tasks=a[6], bn_tasks=a[7])


def out_bytes(BG, Z, R, outMode=nrLDPC_outMode_BIT):
    n = num_llr(BG, Z, R)
    return ((n + 31) // 32) * 4 if outMode == nrLDPC_outMode_BIT else n


def make_dec_params(BG, Z, R, numMaxIter=8, outMode=nrLDPC_outMode_BIT, check_crc=False, E=0, crc_type=CRC24_B):
    p = t_nrLDPC_dec_params(BG=BG, Z=Z, R=R, numMaxIter=numMaxIter, outMode=outMode, E=E, crc_type=crc_type)
    if callable(check_crc):      # a caller's own predicate (decoded_bytes_ptr, n, crc_type) -> int: called on the host
        cb = CHECK_CRC_T(check_crc)
        p._check_crc_keepalive = cb      # lives as long as the parameter block that points at it ...
        _user_predicates.append(cb)      # ... and a little longer for by-value copies of the block (job arrays)
        del _user_predicates[:-64]
        p.check_crc = C.cast(cb, C.c_void_p)
    elif check_crc:
        p.check_crc = device_crc_pointer()
    return p


def LDPCdecoder(p_decParams, p_llr, p_out=None, ab=None, harq_pid=0, ulsch_id=0, C_=0, profiler=None):
    """One code block through the reference's own entry point (nrLDPC_decoder.c:172).
    p_llr: int8[ncols*Z]; returns (numIter, p_out).  profiler: optional t_nrLDPC_time_stats."""
    L = load_library()
    p_llr = np.ascontiguousarray(p_llr, dtype=np.int8)
    nb = out_bytes(p_decParams.BG, p_decParams.Z, p_decParams.R, p_decParams.outMode)
    if p_out is None:
        p_out = np.zeros(nb, dtype=np.uint8)
    n = L.LDPCdecoder(C.byref(p_decParams), harq_pid, ulsch_id, C_, p_llr.ctypes.data, p_out.ctypes.data,
                      C.addressof(profiler) if profiler is not None else None,
                      C.addressof(ab) if ab is not None else None)
    if n < 0:
        raise RuntimeError(f"LDPCdecoder failed: {last_error()}")
    return n, p_out


def raw_decoder_call(p_decParams):
    """(call, keep): `call(llr_address, out_address) -> numIter` is the bare C entry point LDPCdecoder with everything but
    the two buffer addresses bound beforehand -- for callers that time the call itself, as ldpctest does with
    start_meas / stop_meas around `LDPCdecoder(...)` (ldpctest.c:329-334), without this module's per-call marshalling."""
    L = load_library()
    proto = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    fn = proto(("LDPCdecoder", L))   # its own prototype object: L.LDPCdecoder keeps the typed one
    pref = C.cast(C.pointer(p_decParams), C.c_void_p)

    def call(llr_address, out_address):
        return fn(pref, 0, 0, 0, llr_address, out_address, None, None)
    return call, (p_decParams, pref)


def LDPCencoder(inputs, BG, Zc, Kb=None, n_segments=None, macro_num=0, meters=None, block_length=None):
    """Up to 8 segments through the reference entry point (ldpc_encoder_optim8segmulti.c:46).
    inputs: list of uint8[K/8]; returns list of uint8[(66|50)*Zc] (one bit per byte) for ALL n_segments
    (entries outside this macro group are left zero).  block_length: ldpctest's -l when it is shorter than K = 22|10 * Zc
    (impp->K; the code word is then cut to 3|5 * block_length bytes, ldpc_encoder.c:82-92,248-251)."""
    L = load_library()
    kbf = 22 if BG == 1 else 10
    K = kbf * Zc if block_length is None else block_length
    n_segments = len(inputs) if n_segments is None else n_segments
    ins = [np.ascontiguousarray(np.concatenate([np.asarray(i, dtype=np.uint8), np.zeros(8, np.uint8)])) for i in inputs]
    outs = [np.zeros(68 * 384, dtype=np.uint8) for _ in range(n_segments)]
    ip = (C.c_void_p * n_segments)(*[a.ctypes.data for a in ins])
    op = (C.c_void_p * n_segments)(*[a.ctypes.data for a in outs])
    impp = encoder_implemparams_t(n_segments=n_segments, macro_num=macro_num, gen_code=0, Kr=K,
                                  Kb=kbf if Kb is None else Kb, Zc=Zc, BG=BG, K=K, E=K)
    if meters is not None:      # four time_stats_t: tinput, tprep, tparity, toutput (nrLDPC_defs.h:44-47)
        impp.tinput, impp.tprep, impp.tparity, impp.toutput = (C.addressof(m) for m in meters)
    rc = L.LDPCencoder(ip, op, C.byref(impp))
    _check(rc, "LDPCencoder")
    if rc != 0:     # the default reference library's value (ldpc_encoder_optim8segmulti.c:213), not ldpc_encoder.c:251's length
        raise RuntimeError(f"LDPCencoder returned {rc}, expected 0")
    N = (66 if BG == 1 else 50) * Zc if block_length is None else (3 if BG == 1 else 5) * block_length
    return [o[:N] for o in outs]


# ---- the reference's offload plugin slot (ldpc_interface_offload, "_t2"): libldpc_hip_t2.so ------------------------------
EXPORTS += ["nrLDPC_hip_offload_init", "nrLDPC_hip_offload_decoder", "nrLDPC_hip_offload_encoder"]
T2_EXPORTS = ["LDPCinit", "LDPCshutdown", "LDPCdecoder", "LDPCencoder", "ldpc_checkbuildver"]
_t2 = None


def load_offload_library():
    """dlopen libldpc_hip_t2.so the way load_LDPClib("_t2", &ldpc_interface_offload) does (nr_init.c:138-139,
    nrLDPC_load.c:45-75) and run its LDPCinit.  Raises if the library is missing or no GPU is usable."""
    global _t2
    if _t2 is None:
        load_library()
        path = Path(os.environ.get("NRLDPC_HIP_T2_LIB", Path(LIB_PATH).parent / "libldpc_hip_t2.so"))
        if not path.exists():
            raise RuntimeError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
        L = C.CDLL(str(path))
        for name in T2_EXPORTS:
            getattr(L, name)
        L.LDPCdecoder.restype = C.c_int32
        L.LDPCdecoder.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.LDPCencoder.restype = C.c_int32
        L.LDPCencoder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if L.LDPCinit() != 0:
            raise RuntimeError(f"offload LDPCinit failed: {last_error()}")
        _t2 = L
    return _t2


def offload_decoder(BG, Z, R, llr, Qm, rv, F, setCombIn, ulsch_id=0, r=0, harq_pid=0, numMaxIter=8):
    """One segment through the offload slot's LDPCdecoder (nrLDPC_decoder_offload.c:1036, caller nr_ulsch_decoding.c:225-268):
    llr = the E received soft values as int8 in transmission order.  Returns (passes, decoded bytes uint8[ceil(K/8)])."""
    L = load_offload_library()
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    p = t_nrLDPC_dec_params(BG=BG, Z=Z, R=R, F=F, Qm=Qm, rv=rv, numMaxIter=numMaxIter, E=llr.size, setCombIn=1 if setCombIn else 0)
    K = (22 if BG == 1 else 10) * Z
    out = np.zeros((K + 7) // 8 + 64, dtype=np.uint8)
    n = L.LDPCdecoder(C.addressof(p), harq_pid, ulsch_id, r, llr.ctypes.data, out.ctypes.data, None, None)
    if n < 0:
        raise RuntimeError(f"offload LDPCdecoder failed: {last_error()}")
    assert not out[(K + 7) // 8:].any()
    return n, out[:(K + 7) // 8]


def offload_encoder(BG, Zc, segment, F, E, Qm, rv, Kb=None):
    """One segment through the offload slot's LDPCencoder (nrLDPC_decoder_offload.c:1094, caller nr_dlsch_coding.c:366-383):
    segment = its K - F bits (uint8, MSB first).  Returns the E rate-matched, interleaved bits, one per byte."""
    L = load_offload_library()
    K = (22 if BG == 1 else 10) * Zc
    seg = np.ascontiguousarray(np.asarray(segment, dtype=np.uint8)[:(K - F) // 8])
    out = np.full(E + 64, 0xEE, dtype=np.uint8)
    ip = (C.c_void_p * 1)(seg.ctypes.data)
    op = (C.c_void_p * 1)(out.ctypes.data)
    impp = encoder_implemparams_t(n_segments=1, macro_num=0, Kr=K, Kb=(22 if BG == 1 else 10) if Kb is None else Kb, Zc=Zc, BG=BG,
                                  K=K, F=F, Qm=Qm, E=E, rv=rv)
    rc = L.LDPCencoder(C.addressof(ip), C.addressof(op), C.addressof(impp))
    if rc != 0:
        raise RuntimeError(f"offload LDPCencoder failed: {last_error()}")
    assert (out[E:] == 0xEE).all()
    return out[:E]


def decode_batch_host(BG, Z, R, llr, numMaxIter=8, outMode=nrLDPC_outMode_BIT, check_crc=False, E=0,
                      crc_type=CRC24_B, out=None, kernel=0):
    """llr: int8[n_blocks, >= ncols*Z] host array. Returns (n_iter int32[n], out uint8[n, out_bytes])."""
    L = load_library()
    llr = np.ascontiguousarray(llr, dtype=np.int8)
    n = llr.shape[0]
    ob = out_bytes(BG, Z, R, outMode)
    stride = (ob + 3) // 4 * 4
    if out is None:
        out = np.zeros((n, stride), dtype=np.uint8)
    it = np.zeros(n, dtype=np.int32)
    assert llr.ndim == 2 and out.ndim == 2 and out.flags.c_contiguous and out.shape[0] == n
    # (strides of a length-1 axis are arbitrary in numpy, so derive the row pitch from the shape)
    b = nrLDPC_hip_dec_batch_t(params=make_dec_params(BG, Z, R, numMaxIter, outMode, check_crc, E, crc_type),
                               n_blocks=n, llr=llr.ctypes.data, llr_stride=llr.shape[1], out=out.ctypes.data,
                               out_stride=out.shape[1], n_iter=it.ctypes.data, mem=MEM_HOST, stream=None, kernel=kernel)
    _check(L.LDPCdecoder_batch(C.byref(b)), "LDPCdecoder_batch")
    return it, out[:, :ob]


def decode_batch_device(BG, Z, R, llr, out, n_iter, numMaxIter=8, outMode=nrLDPC_outMode_BIT, check_crc=False, E=0,
                        crc_type=CRC24_B, stream=None, kernel=0):
    """Enqueue a decode of device-resident blocks (torch tensors: llr int8[n, stride], out uint8[n, stride'],
    n_iter int32[n]) on `stream` (default: torch's current stream). Asynchronous."""
    import torch
    L = load_library()
    assert llr.is_cuda and out.is_cuda and n_iter.is_cuda and llr.dtype == torch.int8 and n_iter.dtype == torch.int32
    assert llr.stride(1) == 1 and out.stride(1) == 1 and n_iter.is_contiguous()
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    b = nrLDPC_hip_dec_batch_t(params=make_dec_params(BG, Z, R, numMaxIter, outMode, check_crc, E, crc_type),
                               n_blocks=llr.shape[0], llr=llr.data_ptr(), llr_stride=llr.stride(0) * llr.element_size(),
                               out=out.data_ptr(), out_stride=out.stride(0) * out.element_size(),
                               n_iter=n_iter.data_ptr(), mem=MEM_DEVICE, stream=s, kernel=kernel)
    _check(L.LDPCdecoder_batch(C.byref(b)), "LDPCdecoder_batch")


class PreparedDecJobs:
    """A MIXED batch of code blocks (LDPCdecoder_jobs): blocks = list of dict(BG, Z, R, llr=torch int8 row on the GPU,
    out=torch uint8 row, numMaxIter=8, E=0, crc_type=CRC24_B); all with one outMode / stop mode.  Marshalled once,
    submitted many times; decode() only enqueues on the stream.  n_iter: torch int32 [len(blocks)]."""

    def __init__(self, blocks, n_iter, outMode=nrLDPC_outMode_BIT, check_crc=False, stream=None):
        import torch
        self._lib = load_library()
        self._lib.LDPCdecoder_jobs.argtypes = [C.POINTER(nrLDPC_hip_dec_job_t), C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]
        self._lib.LDPCdecoder_jobs.restype = C.c_int32
        self._keep = (blocks, n_iter)
        self.arr = (nrLDPC_hip_dec_job_t * len(blocks))()
        for i, b in enumerate(blocks):
            assert b["llr"].is_cuda and b["out"].is_cuda and b["llr"].is_contiguous() and b["out"].is_contiguous()
            self.arr[i].params = make_dec_params(b["BG"], b["Z"], b["R"], b.get("numMaxIter", 8), outMode, check_crc, b.get("E", 0),
                                                 b.get("crc_type", CRC24_B))
            self.arr[i].llr = b["llr"].data_ptr()
            self.arr[i].out = b["out"].data_ptr()
        assert n_iter.is_cuda and n_iter.dtype == torch.int32 and n_iter.numel() >= len(blocks)
        self.n, self.n_iter = len(blocks), n_iter
        self.stream = torch.cuda.current_stream().cuda_stream if stream is None else stream

    def decode(self):
        _check(self._lib.LDPCdecoder_jobs(self.arr, self.n, self.n_iter.data_ptr(), MEM_DEVICE, self.stream), "LDPCdecoder_jobs")


def encode_batch_host(BG, Zc, info, Kb=None):
    """info: uint8[n_blocks, >= K/8] host array (MSB first). Returns uint8[n_blocks, (66|50)*Zc], one bit per byte."""
    L = load_library()
    info = np.ascontiguousarray(info, dtype=np.uint8)
    assert info.ndim == 2
    n = info.shape[0]
    N = (66 if BG == 1 else 50) * Zc
    out = np.zeros((n, N), dtype=np.uint8)
    b = nrLDPC_hip_enc_batch_t(BG=BG, Zc=Zc, Kb=(22 if BG == 1 else 10) if Kb is None else Kb, n_blocks=n,
                               in_=info.ctypes.data, in_stride=info.shape[1], out=out.ctypes.data,
                               out_stride=out.shape[1], mem=MEM_HOST, stream=None)
    _check(L.LDPCencoder_batch(C.byref(b)), "LDPCencoder_batch")
    return out


def encode_batch_device(BG, Zc, info, out, Kb=None, stream=None):
    """info: torch uint8[n, >= K/8], out: torch uint8[n, >= (66|50)*Zc], both on the GPU. Asynchronous."""
    import torch
    L = load_library()
    assert info.is_cuda and out.is_cuda and info.stride(1) == 1 and out.stride(1) == 1
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    b = nrLDPC_hip_enc_batch_t(BG=BG, Zc=Zc, Kb=(22 if BG == 1 else 10) if Kb is None else Kb, n_blocks=info.shape[0],
                               in_=info.data_ptr(), in_stride=info.stride(0), out=out.data_ptr(),
                               out_stride=out.stride(0), mem=MEM_DEVICE, stream=s)
    _check(L.LDPCencoder_batch(C.byref(b)), "LDPCencoder_batch")


# ---------------------------------------------------------------------------------------------------------
# Transport-block chain (include/nrLDPC_hip.h: nrLDPC_hip_dlsch_encode / nrLDPC_hip_ulsch_decode)
# ---------------------------------------------------------------------------------------------------------
class nrLDPC_hip_tb_t(C.Structure):
    _fields_ = [("A", C.c_uint32), ("G", C.c_uint32), ("tbslbrm", C.c_uint32), ("BG", C.c_uint8), ("Qm", C.c_uint8),
                ("Nl", C.c_uint8), ("rv", C.c_uint8), ("numMaxIter", C.c_uint8), ("round", C.c_uint8),
                ("llrLen", C.c_int32), ("payload_off", C.c_uint64), ("coded_off", C.c_uint64), ("harq_off", C.c_uint64)]


class nrLDPC_hip_tb_batch_t(C.Structure):
    _fields_ = [("n_tb", C.c_uint32), ("tb", C.POINTER(nrLDPC_hip_tb_t)), ("payload", C.c_void_p), ("coded", C.c_void_p),
                ("harq", C.c_void_p), ("harq_stride", C.c_uint32), ("ack", C.c_void_p), ("iter_max", C.c_void_p),
                ("mem", C.c_int32), ("stream", C.c_void_p)]


EXPORTS += ["nrLDPC_hip_dlsch_encode", "nrLDPC_hip_ulsch_decode", "nrLDPC_hip_segmentation", "nrLDPC_hip_get_E",
            "nrLDPC_hip_get_R_ldpc_decoder", "nrLDPC_hip_harq_release", "nrLDPC_hip_harq_release_all", "nrLDPC_hip_harq_read",
            "nrLDPC_hip_host_alloc", "nrLDPC_hip_host_free", "nrLDPC_hip_host_register", "nrLDPC_hip_host_unregister",
            "nrLDPC_hip_chain_timing", "nrLDPC_hip_ulsch_decoder_columns"]
HARQ_STRIDE = 66 * 384


def _tb_lib():
    L = load_library()
    L.nrLDPC_hip_dlsch_encode.argtypes = [C.POINTER(nrLDPC_hip_tb_batch_t)]
    L.nrLDPC_hip_ulsch_decode.argtypes = [C.POINTER(nrLDPC_hip_tb_batch_t)]
    L.nrLDPC_hip_segmentation.argtypes = [C.c_uint32, C.c_uint8] + [C.POINTER(C.c_uint32)] * 4
    L.nrLDPC_hip_segmentation.restype = C.c_int32
    L.nrLDPC_hip_get_E.argtypes = [C.c_uint32] * 5
    L.nrLDPC_hip_get_E.restype = C.c_uint32
    L.nrLDPC_hip_get_R_ldpc_decoder.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_int32), C.c_int32]
    L.nrLDPC_hip_get_R_ldpc_decoder.restype = C.c_int32
    if hasattr(L, "nrLDPC_hip_ulsch_decoder_columns"):
        L.nrLDPC_hip_ulsch_decoder_columns.argtypes = [C.c_int32] + [C.c_uint32] * 5 + [C.c_int32, C.c_uint32, C.c_int32, C.c_int32]
        L.nrLDPC_hip_ulsch_decoder_columns.restype = C.c_int32
    if hasattr(L, "nrLDPC_hip_harq_release"):       # (absent from older builds of the library loaded through NRLDPC_HIP_LIB for A/B runs)
        L.nrLDPC_hip_harq_release.argtypes = [C.c_uint64]
        L.nrLDPC_hip_harq_read.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
        L.nrLDPC_hip_host_alloc.argtypes = [C.c_uint64]
        L.nrLDPC_hip_host_alloc.restype = C.c_void_p
        L.nrLDPC_hip_host_free.argtypes = [C.c_void_p]
        L.nrLDPC_hip_host_free.restype = None
        L.nrLDPC_hip_host_register.argtypes = [C.c_void_p, C.c_uint64]
        L.nrLDPC_hip_host_unregister.argtypes = [C.c_void_p]
        L.nrLDPC_hip_chain_timing.argtypes = [C.c_int32, C.c_void_p]
    return L


class PinnedArray:
    """A numpy array in page-locked host memory from nrLDPC_hip_host_alloc (what a C caller without the HIP runtime
    uses): the GPU reads it in place.  Keep the object alive while `a` is in use."""

    def __init__(self, shape, dtype):
        self._lib = _tb_lib()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = self._lib.nrLDPC_hip_host_alloc(max(n, 1))
        if not self._p:
            raise RuntimeError("nrLDPC_hip_host_alloc failed: " + self._lib.nrLDPC_hip_last_error().decode())
        buf = (C.c_uint8 * max(n, 1)).from_address(self._p)
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.nrLDPC_hip_host_free(self._p)
            self._p = None


def chain_timing(enable=True, read=False):
    """nrLDPC_hip_chain_timing: switch the per-stage HIP events of this thread's UL-SCH calls on / off; read=True returns
    the last recorded call's (de-matching, decoder / fused kernel, reassembly + verdict, sum) in microseconds."""
    L = _tb_lib()
    out = (C.c_float * 4)()
    _check(L.nrLDPC_hip_chain_timing(int(bool(enable)), out if read else None), "nrLDPC_hip_chain_timing")
    return tuple(out) if read else None


def harq_read(harq_id, n, first=0):
    """int16[n] of the soft buffers the library keeps for block `harq_id` (MEM_HARQ_LIBRARY); after a device-memory call:
    synchronise its stream first."""
    out = np.zeros(n, np.int16)
    _check(_tb_lib().nrLDPC_hip_harq_read(harq_id, out.ctypes.data, first, n), "nrLDPC_hip_harq_read")
    return out


def harq_release(harq_id=None):
    L = _tb_lib()
    return L.nrLDPC_hip_harq_release_all() if harq_id is None else L.nrLDPC_hip_harq_release(harq_id)


def nr_segmentation(B, BG):
    """Parameter part of nr_segmentation (nr_segmentation.c:32-140): dict(C, K, Z, F, Kb) or None."""
    L = _tb_lib()
    Cn, K, Z, F = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    kb = L.nrLDPC_hip_segmentation(B, BG, C.byref(Cn), C.byref(K), C.byref(Z), C.byref(F))
    return None if kb < 0 else dict(C=Cn.value, K=K.value, Z=Z.value, F=F.value, Kb=kb)


def nr_get_E(G, C_, Qm, Nl, r):
    return _tb_lib().nrLDPC_hip_get_E(G, C_, Qm, Nl, r)


def nr_get_R_ldpc_decoder(rv, E, BG, Z, llrLen, rnd):
    ll = C.c_int32(llrLen)
    R = _tb_lib().nrLDPC_hip_get_R_ldpc_decoder(rv, E, BG, Z, C.byref(ll), rnd)
    return R, ll.value


def ulsch_decoder_columns(BG, Zc, C_, F, K, tbslbrm, rv, E, rnd, R):
    """columns of the code graph nrLDPC_hip_ulsch_decode() runs a segment on (nrLDPC_hip.h)"""
    return _tb_lib().nrLDPC_hip_ulsch_decoder_columns(BG, Zc, C_, F, K, tbslbrm, rv, E, rnd, R)


def _tb_array(tbs, offs_payload, offs_coded, offs_harq, numMaxIter=8):
    arr = (nrLDPC_hip_tb_t * len(tbs))()
    for i, t in enumerate(tbs):
        arr[i] = nrLDPC_hip_tb_t(A=t["A"], G=t["G"], tbslbrm=t.get("tbslbrm", 0), BG=t["BG"], Qm=t["Qm"], Nl=t["Nl"],
                                 rv=t.get("rv", 0), numMaxIter=t.get("numMaxIter", numMaxIter), round=t.get("round", 0),
                                 llrLen=t.get("llrLen", 0), payload_off=offs_payload[i], coded_off=offs_coded[i],
                                 harq_off=offs_harq[i] if offs_harq is not None else 0)
    return arr


def dlsch_encode_host(tbs, payloads):
    """TX chain for a batch of transport blocks, host buffers.  tbs: list of dict(A, G, BG, Qm, Nl, rv, tbslbrm);
    payloads: list of uint8[A/8].  Returns list of uint8[G] (one bit per byte, the reference's `output`)."""
    L = _tb_lib()
    po = np.cumsum([0] + [(t["A"] // 8 + 15) // 16 * 16 for t in tbs])
    co = np.cumsum([0] + [(t["G"] + 15) // 16 * 16 for t in tbs])
    pay = np.zeros(int(po[-1]) + 16, np.uint8)
    for i, p in enumerate(payloads):
        pay[po[i]:po[i] + tbs[i]["A"] // 8] = np.asarray(p, np.uint8)[:tbs[i]["A"] // 8]
    coded = np.zeros(int(co[-1]) + 16, np.uint8)
    arr = _tb_array(tbs, po, co, None)
    b = nrLDPC_hip_tb_batch_t(n_tb=len(tbs), tb=arr, payload=pay.ctypes.data, coded=coded.ctypes.data, harq=None,
                              harq_stride=0, ack=None, iter_max=None, mem=MEM_HOST, stream=None)
    _check(L.nrLDPC_hip_dlsch_encode(C.byref(b)), "nrLDPC_hip_dlsch_encode")
    return [coded[co[i]:co[i] + tbs[i]["G"]].copy() for i in range(len(tbs))]


def ulsch_decode_host(tbs, llrs, harq, numMaxIter=8, harq_off=None, harq_ids=None, pinned=False, payload_off=None):
    """RX chain for a batch of transport blocks, host buffers.  llrs: list of int16[G]; harq: int16 array
    [total segments, HARQ_STRIDE] holding the soft buffers of all TBs back to back (updated in place); each tb dict
    may carry 'round' and 'llrLen' ('llrLen' is updated).  harq_off: explicit int16 offsets of the TBs' soft buffers in
    `harq` (any order, gaps allowed) instead of back to back.  Returns (payloads, ack bool[n], iter_max int32[n]).
    Soft buffers on the GPU while everything else stays on the host (what nr_ulsch_decoding's caller needs: the LLRs of a
    slot arrive in host memory, d[r] persists per HARQ process): `harq` a torch int16 CUDA tensor (MEM_HARQ_DEVICE), or
    harq=None with harq_ids = one id per TB (MEM_HARQ_LIBRARY: the library keeps them; harq_read() looks at them).
    pinned: the LLR array goes into page-locked memory (nrLDPC_hip_host_alloc) and is pulled by the GPU in place.
    payload_off: explicit byte offsets of the TBs' payloads in the call's payload array (any order) instead of back to back."""
    L = _tb_lib()
    n = len(tbs)
    if harq_ids is not None or not isinstance(harq, np.ndarray):
        return _ulsch_decode_host_resident(L, tbs, llrs, harq, numMaxIter, harq_off, harq_ids, pinned, payload_off)
    po = np.cumsum([0] + [(t["A"] // 8 + 15) // 16 * 16 for t in tbs])
    pay_total = int(po[-1])
    if payload_off is not None:
        po = list(payload_off)
    co = np.cumsum([0] + [(t["G"] + 7) // 8 * 8 for t in tbs])
    segs = [nr_segmentation(t["A"] + (24 if t["A"] > 3824 else 16), t["BG"])["C"] for t in tbs]
    ho = np.cumsum([0] + [c * HARQ_STRIDE for c in segs])
    if harq_off is not None:
        ho = list(harq_off)
        assert len(ho) == n and all(o + c * HARQ_STRIDE <= harq.size for o, c in zip(ho, segs))
    assert harq.dtype == np.int16 and harq.flags.c_contiguous and harq.size >= max(o + c * HARQ_STRIDE for o, c in zip(ho, segs))
    pay = np.zeros(pay_total + 16, np.uint8)
    keep = PinnedArray(int(co[-1]) + 16, np.int16) if pinned else None
    llr = keep.a if pinned else np.zeros(int(co[-1]) + 16, np.int16)
    llr[:] = 0
    for i, x in enumerate(llrs):
        llr[co[i]:co[i] + tbs[i]["G"]] = x
    ack = np.zeros(n, np.uint8)
    itm = np.zeros(n, np.int32)
    arr = _tb_array(tbs, po, co, ho, numMaxIter)
    b = nrLDPC_hip_tb_batch_t(n_tb=n, tb=arr, payload=pay.ctypes.data, coded=llr.ctypes.data, harq=harq.ctypes.data,
                              harq_stride=HARQ_STRIDE, ack=ack.ctypes.data, iter_max=itm.ctypes.data, mem=MEM_HOST,
                              stream=None)
    _check(L.nrLDPC_hip_ulsch_decode(C.byref(b)), "nrLDPC_hip_ulsch_decode")
    for i, t in enumerate(tbs):
        t["llrLen"] = arr[i].llrLen
    return [pay[po[i]:po[i] + tbs[i]["A"] // 8].copy() for i in range(n)], ack.astype(bool), itm


def _ulsch_decode_host_resident(L, tbs, llrs, harq, numMaxIter, harq_off, harq_ids, pinned, payload_off=None):
    """host payload / LLRs / verdicts with the soft buffers resident on the GPU (see ulsch_decode_host)"""
    n = len(tbs)
    po = np.cumsum([0] + [(t["A"] // 8 + 15) // 16 * 16 for t in tbs])
    pay_total = int(po[-1])
    if payload_off is not None:
        po = list(payload_off)
    co = np.cumsum([0] + [(t["G"] + 7) // 8 * 8 for t in tbs])
    segs = [nr_segmentation(t["A"] + (24 if t["A"] > 3824 else 16), t["BG"])["C"] for t in tbs]
    if harq_ids is not None:
        assert harq is None and len(harq_ids) == n
        ho, hp, mem = list(harq_ids), None, MEM_HOST | MEM_HARQ_LIBRARY
    else:
        ho = list(harq_off) if harq_off is not None else list(np.cumsum([0] + [c * HARQ_STRIDE for c in segs])[:n])
        assert harq.is_cuda and harq.is_contiguous() and harq.numel() >= max(o + c * HARQ_STRIDE for o, c in zip(ho, segs))
        hp, mem = harq.data_ptr(), MEM_HOST | MEM_HARQ_DEVICE
    pay = np.zeros(pay_total + 16, np.uint8)
    keep = PinnedArray(int(co[-1]) + 16, np.int16) if pinned else None
    llr = keep.a if pinned else np.zeros(int(co[-1]) + 16, np.int16)
    llr[:] = 0
    for i, x in enumerate(llrs):
        llr[co[i]:co[i] + tbs[i]["G"]] = x
    ack = np.zeros(n, np.uint8)
    itm = np.zeros(n, np.int32)
    arr = _tb_array(tbs, po, co, ho, numMaxIter)
    b = nrLDPC_hip_tb_batch_t(n_tb=n, tb=arr, payload=pay.ctypes.data, coded=llr.ctypes.data, harq=hp,
                              harq_stride=HARQ_STRIDE, ack=ack.ctypes.data, iter_max=itm.ctypes.data, mem=mem, stream=None)
    _check(L.nrLDPC_hip_ulsch_decode(C.byref(b)), "nrLDPC_hip_ulsch_decode")
    for i, t in enumerate(tbs):
        t["llrLen"] = arr[i].llrLen
    return [pay[po[i]:po[i] + tbs[i]["A"] // 8].copy() for i in range(n)], ack.astype(bool), itm


def tb_layout(tbs):
    """Offsets used by the *_device transport-block calls: payload bytes, coded elements, harq int16 elements."""
    po = np.cumsum([0] + [(t["A"] // 8 + 15) // 16 * 16 for t in tbs])
    co = np.cumsum([0] + [(t["G"] + 15) // 16 * 16 for t in tbs])
    segs = [nr_segmentation(t["A"] + (24 if t["A"] > 3824 else 16), t["BG"])["C"] for t in tbs]
    ho = np.cumsum([0] + [c * HARQ_STRIDE for c in segs])
    return po, co, ho, segs


def dlsch_encode_device(tbs, payload, coded, stream=None):
    """payload: torch uint8 [>= tb_layout po[-1]] on the GPU, coded: torch uint8 [>= co[-1]] (out). Asynchronous."""
    import torch
    L = _tb_lib()
    po, co, _, _ = tb_layout(tbs)
    assert payload.is_cuda and coded.is_cuda and payload.numel() >= po[-1] and coded.numel() >= co[-1]
    arr = _tb_array(tbs, po, co, None)
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    b = nrLDPC_hip_tb_batch_t(n_tb=len(tbs), tb=arr, payload=payload.data_ptr(), coded=coded.data_ptr(), harq=None,
                              harq_stride=0, ack=None, iter_max=None, mem=MEM_DEVICE, stream=s)
    _check(L.nrLDPC_hip_dlsch_encode(C.byref(b)), "nrLDPC_hip_dlsch_encode")


def _ptr(x):
    """address of a torch tensor / numpy array / PinnedArray, None for None"""
    if x is None:
        return None
    if isinstance(x, PinnedArray):
        return x.a.ctypes.data
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


class PreparedTbBatch:
    """A transport-block batch descriptor built once and submitted many times (slot after slot with the same
    allocation): keeps the ctypes marshalling out of the caller's per-slot path.  `encode()` / `decode()` are the bare
    C calls nrLDPC_hip_dlsch_encode / nrLDPC_hip_ulsch_decode (asynchronous on the batch's stream with device buffers,
    synchronous with host buffers).  Buffers: torch CUDA tensors (mem = MEM_DEVICE, the default), or numpy arrays /
    PinnedArray objects with mem = MEM_HOST, optionally | MEM_HARQ_DEVICE (harq a CUDA tensor) or | MEM_HARQ_LIBRARY
    (harq None, harq_ids = one id per transport block)."""

    def __init__(self, tbs, payload, coded_or_llr, harq=None, ack=None, iter_max=None, numMaxIter=8, stream=None, mem=MEM_DEVICE,
                 harq_ids=None):
        self._lib = _tb_lib()
        po, co, ho, _ = tb_layout(tbs)
        self._keep = (payload, coded_or_llr, harq, ack, iter_max)
        if harq_ids is not None:
            assert mem & MEM_HARQ_LIBRARY and len(harq_ids) == len(tbs)
            ho = list(harq_ids)
        self.arr = _tb_array(tbs, po, co, ho if (harq is not None or harq_ids is not None) else None, numMaxIter)
        s = None
        if mem & MEM_DEVICE:
            import torch
            s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.batch = nrLDPC_hip_tb_batch_t(
            n_tb=len(tbs), tb=self.arr, payload=_ptr(payload), coded=_ptr(coded_or_llr),
            harq=_ptr(harq), harq_stride=0 if (harq is None and harq_ids is None) else HARQ_STRIDE,
            ack=_ptr(ack), iter_max=_ptr(iter_max), mem=mem, stream=s)

    def encode(self):
        _check(self._lib.nrLDPC_hip_dlsch_encode(C.byref(self.batch)), "nrLDPC_hip_dlsch_encode")

    def decode(self):
        _check(self._lib.nrLDPC_hip_ulsch_decode(C.byref(self.batch)), "nrLDPC_hip_ulsch_decode")

    # HIP graphs: capture on the batch's own stream (torch.cuda.graph(g, stream=<the stream the batch was made on>)) after
    # two warm-up calls there; a call whose descriptors repeat only enqueues kernels and memsets.


def ulsch_decode_device(tbs, llr, harq, payload, ack, iter_max, numMaxIter=8, stream=None, harq_ids=None):
    """llr: torch int16 [>= co[-1]], harq: torch int16 [>= ho[-1]], payload: torch uint8 [>= po[-1]] (out),
    ack: torch uint8 [n], iter_max: torch int32 [n].  Asynchronous; tb dicts get their llrLen updated.
    harq=None with harq_ids: the library keeps the soft buffers (MEM_HARQ_LIBRARY)."""
    import torch
    L = _tb_lib()
    po, co, ho, _ = tb_layout(tbs)
    assert llr.is_cuda and llr.dtype == torch.int16
    mem = MEM_DEVICE
    if harq_ids is not None:
        assert harq is None and len(harq_ids) == len(tbs)
        ho, mem = list(harq_ids), MEM_DEVICE | MEM_HARQ_LIBRARY
    else:
        assert harq.dtype == torch.int16 and harq.numel() >= ho[-1]
    arr = _tb_array(tbs, po, co, ho, numMaxIter)
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    b = nrLDPC_hip_tb_batch_t(n_tb=len(tbs), tb=arr, payload=payload.data_ptr(), coded=llr.data_ptr(), harq=_ptr(harq),
                              harq_stride=HARQ_STRIDE, ack=ack.data_ptr(), iter_max=iter_max.data_ptr(), mem=mem,
                              stream=s)
    _check(L.nrLDPC_hip_ulsch_decode(C.byref(b)), "nrLDPC_hip_ulsch_decode")
    for i, t in enumerate(tbs):
        t["llrLen"] = arr[i].llrLen
