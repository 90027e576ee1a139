#!/usr/bin/env python3
"""Supplementary regression vectors from the SURVEY-STAGE build of the reference (development aid).

PROVENANCE -- NOT A PIN.  The reference LDPC path cannot be compiled in this image without writing
stand-ins for the absent SIMDE headers, which the build rules forbid, so this repository has no
oracle/_ref and declares decoder parity against a reference binary "unpinned" (DESIGN.md "Oracle").
The survey stage that preceded this build left reference shared objects under /tmp/oracle
(libldpc_ref_avx512.so = nrLDPC_decoder.c, libenc_orig.so = ldpc_encoder.c; built there with a
simde->native forwarding shim and the reference's own code generators).  Those objects are not
reproducible from this repository.  This script merely records what they output for seeded inputs so
that the comparison made during development (oracle == those objects on every case tried) stays
checkable as a regression test: tests/golden/survey_ref_*.npz.  Treat them as extra test vectors of
undeclared pedigree, never as the parity pin.

Run (development container only, needs /tmp/oracle/*.so and gcc):  python tools/dev_make_survey_vectors.py
"""
import ctypes as C
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402  (used only to create valid code words / CRC callback)

SURVEY = Path("/tmp/oracle")
HOST_SYMS = r"""
#include <stdio.h>
#include <stdlib.h>
int opp_enabled = 0;
static char glog_buf[1<<20];
void *g_log = glog_buf;
void exit_function(const char *f, const char *fn, const int l, const char *s, const int a){ printf("exit %s:%d\n",f,l); exit(1);}
void logRecord_mt(const char *f, const char *fn, int l, int c, int lv, const char *fmt, ...) {}
"""


class DecP(C.Structure):  # t_nrLDPC_dec_params, nrLDPC_types.h:84-97
    _fields_ = [('BG', C.c_uint8), ('Z', C.c_uint16), ('R', C.c_uint8), ('F', C.c_uint16), ('Qm', C.c_uint8),
                ('rv', C.c_uint8), ('numMaxIter', C.c_uint8), ('E', C.c_int), ('outMode', C.c_int),
                ('crc_type', C.c_int), ('check_crc', C.c_void_p), ('setCombIn', C.c_uint8)]


class EncP(C.Structure):  # encoder_implemparams_t, nrLDPC_defs.h:40-66
    _fields_ = [('n_segments', C.c_uint), ('macro_num', C.c_uint), ('gen_code', C.c_ubyte), ('tinput', C.c_void_p),
                ('tprep', C.c_void_p), ('tparity', C.c_void_p), ('toutput', C.c_void_p), ('Kr', C.c_int),
                ('Kb', C.c_uint32), ('Zc', C.c_uint32), ('harq', C.c_void_p), ('BG', C.c_uint8),
                ('output', C.c_void_p), ('K', C.c_uint32), ('F', C.c_uint32), ('Qm', C.c_uint8), ('E', C.c_uint32),
                ('G', C.c_uint), ('rv', C.c_uint8)]


def main():
    tmp = Path(tempfile.mkdtemp())
    (tmp / "h.c").write_text(HOST_SYMS)
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", str(tmp / "libh.so"), str(tmp / "h.c")], check=True)
    C.CDLL(str(tmp / "libh.so"), mode=C.RTLD_GLOBAL)
    dec = C.CDLL(str(SURVEY / "libldpc_ref_avx512.so"))
    enc = C.CDLL(str(SURVEY / "libenc_orig.so"))
    osor = C.CDLL(str(O.build()))
    osor.oracle_check_crc.argtypes = [C.c_void_p, C.c_uint32, C.c_uint8]
    cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint8)(lambda p, n, t: osor.oracle_check_crc(p, n, t))

    def aligned(n, dtype, fill=0):
        raw = np.full(n + 128, fill, dtype=dtype)
        off = (-raw.ctypes.data) % 64
        return raw, raw[off:off + n]

    def ref_decode(BG, Z, R, llr, it, mode, use_crc, E, crc_type):
        p = DecP(BG=BG, Z=Z, R=R, numMaxIter=it, outMode=mode, E=E, crc_type=crc_type)
        if use_crc:
            p.check_crc = C.cast(cb, C.c_void_p)
        _, a = aligned(68 * 384 + 64, np.int8)
        a[:llr.size] = llr
        _, o = aligned(68 * 384 + 64, np.uint8, 0x55)
        ts = np.zeros(8192, np.uint8)
        ab = np.zeros(64, np.uint8)
        n = dec.LDPCdecoder(C.byref(p), 0, 0, 0, C.c_void_p(a.ctypes.data), C.c_void_p(o.ctypes.data),
                            C.c_void_p(ts.ctypes.data), C.c_void_p(ab.ctypes.data))
        return n, o[:O.out_bytes(BG, Z, R, mode)].copy()

    rng = np.random.default_rng(20260927)
    llrs, runs, outs = [], [], []
    cases = [(1, 384, 13, 3), (1, 384, 23, 2), (1, 384, 89, 2), (1, 176, 13, 3), (2, 64, 15, 4), (2, 64, 13, 3),
             (2, 208, 15, 3), (2, 208, 23, 2), (2, 8, 15, 4), (1, 2, 13, 4), (1, 3, 13, 4), (2, 384, 15, 2),
             (1, 36, 89, 3), (2, 30, 13, 3)]
    for (BG, Z, R, nvec) in cases:
        K = (22 if BG == 1 else 10) * Z
        pad = (-K) % 8
        ntx = (O.NCOLS[(BG, R)] - 2) * Z
        rate = K / ntx
        for v in range(nvec + 2):
            bits = rng.integers(0, 2, K, dtype=np.uint8)
            if K % 8 == 0 and K >= 48:  # put a valid CRC24B at the end so CRC mode can succeed
                ib = np.packbits(bits)
                crc = O.crc("crc24b", ib, K - 24) >> 8
                ib[-3:] = [(crc >> 16) & 255, (crc >> 8) & 255, crc & 255]
                bits = np.unpackbits(ib)
            info = np.packbits(np.concatenate([bits, np.zeros(pad, np.uint8)]))
            cw = O.encode(BG, Z, info)
            if v < nvec:
                snr = [-8.0, 10 * np.log10(rate * 3) - 1.5, 10 * np.log10(rate * 3) + 0.7, 6.0][v % 4]
                llr = O.awgn_llr(rng, cw[:ntx], Z, snr)
            elif v == nvec:
                llr = rng.integers(-128, 128, ntx + 2 * Z).astype(np.int8)
            else:
                llr = rng.choice(np.array([-128, -127, 127, 0, 1, -1], dtype=np.int8), ntx + 2 * Z)
            li = len(llrs)
            llrs.append(llr)
            for it in (0, 1, 2, 8):
                for mode in ((0, 1) if it == 8 else (0,)):
                    n, o = ref_decode(BG, Z, R, llr, it, mode, False, 0, 0)
                    runs.append((li, BG, Z, R, it, mode, 0, 0, 0, n, sum(x.size for x in outs), o.size))
                    outs.append(o)
                if K % 8 == 0 and K >= 48 and it in (2, 8):
                    n, o = ref_decode(BG, Z, R, llr, it, 0, True, K, 1)
                    runs.append((li, BG, Z, R, it, 0, 1, K, 1, n, sum(x.size for x in outs), o.size))
                    outs.append(o)
    gold = ROOT / "tests" / "golden"
    gold.mkdir(exist_ok=True, parents=True)
    np.savez_compressed(gold / "survey_ref_decoder.npz",
                        llr_cat=np.concatenate(llrs), llr_off=np.cumsum([0] + [x.size for x in llrs]),
                        runs=np.array(runs, dtype=np.int64), out_cat=np.concatenate(outs),
                        run_fields=np.array("llr_idx BG Z R numMaxIter outMode use_crc E crc_type n_iter out_off out_len".split()))
    # encoder vectors (ldpc_encoder.c, incl. BG2 Zc=64 where the default optimised encoder is broken)
    evec = []
    for (BG, Z, Kb) in [(1, 384, 22), (1, 176, 22), (1, 208, 22), (1, 8, 22), (2, 64, 10), (2, 208, 10), (2, 384, 10),
                        (2, 16, 6), (2, 40, 8), (2, 64, 9), (1, 352, 22), (2, 120, 10)]:
        K = (22 if BG == 1 else 10) * Z
        bits = rng.integers(0, 2, K, dtype=np.uint8)
        if Kb < 10:
            bits[Kb * Z:] = 0
        info = np.packbits(bits)
        p = EncP(n_segments=1, macro_num=0, gen_code=0, Kr=K, Kb=Kb, Zc=Z, BG=BG, K=K, E=K)
        inb = np.zeros(K // 8 + 64, np.uint8)
        inb[:info.size] = info
        out = np.zeros(68 * 384, np.uint8)
        n = enc.LDPCencoder((C.c_void_p * 1)(inb.ctypes.data), (C.c_void_p * 1)(out.ctypes.data), C.byref(p))
        evec.append((BG, Z, Kb, info, np.packbits(out[:n]), n))
    np.savez_compressed(gold / "survey_ref_encoder.npz",
                        meta=np.array([(a, b, c, n) for (a, b, c, _, _, n) in evec], dtype=np.int64),
                        **{f"info_{i}": e[3] for i, e in enumerate(evec)},
                        **{f"coded_{i}": e[4] for i, e in enumerate(evec)})
    print("decoder runs:", len(runs), "llr vectors:", len(llrs), "encoder vectors:", len(evec))
    for f in ("survey_ref_decoder.npz", "survey_ref_encoder.npz"):
        print(f, (gold / f).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
