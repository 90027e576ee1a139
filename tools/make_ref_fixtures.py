#!/usr/bin/env python3
"""Writes tests/golden/ref_encoder.npz and tests/golden/ref_decoder.npz from oracle/_ref -- the library compiled from the
reference's own sources (oracle/ref_pin/: encoder parity part + generator tables, nrLDPC_init, nrLDPC_mPass.h; see
oracle/ref_pin/ref_pin.h for what is reference-compiled and what is restated).  Development container only (needs
/root/reference to build oracle/_ref).  The fixtures are DATA: seeded inputs and the outputs oracle/_ref gave for them.

  python tools/make_ref_fixtures.py            # regenerate both files (deterministic: same bytes every time)

ref_encoder.npz   meta[i] = (BG, Zc, Kb, K, n_out), info_cat / coded_cat = packed bits (MSB first) at info_off / coded_off:
                  every (BG, Zc) twice + BG2 with Kb = 6, 8, 9.  Source: ref_ldpc_encoder_orig =
                  ldpc_encoder.c:44-252 around the reference-compiled encode_parity_check_part_orig.
ref_decoder.npz   runs[j] = (llr_idx, BG, Z, R, numMaxIter, outMode, use_crc, E, crc_type, n_iter, out_off, out_len),
                  llr_cat at llr_off, out_cat.  Source: ref_hybrid_decode (reference-compiled set-up and data
                  movement, restated node arithmetic), CRC predicate = the oracle's check_crc (CRC catalogue values).
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402  (only for the channel model, the CRC attach and the CRC predicate)
import ref_lib as RL    # noqa: E402

GOLDEN = ROOT / "tests" / "golden"

DEC_CODES = [(1, 384, 13), (1, 384, 23), (1, 384, 89), (1, 176, 13), (1, 352, 23), (1, 44, 13), (1, 15, 89), (1, 2, 13),
             (1, 3, 13), (2, 64, 15), (2, 64, 13), (2, 208, 15), (2, 208, 23), (2, 384, 13), (2, 120, 15), (2, 30, 23),
             (2, 8, 15)]


def info_with_crc(rng, BG, Z):
    kb = 22 if BG == 1 else 10
    K = kb * Z
    bits = rng.integers(0, 2, K, dtype=np.uint8)
    info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
    if K % 8 == 0 and K >= 48:
        crc = O.crc("crc24b", info, K - 24) >> 8
        info[-3:] = [(crc >> 16) & 255, (crc >> 8) & 255, crc & 255]
    return info


def valid_tbs(target_bits, BG):
    """Smallest A >= target whose segmentation is byte aligned, as every 38.214 TBS is."""
    A = (target_bits + 7) // 8 * 8
    while True:
        B = O.len_with_crc(1, A)
        s = O.segmentation(None, B, BG)
        L = 24 if s["C"] > 1 else 0
        if (s["K"] - s["F"] - L) % 8 == 0 and s["F"] % 8 == 0 and (B + L * s["C"]) % s["C"] == 0:
            return A
        A += 8


def make_encoder():
    rng = np.random.default_rng(20260927)
    meta, infos, codeds = [], [], []
    cases = [(BG, Z, 22 if BG == 1 else 10) for BG in (1, 2) for Z in O.LIFT_SIZES for _ in range(2)]
    cases += [(2, Z, Kb) for Z, Kb in ((2, 6), (8, 6), (20, 6), (36, 8), (60, 8), (64, 9), (112, 9), (384, 9), (208, 8))]
    for BG, Z, Kb in cases:
        K = (22 if BG == 1 else 10) * Z
        bits = rng.integers(0, 2, K, dtype=np.uint8)
        bits[Kb * Z:] = 0
        info = np.packbits(bits)
        coded = RL.encode(BG, Z, info, Kb=Kb)
        meta.append((BG, Z, Kb, K, coded.size))
        infos.append(info)
        codeds.append(np.packbits(coded))
    off = lambda xs: np.concatenate([[0], np.cumsum([x.size for x in xs])]).astype(np.int64)
    # transport blocks: payload -> TB CRC + nr_segmentation (oracle: byte copies + CRCs, pinned by the CRC catalogue values)
    # -> every segment through the reference-compiled encoder.  tb_meta[i] = (A, BG, C, Zc, K, F, Kb); the test derives
    # the expected DL-SCH output (rv 0, Qm 2, one layer, E = all transmittable bits) from the stored code words.
    tb_meta, tb_pay, tb_cw = [], [], []
    for bits, BG in ((100000, 1), (213176, 1), (9600, 1), (3840, 1), (5000, 2), (800, 2), (264, 2), (24, 2)):
        A = valid_tbs(bits, BG)
        payload = rng.integers(0, 256, A // 8, dtype=np.uint8)
        a = np.concatenate([payload, np.zeros(4, np.uint8)])
        if A > O.NR_MAX_PDSCH_TBS:
            c = O.crc("crc24a", a, A) >> 8
            a[A // 8:A // 8 + 3] = [(c >> 16) & 255, (c >> 8) & 255, c & 255]
        else:
            c = O.crc("crc16", a, A) >> 16
            a[A // 8:A // 8 + 2] = [(c >> 8) & 255, c & 255]
        s = O.segmentation(a, O.len_with_crc(1, A), BG)
        cws = [np.packbits(RL.encode(BG, s["Z"], seg, Kb=s["Kb"])) for seg in s["segs"]]
        tb_meta.append((A, BG, s["C"], s["Z"], s["K"], s["F"], s["Kb"]))
        tb_pay.append(payload)
        tb_cw.append(np.concatenate(cws))
    np.savez_compressed(GOLDEN / "ref_encoder.npz", meta=np.array(meta, np.int32), info_cat=np.concatenate(infos),
                        info_off=off(infos), coded_cat=np.concatenate(codeds), coded_off=off(codeds),
                        tb_meta=np.array(tb_meta, np.int64), tb_payload_cat=np.concatenate(tb_pay), tb_payload_off=off(tb_pay),
                        tb_cw_cat=np.concatenate(tb_cw), tb_cw_off=off(tb_cw))
    print("ref_encoder.npz:", len(meta), "code words,", len(tb_meta), "transport blocks,", sum(m[2] for m in tb_meta), "segments")


def make_decoder():
    rng = np.random.default_rng(20260928)
    crc_fn = C.cast(O.lib().oracle_check_crc, C.c_void_p)
    L = RL.lib()
    llrs, runs, outs = [], [], []
    out_off = 0
    for BG, Z, R in DEC_CODES:
        kb = 22 if BG == 1 else 10
        K, ncols = kb * Z, O.NCOLS[(BG, R)]
        rate3 = 10 * np.log10(kb / (ncols - 2) * 3)
        cases = []
        for snr in (3.0, 0.5, -4.0):
            cw = O.encode(BG, Z, info_with_crc(rng, BG, Z))
            cases.append(O.awgn_llr(rng, cw, Z, snr + rate3)[:ncols * Z].copy())
        cases.append(rng.choice(np.array([-128, -127, 127, 0, 1, -1], np.int8), ncols * Z))
        for llr in cases:
            li = len(llrs)
            llrs.append(llr)
            combos = [(it, 0, 0) for it in (1, 2, 8)] + [(8, 1, 0)]
            if K % 8 == 0 and K >= 48:
                combos += [(8, 0, 1), (3, 1, 1), (8, 2, 1)]
            for it, mode, use_crc in combos:
                nb = RL.out_bytes(BG, Z, R, mode)
                out = np.full(nb + 64, 0x5A, np.uint8)
                n = L.ref_hybrid_decode(BG, Z, R, it, mode, crc_fn if use_crc else None, K if use_crc else 0, O.CRC24_B, 0,
                                        llr.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
                assert n > 0
                runs.append((li, BG, Z, R, it, mode, use_crc, K if use_crc else 0, O.CRC24_B, n, out_off, nb))
                outs.append(out[:nb].copy())
                out_off += nb
    llr_off = np.concatenate([[0], np.cumsum([x.size for x in llrs])]).astype(np.int64)
    np.savez_compressed(GOLDEN / "ref_decoder.npz", runs=np.array(runs, np.int64), llr_cat=np.concatenate(llrs),
                        llr_off=llr_off, out_cat=np.concatenate(outs), out_init=np.uint8(0x5A))
    conv = sum(1 for r in runs if r[9] <= r[4])
    print("ref_decoder.npz:", len(runs), "runs on", len(llrs), "inputs;", conv, "converged")


if __name__ == "__main__":
    assert RL.available(), "oracle/_ref cannot be built here (no /root/reference)"
    make_encoder()
    make_decoder()
