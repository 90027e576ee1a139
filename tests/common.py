"""Shared helpers for the parity tests (seeded inputs, fixtures)."""
from pathlib import Path

import numpy as np

import oracle_lib as O

GOLDEN = Path(__file__).resolve().parent / "golden"
ALL_RATES = {1: (13, 23, 89), 2: (15, 13, 23)}


def kbits(BG, Z):
    return (22 if BG == 1 else 10) * Z


def random_info(rng, BG, Z, with_crc24b=False):
    """Random information block as packed bytes (K bits, zero padded to a byte)."""
    K = kbits(BG, Z)
    bits = rng.integers(0, 2, K, dtype=np.uint8)
    info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
    if with_crc24b and K % 8 == 0 and K >= 48:
        crc = O.crc("crc24b", info, K - 24) >> 8
        info[-3:] = [(crc >> 16) & 255, (crc >> 8) & 255, crc & 255]
    return info


def make_llr(rng, BG, Z, R, kind, info=None):
    """kind: float = Es/N0-ish SNR in dB relative to the rate-1/3 operating point; 'rand' = uniform int8;
    'sat' = saturation stress alphabet."""
    ntx = (O.NCOLS[(BG, R)] - 2) * Z
    if kind == "rand":
        return rng.integers(-128, 128, ntx + 2 * Z).astype(np.int8)
    if kind == "sat":
        return rng.choice(np.array([-128, -127, 127, 0, 1, -1], dtype=np.int8), ntx + 2 * Z)
    if info is None:
        info = random_info(rng, BG, Z)
    cw = O.encode(BG, Z, info)
    rate = kbits(BG, Z) / ntx
    return O.awgn_llr(rng, cw[:ntx], Z, 10 * np.log10(rate * 3) + float(kind))


def load_survey_decoder_vectors():
    z = np.load(GOLDEN / "survey_ref_decoder.npz")
    runs = z["runs"]
    off = z["llr_off"]
    for row in runs:
        li, BG, Z, R, it, mode, use_crc, E, ct, n_iter, o0, ol = [int(v) for v in row]
        yield dict(BG=BG, Z=Z, R=R, numMaxIter=it, outMode=mode, use_crc=bool(use_crc), E=E, crc_type=ct,
                   n_iter=n_iter, llr=z["llr_cat"][off[li]:off[li + 1]], out=z["out_cat"][o0:o0 + ol], llr_idx=li)
