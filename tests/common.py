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


def load_ref_decoder_vectors():
    """tests/golden/ref_decoder.npz: outputs of oracle/_ref's hybrid decoder (reference-compiled set-up and data movement
    + restated node arithmetic; tools/make_ref_fixtures.py)."""
    z = np.load(GOLDEN / "ref_decoder.npz")
    off, llr_cat, out_cat = z["llr_off"], z["llr_cat"], z["out_cat"]
    for row in z["runs"]:
        li, BG, Z, R, it, mode, use_crc, E, ct, n_iter, o0, ol = [int(v) for v in row]
        yield dict(BG=BG, Z=Z, R=R, numMaxIter=it, outMode=mode, use_crc=bool(use_crc), E=E, crc_type=ct, n_iter=n_iter,
                   llr=llr_cat[off[li]:off[li + 1]], out=out_cat[o0:o0 + ol], llr_idx=li, out_init=int(z["out_init"]))


def load_ref_code_words():
    """tests/golden/ref_encoder.npz: code words of the reference-compiled encoder (parity part + generator tables)."""
    z = np.load(GOLDEN / "ref_encoder.npz")
    io, co = z["info_off"], z["coded_off"]
    for i, (BG, Z, Kb, K, n) in enumerate(z["meta"]):
        yield dict(BG=int(BG), Z=int(Z), Kb=int(Kb), K=int(K), info=z["info_cat"][io[i]:io[i + 1]],
                   coded=np.unpackbits(z["coded_cat"][co[i]:co[i + 1]])[:n])


def load_ref_transport_blocks():
    """Transport blocks of ref_encoder.npz: payload + per-segment reference code words -> the DL-SCH output expected for
    rv 0, QPSK, one layer with every transmittable bit sent once (E = N - F, rounded down to a symbol):
    nr_rate_matching_ldpc (nr_rate_matching.c:424-501) then reads d from 0 and skips the fillers, and
    nr_interleaving_ldpc (:240-303) maps f[i + j Qm] = e[i E/Qm + j]."""
    z = np.load(GOLDEN / "ref_encoder.npz")
    po, wo = z["tb_payload_off"], z["tb_cw_off"]
    for i, (A, BG, C_, Z, K, F, Kb) in enumerate(z["tb_meta"]):
        A, BG, C_, Z, K, F, Kb = (int(v) for v in (A, BG, C_, Z, K, F, Kb))
        N = (66 if BG == 1 else 50) * Z
        nbytes = (N + 7) // 8
        cws = z["tb_cw_cat"][wo[i]:wo[i + 1]].reshape(C_, nbytes)
        Qm = 2
        E = (N - F) // Qm * Qm
        f = []
        for r in range(C_):
            d = np.unpackbits(cws[r])[:N]
            e = np.concatenate([d[:K - F - 2 * Z], d[K - 2 * Z:]])[:E]
            f.append(e.reshape(Qm, E // Qm).T.reshape(-1))
        yield dict(tb=dict(A=A, G=C_ * E, BG=BG, Qm=Qm, Nl=1, rv=0, tbslbrm=0), payload=z["tb_payload_cat"][po[i]:po[i + 1]],
                   coded=np.concatenate(f), C=C_, Z=Z, K=K, F=F, Kb=Kb, code_words=[np.unpackbits(c)[:N] for c in cws])
