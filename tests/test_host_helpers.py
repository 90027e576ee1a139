"""Host-side arithmetic of the transport-block chain (csrc/nr_coding_host.c through the C ABI) vs the oracle
(CPU only: these entry points need no GPU)."""
import numpy as np

import oracle_lib as O


def test_segmentation_E_and_R_match_the_oracle(built):
    import openairinterface5g_amd as pkg
    m = pkg.ldpc
    rng = np.random.default_rng(0)
    n_ok = 0
    for BG in (1, 2):
        for B in list(range(24, 4000, 8)) + [int(x) * 8 for x in rng.integers(500, 160000, 400)]:
            ref = O.segmentation(None, B, BG)
            got = m.nr_segmentation(B, BG)
            L = 24 if ref["C"] > 1 else 0
            aligned = (ref["K"] - ref["F"] - L) % 8 == 0 and ref["F"] % 8 == 0
            if got is None:
                assert ref["Kb"] < 0 or not aligned, (BG, B)      # rejected only outside the reference's byte contract
                continue
            n_ok += 1
            assert got == {k: ref[k] for k in ("C", "K", "Z", "F", "Kb")}, (BG, B)
    assert n_ok > 500
    for G, C_, Qm, Nl in [(60000, 4, 6, 2), (14400, 2, 2, 1), (288000, 12, 8, 4), (26400, 3, 2, 1), (1200, 1, 2, 1)]:
        es = [m.nr_get_E(G, C_, Qm, Nl, r) for r in range(C_)]
        assert es == [O.get_E(G, C_, Qm, Nl, r) for r in range(C_)] and sum(es) == G
    for BG, Z in ((1, 384), (1, 96), (2, 64), (2, 208)):
        for rv in range(4):
            for E in (Z * 11, Z * 24, Z * 30, Z * 50, Z * 70, Z * 140):
                for rnd, ll in ((0, 0), (1, Z * 20), (2, Z * 66)):
                    assert m.nr_get_R_ldpc_decoder(rv, E, BG, Z, ll, rnd) == O.get_R(rv, E, BG, Z, ll, rnd)


# ---- properties that need neither the oracle nor the product's banding code: 38.212 5.2.2 / 5.4.2.1 stated directly -------
TBS_TABLE = [24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 208, 224, 240,
             256, 272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576, 608, 640, 672, 704, 736, 768,
             808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256, 1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672,
             1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216, 2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104,
             3240, 3368, 3496, 3624, 3752, 3824]                      # 38.214 Table 5.1.3.2-1
LIFTING_SET = sorted(a * 2 ** j for a in (2, 3, 5, 7, 9, 11, 13, 15) for j in range(8) if a * 2 ** j <= 384)   # 38.212 Table 5.3.2-1


def large_tbs():
    """38.214 5.1.3.2 step 4 (N_info > 3824) over a sweep of N_info, both code-rate branches."""
    import math
    out = set()
    ninfo = 3825.0
    while ninfo < 1.3e6:
        n = int(math.floor(math.log2(ninfo - 24))) - 5
        npr = max(3840, 2 ** n * int(round((ninfo - 24) / 2 ** n)))
        for low_rate in (True, False):
            if low_rate:
                Cn = -(-(npr + 24) // 3816)
            elif npr > 8424:
                Cn = -(-(npr + 24) // 8424)
            else:
                Cn = 1
            out.add(8 * Cn * (-(-(npr + 24) // (8 * Cn))) - 24)
        ninfo *= 1.013
    return sorted(out)


def _both(built):
    import openairinterface5g_amd as pkg
    m = pkg.ldpc
    seg_o = lambda B, BG: (lambda r: None if r["Kb"] < 0 else {k: r[k] for k in ("C", "K", "Z", "F", "Kb")})(O.segmentation(None, B, BG))
    return (("nr_coding_host.c", m.nr_segmentation, m.nr_get_E, lambda *a: m.nr_get_R_ldpc_decoder(*a)),
            ("oracle", seg_o, O.get_E, lambda rv, E, BG, Z, ll, rnd: O.get_R(rv, E, BG, Z, ll, rnd)))


def test_segmentation_properties_of_38_212_on_every_tbs(built):
    """C (K' - L) = B; Kb from the B thresholds; Zc = the SMALLEST lifting size with Kb Zc >= K' -- looked up in the 51-entry
    set of Table 5.3.2-1, not by the reference's banding arithmetic; K = 22 / 10 Zc; F = K - K'.  Every TBS of 38.214
    (table + formula sweep), BG1 and BG2 where the TBS may use it, on the product's helper AND on the oracle's."""
    sizes = TBS_TABLE + large_tbs()
    assert len(sizes) > 400 and max(sizes) > 1.2e6
    for name, seg, _, _ in _both(built):
        n = 0
        for A in sizes:
            B = A + (24 if A > 3824 else 16)
            for BG in (1, 2):
                if BG == 2 and A > 3824 * 40:
                    continue                       # (BG2 is only selected for small blocks / low rates; keep the sweep bounded)
                Kcb = 8448 if BG == 1 else 3840
                Cn = 1 if B <= Kcb else -(-B // (Kcb - 24))
                L = 24 if Cn > 1 else 0
                Bp = B + Cn * L
                got = seg(B, BG)
                if Bp % Cn or (Bp // Cn - L) % 8:
                    continue                       # not a size the standard's TBS determination produces for this base graph
                Kp = Bp // Cn
                Kb = 22 if BG == 1 else (10 if B > 640 else 9 if B > 560 else 8 if B > 192 else 6)
                fits = [z for z in LIFTING_SET if Kb * z >= Kp]
                if not fits:
                    assert got is None, (name, A, BG)
                    continue
                Zc = fits[0]
                K = (22 if BG == 1 else 10) * Zc
                assert got == dict(C=Cn, K=K, Z=Zc, F=K - Kp, Kb=Kb), (name, A, BG, got)
                assert got["C"] * (Kp - L) == B and got["F"] >= 0
                n += 1
        assert n > 600, (name, n)


def test_rate_matching_lengths_properties(built):
    """38.212 5.4.2.1: the E_r are multiples of Nl Qm, sum to G, take at most two neighbouring values, the smaller ones first:
    the first C - (G / (Nl Qm)) mod C segments get the floor."""
    rng = np.random.default_rng(12)
    for name, _, get_E, _ in _both(built):
        for _ in range(400):
            Qm, Nl, Cn = int(rng.choice([2, 4, 6, 8])), int(rng.integers(1, 5)), int(rng.integers(1, 40))
            G = int(rng.integers(Cn, 60000)) * Qm * Nl
            es = [get_E(G, Cn, Qm, Nl, r) for r in range(Cn)]
            assert sum(es) == G and all(e % (Nl * Qm) == 0 for e in es), (name, G, Cn, Qm, Nl)
            assert es == sorted(es) and es[-1] - es[0] in (0, Nl * Qm)
            n_floor = Cn - (G // (Nl * Qm)) % Cn
            assert es[:n_floor] == [Nl * Qm * (G // (Nl * Qm * Cn))] * n_floor


def test_decoder_rate_mode_at_the_exact_boundaries(built):
    """nr_get_R_ldpc_decoder: mode from sysBits / (min(k0 + E, Ncb) + 2 Z) against 1/3, 2/3, 8/9 written as 0.3333, 0.6667,
    0.8889 (nr_rate_matching.c:405-421).  For every lifting size and rv, the two E on either side of each threshold, decided
    here with exact rationals (the ratio never equals a threshold and its steps are 1000 x a float's resolution); plus the
    llrLen state: round 0 stores k0 + E uncapped, later rounds keep the maximum of it and the capped value."""
    from fractions import Fraction
    K0 = {1: (0, 17, 33, 56), 2: (0, 13, 25, 43)}
    THR = {1: ((Fraction(6667, 10000), 13), (Fraction(8889, 10000), 23)), 2: ((Fraction(3333, 10000), 15), (Fraction(6667, 10000), 13))}
    TOP = {1: 89, 2: 23}
    for name, _, _, get_R in _both(built):
        n = 0
        for BG in (1, 2):
            sys_cols, ncb_cols = (22, 66) if BG == 1 else (10, 50)
            for Z in LIFTING_SET:
                for rv in range(4):
                    k0 = K0[BG][rv] * Z
                    cands = {1, Z, 3 * Z, ncb_cols * Z, 2 * ncb_cols * Z}
                    for thr, _ in THR[BG]:
                        x = Fraction(sys_cols * Z) / thr - 2 * Z - k0          # info bits at which the ratio equals thr
                        cands |= {int(x) - 1, int(x), int(x) + 1, int(x) + 2}
                    for E in sorted(c for c in cands if c > 0):
                        info = min(k0 + E, ncb_cols * Z)
                        ratio = Fraction(sys_cols * Z, info + 2 * Z)
                        want = TOP[BG]
                        for thr, mode in reversed(THR[BG]):
                            assert ratio != thr
                            if ratio < thr:
                                want = mode
                        R, ll = get_R(rv, E, BG, Z, 12345, 0)
                        assert R == want and ll == max(k0 + E, info), (name, BG, Z, rv, E, R, want)
                        R1, ll1 = get_R(rv, E, BG, Z, 7 * Z, 1)
                        assert R1 == want and ll1 == max(7 * Z, info), (name, BG, Z, rv, E)
                        n += 1
        assert n > 3000


def test_columns_a_first_transmission_can_reach_against_the_literal_rate_matching_loop(built):
    """The chain decodes a first transmission on the rate mode's graph cut behind the last column that received anything
    (nrLDPC_hip_ulsch_decoder_columns).  The claim that everything behind is zero is checked against the oracle's restatement
    of nr_rate_matching_ldpc_rx (nr_rate_matching.c:507-603: the literal walk of the circular buffer): E ones go in, the
    cleared soft buffer comes back, and NO position at or behind the reported column may be non-zero -- while the column in
    front of it must hold something (the cut is tight), for every rv, with and without LBRM, with fillers, with wrap-around."""
    import openairinterface5g_amd as pkg
    m = pkg.ldpc
    rng = np.random.default_rng(5)
    n_cut = n_all = n_bad = 0
    while n_all < 300:
        BG = int(rng.integers(1, 3))
        B = int(rng.integers(40, 60000)) * 8
        sg = m.nr_segmentation(B, BG)
        if sg is None:
            continue
        Z, K, F, C_ = sg["Z"], sg["K"], sg["F"], sg["C"]
        N = (66 if BG == 1 else 50) * Z
        rv = int(rng.integers(0, 4))
        tbslbrm = 0 if rng.random() < 0.6 else int(rng.integers(C_ * N // 3, C_ * N))
        E = int(rng.integers(K - F, 3 * N)) if rng.random() < 0.3 else int(rng.integers((K - F) // 1, max(K - F + 1, N)))
        E -= E % 2
        Foffset = K - F - 2 * Z
        w = np.zeros(N + 8 * Z, dtype=np.int16)
        rc, w = O.rate_match_rx(tbslbrm, BG, Z, w, np.ones(E, dtype=np.int16), C_, rv, 1, E, F, Foffset)
        R, _ = m.nr_get_R_ldpc_decoder(rv, E, BG, Z, 0, 0)
        cols = m.ulsch_decoder_columns(BG, Z, C_, F, K, tbslbrm, rv, E, 0, R)
        if rc != 0:
            assert cols == -1, (BG, B, rv, tbslbrm, E)
            n_bad += 1
            continue
        ncols_mode = m.NCOLS[(BG, R)]
        ncore = 26 if BG == 1 else 14
        assert ncore < cols <= ncols_mode
        nz = np.flatnonzero(w)
        reach_col = (2 * Z + int(nz[-1])) // Z + 1            # columns 0 .. reach_col - 1 hold something
        assert cols >= min(reach_col, ncols_mode), (BG, Z, rv, tbslbrm, E, cols, reach_col)      # nothing non-zero is cut off
        assert cols == min(max(reach_col, ncore + 1), ncols_mode), (BG, Z, rv, tbslbrm, E, cols, reach_col)   # and the cut is tight
        n_all += 1
        n_cut += cols < ncols_mode
        # a retransmission is never cut: the soft buffer's history is the caller's
        assert m.ulsch_decoder_columns(BG, Z, C_, F, K, tbslbrm, rv, E, 1, R) == ncols_mode
    assert n_cut > 80 and n_all - n_cut > 80, (n_all, n_cut, n_bad)
