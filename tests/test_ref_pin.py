"""oracle/_ref (reference-COMPILED: encoder parity part + generator tables, nrLDPC_init LUT selection, all decoder data
movement of nrLDPC_mPass.h) against the oracle's restatement.  Runs wherever oracle/_ref can be built or was shipped
(the development container; the GPU box when the snapshot carried the .so); the committed fixtures
tests/golden/ref_*.npz carry the same outputs everywhere else (test_ref_fixtures.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ref_lib as RL

pytestmark = pytest.mark.skipif(not RL.available(), reason="oracle/_ref not built (needs /root/reference)")

MODES = [(1, 13), (1, 23), (1, 89), (2, 15), (2, 13), (2, 23)]


def test_ref_library_needs_nothing_but_libc():
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", str(RL.LIB_PATH)], capture_output=True, text=True).stdout
    names = {l.split()[-1].split("@")[0] for l in und.splitlines() if l.strip()}
    names -= {n for n in names if n.startswith("_ITM") or n.startswith("__gmon") or n.startswith("__cxa")}
    assert names <= {"memcpy", "memmove", "memset", "calloc", "free", "puts", "printf", "__memset_chk", "__memcpy_chk",
                     "__memmove_chk", "__stack_chk_fail", "__tls_get_addr", "__printf_chk"}, names


@pytest.mark.parametrize("BG", [1, 2])
def test_reference_encoder_equals_oracle_every_lifting_size(BG):
    """a14/a21: ldpc_encoder.c's code word, parity part by reference-compiled encode_parity_check_part_orig with the
    reference's generator tables, = the oracle's (which solves H x = 0 from the BG*_I* shift tables)."""
    rng = np.random.default_rng(100 + BG)
    kb = 22 if BG == 1 else 10
    for Z in O.LIFT_SIZES:
        assert RL.lib().ref_has_generator_matrix(BG, Z)
        for trial in range(2):
            K = kb * Z
            bits = rng.integers(0, 2, K, dtype=np.uint8)
            info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
            ref = RL.encode(BG, Z, info)
            assert np.array_equal(ref, O.encode(BG, Z, info)), (BG, Z)
            full = np.concatenate([bits[:2 * Z], ref])
            assert O.syndrome_weight(BG, Z, full) == 0


def test_reference_encoder_kb_below_ten_and_shortened_lengths():
    """BG2 with Kb = 6, 8, 9 (nr_segmentation.c:69-82) and ldpctest's shortened block lengths."""
    rng = np.random.default_rng(7)
    for Z, Kb in ((8, 6), (36, 8), (64, 9), (208, 10), (384, 9)):
        K = 10 * Z
        bits = rng.integers(0, 2, K, dtype=np.uint8)
        bits[Kb * Z:] = 0                      # columns Kb..9 are fillers = 0 when Kb < 10
        info = np.packbits(np.concatenate([bits, np.zeros((-K) % 8, np.uint8)]))
        assert np.array_equal(RL.encode(2, Z, info, Kb=Kb), O.encode(2, Z, info, Kb=Kb)), (Z, Kb)


def test_reference_parity_part_bit_sliced_eight_segments():
    """a15: ldpc_encoder_optim8segmulti.c:132-208 feeds the SAME function with 8 segments sliced into the bits of a
    byte; the result de-sliced = 8 independent code words."""
    rng = np.random.default_rng(8)
    for BG, Z in ((1, 384), (1, 22), (2, 64), (2, 15)):
        kb, nrows = (22, 46) if BG == 1 else (10, 42)
        bits = rng.integers(0, 2, (8, kb * Z), dtype=np.uint8)
        c = np.zeros(kb * Z, np.uint8)
        for s in range(8):
            c |= bits[s] << s
        d = RL.parity_part(BG, Z, c)
        for s in range(8):
            info = np.packbits(np.concatenate([bits[s], np.zeros((-kb * Z) % 8, np.uint8)]))
            cw = O.encode(BG, Z, info)
            assert np.array_equal((d >> s) & 1, cw[(kb - 2) * Z:]), (BG, Z, s)


@pytest.mark.parametrize("BG,R", MODES)
def test_reference_init_numllr_and_group_luts(BG, R):
    """a3: nrLDPC_init selects a LUT set for every lifting size; numLLR, CN-group populations and BN-group
    populations agree with the oracle's graph."""
    for Z in O.LIFT_SIZES:
        h = RL.lib().ref_dec_new(BG, Z, R)
        assert h, (BG, Z, R)
        g = O.graph(BG, Z, R)
        assert RL.lib().ref_dec_numLLR(h) == g.ncols * Z == O.NCOLS[(BG, R)] * Z
        G = RL.lib().ref_dec_numCnGroups(h)
        ncn = RL.lib().ref_dec_numCnInCnGroups(h)
        degs = [g.row_ptr[r + 1] - g.row_ptr[r] for r in range(g.nrows)]
        for k in range(G):
            assert ncn[k] == degs.count(RL.lib().ref_dec_bnInCnGroup(h, k)), (BG, Z, R, k)
        assert sum(ncn[k] for k in range(G)) == g.nrows
        nbn = RL.lib().ref_dec_numBnInBnGroups(h)
        cols = [g.col[e] for e in range(g.nedges)]
        hist = [0] * 31
        for c in range(g.ncols):
            hist[cols.count(c)] += 1
        assert [nbn[d - 1] for d in range(1, 31)] == hist[1:31], (BG, Z, R)
        RL.lib().ref_dec_free(h)
    assert not RL.lib().ref_dec_new(BG, 7 * 64 + 1, R)   # not a lifting size


def _llr_cases(rng, BG, Z, R, n):
    """Inputs that exercise both outcomes: noisy code words over a range of SNRs (converging and not), uniform
    garbage, and the saturation corners."""
    kb = 22 if BG == 1 else 10
    ncols = O.NCOLS[(BG, R)]
    for i in range(n):
        info = rng.integers(0, 256, (kb * Z + 7) // 8, dtype=np.uint8)
        if (kb * Z) % 8:
            info[-1] &= 0xFF << (8 - (kb * Z) % 8) & 0xFF
        cw = O.encode(BG, Z, info)
        rate = kb / (ncols - 2)
        snr = rng.choice([-3.0, 0.0, 1.0, 2.0, 4.0, 8.0]) + 10 * np.log10(rate * 3)
        yield O.awgn_llr(rng, cw, Z, snr)[:ncols * Z].copy()
    yield rng.integers(-128, 128, ncols * Z, dtype=np.int8)
    yield rng.choice(np.array([-128, -127, 127, 0], np.int8), ncols * Z)
    yield np.full(ncols * Z, -128, np.int8)
    yield np.zeros(ncols * Z, np.int8)


@pytest.mark.parametrize("BG,R", MODES)
def test_hybrid_reference_decoder_equals_oracle_every_lifting_size(BG, R):
    """a2-a12: the decoder whose set-up and data movement are reference-compiled (nrLDPC_init, nrLDPC_mPass.h) and
    whose node arithmetic is restated on the reference's buffer layouts gives the oracle's pass counts and bits for
    every lifting size -- i.e. the oracle's (edge, lane) formulation, its graph and its parity-check group order
    [F6] are the reference's."""
    rng = np.random.default_rng(BG * 100 + R)
    for Z in O.LIFT_SIZES:
        n = 1 if Z > 100 else 2
        for llr in _llr_cases(rng, BG, Z, R, n):
            for max_iter in (8,) if Z > 100 else (1, 8):
                n_ref, out_ref = RL.decode(BG, Z, R, llr, max_iter, 0, out_init=0xA5)
                n_or, out_or = O.decode(BG, Z, R, llr, max_iter, O.OUT_BIT, out_init=0xA5)
                assert n_ref == n_or, (BG, Z, R, max_iter, n_ref, n_or)
                assert np.array_equal(out_ref, out_or), (BG, Z, R, max_iter)


@pytest.mark.parametrize("BG,Z,R", [(1, 384, 13), (1, 176, 23), (1, 36, 89), (2, 64, 15), (2, 208, 13), (2, 15, 23)])
def test_hybrid_reference_decoder_output_modes_and_crc_stop(BG, Z, R):
    """a10/a11: BITINT8 / LLRINT8 outputs and the CRC stop, the predicate called as decoder.c:857 calls it."""
    rng = np.random.default_rng(Z)
    kb = 22 if BG == 1 else 10
    K = kb * Z
    for snr in (6.0, 1.0, -6.0):
        nbytes = K // 8
        payload = rng.integers(0, 256, nbytes - 3, dtype=np.uint8)
        crc = O.crc("crc24b", np.concatenate([payload, np.zeros(4, np.uint8)]), (nbytes - 3) * 8) >> 8
        info = np.concatenate([payload, np.array([(crc >> 16) & 255, (crc >> 8) & 255, crc & 255], np.uint8)])
        if K % 8:
            info = np.concatenate([info, np.zeros(1, np.uint8)])
        cw = O.encode(BG, Z, info)
        llr = O.awgn_llr(rng, cw, Z, snr + 10 * np.log10(kb / (O.NCOLS[(BG, R)] - 2) * 3))[:O.NCOLS[(BG, R)] * Z]
        E = nbytes * 8
        calls = []

        def pred(p, n, t):
            calls.append((n, t))
            buf = np.ctypeslib.as_array(p, shape=(n // 8 + 4,))
            return O.check_crc(buf, n, t)
        for mode in (0, 1, 2):
            calls.clear()
            n_ref, out_ref = RL.decode(BG, Z, R, llr, 8, mode, check_crc=pred, E=E, crc_type=O.CRC24_B, out_init=0x5A)
            n_or, out_or = O.decode(BG, Z, R, llr, 8, mode, use_crc=True, E=E, crc_type=O.CRC24_B, out_init=0x5A)
            assert n_ref == n_or and np.array_equal(out_ref, out_or), (BG, Z, R, snr, mode)
            assert calls and all(c == (E, O.CRC24_B) for c in calls) and len(calls) == max(0, n_ref - 2)
            n_ref, out_ref = RL.decode(BG, Z, R, llr, 8, mode, out_init=0x5A)
            n_or, out_or = O.decode(BG, Z, R, llr, 8, mode, out_init=0x5A)
            assert n_ref == n_or and np.array_equal(out_ref, out_or)


def test_generic_bnprocpc_differs_only_in_parity_columns():
    """[F5]: following the generic bnProcPc (degree-1 group summed) instead of the generated one changes the output
    behind the core columns only -- and never the pass count."""
    rng = np.random.default_rng(55)
    for BG, Z, R in ((1, 96, 13), (2, 52, 15), (1, 384, 23)):
        ncore = 26 if BG == 1 else 14
        for llr in _llr_cases(rng, BG, Z, R, 2):
            n0, o0 = RL.decode(BG, Z, R, llr, 8, 1)
            n1, o1 = RL.decode(BG, Z, R, llr, 8, 1, deg1_generic=True)
            assert n0 == n1 and np.array_equal(o0[:ncore * Z], o1[:ncore * Z])
            assert not o0[ncore * Z:].any()


# ---- the SHIPPED node functions' text, written by the reference's own generators (oracle/ref_pin/ref_gen_main.c) --------------
GEN = RL.LIB_PATH.parent / "gen"
RATES = {1: (13, 23, 89), 2: (15, 13, 23)}


def _lut(BG, R):
    """group LUTs of (BG, R) as nrLDPC_init hands them out (reference-compiled), plain Python lists"""
    L = RL.lib()
    h = L.ref_dec_new(BG, 384, R)
    G = L.ref_dec_numCnGroups(h)
    d = dict(G=G, numCn=[L.ref_dec_numCnInCnGroups(h)[g] for g in range(G)], startCn=[L.ref_dec_startAddrCnGroups(h)[g] for g in range(G)],
             bnInCn=[L.ref_dec_bnInCnGroup(h, g) for g in range(G)], cnFull=[L.ref_dec_cnInCnGroupFull(h, g) for g in range(G)],
             numBn=[L.ref_dec_numBnInBnGroups(h)[k] for k in range(30)])
    nz = sum(1 for x in d["numBn"] if x)
    d["startBn"] = [L.ref_dec_startAddrBnGroups(h)[k] for k in range(nz)]
    d["startLlr"] = [L.ref_dec_startAddrBnGroupsLlr(h)[k] for k in range(nz)]
    L.ref_dec_free(h)
    return d


CN_VARIANTS = {
    # AVX2: sign handling by sign_epi8 (zero inputs zero the product); word = 32 bytes
    "avx2": dict(dir="cnProc", suffix="AVX2", word=32, vec="simde__m256i", reg="ymm0", p="simde_mm256_", bgs=(1,),
                 consts=("ones   = simde_mm256_set1_epi8((int8_t)1);", "maxLLR = simde_mm256_set1_epi8((int8_t)127);"),
                 sgn_first="sgn  = simde_mm256_sign_epi8(ones, ymm0);", sgn_next="sgn  = simde_mm256_sign_epi8(sgn, ymm0);",
                 store=r"simde_mm256_sign_epi8\(min, sgn\);"),
    # AVX-512 (the canonical build): sign bit by xor, applied by conditional_negate = mask_sub_epi8(a, movepi8_mask(b), z, a)
    # (nrLDPC_cnProc_avx512.h:35): -min where the xor of the others' sign bits is set; word = 64 bytes; both base graphs
    "avx512": dict(dir="cnProc_avx512", suffix="AVX512", word=64, vec="simde__m512i", reg="zmm0", p="simde_mm512_", bgs=(1, 2),
                   consts=("ones = simde_mm512_set1_epi8((char)1);", "maxLLR = simde_mm512_set1_epi8((char)127);", "zeros  = simde_mm512_setzero_si512();"),
                   sgn_first="sgn  = simde_mm512_xor_si512(ones, zmm0);", sgn_next="sgn  = simde_mm512_xor_si512(sgn, zmm0);",
                   store=r"conditional_negate\(min, sgn,zeros\);"),
}


@pytest.mark.skipif(not (GEN / ".done").exists(), reason="generated headers not built")
@pytest.mark.parametrize("variant", ["avx2", "avx512"])
def test_generated_check_node_function_is_the_generic_formula_with_all_others_wiring(variant):
    """[D2] on the shipped code: nrLDPC_cnProc_BG*_R*_{AVX2,AVX512}.h as written by the reference's generators
    (generator_cnProc/cnProc_gen_BG1_avx2.c; generator_cnProc_avx512/cnProc_gen_BG{1,2}_avx512.c -- compiled from the
    reference tree) -- every loop of them must be
        sgn = SIGN(ones, x_a); min = abs_epi8(x_a);  then for every further input  min = min_epu8(min, abs_epi8(x)); sgn = SIGN(sgn, x);
        min = min_epu8(min, maxLLR = 127);  out = APPLY(min, sgn)
    (SIGN / APPLY = sign_epi8 / sign_epi8 in the AVX2 text, xor / conditional_negate in the AVX-512 text: the same value
    wherever the magnitude is not 0, and 0 either way where it is), and its inputs must be exactly the OTHER bit nodes' words
    of the same check-node group -- addresses from the reference's own LUTs -- for every output of every group,
    M = ceil(numCn Z / word) words each.  That is the formula the oracle restates from the generic nrLDPC_cnProc.h:81-118; the
    intrinsics' semantics are Intel's."""
    import re
    V = CN_VARIANTS[variant]
    W, reg, P, vec = V["word"], V["reg"], re.escape(V["p"]), re.escape(V["vec"])
    n_files = 0
    for BG in V["bgs"]:
      for R in RATES[BG]:
        txt = (GEN / V["dir"] / f"nrLDPC_cnProc_BG{BG}_R{R}_{V['suffix']}.h").read_text()
        for c in V["consts"]:
            assert re.sub(r"\s+", "", c) in re.sub(r"\s+", "", txt), c
        lut = _lut(BG, R)
        n_files += 1
        # every loop, statement by statement (white space ignored: the generators' spelling of it varies), as sets of the
        # input words that have gone into `min` and into `sgn`
        nows = lambda x: re.sub(r"\s+", "", x)
        S_FIRST, S_NEXT = nows(V["sgn_first"]), nows(V["sgn_next"])
        M_FIRST, M_NEXT = nows(f"min = {V['p']}abs_epi8({reg});"), nows(f"min = {V['p']}min_epu8(min, {V['p']}abs_epi8({reg}));")
        CAP = nows(f"min = {V['p']}min_epu8(min, maxLLR);")
        blocks, cur_m = [], None
        it = iter(txt.splitlines())
        for line in it:
            s = nows(line)
            m = re.fullmatch(rf"M=\((\d+)\*Z\+{W - 1}\)>>{W.bit_length() - 1};", s)
            if m:
                cur_m = int(m.group(1))
            if s not in ("for(inti=0;i<M;i++){", "for(i=0;i<M;i++){"):
                continue
            reads, in_min, in_sgn, cur, capped, out = [], None, None, None, False, None
            for line in it:
                s = nows(line)
                if s == "}":
                    break
                m = re.fullmatch(rf"{reg}=\(\({vec}\*\)cnProcBuf\)\[(\d+)\+i\];", s)
                w = re.fullmatch(rf"\(\({vec}\*\)cnProcBufRes\)\[(\d+)\+i\]=" + nows(V["store"]), s)
                assert out is None, "a statement behind the store"
                if m:
                    assert not capped
                    cur = int(m.group(1))
                    assert cur not in reads
                    reads.append(cur)
                elif s == S_FIRST:
                    assert in_sgn is None and cur is not None
                    in_sgn = {cur}
                elif s == M_FIRST:
                    assert in_min is None and cur is not None
                    in_min = {cur}
                elif s == M_NEXT:
                    assert in_min is not None and cur not in in_min and not capped
                    in_min.add(cur)
                elif s == S_NEXT:
                    assert in_sgn is not None and cur not in in_sgn
                    in_sgn.add(cur)
                elif s == CAP:
                    assert not capped and in_min == set(reads)
                    capped = True
                elif w:
                    assert capped and in_min == in_sgn == set(reads)
                    out = int(w.group(1))
                else:
                    raise AssertionError("statement outside the formula: " + line)
            assert out is not None
            blocks.append((cur_m, sorted(reads), out))
        assert blocks
        want = []
        for g in range(lut["G"]):
            if not lut["numCn"][g]:
                continue
            d, base, off = lut["bnInCn"][g], lut["startCn"][g] // W, lut["cnFull"][g] * 384 // W
            for j in range(d):
                want.append((lut["numCn"][g], sorted(base + k * off for k in range(d) if k != j), base + j * off))
        assert blocks == want, (variant, BG, R, len(blocks), len(want))
    assert n_files == 3 * len(V["bgs"])


@pytest.mark.skipif(not (GEN / ".done").exists(), reason="generated headers not built")
def test_generated_bit_node_sum_skips_the_one_check_columns_and_sums_every_message_once():
    """[D3] / [F5] on the shipped code: nrLDPC_bnProcPc_BG1_R*_AVX2.h as written by generator_bnProc/bnProcPc_gen_BG1_avx2.c.
    A group of bit nodes with N >= 2 check nodes: widen (cvtepi8_epi16) and add (adds_epi16) the N messages at
    start + k cnOffset, add the channel LLR, pack with saturation (packs_epi16 + the lane fix-up), store to llrRes -- addresses
    from the reference's LUTs; and there is NO code for the 1-check group: llrRes of the degree-1 parity columns (its
    [0, startAddrBnGroupsLlr[1]) entries) is never written, which is why the oracle reports them as 0."""
    import re
    for R in RATES[1]:
        txt = (GEN / "bnProcPc" / f"nrLDPC_bnProcPc_BG1_R{R}_AVX2.h").read_text()
        lut = _lut(1, R)
        secs = re.split(r"// Process group with (\d+) CNs", txt)
        groups = {int(secs[i]): secs[i + 1] for i in range(1, len(secs), 2)}
        assert 1 in groups and all(N in groups for N in range(2, 31) if lut["numBn"][N - 1])
        idx = 0
        for N in sorted(groups):
            body = groups[N]
            if N == 1 or lut["numBn"][N - 1] == 0:
                assert "simde" not in body and "=" not in body.replace("==", ""), (R, N)      # nothing but the comment
                continue
            idx += 1
            nb = lut["numBn"][N - 1]
            assert re.search(rf"M = \({nb}\*Z \+ 31\)>>5;", body), (R, N)
            assert f"&bnProcBuf    [{lut['startBn'][idx]}];" in body and f"&llrProcBuf   [{lut['startLlr'][idx]}];" in body
            assert f"&llrRes       [{lut['startLlr'][idx]}];" in body
            offs = sorted(int(x) for x in re.findall(r"ymm0 = simde_mm256_cvtepi8_epi16\(p_bnProcBuf\[(\d+) \+ j\]\);", body))
            assert offs == [k * nb * 384 // 16 for k in range(1, N)], (R, N, offs)
            assert body.count("ymmRes0 = simde_mm256_cvtepi8_epi16(p_bnProcBuf [j]);") == 1
            assert body.count("simde_mm256_adds_epi16(ymmRes0,") == N and body.count("simde_mm256_adds_epi16(ymmRes1,") == N   # N-1 messages + the LLR
            assert body.count("simde_mm256_cvtepi8_epi16(p_llrProcBuf[j]);") == 1
            assert "ymm0 = simde_mm256_packs_epi16(ymmRes0, ymmRes1);" in body and "p_llrRes[i] = simde_mm256_permute4x64_epi64(ymm0, 0xD8);" in body
        assert idx == sum(1 for x in lut["numBn"][1:] if x) and lut["startLlr"][1] == lut["numBn"][0] * 384


@pytest.mark.skipif(not (GEN / ".done").exists(), reason="generated headers not built")
def test_generated_bit_to_check_messages_subtract_from_the_clamped_sum():
    """[D5] on the shipped code: nrLDPC_bnProc_BG2_R*_AVX2.h as written by generator_bnProc/bnProc_gen_BG2_avx2.c: for every
    group with N >= 2 checks and every k < N:  bnProcBufRes[start + k off + i] = subs_epi8(llrRes[startLlr + i], bnProcBuf[start + k off + i])
    -- the saturating byte subtraction from llrRes, i.e. from the ALREADY CLAMPED sum -- and nothing for the 1-check group."""
    import re
    for R in RATES[2]:
        txt = (GEN / "bnProc" / f"nrLDPC_bnProc_BG2_R{R}_AVX2.h").read_text()
        lut = _lut(2, R)
        # (the generator does not print a heading for every group: the file is read as one sequence of stores)
        st = re.findall(r"\(\(simde__m256i\*\)bnProcBufRes\)\[(\d+) \+ i \] = simde_mm256_subs_epi8\(\(\(simde__m256i\*\)llrRes\)\[(\d+) \+ i \], "
                        r"\(\(simde__m256i\*\) bnProcBuf\)\[(\d+) \+ i\]\);", txt)
        ms = [int(x) for x in re.findall(r"M = \((\d+)\*Z \+ 31\)>>5;", txt)]
        want, want_m, idx = [], [], 0
        for N in range(2, 31):
            nb = lut["numBn"][N - 1]
            if not nb:
                continue
            idx += 1
            S, Lr, off = lut["startBn"][idx] // 32, lut["startLlr"][idx] // 32, nb * 384 // 32
            want += [(S + k * off, Lr, S + k * off) for k in range(N)]
            want_m.append(nb)
        assert [tuple(int(v) for v in t) for t in st] == want, R
        assert ms == want_m and txt.count("simde_mm256") == len(want)                       # nothing else: no 1-check group
        assert min(w[1] for w in want) == lut["startLlr"][1] // 32 == lut["numBn"][0] * 384 // 32


def test_the_recipe_holds_no_stand_ins():
    """What makes oracle/_ref a pin and not a restatement: the recipe compiles reference files WHERE THEY LIE with include paths
    into the reference tree only, the wrapper TU contains no arithmetic of its own, and nothing under oracle/ref_pin/ defines
    or fakes a SIMDE name (the survey-stage build did, which is why its vectors pin nothing)."""
    import re
    pin = RL.PIN_DIR
    mk = (pin / "Makefile").read_text()
    incs = re.findall(r"-I(\S+)", mk)
    assert incs and all(i.startswith("$(REF)") or i.startswith("$(CODING)") for i in incs), incs
    assert "REF ?= /root/reference" in mk and "CODING = $(REF)/openair1/PHY/CODING" in mk
    wrap = (pin / "ref_wrap.c").read_text()
    quoted = re.findall(r'#include "([^"]+)"', wrap)
    assert quoted == ["ldpc_generate_coefficient.c", "nrLDPC_types.h", "nrLDPC_init.h", "nrLDPC_mPass.h", "ref_pin.h"], quoted
    for name in ("ldpc_generate_coefficient.c", "nrLDPC_types.h", "nrLDPC_init.h", "nrLDPC_mPass.h"):
        assert not (pin / name).exists() and not list(RL.ROOT.glob(f"oracle/**/{name}"))       # included from the reference tree, not copied
    body = wrap[wrap.index('#include "ref_pin.h"'):]
    code = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    assert not re.search(r"\b(for|while)\s*\(", code) and not re.search(r"\^|<<|>>|%|\+\+", code)   # plumbing only: no loops, no arithmetic of the path
    for f in pin.iterdir():
        if f.suffix in (".c", ".h"):
            code = re.sub(r"/\*.*?\*/", "", f.read_text(), flags=re.S)
            assert not re.search(r"#\s*define\s+simde|typedef[^;]*simde_|simde_mm\w*\s*\(", code), f.name


# ---- the GENERIC functions no generator writes: parity check [D7][F6], hard decision [D9], the pass loop [D8], read as text ------
DEC_DIR = RL.REFERENCE / "openair1" / "PHY" / "CODING" / "nrLDPC_decoder"
needs_reference_text = pytest.mark.skipif(not (DEC_DIR / "nrLDPC_cnProc.h").exists(), reason="reference tree not present")


def _code(text):
    """C text without comments and without white space"""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^[ \t]*#[^\n]*$", "", text, flags=re.M)       # preprocessor lines
    return re.sub(r"\s+", "", text)


def _function_body(text, signature_start):
    """the text from `signature_start` to the closing brace of that function"""
    a = text.index(signature_start)
    while text.index(";", a) < text.index("{", a):          # a prototype: the definition follows
        a = text.index(signature_start, a + 1)
    depth, i = 0, text.index("{", a)
    while True:
        depth += text[i] == "{"
        depth -= text[i] == "}"
        i += 1
        if depth == 0:
            return text[a:i]


@needs_reference_text
@pytest.mark.parametrize("BG", [1, 2])
def test_generic_parity_check_text_group_by_group_and_the_mrem_rule(BG):
    """[D7][F6] nrLDPC_cnProcPc_BG1 / _BG2 (nrLDPC_cnProc.h:887-1526, 1528-1946), the function the shipped decoder calls for the
    parity-check stop (decoder.c:842-846) and for which no generator exists.  Read statement by statement: EVERY check-node group
    is `if (lut_numCnInCnGroups[g] > 0)`: M = numCn*Z lanes, Mrem = M & 31, M32 = ceil(M / 32); chunks 0 .. M32-2 in full,
    pcRes ^= movemask_epi8(adds_epi8(cnProcBuf, cnProcBufRes)) over exactly the group's D bit nodes at word offsets j*OFF + i; the
    LAST chunk with the same wiring, masked 0xFFFFFFFF >> (32 - Mrem) and counted ONLY `if (Mrem)` [F6]; early return of the group's
    word.  D and OFF must be the group's degree and stride in the reference-compiled LUTs.  The text model is then EXECUTED (numpy) on
    random buffers beside the hybrid's restatement -- which is what tests/golden/ref_decoder.npz was produced with."""
    import re
    text = (DEC_DIR / "nrLDPC_cnProc.h").read_text()
    body = _code(_function_body(text, f"static inline uint32_t nrLDPC_cnProcPc_BG{BG}("))
    head = (f"staticinlineuint32_tnrLDPC_cnProcPc_BG{BG}(t_nrLDPC_lut*p_lut,int8_t*cnProcBuf,int8_t*cnProcBufRes,uint16_tZ){{"
            "constuint8_t*lut_numCnInCnGroups=p_lut->numCnInCnGroups;constuint32_t*lut_startAddrCnGroups=p_lut->startAddrCnGroups;"
            "simde__m256i*p_cnProcBuf;simde__m256i*p_cnProcBufRes;uint32_tM;uint32_ti;uint32_tj;uint32_tpcRes=0;uint32_tpcResSum=0;"
            "uint32_tMrem;uint32_tM32;simde__m256iymm0,ymm1;")
    assert body.startswith(head), body[:400]
    rest = body[len(head):]
    inner = (r"for\(j=0;j<(?P<D{k}>\d+);j\+\+\)\{{ymm0=p_cnProcBuf\[j\*(?P<O{k}>\d+)\+i\];ymm1=p_cnProcBufRes\[j\*(?P<P{k}>\d+)\+i\];"
             r"pcRes\^=simde_mm256_movemask_epi8\(simde_mm256_adds_epi8\(ymm0,ymm1\)\);\}}")
    group = re.compile(
        r"if\(lut_numCnInCnGroups\[(?P<g>\d+)\]>0\)\{pcResSum=0;M=lut_numCnInCnGroups\[(?P<g2>\d+)\]\*Z;Mrem=M&31;M32=\(M\+31\)>>5;"
        r"p_cnProcBuf=\(simde__m256i\*\)&cnProcBuf\[lut_startAddrCnGroups\[(?P<g3>\d+)\]\];"
        r"p_cnProcBufRes=\(simde__m256i\*\)&cnProcBufRes\[lut_startAddrCnGroups\[(?P<g4>\d+)\]\];"
        r"for\(i=0;i<\(M32-1\);i\+\+\)\{pcRes=0;" + inner.format(k=1) + r"pcResSum\|=pcRes;\}"
        r"pcRes=0;" + inner.format(k=2) +
        r"if\(Mrem\)pcResSum\|=\(pcRes&\(0xFFFFFFFF>>\(32-Mrem\)\)\);if\(pcResSum>0\)\{returnpcResSum;\}\}")
    groups, pos = [], 0
    while True:
        m = group.match(rest, pos)
        if not m:
            break
        d = m.groupdict()
        assert d["g"] == d["g2"] == d["g3"] == d["g4"] and d["D1"] == d["D2"] and len({d["O1"], d["P1"], d["O2"], d["P2"]}) == 1, d
        groups.append((int(d["g"]), int(d["D1"]), int(d["O1"])))
        pos = m.end()
    assert rest[pos:] == "returnpcResSum;}", rest[pos:pos + 200]       # nothing else in the function
    lut = _lut(BG, 13 if BG == 1 else 15)
    assert [g for g, _, _ in groups] == list(range(lut["G"]))           # every group, in LUT order
    for g, D, off in groups:
        assert D == lut["bnInCn"][g] and off == lut["cnFull"][g] * 384 // 32, (g, D, off)
    # ---- run the text's model beside the restatement --------------------------------------------------------------------------
    L = RL.lib()
    rng = np.random.default_rng(20 + BG)
    ncn = L.ref_size_cn_proc_buf()

    def text_model(numCn, start, Z, a, b):
        sgn = (np.clip(a.astype(np.int16) + b.astype(np.int16), -128, 127) < 0)          # movemask(adds_epi8)
        for g, D, off in groups:
            if numCn[g] == 0:
                continue
            M = numCn[g] * Z
            Mrem, M32 = M & 31, (M + 31) >> 5
            word = 0
            for i in range(M32):
                pc = 0
                for j in range(D):
                    lanes = sgn[start[g] + 32 * (j * off + i):start[g] + 32 * (j * off + i) + 32]
                    pc ^= int(np.packbits(lanes, bitorder="little").view(np.uint32)[0])
                if i < M32 - 1:
                    word |= pc
                elif Mrem:
                    word |= pc & (0xFFFFFFFF >> (32 - Mrem))
            if word:
                return word
        return 0

    seen_nonzero = seen_zero = seen_f6 = 0
    for Z in (384, 352, 208, 36, 22, 13, 8, 2):
        for R in RATES[BG]:
            h = L.ref_dec_new(BG, Z, R)
            if not h:
                continue
            G = L.ref_dec_numCnGroups(h)
            numCn = [L.ref_dec_numCnInCnGroups(h)[g] for g in range(G)]
            start = [L.ref_dec_startAddrCnGroups(h)[g] for g in range(G)]
            for case in range(6):
                a = np.zeros(ncn + 64, np.int8)
                b = np.zeros(ncn + 64, np.int8)
                if case < 2:     # dense random: some group fails
                    a[:ncn] = rng.integers(-128, 128, ncn)
                    b[:ncn] = rng.integers(-128, 128, ncn)
                else:            # all satisfied, then ONE lane of ONE group flipped: in a counted chunk, or in the last chunk
                    a[:ncn] = rng.integers(0, 128, ncn)
                    b[:ncn] = rng.integers(0, 100, ncn)
                    if case >= 3:
                        live = [g for g in range(G) if numCn[g]]
                        g = live[int(rng.integers(len(live)))]
                        M = numCn[g] * Z
                        lane = M - 1 - int(rng.integers(min(M, 32))) if case >= 4 else int(rng.integers(M))
                        a[start[g] + lane] = -100
                        seen_f6 += (M & 31) == 0 and lane >= M - 32
                want = L.ref_hybrid_cnProcPc(h, Z, a.ctypes.data, b.ctypes.data)
                got = text_model(numCn, start, Z, a, b)
                assert got == want, (BG, Z, R, case, hex(got), hex(want))
                seen_nonzero += want != 0
                seen_zero += want == 0
            L.ref_dec_free(h)
    assert seen_nonzero > 20 and seen_zero > 20 and seen_f6 > 3, (seen_nonzero, seen_zero, seen_f6)


@needs_reference_text
def test_generic_hard_decision_text_and_bit_order():
    """[D9] nrLDPC_llr2bit / nrLDPC_llr2bitPacked (nrLDPC_bnProc.h:1321-1345, 1353-1380) as text: bit = (0 > llr) per byte resp.
    movemask of the byte-reversed-within-8 load -- bit 8b of the input in the MSB of output byte b -- for the numLLR >> 5 full
    words, and a scalar tail `(p[i] < 0) << ((7 - i) + 16 * (i / 8))` written only `if (Mr)`.  The text's model runs beside the
    hybrid's restatement for lengths with and without a tail."""
    text = (DEC_DIR / "nrLDPC_bnProc.h").read_text()
    b1 = _code(_function_body(text, "static inline void nrLDPC_llr2bit("))
    assert b1 == ("staticinlinevoidnrLDPC_llr2bit(int8_t*out,int8_t*llrOut,uint16_tnumLLR){simde__m256i*p_llrOut=(simde__m256i*)llrOut;"
                  "simde__m256i*p_out=(simde__m256i*)out;constuint32_tM=numLLR>>5;constuint32_tMr=numLLR&31;"
                  "constsimde__m256i*p_zeros=(simde__m256i*)zeros256_epi8;constsimde__m256i*p_ones=(simde__m256i*)ones256_epi8;"
                  "for(uint32_ti=0;i<M;i++){*p_out++=simde_mm256_and_si256(*p_ones,simde_mm256_cmpgt_epi8(*p_zeros,*p_llrOut));p_llrOut++;}"
                  "int8_t*p_llrOut8=(int8_t*)p_llrOut;int8_t*p_out8=(int8_t*)p_out;for(uint32_ti=0;i<Mr;i++)p_out8[i]=p_llrOut8[i]<0;}"), b1
    b2 = _code(_function_body(text, "static inline void nrLDPC_llr2bitPacked("))
    assert b2 == ("staticinlinevoidnrLDPC_llr2bitPacked(int8_t*out,int8_t*llrOut,uint16_tnumLLR){"
                  "constuint8_tconstShuffle_256_epi8[32]__attribute__((aligned(32)))={7,6,5,4,3,2,1,0,15,14,13,12,11,10,9,8,7,6,5,4,3,2,1,0,15,14,13,12,11,10,9,8};"
                  "constsimde__m256i*p_shuffle=(simde__m256i*)constShuffle_256_epi8;simde__m256i*p_llrOut=(simde__m256i*)llrOut;"
                  "uint32_t*p_bits=(uint32_t*)out;constuint32_tM=numLLR>>5;constuint32_tMr=numLLR&31;"
                  "for(uint32_ti=0;i<M;i++){constsimde__m256iinPerm=simde_mm256_shuffle_epi8(*p_llrOut,*p_shuffle);"
                  "*p_bits++=simde_mm256_movemask_epi8(inPerm);p_llrOut++;}"
                  "if(Mr){constint8_t*p_llrOut8=(int8_t*)p_llrOut;uint32_tbitsTmp=0;"
                  "for(uint32_ti=0;i<Mr;i++)bitsTmp|=(p_llrOut8[i]<0)<<((7-i)+(16*(i/8)));*p_bits=bitsTmp;}}"), b2
    shuffle = [7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8] * 2      # (per 128-bit lane, as shuffle_epi8 works)

    def packed_model(llr):
        n = len(llr)
        M, Mr = n >> 5, n & 31
        words = []
        for i in range(M):
            w = llr[32 * i:32 * i + 32]
            perm = np.concatenate([w[:16][shuffle[:16]], w[16:][shuffle[16:]]])
            words.append(int(np.packbits(perm < 0, bitorder="little").view(np.uint32)[0]))
        if Mr:
            t = 0
            for i in range(Mr):
                t |= int(llr[32 * M + i] < 0) << ((7 - i) + 16 * (i // 8))
            words.append(t)
        return np.array(words, dtype=np.uint32).view(np.uint8)

    L = RL.lib()
    rng = np.random.default_rng(9)
    for n in (64, 26112, 52 * 2, 68 * 3, 27 * 5, 35 * 7, 17 * 13, 32 * 9, 27 * 384):
        llr = np.zeros(n + 64, np.int8)
        llr[:n] = rng.integers(-128, 128, n)
        out = np.full(n + 64, 0x5A, np.uint8)
        L.ref_hybrid_llr2bit(out.ctypes.data, llr.ctypes.data, n, 1)
        want = packed_model(llr[:n])
        assert np.array_equal(out[:len(want)], want) and (out[len(want):] == 0x5A).all(), n
        assert np.array_equal(np.unpackbits(want)[:n], (llr[:n] < 0).astype(np.uint8))       # = MSB first, as the oracle packs
        L.ref_hybrid_llr2bit(out.ctypes.data, llr.ctypes.data, n, 0)
        assert np.array_equal(out[:n], (llr[:n] < 0).astype(np.uint8)), n                   # and(ones, cmpgt(zeros, llr)) / llr < 0


@needs_reference_text
def test_pass_loop_text_of_the_decoder_core():
    """[D8] + iteration control (a10, a12) as text, nrLDPC_decoder.c: the entry point aborts the transport block when the core
    returns more than numMaxIter (:189-193); the core counts the unconditional first pass as 1 and loops
    `while ((numIter <= numMaxIter) && (pcRes != 0))`, incrementing first, returning numMaxIter + 2 on an abort (:552-559); the
    stop test at the end of a pass is cnProcPc when there is no predicate, else -- only `if (numIter > 2)` -- reorder, hard decision
    in the caller's output mode INTO p_out, predicate on (p_out, E, crc_type), break (:841-861); the output is produced after the
    loop only when there is no predicate (:864-879); the pass count is returned (:880).  The hybrid decoder and both oracle
    restatements follow exactly this skeleton."""
    text = (DEC_DIR / "nrLDPC_decoder.c").read_text()
    entry = _code(_function_body(text, "int32_t LDPCdecoder(t_nrLDPC_dec_params* p_decParams,"))
    assert ("numLLR=nrLDPC_init(p_decParams,p_lut);intnumIter=nrLDPC_decoder_core(p_llr,p_out,numLLR,p_lut,p_decParams,p_profiler,ab);"
            "if(numIter>p_decParams->numMaxIter){") in entry and entry.endswith("set_abort(ab,true);}returnnumIter;}"), entry[-300:]
    core = _code(_function_body(text, "static inline uint32_t nrLDPC_decoder_core(int8_t* p_llr,"))
    order = [
        "uint32_tnumIter=1;int32_tpcRes=1;while((numIter<=numMaxIter)&&(pcRes!=0)){numIter++;if(check_abort(ab)){numIter=numMaxIter+2;break;}",
        "if(!p_decParams->check_crc){",
        "if(BG==1)pcRes=nrLDPC_cnProcPc_BG1(p_lut,cnProcBuf,cnProcBufRes,Z);elsepcRes=nrLDPC_cnProcPc_BG2(p_lut,cnProcBuf,cnProcBufRes,Z);",
        "}else{if(numIter>2){int8_tllrOut[NR_LDPC_MAX_NUM_LLR]__attribute__((aligned(64)))={0};"
        "int8_t*p_llrOut=outMode==nrLDPC_outMode_LLRINT8?p_out:llrOut;nrLDPC_llrRes2llrOut(p_lut,p_llrOut,llrRes,Z,BG);"
        "if(outMode==nrLDPC_outMode_BIT)nrLDPC_llr2bitPacked(p_out,p_llrOut,numLLR);elsenrLDPC_llr2bit(p_out,p_llrOut,numLLR);"
        "if(p_decParams->check_crc((uint8_t*)p_out,p_decParams->E,p_decParams->crc_type)){",
        "break;}}}}if(!p_decParams->check_crc){int8_tllrOut[NR_LDPC_MAX_NUM_LLR]__attribute__((aligned(64)))={0};"
        "int8_t*p_llrOut=outMode==nrLDPC_outMode_LLRINT8?p_out:llrOut;",
        "nrLDPC_llrRes2llrOut(p_lut,p_llrOut,llrRes,Z,BG);",
        "if(outMode==nrLDPC_outMode_BIT)nrLDPC_llr2bitPacked(p_out,p_llrOut,numLLR);elsenrLDPC_llr2bit(p_out,p_llrOut,numLLR);",
        "}returnnumIter;}",
    ]
    pos = 0
    for frag in order:
        at = core.find(frag, pos)
        assert at >= 0, frag
        pos = at + len(frag)
    assert pos == len(core)                                            # the function ends with the last fragment
    # nothing between the pieces of the stop test but profiler / debug macros
    tail = core[core.index(order[1]):]
    import re
    between = tail
    for frag in order[1:]:
        between = between.replace(frag, "", 1)
    between = re.sub(r"NR_LDPC_PROFILER_DETAIL\((start|stop)_meas\(&p_profiler->\w+\)\);", "", between)
    between = re.sub(r"LOG_D\(PHY,\"[^\"]*\"\);", "", between)
    assert between == "", between[:300]
    # the first pass runs no parity check and the loop body holds exactly one call of each node function per base graph / rate
    first = core[:core.index(order[0])]
    assert "nrLDPC_cnProcPc" not in first and "check_crc" not in first
