"""38.212 base-graph tables and the code descriptors built from them (CPU only).

When the reference tree is present (development container) the tables are cross-checked against BOTH copies the
reference holds as data: nrLDPC_decoder_LYC/bgs/BG*_I* (the generator's source) and, independently formatted,
nrLDPC_decoder/nrLDPC_lut.h (circShift_*, posBnInCnProcBuf_*) + nrLDPCdecoder_defs.h (group sizes)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as O

REF = Path("/root/reference/openair1/PHY/CODING")
RATES = {1: (13, 23, 89), 2: (15, 13, 23)}


@pytest.fixture(scope="module")
def emul(built):
    L = C.CDLL(str(Path(__file__).resolve().parent / "emul" / "libldpc_emul.so"))
    L.ldpc_emul_desc_edges.argtypes = [C.c_int] * 3 + [C.c_void_p] * 5
    return L


def product_graph(emul, BG, Z, R):
    dims = np.zeros(8, np.int32)
    rp, col, sh, pc = (np.zeros(n, np.int32) for n in (48, 320, 320, 48))
    rc = emul.ldpc_emul_desc_edges(BG, Z, R, *(a.ctypes.data for a in (dims, rp, col, sh, pc)))
    assert rc == 0, (BG, Z, R, rc)
    nrows, ncols, ncore, ne = (int(v) for v in dims[:4])
    return dict(nrows=nrows, ncols=ncols, ncore=ncore, nedges=ne, row_ptr=rp[:nrows + 1], col=col[:ne], shift=sh[:ne],
                pc_lo=pc[:nrows], f_ok=int(dims[4]), lds=int(dims[5]), f_lds=int(dims[6]))


def test_structure_and_product_vs_oracle_graph(emul):
    """The product's descriptor and the oracle's graph agree; 38.212 structure the kernels rely on holds."""
    for BG in (1, 2):
        kbf = 22 if BG == 1 else 10
        for Z in O.LIFT_SIZES:
            for R in RATES[BG]:
                g = O.graph(BG, Z, R)
                p = product_graph(emul, BG, Z, R)
                assert (g.nrows, g.ncols, g.ncore, g.nedges) == (p["nrows"], p["ncols"], p["ncore"], p["nedges"])
                assert list(g.row_ptr[:g.nrows + 1]) == list(p["row_ptr"])
                assert list(g.col[:g.nedges]) == list(p["col"]) and list(g.shift[:g.nedges]) == list(p["shift"])
                assert p["ncols"] == O.NCOLS[(BG, R)] and p["nrows"] == p["ncols"] - kbf
                for r in range(p["nrows"]):
                    cols = p["col"][p["row_ptr"][r]:p["row_ptr"][r + 1]]
                    assert list(cols) == sorted(set(cols)) and cols[-1] < p["ncols"]
                    if r >= 4:   # one degree-1 extension column per extension row, identity circulant
                        assert cols[-1] == kbf + r and p["shift"][p["row_ptr"][r + 1] - 1] == 0
                        assert (cols[:-1] < p["ncore"]).all()
                    else:
                        assert (cols < p["ncore"]).all()
                assert p["lds"] <= 160 * 1024 and (not p["f_ok"] or p["f_lds"] <= 160 * 1024)
                assert bool(p["f_ok"]) == (Z % 4 == 0 and Z >= 8)
    assert O.lib().oracle_ldpc_ils(384) == 1 and O.lib().oracle_ldpc_ils(256) == 0 and O.lib().oracle_ldpc_ils(17) == -1


def test_parity_check_lane_exemption(emul):
    """[F6] pc_lo: the reference's parity check drops the last 32-lane chunk of a CN group holding a multiple of
    32 lanes (nrLDPC_cnProc.h:964-965)."""
    p = product_graph(emul, 1, 384, 13)
    deg = np.diff(p["row_ptr"])
    for d in set(deg):
        rows = np.flatnonzero(deg == d)
        assert (p["pc_lo"][rows[:-1]] == 384).all() and p["pc_lo"][rows[-1]] == 352   # every group: n*384 % 32 == 0
    p = product_graph(emul, 2, 208, 15)          # 208 = 6.5 * 32: groups with an odd number of rows are fully checked
    deg = np.diff(p["row_ptr"])
    for d in set(deg):
        rows = np.flatnonzero(deg == d)
        expect = 208 - 32 if (len(rows) * 208) % 32 == 0 else 208
        assert p["pc_lo"][rows[-1]] == expect and (p["pc_lo"][rows[:-1]] == 208).all()
    p = product_graph(emul, 2, 8, 15)            # Z < 32: the dropped chunk spans several rows
    deg = np.diff(p["row_ptr"])
    rows = np.flatnonzero(deg == 4)              # 20 rows * 8 lanes = 160 = 5 * 32 -> last 32 lanes = last 4 rows
    assert (p["pc_lo"][rows[-4:]] == 0).all() and (p["pc_lo"][rows[:-4]] == 8).all()


def _parse_arrays(text, pattern):
    out = {}
    for m in re.finditer(pattern + r"\s*(\[\d+\])+\s*=\s*(\{.*?\});", text, re.S):
        body = m.group(m.lastindex)
        rows = [[int(v) for v in re.findall(r"-?\d+", r)] for r in re.findall(r"\{([^{}]*)\}", body)]
        out[m.groups()[:-2]] = rows
    return out


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present (development container only)")
def test_tables_against_reference_lut(emul):
    lut = (REF / "nrLDPC_decoder" / "nrLDPC_lut.h").read_text()
    defs = (REF / "nrLDPC_decoder" / "nrLDPCdecoder_defs.h").read_text()
    shifts = _parse_arrays(lut, r"circShift_BG(\d)_Z(\d+)_CNG(\d+)")
    posbn = _parse_arrays(lut, r"posBnInCnProcBuf_BG(\d)_CNG(\d+)")
    ncn = {}
    for m in re.finditer(r"lut_numCnInCnGroups_BG(\d)_R(\d+)\[[^\]]*\]\s*=\s*\{([^}]*)\}", defs):
        ncn[(int(m.group(1)), int(m.group(2)))] = [int(v) for v in re.findall(r"\d+", m.group(3))]
    groups = {1: [3, 4, 5, 6, 7, 8, 9, 10, 19], 2: [3, 4, 5, 6, 8, 10]}
    checked = 0
    for BG in (1, 2):
        for Z in O.LIFT_SIZES:
            p = product_graph(emul, BG, Z, 13 if BG == 1 else 15)
            deg = np.diff(p["row_ptr"])
            for d in groups[BG]:
                rows = np.flatnonzero(deg == d)
                ref_s = shifts[(str(BG), str(Z), str(d))]     # [edge j in row][CN i of the group]
                ref_c = posbn[(str(BG), str(d))]
                assert len(ref_s) == d and len(ref_s[0]) == len(rows), (BG, Z, d)
                for i, r in enumerate(rows):
                    e0 = p["row_ptr"][r]
                    for j in range(d):
                        assert p["col"][e0 + j] == ref_c[j][i], (BG, Z, d, i, j)
                        assert p["shift"][e0 + j] == ref_s[j][i], (BG, Z, d, i, j)
                        checked += 1
        for R in RATES[BG]:                                  # rows per group in every decoder-rate mode
            p = product_graph(emul, BG, 384, R)
            deg = np.diff(p["row_ptr"])
            assert [int((deg == d).sum()) for d in groups[BG]] == ncn[(BG, R)], (BG, R)
    assert checked == 51 * (316 + 197)


def _header_arrays(path, prefix):
    txt = Path(path).read_text()
    return {m.group(1)[len(prefix):]: [int(v) for v in re.findall(r"-?\d+", m.group(2))]
            for m in re.finditer(r"(%s\w+)\s*(?:\[\d+\])+\s*=\s*\{(.*?)\};" % prefix, txt, re.S)}


ROOT = Path(__file__).resolve().parent.parent
PRODUCT_TABLES = ROOT / "openairinterface5g_amd" / "csrc" / "nr_ldpc_bg_tables.h"
ORACLE_TABLES = ROOT / "oracle" / "oracle_bg_tables.h"


def test_product_and_oracle_tables_come_from_two_routes_and_agree():
    """The product's table header (generated from nrLDPC_decoder_LYC/bgs/BG*_I*, tools/gen_bg_tables.py) and the oracle's
    own (generated from nrLDPC_lut.h + the .cu row-degree lists, tools/gen_oracle_tables.py) are different files that
    hold the same numbers; both are the validated ones (sha256 recorded in the development container, where
    test_tables_against_reference_lut and the generators' own cross-checks run) -- this part also runs on the GPU box."""
    import hashlib
    import json
    a, b = _header_arrays(PRODUCT_TABLES, "nr_ldpc_"), _header_arrays(ORACLE_TABLES, "oracle_")
    assert sorted(a) == sorted(b) == ["bg1_col", "bg1_row_deg", "bg1_shift", "bg2_col", "bg2_row_deg", "bg2_shift", "lift_sizes"]
    for k in a:
        assert a[k] == b[k], k
    assert "nr_ldpc_bg_tables.h" not in "".join((ROOT / "oracle" / f).read_text() for f in
                                                ("oracle_ldpc_decoder.c", "oracle_ldpc_encoder.c", "oracle_ldpc_decoder_vec.c"))
    pinned = json.loads((GOLDEN_DIR / "table_hashes.json").read_text())
    for name, path in (("product", PRODUCT_TABLES), ("oracle", ORACLE_TABLES)):
        assert hashlib.sha256(path.read_bytes()).hexdigest() == pinned[name]["sha256"], name


GOLDEN_DIR = ROOT / "tests" / "golden"


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present (development container only)")
def test_bit_node_side_and_k0_against_reference_data(emul):
    """The bit-node side of the reference's tables, read as data: lut_numBnInBnGroups_BG*_R* (nrLDPCdecoder_defs.h: how
    many columns of each degree a rate mode has) against the column degrees of the oracle's / product's graph, and
    index_k0 (nr_rate_matching.c:34) against the k0 the oracle and the product compute."""
    defs = (REF / "nrLDPC_decoder" / "nrLDPCdecoder_defs.h").read_text()
    for m in re.finditer(r"lut_numBnInBnGroups_BG(\d)_R(\d+)\[[^\]]*\]\s*=\s*\{([^}]*)\}", defs):
        BG, R = int(m.group(1)), int(m.group(2))
        ref = [int(v) for v in re.findall(r"\d+", m.group(3))]           # ref[d-1] = columns of degree d
        for Z in (384, 208, 6):
            g = O.graph(BG, Z, R)
            deg = np.bincount(np.asarray(g.col[:g.nedges]), minlength=g.ncols)
            got = [int((deg == d).sum()) for d in range(1, len(ref) + 1)]
            assert got == ref, (BG, R, Z)
            p = product_graph(emul, BG, Z, R)
            assert np.array_equal(np.bincount(p["col"], minlength=p["ncols"]), deg)
    rm = (REF / "nr_rate_matching.c").read_text()
    m = re.search(r"index_k0\[2\]\[4\]\s*=\s*\{\{([^}]*)\},\s*\{([^}]*)\}\}", rm)
    k0 = [[int(v) for v in re.findall(r"\d+", m.group(i))] for i in (1, 2)]
    assert k0 == [[0, 17, 33, 56], [0, 13, 25, 43]]
    import openairinterface5g_amd as pkg
    for BG in (1, 2):
        for rv in range(4):
            for Z in (384, 96, 7):
                N = (66 if BG == 1 else 50) * Z
                E = 4 * Z
                # nr_get_R_ldpc_decoder's infoBits = k0 index * Z + E (nr_rate_matching.c:399) is llrLen on round 0
                for f in (O.get_R, lambda rv_, E_, BG_, Z_, a, b: pkg.ldpc.nr_get_R_ldpc_decoder(rv_, E_, BG_, Z_, a, b)):
                    _, llrlen = f(rv, E, BG, Z, 0, 0)
                    assert llrlen == min(k0[BG - 1][rv] * Z + E, N), (BG, rv, Z)
