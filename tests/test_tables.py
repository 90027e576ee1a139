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
