"""The reference's transport-channel acceptance runs (nr_ulschsim / nr_dlschsim, cmake_targets/autotests/
test_case_list.xml:232-277) reproduced on libldpc_hip.so -- SURVEY 8(f) row f4, second half.

CPU part: the restated MAC helpers the sims call (MCS tables, nr_compute_tbs, nr_get_G, get_BG) and the whole flow on the
oracle back end.  GPU part: the CI argument sets through the transport-block chain (`PUSCH test OK` / `PDSCH test OK`),
and near the waterfall identical ACK / pass-count / payload records from the chain, from per-segment
LDPCencoder / LDPCdecoder calls and from the oracle on the same seeds.
"""
import io
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import nr_schsim as S

TESTS = Path(__file__).resolve().parent
# test_case_list.xml:240-242 (nr_dlschsim) and :268-271 (nr_ulschsim)
CI_DL = ["-R 106 -m9 -s13 -n100", "-R 217 -m15 -s15 -n100", "-R 273 -m19 -s20 -n100"]
CI_UL = CI_DL + ["-R 106 -m9 -s13 -n100 -y4 -z4 -W4"]


def _run(direction, argline, backend, n=None, seed=7):
    argv = argline.split() + ["--backend", backend, "--seed", str(seed)]
    if n is not None:
        argv += ["-n", str(n)]
    buf = io.StringIO()
    res = S.run(direction, S.parser(direction).parse_args(argv), out=buf)
    return res, buf.getvalue()


def test_mac_helpers_the_sims_call():
    # TS 38.214 5.1.3.2 worked by hand: 106 PRB x 12 symbols, 6 DMRS REs, QPSK 679/1024 -> N_info 19399, n = 9,
    # N'_info = 19456, C = 3, TBS = 19464; the sims' default allocation (50 PRB) -> 9224
    assert S.nr_compute_tbs(2, 6790, 106, 12, 6, 0, 0, 1) == 19464
    assert S.nr_compute_tbs(2, 6790, 50, 12, 6, 0, 0, 1) == 9224
    # small blocks come from the table (3824 is its last entry), R <= 1/4 uses 3816-bit segments
    assert S.nr_compute_tbs(2, 1200, 1, 12, 6, 0, 0, 1) in S.TBS_TABLE
    assert S.nr_compute_tbs(2, 1200, 273, 12, 6, 0, 0, 1) % 8 == 0
    assert S.nr_get_G(50, 12, 6, 1, 0, 2, 1) == 13800 and S.nr_get_G(50, 12, 12, 1, 0, 2, 4) == 52800
    assert S.get_BG(292, 9480) == 2 and S.get_BG(3824, 6660) == 2 and S.get_BG(3832, 6660) == 1 and S.get_BG(100000, 2500) == 2
    # every TBS the five MCS tables produce for the sims' allocation is a multiple of 8 and segments legally
    import oracle_lib as O
    for direction, tabs in S.MCS_TABLES.items():
        for ti, tab in enumerate(tabs):
            assert len(tab) == 32
            for Imcs, (Qm, R) in enumerate(tab):
                if R == 0:
                    continue
                tbs = S.nr_compute_tbs(Qm, R, 50, 12, 6, 0, 0, 1)
                assert tbs % 8 == 0
                assert O.segmentation(None, O.len_with_crc(1, tbs), S.get_BG(tbs, R))["Kb"] > 0, (direction, ti, Imcs)


@pytest.mark.parametrize("direction,argline,ok", [("ul", CI_UL[0], "PUSCH test OK"), ("ul", CI_UL[3], "PUSCH test OK"),
                                                  ("dl", CI_DL[1], "PDSCH test OK"), ("dl", CI_DL[2], "PDSCH test OK")])
def test_ci_argument_sets_on_the_oracle_backend(built, direction, argline, ok):
    res, text = _run(direction, argline, "oracle", n=4)
    assert ok in text and res["n_errors"] == 0
    assert "BLER 0.000000 (false positive 0.000000)" in text


def test_waterfall_and_exit_code_on_the_oracle_backend(built):
    res, text = _run("dl", "-m19 -s0.4 -S1.7 --snr-step 0.4", "oracle", n=12)
    blers = [1.0 - r["ack"].mean() for r in res["records"]]
    assert blers[0] > 0.5 and blers[-1] < 0.2 and blers == sorted(blers, reverse=True)
    # the executable's exit code is n_errors of the last SNR point (dlschsim.c:682)
    p = subprocess.run([sys.executable, str(TESTS / "ulschsim_hip.py"), "-m9", "-s0", "-S0.05", "-n3", "--oracle"],
                       capture_output=True, text=True)
    assert p.returncode == 3 and "PUSCH test OK" not in p.stdout and "BLER 1.000000" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("direction,argline", [("ul", a) for a in CI_UL] + [("dl", a) for a in CI_DL])
def test_reference_ci_acceptance_through_the_transport_block_chain(hip, direction, argline):
    res, text = _run(direction, argline, "chain")
    assert ("PUSCH test OK" if direction == "ul" else "PDSCH test OK") in text, text
    assert res["n_errors"] == 0 and len(res["records"]) == 1 and len(res["records"][0]["ack"]) == 100
    assert res["records"][0]["payload_ok"].all()


@pytest.mark.gpu
@pytest.mark.parametrize("direction,argline", [("ul", "-R 106 -m9 -s2.3 -S2.35"), ("dl", "-R 217 -m15 -s1.8 -S1.85"),
                                               ("dl", "-R 273 -m19 -s0.8 -S0.85"), ("ul", "-m9 -s2.5 -S2.55 -y4 -z4 -W4"),
                                               ("ul", "-m3 -r 20 -q 2 -s-9 -S-8.95"), ("dl", "-m27 -r 273 -l 13 -s5.2 -S5.25")])
def test_identical_records_from_chain_segment_calls_and_oracle_near_the_waterfall(hip, direction, argline):
    n = 24
    ref, _ = _run(direction, argline, "oracle", n=n)
    frac = ref["records"][0]["ack"].mean()
    for backend in ("chain", "segment"):
        res, text = _run(direction, argline, backend, n=n)
        for a, b in zip(ref["records"], res["records"]):
            assert np.array_equal(a["ack"], b["ack"]), (backend, text)
            assert np.array_equal(a["iter_max"], b["iter_max"]), (backend, a["iter_max"], b["iter_max"])
            assert np.array_equal(a["payload_ok"], b["payload_ok"]), backend
        if backend == "segment":      # per-segment pass counts of every trial, up to the first failed segment (abort)
            for p_ref, p_seg in zip(ref["backend"].segment_passes, res["backend"].segment_passes):
                k = next((i for i, v in enumerate(p_ref) if v > 5), len(p_ref) - 1)
                assert p_ref[:k + 1] == p_seg[:k + 1]
    assert 0.0 < frac < 1.0 or n < 8, f"operating point not on the waterfall (ACK fraction {frac})"
