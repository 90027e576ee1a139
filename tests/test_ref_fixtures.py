"""The oracle against the committed outputs of oracle/_ref (tests/golden/ref_*.npz, tools/make_ref_fixtures.py): runs
everywhere, with or without the reference tree.  What the fixtures pin and what they cannot: oracle/ref_pin/ref_pin.h."""
import numpy as np

import oracle_lib as O
from common import load_ref_code_words, load_ref_decoder_vectors, load_ref_transport_blocks


def test_oracle_encoder_reproduces_reference_code_words():
    n = 0
    for v in load_ref_code_words():
        assert np.array_equal(O.encode(v["BG"], v["Z"], v["info"], v["Kb"]), v["coded"]), (v["BG"], v["Z"], v["Kb"])
        n += 1
    assert n >= 2 * 2 * 51


def test_oracle_decoders_reproduce_reference_hybrid_runs():
    """Scalar AND vectorised restatement: pass counts and every output byte, all output modes, parity and CRC stop."""
    n = 0
    for v in load_ref_decoder_vectors():
        for vec in (False, True):
            if not vec and v["Z"] > 200 and v["numMaxIter"] == 8 and v["outMode"] == 0 and v["use_crc"]:
                continue                                   # keep the scalar O(d^2) restatement's share of the run short
            n_it, out = O.decode(v["BG"], v["Z"], v["R"], v["llr"], v["numMaxIter"], v["outMode"], v["use_crc"], v["E"],
                                 v["crc_type"], out_init=v["out_init"], vec=vec)
            assert n_it == v["n_iter"], (vec, v["BG"], v["Z"], v["R"], v["numMaxIter"], v["outMode"], v["use_crc"])
            assert np.array_equal(out, v["out"]), (vec, v["BG"], v["Z"], v["R"], v["numMaxIter"], v["outMode"])
        n += 1
    assert n >= 400


def test_oracle_dlsch_chain_reproduces_reference_code_words():
    """TB CRC + segmentation + encoder + rate matching + interleaving of the oracle chain = the expected output derived
    from the reference-compiled code words."""
    for t in load_ref_transport_blocks():
        assert np.array_equal(O.dlsch_encode(t["tb"], t["payload"]), t["coded"]), t["tb"]
