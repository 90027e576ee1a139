import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the product library, the oracle and the emulator are built (all compile without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def hip(built):
    """The product library, initialised on the GPU. Fails loudly (no fallback) when there is no device."""
    import hashlib
    import json
    import openairinterface5g_amd as pkg
    # the GPU box has no reference tree: prove here, in every `-m gpu` run, that the code tables compiled into the product
    # and into the oracle are the ones validated against nrLDPC_lut.h / bgs/BG*_I* in the development container
    # (tests/test_tables.py, which is not gpu-marked and therefore not part of the driver's GPU run)
    pinned = json.loads((ROOT / "tests" / "golden" / "table_hashes.json").read_text())
    for name, path in (("product", ROOT / "openairinterface5g_amd" / "csrc" / "nr_ldpc_bg_tables.h"),
                       ("oracle", ROOT / "oracle" / "oracle_bg_tables.h")):
        assert hashlib.sha256(path.read_bytes()).hexdigest() == pinned[name]["sha256"], f"{name} tables differ from the validated ones"
    lib, so_src = pkg.ldpc.LIB_PATH, ROOT / "openairinterface5g_amd" / "csrc" / "nr_ldpc_bg_tables.h"
    assert lib.stat().st_mtime >= so_src.stat().st_mtime, "libldpc_hip.so is older than its table header"
    pkg.LDPCinit()
    return pkg
