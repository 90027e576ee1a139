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
    import openairinterface5g_amd as pkg
    pkg.LDPCinit()
    return pkg
