import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """One HIP context on device 0 for the whole GPU test session (fails loudly without a GPU)."""
    from dot_ring_amd import _native

    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def srs_bytes():
    path = os.path.join(ROOT, "dot_ring_amd", "data", "bls12-381-srs-2-11-uncompressed-zcash.bin")
    with open(path, "rb") as f:
        blob = f.read()
    count = int.from_bytes(blob[:8], "little")
    return blob[8 : 8 + 96 * count]
