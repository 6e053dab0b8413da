import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


@pytest.fixture(scope="session")
def ctx():
    """The HIP context.  No skip-if-missing: on a GPU box a missing library or device is a failure."""
    import plslam_amd
    c = plslam_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O
