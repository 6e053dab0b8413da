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


def set_mfma_form(ctx, form: int) -> bool:
    """ctx.set_option("mfma_form", form); False when the form is one of the earlier generations of the matrix-core scan
    (1 K1e, 3 K1g, 4 K1h) and the library was built without them (the product: PLSLAM_BUILD_LEGACY_SCANS=1 builds them)."""
    from plslam_amd.capi import ENOTSUP, PlslamError
    try:
        ctx.set_option("mfma_form", form)
        return True
    except PlslamError as e:
        if e.code != ENOTSUP:
            raise
        return False
