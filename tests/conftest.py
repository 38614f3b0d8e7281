import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rs-image-segmentation_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import ref_np
    return ref_np


@pytest.fixture(scope="session")
def ctx():
    """GPU context; fails loudly if the HIP library or the GPU is missing (no CPU fallback)."""
    from rsseg.runtime import Context
    c = Context(0, use_dist=False)
    yield c
    c.close()
