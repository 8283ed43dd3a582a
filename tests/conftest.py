import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_png_rgb(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))


@pytest.fixture(scope="session")
def marlene():
    return load_png_rgb("marlene.png")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The shared library must exist; build it if the tree is fresh."""
    from mathmap_amd._lib import LIB_PATH
    if not os.path.exists(LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
