import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gs-lora_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def dev_lib():
    """`dev_lib(L)` routes the rest of the test through libgslora_hip_dev.so (the -DGSL_DEV build with the lab kernels and the GSL_* knobs);
    the product library is restored at teardown."""
    import contextlib
    with contextlib.ExitStack() as stack:
        yield lambda L: stack.enter_context(L.use_dev())
