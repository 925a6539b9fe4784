import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "ntsc-crt_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """Build the CPU checkers once per session: the oracle always, oracle/_ref only where
    /root/reference exists (this container).  Building the checker is not using it."""
    import crtref
    crtref.build_oracle()
    if os.path.exists("/root/reference/crt_core.c"):
        crtref.build_ref()
    yield
