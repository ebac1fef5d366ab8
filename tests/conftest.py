import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """Builds (if needed) and loads libmsi_hip.so; host-only entry points work without a GPU."""
    from matryodshka_amd import build
    build.build(verbose=False)
    from matryodshka_amd import _native
    return _native
