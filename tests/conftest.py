import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def ctx():
    """HIP context on cuda:0 -- fails loudly when the extension or the GPU is missing."""
    from ov2slam_amd import frontend
    c = frontend.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def stream():
    from ov2slam_amd import synth
    return synth.StereoStream()
