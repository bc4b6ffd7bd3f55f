import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU restatement (oracle/) — the checker every parity test compares against."""
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def pair2000():
    from hso_amd import synth
    return synth.config2_pair(2000)


@pytest.fixture(scope="session")
def pair200():
    from hso_amd import synth
    return synth.config2_pair(200, seed=77)


@pytest.fixture(scope="session")
def cam():
    from hso_amd import synth
    return synth.camera()


@pytest.fixture(scope="session")
def gpu_ctx():
    from hso_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
