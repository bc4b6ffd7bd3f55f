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


import contextlib


@contextlib.contextmanager
def track_env(ctx, mode, feats_per_wg=None):
    """Select the tracker shape for the calls of `ctx` inside the block (hso_gpu_configure; the library reads it at prepare time):
    "coop" (default for <= 8 jobs: several workgroups per job), "scatter" (the same with a job's workgroups spread over the
    XCDs, i.e. the placement-independent transport), "one_wg" (the batch shapes: one workgroup per job).  feats_per_wg
    overrides the features per workgroup the host aims for (default 256), i.e. the number of workgroups per job."""
    ctx.configure(track_no_coop=(mode == "one_wg"), track_coop_scatter=(mode == "scatter"), track_coop_feats_per_wg=feats_per_wg or 0)
    try:
        yield
    finally:
        ctx.configure()
