"""The native result gather (include/hso_vo.h: hso_gather_*, libhso_gather.so): ncclAllGather of per-frame records.
CPU: the library loads and exports what the header declares, argument errors come back as codes.  GPU: a one-rank
communicator on cuda:0 returns the records it was given (the N > 1 exchange is the same call on every rank; an 8-GPU node is
the driver's to run), and agrees with hso_amd.dist.gather_records."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_library_exports_its_c_interface():
    from hso_amd import dist
    src = open(os.path.join(ROOT, "include", "hso_vo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(hso_gather_\w+)\s*\(", src)))
    lib = dist.load_gather()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(dist.GATHER_SYMBOLS) == names
    # argument errors, no device needed
    h = C.c_void_p()
    assert lib.hso_gather_create(C.byref(h), None, 0, 1, 0) == -1 and not h.value
    uid = (C.c_uint8 * 128)()
    assert lib.hso_gather_create(C.byref(h), uid, 2, 2, 0) == -1 and not h.value      # rank outside the world
    assert lib.hso_gather_records(None, None, 1, None) == -1
    assert b"bad arguments" in lib.hso_gather_last_error()
    assert lib.hso_gather_size(None) == 0 and lib.hso_gather_rank(None) == -1
    lib.hso_gather_destroy(None)
    with pytest.raises(ValueError):
        dist.NativeGather(b"short", 0, 1)


@pytest.mark.gpu
def test_one_rank_gather_returns_the_records():
    from hso_amd import dist
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")   # one rank, one node: the bootstrap needs no other interface
    rng = np.random.default_rng(5)
    g = dist.NativeGather(dist.NativeGather.unique_id(), 0, 1, 0)
    try:
        traj = dist.pack_trajectories([[(0.1 * k, (rng.normal(size=4), rng.normal(size=3))) for k in range(n)] for n in (7, 5, 0)], 7)
        out = g.gather(traj)                       # [1, 3, 7, 8], NaN rows of the shorter sequences travel as they are
        assert out.shape == (1, 3, 7, 8)
        assert np.array_equal(out[0], traj, equal_nan=True)
        assert np.array_equal(dist.gather_records(traj)[0], out[0], equal_nan=True)
        big = rng.normal(size=(4096, 8))           # grows the staging buffers
        assert np.array_equal(g.gather(big)[0], big)
        assert g.gather(np.zeros((0, 8))).shape == (1, 0, 8)
        with pytest.raises(ValueError):
            g.gather(np.zeros((3, 7)))
    finally:
        g.close()
