"""N > 1 logic on CPU: two gloo ranks shard sequences, run their (stand-in) work, gather the
per-frame records and reduce the timing exactly as bench.py does on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from hso_amd import capi
from hso_amd import dist as hdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = hdist.shard_sequences(7, rank, world)
    # stand-in results: one record per local frame, tagged by (rank, sequence)
    res = []
    for k in range(4):
        r = capi.TrackResult()
        r.T_cur_ref = capi.SE3.from_arrays([0, 0, 0, 1], [rank, k, len(seqs)])
        r.exposure_rat = 1.0 + 0.01 * rank
        res.append(r)
    rec = hdist.pack_records(res)
    allrec = hdist.gather_records(rec)
    tmax = hdist.max_over_ranks(0.5 + rank)
    # the chain's records: per-frame trajectories of this rank's sequences (bench.py "sequences": hso_vo_multi_* per GPU), NaN-padded
    trajs = [[(float(k), ([0, 0, 0, 1], [rank, s, k])) for k in range(3 + s)] for s in range(2)]
    alltr = hdist.gather_records(hdist.pack_trajectories(trajs, 5))
    dist.barrier()
    q.put((rank, seqs, allrec, tmax, alltr))
    dist.destroy_process_group()


def test_two_rank_gather_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, a0, t0, tr0), (r1, s1, a1, t1, tr1) = out
    # trajectories: [world, sequences, frames, 8], every rank holds all; short sequences padded with NaN
    assert tr0.shape == (2, 2, 5, 8) and np.array_equal(tr0, tr1, equal_nan=True)
    assert np.allclose(tr0[1, 1, 3, 4:8], [1, 1, 3, 3]) and np.isnan(tr0[0, 0, 3:, :]).all() and not np.isnan(tr0[0, 1, :4, :]).any()
    assert s0 == [0, 1, 2, 3] and s1 == [4, 5, 6]          # 7 sequences over 2 ranks
    assert a0.shape == (2, 4, 8) and np.array_equal(a0, a1)  # every rank holds every record
    assert np.allclose(a0[1, 2, 4:7], [1, 2, 3]) and np.isclose(a0[1, 0, 7], 1.01, atol=1e-6)
    assert t0 == t1 == 1.5                                  # MAX over ranks


def test_shard_sequences_partition():
    for n in (1, 8, 13):
        for world in (1, 2, 8):
            parts = [hdist.shard_sequences(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_paths():
    rec = np.ones((3, 8))
    assert hdist.gather_records(rec).shape == (1, 3, 8)
    assert hdist.max_over_ranks(2.5) == 2.5
