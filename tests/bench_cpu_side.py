"""TEST INFRASTRUCTURE: a stand-in for bench.py's GpuSide over the CPU restatement (oracle/) and gloo, selected with
HSO_BENCH_SIDE=bench_cpu_side:CpuSide.  It exists so that bench.py's multi-rank control flow — self-launch, the shard of distinct
scenes, the per-frame record gather, the trajectory gather of the sequence engine, max-over-ranks timing, the JSON line — runs
under `torch.distributed.run` without a GPU (tests/test_bench_cpu.py).  Numbers it produces mean nothing."""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        pass


class _On:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Event:
    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * max(other.t - self.t, 1e-6)


class _Tensor:
    def __init__(self, img):
        self.a = np.ascontiguousarray(img, np.uint8)

    def data_ptr(self):
        return self.a.ctypes.data


class _OracleContext:
    """the handful of capi.Context methods bench.py's timed loop uses, on the restatement"""

    def __init__(self):
        from hso_amd import capi
        from oracle import oracle_py
        self.capi, self.orc = capi, oracle_py
        oracle_py.build(); oracle_py.load()
        self.frames = {}
        self.make_job = capi.Context.make_job

    def frame_upload_batch(self, frame_ids, imgs=None, device_ptrs=None, width=None, height=None, want_stats=True):
        out = []
        for i, fid in enumerate(frame_ids):
            img = imgs[i] if imgs is not None else np.frombuffer(C.string_at(int(device_ptrs[i]), width * height), np.uint8).reshape(height, width)
            pyr = self.orc.create_pyramid(np.ascontiguousarray(img))
            self.frames[int(fid)] = pyr
            if want_stats:
                out.append(self.orc.frame_stats(pyr[0], *self.orc.sobel5(np.ascontiguousarray(pyr[0]))))
        return out if want_stats else None

    def coarse_track_prepare(self, cam, params, jobs):
        self.cam, self.params, self.jobs = cam, params, list(jobs)

    def coarse_track_launch(self):
        pass

    def coarse_track_collect(self, as_list=True):
        res = []
        for j in self.jobs:
            feats = np.frombuffer(C.string_at(j.feats, j.n_feats * self.capi.REF_FEAT_DTYPE.itemsize), self.capi.REF_FEAT_DTYPE)
            tr = self.orc.Tracker(self.cam, self.params, self.frames[j.ref_frame_id], self.frames[j.cur_frame_id], feats)
            res.append(tr.run(j.T_cur_ref, j.exposure_rat))
        return res

    def close(self):
        self.frames.clear()


class CpuSide:
    backend = "gloo"
    engine_lib = os.path.join(HERE, "fakegpu", "libhso_host_fake.so")

    def __init__(self, local_rank):
        import torch
        self.torch, self.rank = torch, local_rank
        self.dev = torch.device("cpu")

    def init_group(self, rank, world):
        import torch.distributed as dist
        dist.init_process_group(self.backend, rank=rank, world_size=world)

    def new_stream(self):
        return _Stream()

    def on(self, stream):
        return _On()

    def context(self, stream):
        return _OracleContext()

    def to_device(self, img):
        return _Tensor(img)

    def event(self):
        return _Event()

    def synchronize(self):
        pass

    def release_cached(self):
        pass
