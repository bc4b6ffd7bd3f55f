"""Host memory handed to the entry points (hso_amd/csrc/hso_ctx.h: hso_copy_async / hso_stream_sync).

Why the wrappers exist: on this stack, host memory that was the source or destination of a runtime copy as PAGEABLE memory and
is then returned to the kernel (munmap of a large block, trimming of a thread's malloc arena) makes the kernel driver evict the
process's GPU queues; the next launch starts 10-35 ms late.  Found in the multi-sequence driver at 32 sequences
(profiles/r3_host_memory_eviction.txt); the library now stages pageable memory itself, so nothing the caller does with its buffers
afterwards can have that effect.  The first test drives the caller-side pattern — multi-megabyte tables in and out through the
direct-copy path of a value-passing call, freed at once, then a small call — and requires the small call to keep its normal
latency.  It is a guard, not a reproduction: from a single-threaded caller glibc keeps the freed memory after the first large
free; the reproduction is the multi-sequence run recorded in the profile note (made with a build that handed pageable memory to the
runtime as rounds 1-2 did)."""
import gc
import time

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_freeing_large_pageable_tables_does_not_stall_later_calls(gpu_ctx):
    spec = synth.ICL_NUIM
    cam = synth.camera(spec)
    P = synth.map_problem(n_points=600, spec=spec, first_frame_id=9700)
    ids = [int(k["frame_id"]) for k in P["kfs"]]
    for i, f in zip(ids, P["frames"]):
        gpu_ctx.frame_upload(i, f)
    gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"])
    try:
        f, p, T0, _ = synth.pose_problem(200, seed=5)
        job = capi.make_pose_job(f, p, T0)
        # the point table repeated to ~2.9 MB (+ 3.8 MB of results): hso_gpu_reproject_match's large-table path copies the caller's
        # tables and results directly; every array is an mmap'd region, freed right after the call
        reps = 64
        small = []
        for rep in range(12):
            pts = np.tile(P["points"], reps)
            proj, match = gpu_ctx.reproject_match(cam, P["cur_frame_id"], P["T_cur_w"], P["cur_exposure"], P["cur_keyframe_id"], P["kfs"], pts,
                                                  P["obs"], P["cell_size"], P["grid_n_cols"])
            assert len(proj) == len(pts)
            del pts, proj, match
            gc.collect()
            t0 = time.perf_counter()
            res, _ = gpu_ctx.pose_optimize_batch(cam, [job])
            small.append((time.perf_counter() - t0) * 1e3)
            assert res[0].status == 0
        print("small call after freed multi-megabyte tables: ms", ["%.2f" % t for t in small])
        # normal: 0.15-0.3 ms; evicted queues: 10-35 ms.  The pattern that stalled the multi-sequence driver did so in most steps; a
        # single late call can still happen for reasons outside the library (seen once in ~60 calls inside the full test-suite
        # process, whose earlier tests leave threads and mappings behind), so the bound is on the median and on the number of stalls
        stalls = sum(t > 8.0 for t in small[1:])
        assert np.median(small) < 3.0 and stalls <= 2, small
    finally:
        for i in ids + [P["cur_frame_id"]]:
            gpu_ctx.frame_release(i)


def test_results_arrive_in_pageable_and_page_locked_tables_alike(gpu_ctx, cam, pair2000):
    """The same call into a numpy array (staged: copied out by the call's own synchronisation) and into a table of
    hso_gpu_host_alloc (DMA target as it is) returns the same bytes."""
    d = pair2000
    gpu_ctx.frame_upload(9711, d["ref"]); gpu_ctx.frame_upload(9712, d["cur"])
    try:
        seeds, T_cur, _ = synth.seeds_for_pair(d, 500, 9711, seed=4)
        t = gpu_ctx.seed_table_create()
        gpu_ctx.seed_table_append(t, seeds)
        pea = 2 * np.arctan(1.0 / (2.0 * 480.6))
        a, _ = gpu_ctx.seed_table_observe(cam, t, [(9712, T_cur, 1.05)], pea)
        gpu_ctx.seed_table_destroy(t)
        t = gpu_ctx.seed_table_create()
        gpu_ctx.seed_table_append(t, seeds)
        locked = gpu_ctx.host_array(500, capi.SEED_BRIEF_DTYPE)
        b, _ = gpu_ctx.seed_table_observe(cam, t, [(9712, T_cur, 1.05)], pea, brief_out=locked)
        gpu_ctx.seed_table_destroy(t)
        assert a.tobytes() == b.tobytes() and np.shares_memory(b, locked)
    finally:
        gpu_ctx.frame_release(9711); gpu_ctx.frame_release(9712)
