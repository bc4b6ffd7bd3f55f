"""Error behaviour of the C-ABI: every entry point rejects null / inconsistent arguments with a
negative status and a readable hso_gpu_last_error(), never crashes, and leaves the context usable
(the reference throws std::runtime_error at the corresponding places, e.g. src/frame.cpp:85-86)."""
import ctypes as C

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu

E_INVALID, E_NOFRAME = -1, -2


def _err(ctx):
    return capi.load().hso_gpu_last_error(ctx.h).decode()


def test_null_and_inconsistent_arguments(gpu_ctx, cam, pair200):
    lib = capi.load()
    h = gpu_ctx.h
    img = pair200["ref"]
    # frames
    assert lib.hso_gpu_frame_upload(h, 9600, None, 640, 480, 0, None) == E_INVALID and "null" in _err(gpu_ctx)
    assert lib.hso_gpu_frame_upload(h, 9600, img.ctypes.data_as(C.POINTER(C.c_uint8)), 0, 480, 0, None) == E_INVALID
    assert lib.hso_gpu_frame_release(h, 9600) == E_NOFRAME
    assert lib.hso_gpu_frame_download_level(h, 9600, 0, None, None, None) < 0
    gpu_ctx.frame_upload(9600, img)
    try:
        out = np.zeros((480, 640), np.uint8)
        w, hh = C.c_int(), C.c_int()
        assert lib.hso_gpu_frame_download_level(h, 9600, 7, out.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(hh)) == E_INVALID
        # tracker: null job table, negative count, null results
        p = capi.TrackParams(0, 4, 1, 50)
        assert lib.hso_gpu_coarse_track_batch(h, C.byref(cam), C.byref(p), None, 1, None) == E_INVALID
        job = gpu_ctx.make_job(9600, 9600, pair200["feats"], capi.SE3.identity(), 1.0)
        res = capi.TrackResult()
        assert lib.hso_gpu_coarse_track_batch(h, None, C.byref(p), C.byref(job), 1, C.byref(res)) == E_INVALID
        bad = capi.TrackParams(0, 1, 4, 50)                      # max_level < min_level
        assert lib.hso_gpu_coarse_track_batch(h, C.byref(cam), C.byref(bad), C.byref(job), 1, C.byref(res)) == E_INVALID
        cam2 = synth.camera(synth.EUROC)                         # camera size differs from the frames'
        assert lib.hso_gpu_coarse_track_batch(h, C.byref(cam2), C.byref(p), C.byref(job), 1, C.byref(res)) == E_INVALID
        assert lib.hso_gpu_coarse_track_collect(h, None) == E_INVALID
        # matcher / pose / seeds / BA / FAST: null tables with a positive count
        assert lib.hso_gpu_align_batch(h, C.byref(cam), 9600, None, 3, None) == E_INVALID
        assert lib.hso_gpu_align_multi(h, C.byref(cam), None, None, 3, None) == E_INVALID
        assert lib.hso_gpu_pose_optimize_batch(h, C.byref(cam), None, 2, None, None) == E_INVALID
        assert lib.hso_gpu_seed_observe(h, C.byref(cam), 9600, None, 1.0, 1e-3, None, 4, None) == E_INVALID
        assert lib.hso_gpu_seed_activate(h, C.byref(cam), None, 2, None, None, 6, None, None) == E_INVALID
        assert lib.hso_gpu_ba_linearize(h, None, None, 0, None, 0, None, 0, 1.0, 1.0, *([None] * 8)) == E_INVALID
        br = capi.BaResult()
        assert lib.hso_gpu_ba_optimize(h, None, None, 0, None, 0, None, 0, 1.0, 1.0, 10, None, C.byref(br)) == E_INVALID
        hc, he = C.c_float(), C.c_float()
        assert lib.hso_gpu_ba_huber_deltas(h, None, 0, None, 0, None, None, 0, 480.0, C.byref(hc), C.byref(he)) == E_INVALID
        counts = (C.c_int32 * 3)()
        assert lib.hso_gpu_fast_detect(h, 9600, 9, 20, 8, None, 0, counts) == E_INVALID      # more levels than the pyramid has
        assert lib.hso_gpu_fast_detect(h, 9600, 3, 300, 8, None, 0, counts) == E_INVALID     # barrier outside 0..255
        assert lib.hso_gpu_fast_detect(h, 9600, 3, 20, 8, None, 16, counts) == E_INVALID     # cap > 0 without an output buffer
        ids1 = (C.c_int64 * 1)(9600)
        ec = (C.c_int32 * 3)()
        assert lib.hso_gpu_detect_candidates(h, ids1, 1, 4, 20, None, 0, counts, None, 0, ec) == E_INVALID   # Sobel images exist for 3 levels
        assert lib.hso_gpu_detect_candidates(h, ids1, 1, 3, 20, None, 0, counts, None, 8, ec) == E_INVALID   # cap > 0 without a buffer
        assert lib.hso_gpu_detect_candidates(h, ids1, 1, 3, 20, None, 0, None, None, 0, ec) == E_INVALID
        assert lib.hso_gpu_detect_candidates(h, (C.c_int64 * 1)(123456), 1, 3, 20, None, 0, counts, None, 0, ec) == E_NOFRAME
        # empty batches are fine
        assert lib.hso_gpu_align_batch(h, C.byref(cam), 9600, None, 0, None) == 0
        assert lib.hso_gpu_pose_optimize_batch(h, C.byref(cam), None, 0, None, None) == 0
        # the context still works after all of the above
        r = gpu_ctx.coarse_track_batch(cam, p, [job])[0]
        assert r.status == 0 and r.n_tracked > 0
    finally:
        gpu_ctx.frame_release(9600)
    # a null context never dereferences
    assert lib.hso_gpu_frame_release(None, 1) == E_INVALID
    assert lib.hso_gpu_synchronize(None) == E_INVALID


class _Kf(C.Structure):
    _fields_ = [("frame_id", C.c_int64), ("T_f_w", capi.SE3), ("exposure_time", C.c_double), ("keyframe_id", C.c_int32), ("pad_", C.c_int32)]


class _MapPoint(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("idist", C.c_double), ("host_f", C.c_double * 3), ("host_kf", C.c_int32),
                ("obs_begin", C.c_int32), ("obs_count", C.c_int32), ("pad_", C.c_int32)]


class _Obs(C.Structure):
    _fields_ = [("kf", C.c_int32), ("level", C.c_int32), ("type", C.c_int32), ("pad_", C.c_int32),
                ("px", C.c_double * 2), ("f", C.c_double * 3), ("grad", C.c_double * 2)]


class _Rows(C.Structure):
    _fields_ = [("map", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("pad_", C.c_int32),
                ("point_ids", C.POINTER(C.c_int32)), ("points", C.POINTER(_MapPoint)),
                ("obs_ids", C.POINTER(C.c_int32)), ("obs", C.POINTER(_Obs)), ("obs_point", C.POINTER(C.c_int32))]


def test_sequence_map_arguments(gpu_ctx, pair200):
    """hso_gpu_seqmap_*: what the engine sends every frame, checked before it reaches the kernels (they trust the tables)."""
    lib = capi.load()
    h = gpu_ctx.h
    for fn in ("hso_gpu_seqmap_create", "hso_gpu_seqmap_destroy", "hso_gpu_seqmap_set_keyframes", "hso_gpu_seqmap_patch_multi",
               "hso_gpu_seqmap_size", "hso_gpu_seqmap_set_key_points", "hso_gpu_seqmap_configure", "hso_gpu_seq_chain"):
        getattr(lib, fn).restype = C.c_int
    lib.hso_gpu_seqmap_patch_multi.argtypes = [C.c_void_p, C.POINTER(_Rows), C.c_int]
    lib.hso_gpu_seqmap_set_keyframes.argtypes = [C.c_void_p, C.c_int, C.POINTER(_Kf), C.c_int]
    lib.hso_gpu_seqmap_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    m0, m1 = C.c_int(-1), C.c_int(-1)
    assert lib.hso_gpu_seqmap_create(h, C.byref(m0)) == 0 and lib.hso_gpu_seqmap_create(h, C.byref(m1)) == 0 and m0.value != m1.value
    gpu_ctx.frame_upload(9700, pair200["ref"])
    try:
        assert lib.hso_gpu_seqmap_destroy(h, 12345) == E_INVALID
        kf = _Kf(9701, capi.SE3.identity(), 1.0, 0, 0)
        assert lib.hso_gpu_seqmap_set_keyframes(h, m0.value, C.byref(kf), 1) == E_NOFRAME          # keyframe not resident
        kf.frame_id = 9700
        assert lib.hso_gpu_seqmap_set_keyframes(h, m0.value, C.byref(kf), 1) == 0
        assert lib.hso_gpu_seqmap_set_keyframes(h, m0.value, None, 1) == E_INVALID
        ids = (C.c_int32 * 2)(0, 1)
        pts = (_MapPoint * 2)()
        for p in pts:
            p.host_kf = 0; p.obs_begin = 0; p.obs_count = 0
        obs_ids = (C.c_int32 * 1)(0)
        obs = (_Obs * 1)()
        obs[0].kf = 0; obs[0].pad_ = -1
        good = _Rows(m0.value, 2, 1, 0, ids, pts, obs_ids, obs, None)
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(good), 1) == 0
        nk, np_, no = C.c_int(), C.c_int(), C.c_int()
        assert lib.hso_gpu_seqmap_size(h, m0.value, C.byref(nk), C.byref(np_), C.byref(no)) == 0 and (nk.value, np_.value, no.value) == (1, 2, 1)
        twice = (_Rows * 2)(good, good)
        assert lib.hso_gpu_seqmap_patch_multi(h, twice, 2) == E_INVALID and "twice" in _err(gpu_ctx)
        pts[1].host_kf = 3                                                                         # keyframe row the map does not have
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(good), 1) == E_INVALID
        pts[1].host_kf = 0
        ids[1] = -4
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(good), 1) == E_INVALID and "negative" in _err(gpu_ctx)
        ids[1] = 1
        obs[0].kf = 5
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(good), 1) == E_INVALID
        obs[0].kf = 0
        link = (C.c_int32 * 1)(7)                                                                  # Feature::point beyond the point table
        bad_link = _Rows(m0.value, 2, 1, 0, ids, pts, obs_ids, obs, link)
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(bad_link), 1) == E_INVALID
        other = _Rows(m1.value, 2, 0, 0, ids, pts, None, None, None)                               # the second map has no keyframe table yet
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(other), 1) == E_INVALID
        assert lib.hso_gpu_seqmap_patch_multi(h, C.byref(good), 1) == 0                            # the context is still usable
        assert lib.hso_gpu_seq_chain(h, None, None, None, 1, None, 0, None) == E_INVALID
    finally:
        assert lib.hso_gpu_seqmap_destroy(h, m0.value) == 0 and lib.hso_gpu_seqmap_destroy(h, m1.value) == 0
        gpu_ctx.frame_release(9700)


def test_options_and_debug_read_backs_refuse_bad_arguments(gpu_ctx):
    """hso_gpu_configure (round 6: per-context options instead of environment switches) and the parity read-backs of
    include/hso_gpu_debug.h: unknown sizes, out-of-range values and missing tables are refused, the context stays usable."""
    lib = capi.load()
    h = gpu_ctx.h
    lib.hso_gpu_seqmap_debug_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    assert lib.hso_gpu_configure(None, None) == E_INVALID
    assert lib.hso_gpu_configure(h, None) == E_INVALID and "options" in _err(gpu_ctx)
    o = capi.GpuOptions()
    assert lib.hso_gpu_configure(h, C.byref(o)) == E_INVALID                                 # size 0: not a struct this library knows
    o.size = C.sizeof(capi.GpuOptions) + 64
    assert lib.hso_gpu_configure(h, C.byref(o)) == E_INVALID                                 # a newer, larger struct
    o.size = C.sizeof(capi.GpuOptions); o.wait_mode = 9
    assert lib.hso_gpu_configure(h, C.byref(o)) == E_INVALID and "range" in _err(gpu_ctx)
    o.wait_mode = capi.WAIT_NAP; o.track_coop_feats_per_wg = -1
    assert lib.hso_gpu_configure(h, C.byref(o)) == E_INVALID
    o.track_coop_feats_per_wg = 0
    assert lib.hso_gpu_configure(h, C.byref(o)) == 0                                         # a napping wait on a lone context is legal
    short = capi.GpuOptions(16, capi.WAIT_DEFAULT, 1, 0)                                     # a caller built against a shorter struct: the tail defaults
    assert lib.hso_gpu_configure(h, C.byref(short)) == 0
    gpu_ctx.configure()
    # the dump of a map that does not exist / a table of another size
    buf = np.zeros(16, np.int64)
    assert lib.hso_gpu_seqmap_debug_dump(h, 12345, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == E_INVALID
    m = C.c_int(-1)
    assert lib.hso_gpu_seqmap_create(h, C.byref(m)) == 0
    try:
        assert lib.hso_gpu_seqmap_debug_dump(h, m.value, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0 and buf[0] == 0 and buf[1] == 0
        assert lib.hso_gpu_seqmap_debug_dump(h, m.value, 0, buf.ctypes.data_as(C.c_void_p), 8) == E_INVALID and "size" in _err(gpu_ctx)
        assert lib.hso_gpu_seqmap_debug_dump(h, m.value, 99, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == E_INVALID
        assert lib.hso_gpu_seqmap_debug_dump(h, m.value, 2, buf.ctypes.data_as(C.c_void_p), 0) == 0      # an empty point table: nothing to copy
    finally:
        assert lib.hso_gpu_seqmap_destroy(h, m.value) == 0
    # a keyframe table beyond HSO_SEQ_MAX_KFS rows is refused by the chain, not silently truncated: checked on the header's constant
    assert "HSO_SEQ_MAX_KFS 2048" in open(__import__("os").path.join(__import__("os").path.dirname(capi.__file__), "..", "include", "hso_gpu.h")).read()


def test_resident_local_ba_and_cpulist_refuse_bad_arguments(gpu_ctx):
    """hso_gpu_seq_local_ba, its window read-back and hso_gpu_device_cpulist: null / out-of-range arguments -> negative status with a
    message, nothing touched, the context stays usable (the parity half is tests/test_seq_ba.py)."""
    lib = capi.load()
    h = gpu_ctx.h
    lib.hso_gpu_seq_local_ba.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.hso_gpu_seq_ba_debug_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.hso_gpu_device_cpulist.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(64)
    assert lib.hso_gpu_device_cpulist(None, buf, 64) == E_INVALID
    assert lib.hso_gpu_device_cpulist(h, None, 64) == E_INVALID and lib.hso_gpu_device_cpulist(h, buf, 0) == E_INVALID
    assert lib.hso_gpu_device_cpulist(h, buf, 64) == 0                                           # a list, possibly cut, or an empty string
    assert all(c in b"0123456789,-" for c in buf.value)
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import chain_state as cs
    job = np.zeros(1, cs.SEQ_BA_JOB); res = np.zeros(1, cs.SEQ_BA_RESULT)
    pj, pr = job.ctypes.data_as(C.c_void_p), res.ctypes.data_as(C.c_void_p)
    assert lib.hso_gpu_seq_local_ba(None, pj, 1, 1.0, 1.0, 1.0, pr) == E_INVALID
    assert lib.hso_gpu_seq_local_ba(h, None, 1, 1.0, 1.0, 1.0, pr) == E_INVALID and lib.hso_gpu_seq_local_ba(h, pj, 1, 1.0, 1.0, 1.0, None) == E_INVALID
    assert lib.hso_gpu_seq_local_ba(h, pj, -1, 1.0, 1.0, 1.0, pr) == E_INVALID
    assert lib.hso_gpu_seq_local_ba(h, pj, 0, 1.0, 1.0, 1.0, pr) == 0                            # no jobs: nothing to do
    job["map"] = 0; job["n_core"] = 0
    assert lib.hso_gpu_seq_local_ba(h, pj, 1, 1.0, 1.0, 1.0, pr) == E_INVALID                    # a window needs a core keyframe
    job["n_core"] = 17
    assert lib.hso_gpu_seq_local_ba(h, pj, 1, 1.0, 1.0, 1.0, pr) == E_INVALID                    # more than HSO_SEQ_BA_MAX_CORE
    job["n_core"] = 1; job["map"] = 4242
    assert lib.hso_gpu_seq_local_ba(h, pj, 1, 1.0, 1.0, 1.0, pr) == E_INVALID and "map" in _err(gpu_ctx)
    m = C.c_int(-1)
    assert lib.hso_gpu_seqmap_create(h, C.byref(m)) == 0
    try:
        job["map"] = m.value
        assert lib.hso_gpu_seq_local_ba(h, pj, 1, 1.0, 1.0, 1.0, pr) == E_INVALID and "keyframes" in _err(gpu_ctx)   # an empty map
        two = np.zeros(2, cs.SEQ_BA_JOB); two["map"] = m.value; two["n_core"] = 1
        res2 = np.zeros(2, cs.SEQ_BA_RESULT)
        assert lib.hso_gpu_seq_local_ba(h, two.ctypes.data_as(C.c_void_p), 2, 1.0, 1.0, 1.0, res2.ctypes.data_as(C.c_void_p)) == E_INVALID and "twice" in _err(gpu_ctx)
    finally:
        assert lib.hso_gpu_seqmap_destroy(h, m.value) == 0
    four = np.zeros(4, np.int32)
    assert lib.hso_gpu_seq_ba_debug_window(None, 0, 0, four.ctypes.data_as(C.c_void_p), 16) == E_INVALID
    assert lib.hso_gpu_seq_ba_debug_window(h, 7, 0, four.ctypes.data_as(C.c_void_p), 16) == E_INVALID   # no such job in the last call
    assert lib.hso_gpu_synchronize(h) == 0
