"""ctypes binding of the sequence engine (include/hso_vo.h, hso_amd/host/libhso_host.so): FrameHandlerMono::addImage's
pipeline for one or many sequences over the device library, plus a reader for the engine's C-ABI call trace
(hso_amd/host/hso_engine.h: Trace).  Plumbing for the harness and the tests."""
import ctypes as C
import os
import struct

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libhso_host.so")


class VoStatus(C.Structure):
    _fields_ = [("T_f_w", capi.SE3), ("timestamp", C.c_double), ("exposure_time", C.c_double),
                ("frame_id", C.c_int32), ("keyframe_id", C.c_int32), ("is_keyframe", C.c_int32), ("stage", C.c_int32),
                ("tracking_quality", C.c_int32), ("result", C.c_int32),
                ("n_features", C.c_int32), ("n_inliers", C.c_int32), ("n_tracked", C.c_int32), ("n_matches", C.c_int32),
                ("n_trials", C.c_int32), ("n_seed_matches", C.c_int32), ("n_seeds", C.c_int32), ("n_candidates", C.c_int32),
                ("n_keyframes", C.c_int32), ("used_inverse", C.c_int32), ("ba_removed_1", C.c_int32), ("ba_removed_2", C.c_int32),
                ("pose_error_init", C.c_double), ("pose_error_final", C.c_double), ("ba_error_init", C.c_double),
                ("ba_error_final", C.c_double)]


class VoOptions(C.Structure):
    # hso_vo_options (include/hso_vo.h)
    _fields_ = [("size", C.c_int32), ("sync_previous", C.c_int32), ("track_no_coop", C.c_int32), ("no_numa_pin", C.c_int32), ("reserved", C.c_int32 * 4)]


_lib = None


def _declare(lib):
    vp, i32, P = C.c_void_p, C.c_int, C.POINTER
    lib.hso_vo_create.argtypes = [P(vp), P(capi.Camera), i32, i32]
    lib.hso_vo_destroy.argtypes = [vp]
    lib.hso_vo_destroy.restype = None
    lib.hso_vo_last_error.argtypes = [vp]
    lib.hso_vo_last_error.restype = C.c_char_p
    lib.hso_vo_trace.argtypes = [vp, C.c_char_p]
    lib.hso_vo_trace_state.argtypes = [vp, i32]
    lib.hso_vo_set_options.argtypes = [vp, P(VoOptions)]
    lib.hso_vo_multi_set_options.argtypes = [vp, P(VoOptions)]
    lib.hso_vo_multi_trace_state.argtypes = [vp, i32, i32]
    lib.hso_vo_set_first_frame.argtypes = [vp, vp, i32, i32, C.c_double, vp, P(capi.SE3)]
    lib.hso_vo_add_image.argtypes = [vp, vp, i32, i32, C.c_double]
    lib.hso_vo_start.argtypes = [vp]
    lib.hso_vo_init_compute_matrix.argtypes = [vp, vp, i32, C.c_double, C.c_double, P(capi.SE3), vp, i32, vp, vp]
    lib.hso_vo_get_status.argtypes = [vp, P(VoStatus)]
    lib.hso_vo_get_keyframes.argtypes = [vp, vp, vp, vp, i32]
    lib.hso_vo_multi_create.argtypes = [P(vp), P(capi.Camera), i32, i32, i32]
    lib.hso_vo_multi_destroy.argtypes = [vp]
    lib.hso_vo_multi_destroy.restype = None
    lib.hso_vo_multi_last_error.argtypes = [vp]
    lib.hso_vo_multi_last_error.restype = C.c_char_p
    lib.hso_vo_multi_size.argtypes = [vp]
    lib.hso_vo_multi_set_first_frames.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.hso_vo_multi_add_images.argtypes = [vp, vp, i32, i32, vp]
    lib.hso_vo_multi_start.argtypes = [vp, vp]
    if hasattr(lib, "hso_vo_multi_add_images_device"):
        lib.hso_vo_multi_add_images_device.argtypes = [vp, vp, i32, i32, vp]
    lib.hso_vo_multi_trace.argtypes = [vp, i32, C.c_char_p]
    lib.hso_vo_multi_get_status.argtypes = [vp, i32, P(VoStatus)]
    lib.hso_vo_multi_get_keyframes.argtypes = [vp, i32, vp, vp, vp, i32]
    lib.hso_vo_multi_call_counts.argtypes = [vp, vp, vp, i32]
    lib.hso_vo_host_share.argtypes = [i32]
    lib.hso_vo_multi_alg_bytes.argtypes = [vp, vp, i32]
    lib.hso_vo_multi_threads.argtypes = [vp]
    lib.hso_vo_host_cpu_quota.argtypes = []
    lib.hso_vo_multi_get_trajectory.argtypes = [vp, i32, vp, vp, i32]
    lib.hso_vo_get_trajectory.argtypes = [vp, vp, vp, i32]
    return lib


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()                      # the device library (and torch's HIP runtime) first
    if not os.path.exists(LIB_PATH):
        raise capi.HsoGpuError("libhso_host.so is not built (%s): run `python -m hso_amd.build`" % LIB_PATH)
    _lib = _declare(C.CDLL(LIB_PATH))
    return _lib


def load_from(path):
    """The same interface from another build of the engine (tests: the engine over the CPU restatement, tests/fakegpu)."""
    return _declare(C.CDLL(path))


EXPORTED_SYMBOLS = ["hso_vo_create", "hso_vo_destroy", "hso_vo_last_error", "hso_vo_trace", "hso_vo_set_first_frame",
                    "hso_vo_add_image", "hso_vo_get_status", "hso_vo_get_keyframes", "hso_vo_start", "hso_vo_init_compute_matrix",
                    "hso_vo_multi_create", "hso_vo_multi_destroy", "hso_vo_multi_last_error", "hso_vo_multi_size",
                    "hso_vo_multi_set_first_frames", "hso_vo_multi_add_images", "hso_vo_multi_get_status", "hso_vo_multi_get_keyframes",
                    "hso_vo_multi_call_counts", "hso_vo_host_share", "hso_vo_multi_alg_bytes", "hso_vo_multi_threads", "hso_vo_host_cpu_quota", "hso_vo_multi_start", "hso_vo_multi_trace", "hso_vo_multi_add_images_device", "hso_vo_multi_get_trajectory", "hso_vo_get_trajectory", "hso_vo_trace_state", "hso_vo_multi_trace_state", "hso_vo_set_options", "hso_vo_multi_set_options"]

CALL_KINDS = ["frame_upload", "frame_release", "track", "reproject_select_pose", "align", "pose", "seed_observe", "seed_activate", "ba", "other"]


class MultiVisualOdometry:
    """N FrameHandlerMono over one device context, advancing in lockstep (include/hso_vo.h: hso_vo_multi_*)."""

    def __init__(self, cam, n_sequences, max_fts=200, device=0, lib=None):
        self.lib = lib or load()
        self.h = C.c_void_p()
        rc = self.lib.hso_vo_multi_create(C.byref(self.h), C.byref(cam), int(max_fts), int(n_sequences), int(device))
        if rc < 0:
            raise capi.HsoGpuError("hso_vo_multi_create failed: %d" % rc)
        self.n = n_sequences

    def close(self):
        if self.h:
            self.lib.hso_vo_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise capi.HsoGpuError("%s failed (%d): %s" % (what, rc, (self.lib.hso_vo_multi_last_error(self.h) or b"?").decode()))

    @staticmethod
    def _ptrs(arrs):
        return (C.c_void_p * len(arrs))(*[a.ctypes.data if a is not None else None for a in arrs])

    def set_first_frames(self, imgs, depths, timestamps=None):
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        depths = [np.ascontiguousarray(d, np.float32) for d in depths]
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.zeros(self.n), np.float64)
        h, w = imgs[0].shape
        self._check(self.lib.hso_vo_multi_set_first_frames(self.h, self._ptrs(imgs), w, h, ts.ctypes.data, self._ptrs(depths), None), "set_first_frames")

    def trace(self, k, path):
        self._check(self.lib.hso_vo_multi_trace(self.h, int(k), path.encode() if path else None), "trace")

    def set_options(self, sync_previous=False, track_no_coop=False, no_numa_pin=False):
        o = VoOptions(C.sizeof(VoOptions), int(bool(sync_previous)), int(bool(track_no_coop)), int(bool(no_numa_pin)))
        self._check(self.lib.hso_vo_multi_set_options(self.h, C.byref(o)), "set_options")

    def start(self, which=None):
        w = np.ascontiguousarray(which, np.uint8) if which is not None else None
        self._check(self.lib.hso_vo_multi_start(self.h, w.ctypes.data if w is not None else None), "start")

    def add_images(self, imgs, timestamps):
        """imgs: one image per sequence, None = the sequence sits this step out."""
        imgs = [np.ascontiguousarray(i, np.uint8) if i is not None else None for i in imgs]
        ts = np.ascontiguousarray(timestamps, np.float64)
        shape = next(i.shape for i in imgs if i is not None)
        self._check(self.lib.hso_vo_multi_add_images(self.h, self._ptrs(imgs), shape[1], shape[0], ts.ctypes.data), "add_images")

    def add_images_host_ptrs(self, ptrs, width, height, timestamps):
        """ptrs: one HOST pointer (int) per sequence (e.g. page-locked buffers a camera driver fills), 0 / None = sits out"""
        arr = (C.c_void_p * len(ptrs))(*[p if p else None for p in ptrs])
        ts = np.ascontiguousarray(timestamps, np.float64)
        self._check(self.lib.hso_vo_multi_add_images(self.h, arr, width, height, ts.ctypes.data), "add_images")

    def add_images_device(self, ptrs, width, height, timestamps):
        """ptrs: one device pointer (int) per sequence, 0 / None = the sequence sits this step out."""
        arr = (C.c_void_p * len(ptrs))(*[p if p else None for p in ptrs])
        ts = np.ascontiguousarray(timestamps, np.float64)
        self._check(self.lib.hso_vo_multi_add_images_device(self.h, arr, width, height, ts.ctypes.data), "add_images_device")

    def status(self, k):
        st = VoStatus()
        self._check(self.lib.hso_vo_multi_get_status(self.h, k, C.byref(st)), "get_status")
        return st

    def keyframes(self, k):
        n = self.lib.hso_vo_multi_get_keyframes(self.h, k, None, None, None, 0)
        ts = np.zeros(max(n, 1)); T = (capi.SE3 * max(n, 1))(); ids = np.zeros(max(n, 1), np.int32)
        self.lib.hso_vo_multi_get_keyframes(self.h, k, capi._ptr(ts), C.cast(T, C.c_void_p), capi._ptr(ids), n)
        return [(float(ts[i]), T[i], int(ids[i])) for i in range(n)]

    def trajectory(self, k):
        """-> (timestamps [n], poses [n, 7] as qx qy qz qw tx ty tz) of every frame sequence k has processed"""
        n = self.lib.hso_vo_multi_get_trajectory(self.h, k, None, None, 0)
        ts = np.zeros(max(n, 1)); T = np.zeros((max(n, 1), 7))
        self.lib.hso_vo_multi_get_trajectory(self.h, k, ts.ctypes.data, T.ctypes.data, n)
        return ts[:n], T[:n]

    def alg_bytes(self):
        out = np.zeros(5)
        self.lib.hso_vo_multi_alg_bytes(self.h, out.ctypes.data, 5)
        return dict(zip(("frame", "track", "match", "pose", "seed"), [float(x) for x in out]))

    def call_counts(self):
        calls = np.zeros(16, np.int64); items = np.zeros(16, np.int64)
        n = self.lib.hso_vo_multi_call_counts(self.h, calls.ctypes.data, items.ctypes.data, 16)
        return {CALL_KINDS[k]: (int(calls[k]), int(items[k])) for k in range(n)}


class VisualOdometry:
    """FrameHandlerMono behind the C interface."""

    def __init__(self, cam, max_fts=200, device=0, lib=None):
        self.lib = lib or load()
        self.h = C.c_void_p()
        rc = self.lib.hso_vo_create(C.byref(self.h), C.byref(cam), int(max_fts), int(device))
        if rc < 0:
            raise capi.HsoGpuError("hso_vo_create failed: %d" % rc)
        self.cam = cam

    def close(self):
        if self.h:
            self.lib.hso_vo_trace(self.h, None)
            self.lib.hso_vo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise capi.HsoGpuError("%s failed (%d): %s" % (what, rc, (self.lib.hso_vo_last_error(self.h) or b"?").decode()))

    def trace(self, path, state=False):
        """record the device calls to `path` (None stops); state=True: every chain call with the sequence map it ran on (hso_vo_trace_state)"""
        self._check(self.lib.hso_vo_trace_state(self.h, 1 if state else 0), "trace_state")
        self._check(self.lib.hso_vo_trace(self.h, path.encode() if path else None), "trace")

    def set_options(self, sync_previous=False, track_no_coop=False, no_numa_pin=False):
        o = VoOptions(C.sizeof(VoOptions), int(bool(sync_previous)), int(bool(track_no_coop)), int(bool(no_numa_pin)))
        self._check(self.lib.hso_vo_set_options(self.h, C.byref(o)), "set_options")

    def set_first_frame(self, img, depth_z, timestamp=0.0, T_f_w=None):
        img = np.ascontiguousarray(img, np.uint8)
        depth_z = np.ascontiguousarray(depth_z, np.float32)
        assert depth_z.shape == img.shape
        self._check(self.lib.hso_vo_set_first_frame(self.h, capi._ptr(img), img.shape[1], img.shape[0], float(timestamp),
                                                    capi._ptr(depth_z), C.byref(T_f_w) if T_f_w is not None else None),
                    "set_first_frame")

    def start(self):
        """FrameHandlerBase::start(): the next images run the two-view initialisation."""
        self._check(self.lib.hso_vo_start(self.h), "start")

    def add_image(self, img, timestamp):
        img = np.ascontiguousarray(img, np.uint8)
        self._check(self.lib.hso_vo_add_image(self.h, capi._ptr(img), img.shape[1], img.shape[0], float(timestamp)), "add_image")
        return self.status()

    def status(self):
        st = VoStatus()
        self._check(self.lib.hso_vo_get_status(self.h, C.byref(st)), "get_status")
        return st

    def keyframes(self):
        n = self.lib.hso_vo_get_keyframes(self.h, None, None, None, 0)
        ts = np.zeros(max(n, 1)); T = (capi.SE3 * max(n, 1))(); ids = np.zeros(max(n, 1), np.int32)
        self.lib.hso_vo_get_keyframes(self.h, capi._ptr(ts), C.cast(T, C.c_void_p), capi._ptr(ids), n)
        return [(float(ts[i]), T[i], int(ids[i])) for i in range(n)]


def init_compute_matrix(f_ref, f_cur, focal_length, reproj_thresh=2.0):
    """initialization::computeInitializeMatrix on unit bearings (n, 3) -> (T_cur_from_ref SE3, inlier indices, xyz_in_cur (n, 3),
    used_homography).  Host code only."""
    lib = load()
    a = np.ascontiguousarray(f_ref, np.float64); b = np.ascontiguousarray(f_cur, np.float64)
    n = len(a)
    T = capi.SE3(); inl = np.zeros(max(n, 1), np.int32); xyz = np.zeros((max(n, 1), 3)); used = np.zeros(1, np.int32)
    k = lib.hso_vo_init_compute_matrix(capi._ptr(a), capi._ptr(b), n, float(focal_length), float(reproj_thresh), C.byref(T), capi._ptr(inl), n,
                                    capi._ptr(xyz), capi._ptr(used))
    if k < 0:
        raise capi.HsoGpuError("hso_vo_init_compute_matrix failed: %d" % k)
    return T, inl[:k].copy(), xyz[:n], int(used[0])


def read_trace(path):
    """-> list of (call name, {field: bytes}) in call order; scalars are 8-byte doubles (scalar())."""
    data = open(path, "rb").read()
    pos, out = 0, []
    while pos + 12 <= len(data):
        magic, nl = struct.unpack_from("<II", data, pos); pos += 8
        if magic != 0x52545348:
            raise ValueError("bad trace record at %d" % (pos - 8))
        name = data[pos:pos + nl].decode(); pos += nl
        (nf,) = struct.unpack_from("<I", data, pos); pos += 4
        rec = {}
        for _ in range(nf):
            (kl,) = struct.unpack_from("<I", data, pos); pos += 4
            key = data[pos:pos + kl].decode(); pos += kl
            (nb,) = struct.unpack_from("<Q", data, pos); pos += 8
            rec[key] = data[pos:pos + nb]; pos += nb
        out.append((name, rec))
    return out


def scalar(rec, key):
    return struct.unpack("<d", rec[key])[0]
