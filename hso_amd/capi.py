"""ctypes binding of the C-ABI in include/hso_gpu.h (libhso_gpu.so).

This is plumbing for tests and bench.py: the product is the shared library.
Loading fails loudly when the HIP extension has not been built — there is no
CPU fallback anywhere in this package.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HSO_GPU_LIB") or os.path.join(_HERE, "csrc", "libhso_gpu.so")   # HSO_GPU_LIB: developer experiments only

ABI_VERSION = 2      # include/hso_gpu.h: HSO_GPU_ABI_VERSION
N_PYR_LEVELS = 5
N_SOBEL_LEVELS = 3
CAM_PINHOLE, CAM_FOV, CAM_EQUIDISTANT = 0, 1, 2


class Camera(C.Structure):
    _fields_ = [("model", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("distortion", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("d", C.c_double * 5)]


class SE3(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]

    @staticmethod
    def from_arrays(q, t):
        s = SE3()
        s.q[:] = [float(x) for x in q]
        s.t[:] = [float(x) for x in t]
        return s

    @staticmethod
    def identity():
        return SE3.from_arrays([0, 0, 0, 1], [0, 0, 0])

    def to_arrays(self):
        return np.array(self.q[:]), np.array(self.t[:])


class KltParams(C.Structure):
    # hso_klt_params: the constants of initialization::trackKlt (src/initialization.cpp:236-245)
    _fields_ = [("win_size", C.c_int32), ("max_level", C.c_int32), ("max_iter", C.c_int32), ("use_initial_flow", C.c_int32),
                ("epsilon", C.c_double)]

    def __init__(self, win_size=30, max_level=4, max_iter=30, use_initial_flow=1, epsilon=1e-4):
        super().__init__(win_size, max_level, max_iter, use_initial_flow, epsilon)


KLT_RESULT_DTYPE = np.dtype([("px", np.float32, 2), ("ncc", np.float32), ("status", np.int32)])
KLT_TRACKED, KLT_PATCH_OK = 1, 2


class FrameStats(C.Structure):
    _fields_ = [("integral_image", C.c_float), ("grad_mean", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class RefFeat(C.Structure):
    _fields_ = [("px", C.c_double * 2), ("f", C.c_double * 3), ("dist", C.c_double)]


REF_FEAT_DTYPE = np.dtype([("px", "<f8", 2), ("f", "<f8", 3), ("dist", "<f8")])
assert REF_FEAT_DTYPE.itemsize == C.sizeof(RefFeat)


class DepthRefIn(C.Structure):
    _fields_ = [("has_point", C.c_int32), ("host_pose", C.c_int32),
                ("host_f", C.c_double * 3), ("idist", C.c_double)]


DEPTH_REF_IN_DTYPE = np.dtype([("has_point", "<i4"), ("host_pose", "<i4"),
                               ("host_f", "<f8", 3), ("idist", "<f8")])
assert DEPTH_REF_IN_DTYPE.itemsize == C.sizeof(DepthRefIn)


class TrackParams(C.Structure):
    _fields_ = [("inverse_composition", C.c_int32), ("max_level", C.c_int32),
                ("min_level", C.c_int32), ("n_iter", C.c_int32)]


class TrackJob(C.Structure):
    _fields_ = [("ref_frame_id", C.c_int64), ("cur_frame_id", C.c_int64),
                ("feats", C.c_void_p), ("n_feats", C.c_int32), ("feats_soa", C.c_int32),
                ("T_cur_ref", SE3), ("exposure_rat", C.c_float), ("_pad2", C.c_float)]


class TrackResult(C.Structure):
    _fields_ = [("T_cur_ref", SE3), ("exposure_rat", C.c_float), ("n_tracked", C.c_int32),
                ("n_terms_last", C.c_int32), ("n_saturated_last", C.c_int32),
                ("iters", C.c_int32 * N_PYR_LEVELS), ("n_eval", C.c_int32 * N_PYR_LEVELS),
                ("accept_mask", C.c_uint64 * N_PYR_LEVELS),
                ("huber", C.c_float * N_PYR_LEVELS), ("outlier", C.c_float * N_PYR_LEVELS),
                ("n_select", C.c_int32 * N_PYR_LEVELS), ("energy", C.c_double * N_PYR_LEVELS),
                ("phase_cycles", C.c_uint64 * 10),
                ("status", C.c_int32), ("coop_workgroups", C.c_int16), ("coop_same_xcd", C.c_int16)]


class EvalOut(C.Structure):
    _fields_ = [("H", C.c_double * 49), ("b", C.c_double * 7), ("energy", C.c_double),
                ("energy_sum", C.c_double), ("n_terms", C.c_int32), ("n_saturated", C.c_int32),
                ("n_select", C.c_int32), ("n_visible", C.c_int32),
                ("huber", C.c_float), ("outlier", C.c_float)]


FTR_CORNER, FTR_EDGELET, FTR_GRADIENT = 0, 1, 2


class AlignJob(C.Structure):
    _fields_ = [("ref_frame_id", C.c_int64), ("ref_level", C.c_int32), ("type", C.c_int32),
                ("px_ref", C.c_double * 2), ("f_ref", C.c_double * 3), ("depth", C.c_double),
                ("grad", C.c_double * 2), ("T_cur_ref", SE3), ("px_cur", C.c_double * 2),
                ("exposure_rat", C.c_float), ("kf_gap_lt4", C.c_int32)]


class AlignOut(C.Structure):
    _fields_ = [("success", C.c_int32), ("stage", C.c_int32), ("search_level", C.c_int32), ("iters", C.c_int32),
                ("px_cur", C.c_double * 2), ("A_cur_ref", C.c_double * 4), ("h_inv", C.c_double),
                ("ncc", C.c_float), ("chi2", C.c_float)]


# tables of hso_gpu_reproject_match (hso_kf, hso_obs, hso_map_point, hso_reproj_point)
KF_DTYPE = np.dtype([("frame_id", "<i8"), ("q", "<f8", 4), ("t", "<f8", 3), ("exposure_time", "<f8"),
                     ("keyframe_id", "<i4"), ("pad_", "<i4")])
OBS_DTYPE = np.dtype([("kf", "<i4"), ("level", "<i4"), ("type", "<i4"), ("pad_", "<i4"), ("px", "<f8", 2), ("f", "<f8", 3),
                      ("grad", "<f8", 2)])
MAP_POINT_DTYPE = np.dtype([("pos", "<f8", 3), ("idist", "<f8"), ("host_f", "<f8", 3), ("host_kf", "<i4"),
                            ("obs_begin", "<i4"), ("obs_count", "<i4"), ("pad_", "<i4")])
REPROJ_POINT_DTYPE = np.dtype([("projected", "<i4"), ("cell", "<i4"), ("px", "<f8", 2), ("ref_obs", "<i4"), ("pad_", "<i4")])
REPROJ_FRAME_DTYPE = np.dtype([("cur_frame_id", "<i8"), ("q", "<f8", 4), ("t", "<f8", 3), ("cur_exposure_time", "<f8"),
                               ("cur_keyframe_id", "<i4"), ("kf_begin", "<i4"), ("kf_count", "<i4"), ("point_begin", "<i4"),
                               ("point_count", "<i4"), ("pad_", "<i4")])
assert REPROJ_FRAME_DTYPE.itemsize == 96
assert (KF_DTYPE.itemsize, OBS_DTYPE.itemsize, MAP_POINT_DTYPE.itemsize, REPROJ_POINT_DTYPE.itemsize) == (80, 72, 72, 32)


class PoseFeat(C.Structure):
    _fields_ = [("has_point", C.c_int32), ("type", C.c_int32), ("level", C.c_int32), ("temporary", C.c_int32),
                ("host_pose", C.c_int32), ("_pad", C.c_int32), ("f", C.c_double * 3), ("grad", C.c_double * 2),
                ("host_f", C.c_double * 3), ("idist", C.c_double)]


POSE_FEAT_DTYPE = np.dtype([("has_point", "<i4"), ("type", "<i4"), ("level", "<i4"), ("temporary", "<i4"),
                            ("host_pose", "<i4"), ("_pad", "<i4"), ("f", "<f8", 3), ("grad", "<f8", 2),
                            ("host_f", "<f8", 3), ("idist", "<f8")])
assert POSE_FEAT_DTYPE.itemsize == C.sizeof(PoseFeat)


class PoseJob(C.Structure):
    _fields_ = [("feats", C.c_void_p), ("n_feats", C.c_int32), ("n_poses", C.c_int32), ("poses_f_w", C.c_void_p),
                ("T_f_w", SE3), ("reproj_thresh", C.c_double), ("n_iter", C.c_int32), ("_pad", C.c_int32)]


class PoseResult(C.Structure):
    _fields_ = [("T_f_w", SE3), ("cov", C.c_double * 36), ("estimated_scale", C.c_double),
                ("error_init", C.c_double), ("error_final", C.c_double), ("error_in_px", C.c_float),
                ("num_obs", C.c_int32), ("n_deleted", C.c_int32), ("iters", C.c_int32),
                ("n_trials_total", C.c_int32), ("status", C.c_int32)]


def make_pose_job(feats, poses, T_f_w, reproj_thresh=2.0, n_iter=12):
    """feats: POSE_FEAT_DTYPE array; poses: list of SE3 (host keyframe T_f_w)."""
    feats = np.ascontiguousarray(feats, dtype=POSE_FEAT_DTYPE)
    arr = (SE3 * len(poses))(*poses)
    j = PoseJob()
    j.feats = feats.ctypes.data
    j.n_feats = len(feats)
    j.n_poses = len(poses)
    j.poses_f_w = C.cast(arr, C.c_void_p).value
    j.T_f_w = T_f_w
    j.reproj_thresh = reproj_thresh
    j.n_iter = n_iter
    j._keep = (feats, arr)
    return j


class Seed(C.Structure):
    _fields_ = [("ref_frame_id", C.c_int64), ("level", C.c_int32), ("type", C.c_int32), ("px", C.c_double * 2),
                ("f", C.c_double * 3), ("grad", C.c_double * 2), ("T_ref_w", SE3), ("ref_exposure", C.c_double),
                ("mu", C.c_float), ("sigma2", C.c_float), ("b", C.c_float), ("_pad", C.c_float)]


class SeedOut(C.Structure):
    _fields_ = [("mu", C.c_float), ("sigma2", C.c_float), ("b", C.c_float), ("result", C.c_int32),
                ("is_update", C.c_int32), ("is_valid", C.c_int32), ("search_level", C.c_int32),
                ("epl_start", C.c_int32 * 2), ("epl_end", C.c_int32 * 2), ("n_steps", C.c_int32),
                ("px_cur", C.c_double * 2), ("z", C.c_double), ("zmncc_best", C.c_float), ("zmncc_second", C.c_float)]


SEED_BRIEF_DTYPE = np.dtype([("mu", "<f4"), ("sigma2", "<f4"), ("b", "<f4"), ("result", "i1"), ("is_update", "i1"), ("is_valid", "i1"),
                             ("search_level", "i1")])
assert SEED_BRIEF_DTYPE.itemsize == 16
MATCH_BRIEF_DTYPE = np.dtype([("px", "<f8", 2), ("px_cur", "<f8", 2), ("grad", "<f4", 2), ("cell", "<i4"), ("ref_obs", "<i4"),
                              ("success", "i1"), ("stage", "i1"), ("search_level", "i1"), ("ref_type", "i1"), ("pad_", "<i4")])
assert MATCH_BRIEF_DTYPE.itemsize == 56


class SeedFrame(C.Structure):
    _fields_ = [("frame_id", C.c_int64), ("T_f_w", SE3), ("exposure_time", C.c_double)]


CORNER_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("score", "<i4"), ("response", "<f4")])   # hso_corner
assert CORNER_DTYPE.itemsize == 12
EDGELET_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("gx", "<i2"), ("gy", "<i2"), ("grad", "<f4")])   # hso_edgelet
assert EDGELET_DTYPE.itemsize == 12
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("level", "<i4"), ("species", "<i4"),
                           ("gx", "<i4"), ("gy", "<i4")])   # hso_keypoint
assert KEYPOINT_DTYPE.itemsize == 28
KP_CORNER_HIGH, KP_EDGELET, KP_GRAD, KP_OCCUR = 0, 1, 2, 3
ACTIVATE_MAX_TARGETS = 64


class ActivateTarget(C.Structure):
    _fields_ = [("frame_id", C.c_int64), ("T_f_w", SE3), ("exposure", C.c_double)]


class ActivateOut(C.Structure):
    _fields_ = [("activated", C.c_int32), ("is_valid", C.c_int32), ("n_targets", C.c_int32), ("n_matched", C.c_int32),
                ("dist_mean", C.c_double), ("huber", C.c_double), ("opt_id", C.c_double), ("energy", C.c_double),
                ("n_iter", C.c_int32), ("_pad", C.c_int32)]


class BaDeltasJob(C.Structure):   # hso_ba_deltas_job
    _fields_ = [("poses_f_w", C.c_void_p), ("idist", C.c_void_p), ("edges", C.c_void_p), ("obs_uv", C.c_void_p),
                ("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("huber_corner", C.c_float), ("huber_edge", C.c_float)]


BA_EDGE_DTYPE = np.dtype([("point", "<i4"), ("host", "<i4"), ("target", "<i4"), ("type", "<i4"), ("level", "<i4"),
                          ("_pad", "<i4"), ("fH", "<f8", 3), ("meas", "<f8", 2), ("normal", "<f8", 2)])
assert BA_EDGE_DTYPE.itemsize == 80


class BaResult(C.Structure):
    _fields_ = [("init_chi2", C.c_double), ("final_chi2", C.c_double), ("robust_chi2", C.c_double), ("lambda_", C.c_double),
                ("iterations", C.c_int32), ("n_solves", C.c_int32), ("n_accepted", C.c_int32), ("stop", C.c_int32)]


class BaProblem(C.Structure):
    _fields_ = [("poses_f_w", C.c_void_p), ("pose_fixed", C.c_void_p), ("idist", C.c_void_p), ("edges", C.c_void_p),
                ("edge_chi2_out", C.c_void_p), ("result", C.c_void_p),
                ("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("n_iter", C.c_int32),
                ("huber_corner", C.c_double), ("huber_edge", C.c_double)]


def ba_alloc(n_poses, n_points, n_edges):
    """Output buffers of hso_gpu_ba_linearize."""
    return dict(Hpp=np.zeros(n_points), bp=np.zeros(n_points), Hpc=np.zeros((n_points, n_poses, 6)),
                Hcc=np.zeros((n_poses, n_poses, 6, 6)), bc=np.zeros((n_poses, 6)),
                edge_err=np.zeros((n_edges, 2)), edge_chi2=np.zeros(n_edges), chi2_sum=np.zeros(2))


def make_camera(model, width, height, fx, fy, cx, cy, d=(0, 0, 0, 0, 0), distortion=None):
    cam = Camera()
    cam.model = model
    cam.width, cam.height = int(width), int(height)
    cam.fx, cam.fy, cam.cx, cam.cy = float(fx), float(fy), float(cx), float(cy)
    dd = list(d) + [0.0] * (5 - len(d))
    cam.d[:] = [float(x) for x in dd]
    if distortion is None:
        distortion = int(model == CAM_PINHOLE and abs(dd[0]) > 1e-7)  # src/camera.cpp:37
    cam.distortion = int(distortion)
    return cam


WAIT_DEFAULT, WAIT_POLL, WAIT_NAP, WAIT_BLOCK = 0, 1, 2, 3


class GpuOptions(C.Structure):
    # hso_gpu_options (include/hso_gpu.h): per-context kernel-shape / wait choices; zero = default
    _fields_ = [("size", C.c_int32), ("wait_mode", C.c_int32), ("track_no_coop", C.c_int32), ("track_coop_scatter", C.c_int32),
                ("track_coop_feats_per_wg", C.c_int32), ("track_coop_workgroups", C.c_int32), ("reserved", C.c_int32 * 2)]


class HsoGpuError(RuntimeError):
    pass


_lib = None


def load():
    """Load libhso_gpu.so.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HsoGpuError(
            "libhso_gpu.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python -m hso_amd.build`. There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own libamdhip64; when a process uses both torch and
    # this library (bench.py, tests), torch's copy must be the one HIP runtime of the
    # process, so let it load first.  The shared library itself has no torch dependency.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    P = C.POINTER
    lib.hso_gpu_create.argtypes = [P(vp), i32, vp]
    lib.hso_gpu_destroy.argtypes = [vp]
    lib.hso_gpu_destroy.restype = None
    lib.hso_gpu_last_error.argtypes = [vp]
    lib.hso_gpu_last_error.restype = C.c_char_p
    lib.hso_gpu_abi_version.argtypes = []
    lib.hso_gpu_synchronize.argtypes = [vp]
    lib.hso_gpu_set_shared_device.argtypes = [vp, i32]
    lib.hso_gpu_configure.argtypes = [vp, P(GpuOptions)]
    lib.hso_gpu_set_host_parallel.argtypes = [vp, vp, vp]
    lib.hso_gpu_frame_upload.argtypes = [vp, i64, vp, i32, i32, i32, P(FrameStats)]
    lib.hso_gpu_frame_upload_resized.argtypes = [vp, i64, vp, i32, i32, i32, i32, i32, P(FrameStats)]
    lib.hso_gpu_frame_upload_batch.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.hso_gpu_frame_release.argtypes = [vp, i64]
    lib.hso_gpu_frame_release_batch.argtypes = [vp, vp, i32]
    lib.hso_gpu_frame_download_level.argtypes = [vp, i64, i32, vp, P(i32), P(i32)]
    lib.hso_gpu_frame_download_sobel.argtypes = [vp, i64, i32, vp, vp]
    lib.hso_gpu_make_depth_ref.argtypes = [vp, vp, i32, vp, i32, P(SE3), vp]
    lib.hso_gpu_coarse_track_batch.argtypes = [vp, P(Camera), P(TrackParams), P(TrackJob), i32, P(TrackResult)]
    lib.hso_gpu_coarse_track_prepare.argtypes = [vp, P(Camera), P(TrackParams), P(TrackJob), i32]
    lib.hso_gpu_coarse_track_launch.argtypes = [vp]
    lib.hso_gpu_coarse_track_collect.argtypes = [vp, P(TrackResult)]
    lib.hso_gpu_coarse_track_collect_begin.argtypes = [vp]
    lib.hso_gpu_coarse_track_collect_end.argtypes = [vp, P(TrackResult)]
    lib.hso_gpu_tracker_eval.argtypes = [vp, P(Camera), P(TrackParams), P(TrackJob), i32, P(SE3),
                                         C.c_float, C.c_float, C.c_float, P(EvalOut), vp, vp, vp]
    lib.hso_gpu_tracker_pattern.argtypes = [i32, i32, P(i32), P(i32), vp]
    lib.hso_gpu_align_batch.argtypes = [vp, P(Camera), i64, P(AlignJob), i32, P(AlignOut)]
    lib.hso_gpu_align_multi.argtypes = [vp, P(Camera), P(i64), P(AlignJob), i32, P(AlignOut)]
    lib.hso_gpu_pose_optimize_batch.argtypes = [vp, P(Camera), P(PoseJob), i32, P(PoseResult), vp]
    lib.hso_gpu_ba_linearize.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, C.c_double, C.c_double] + [vp] * 8
    lib.hso_gpu_ba_huber_deltas.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, C.c_double, P(C.c_float), P(C.c_float)]
    lib.hso_gpu_ba_huber_deltas_multi.argtypes = [vp, P(BaDeltasJob), i32, C.c_double]
    lib.hso_gpu_ba_optimize.argtypes = [vp, vp, vp, i32, vp, i32, vp, i32, C.c_double, C.c_double, i32, vp, P(BaResult)]
    lib.hso_gpu_ba_optimize_multi.argtypes = [vp, P(BaProblem), i32]
    lib.hso_gpu_ba_local_multi.argtypes = [vp, P(BaProblem), P(vp), i32, C.c_double, P(C.c_float)]
    lib.hso_gpu_reproject_select.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp]
    lib.hso_gpu_host_alloc.argtypes = [vp, C.c_size_t, P(vp)]
    lib.hso_gpu_host_free.argtypes = [vp, vp]
    lib.hso_gpu_klt_track.argtypes = [vp, i64, i64, vp, vp, i32, P(KltParams), vp]
    lib.hso_gpu_klt_levels.argtypes = [i32, i32, i32, i32]
    lib.hso_gpu_klt_debug_level.argtypes = [vp, i64, i32, vp, vp]
    lib.hso_gpu_seed_activate_multi.argtypes = [vp, P(Camera), P(Seed), i32, P(i32), P(ActivateTarget), P(i32), P(ActivateOut),
                                                P(AlignOut)]
    lib.hso_gpu_seed_observe.argtypes = [vp, P(Camera), i64, P(SE3), C.c_double, C.c_double, P(Seed), i32, P(SeedOut)]
    lib.hso_gpu_seed_activate_frames.argtypes = [vp, P(Camera), P(Seed), i32, P(i32), P(i32), P(ActivateTarget), i32, P(i32), P(ActivateOut)]
    lib.hso_gpu_seed_table_activate.argtypes = [vp, P(Camera), i32, P(i32), i32, P(i32), P(i32), P(ActivateTarget), i32, P(i32), P(ActivateOut)]
    lib.hso_gpu_seed_observe_multi.argtypes = [vp, P(Camera), P(SeedFrame), i32, vp, C.c_double, P(Seed), i32, P(SeedOut)]
    lib.hso_gpu_seed_activate.argtypes = [vp, P(Camera), P(Seed), i32, P(i32), P(ActivateTarget), i32, P(ActivateOut),
                                          P(AlignOut)]
    lib.hso_gpu_seed_table_create.argtypes = [vp, P(i32)]
    lib.hso_gpu_seed_table_destroy.argtypes = [vp, i32]
    lib.hso_gpu_seed_table_append.argtypes = [vp, i32, vp, vp, i32, P(C.c_int32)]
    lib.hso_gpu_seed_table_erase.argtypes = [vp, i32, vp, i32]
    lib.hso_gpu_seed_table_size.argtypes = [vp, i32, P(i32), P(i32)]
    lib.hso_gpu_seed_table_compact.argtypes = [vp, i32, vp]
    lib.hso_gpu_seed_table_observe.argtypes = [vp, P(Camera), i32, P(SeedFrame), i32, C.c_double, vp, vp]
    lib.hso_gpu_seed_table_read.argtypes = [vp, i32, i32, i32, vp]
    lib.hso_gpu_seed_table_observe_previous.argtypes = [vp, P(Camera), i32, vp, P(SeedFrame), i32, C.c_double, vp, vp]
    lib.hso_gpu_seed_table_observe_previous_begin.argtypes = [vp, P(Camera), i32, vp, P(SeedFrame), i32, C.c_double]
    lib.hso_gpu_seed_table_observe_previous_end.argtypes = [vp, i32, vp, i32]
    lib.hso_gpu_seed_reproject_match.argtypes = [vp, P(Camera), i64, P(SE3), C.c_double, vp, i32, i32, i32, vp, vp]
    lib.hso_gpu_fast_detect.argtypes = [vp, i64, i32, i32, i32, vp, i32, P(i32)]
    lib.hso_gpu_fast_detect_batch.argtypes = [vp, P(i64), i32, i32, i32, i32, vp, i32, vp]
    lib.hso_gpu_detect_candidates.argtypes = [vp, P(i64), i32, i32, i32, vp, i32, vp, vp, i32, vp]
    lib.hso_gpu_detect_candidates_multi.argtypes = [vp, P(i64), i32, i32, vp, vp, i32, vp, vp, i32, vp]
    lib.hso_gpu_detect_candidates_init.argtypes = [vp, P(i64), i32, i32, i32, vp, i32, vp, vp, i32, vp]
    lib.hso_gpu_select_octree.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32]
    lib.hso_gpu_reproject_match_multi.argtypes = [vp, P(Camera), vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.hso_gpu_reproject_match.argtypes = [vp, P(Camera), i64, P(SE3), C.c_double, i32, vp, i32, vp, i32, vp, i32, i32, i32,
                                            vp, vp]
    if lib.hso_gpu_abi_version() != ABI_VERSION:
        raise HsoGpuError("libhso_gpu.so reports ABI version %d, this binding is written against %d (include/hso_gpu.h: HSO_GPU_ABI_VERSION); rebuild"
                          % (lib.hso_gpu_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


# Every symbol include/hso_gpu.h declares; tests check the library exports all of them.
EXPORTED_SYMBOLS = [
    "hso_gpu_ba_huber_deltas_multi", "hso_gpu_ba_local_multi",
    "hso_gpu_create", "hso_gpu_destroy", "hso_gpu_last_error", "hso_gpu_abi_version",
    "hso_gpu_synchronize", "hso_gpu_set_shared_device", "hso_gpu_device_cpulist", "hso_gpu_configure", "hso_gpu_set_host_parallel", "hso_gpu_frame_upload", "hso_gpu_frame_upload_batch", "hso_gpu_frame_release", "hso_gpu_frame_release_batch",
    "hso_gpu_frame_download_level", "hso_gpu_frame_download_sobel", "hso_gpu_make_depth_ref",
    "hso_gpu_coarse_track_batch", "hso_gpu_coarse_track_prepare", "hso_gpu_coarse_track_launch",
    "hso_gpu_coarse_track_collect", "hso_gpu_coarse_track_collect_begin", "hso_gpu_coarse_track_collect_end", "hso_gpu_tracker_eval", "hso_gpu_tracker_pattern",
    "hso_gpu_align_batch", "hso_gpu_align_multi", "hso_gpu_pose_optimize_batch", "hso_gpu_ba_linearize",
    "hso_gpu_seed_observe", "hso_gpu_seed_activate", "hso_gpu_fast_detect", "hso_gpu_fast_detect_batch",
    "hso_gpu_detect_candidates", "hso_gpu_select_octree", "hso_gpu_reproject_match", "hso_gpu_seed_observe_multi",
    "hso_gpu_detect_candidates_init", "hso_gpu_detect_candidates_multi", "hso_gpu_frame_upload_resized",
    "hso_gpu_reproject_match_multi", "hso_gpu_ba_huber_deltas", "hso_gpu_ba_optimize", "hso_gpu_ba_optimize_multi",
    "hso_gpu_seed_activate_multi", "hso_gpu_seed_activate_frames", "hso_gpu_seed_table_activate", "hso_gpu_reproject_select",
    "hso_gpu_seed_reproject_match",
    "hso_gpu_seed_table_create", "hso_gpu_seed_table_destroy", "hso_gpu_seed_table_append", "hso_gpu_seed_table_erase",
    "hso_gpu_seed_table_size", "hso_gpu_seed_table_observe", "hso_gpu_seed_table_read",
    "hso_gpu_klt_track", "hso_gpu_klt_levels", "hso_gpu_host_alloc", "hso_gpu_host_free", "hso_gpu_seed_table_compact",
    "hso_gpu_seqmap_create", "hso_gpu_seqmap_destroy", "hso_gpu_seqmap_set_keyframes", "hso_gpu_seqmap_patch", "hso_gpu_seqmap_patch_multi", "hso_gpu_seqmap_size", "hso_gpu_seqmap_read",
    "hso_gpu_seqmap_configure", "hso_gpu_seqmap_patch_lists", "hso_gpu_seqmap_patch_links", "hso_gpu_seqmap_set_key_points",
    "hso_gpu_seq_chain", "hso_gpu_seq_local_ba", "hso_gpu_seq_events", "hso_gpu_seq_frame_features", "hso_gpu_seq_set_frame_features",
    "hso_gpu_seed_table_observe_groups", "hso_gpu_seed_table_set_host_pose",
    "hso_gpu_seed_table_observe_previous", "hso_gpu_seed_table_observe_previous_begin", "hso_gpu_seed_table_observe_previous_end",
]


# ... and every symbol include/hso_gpu_debug.h declares (parity / trace read-backs, developer probes: not part of the boundary)
DEBUG_SYMBOLS = ["hso_gpu_debug_census", "hso_gpu_klt_debug_level", "hso_gpu_seq_debug_list", "hso_gpu_seq_debug_ref_table", "hso_gpu_debug_fetch",
                 "hso_gpu_seqmap_debug_dump", "hso_gpu_seq_ba_debug_window"]


def select_octree(keys, width, height, n_features):
    """computeKeyPointsOctTree over (0, width, 0, height): keys = KEYPOINT_DTYPE array in the
    reference's allFeturesToDistribute_ order -> the selected keys in node-list order."""
    keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    out = np.zeros(max(len(keys), 1), KEYPOINT_DTYPE)
    n = load().hso_gpu_select_octree(_ptr(keys), len(keys), 0, width, 0, height, n_features, _ptr(out), len(out))
    if n < 0:
        raise HsoGpuError("select_octree: status %d" % n)
    return out[:n].copy()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """Thin RAII wrapper over hso_gpu_ctx."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.hso_gpu_create(C.byref(self.h), int(device), C.c_void_p(stream or 0))
        if rc < 0:
            raise HsoGpuError("hso_gpu_create failed: %d" % rc)
        self._keep = []
        self._n_host_live, self._close_pending = 0, False

    def close(self):
        """Destroys the context — once no array of host_array() is alive any more: those arrays are views of memory the context
        owns, so closing earlier would leave them dangling; the destruction then happens when the last of them goes."""
        if self._n_host_live > 0:
            self._close_pending = True
            return
        if self.h:
            self.lib.hso_gpu_destroy(self.h)
            self.h = C.c_void_p()

    def _host_array_gone(self):
        self._n_host_live -= 1
        if self._close_pending and self._n_host_live == 0:
            self._close_pending = False
            self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.hso_gpu_last_error(self.h)
            raise HsoGpuError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
        return rc

    def synchronize(self):
        self._check(self.lib.hso_gpu_synchronize(self.h), "synchronize")

    def configure(self, wait_mode=0, track_no_coop=False, track_coop_scatter=False, track_coop_feats_per_wg=0, track_coop_workgroups=0):
        """hso_gpu_configure: the context's kernel-shape / wait choices (every call sets all of them; no arguments = the defaults)"""
        o = GpuOptions(C.sizeof(GpuOptions), int(wait_mode), int(bool(track_no_coop)), int(bool(track_coop_scatter)), int(track_coop_feats_per_wg),
                       int(track_coop_workgroups))
        self._check(self.lib.hso_gpu_configure(self.h, C.byref(o)), "configure")

    def device_cpulist(self):
        """hso_gpu_device_cpulist: the CPUs of the NUMA node the device is attached to, as a set of ints (empty: unknown)"""
        buf = C.create_string_buffer(2048)
        self.lib.hso_gpu_device_cpulist.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._check(self.lib.hso_gpu_device_cpulist(self.h, buf, len(buf)), "device_cpulist")
        cpus = set()
        for part in buf.value.decode().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus

    def host_array(self, shape, dtype):
        """A numpy array over page-locked memory of hso_gpu_host_alloc (zeroed): pass it as an `out=` / input table and the DMA goes
        straight to it.  Lives as long as the context."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        p = C.c_void_p()
        self._check(self.lib.hso_gpu_host_alloc(self.h, max(n * dt.itemsize, 1), C.byref(p)), "host_alloc")
        buf = (C.c_char * max(n * dt.itemsize, 1)).from_address(p.value)
        # the memory belongs to the context: the buffer object keeps the Context alive (an array that outlives its creator's
        # last reference keeps the context, and with it the allocation, open), and close() refuses while such arrays exist
        buf._hso_owner = self
        self._n_host_live += 1
        weakref.finalize(buf, self._host_array_gone)
        a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
        a[...] = np.zeros((), dt)
        return a

    def host_free(self, a):
        """Give a host_array's memory back now (the array must not be used afterwards)."""
        base = a
        while getattr(base, "base", None) is not None:
            base = base.base
        addr = C.addressof(base.obj) if isinstance(base, memoryview) else C.addressof(base)
        self._check(self.lib.hso_gpu_host_free(self.h, C.c_void_p(addr)), "host_free")

    # -- frames
    def frame_upload(self, frame_id, img, device_ptr=None, width=None, height=None):
        st = FrameStats()
        if device_ptr is not None:
            rc = self.lib.hso_gpu_frame_upload(self.h, frame_id, C.c_void_p(device_ptr), width, height, 1, C.byref(st))
        else:
            img = np.ascontiguousarray(img, dtype=np.uint8)
            h, w = img.shape
            rc = self.lib.hso_gpu_frame_upload(self.h, frame_id, _ptr(img), w, h, 0, C.byref(st))
        self._check(rc, "frame_upload")
        return st

    def frame_upload_resized(self, frame_id, img, width, height):
        """Sensor image -> cv::resize to the camera size on the device -> Frame."""
        st = FrameStats()
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        self._check(self.lib.hso_gpu_frame_upload_resized(self.h, frame_id, _ptr(img), w, h, width, height, 0, C.byref(st)),
                    "frame_upload_resized")
        return st

    def frame_upload_batch(self, frame_ids, imgs=None, device_ptrs=None, width=None, height=None, want_stats=True):
        n = len(frame_ids)
        ids = np.ascontiguousarray(frame_ids, np.int64)
        if device_ptrs is not None:
            ptrs = np.ascontiguousarray(device_ptrs, np.uint64)
            is_dev = 1
        else:
            imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
            height, width = imgs[0].shape
            ptrs = np.array([im.ctypes.data for im in imgs], np.uint64)
            is_dev = 0
        stats = (FrameStats * n)() if want_stats else None
        rc = self.lib.hso_gpu_frame_upload_batch(self.h, _ptr(ids), _ptr(ptrs), n, width, height, is_dev,
                                                 C.cast(stats, C.c_void_p) if want_stats else None)
        self._check(rc, "frame_upload_batch")
        return list(stats) if want_stats else None

    def frame_release(self, frame_id):
        self._check(self.lib.hso_gpu_frame_release(self.h, frame_id), "frame_release")

    @staticmethod
    def level_dims(w0, h0, level):
        """(w, h) of a pyramid level: halfSample sizes for multiples of 16, else the cvRound sizes of
        the cv::resize branch (src/frame.cpp:302-312)."""
        if w0 % 16 == 0 and h0 % 16 == 0:
            return w0 >> level, h0 >> level
        sc = np.float32(1.0 / (1 << level))
        return int(np.rint(np.float32(w0) * sc)), int(np.rint(np.float32(h0) * sc))

    def frame_level(self, frame_id, level, w0, h0):
        lw, lh = self.level_dims(w0, h0, level)
        out = np.empty((lh, lw), np.uint8)
        w, h = C.c_int(), C.c_int()
        self._check(self.lib.hso_gpu_frame_download_level(self.h, frame_id, level, _ptr(out), C.byref(w), C.byref(h)),
                    "frame_download_level")
        assert (h.value, w.value) == out.shape
        return out

    def frame_sobel(self, frame_id, level, w0, h0):
        lw, lh = self.level_dims(w0, h0, level)
        gx = np.empty((lh, lw), np.int16)
        gy = np.empty_like(gx)
        self._check(self.lib.hso_gpu_frame_download_sobel(self.h, frame_id, level, _ptr(gx), _ptr(gy)),
                    "frame_download_sobel")
        return gx, gy

    # -- tracker
    def make_depth_ref(self, din, poses, T_ref_w):
        din = np.ascontiguousarray(din, dtype=DEPTH_REF_IN_DTYPE)
        poses_arr = (SE3 * len(poses))(*poses)
        out = np.empty(len(din), np.float64)
        self._check(self.lib.hso_gpu_make_depth_ref(self.h, _ptr(din), len(din), C.cast(poses_arr, C.c_void_p),
                                                    len(poses), C.byref(T_ref_w), _ptr(out)), "make_depth_ref")
        return out

    @staticmethod
    def make_job(ref_id, cur_id, feats, T_cur_ref, exposure_rat):
        feats = np.ascontiguousarray(feats, dtype=REF_FEAT_DTYPE)
        job = TrackJob()
        job.ref_frame_id, job.cur_frame_id = ref_id, cur_id
        job.feats = feats.ctypes.data
        job.n_feats = len(feats)
        job.T_cur_ref = T_cur_ref
        job.exposure_rat = float(exposure_rat)
        job._feats_keepalive = feats
        return job

    def coarse_track_batch(self, cam, params, jobs):
        arr = (TrackJob * len(jobs))(*jobs)
        res = (TrackResult * len(jobs))()
        self._check(self.lib.hso_gpu_coarse_track_batch(self.h, C.byref(cam), C.byref(params), arr, len(jobs), res),
                    "coarse_track_batch")
        return list(res)

    def coarse_track_prepare(self, cam, params, jobs):
        arr = (TrackJob * len(jobs))(*jobs)
        self._n_prepared = len(jobs)
        self._check(self.lib.hso_gpu_coarse_track_prepare(self.h, C.byref(cam), C.byref(params), arr, len(jobs)),
                    "coarse_track_prepare")

    def coarse_track_launch(self):
        self._check(self.lib.hso_gpu_coarse_track_launch(self.h), "coarse_track_launch")

    def coarse_track_collect(self, as_list=True):
        """as_list=False returns the ctypes array itself (records are wrapped on access): a list of 4096 wrappers per
        call is ~1 ms of interpreter time and feeds the garbage collector."""
        res = (TrackResult * self._n_prepared)()
        self._check(self.lib.hso_gpu_coarse_track_collect(self.h, res), "coarse_track_collect")
        return list(res) if as_list else res

    def coarse_track_collect_begin(self):
        self._check(self.lib.hso_gpu_coarse_track_collect_begin(self.h), "coarse_track_collect_begin")

    def coarse_track_collect_end(self, as_list=True):
        res = (TrackResult * self._n_prepared)()
        self._check(self.lib.hso_gpu_coarse_track_collect_end(self.h, res), "coarse_track_collect_end")
        return list(res) if as_list else res

    # -- reprojection matching
    def align_batch(self, cam, cur_frame_id, jobs, as_list=True):
        """jobs: list of AlignJob or a ready ctypes array (no per-call marshalling)."""
        arr = jobs if isinstance(jobs, C.Array) else (AlignJob * len(jobs))(*jobs)
        out = (AlignOut * len(jobs))()
        self._check(self.lib.hso_gpu_align_batch(self.h, C.byref(cam), cur_frame_id, arr, len(jobs), out), "align_batch")
        return list(out) if as_list else out

    def reproject_match(self, cam, cur_frame_id, T_cur_w, cur_exposure_time, cur_keyframe_id, kfs, points, obs,
                        cell_size, grid_n_cols):
        """kfs / points / obs: KF_DTYPE / MAP_POINT_DTYPE / OBS_DTYPE arrays.
        Returns (REPROJ_POINT_DTYPE array, ctypes AlignOut array), both in point order."""
        kfs = np.ascontiguousarray(kfs, KF_DTYPE); points = np.ascontiguousarray(points, MAP_POINT_DTYPE)
        obs = np.ascontiguousarray(obs, OBS_DTYPE)
        proj = np.zeros(len(points), REPROJ_POINT_DTYPE)
        match = (AlignOut * max(len(points), 1))()
        self._check(self.lib.hso_gpu_reproject_match(self.h, C.byref(cam), cur_frame_id, C.byref(T_cur_w), cur_exposure_time,
                                                     cur_keyframe_id, _ptr(kfs), len(kfs), _ptr(points), len(points), _ptr(obs),
                                                     len(obs), cell_size, grid_n_cols, _ptr(proj), match), "reproject_match")
        return proj, match

    def reproject_match_multi(self, cam, frames, kfs, points, obs, cell_size, grid_n_cols):
        """frames: REPROJ_FRAME_DTYPE array (one row per current frame); tables as in reproject_match,
        with host_kf / obs kf relative to the owning frame's kf_begin."""
        frames = np.ascontiguousarray(frames, REPROJ_FRAME_DTYPE)
        kfs = np.ascontiguousarray(kfs, KF_DTYPE); points = np.ascontiguousarray(points, MAP_POINT_DTYPE)
        obs = np.ascontiguousarray(obs, OBS_DTYPE)
        proj = np.zeros(len(points), REPROJ_POINT_DTYPE)
        match = (AlignOut * max(len(points), 1))()
        self._check(self.lib.hso_gpu_reproject_match_multi(self.h, C.byref(cam), _ptr(frames), len(frames), _ptr(kfs), len(kfs),
                                                           _ptr(points), len(points), _ptr(obs), len(obs), cell_size, grid_n_cols,
                                                           _ptr(proj), match), "reproject_match_multi")
        return proj, match

    def align_multi(self, cam, cur_frame_ids, jobs, as_list=True):
        """jobs[i] is searched in frame cur_frame_ids[i]; jobs may be a ready ctypes array."""
        n = len(jobs)
        arr = jobs if isinstance(jobs, C.Array) else (AlignJob * n)(*jobs)
        ids = (C.c_int64 * n)(*cur_frame_ids)
        out = (AlignOut * n)()
        self._check(self.lib.hso_gpu_align_multi(self.h, C.byref(cam), ids, arr, n, out), "align_multi")
        return list(out) if as_list else out

    # -- pose optimiser
    def pose_optimize_batch(self, cam, jobs):
        arr = (PoseJob * len(jobs))(*jobs)
        res = (PoseResult * len(jobs))()
        masks = [np.zeros(max(j.n_feats, 1), np.uint8) for j in jobs]
        mptr = (C.c_void_p * len(jobs))(*[m.ctypes.data for m in masks])
        self._check(self.lib.hso_gpu_pose_optimize_batch(self.h, C.byref(cam), arr, len(jobs), res, mptr), "pose_optimize_batch")
        return list(res), [m[:j.n_feats] for m, j in zip(masks, jobs)]

    # -- bundle adjustment linearisation
    def ba_linearize(self, poses, fixed, idist, edges, huber_corner, huber_edge):
        parr = (SE3 * len(poses))(*poses)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        idist = np.ascontiguousarray(idist, np.float64)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        o = ba_alloc(len(poses), len(idist), len(edges))
        rc = self.lib.hso_gpu_ba_linearize(self.h, C.cast(parr, C.c_void_p), _ptr(fixed), len(poses), _ptr(idist), len(idist),
                                           _ptr(edges), len(edges), huber_corner, huber_edge, _ptr(o["Hpp"]), _ptr(o["bp"]),
                                           _ptr(o["Hpc"]), _ptr(o["Hcc"]), _ptr(o["bc"]), _ptr(o["edge_err"]),
                                           _ptr(o["edge_chi2"]), _ptr(o["chi2_sum"]))
        self._check(rc, "ba_linearize")
        return o

    def ba_huber_deltas(self, poses, idist, edges, obs_uv, error_multiplier2):
        """Huber deltas of LocalBundleAdjustment (bundle_adjustment.cpp:618-680) -> (huber_corner, huber_edge) as floats."""
        parr = (SE3 * len(poses))(*poses)
        idist = np.ascontiguousarray(idist, np.float64)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        obs_uv = np.ascontiguousarray(obs_uv, np.float64)
        hc, he = C.c_float(), C.c_float()
        self._check(self.lib.hso_gpu_ba_huber_deltas(self.h, C.cast(parr, C.c_void_p), len(poses), _ptr(idist), len(idist), _ptr(edges),
                                                     _ptr(obs_uv), len(edges), error_multiplier2, C.byref(hc), C.byref(he)),
                    "ba_huber_deltas")
        return hc.value, he.value

    def ba_huber_deltas_multi(self, windows, error_multiplier2):
        """hso_gpu_ba_huber_deltas_multi: windows = [(poses, idist, edges, obs_uv), ...] -> [(huber_corner, huber_edge), ...]."""
        jobs = (BaDeltasJob * max(len(windows), 1))()
        keep = []
        for j, (poses, idist, edges, obs_uv) in zip(jobs, windows):
            parr = (SE3 * len(poses))(*poses)
            idist = np.ascontiguousarray(idist, np.float64)
            edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
            obs_uv = np.ascontiguousarray(obs_uv, np.float64)
            keep.append((parr, idist, edges, obs_uv))
            j.poses_f_w = C.cast(parr, C.c_void_p); j.idist = idist.ctypes.data; j.edges = edges.ctypes.data if len(edges) else None
            j.obs_uv = obs_uv.ctypes.data if len(edges) else None
            j.n_poses = len(poses); j.n_points = len(idist); j.n_edges = len(edges)
        self._check(self.lib.hso_gpu_ba_huber_deltas_multi(self.h, jobs, len(windows), error_multiplier2), "ba_huber_deltas_multi")
        return [(jobs[i].huber_corner, jobs[i].huber_edge) for i in range(len(windows))]

    def ba_optimize(self, poses, fixed, idist, edges, huber_corner, huber_edge, n_iter):
        """The LM optimisation of LocalBundleAdjustment.  Returns (poses, idist, edge_chi2, BaResult)."""
        parr = (SE3 * len(poses))(*poses)
        fixed = np.ascontiguousarray(fixed, np.uint8)
        idist = np.array(idist, np.float64)
        edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
        chi2 = np.zeros(len(edges))
        res = BaResult()
        self._check(self.lib.hso_gpu_ba_optimize(self.h, C.cast(parr, C.c_void_p), _ptr(fixed), len(poses), _ptr(idist), len(idist),
                                                 _ptr(edges), len(edges), huber_corner, huber_edge, n_iter, _ptr(chi2), C.byref(res)),
                    "ba_optimize")
        return list(parr), idist, chi2, res

    def reproject_select(self, frame_begin, cell, quality, flags, cell_order, max_fts):
        """The cell passes of Reprojector::reprojectMap for any number of frames.  Returns (per-frame list of
        (index, taken) in examination order, counts[n_frames, 4])."""
        fb = np.ascontiguousarray(frame_begin, np.int32)
        cell = np.ascontiguousarray(cell, np.int32); quality = np.ascontiguousarray(quality, np.uint8)
        flags = np.ascontiguousarray(flags, np.uint8); order = np.ascontiguousarray(cell_order, np.int32)
        nf = len(fb) - 1
        out = np.zeros(max(len(cell), 1), np.int32)
        counts = np.zeros((nf, 4), np.int32)
        self._check(self.lib.hso_gpu_reproject_select(self.h, _ptr(fb), nf, _ptr(cell), _ptr(quality), _ptr(flags), _ptr(order),
                                                      len(order), max_fts, _ptr(out), _ptr(counts)), "reproject_select")
        res = []
        for f in range(nf):
            ex = out[fb[f]:fb[f] + counts[f, 0]].astype(np.int64)
            res.append([(int(v & 0x7fffffff), bool(v & 0x80000000)) for v in ex])
        return res, counts

    def ba_optimize_multi(self, problems):
        """problems: list of (poses, fixed, idist, edges, huber_corner, huber_edge, n_iter); one call, the windows advance in
        lockstep.  Returns a list of (poses, idist, edge_chi2, BaResult)."""
        keep, arr = [], (BaProblem * len(problems))()
        for q, (poses, fixed, idist, edges, hc, he, n_iter) in enumerate(problems):
            parr = (SE3 * len(poses))(*poses)
            fixed = np.ascontiguousarray(fixed, np.uint8)
            idist = np.array(idist, np.float64)
            edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
            chi2 = np.zeros(len(edges))
            res = BaResult()
            keep.append((parr, fixed, idist, edges, chi2, res))
            P_ = arr[q]
            P_.poses_f_w = C.cast(parr, C.c_void_p).value; P_.pose_fixed = fixed.ctypes.data; P_.idist = idist.ctypes.data
            P_.edges = edges.ctypes.data; P_.edge_chi2_out = chi2.ctypes.data; P_.result = C.addressof(res)
            P_.n_poses, P_.n_points, P_.n_edges, P_.n_iter = len(poses), len(idist), len(edges), n_iter
            P_.huber_corner, P_.huber_edge = hc, he
        self._check(self.lib.hso_gpu_ba_optimize_multi(self.h, arr, len(problems)), "ba_optimize_multi")
        return [(list(k[0]), k[2], k[4], k[5]) for k in keep]

    def ba_local_multi(self, problems, error_multiplier2):
        """hso_gpu_ba_local_multi: problems = list of (poses, fixed, idist, edges, obs_uv, n_iter); the Huber deltas are formed on
        the device, then the windows are optimised with them.  Returns (list of (poses, idist, edge_chi2, BaResult), hubers[n, 2])."""
        keep, arr = [], (BaProblem * len(problems))()
        uvp = (C.c_void_p * max(len(problems), 1))()
        for q, (poses, fixed, idist, edges, obs_uv, n_iter) in enumerate(problems):
            parr = (SE3 * len(poses))(*poses)
            fixed = np.ascontiguousarray(fixed, np.uint8)
            idist = np.array(idist, np.float64)
            edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
            uv = np.ascontiguousarray(obs_uv, np.float64)
            chi2 = np.zeros(len(edges))
            res = BaResult()
            keep.append((parr, fixed, idist, edges, chi2, res, uv))
            P_ = arr[q]
            P_.poses_f_w = C.cast(parr, C.c_void_p).value; P_.pose_fixed = fixed.ctypes.data; P_.idist = idist.ctypes.data
            P_.edges = edges.ctypes.data; P_.edge_chi2_out = chi2.ctypes.data; P_.result = C.addressof(res)
            P_.n_poses, P_.n_points, P_.n_edges, P_.n_iter = len(poses), len(idist), len(edges), n_iter
            P_.huber_corner, P_.huber_edge = 0.0, 0.0
            uvp[q] = uv.ctypes.data
        hub = np.zeros((max(len(problems), 1), 2), np.float32)
        self._check(self.lib.hso_gpu_ba_local_multi(self.h, arr, uvp, len(problems), float(error_multiplier2), hub.ctypes.data_as(C.POINTER(C.c_float))), "ba_local_multi")
        return [(list(k[0]), k[2], k[4], k[5]) for k in keep], hub[:len(problems)]

    # -- depth-filter seed observation
    def seed_observe(self, cam, cur_frame_id, cur_T_f_w, cur_exposure, px_error_angle, seeds, as_list=True):
        arr = seeds if isinstance(seeds, C.Array) else (Seed * len(seeds))(*seeds)
        out = (SeedOut * len(seeds))()
        self._check(self.lib.hso_gpu_seed_observe(self.h, C.byref(cam), cur_frame_id, C.byref(cur_T_f_w), cur_exposure,
                                                  px_error_angle, arr, len(seeds), out), "seed_observe")
        return list(out) if as_list else out

    def seed_observe_multi(self, cam, frames, seed_frame, px_error_angle, seeds, as_list=True):
        """frames: list of (frame_id, SE3 T_f_w, exposure_time); seed i is observed in frames[seed_frame[i]]."""
        fr = (SeedFrame * len(frames))()
        for k, (fid, T, expo) in enumerate(frames):
            fr[k].frame_id, fr[k].T_f_w, fr[k].exposure_time = fid, T, expo
        n = len(seeds)
        arr = seeds if isinstance(seeds, C.Array) else (Seed * n)(*seeds)
        idx = np.ascontiguousarray(seed_frame, np.int32)
        out = (SeedOut * max(n, 1))()
        self._check(self.lib.hso_gpu_seed_observe_multi(self.h, C.byref(cam), fr, len(frames), _ptr(idx), px_error_angle, arr, n, out),
                    "seed_observe_multi")
        return list(out)[:n] if as_list else out

    # -- resident seed tables
    def seed_table_create(self):
        t = C.c_int()
        self._check(self.lib.hso_gpu_seed_table_create(self.h, C.byref(t)), "seed_table_create")
        return t.value

    def seed_table_destroy(self, table):
        self._check(self.lib.hso_gpu_seed_table_destroy(self.h, table), "seed_table_destroy")

    def seed_table_append(self, table, seeds, group=None):
        n = len(seeds)
        arr = seeds if isinstance(seeds, C.Array) else (Seed * max(n, 1))(*seeds)
        grp = np.ascontiguousarray(group, np.int32) if group is not None else None
        first = C.c_int32()
        self._check(self.lib.hso_gpu_seed_table_append(self.h, table, C.cast(arr, C.c_void_p), _ptr(grp), n, C.byref(first)), "seed_table_append")
        return first.value

    def seed_table_erase(self, table, slots):
        sl = np.ascontiguousarray(slots, np.int32)
        self._check(self.lib.hso_gpu_seed_table_erase(self.h, table, _ptr(sl), len(sl)), "seed_table_erase")

    def seed_table_size(self, table):
        a, b = C.c_int(), C.c_int()
        self._check(self.lib.hso_gpu_seed_table_size(self.h, table, C.byref(a), C.byref(b)), "seed_table_size")
        return a.value, b.value

    def seed_table_observe(self, cam, table, frames, px_error_angle, want_brief=True, want_full=False, brief_out=None):
        """frames: list of (frame_id, SE3 T_f_w, exposure_time), one per group.  -> (brief array or None, full ctypes array or None)
        brief_out: a SEED_BRIEF_DTYPE array of >= the table's size to receive the briefs (host_array(): no staging copy)."""
        fr = (SeedFrame * len(frames))()
        for k, (fid, T, expo) in enumerate(frames):
            fr[k].frame_id, fr[k].T_f_w, fr[k].exposure_time = fid, T, expo
        n, _ = self.seed_table_size(table)
        if brief_out is not None:
            assert brief_out.dtype == SEED_BRIEF_DTYPE and len(brief_out) >= n and brief_out.flags.c_contiguous
            brief = brief_out[:n]
        else:
            brief = np.zeros(n, SEED_BRIEF_DTYPE) if want_brief else None
        full = (SeedOut * max(n, 1))() if want_full else None
        self._check(self.lib.hso_gpu_seed_table_observe(self.h, C.byref(cam), table, fr, len(frames), px_error_angle, _ptr(brief),
                                                        C.cast(full, C.c_void_p) if want_full else None), "seed_table_observe")
        return brief, full

    def seed_table_observe_previous(self, cam, table, pairs, px_error_angle, want_brief=True, want_full=False):
        """pairs: list of (host keyframe id, (earlier frame id, SE3 T_f_w, exposure_time)).  -> (brief array or None, full ctypes array or None)"""
        n = len(pairs)
        ids = np.ascontiguousarray([p[0] for p in pairs], np.int64)
        fr = (SeedFrame * max(n, 1))()
        for k, (_, (fid, T, expo)) in enumerate(pairs):
            fr[k].frame_id, fr[k].T_f_w, fr[k].exposure_time = fid, T, expo
        m, _ = self.seed_table_size(table)
        brief = np.zeros(m, SEED_BRIEF_DTYPE) if want_brief else None
        full = (SeedOut * max(m, 1))() if want_full else None
        self._check(self.lib.hso_gpu_seed_table_observe_previous(self.h, C.byref(cam), table, _ptr(ids), fr, n, px_error_angle, _ptr(brief),
                                                                 C.cast(full, C.c_void_p) if want_full else None), "seed_table_observe_previous")
        return brief, full

    def seed_table_observe_previous_begin(self, cam, table, pairs, px_error_angle):
        """the same pass queued on the depth filter's stream; collect with seed_table_observe_previous_end"""
        n = len(pairs)
        ids = np.ascontiguousarray([p[0] for p in pairs], np.int64)
        fr = (SeedFrame * max(n, 1))()
        for k, (_, (fid, T, expo)) in enumerate(pairs):
            fr[k].frame_id, fr[k].T_f_w, fr[k].exposure_time = fid, T, expo
        self._check(self.lib.hso_gpu_seed_table_observe_previous_begin(self.h, C.byref(cam), table, _ptr(ids), fr, n, px_error_angle), "seed_table_observe_previous_begin")

    def seed_table_observe_previous_end(self, table, n_slots):
        brief = np.zeros(max(n_slots, 1), SEED_BRIEF_DTYPE)
        self._check(self.lib.hso_gpu_seed_table_observe_previous_end(self.h, table, _ptr(brief), len(brief)), "seed_table_observe_previous_end")
        return brief[:n_slots]

    def seed_table_compact(self, table):
        """-> (new number of slots, remap[old slot] = new slot or -1)"""
        n, _ = self.seed_table_size(table)
        remap = np.full(max(n, 1), -1, np.int32)
        k = self._check(self.lib.hso_gpu_seed_table_compact(self.h, table, _ptr(remap)), "seed_table_compact")
        return k, remap[:n]

    def seed_table_read(self, table, first, n):
        out = (Seed * max(n, 1))()
        self._check(self.lib.hso_gpu_seed_table_read(self.h, table, first, n, C.cast(out, C.c_void_p)), "seed_table_read")
        return out

    def klt_track(self, frame_prev, frame_cur, px_prev, px_init, params=None):
        """initialization::trackKlt's device part between two resident frames -> KLT_RESULT_DTYPE array."""
        a = np.ascontiguousarray(px_prev, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(px_init, np.float32).reshape(-1, 2)
        if len(a) != len(b):
            raise ValueError("klt_track: px_prev and px_init differ in length")
        out = np.zeros(len(a), KLT_RESULT_DTYPE)
        params = params or KltParams()
        self._check(self.lib.hso_gpu_klt_track(self.h, frame_prev, frame_cur, _ptr(a), _ptr(b), len(a), C.byref(params), _ptr(out)), "klt_track")
        return out

    def klt_debug_level(self, frame, level, width, height):
        """Gaussian pyramid level + Scharr image the KLT tracker uses (test hook) -> (img uint8 [h, w], deriv int16 [h, w, 2])."""
        w, h = width, height
        for _ in range(level):
            w, h = (w + 1) // 2, (h + 1) // 2
        img = np.zeros((h, w), np.uint8); der = np.zeros((h, w, 2), np.int16)
        self._check(self.lib.hso_gpu_klt_debug_level(self.h, frame, level, _ptr(img), _ptr(der)), "klt_debug_level")
        return img, der

    # -- FAST-9 corner candidates
    def fast_detect(self, frame_id, n_levels=3, threshold=20, border=8, cap=20000):
        """Returns ([structured array per level], [count per level])."""
        out = np.zeros((n_levels, cap), CORNER_DTYPE)
        counts = (C.c_int32 * n_levels)()
        self._check(self.lib.hso_gpu_fast_detect(self.h, frame_id, n_levels, threshold, border, _ptr(out), cap, counts),
                    "fast_detect")
        return [out[l, :min(counts[l], cap)].copy() for l in range(n_levels)], list(counts)

    def fast_detect_batch(self, frame_ids, n_levels=3, threshold=20, border=8, cap=4096):
        """Returns (out[n_frames, n_levels, cap] structured array or None when cap == 0, counts[n_frames, n_levels])."""
        n = len(frame_ids)
        ids = (C.c_int64 * n)(*frame_ids)
        out = np.zeros((n, n_levels, cap), CORNER_DTYPE) if cap > 0 else None
        counts = np.zeros((n, n_levels), np.int32)
        self._check(self.lib.hso_gpu_fast_detect_batch(self.h, ids, n, n_levels, threshold, border, _ptr(out), cap, _ptr(counts)),
                    "fast_detect_batch")
        return out, counts

    def detect_candidates(self, frame_ids, n_levels=3, min_thresh=20, corner_cap=8192, edgelet_cap=4800):
        """fastDetectMT + edgeLetDetectMT of FeatureExtractor::detect for a batch of keyframes.
        Returns (corners[n, L, cap], corner_counts[n, L], edgelets[n, L, cap], edgelet_counts[n, L])."""
        n = len(frame_ids)
        ids = (C.c_int64 * n)(*frame_ids)
        co = np.zeros((n, n_levels, corner_cap), CORNER_DTYPE) if corner_cap > 0 else None
        eo = np.zeros((n, n_levels, edgelet_cap), EDGELET_DTYPE) if edgelet_cap > 0 else None
        cc, ec = np.zeros((n, n_levels), np.int32), np.zeros((n, n_levels), np.int32)
        self._check(self.lib.hso_gpu_detect_candidates(self.h, ids, n, n_levels, min_thresh, _ptr(co), corner_cap, _ptr(cc),
                                                       _ptr(eo), edgelet_cap, _ptr(ec)), "detect_candidates")
        return co, cc, eo, ec

    def detect_candidates_multi(self, frame_ids, min_thresh, n_levels=3, corner_cap=8192, edgelet_cap=4800):
        """hso_gpu_detect_candidates_multi: one barrier per frame (min_thresh[i] of frame i); returns what detect_candidates does."""
        n = len(frame_ids)
        ids = (C.c_int64 * n)(*frame_ids)
        th = np.ascontiguousarray(min_thresh, np.int32)
        assert len(th) == n
        co = np.zeros((n, n_levels, corner_cap), CORNER_DTYPE) if corner_cap > 0 else None
        eo = np.zeros((n, n_levels, edgelet_cap), EDGELET_DTYPE) if edgelet_cap > 0 else None
        cc, ec = np.zeros((n, n_levels), np.int32), np.zeros((n, n_levels), np.int32)
        self._check(self.lib.hso_gpu_detect_candidates_multi(self.h, ids, n, n_levels, _ptr(th), _ptr(co), corner_cap, _ptr(cc),
                                                             _ptr(eo), edgelet_cap, _ptr(ec)), "detect_candidates_multi")
        return co, cc, eo, ec

    def detect_candidates_init(self, frame_ids, n_levels=3, min_thresh=20, corner_cap=8192, fill_cap=4800):
        """fastDetectMT + fillingHole (the initialisation branch of FeatureExtractor::detect).
        Returns (corners[n, L, cap], corner_counts[n, L], fill[n, cap], fill_counts[n])."""
        n = len(frame_ids)
        ids = (C.c_int64 * n)(*frame_ids)
        co = np.zeros((n, n_levels, corner_cap), CORNER_DTYPE) if corner_cap > 0 else None
        fo = np.zeros((n, fill_cap), CORNER_DTYPE) if fill_cap > 0 else None
        cc, fc = np.zeros((n, n_levels), np.int32), np.zeros(n, np.int32)
        self._check(self.lib.hso_gpu_detect_candidates_init(self.h, ids, n, n_levels, min_thresh, _ptr(co), corner_cap, _ptr(cc),
                                                            _ptr(fo), fill_cap, _ptr(fc)), "detect_candidates_init")
        return co, cc, fo, fc

    def seed_activate(self, cam, seeds, targets_per_seed, n_mean_converge_frame=6, want_matches=False):
        """targets_per_seed: one list of ActivateTarget per seed (optFrames_P + optFrames_A order)."""
        begin = np.zeros(len(seeds) + 1, np.int32)
        begin[1:] = np.cumsum([len(t) for t in targets_per_seed])
        flat = [t for ts in targets_per_seed for t in ts]
        sarr = (Seed * len(seeds))(*seeds)
        tarr = (ActivateTarget * max(len(flat), 1))(*flat)
        out = (ActivateOut * len(seeds))()
        mo = (AlignOut * max(len(flat), 1))() if want_matches else None
        self._check(self.lib.hso_gpu_seed_activate(self.h, C.byref(cam), sarr, len(seeds),
                                                   begin.ctypes.data_as(C.POINTER(C.c_int32)), tarr,
                                                   n_mean_converge_frame, out, mo), "seed_activate")
        if want_matches:
            return list(out), [list(mo[begin[i]:begin[i + 1]]) for i in range(len(seeds))]
        return list(out)

    def seed_activate_multi(self, cam, seeds, targets_per_seed, n_mean_converge_frame):
        """The converged seeds of many sequences in one call; n_mean_converge_frame: one value per seed."""
        begin = np.zeros(len(seeds) + 1, np.int32)
        begin[1:] = np.cumsum([len(t) for t in targets_per_seed])
        flat = [t for ts in targets_per_seed for t in ts]
        sarr = (Seed * len(seeds))(*seeds)
        tarr = (ActivateTarget * max(len(flat), 1))(*flat)
        nm = np.ascontiguousarray(n_mean_converge_frame, np.int32)
        assert len(nm) == len(seeds)
        out = (ActivateOut * len(seeds))()
        self._check(self.lib.hso_gpu_seed_activate_multi(self.h, C.byref(cam), sarr, len(seeds),
                                                         begin.ctypes.data_as(C.POINTER(C.c_int32)), tarr,
                                                         nm.ctypes.data_as(C.POINTER(C.c_int32)), out, None), "seed_activate_multi")
        return list(out)

    def _activate_tables(self, targets_per_seed, frames, n_mean_converge_frame, n):
        begin = np.zeros(n + 1, np.int32)
        begin[1:] = np.cumsum([len(t) for t in targets_per_seed])
        pair = np.ascontiguousarray([q for ts in targets_per_seed for q in ts] or [0], np.int32)
        farr = (ActivateTarget * max(len(frames), 1))(*frames)
        nm = np.ascontiguousarray(n_mean_converge_frame, np.int32)
        assert len(nm) == n and len(targets_per_seed) == n
        return begin, pair, farr, nm

    def seed_activate_frames(self, cam, seeds, targets_per_seed, frames, n_mean_converge_frame):
        """targets_per_seed[i]: indices into `frames` (the unique target frames of the call)."""
        begin, pair, farr, nm = self._activate_tables(targets_per_seed, frames, n_mean_converge_frame, len(seeds))
        sarr = (Seed * len(seeds))(*seeds)
        out = (ActivateOut * len(seeds))()
        i32p = C.POINTER(C.c_int32)
        self._check(self.lib.hso_gpu_seed_activate_frames(self.h, C.byref(cam), sarr, len(seeds), begin.ctypes.data_as(i32p), pair.ctypes.data_as(i32p),
                                                          farr, len(frames), nm.ctypes.data_as(i32p), out), "seed_activate_frames")
        return list(out)

    def seed_table_activate(self, cam, table, slots, targets_per_seed, frames, n_mean_converge_frame):
        """The same for seeds named by their slots in a resident seed table."""
        sl = np.ascontiguousarray(slots, np.int32)
        begin, pair, farr, nm = self._activate_tables(targets_per_seed, frames, n_mean_converge_frame, len(sl))
        out = (ActivateOut * len(sl))()
        i32p = C.POINTER(C.c_int32)
        self._check(self.lib.hso_gpu_seed_table_activate(self.h, C.byref(cam), table, sl.ctypes.data_as(i32p), len(sl), begin.ctypes.data_as(i32p),
                                                         pair.ctypes.data_as(i32p), farr, len(frames), nm.ctypes.data_as(i32p), out), "seed_table_activate")
        return list(out)

    def tracker_eval(self, cam, params, job, level, T, exposure_rat, huber=-1.0, outlier=-1.0,
                     want_cache=False, want_errors=False):
        out = EvalOut()
        pa = C.c_int()
        self.lib.hso_gpu_tracker_pattern(params.max_level, level, C.byref(pa), None, None)
        n = job.n_feats
        ref_patch = np.zeros((n, pa.value), np.float32) if want_cache else None
        visible = np.zeros(n, np.uint8) if want_cache else None
        errs = np.zeros(n * pa.value, np.float32) if want_errors else None
        self._check(self.lib.hso_gpu_tracker_eval(self.h, C.byref(cam), C.byref(params), C.byref(job), level,
                                                  C.byref(T), exposure_rat, huber, outlier, C.byref(out),
                                                  _ptr(ref_patch), _ptr(visible), _ptr(errs)), "tracker_eval")
        return out, ref_patch, visible, errs
