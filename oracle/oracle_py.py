"""ctypes wrapper over oracle/libhso_oracle.so — the CPU restatement.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing under hso_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hso_amd.capi import (Camera, SE3, FrameStats, TrackParams, TrackResult, EvalOut,
                          REF_FEAT_DTYPE, DEPTH_REF_IN_DTYPE, N_PYR_LEVELS)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhso_oracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "librobust_cost_ref.so")

_lib = None


def build(force=False):
    if force:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"])
    else:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])      # incremental: a no-op when up to date
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


PORTABLE_FLAGS = "gcc -O3 -std=gnu11 -ffp-contract=off -fno-fast-math (portable x86-64, the parity build)"


def use_native_build():
    """bench.py's cpu_baseline leg only: rebuild the restatement for THIS host with the reference's
    own optimisation level (-O3 -march=native, CMakeLists.txt:24-41; contraction left to the compiler)
    into oracle/_native/ and make it the library the wrappers below call.  Returns the flags in use
    (the portable parity build stays in use if the compiler is unavailable)."""
    global _lib, LIB_PATH
    import glob
    out_dir = os.path.join(_HERE, "_native")
    out = os.path.join(out_dir, "libhso_oracle_native.so")
    flags = ["-O3", "-march=native", "-std=gnu11", "-fPIC", "-shared"]
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc"] + flags + ["-o", out] + sorted(glob.glob(os.path.join(_HERE, "hso_oracle_*.c"))) + ["-lm", "-lpthread"],
                              stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return PORTABLE_FLAGS
    LIB_PATH, _lib = out, None
    load()
    return "gcc " + " ".join(flags[:3]) + " (rebuilt on the timing host)"


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    P = C.POINTER
    lib.hso_or_se3_mul.argtypes = [P(SE3), P(SE3), P(SE3)]
    lib.hso_or_se3_inverse.argtypes = [P(SE3), P(SE3)]
    lib.hso_or_se3_apply.argtypes = [P(SE3), vp, vp]
    lib.hso_or_se3_exp.argtypes = [vp, P(SE3)]
    lib.hso_or_se3_log.argtypes = [P(SE3), vp]
    lib.hso_or_so3_matrix.argtypes = [vp, vp]
    lib.hso_or_ldlt_solve.argtypes = [vp, vp, i32, vp]
    lib.hso_or_median_f.argtypes = [vp, i32]
    lib.hso_or_median_f.restype = C.c_float
    lib.hso_or_world2cam.argtypes = [P(Camera), vp, vp]
    lib.hso_or_cam2world.argtypes = [P(Camera), C.c_double, C.c_double, vp]
    lib.hso_or_error_multiplier2.argtypes = [P(Camera)]
    lib.hso_or_error_multiplier2.restype = C.c_double
    lib.hso_or_half_sample.argtypes = [vp, i32, i32, vp]
    lib.hso_or_create_pyramid.argtypes = [vp, i32, i32, P(vp)]
    lib.hso_or_sobel5.argtypes = [vp, i32, i32, vp, vp]
    lib.hso_or_frame_stats.argtypes = [vp, vp, vp, i32, i32, P(FrameStats)]
    lib.hso_or_make_depth_ref.argtypes = [vp, i32, vp, P(SE3), vp]
    lib.hso_or_tracker_create.argtypes = [P(Camera), P(TrackParams), P(vp), P(vp), i32, i32, vp, i32]
    lib.hso_or_tracker_create.restype = vp
    lib.hso_or_tracker_destroy.argtypes = [vp]
    lib.hso_or_tracker_destroy.restype = None
    lib.hso_or_tracker_set_level.argtypes = [vp, i32]
    lib.hso_or_tracker_set_level.restype = None
    lib.hso_or_tracker_select.argtypes = [vp, P(SE3), C.c_float, P(C.c_float), P(C.c_float), vp]
    lib.hso_or_tracker_set_thresholds.argtypes = [vp, C.c_float, C.c_float]
    lib.hso_or_tracker_set_thresholds.restype = None
    lib.hso_or_tracker_eval.argtypes = [vp, P(SE3), C.c_float, P(EvalOut)]
    lib.hso_or_tracker_eval.restype = None
    lib.hso_or_tracker_energy_f64.argtypes = [vp]
    lib.hso_or_tracker_energy_f64.restype = C.c_double
    lib.hso_or_tracker_decide_on_f64_sum.argtypes = [vp, C.c_int]
    lib.hso_or_tracker_decide_on_f64_sum.restype = None
    lib.hso_or_tracker_get_cache.argtypes = [vp, vp, vp, P(i32)]
    lib.hso_or_tracker_get_cache.restype = None
    lib.hso_or_tracker_run.argtypes = [vp, P(SE3), C.c_float, P(TrackResult)]
    lib.hso_or_tracker_run.restype = None
    lib.hso_or_tracker_pattern.argtypes = [i32, i32, P(i32), P(i32), vp]
    for name in ("hso_or_huber_weight", "hso_or_tukey_weight", "hso_or_tdist_weight"):
        getattr(lib, name).argtypes = [C.c_float, C.c_float]
        getattr(lib, name).restype = C.c_float
    lib.hso_or_mad_scale.argtypes = [vp, i32]
    lib.hso_or_mad_scale.restype = C.c_float
    lib.hso_or_tdist_scale.argtypes = [C.c_float, vp, i32]
    lib.hso_or_tdist_scale.restype = C.c_float
    _lib = lib
    return lib


def load_ref():
    """The compiled reference robust_cost.cpp (oracle/_ref), or None if absent."""
    if not os.path.exists(REF_LIB_PATH):
        return None
    lib = C.CDLL(REF_LIB_PATH)
    for name in ("ref_huber_weight", "ref_tukey_weight", "ref_tdist_weight"):
        getattr(lib, name).argtypes = [C.c_float, C.c_float]
        getattr(lib, name).restype = C.c_float
    lib.ref_mad_scale.argtypes = [C.c_void_p, C.c_int]
    lib.ref_mad_scale.restype = C.c_float
    lib.ref_tdist_scale.argtypes = [C.c_float, C.c_void_p, C.c_int]
    lib.ref_tdist_scale.restype = C.c_float
    lib.ref_huber_default_k.restype = C.c_float
    lib.ref_tukey_default_b.restype = C.c_float
    return lib


# ---------------------------------------------------------------- SE3 helpers
def se3_mul(a, b):
    o = SE3(); load().hso_or_se3_mul(C.byref(a), C.byref(b), C.byref(o)); return o


def se3_inverse(a):
    o = SE3(); load().hso_or_se3_inverse(C.byref(a), C.byref(o)); return o


def se3_exp(v):
    v = np.ascontiguousarray(v, np.float64)
    o = SE3(); load().hso_or_se3_exp(_ptr(v), C.byref(o)); return o


def se3_log(T):
    o = np.zeros(6); load().hso_or_se3_log(C.byref(T), _ptr(o)); return o


def se3_apply(T, p):
    p = np.ascontiguousarray(p, np.float64)
    o = np.zeros(3); load().hso_or_se3_apply(C.byref(T), _ptr(p), _ptr(o)); return o


def so3_matrix(q):
    q = np.ascontiguousarray(q, np.float64)
    R = np.zeros(9); load().hso_or_so3_matrix(_ptr(q), _ptr(R)); return R.reshape(3, 3)


def ldlt_solve(A, b):
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
    x = np.zeros(len(b)); load().hso_or_ldlt_solve(_ptr(A), _ptr(b), len(b), _ptr(x)); return x


def world2cam(cam, xyz):
    xyz = np.ascontiguousarray(xyz, np.float64)
    px = np.zeros(2); load().hso_or_world2cam(C.byref(cam), _ptr(xyz), _ptr(px)); return px


def cam2world(cam, u, v):
    f = np.zeros(3); load().hso_or_cam2world(C.byref(cam), float(u), float(v), _ptr(f)); return f


# -------------------------------------------------------------------- frames
def _padded_zeros(h, w):
    """(h, w) view of a buffer with two extra zero rows: the reference's dy taps read
    row `rows` of a level for features on the bottom border (src/CoarseTracker.cpp:370,491,
    a heap over-read there); oracle and GPU both define those bytes as 0."""
    return np.zeros((h + 2, w), np.uint8)[:h]


def _padded_copy(a):
    a = np.asarray(a, np.uint8)
    out = _padded_zeros(*a.shape)
    out[:] = a
    return out


def create_pyramid(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    levels = [_padded_zeros(*pyramid_dims(w, h, i)[::-1]) for i in range(N_PYR_LEVELS)]
    ptrs = (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in levels])
    rc = load().hso_or_create_pyramid(_ptr(img), w, h, ptrs)
    if rc != 0:
        raise ValueError("pyramid: failed")
    return levels


def pyramid_dims(w, h, level):
    """(width, height) of a pyramid level: w >> l for sizes that are multiples of 16, else the
    cvRound sizes of the cv::resize branch (frame.cpp:302-312)."""
    lw, lh = C.c_int(), C.c_int()
    load().hso_or_pyramid_dims(int(w), int(h), int(level), C.byref(lw), C.byref(lh))
    return lw.value, lh.value


def resize_linear(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    load().hso_or_resize_linear_8u(_ptr(img), img.shape[1], img.shape[0], _ptr(out), dw, dh)
    return out


def half_sample(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), np.uint8)
    load().hso_or_half_sample(_ptr(img), w, h, _ptr(out))
    return out


def sobel5(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    gx = np.zeros((h, w), np.int16); gy = np.zeros((h, w), np.int16)
    load().hso_or_sobel5(_ptr(img), w, h, _ptr(gx), _ptr(gy))
    return gx, gy


def frame_stats(img0, gx0, gy0):
    st = FrameStats()
    h, w = img0.shape
    load().hso_or_frame_stats(_ptr(np.ascontiguousarray(img0)), _ptr(np.ascontiguousarray(gx0)),
                              _ptr(np.ascontiguousarray(gy0)), w, h, C.byref(st))
    return st


def make_depth_ref(din, poses, T_ref_w):
    din = np.ascontiguousarray(din, dtype=DEPTH_REF_IN_DTYPE)
    arr = (SE3 * len(poses))(*poses)
    out = np.zeros(len(din))
    load().hso_or_make_depth_ref(_ptr(din), len(din), C.cast(arr, C.c_void_p), C.byref(T_ref_w), _ptr(out))
    return out


# ------------------------------------------------------------------- tracker
class Tracker:
    def __init__(self, cam, params, ref_pyr, cur_pyr, feats):
        self.lib = load()
        self.ref_pyr = [_padded_copy(l) for l in ref_pyr]
        self.cur_pyr = [_padded_copy(l) for l in cur_pyr]
        self.feats = np.ascontiguousarray(feats, dtype=REF_FEAT_DTYPE)
        self.params = params
        h, w = self.ref_pyr[0].shape
        rp = (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in self.ref_pyr])
        cp = (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in self.cur_pyr])
        self.h = self.lib.hso_or_tracker_create(C.byref(cam), C.byref(params), rp, cp, w, h,
                                                _ptr(self.feats), len(self.feats))
        self.n = len(self.feats)

    def __del__(self):
        try:
            if self.h:
                self.lib.hso_or_tracker_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_level(self, level):
        self.lib.hso_or_tracker_set_level(self.h, level)

    def select(self, T, a, want_errors=False):
        hu, ou = C.c_float(), C.c_float()
        errs = np.zeros(self.n * 25, np.float32) if want_errors else None
        n = self.lib.hso_or_tracker_select(self.h, C.byref(T), a, C.byref(hu), C.byref(ou), _ptr(errs))
        return n, hu.value, ou.value, (errs[:n] if want_errors else None)

    def set_thresholds(self, huber, outlier):
        self.lib.hso_or_tracker_set_thresholds(self.h, huber, outlier)

    def eval(self, T, a):
        out = EvalOut()
        self.lib.hso_or_tracker_eval(self.h, C.byref(T), a, C.byref(out))
        return out

    def energy_f64(self):
        return self.lib.hso_or_tracker_energy_f64(self.h)

    def decide_on_f64_sum(self, on=True):
        """Diagnostics (not the reference's behaviour): accept decisions on the fp64 sum of the same fp32 energy terms."""
        self.lib.hso_or_tracker_decide_on_f64_sum(self.h, 1 if on else 0)

    def cache(self):
        pa = C.c_int()
        self.lib.hso_or_tracker_get_cache(self.h, None, None, C.byref(pa))
        rp = np.zeros((self.n, pa.value), np.float32)
        vis = np.zeros(self.n, np.uint8)
        self.lib.hso_or_tracker_get_cache(self.h, _ptr(rp), _ptr(vis), C.byref(pa))
        return rp, vis

    def run(self, T_init, a_init):
        res = TrackResult()
        self.lib.hso_or_tracker_run(self.h, C.byref(T_init), a_init, C.byref(res))
        return res


def find_match_direct(cam, job, ref_pyr, cur_pyr, cur_sobel):
    """Matcher::findMatchDirect (after the reference observation was chosen) on host arrays.
    ref_pyr/cur_pyr: 5 level images; cur_sobel: [(gx, gy)] for levels 0..2."""
    from hso_amd.capi import AlignJob, AlignOut
    lib = load()
    lib.hso_or_find_match_direct.argtypes = [C.POINTER(Camera), C.POINTER(AlignJob), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                             C.c_int, C.c_int, C.POINTER(AlignOut)]
    lib.hso_or_find_match_direct.restype = None
    rp = [np.ascontiguousarray(l) for l in ref_pyr]
    cp = [np.ascontiguousarray(l) for l in cur_pyr]
    gx = [np.ascontiguousarray(g[0]) for g in cur_sobel]
    gy = [np.ascontiguousarray(g[1]) for g in cur_sobel]
    h, w = rp[0].shape
    out = AlignOut()
    lib.hso_or_find_match_direct(C.byref(cam), C.byref(job),
                                 (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in rp]),
                                 (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in cp]),
                                 (C.c_void_p * 3)(*[g.ctypes.data for g in gx]),
                                 (C.c_void_p * 3)(*[g.ctypes.data for g in gy]), w, h, C.byref(out))
    return out


def _pyr_ptrs(pyr, n):
    arrs = [np.ascontiguousarray(l) for l in pyr]
    return arrs, (C.c_void_p * n)(*[a.ctypes.data for a in arrs])


def find_match_direct_batch(cam, jobs, job_kf, kf_pyrs, cur_pyr, cur_sobel):
    """bench.py's cpu_baseline leg: hso_or_find_match_direct over a ctypes array of AlignJob (job i against keyframe
    job_kf[i]'s pyramid) in one call (no per-item interpreter time).  Returns the AlignOut array."""
    from hso_amd.capi import AlignJob, AlignOut
    lib = load()
    vp = C.c_void_p
    lib.hso_or_find_match_direct_batch.argtypes = [C.POINTER(Camera), vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp]
    lib.hso_or_find_match_direct_batch.restype = None
    keep, flat = [], []
    for p in kf_pyrs:
        a, _ = _pyr_ptrs(p, N_PYR_LEVELS)
        keep.append(a); flat += [x.ctypes.data for x in a]
    kfp = (C.c_void_p * len(flat))(*flat)
    ca, cp = _pyr_ptrs(cur_pyr, N_PYR_LEVELS)
    gxa, gx = _pyr_ptrs([g[0] for g in cur_sobel], 3)
    gya, gy = _pyr_ptrs([g[1] for g in cur_sobel], 3)
    n = len(jobs)
    h, w = ca[0].shape
    idx = np.ascontiguousarray(job_kf, np.int32)
    out = (AlignOut * max(n, 1))()
    lib.hso_or_find_match_direct_batch(C.byref(cam), C.cast(jobs, vp), _ptr(idx), n, C.cast(kfp, vp), C.cast(cp, vp), C.cast(gx, vp),
                                       C.cast(gy, vp), w, h, C.cast(out, vp))
    return out


def seed_observe_batch(cam, seeds, cur_T_f_w, cur_exposure, px_error_angle, ref_pyr, cur_pyr, cur_sobel, n_threads=1):
    """bench.py's cpu_baseline leg: hso_or_seed_observe over a ctypes array of Seed hosted in ONE frame, in one call, on
    n_threads threads (4 = the reference's depth-filter workers, include/hso/IndexThreadReduce.h:27)."""
    from hso_amd.capi import SeedOut
    lib = load()
    vp = C.c_void_p
    lib.hso_or_seed_observe_batch.argtypes = [C.POINTER(Camera), vp, C.c_int, C.POINTER(SE3), C.c_double, C.c_double, vp, vp, vp, vp,
                                              C.c_int, C.c_int, vp, C.c_int]
    lib.hso_or_seed_observe_batch.restype = None
    ra, rp = _pyr_ptrs(ref_pyr, N_PYR_LEVELS)
    ca, cp = _pyr_ptrs(cur_pyr, N_PYR_LEVELS)
    gxa, gx = _pyr_ptrs([g[0] for g in cur_sobel], 3)
    gya, gy = _pyr_ptrs([g[1] for g in cur_sobel], 3)
    n = len(seeds)
    h, w = ca[0].shape
    out = (SeedOut * max(n, 1))()
    lib.hso_or_seed_observe_batch(C.byref(cam), C.cast(seeds, vp), n, C.byref(cur_T_f_w), cur_exposure, px_error_angle, C.cast(rp, vp),
                                  C.cast(cp, vp), C.cast(gx, vp), C.cast(gy, vp), w, h, C.cast(out, vp), n_threads)
    return out


def pose_optimize(cam, job):
    """optimizeLevenbergMarquardt3rd on a hso_amd.capi.PoseJob; returns (PoseResult, outlier mask)."""
    from hso_amd.capi import PoseJob, PoseResult
    lib = load()
    lib.hso_or_pose_optimize.argtypes = [C.POINTER(Camera), C.POINTER(PoseJob), C.POINTER(PoseResult), C.c_void_p]
    lib.hso_or_pose_optimize.restype = None
    res = PoseResult()
    mask = np.zeros(max(job.n_feats, 1), np.uint8)
    lib.hso_or_pose_optimize(C.byref(cam), C.byref(job), C.byref(res), _ptr(mask))
    return res, mask[:job.n_feats]


def ba_linearize(poses, fixed, idist, edges, huber_corner, huber_edge):
    """g2o-style build of the robustified normal equations (see hso_oracle_ba.c)."""
    from hso_amd.capi import BA_EDGE_DTYPE, ba_alloc
    lib = load()
    lib.hso_or_ba_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_double, C.c_double] + [C.c_void_p] * 8
    lib.hso_or_ba_linearize.restype = None
    parr = (SE3 * len(poses))(*poses)
    fixed = np.ascontiguousarray(fixed, np.uint8)
    idist = np.ascontiguousarray(idist, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    o = ba_alloc(len(poses), len(idist), len(edges))
    lib.hso_or_ba_linearize(C.cast(parr, C.c_void_p), _ptr(fixed), len(poses), _ptr(idist), len(idist), _ptr(edges),
                            len(edges), huber_corner, huber_edge, _ptr(o["Hpp"]), _ptr(o["bp"]), _ptr(o["Hpc"]),
                            _ptr(o["Hcc"]), _ptr(o["bc"]), _ptr(o["edge_err"]), _ptr(o["edge_chi2"]), _ptr(o["chi2_sum"]))
    return o


def seed_observe(cam, seed, cur_T_f_w, cur_exposure, px_error_angle, ref_pyr, cur_pyr, cur_sobel):
    """DepthFilter::observeDepthRow for one seed on host arrays."""
    from hso_amd.capi import Seed, SeedOut
    lib = load()
    lib.hso_or_seed_observe.argtypes = [C.POINTER(Camera), C.POINTER(Seed), C.POINTER(SE3), C.c_double, C.c_double,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(SeedOut)]
    lib.hso_or_seed_observe.restype = None
    rp = [np.ascontiguousarray(l) for l in ref_pyr]
    cp = [np.ascontiguousarray(l) for l in cur_pyr]
    gx = [np.ascontiguousarray(g[0]) for g in cur_sobel]
    gy = [np.ascontiguousarray(g[1]) for g in cur_sobel]
    h, w = rp[0].shape
    out = SeedOut()
    lib.hso_or_seed_observe(C.byref(cam), C.byref(seed), C.byref(cur_T_f_w), cur_exposure, px_error_angle,
                            (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in rp]),
                            (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in cp]),
                            (C.c_void_p * 3)(*[g.ctypes.data for g in gx]),
                            (C.c_void_p * 3)(*[g.ctypes.data for g in gy]), w, h, C.byref(out))
    return out


def seed_observe_previous(cam, seed, pre_T_f_w, pre_exposure, px_error_angle, ref_pyr, pre_pyr, pre_sobel):
    """DepthFilter::observeDepthWithPreviousFrameOnce for one seed and one earlier frame, on host arrays."""
    from hso_amd.capi import Seed, SeedOut
    lib = load()
    lib.hso_or_seed_observe_previous.argtypes = [C.POINTER(Camera), C.POINTER(Seed), C.POINTER(SE3), C.c_double, C.c_double,
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(SeedOut)]
    lib.hso_or_seed_observe_previous.restype = None
    rp = [np.ascontiguousarray(l) for l in ref_pyr]
    cp = [np.ascontiguousarray(l) for l in pre_pyr]
    gx = [np.ascontiguousarray(g[0]) for g in pre_sobel]
    gy = [np.ascontiguousarray(g[1]) for g in pre_sobel]
    h, w = rp[0].shape
    out = SeedOut()
    lib.hso_or_seed_observe_previous(C.byref(cam), C.byref(seed), C.byref(pre_T_f_w), pre_exposure, px_error_angle,
                                     (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in rp]),
                                     (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in cp]),
                                     (C.c_void_p * 3)(*[g.ctypes.data for g in gx]),
                                     (C.c_void_p * 3)(*[g.ctypes.data for g in gy]), w, h, C.byref(out))
    return out


def seed_activate(cam, seed, targets, ref_pyr, tgt_pyrs, tgt_sobels, n_mean_converge_frame=6):
    """DepthFilter::activatePoint for one seed.  targets: list of ActivateTarget; tgt_pyrs[i]: the
    5 levels of target i; tgt_sobels[i]: [(gx, gy)] * 3.  Returns (ActivateOut, [AlignOut])."""
    from hso_amd.capi import Seed, ActivateTarget, ActivateOut, AlignOut
    lib = load()
    vpp = C.POINTER(C.c_void_p)
    lib.hso_or_seed_activate.argtypes = [C.POINTER(Camera), C.POINTER(Seed), C.POINTER(ActivateTarget), C.c_int, vpp, vpp, vpp,
                                         vpp, C.c_int, C.c_int, C.c_int, C.POINTER(ActivateOut), C.POINTER(AlignOut)]
    lib.hso_or_seed_activate.restype = None
    n = len(targets)
    rp = [np.ascontiguousarray(l) for l in ref_pyr]
    h, w = rp[0].shape
    keep = [rp]
    pyr_ptrs, gx_ptrs, gy_ptrs = [], [], []
    for i in range(n):
        tp = [np.ascontiguousarray(l) for l in tgt_pyrs[i]]
        gx = [np.ascontiguousarray(g[0]) for g in tgt_sobels[i]]
        gy = [np.ascontiguousarray(g[1]) for g in tgt_sobels[i]]
        keep += [tp, gx, gy]
        pyr_ptrs += [l.ctypes.data for l in tp]
        gx_ptrs += [g.ctypes.data for g in gx]
        gy_ptrs += [g.ctypes.data for g in gy]
    tarr = (ActivateTarget * max(n, 1))(*targets)
    out = ActivateOut()
    mo = (AlignOut * max(n, 1))()
    lib.hso_or_seed_activate(C.byref(cam), C.byref(seed), tarr, n,
                             (C.c_void_p * N_PYR_LEVELS)(*[l.ctypes.data for l in rp]),
                             (C.c_void_p * max(5 * n, 1))(*pyr_ptrs), (C.c_void_p * max(3 * n, 1))(*gx_ptrs),
                             (C.c_void_p * max(3 * n, 1))(*gy_ptrs), w, h, n_mean_converge_frame, C.byref(out), mo)
    return out, list(mo[:n])


CORNER_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("score", "<i4"), ("response", "<f4")])
REF_FAST_PATH = os.path.join(_HERE, "_ref", "libfast_ref.so")


def fast_detect_level(img, threshold, border=8, cap=None):
    """FeatureExtractor::fastDetect for one level -> structured array of surviving corners."""
    lib = load()
    lib.hso_or_fast_detect_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.hso_or_fast_detect_level.restype = C.c_int
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = cap or w * h
    out = np.zeros(cap, CORNER_DTYPE)
    n = lib.hso_or_fast_detect_level(img.ctypes.data, w, h, threshold, border, out.ctypes.data, cap)
    return out[:min(n, cap)].copy(), n


def fast9_detect(img, threshold):
    """All FAST-9 corners (raster order) and their scores, before non-max suppression."""
    lib = load()
    lib.hso_or_fast9_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.hso_or_fast9_detect.restype = C.c_int
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xy = np.zeros((w * h, 2), np.int16); sc = np.zeros(w * h, np.int32)
    n = lib.hso_or_fast9_detect(img.ctypes.data, w, h, threshold, xy.ctypes.data, sc.ctypes.data, w * h)
    return xy[:n].copy(), sc[:n].copy()


def fast_detect_arc(img, threshold, arc):
    """All FAST-`arc` corners and their scores in raster order (arc = 9 or 12)."""
    lib = load()
    lib.hso_or_fast_detect_arc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.hso_or_fast_detect_arc.restype = C.c_int
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xy = np.zeros((w * h, 2), np.int16); sc = np.zeros(w * h, np.int32)
    n = lib.hso_or_fast_detect_arc(img.ctypes.data, w, h, threshold, arc, xy.ctypes.data, sc.ctypes.data, w * h)
    return xy[:n].copy(), sc[:n].copy()


def filling_hole_level(img, level, frame_w, frame_h, min_thresh, have):
    """FeatureExtractor::fillingHole of one level; `have` is updated in place."""
    lib = load()
    lib.hso_or_filling_hole_level.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int]
    lib.hso_or_filling_hole_level.restype = C.c_int
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(len(have), CORNER_DTYPE)
    n = lib.hso_or_filling_hole_level(img.ctypes.data, w, h, level, frame_w, frame_h, int(min_thresh), have.ctypes.data,
                                      out.ctypes.data, len(out))
    return out[:n].copy()


def ref_fast12(img, threshold):
    """FAST-12 of the compiled reference library: corners, scores, fast_nonmax_3x3 survivors; None if absent."""
    if not os.path.exists(REF_FAST_PATH):
        return None
    lib = C.CDLL(REF_FAST_PATH)
    if not hasattr(lib, "ref_fast12_detect"):
        return None
    lib.ref_fast12_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.ref_fast12_score.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.ref_fast_nonmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xy = np.zeros((w * h, 2), np.int16)
    n = lib.ref_fast12_detect(img.ctypes.data, w, h, w, threshold, xy.ctypes.data, w * h)
    xy = xy[:n].copy()
    sc = np.zeros(max(n, 1), np.int32)
    lib.ref_fast12_score(img.ctypes.data, w, xy.ctypes.data, n, threshold, sc.ctypes.data)
    keep = np.zeros(max(n, 1), np.int32)
    nk = lib.ref_fast_nonmax(xy.ctypes.data, sc.ctypes.data, n, keep.ctypes.data) if n else 0
    return xy, sc[:n].copy(), keep[:nk].copy()


def shi_tomasi(img, u, v):
    lib = load()
    lib.hso_or_shi_tomasi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.hso_or_shi_tomasi.restype = C.c_float
    img = np.ascontiguousarray(img, np.uint8)
    return lib.hso_or_shi_tomasi(img.ctypes.data, img.shape[1], img.shape[0], int(u), int(v))


def ref_fast(img, threshold):
    """The compiled reference FAST library (oracle/_ref/libfast_ref.so): corners, scores and the
    indices fast_nonmax_3x3 keeps; None if the library is absent (GPU box)."""
    if not os.path.exists(REF_FAST_PATH):
        return None
    lib = C.CDLL(REF_FAST_PATH)
    lib.ref_fast9_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    lib.ref_fast9_score.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.ref_fast_nonmax.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    xy = np.zeros((w * h, 2), np.int16)
    n = lib.ref_fast9_detect(img.ctypes.data, w, h, w, threshold, xy.ctypes.data, w * h)
    xy = xy[:n].copy()
    sc = np.zeros(max(n, 1), np.int32)
    lib.ref_fast9_score(img.ctypes.data, w, xy.ctypes.data, n, threshold, sc.ctypes.data)
    keep = np.zeros(max(n, 1), np.int32)
    nk = lib.ref_fast_nonmax(xy.ctypes.data, sc.ctypes.data, n, keep.ctypes.data) if n else 0
    return xy, sc[:n].copy(), keep[:nk].copy()


EDGELET_DTYPE = np.dtype([("x", "<i2"), ("y", "<i2"), ("gx", "<i2"), ("gy", "<i2"), ("grad", "<f4")])


def canny_l2(gx, gy, low, high):
    """cv::Canny(dx, dy, edges, low, high, L2gradient=True) restated -> u8 edge map (0 / 255)."""
    lib = load()
    lib.hso_or_canny_l2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
    lib.hso_or_canny_l2.restype = None
    gx, gy = np.ascontiguousarray(gx, np.int16), np.ascontiguousarray(gy, np.int16)
    h, w = gx.shape
    out = np.zeros((h, w), np.uint8)
    lib.hso_or_canny_l2(gx.ctypes.data, gy.ctypes.data, w, h, float(low), float(high), out.ctypes.data)
    return out


def detect_grid(width, height, level):
    """-> (grid, gcols, grows) of FeatureExtractor's occupancy grid on `level`."""
    lib = load()
    v = [C.c_int() for _ in range(5)]
    lib.hso_or_detect_grid.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 5
    lib.hso_or_detect_grid.restype = None
    lib.hso_or_detect_grid(width, height, level, *[C.byref(x) for x in v])
    return v[0].value, v[1].value, v[2].value


def cell_index(x, y, grid, gcols, grows):
    lib = load()
    lib.hso_or_detect_cell_index.argtypes = [C.c_int] * 5
    lib.hso_or_detect_cell_index.restype = C.c_int
    return lib.hso_or_detect_cell_index(int(x), int(y), grid, gcols, grows)


def edgelet_level(gx, gy, level, frame_w, frame_h, min_thresh, have):
    """edgeLetDetectST of one level; `have` (uint8 flags per grid index) is updated in place."""
    lib = load()
    lib.hso_or_edgelet_level.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int]
    lib.hso_or_edgelet_level.restype = C.c_int
    gx, gy = np.ascontiguousarray(gx, np.int16), np.ascontiguousarray(gy, np.int16)
    h, w = gx.shape
    out = np.zeros(len(have), EDGELET_DTYPE)
    n = lib.hso_or_edgelet_level(gx.ctypes.data, gy.ctypes.data, w, h, level, frame_w, frame_h, int(min_thresh),
                                 have.ctypes.data, out.ctypes.data, len(out))
    return out[:n].copy()


def detect_candidates_level(img, gx, gy, level, frame_w, frame_h, min_thresh):
    """fastDetectST + edgeLetDetectST of one level -> (corners, edgelets, have)."""
    corners, _ = fast_detect_level(img, min_thresh, 8)
    g, gc, gr = detect_grid(frame_w, frame_h, level)
    have = np.zeros(gc * gr, np.uint8)
    for c in corners:
        have[cell_index(c["x"], c["y"], g, gc, gr)] = 1
    return corners, edgelet_level(gx, gy, level, frame_w, frame_h, min_thresh, have), have


def reproject_match(cam, T_cur_w, cur_exposure_time, cur_keyframe_id, kfs, points, obs, cell_size, grid_n_cols,
                    kf_pyrs, cur_pyr, cur_sobel, margins_out=None):
    """Reprojector::reprojectPoint + getCloseViewObs + findMatchDirect per map point.
    kf_pyrs[k]: pyramid of keyframe k.  Returns (proj array, list of AlignOut or None per point); margins_out (a list)
    receives the decision margins of every point's findMatchDirect (None where none ran)."""
    from hso_amd.capi import AlignJob, KF_DTYPE, MAP_POINT_DTYPE, OBS_DTYPE, REPROJ_POINT_DTYPE
    lib = load()
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    lib.hso_or_reproject_point.argtypes = [C.POINTER(Camera), C.POINTER(SE3), vp, vp, dbl, i32, i32, vp, C.POINTER(i32)]
    lib.hso_or_reproject_point.restype = i32
    lib.hso_or_close_view_obs.argtypes = [vp, vp, vp, vp, i32]
    lib.hso_or_close_view_obs.restype = i32
    lib.hso_or_reproject_make_job.argtypes = [C.POINTER(SE3), dbl, i32, vp, vp, vp, vp, C.POINTER(AlignJob)]
    lib.hso_or_reproject_make_job.restype = None
    kfs = np.ascontiguousarray(kfs, KF_DTYPE); points = np.ascontiguousarray(points, MAP_POINT_DTYPE)
    obs = np.ascontiguousarray(obs, OBS_DTYPE)
    cur_pos = np.array(se3_inverse(T_cur_w).t[:])
    proj = np.zeros(len(points), REPROJ_POINT_DTYPE)
    proj["ref_obs"] = -1
    matches = []
    for i, p in enumerate(points):
        px = np.zeros(2); cell = i32(0)
        T_host = kfs[p["host_kf"]:p["host_kf"] + 1]          # q, t follow frame_id: an hso_se3 in place
        ok = lib.hso_or_reproject_point(C.byref(cam), C.byref(T_cur_w), T_host.ctypes.data + 8, p["host_f"].ctypes.data,
                                        float(p["idist"]), cell_size, grid_n_cols, px.ctypes.data, C.byref(cell))
        m = mg = None
        if ok:
            proj[i]["projected"], proj[i]["cell"], proj[i]["px"] = 1, cell.value, px
            o = obs[p["obs_begin"]:p["obs_begin"] + p["obs_count"]]
            k = lib.hso_or_close_view_obs(cur_pos.ctypes.data, p["pos"].ctypes.data, kfs.ctypes.data, o.ctypes.data, len(o)) if len(o) else -1
            if k >= 0:
                proj[i]["ref_obs"] = p["obs_begin"] + k
                job = AlignJob()
                lib.hso_or_reproject_make_job(C.byref(T_cur_w), cur_exposure_time, cur_keyframe_id, kfs.ctypes.data,
                                              points[i:i + 1].ctypes.data, o[k:k + 1].ctypes.data, px.ctypes.data, C.byref(job))
                margins_reset()
                m = find_match_direct(cam, job, kf_pyrs[o[k]["kf"]], cur_pyr, cur_sobel)
                mg = margins()
        matches.append(m)
        if margins_out is not None:
            margins_out.append(mg if m is not None else None)
    return proj, matches


REF_PATCH_SCORE_PATH = os.path.join(_HERE, "_ref", "libpatch_score_ref.so")


def zmncc_f8(host, target):
    lib = load()
    lib.hso_or_zmncc_f8.argtypes = [C.c_void_p, C.c_void_p]
    lib.hso_or_zmncc_f8.restype = C.c_float
    host, target = np.ascontiguousarray(host, np.float32), np.ascontiguousarray(target, np.float32)
    return lib.hso_or_zmncc_f8(host.ctypes.data, target.ctypes.data)


def ref_zmncc_f8(host, target):
    """ZMNCC_F<4> of the compiled reference header (oracle/_ref/libpatch_score_ref.so); None if absent."""
    if not os.path.exists(REF_PATCH_SCORE_PATH):
        return None
    lib = C.CDLL(REF_PATCH_SCORE_PATH)
    lib.ref_zmncc_f8.argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_zmncc_f8.restype = C.c_float
    host, target = np.ascontiguousarray(host, np.float32), np.ascontiguousarray(target, np.float32)
    return lib.ref_zmncc_f8(host.ctypes.data, target.ctypes.data)


def pattern(max_level, level):
    pa, hp = C.c_int(), C.c_int()
    offs = np.zeros((40, 2), np.int8)
    rc = load().hso_or_tracker_pattern(max_level, level, C.byref(pa), C.byref(hp), _ptr(offs))
    return rc, pa.value, hp.value, offs[:pa.value].copy()


def ba_huber_deltas(poses, idist, edges, obs_uv, error_multiplier2):
    from hso_amd.capi import BA_EDGE_DTYPE
    lib = load()
    lib.hso_or_ba_huber_deltas.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                           C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.hso_or_ba_huber_deltas.restype = None
    parr = (SE3 * len(poses))(*poses)
    idist = np.ascontiguousarray(idist, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    obs_uv = np.ascontiguousarray(obs_uv, np.float64)
    hc, he = C.c_float(), C.c_float()
    lib.hso_or_ba_huber_deltas(C.cast(parr, C.c_void_p), len(poses), _ptr(idist), len(idist), _ptr(edges), _ptr(obs_uv), len(edges),
                               error_multiplier2, C.byref(hc), C.byref(he))
    return hc.value, he.value


def ba_optimize(poses, fixed, idist, edges, huber_corner, huber_edge, n_iter):
    """g2o's LM over the local-BA graph (hso_oracle_ba.c) -> (poses, idist, edge_chi2, BaResult)."""
    from hso_amd.capi import BA_EDGE_DTYPE, BaResult
    lib = load()
    lib.hso_or_ba_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                       C.c_double, C.c_int, C.c_void_p, C.POINTER(BaResult)]
    lib.hso_or_ba_optimize.restype = None
    parr = (SE3 * len(poses))(*poses)
    fixed = np.ascontiguousarray(fixed, np.uint8)
    idist = np.array(idist, np.float64)
    edges = np.ascontiguousarray(edges, BA_EDGE_DTYPE)
    chi2 = np.zeros(len(edges))
    res = BaResult()
    lib.hso_or_ba_optimize(C.cast(parr, C.c_void_p), _ptr(fixed), len(poses), _ptr(idist), len(idist), _ptr(edges), len(edges),
                           huber_corner, huber_edge, n_iter, _ptr(chi2), C.byref(res))
    return list(parr), idist, chi2, res


def se3quat_exp(update):
    lib = load()
    lib.hso_or_se3quat_exp.argtypes = [C.c_void_p, C.POINTER(SE3)]
    lib.hso_or_se3quat_exp.restype = None
    u = np.ascontiguousarray(update, np.float64)
    o = SE3(); lib.hso_or_se3quat_exp(_ptr(u), C.byref(o)); return o


def se3quat_mul(a, b):
    lib = load()
    lib.hso_or_se3quat_mul.argtypes = [C.POINTER(SE3)] * 3
    lib.hso_or_se3quat_mul.restype = None
    o = SE3(); lib.hso_or_se3quat_mul(C.byref(a), C.byref(b), C.byref(o)); return o


MARGIN_FIELDS = ("lk_update", "lk_chi2", "ncc", "normal", "jump", "zmncc_best", "zmncc_ambig", "zmncc_order", "klt_energy",
                 "klt_step", "klt_accept", "pose_rho", "march_end", "track_accept")


class Margins(C.Structure):
    _fields_ = [(n, C.c_double) for n in MARGIN_FIELDS]


def margins_reset():
    load().hso_or_margins_reset()


def margins():
    """The smallest distance of every gating comparison to its threshold since margins_reset() (hso_oracle.h)."""
    lib = load()
    lib.hso_or_margins_get.argtypes = [C.POINTER(Margins)]
    m = Margins()
    lib.hso_or_margins_get(C.byref(m))
    return m


# ---- two-view initialisation, image side (hso_oracle_klt.c) ----
def pyr_down(img):
    """cv::pyrDown (8-bit, BORDER_REFLECT_101)."""
    lib = load()
    lib.hso_or_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib.hso_or_pyr_down(img.ctypes.data, w, h, out.ctypes.data)
    return out


def scharr_deriv(img):
    """calcSharrDeriv -> (h, w, 2) int16 (Ix, Iy)."""
    lib = load()
    lib.hso_or_scharr_deriv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib.hso_or_scharr_deriv(img.ctypes.data, w, h, out.ctypes.data)
    return out


def klt_levels(w, h, win=30, max_level=4):
    lib = load()
    lib.hso_or_klt_levels.restype = C.c_int
    return lib.hso_or_klt_levels(C.c_int(w), C.c_int(h), C.c_int(win), C.c_int(max_level))


def klt_track(prev, cur, px_prev, px_init, win=30, max_level=4, max_count=30, epsilon=1e-4, use_initial_flow=True):
    """initialization::trackKlt's cv::calcOpticalFlowPyrLK call -> (px_cur (n, 2) float32, status (n,) uint8, margin (n,) float32)."""
    lib = load()
    lib.hso_or_klt_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_double, C.c_int, C.c_void_p]
    prev = np.ascontiguousarray(prev, np.uint8); cur = np.ascontiguousarray(cur, np.uint8)
    h, w = prev.shape
    a = np.ascontiguousarray(px_prev, np.float32).reshape(-1, 2)
    b = np.array(px_init, np.float32).reshape(-1, 2).copy()
    n = len(a)
    st = np.zeros(n, np.uint8); mg = np.zeros(n, np.float32)
    lib.hso_or_klt_track(prev.ctypes.data, cur.ctypes.data, w, h, a.ctypes.data, b.ctypes.data, st.ctypes.data, n, win, max_level, max_count,
                         epsilon, 1 if use_initial_flow else 0, mg.ctypes.data)
    return b, st, mg


def patch_check(img_pre, img_cur, px_pre, px_cur):
    """initialization::patchCheck -> (ok, ncc); ncc = -2 when a patch leaves the image."""
    lib = load()
    lib.hso_or_patch_check.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hso_or_patch_check.restype = C.c_int
    img_pre = np.ascontiguousarray(img_pre, np.uint8); img_cur = np.ascontiguousarray(img_cur, np.uint8)
    h, w = img_pre.shape
    a = np.ascontiguousarray(px_pre, np.float32); b = np.ascontiguousarray(px_cur, np.float32)
    ncc = C.c_float(0)
    ok = lib.hso_or_patch_check(img_pre.ctypes.data, img_cur.ctypes.data, w, h, a.ctypes.data, b.ctypes.data, C.byref(ncc))
    return bool(ok), ncc.value
