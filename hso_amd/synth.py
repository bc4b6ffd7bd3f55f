"""Synthetic scenes for parity tests and bench.py.

A band-limited analytic texture is draped over a smooth depth surface defined
in frame 0; any other frame is rendered by inverse warping through a known
SE(3), an exposure ratio and Gaussian noise.  Everything is numpy; nothing here
touches the GPU or the oracle.
"""
import numpy as np

from .capi import REF_FEAT_DTYPE, CAM_PINHOLE, CAM_FOV, make_camera

# test/cameras/icl-nuim.txt of the reference: pinhole 481.2 480 319.5 239.5, 640x480
ICL_NUIM = dict(model=CAM_PINHOLE, width=640, height=480, fx=481.2, fy=480.0, cx=319.5, cy=239.5)
# test/cameras/euroc.txt: pinhole + radtan, 752x480
EUROC = dict(model=CAM_PINHOLE, width=752, height=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375,
             d=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0))


# test/cameras/tum_mono_vo_wide.txt at the reference's internal 920x736 (test/test_dataset.cpp:218-224).  Its third line is
# "true": the image is rectified first (cv::remap, outside the hot path) and the FOV camera then projects without distortion
# (src/camera.cpp:171-221, undistort_ branch).
TUM_WIDE = dict(model=CAM_FOV, width=920, height=736, fx=0.349153 * 920, fy=0.436593 * 736, cx=0.49314 * 920, cy=0.499021 * 736,
                d=(0.933271,), distortion=0)
# the same sensor geometry with the FOV distortion left in the projection (third line "false").  omega is reduced so that
# dist * omega < pi / 2 up to the image corners (with the file's 0.933 the model's tan() turns over inside the 920x736 frame,
# which is why the reference rectifies that camera)
FOV_920 = dict(TUM_WIDE, d=(0.6,), distortion=1)


def camera(spec=ICL_NUIM):
    return make_camera(**{k: v for k, v in spec.items() if k != "texture_om"})    # texture_om describes the scene, not the camera


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rotvec_to_quat(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.array([0.5 * rv[0], 0.5 * rv[1], 0.5 * rv[2], 1.0])
    s = np.sin(th / 2) / th
    return np.array([rv[0] * s, rv[1] * s, rv[2] * s, np.cos(th / 2)])


class Scene:
    """Texture + depth surface in frame 0, pinhole (optionally radtan) camera."""

    def __init__(self, spec=ICL_NUIM, seed=1234, n_waves=32):
        self.spec = dict(spec)
        self.w, self.h = spec["width"], spec["height"]
        self.fx, self.fy, self.cx, self.cy = spec["fx"], spec["fy"], spec["cx"], spec["cy"]
        self.d = np.array(list(spec.get("d", (0, 0, 0, 0, 0))) + [0.0] * 5, float)[:5]
        self.model = spec.get("model", CAM_PINHOLE)
        self.distortion = bool(spec.get("distortion", 1)) if self.model == CAM_FOV else abs(self.d[0]) > 1e-7
        rng = np.random.default_rng(seed)
        om = rng.uniform(0.02, 0.6, n_waves)
        if "texture_om" in spec:
            # optional bands of spatial frequency (rad / px), the waves split evenly among them: a scene with the low-frequency
            # content natural images have (the default band has none below a 314 px wavelength; a large misprediction of the
            # tracker's start, as after the two-view initialisation, then has no coarse level to recover on)
            bands = list(spec["texture_om"])
            om = np.concatenate([rng.uniform(lo, hi, len(part)) for (lo, hi), part in zip(bands, np.array_split(np.arange(n_waves), len(bands)))])
        ang = rng.uniform(0, 2 * np.pi, n_waves)
        self.wx, self.wy = om * np.cos(ang), om * np.sin(ang)
        self.ph = rng.uniform(0, 2 * np.pi, n_waves)
        self.amp = 45.0 * np.sqrt(2.0 / n_waves)
        rng_d = np.random.default_rng(seed + 1)
        self.dwx = rng_d.uniform(-0.012, 0.012, 4)
        self.dwy = rng_d.uniform(-0.012, 0.012, 4)
        self.dph = rng_d.uniform(0, 2 * np.pi, 4)

    # analytic texture / depth of frame 0 at real-valued pixel coordinates
    def texture(self, x, y):
        acc = np.zeros_like(x, dtype=np.float64)
        for k in range(len(self.wx)):
            acc += np.sin(self.wx[k] * x + self.wy[k] * y + self.ph[k])
        return 128.0 + self.amp * acc

    def depth0(self, x, y):
        acc = np.zeros_like(x, dtype=np.float64)
        for k in range(4):
            acc += np.sin(self.dwx[k] * x + self.dwy[k] * y + self.dph[k])
        return 4.0 + 0.5 * acc  # z in [2, 6]

    # camera (src/camera.cpp:94-125 formulas, vectorised for data generation only)
    def project(self, X):
        u, v = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
        if self.model == CAM_FOV:
            if self.distortion:                      # FOVCamera::world2cam, src/camera.cpp:196-221
                om = self.d[0]
                dist = np.sqrt(u * u + v * v)
                ratio = np.where(dist > 1e-12, np.arctan(2 * dist * np.tan(om / 2)) / (np.maximum(dist, 1e-12) * om), 1.0)
                u, v = ratio * u, ratio * v
            return self.fx * u + self.cx, self.fy * v + self.cy
        if self.distortion:
            d = self.d
            r2 = u * u + v * v
            cd = 1 + d[0] * r2 + d[1] * r2 * r2 + d[4] * r2 * r2 * r2
            a1, a2, a3 = 2 * u * v, r2 + 2 * u * u, r2 + 2 * v * v
            u, v = u * cd + d[2] * a1 + d[3] * a2, v * cd + d[2] * a3 + d[3] * a1
        return self.fx * u + self.cx, self.fy * v + self.cy

    def unproject(self, px, py):
        x, y = (px - self.cx) / self.fx, (py - self.cy) / self.fy
        if self.model == CAM_FOV:
            if self.distortion:                      # FOVCamera::cam2world, src/camera.cpp:171-194
                om = self.d[0]
                dist = np.sqrt(x * x + y * y)
                rd = np.where(dist > 1e-12, np.tan(np.maximum(dist, 1e-12) * om) / (2 * np.maximum(dist, 1e-12) * np.tan(om / 2)), 1.0)
                x, y = rd * x, rd * y
            return np.stack([x, y, np.ones_like(x)], -1)
        if self.distortion:
            d = self.d
            x0, y0 = x, y
            for _ in range(20):
                r2 = x * x + y * y
                icd = 1.0 / (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2)
                dx = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
                dy = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
                x, y = (x0 - dx) * icd, (y0 - dy) * icd
        return np.stack([x, y, np.ones_like(x)], -1)

    def points0(self, px, py):
        """3-D points (frame 0) seen at frame-0 pixels."""
        return self.unproject(px, py) * self.depth0(px, py)[..., None]

    def warp_from0(self, q, t, px, py):
        X = self.points0(px, py) @ quat_to_R(q).T + np.asarray(t)
        return self.project(X), X

    def pixels0_of(self, q, t, px, py, iters=8):
        """Frame-0 pixels whose surface points project to (px, py) in frame (q,t)."""
        x0, y0 = px.astype(np.float64).copy(), py.astype(np.float64).copy()
        for _ in range(iters):
            (u, v), _ = self.warp_from0(q, t, x0, y0)
            x0 -= (u - px)
            y0 -= (v - py)
        return x0, y0

    def render(self, q, t, exposure=1.0, noise_sigma=0.0, seed=0):
        ys, xs = np.mgrid[0:self.h, 0:self.w].astype(np.float64)
        if np.allclose(q, [0, 0, 0, 1]) and np.allclose(t, 0):
            x0, y0 = xs, ys
        else:
            x0, y0 = self.pixels0_of(q, t, xs, ys)
        img = exposure * self.texture(x0, y0)
        if noise_sigma > 0:
            img = img + np.random.default_rng(seed).normal(0, noise_sigma, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def features(self, q, t, n, seed=1237, margin=16, frac_invalid=0.0):
        """n reference features in frame (q,t): px uniform in the margin box, unit
        bearing f = normalize(K^-1 px) and dist = |X| (CoarseTracker.cpp:219-235)."""
        rng = np.random.default_rng(seed)
        px = rng.uniform(margin, self.w - margin, n)
        py = rng.uniform(margin, self.h - margin, n)
        x0, y0 = self.pixels0_of(q, t, px, py)
        _, X = self.warp_from0(q, t, x0, y0)
        feats = np.zeros(n, REF_FEAT_DTYPE)
        feats["px"][:, 0], feats["px"][:, 1] = px, py
        b = self.unproject(px, py)
        feats["f"] = b / np.linalg.norm(b, axis=1, keepdims=True)
        feats["dist"] = np.linalg.norm(X, axis=1)
        if frac_invalid > 0:
            bad = rng.uniform(size=n) < frac_invalid
            feats["dist"][bad] = -1.0
        return feats


def align_jobs(pair, n, ref_frame_id, seed=7, px_noise=1.5, edgelet_frac=0.3, gx=None, gy=None,
               pose_noise=0.0, exposure_rat=1.0, kf_gap_lt4=1):
    """Reprojection candidates for Matcher::findMatchDirect: reference features of `pair`
    (levels 0-2, corners and edgelets), their depth along the bearing, the pair's true
    T_cur_ref (optionally perturbed) and a noisy projected start position."""
    from .capi import AlignJob, SE3, FTR_CORNER, FTR_EDGELET
    sc = pair["scene"]
    rng = np.random.default_rng(seed)
    feats = sc.features(np.array([0, 0, 0, 1.0]), np.zeros(3), n, seed=seed + 1, margin=40)
    q, t = np.array(pair["q_true"], float), np.array(pair["t_true"], float)
    if pose_noise > 0:
        t = t + rng.normal(0, pose_noise, 3)
    R = quat_to_R(q)
    jobs = []
    for i in range(n):
        j = AlignJob()
        j.ref_frame_id = ref_frame_id
        j.ref_level = int(rng.integers(0, 3))
        px = feats["px"][i]
        j.px_ref[:] = [float(px[0]), float(px[1])]
        j.f_ref[:] = [float(v) for v in feats["f"][i]]
        j.depth = float(feats["dist"][i])
        is_edge = rng.uniform() < edgelet_frac and gx is not None
        j.type = FTR_EDGELET if is_edge else FTR_CORNER
        if is_edge:
            g = np.array([gx[int(px[1]), int(px[0])], gy[int(px[1]), int(px[0])]], float)
            g = g / (np.linalg.norm(g) + 1e-9)
            j.grad[:] = [float(g[0]), float(g[1])]
        else:
            j.grad[:] = [1.0, 0.0]
        j.T_cur_ref = SE3.from_arrays(q, t)
        X = R @ (feats["f"][i] * feats["dist"][i]) + t
        u, v = sc.project(X[None, :])
        j.px_cur[:] = [float(u[0] + rng.normal(0, px_noise)), float(v[0] + rng.normal(0, px_noise))]
        j.exposure_rat = float(exposure_rat)
        j.kf_gap_lt4 = int(kf_gap_lt4)
        jobs.append(j)
    return jobs


def pose_problem(n_feats=200, n_hosts=5, seed=3, px_noise=0.4, outlier_frac=0.05, edgelet_frac=0.3,
                 temp_frac=0.1, nopoint_frac=0.1, pose_err=(0.02, 0.01), spec=ICL_NUIM):
    """A frame observing points hosted in `n_hosts` keyframes (inverse-depth parameterisation),
    as pose_optimizer::optimizeLevenbergMarquardt3rd sees it: returns (feats, host poses,
    initial T_f_w, true T_f_w)."""
    from .capi import POSE_FEAT_DTYPE, SE3, FTR_CORNER, FTR_EDGELET
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, w, h = spec["fx"], spec["fy"], spec["cx"], spec["cy"], spec["width"], spec["height"]

    def rand_pose(sr, st):
        return rotvec_to_quat(rng.normal(0, sr, 3)), rng.normal(0, st, 3)
    hosts = [rand_pose(0.05, 0.3) for _ in range(n_hosts)]
    q_true, t_true = rand_pose(0.05, 0.3)
    R_true = quat_to_R(q_true)
    feats = np.zeros(n_feats, POSE_FEAT_DTYPE)
    for i in range(n_feats):
        hidx = int(rng.integers(0, n_hosts))
        qh, th = hosts[hidx]
        Rh = quat_to_R(qh)
        # a point in front of both cameras: sample in the host frame, express in world
        upx = rng.uniform(30, w - 30), rng.uniform(30, h - 30)
        bear = np.array([(upx[0] - cx) / fx, (upx[1] - cy) / fy, 1.0]); bear /= np.linalg.norm(bear)
        dist = rng.uniform(2, 8)
        Xw = Rh.T @ (bear * dist - th)
        Xc = R_true @ Xw + t_true
        u = fx * Xc[0] / Xc[2] + cx + rng.normal(0, px_noise)
        v = fy * Xc[1] / Xc[2] + cy + rng.normal(0, px_noise)
        if rng.uniform() < outlier_frac:
            u += rng.normal(0, 25); v += rng.normal(0, 25)
        fo = np.array([(u - cx) / fx, (v - cy) / fy, 1.0]); fo /= np.linalg.norm(fo)
        feats["has_point"][i] = 0 if rng.uniform() < nopoint_frac else 1
        feats["type"][i] = FTR_EDGELET if rng.uniform() < edgelet_frac else FTR_CORNER
        feats["level"][i] = int(rng.integers(0, 3))
        feats["temporary"][i] = 1 if rng.uniform() < temp_frac else 0
        feats["host_pose"][i] = hidx
        feats["f"][i] = fo
        g = rng.normal(size=2); feats["grad"][i] = g / np.linalg.norm(g)
        feats["host_f"][i] = bear
        feats["idist"][i] = 1.0 / dist
    q0 = rotvec_to_quat(rng.normal(0, pose_err[1], 3))
    # initial guess = true pose perturbed on the left
    R0 = quat_to_R(q0)
    t0 = R0 @ t_true + rng.normal(0, pose_err[0], 3)
    # quaternion of R0 * R_true
    def qmul(a, b):
        ax, ay, az, aw = a; bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                         aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])
    q_init = qmul(q0, q_true)
    poses = [SE3.from_arrays(q, t) for q, t in hosts]
    return feats, poses, SE3.from_arrays(q_init, t0), SE3.from_arrays(q_true, t_true)


def ba_problem(n_poses=9, n_points=300, obs_per_point=4, seed=9, px_noise=0.5, edgelet_frac=0.3,
               n_fixed=2, spec=ICL_NUIM):
    """A local-BA graph as ba::LocalBundleAdjustment builds it: keyframe poses (some fixed),
    inverse-depth points hosted in one keyframe, one edge per non-host observation."""
    from .capi import BA_EDGE_DTYPE, SE3, FTR_CORNER, FTR_EDGELET
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, w, h = spec["fx"], spec["fy"], spec["cx"], spec["cy"], spec["width"], spec["height"]
    poses = [(rotvec_to_quat(rng.normal(0, 0.04, 3)), rng.normal(0, 0.25, 3)) for _ in range(n_poses)]
    fixed = np.zeros(n_poses, np.uint8); fixed[:n_fixed] = 1
    idist = np.zeros(n_points)
    edges = []
    for p in range(n_points):
        host = int(rng.integers(0, n_poses))
        qh, th = poses[host]; Rh = quat_to_R(qh)
        upx = rng.uniform(40, w - 40), rng.uniform(40, h - 40)
        bear = np.array([(upx[0] - cx) / fx, (upx[1] - cy) / fy, 1.0]); bear /= np.linalg.norm(bear)
        dist = rng.uniform(2, 8)
        idist[p] = (1.0 / dist) * (1 + rng.normal(0, 0.03))         # perturbed estimate
        Xw = Rh.T @ (bear * dist - th)
        others = [k for k in range(n_poses) if k != host]
        for tgt in rng.choice(others, size=min(obs_per_point, len(others)), replace=False):
            qt, tt = poses[int(tgt)]
            Xc = quat_to_R(qt) @ Xw + tt
            uv = np.array([Xc[0] / Xc[2] + rng.normal(0, px_noise / fx), Xc[1] / Xc[2] + rng.normal(0, px_noise / fy)])
            e = np.zeros((), BA_EDGE_DTYPE)
            e["point"], e["host"], e["target"] = p, host, int(tgt)
            e["level"] = int(rng.integers(0, 3))
            e["fH"] = bear
            if rng.uniform() < edgelet_frac:
                g = rng.normal(size=2); g /= np.linalg.norm(g)
                e["type"] = FTR_EDGELET
                e["normal"] = g
                e["meas"] = [g @ uv, 0.0]
            else:
                e["type"] = FTR_CORNER
                e["normal"] = [1.0, 0.0]
                e["meas"] = uv
            edges.append(e)
    return [SE3.from_arrays(q, t) for q, t in poses], fixed, idist, np.array(edges, BA_EDGE_DTYPE)


def seeds_for_pair(pair, n, ref_frame_id, seed=17, depth_noise=0.15, edgelet_frac=0.3, gx=None, gy=None,
                   depth_mean=4.0, depth_min=2.0):
    """Depth-filter seeds hosted in the reference frame of `pair` (identity pose), initialised the
    way Seed::Seed does (mu = 1/depth_mean-ish, z_range = 1/depth_min, sigma2 = z_range^2/36;
    src/depth_filter.cpp:49-68) with a noisy inverse depth, plus the active frame's pose."""
    from .capi import Seed, SE3, FTR_CORNER, FTR_EDGELET
    sc = pair["scene"]
    rng = np.random.default_rng(seed)
    feats = sc.features(np.array([0, 0, 0, 1.0]), np.zeros(3), n, seed=seed + 1, margin=40)
    seeds = []
    for i in range(n):
        s = Seed()
        s.ref_frame_id = ref_frame_id
        s.level = int(rng.integers(0, 3))
        px = feats["px"][i]
        is_edge = rng.uniform() < edgelet_frac and gx is not None
        s.type = FTR_EDGELET if is_edge else FTR_CORNER
        s.px[:] = [float(px[0]), float(px[1])]
        s.f[:] = [float(v) for v in feats["f"][i]]
        if is_edge:
            g = np.array([gx[int(px[1]), int(px[0])], gy[int(px[1]), int(px[0])]], float)
            g = g / (np.linalg.norm(g) + 1e-9)
            s.grad[:] = [float(g[0]), float(g[1])]
        else:
            s.grad[:] = [1.0, 0.0]
        s.T_ref_w = SE3.identity()
        s.ref_exposure = 1.0
        true_idist = 1.0 / feats["dist"][i]
        s.mu = float(true_idist * (1 + rng.normal(0, depth_noise)))
        z_range = 1.0 / depth_min
        s.sigma2 = float(z_range * z_range / 36)
        s.b = 10.0
        seeds.append(s)
    T_cur = SE3.from_arrays(pair["q_true"], pair["t_true"])
    return seeds, T_cur, feats


def activation_problem(n_seeds=120, n_targets=8, seed=31, spec=ICL_NUIM, depth_noise=0.03, edgelet_frac=0.3,
                       trans_frac=0.04, rot_deg=0.6, noise=1.0, host_frame_id=9200):
    """A host frame (identity pose) plus n_targets frames at small random motions, and seeds hosted
    in the host frame whose inverse depth is close to the truth (converged seeds, the state
    DepthFilter::activatePoint sees).  Every seed lists all target frames; a few seeds get
    truncated lists, wrong depths or a far-away target to exercise the gates."""
    from .capi import Seed, SE3, ActivateTarget, FTR_CORNER, FTR_EDGELET
    sc = Scene(spec, seed)
    rng = np.random.default_rng(seed + 1)
    qi = np.array([0, 0, 0, 1.0]); ti = np.zeros(3)
    host = sc.render(qi, ti, 1.0, noise, seed + 2)
    frames, targets = [], []
    for k in range(n_targets):
        tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
        rdir = rng.normal(size=3); rdir /= np.linalg.norm(rdir)
        t = tdir * trans_frac * 4.0 * rng.uniform(0.4, 1.0)
        q = rotvec_to_quat(rdir * np.deg2rad(rot_deg) * rng.uniform(0.3, 1.0))
        expo = float(rng.uniform(0.9, 1.1)) if k != 2 else 1.4     # one frame triggers the exposure compensation
        frames.append(sc.render(q, t, expo, noise, seed + 10 + k))
        tg = ActivateTarget()
        tg.frame_id = host_frame_id + 1 + k
        tg.T_f_w = SE3.from_arrays(q, t)
        tg.exposure = expo
        targets.append(tg)
    feats = sc.features(qi, ti, n_seeds, seed=seed + 3, margin=48)
    gy0, gx0 = np.gradient(host.astype(np.float64))
    seeds, per_seed = [], []
    for i in range(n_seeds):
        s = Seed()
        s.ref_frame_id = host_frame_id
        s.level = int(rng.integers(0, 3))
        px = feats["px"][i]
        is_edge = rng.uniform() < edgelet_frac
        s.type = FTR_EDGELET if is_edge else FTR_CORNER
        s.px[:] = [float(px[0]), float(px[1])]
        s.f[:] = [float(v) for v in feats["f"][i]]
        g = np.array([gx0[int(px[1]), int(px[0])], gy0[int(px[1]), int(px[0])]])
        g = g / (np.linalg.norm(g) + 1e-9)
        s.grad[:] = [float(g[0]), float(g[1])] if is_edge else [1.0, 0.0]
        s.T_ref_w = SE3.identity()
        s.ref_exposure = 1.0
        true_idist = 1.0 / feats["dist"][i]
        s.mu = float(true_idist * (1 + rng.normal(0, depth_noise)))
        s.sigma2 = 1e-4
        s.b = 10.0
        tl = list(targets)
        if i % 17 == 3:
            tl = tl[:2]                      # fewer targets than the frame threshold
        if i % 19 == 5:
            s.mu = float(true_idist * 1.6)   # wrong depth: large drift -> isValid = false
        if i % 23 == 7:
            tl = []
        seeds.append(s)
        per_seed.append(tl)
    return dict(scene=sc, host=host, frames=frames, targets=targets, seeds=seeds, per_seed=per_seed, feats=feats,
                host_frame_id=host_frame_id)


def config2_pair(n_feats=2000, spec=ICL_NUIM, seed=1234, exposure=1.05, noise=1.0,
                 trans_frac=0.02, rot_deg=0.5):
    """reference = frame 0, current = known SE(3) away."""
    sc = Scene(spec, seed)
    rng = np.random.default_rng(seed + 5)
    tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
    rdir = rng.normal(size=3); rdir /= np.linalg.norm(rdir)
    t_true = tdir * trans_frac * 4.0
    q_true = rotvec_to_quat(rdir * np.deg2rad(rot_deg))
    qi = np.array([0, 0, 0, 1.0]); ti = np.zeros(3)
    ref = sc.render(qi, ti, 1.0, noise, seed + 2)
    cur = sc.render(q_true, t_true, exposure, noise, seed + 3)
    feats = sc.features(qi, ti, n_feats, seed + 4)
    return dict(scene=sc, ref=ref, cur=cur, feats=feats, q_true=q_true, t_true=t_true,
                exposure=exposure)


def map_problem(n_points=1500, n_kfs=6, seed=71, spec=ICL_NUIM, noise=1.0, trans_frac=0.04, rot_deg=0.8,
                edgelet_frac=0.3, first_frame_id=9800, max_fts=200):
    """A small map the way Reprojector::reprojectMap sees it: n_kfs keyframes (kf 0 at the
    identity) plus one far-away keyframe whose viewing direction is useless (> 60 degrees), map
    points hosted in different keyframes with 1-4 keyframe observations each, and a current frame.
    Includes points that project outside the frame, behind the camera, or have no usable
    observation.  Returns the tables of hso_gpu_reproject_match plus the images."""
    from .capi import KF_DTYPE, OBS_DTYPE, MAP_POINT_DTYPE, SE3, FTR_CORNER, FTR_EDGELET
    sc = Scene(spec, seed)
    rng = np.random.default_rng(seed + 1)
    poses = [(np.array([0, 0, 0, 1.0]), np.zeros(3))]
    for k in range(1, n_kfs):
        tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
        rdir = rng.normal(size=3); rdir /= np.linalg.norm(rdir)
        poses.append((rotvec_to_quat(rdir * np.deg2rad(rot_deg) * rng.uniform(0.3, 1.0)), tdir * trans_frac * 4.0 * rng.uniform(0.4, 1.0)))
    frames = [sc.render(q, t, float(rng.uniform(0.9, 1.1)) if k else 1.0, noise, seed + 10 + k) for k, (q, t) in enumerate(poses)]
    kfs = np.zeros(n_kfs + 1, KF_DTYPE)
    for k, (q, t) in enumerate(poses):
        kfs[k]["frame_id"], kfs[k]["q"], kfs[k]["t"] = first_frame_id + k, q, t
        kfs[k]["exposure_time"], kfs[k]["keyframe_id"] = 1.0 + 0.02 * k, k
    # the far keyframe: 30 m away behind the surface, looking back at it (its image is never sampled: no point chooses it)
    far = n_kfs
    kfs[far]["frame_id"], kfs[far]["q"], kfs[far]["t"] = first_frame_id + far, [0, 0, 0, 1.0], [0, 0, -30.0]
    kfs[far]["exposure_time"], kfs[far]["keyframe_id"] = 1.0, far
    frames.append(frames[0])
    # current frame
    tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
    q_cur, t_cur = rotvec_to_quat(np.array([0.3, -0.8, 0.5]) * np.deg2rad(rot_deg)), tdir * trans_frac * 4.0
    cur = sc.render(q_cur, t_cur, 1.06, noise, seed + 3)
    feats = sc.features(poses[0][0], poses[0][1], n_points, seed=seed + 4, margin=12)
    X0 = feats["f"] * feats["dist"][:, None]                       # world = frame 0
    grads = [np.gradient(f.astype(np.float64)) for f in frames]
    points = np.zeros(n_points, MAP_POINT_DTYPE)
    obs = []
    w, h = sc.w, sc.h

    def observe(k, X):
        q, t = poses[k]
        Xk = quat_to_R(q) @ X + t
        u, v = sc.project(Xk[None])
        u, v = float(u[0]), float(v[0])
        b = sc.unproject(np.array([u]), np.array([v]))[0]
        return u, v, b / np.linalg.norm(b), Xk
    for i in range(n_points):
        X = X0[i].copy()
        if i % 29 == 4:
            X[0] += 30.0                                           # far outside the image
        if i % 31 == 6:
            X[2] = -1.0                                            # behind every camera
        host = int(rng.integers(0, n_kfs))
        u, v, f, Xh = observe(host, X)
        if not (10 <= u < w - 10 and 10 <= v < h - 10) or Xh[2] <= 0.1:
            host = 0
            u, v, f, Xh = observe(0, X0[i]) if i % 29 != 4 and i % 31 != 6 else (feats["px"][i][0], feats["px"][i][1], feats["f"][i], X0[i])
            if i % 29 == 4 or i % 31 == 6:
                # keep the bad geometry: the host feature is consistent with the displaced point
                Xh = X
                f = X / np.linalg.norm(X) if X[2] > 0 else np.array([0.0, 0.0, 1.0])
        points[i]["pos"] = X
        points[i]["idist"] = 1.0 / max(np.linalg.norm(Xh), 1e-3) if i % 31 != 6 else -0.5
        points[i]["host_f"], points[i]["host_kf"] = f, host
        ks = [host] + [int(k) for k in rng.permutation(n_kfs)[:int(rng.integers(0, 4))] if k != host]
        if i % 37 == 9:
            ks = [far]                                             # only a useless observation
        if i % 41 == 11:
            ks = []                                                # no observation at all
        if i % 13 == 2 and ks and ks != [far]:
            ks.insert(int(rng.integers(0, len(ks) + 1)), far)
        points[i]["obs_begin"], points[i]["obs_count"] = len(obs), len(ks)
        for k in ks:
            o = np.zeros(1, OBS_DTYPE)[0]
            kk = k if k != far else 0
            uu, vv, ff, _ = observe(kk, X0[i])
            uu, vv = float(np.clip(uu, 0, w - 1)), float(np.clip(vv, 0, h - 1))
            is_edge = rng.uniform() < edgelet_frac
            gy, gx = grads[kk]
            g = np.array([gx[int(vv), int(uu)], gy[int(vv), int(uu)]])
            g = g / (np.linalg.norm(g) + 1e-9)
            lvl = int(rng.integers(0, 3))
            o["kf"], o["level"], o["type"] = k, lvl, FTR_EDGELET if is_edge else FTR_CORNER
            o["px"] = [np.floor(uu / (1 << lvl)) * (1 << lvl), np.floor(vv / (1 << lvl)) * (1 << lvl)]   # detected on the level's lattice
            bb = sc.unproject(np.array([o["px"][0]]), np.array([o["px"][1]]))[0]
            o["f"] = bb / np.linalg.norm(bb)
            o["grad"] = g if is_edge else [1.0, 0.0]
            obs.append(o)
    obs = np.array(obs, OBS_DTYPE) if obs else np.zeros(0, OBS_DTYPE)
    cell_size = int(np.floor(np.float32(np.sqrt(np.float32(w * h) / max_fts)) * 0.6))    # reprojector.cpp:53-56
    return dict(scene=sc, frames=frames, cur=cur, kfs=kfs, points=points, obs=obs, T_cur_w=SE3.from_arrays(q_cur, t_cur),
                cur_exposure=1.06, cur_keyframe_id=n_kfs + 2, cell_size=cell_size, grid_n_cols=int(np.ceil(w / cell_size)),
                cur_frame_id=first_frame_id + n_kfs + 1)


def _render_frame(args):
    spec, seed, q, t, expo, noise, k = args
    return Scene(spec, seed).render(np.asarray(q), np.asarray(t), expo, noise, seed + 100 + k)


def sequence(n_frames=60, spec=EUROC, seed=2024, step=(0.016, 0.005, 0.007), rot_deg_per_frame=(0.05, -0.12, 0.03), noise=1.0,
             exposure_wobble=0.03, workers=None):
    """A synthetic image sequence over the analytic scene: frame k sees the scene from T_k_0 = (q_k, t_k), a smooth
    constant-velocity-like path with a slow sinusoidal modulation (so the motion model is good but not exact).
    Returns dict(images [n], T_f_w [(q, t)], exposure [n], depth0 (optical-axis depth image of frame 0), scene, spec)."""
    import multiprocessing as mp
    import os
    poses, expos = [], []
    for k in range(n_frames):
        s = k + 1.5 * np.sin(k / 9.0)                       # modulated arc length
        rv = np.deg2rad(np.array(rot_deg_per_frame)) * s
        poses.append((rotvec_to_quat(rv), np.array(step) * s))
        expos.append(1.0 + exposure_wobble * np.sin(k / 5.0))
    jobs = [(dict(spec), seed, list(q), list(t), expos[k], noise, k) for k, (q, t) in enumerate(poses)]
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 32))
    if workers == 1:
        images = [_render_frame(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            images = pool.map(_render_frame, jobs, chunksize=1)
    sc = Scene(spec, seed)
    ys, xs = np.mgrid[0:sc.h, 0:sc.w].astype(np.float64)
    depth0 = sc.points0(xs, ys)[..., 2].astype(np.float32)
    return dict(images=images, T_f_w=poses, exposure=expos, depth0=depth0, scene=sc, spec=dict(spec))


def sequences(n_seq, n_frames, spec=EUROC, seed0=2024, workers=None):
    """n_seq synthetic sequences (different scenes and motions) rendered through ONE worker pool: what the multi-sequence
    driver / bench feed to hso_vo_multi_*.  Returns a list of dicts like sequence()."""
    import multiprocessing as mp
    import os
    metas, jobs = [], []
    for q in range(n_seq):
        seed = seed0 + 17 * q
        step = (0.016 + 0.002 * (q % 4), 0.005, 0.007 - 0.001 * (q % 3))
        rot = (0.05, -0.12 + 0.01 * (q % 5), 0.03)
        poses, expos = [], []
        for k in range(n_frames):
            s_ = k + 1.5 * np.sin(k / 9.0)
            rv = np.deg2rad(np.array(rot)) * s_
            poses.append((rotvec_to_quat(rv), np.array(step) * s_))
            expos.append(1.0 + 0.03 * np.sin(k / 5.0))
        metas.append((seed, poses, expos))
        jobs += [(dict(spec), seed, list(q_), list(t_), expos[k], 1.0, k) for k, (q_, t_) in enumerate(poses)]
    workers = workers or max(1, min(len(jobs), (os.cpu_count() or 2) - 1, 32))
    if workers == 1:
        images = [_render_frame(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            images = pool.map(_render_frame, jobs, chunksize=1)
    out = []
    for q, (seed, poses, expos) in enumerate(metas):
        sc = Scene(spec, seed)
        ys, xs = np.mgrid[0:sc.h, 0:sc.w].astype(np.float64)
        depth0 = sc.points0(xs, ys)[..., 2].astype(np.float32)
        out.append(dict(images=images[q * n_frames:(q + 1) * n_frames], T_f_w=poses, exposure=expos, depth0=depth0, scene=sc, spec=dict(spec)))
    return out
