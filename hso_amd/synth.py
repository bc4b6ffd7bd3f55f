"""Synthetic scenes for parity tests and bench.py.

A band-limited analytic texture is draped over a smooth depth surface defined
in frame 0; any other frame is rendered by inverse warping through a known
SE(3), an exposure ratio and Gaussian noise.  Everything is numpy; nothing here
touches the GPU or the oracle.
"""
import numpy as np

from .capi import REF_FEAT_DTYPE, CAM_PINHOLE, make_camera

# test/cameras/icl-nuim.txt of the reference: pinhole 481.2 480 319.5 239.5, 640x480
ICL_NUIM = dict(model=CAM_PINHOLE, width=640, height=480, fx=481.2, fy=480.0, cx=319.5, cy=239.5)
# test/cameras/euroc.txt: pinhole + radtan, 752x480
EUROC = dict(model=CAM_PINHOLE, width=752, height=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375,
             d=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0))


def camera(spec=ICL_NUIM):
    return make_camera(**spec)


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
        self.d = np.array(list(spec.get("d", (0, 0, 0, 0, 0))), float)
        self.distortion = abs(self.d[0]) > 1e-7
        rng = np.random.default_rng(seed)
        om = rng.uniform(0.02, 0.6, n_waves)
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
        if self.distortion:
            d = self.d
            r2 = u * u + v * v
            cd = 1 + d[0] * r2 + d[1] * r2 * r2 + d[4] * r2 * r2 * r2
            a1, a2, a3 = 2 * u * v, r2 + 2 * u * u, r2 + 2 * v * v
            u, v = u * cd + d[2] * a1 + d[3] * a2, v * cd + d[2] * a3 + d[3] * a1
        return self.fx * u + self.cx, self.fy * v + self.cy

    def unproject(self, px, py):
        x, y = (px - self.cx) / self.fx, (py - self.cy) / self.fy
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
