"""Image side of the two-view initialisation (initialization::trackKlt, reference src/initialization.cpp:225-300: cv::calcOpticalFlowPyrLK
with a 30x30 window, 4 levels, 30 iterations, eps 1e-4, OPTFLOW_USE_INITIAL_FLOW, then patchCheck :476-563).

CPU: the restatement (oracle/hso_oracle_klt.c; OpenCV itself is absent, "parity unpinned" in its header) checked through properties
that hold for the published algorithm: pyrDown of a constant / a linear ramp, Scharr of a ramp (the kernel sums to 32 per unit
slope), the level rule of buildOpticalFlowPyramid, recovery of a known sub-pixel translation, and the ground-truth flow of the
synthetic scene.
GPU: hso_gpu_klt_track / the two image kernels against the restatement: pyramid levels and derivative images bit-exact; per point
the status equal and the position within 2e-3 px (the device sums the window exactly in 64-bit integers, OpenCV's scalar path in
float; the fixed point of the iteration is the same, the stop tests differ by rounding) unless the restatement's own decision margin
(minimum-eigenvalue test, epsilon stop, oscillation stop) was within 1e-3 of its bound or a level ran into the iteration cap; the patch check's NCC within 1e-5 and its
decision equal unless |ncc - 0.8| < 1e-4."""
import numpy as np
import pytest

from hso_amd import capi, synth


@pytest.fixture(scope="module")
def klt_seq():
    return synth.sequence(n_frames=4, step=(0.06, 0.02, 0.015), workers=4)


def _points(img_shape, n, seed, border=12):
    h, w = img_shape
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(border, w - border, n), rng.uniform(border, h - border, n)], 1).astype(np.float32)


def test_pyr_down_and_scharr_properties(orc):
    const = np.full((37, 53), 77, np.uint8)
    assert (orc.pyr_down(const) == 77).all() and orc.pyr_down(const).shape == (19, 27)
    ramp = np.tile((np.arange(64, dtype=np.int32) * 3).astype(np.uint8), (40, 1))       # slope 3 in x, no wrap (max 189)
    d = orc.pyr_down(ramp)
    assert d.shape == (20, 32)
    assert (d[:, 1:-1] == (np.arange(1, 31) * 6)[None, :]).all()                       # interior: the value at 2x (the kernel is symmetric)
    s = orc.scharr_deriv(ramp)
    assert (s[1:-1, 1:-1, 0] == 3 * 32).all() and (s[..., 1] == 0).all()               # (3 + 10 + 3) * 2 per unit slope
    assert (s[:, 0, 0] == 0).all() and (s[:, -1, 0] == 0).all()                         # BORDER_REFLECT_101: the mirrored neighbours cancel
    assert (orc.scharr_deriv(ramp.T.copy())[1:-1, 1:-1, 1] == 3 * 32).all()


def test_klt_level_rule(orc):
    # buildOpticalFlowPyramid stops before a level that is not larger than the window in both dimensions
    assert orc.klt_levels(752, 480) == 3        # 376x240, 188x120, 94x60, (47x30: no)
    assert orc.klt_levels(640, 480) == 3
    assert orc.klt_levels(920, 736) == 4        # 58x46 is still larger than 30
    assert orc.klt_levels(1240, 376) == 3       # KITTI: 78x24 ends it
    assert orc.klt_levels(64, 64) == 1 and orc.klt_levels(60, 60, win=30, max_level=4) == 0


def test_oracle_klt_recovers_translation(orc):
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 255, (64, 80))
    big = np.kron(base, np.ones((8, 8)))                     # 512 x 640 blocks of 8
    from scipy.ndimage import gaussian_filter, shift
    big = gaussian_filter(big, 3.0)
    a = np.clip(big, 0, 255).astype(np.uint8)
    b = np.clip(shift(big, (2.25, -3.5), order=3, mode="nearest"), 0, 255).astype(np.uint8)   # content moves by (+2.25 rows, -3.5 cols)
    px = _points(a.shape, 200, 1, border=60)
    cur, st, _ = orc.klt_track(a, b, px, px)
    assert st.all()
    err = cur - px - np.array([-3.5, 2.25], np.float32)
    assert np.abs(err).max() < 0.08, np.abs(err).max()
    # without the initial-flow flag the start is px_prev whatever px_init holds
    cur2, st2, _ = orc.klt_track(a, b, px, px + 9.0, use_initial_flow=False)
    assert np.array_equal(cur, cur2) and np.array_equal(st, st2)


def test_oracle_klt_against_scene_flow(orc, klt_seq):
    """The synthetic scene knows every pixel's depth and both poses: the tracked positions must agree with the projected ones."""
    im = klt_seq["images"]
    cam = synth.camera(klt_seq["spec"])
    px = _points(im[0].shape, 300, 3, border=40)
    cur, st, _ = orc.klt_track(im[0], im[1], px, px)
    q, t = klt_seq["T_f_w"][1]
    P = klt_seq["scene"].points0(px[:, 0].astype(np.float64), px[:, 1].astype(np.float64)) @ synth.quat_to_R(q).T + t
    uv = np.array([orc.world2cam(cam, p) for p in P])
    flow = np.linalg.norm(uv - px, axis=1)
    err = np.linalg.norm(uv - cur, axis=1)[st > 0]
    assert np.median(flow) > 3.0                              # a real displacement, so the pyramid matters
    assert st.mean() > 0.95 and np.median(err) < 0.1 and np.percentile(err, 90) < 0.5, (st.mean(), np.median(err), np.percentile(err, 90))
    ok = [orc.patch_check(im[0], im[1], px[i], cur[i]) for i in range(len(px)) if st[i]]
    assert np.mean([o for o, _ in ok]) > 0.9
    # a patch off the image fails the check whatever it looks like
    assert orc.patch_check(im[0], im[1], np.array([3.9, 100], np.float32), np.array([50, 50], np.float32)) == (False, -2.0)


@pytest.mark.gpu
def test_klt_image_kernels_bit_exact(orc, gpu_ctx, klt_seq):
    img = klt_seq["images"][0]
    h, w = img.shape
    gpu_ctx.frame_upload(88001, img)
    try:
        lvl = img
        for level in range(4):
            if level:
                lvl = orc.pyr_down(lvl)
            g_img, g_der = gpu_ctx.klt_debug_level(88001, level, w, h)
            assert np.array_equal(g_img, lvl), level
            assert np.array_equal(g_der, orc.scharr_deriv(lvl)), level
    finally:
        gpu_ctx.frame_release(88001)


def _compare(orc, gpu_ctx, a, b, px, init, tag):
    gpu_ctx.frame_upload(88002, a); gpu_ctx.frame_upload(88003, b)
    try:
        res = gpu_ctx.klt_track(88002, 88003, px, init)
    finally:
        gpu_ctx.frame_release(88002); gpu_ctx.frame_release(88003)
    cur, st, mg = orc.klt_track(a, b, px, init)
    excused = mg < 1e-3
    tracked = (res["status"] & capi.KLT_TRACKED) != 0
    bad_status = (tracked != (st > 0)) & ~excused
    assert not bad_status.any(), (tag, np.flatnonzero(bad_status)[:10])
    both = tracked & (st > 0)
    d = np.abs(res["px"] - cur).max(axis=1)
    bad_px = both & (d > 2e-3) & ~excused
    assert not bad_px.any(), (tag, np.flatnonzero(bad_px)[:10], d[bad_px][:10])
    # the patch check runs on the DEVICE's position: restate it there
    n_ncc = 0
    for i in np.flatnonzero(both):
        ok, ncc = orc.patch_check(a, b, px[i], res["px"][i])
        g_ok = bool(res["status"][i] & capi.KLT_PATCH_OK)
        assert abs(ncc - res["ncc"][i]) <= 1e-5 or (ncc == -2.0 and res["ncc"][i] == -2.0), (tag, i, ncc, res["ncc"][i])
        assert ok == g_ok or abs(ncc - 0.8) < 1e-4, (tag, i, ncc)
        n_ncc += 1
    return dict(n=len(px), tracked=int(both.sum()), excused=int(excused.sum()), max_dpx=float(d[both & ~excused].max()) if both.any() else 0.0,
                patch_ok=int(((res["status"] & capi.KLT_PATCH_OK) != 0).sum()), checked=n_ncc)


@pytest.mark.gpu
def test_klt_track_parity(orc, gpu_ctx, klt_seq):
    im = klt_seq["images"]
    px = _points(im[0].shape, 1500, 11, border=6)             # includes windows that hang over the image edge (reflected / zero-derivative taps)
    s1 = _compare(orc, gpu_ctx, im[0], im[1], px, px, "f0->f1")
    assert s1["tracked"] > 0.9 * s1["n"] and s1["excused"] < 0.02 * s1["n"], s1
    # a bad initial flow: some points diverge / leave the image, the status bytes must still agree
    rng = np.random.default_rng(2)
    init = px + rng.normal(0, 6, px.shape).astype(np.float32)
    s2 = _compare(orc, gpu_ctx, im[0], im[3], px, init, "f0->f3 noisy start")
    # flat image: the minimum-eigenvalue test must drop every point on both sides
    flat = np.full_like(im[0], 90)
    s3 = _compare(orc, gpu_ctx, flat, flat, px[:64], px[:64], "flat")
    assert s3["tracked"] == 0
    assert s2["excused"] < 0.1 * s2["n"], s2
    print("klt parity:", s1, s2, s3)


@pytest.mark.gpu
def test_klt_track_argument_errors(gpu_ctx, klt_seq):
    im = klt_seq["images"]
    px = _points(im[0].shape, 4, 1)
    with pytest.raises(capi.HsoGpuError):
        gpu_ctx.klt_track(88010, 88011, px, px)                # frames not resident
    gpu_ctx.frame_upload(88010, im[0])
    try:
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.klt_track(88010, 88010, px, px, capi.KltParams(win_size=40))
        assert len(gpu_ctx.klt_track(88010, 88010, px[:0], px[:0])) == 0
        res = gpu_ctx.klt_track(88010, 88010, px, px)           # a frame against itself: nothing moves
        assert np.abs(res["px"] - px).max() < 1e-3 and (res["status"] == 3).all() and (res["ncc"] > 0.999).all()
    finally:
        gpu_ctx.frame_release(88010)


@pytest.mark.gpu
def test_klt_parity_on_a_size_with_odd_pyramid_levels(orc, gpu_ctx):
    """920x736 (the TUM-mono camera after its downscale rule): Gaussian levels 460x368, 230x184, 115x92, 58x46 — an odd width on the
    way down ((115 + 1) / 2 = 58: the reflected column takes part) and five levels instead of four."""
    S = synth.sequence(n_frames=3, spec=synth.TUM_WIDE, step=(0.05, 0.015, 0.01), workers=3)
    im = S["images"]
    h, w = im[0].shape
    assert (w, h) == (920, 736) and orc.klt_levels(w, h) == 4
    gpu_ctx.frame_upload(88020, im[0])
    try:
        lvl = im[0]
        for level in range(5):
            if level:
                lvl = orc.pyr_down(lvl)
            g_img, g_der = gpu_ctx.klt_debug_level(88020, level, w, h)
            assert g_img.shape == lvl.shape and np.array_equal(g_img, lvl), level
            assert np.array_equal(g_der, orc.scharr_deriv(lvl)), level
    finally:
        gpu_ctx.frame_release(88020)
    px = _points(im[0].shape, 800, 21, border=6)
    s = _compare(orc, gpu_ctx, im[0], im[2], px, px, "tum 920x736")
    assert s["tracked"] > 0.9 * s["n"] and s["excused"] < 0.15 * s["n"], s      # two frames apart: more points end a level at the iteration cap
    print("klt parity 920x736:", s)
