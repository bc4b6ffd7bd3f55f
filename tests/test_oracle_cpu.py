"""CPU tests (no GPU): the oracle against the reference-derived golden vectors, its own
self-consistency, and the committed self-golden of the tracker."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from hso_amd import capi, synth


def f32(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


# ---------------------------------------------------------------- robust cost: pinned
def test_robust_cost_matches_reference_golden(orc):
    """tests/golden/robust_cost.json was produced by the compiled reference
    (src/vikit/robust_cost.cpp); the restatement must reproduce it bit for bit."""
    g = json.load(open(os.path.join(GOLDEN, "robust_cost.json")))
    lib = orc.load()
    k, b = f32(g["huber_k"]), f32(g["tukey_b"])
    assert k == np.float32(1.345) and b == np.float32(4.6851)
    for i, xb in enumerate(g["x"]):
        x = f32(xb)
        assert bits(lib.hso_or_huber_weight(k, x)) == g["huber"][i]
        assert bits(lib.hso_or_tukey_weight(b, x)) == g["tukey"][i]
        assert bits(lib.hso_or_tdist_weight(5.0, x)) == g["tdist"][i]
    for c in g["mad"]:
        e = np.array([f32(v) for v in c["errors"]], np.float32)
        assert bits(lib.hso_or_mad_scale(e.ctypes.data, len(e))) == c["scale"]
    for c in g["tdist_scale"]:
        e = np.array([f32(v) for v in c["errors"]], np.float32)
        assert bits(lib.hso_or_tdist_scale(5.0, e.ctypes.data, len(e))) == c["scale"]


def test_robust_cost_matches_live_reference(orc):
    """When oracle/_ref is present (authoring container / shipped .so) compare live."""
    ref = orc.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    lib = orc.load()
    rng = np.random.default_rng(5)
    for x in rng.normal(0, 4, 500).astype(np.float32):
        assert bits(lib.hso_or_huber_weight(1.345, float(x))) == bits(ref.ref_huber_weight(1.345, float(x)))
    for n in (5, 64, 999):
        e = np.abs(rng.normal(0, 1, n)).astype(np.float32)
        assert bits(lib.hso_or_mad_scale(e.ctypes.data, n)) == bits(ref.ref_mad_scale(e.ctypes.data, n))


# ---------------------------------------------------------------- math
def test_median_is_upper_median(orc):
    lib = orc.load()
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 4, 5, 30, 31, 1000, 4097):
        a = rng.normal(size=n).astype(np.float32)
        b = a.copy()
        m = lib.hso_or_median_f(b.ctypes.data, n)
        assert m == np.sort(a)[n // 2]  # nth_element at floor(n/2), math_utils.h:123


def test_se3_exp_log_roundtrip_and_group_laws(orc):
    rng = np.random.default_rng(2)
    for _ in range(50):
        v = rng.normal(0, 0.3, 6)
        T = orc.se3_exp(v)
        assert np.allclose(orc.se3_log(T), v, atol=1e-12)
        q, _ = T.to_arrays()
        assert abs(np.linalg.norm(q) - 1) < 1e-15
        Ti = orc.se3_inverse(T)
        I = orc.se3_mul(T, Ti)
        qi, ti = I.to_arrays()
        assert np.allclose(qi, [0, 0, 0, 1], atol=1e-14) and np.allclose(ti, 0, atol=1e-14)
        p = rng.normal(size=3)
        R = orc.so3_matrix(q)
        assert np.allclose(orc.se3_apply(T, p), R @ p + np.array(T.t[:]), atol=1e-14)
    # tangent order is [upsilon, omega] (thirdparty/Sophus/sophus/se3.cpp:170-173)
    T = orc.se3_exp([1, 2, 3, 0, 0, 0])
    assert np.allclose(T.t[:], [1, 2, 3]) and np.allclose(T.q[:], [0, 0, 0, 1])
    # small-angle branch (theta < 1e-10, so3.cpp:188-193)
    T = orc.se3_exp([0.1, 0, 0, 1e-12, 0, 0])
    assert np.isfinite(T.q[:]).all() and abs(T.t[0] - 0.1) < 1e-12


def test_ldlt_solves_spd_and_semidefinite(orc):
    rng = np.random.default_rng(3)
    for n in (6, 7):
        for _ in range(20):
            A = rng.normal(size=(n, 40))
            H = A @ A.T
            b = rng.normal(size=n)
            assert np.allclose(orc.ldlt_solve(H, b), np.linalg.solve(H, b), rtol=1e-9, atol=1e-12)
    # all-zero matrix -> zero step (Eigen pseudo-inverse in LDLT::_solve_impl)
    assert np.all(orc.ldlt_solve(np.zeros((7, 7)), np.ones(7)) == 0)
    # badly scaled diagonal exercises the pivoting
    H = np.diag([1e-6, 1e6, 1.0, 10.0, 1e-3, 5.0, 2.0]) + 1e-8
    b = np.arange(7.0)
    assert np.allclose(orc.ldlt_solve(H, b), np.linalg.solve(H, b), rtol=1e-9)


# ---------------------------------------------------------------- camera
def test_world2cam_cam2world_models(orc):
    pin = capi.make_camera(**synth.ICL_NUIM)
    eu = capi.make_camera(**synth.EUROC)
    assert pin.distortion == 0 and eu.distortion == 1
    rng = np.random.default_rng(4)
    for cam in (pin, eu):
        for _ in range(100):
            u, v = rng.uniform(20, cam.width - 20), rng.uniform(20, cam.height - 20)
            f = orc.cam2world(cam, u, v)
            assert abs(np.linalg.norm(f) - 1) < 1e-12
            px = orc.world2cam(cam, f * rng.uniform(0.5, 10))
            # radtan cam2world goes through the 5-iteration fp32 undistortPoints (camera.cpp:78-85):
            # five fixed-point iterations leave up to ~0.3 px at the EuRoC image corners
            r = np.hypot(u - cam.cx, v - cam.cy)
            tol = (0.5 if r > 200 else 2e-2) if cam.distortion else 1e-9
            assert np.allclose(px, [u, v], atol=tol)
    fov = capi.make_camera(capi.CAM_FOV, 640, 480, 300, 300, 320, 240, d=(0.9, 0, 0, 0, 0), distortion=1)
    f = orc.cam2world(fov, 100.0, 50.0)
    assert np.allclose(orc.world2cam(fov, f), [100, 50], atol=1e-9)
    assert orc.load().hso_or_error_multiplier2(C.byref(eu)) == pytest.approx((458.654 + 457.296) / 2)


# ---------------------------------------------------------------- pyramid / sobel
def test_half_sample_roundings(orc):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (32, 64), dtype=np.uint8)  # 64 % 16 == 0 -> SSE2 rounding
    a, b, c, d = (img[0::2, 0::2].astype(int), img[0::2, 1::2].astype(int),
                  img[1::2, 0::2].astype(int), img[1::2, 1::2].astype(int))
    exp_sse = ((((a + c + 1) >> 1) + ((b + d + 1) >> 1) + 1) >> 1).astype(np.uint8)
    assert np.array_equal(orc.half_sample(img), exp_sse)
    img2 = rng.integers(0, 256, (30, 376), dtype=np.uint8)  # 376 % 16 != 0 -> truncating scalar loop
    a, b, c, d = (img2[0::2, 0::2].astype(int), img2[0::2, 1::2].astype(int),
                  img2[1::2, 0::2].astype(int), img2[1::2, 1::2].astype(int))
    assert np.array_equal(orc.half_sample(img2), ((a + b + c + d) // 4).astype(np.uint8))
    # the two roundings really differ
    assert not np.array_equal(exp_sse, ((img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) // 4))


def test_pyramid_euroc_level_paths(orc):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (480, 752), dtype=np.uint8)
    lv = orc.create_pyramid(img)
    assert [l.shape for l in lv] == [(480, 752), (240, 376), (120, 188), (60, 94), (30, 47)]
    assert np.array_equal(lv[1], orc.half_sample(lv[0])) and np.array_equal(lv[4], orc.half_sample(lv[3]))


def test_pyramid_resize_branch_tum_mono(orc):
    """Level-0 sizes that are not multiples of 16 take cv::resize (frame.cpp:307-312): TUM-mono's
    920x736.  Level sizes follow cvRound (58, not 57, at level 4); an exact 2x step is OpenCV's
    area-fast path (a+b+c+d+2)>>2, the 115->58 step the fixed-point bilinear kernel."""
    rng = np.random.default_rng(16)
    img = rng.integers(0, 256, (736, 920), dtype=np.uint8)
    lv = orc.create_pyramid(img)
    assert [l.shape for l in lv] == [(736, 920), (368, 460), (184, 230), (92, 115), (46, 58)]
    for l in (1, 2, 3):
        a = lv[l - 1].astype(int)
        assert np.array_equal(lv[l], ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # 115x92 -> 58x46: within one grey level of real-valued bilinear sampling at aligned pixel centres
    src = lv[3].astype(float)
    sx = (np.arange(58) + 0.5) * (115 / 58) - 0.5; sy = (np.arange(46) + 0.5) * (92 / 46) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 113); fx = np.clip(sx - x0, 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 90); fy = np.clip(sy - y0, 0, 1)
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x0 + 1] * fx
    bot = src[y0 + 1][:, x0] * (1 - fx) + src[y0 + 1][:, x0 + 1] * fx
    assert np.abs(lv[4].astype(float) - (top * (1 - fy)[:, None] + bot * fy[:, None])).max() <= 1.0
    # a case small enough to do by hand: [0 100 200] -> 2 pixels, weights (0.75, 0.25) and (0.25, 0.75)
    tiny = np.array([[0, 100, 200], [0, 100, 200]], np.uint8)
    assert orc.resize_linear(tiny, 2, 2).tolist() == [[25, 175], [25, 175]]
    assert orc.pyramid_dims(922, 738, 1) == (461, 369) and orc.pyramid_dims(926, 730, 2) == (232, 182)   # 231.5 -> 232, 182.5 -> 182: half to even


def test_sobel5_against_direct_convolution(orc):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    gx, gy = orc.sobel5(img)
    kd, ks = np.array([-1, -2, 0, 2, 1]), np.array([1, 4, 6, 4, 1])
    p = np.pad(img.astype(int), 2, mode="edge")
    ex = np.zeros(img.shape, int); ey = np.zeros(img.shape, int)
    for j in range(5):
        for i in range(5):
            ex += ks[j] * kd[i] * p[j:j + 37, i:i + 53]
            ey += kd[j] * ks[i] * p[j:j + 37, i:i + 53]
    assert np.array_equal(gx, ex) and np.array_equal(gy, ey)
    # constant image -> zero gradient, incl. the replicated border
    gx, gy = orc.sobel5(np.full((20, 20), 200, np.uint8))
    assert not gx.any() and not gy.any()


# ---------------------------------------------------------------- tracker
def test_pattern_tables_keep_the_reference_quirk(orc):
    rc, pa, hp, offs = orc.pattern(4, 4)
    assert (pa, hp) == (9, 1)
    # include/hso/CoarseTracker.h:69: {-1,0} twice, {0,-1} missing
    lst = [tuple(o) for o in offs]
    assert lst.count((-1, 0)) == 2 and (0, -1) not in lst
    assert [orc.pattern(4, l)[1] for l in (4, 3, 2, 1, 0)] == [9, 13, 13, 21, 25]
    assert [orc.pattern(4, l)[2] for l in (4, 3, 2, 1, 0)] == [1, 2, 2, 3, 2]


def test_tracker_normal_equations_match_finite_differences(orc, pair200, cam):
    """b = -J^T W r and H ~ J^T W J: check b against a numeric derivative of the
    unsaturated weighted residual sum (forward mode, fixed weights)."""
    d = pair200
    rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
    p = capi.TrackParams(0, 4, 1, 50)
    tr = orc.Tracker(cam, p, rp, cp, d["feats"])
    tr.set_level(1)
    tr.set_thresholds(1e9, 1e9)  # hw = 1 everywhere, no saturation: E = sum r^2
    T0 = capi.SE3.identity()     # away from the optimum, where the gradient is well above the noise
    e0 = tr.eval(T0, 1.05)
    b = np.array(e0.b[:])
    # jacobian_xyz2uv is negated (frame.h:192, "for r = obs - proj"), so J = -dr/dxi and
    # b = -sum w r J = +1/2 dE/dxi; the update exp(-H^-1 b) then descends (CoarseTracker.cpp:131).
    # E is a serial fp32 sum (:272), so the step must be large enough to rise above its rounding.
    num = np.zeros(6)
    for k in range(6):
        eps = 2e-3 if k < 3 else 5e-4
        v = np.zeros(6); v[k] = eps
        ep = tr.eval(orc.se3_mul(orc.se3_exp(v), T0), 1.05)
        em = tr.eval(orc.se3_mul(orc.se3_exp(-v), T0), 1.05)
        num[k] = (ep.energy_sum - em.energy_sum) / (2 * eps)
    # the photometric J is itself a central difference of a bilinear surface over a
    # band-limited-but-busy texture: direction and magnitude agree, not every component
    ana = 2 * b[1:]
    assert ana @ num / (np.linalg.norm(ana) * np.linalg.norm(num)) > 0.98
    assert 0.8 < np.linalg.norm(ana) / np.linalg.norm(num) < 1.25
    # exposure: r = I_cur - a I_ref, J_e = -I_ref = dr/da  =>  dE/da = 2 sum r J_e = -2 b[0]
    ea = 1e-3
    dEa = (tr.eval(T0, 1.05 + ea).energy_sum - tr.eval(T0, 1.05 - ea).energy_sum) / (2 * ea)
    assert abs(-2 * b[0] - dEa) <= 0.02 * abs(dEa) + 1.0
    H = np.array(e0.H[:]).reshape(7, 7)
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)


def test_tracker_converges_to_ground_truth(orc, pair2000, cam):
    d = pair2000
    rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
    for inv in (0, 1):
        tr = orc.Tracker(cam, capi.TrackParams(inv, 4, 1, 50), rp, cp, d["feats"])
        r = tr.run(capi.SE3.identity(), 1.0)
        q, t = r.T_cur_ref.to_arrays()
        assert np.linalg.norm(q - d["q_true"]) < 2e-4
        assert np.linalg.norm(t - d["t_true"]) < 2e-3
        assert abs(r.exposure_rat - d["exposure"]) < 5e-3
        assert r.n_tracked > 1900


def test_tracker_edge_cases(orc, pair200, cam):
    d = pair200
    rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
    p = capi.TrackParams(0, 4, 1, 50)
    # no features: run returns the initial state (CoarseTracker.cpp:53-54)
    r = orc.Tracker(cam, p, rp, cp, d["feats"][:0]).run(capi.SE3.identity(), 1.25)
    assert r.n_tracked == 0 and r.exposure_rat == 1.25
    # all features without a point / behind the camera: < 30 terms -> default thresholds (5.2, 100)
    f = d["feats"].copy(); f["dist"] = -1
    r = orc.Tracker(cam, p, rp, cp, f).run(capi.SE3.identity(), 1.0)
    assert r.n_tracked == 0 and r.huber[4] == np.float32(5.2) and r.outlier[4] == 100
    assert np.allclose(r.T_cur_ref.q[:], [0, 0, 0, 1]) and r.iters[4] == 1
    # features on the image border are invisible at coarse levels but keep their index
    f = d["feats"].copy(); f["px"][:10] = [[1.0, 1.0]] * 10
    tr = orc.Tracker(cam, p, rp, cp, f); tr.set_level(4)
    _, vis = tr.cache()
    assert not vis[:10].any() and vis[10:].sum() > 150


def test_make_depth_ref(orc):
    rng = np.random.default_rng(8)
    poses = [orc.se3_exp(rng.normal(0, 0.2, 6)) for _ in range(4)]
    T_ref = orc.se3_exp(rng.normal(0, 0.2, 6))
    din = np.zeros(50, capi.DEPTH_REF_IN_DTYPE)
    din["has_point"] = rng.integers(0, 2, 50)
    din["host_pose"] = rng.integers(0, 4, 50)
    f = rng.normal(size=(50, 3)); f[:, 2] = np.abs(f[:, 2]) + 1
    din["host_f"] = f / np.linalg.norm(f, axis=1, keepdims=True)
    din["idist"] = rng.uniform(0.1, 1, 50)
    din["idist"][:5] = -0.5  # behind the camera
    out = orc.make_depth_ref(din, poses, T_ref)
    for i in range(50):
        if not din["has_point"][i]:
            assert out[i] == -1
            continue
        p = orc.se3_apply(orc.se3_mul(T_ref, orc.se3_inverse(poses[din["host_pose"][i]])), din["host_f"][i] / din["idist"][i])
        assert out[i] == (-1 if p[2] < 1e-5 else pytest.approx(np.linalg.norm(p), rel=1e-14))


def test_tracker_self_golden(orc):
    """tests/golden/tracker_small.json (made by tests/golden/make_tracker_golden.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLDEN, "make_tracker_golden.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    g = json.load(open(os.path.join(GOLDEN, "tracker_small.json")))
    d = mk.case()
    assert [int(d["ref"].astype(np.uint64).sum()), int(d["cur"].astype(np.uint64).sum())] == g["image_sha"]
    cam = synth.camera(mk.SPEC)
    rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
    st_r = orc.frame_stats(rp[0], *orc.sobel5(rp[0])); st_c = orc.frame_stats(cp[0], *orc.sobel5(cp[0]))
    assert [st_r.integral_image, st_r.grad_mean, st_c.integral_image, st_c.grad_mean] == g["stats"]
    a0 = float(np.float32(st_c.integral_image / st_r.integral_image))
    for inv in (0, 1):
        r = orc.Tracker(cam, capi.TrackParams(inv, 4, 1, 50), rp, cp, d["feats"]).run(capi.SE3.identity(), a0)
        e = g["runs"][str(inv)]
        assert list(r.iters) == e["iters"] and [int(x) for x in r.accept_mask] == e["accept"]
        assert [float(x) for x in r.huber] == e["huber"] and list(r.n_select) == e["n_select"]
        # libm (sin/cos/sqrt) may differ by an ulp between hosts: poses to 1e-12
        assert np.allclose(r.T_cur_ref.q[:], e["q"], atol=1e-12) and np.allclose(r.T_cur_ref.t[:], e["t"], atol=1e-12)
        assert r.exposure_rat == pytest.approx(e["a"], abs=1e-7) and r.n_tracked == e["n_tracked"]
