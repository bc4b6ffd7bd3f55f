"""Reprojector candidate generation chained into the matcher:
Reprojector::reprojectPoint src/reprojector.cpp:504-529, Point::getCloseViewObs src/point.cpp:116-136,
Matcher::findMatchDirect src/matcher.cpp:270-375 — hso_gpu_reproject_match against the oracle."""
import ctypes as C
import math

import numpy as np
import pytest

from hso_amd import capi, synth


def _tables(pos, host_f, idist, obs_kfs, kf_t):
    kfs = np.zeros(len(kf_t), capi.KF_DTYPE)
    for k, t in enumerate(kf_t):
        kfs[k]["frame_id"], kfs[k]["q"], kfs[k]["t"], kfs[k]["exposure_time"], kfs[k]["keyframe_id"] = 100 + k, [0, 0, 0, 1], t, 1.0, k
    pts = np.zeros(1, capi.MAP_POINT_DTYPE)
    pts[0]["pos"], pts[0]["idist"], pts[0]["host_f"], pts[0]["host_kf"] = pos, idist, host_f, 0
    pts[0]["obs_begin"], pts[0]["obs_count"] = 0, len(obs_kfs)
    obs = np.zeros(len(obs_kfs), capi.OBS_DTYPE)
    obs["kf"] = obs_kfs
    return kfs, pts, obs


def test_oracle_reproject_point_and_cell(orc):
    cam = synth.camera()
    lib = orc.load()
    lib.hso_or_reproject_point.argtypes = [C.POINTER(capi.Camera), C.POINTER(capi.SE3), C.c_void_p, C.c_void_p, C.c_double,
                                           C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    I = capi.SE3.identity()
    Th = np.zeros(7); Th[3] = 1.0                             # q = (0,0,0,1), t = 0

    def run(f, idist, T_cur=I):
        f = np.array(f, float); px = np.zeros(2); cell = C.c_int(-1)
        ok = lib.hso_or_reproject_point(C.byref(cam), C.byref(T_cur), Th.ctypes.data, f.ctypes.data, idist, 23, 28,
                                        px.ctypes.data, C.byref(cell))
        return ok, px, cell.value
    # bearing through pixel (400.25, 100.75) at depth 2: lands there, cell = (100/23)*28 + 400/23
    b = np.array([(400.25 - cam.cx) / cam.fx, (100.75 - cam.cy) / cam.fy, 1.0])
    ok, px, cell = run(b / np.linalg.norm(b), 1.0 / (2.0 * np.linalg.norm(b)))
    assert ok == 1 and np.allclose(px, [400.25, 100.75], atol=1e-9) and cell == 4 * 28 + 17
    # the 8-pixel border is tested on the truncated position: 7.99 -> 7 fails, 8.0 passes; 631.9 -> 631 passes, 632 fails
    for u, want in ((7.99, 0), (8.0, 1), (631.9, 1), (632.0, 0)):
        b = np.array([(u - cam.cx) / cam.fx, (200.0 - cam.cy) / cam.fy, 1.0])
        assert run(b, 0.5)[0] == want, u
    # behind the camera / closer than 1e-5
    assert run([0, 0, 1.0], -0.5)[0] == 0
    assert run([0, 0, 1.0], 1.0 / 0.9e-5)[0] == 0 and run([0, 0, 1.0], 1.0 / 1.1e-5)[0] == 1
    # a translated current frame shifts the projection by fx * tx / z
    T = capi.SE3.from_arrays([0, 0, 0, 1.0], [0.1, 0, 0])
    ok, px, _ = run([0, 0, 1.0], 0.5, T)
    assert ok == 1 and px[0] == pytest.approx(cam.cx + cam.fx * 0.1 / 2.0, abs=1e-9)


def test_oracle_close_view_obs(orc):
    lib = orc.load()
    lib.hso_or_close_view_obs.argtypes = [C.c_void_p] * 4 + [C.c_int]
    lib.hso_or_close_view_obs.restype = C.c_int
    pos = np.array([0.0, 0.0, 4.0])
    cur = np.array([0.0, 0.0, 0.0])
    # keyframe positions = -t for identity rotations; angles seen from the point: 0, ~14, 90 degrees
    s60 = 4.0 * math.tan(math.radians(59.0)); s61 = 4.0 * math.tan(math.radians(61.0))
    kfs, pts, obs = _tables(pos, [0, 0, 1], 0.25, [2, 1, 0], [[0, 0, 0], [-1.0, 0, 0], [-4.0, 0, -4.0]])
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 3) == 2   # kf 0 is at the frame position
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 2) == 1   # then the 14 degree one
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 1) == -1  # 90 degrees: useless
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 0) == -1
    # the 60 degree threshold (cos 0.5), and the first of two equal views wins (strict >)
    kfs, pts, obs = _tables(pos, [0, 0, 1], 0.25, [0, 1, 2, 2], [[-s61, 0, 0], [-s60, 0, 0], [0, -1.0, 0]])
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 1) == -1
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs.ctypes.data, 2) == 1
    assert lib.hso_or_close_view_obs(cur.ctypes.data, pos.ctypes.data, kfs.ctypes.data, obs[2:].ctypes.data, 2) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [synth.ICL_NUIM, synth.EUROC], ids=["pinhole", "radtan"])
def test_reproject_match_equals_oracle(gpu_ctx, orc, spec):
    P = synth.map_problem(n_points=900, spec=spec, first_frame_id=9800 if spec is synth.ICL_NUIM else 9830)
    cam = synth.camera(spec)
    ids = [int(k["frame_id"]) for k in P["kfs"]]
    for i, f in zip(ids, P["frames"]):
        gpu_ctx.frame_upload(i, f)
    gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"])
    try:
        proj, match = gpu_ctx.reproject_match(cam, P["cur_frame_id"], P["T_cur_w"], P["cur_exposure"], P["cur_keyframe_id"],
                                              P["kfs"], P["points"], P["obs"], P["cell_size"], P["grid_n_cols"])
        kf_pyrs = [orc.create_pyramid(f) for f in P["frames"]]
        cur_pyr = orc.create_pyramid(P["cur"])
        cur_sobel = [orc.sobel5(np.ascontiguousarray(cur_pyr[L])) for L in range(3)]
        margins = []
        wproj, wmatch = orc.reproject_match(cam, P["T_cur_w"], P["cur_exposure"], P["cur_keyframe_id"], P["kfs"], P["points"],
                                            P["obs"], P["cell_size"], P["grid_n_cols"], kf_pyrs, cur_pyr, cur_sobel, margins_out=margins)
        kd = 1.0 if spec is synth.ICL_NUIM else 5.0   # radtan: A_cur_ref carries 2e-5 instead of 1e-9 (see below)
        n_proj = n_ref = n_ok = n_tie = 0
        for i in range(len(proj)):
            g, w = proj[i], wproj[i]
            if g["projected"] != w["projected"] or (g["projected"] and g["cell"] != w["cell"]):
                # only a projection within rounding of an integer pixel / cell border may decide differently
                px = w["px"] if w["projected"] else g["px"]
                assert min(abs(px[0] - round(px[0])), abs(px[1] - round(px[1]))) < 1e-9, i
                n_tie += 1
                continue
            if not g["projected"]:
                assert g["ref_obs"] == -1 and match[i].success == 0 and match[i].iters == 0
                continue
            n_proj += 1
            assert np.allclose(g["px"], w["px"], atol=1e-9, rtol=0)
            assert g["ref_obs"] == w["ref_obs"], i
            m = match[i]
            if g["ref_obs"] < 0:
                assert (m.success, m.stage, m.iters) == (0, 0, 0) and wmatch[i] is None
                continue
            n_ref += 1
            o = wmatch[i]
            assert m.search_level == o.search_level
            mg = margins[i]
            if m.iters != o.iters:
                assert mg.lk_update < 1e-2 * kd, (i, m.iters, o.iters, mg.lk_update)     # the margin rule of tests/test_align.py
                n_tie += 1
                continue
            if (m.success, m.stage) != (o.success, o.stage):
                assert min(mg.ncc / 1e-3, mg.normal / 1e-3, mg.lk_chi2 / 1e-2, mg.lk_update / 1e-2, mg.jump / 1e-2) < kd, \
                    (i, m.stage, o.stage, [getattr(mg, f) for f in orc.MARGIN_FIELDS])
                n_tie += 1
                continue
            # radtan: cam2world runs OpenCV's five fp32 undistortion iterations (src/camera.cpp:171-194),
            # whose rounding differs between host and device at the 1e-7 level of the bearing
            assert np.allclose(m.A_cur_ref[:], o.A_cur_ref[:], atol=1e-9 if spec is synth.ICL_NUIM else 2e-5)
            if o.success:
                n_ok += 1
                assert np.allclose(m.px_cur[:], o.px_cur[:], atol=2e-3)
        assert n_proj > 600 and n_ref > 500 and n_ok > 350, (n_proj, n_ref, n_ok, n_tie)   # coverage; ties excused by margin only
        # the cases the generator plants: outside / behind -> not projected; useless or no observation -> no reference
        idx = np.arange(len(proj))
        assert not proj["projected"][(idx % 29 == 4) | (idx % 31 == 6)].any()
        sel = proj["projected"].astype(bool) & ((idx % 37 == 9) | (idx % 41 == 11))
        assert sel.sum() > 20 and (proj["ref_obs"][sel] == -1).all()
        far = len(P["kfs"]) - 1
        chosen = proj["ref_obs"][proj["ref_obs"] >= 0]
        assert (P["obs"]["kf"][chosen] != far).all()
        # equal to hso_gpu_align_batch on the jobs the host adapter would have built
        jobs, slots = [], []
        lib = orc.load()
        for i in np.nonzero(proj["ref_obs"] >= 0)[0][:200]:
            j = capi.AlignJob()
            lib.hso_or_reproject_make_job(C.byref(P["T_cur_w"]), P["cur_exposure"], P["cur_keyframe_id"], P["kfs"].ctypes.data,
                                          P["points"][i:i + 1].ctypes.data, P["obs"][proj["ref_obs"][i]:proj["ref_obs"][i] + 1].ctypes.data,
                                          proj["px"][i].copy().ctypes.data, C.byref(j))
            jobs.append(j); slots.append(i)
        direct = gpu_ctx.align_batch(cam, P["cur_frame_id"], jobs)
        same = sum((d.success, d.stage, d.iters, d.search_level) == (match[i].success, match[i].stage, match[i].iters, match[i].search_level)
                   and abs(d.px_cur[0] - match[i].px_cur[0]) < 1e-6 for d, i in zip(direct, slots))
        assert same >= len(slots) - 2
    finally:
        for i in ids + [P["cur_frame_id"]]:
            gpu_ctx.frame_release(i)


@pytest.mark.gpu
def test_reproject_match_argument_errors(gpu_ctx):
    P = synth.map_problem(n_points=40, first_frame_id=9860)
    cam = synth.camera()
    ids = [int(k["frame_id"]) for k in P["kfs"]]
    for i, f in zip(ids, P["frames"]):
        gpu_ctx.frame_upload(i, f)
    gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"])
    try:
        args = (cam, P["cur_frame_id"], P["T_cur_w"], 1.0, 9, P["kfs"], P["points"], P["obs"], P["cell_size"], P["grid_n_cols"])
        gpu_ctx.reproject_match(*args)
        bad = P["points"].copy(); bad["host_kf"][3] = 99
        with pytest.raises(capi.HsoGpuError, match="out of range"):
            gpu_ctx.reproject_match(*args[:6], bad, *args[7:])
        bad = P["points"].copy(); bad["obs_begin"][5] = len(P["obs"])
        bad["obs_count"][5] = 1
        with pytest.raises(capi.HsoGpuError, match="out of range"):
            gpu_ctx.reproject_match(*args[:6], bad, *args[7:])
        kf2 = P["kfs"].copy(); kf2["frame_id"][1] = 424242
        with pytest.raises(capi.HsoGpuError, match="not resident"):
            gpu_ctx.reproject_match(*args[:5], kf2, *args[6:])
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.reproject_match(*args[:8], 0, 28)
        empty = gpu_ctx.reproject_match(*args[:6], P["points"][:0], *args[7:])
        assert len(empty[0]) == 0
    finally:
        for i in ids + [P["cur_frame_id"]]:
            gpu_ctx.frame_release(i)


@pytest.mark.gpu
def test_reproject_match_multi_equals_per_frame_calls(gpu_ctx):
    """Two sequences (their own keyframes, points, current frames) in one launch: every point gets
    exactly the result of its sequence's single call."""
    cam = synth.camera()
    A = synth.map_problem(n_points=400, first_frame_id=9870, seed=71)
    B = synth.map_problem(n_points=300, first_frame_id=9885, seed=93, n_kfs=4)
    up = []
    for P in (A, B):
        for k, f in zip(P["kfs"], P["frames"]):
            gpu_ctx.frame_upload(int(k["frame_id"]), f); up.append(int(k["frame_id"]))
        gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"]); up.append(P["cur_frame_id"])
    try:
        solo = [gpu_ctx.reproject_match(cam, P["cur_frame_id"], P["T_cur_w"], P["cur_exposure"], P["cur_keyframe_id"], P["kfs"],
                                        P["points"], P["obs"], P["cell_size"], P["grid_n_cols"]) for P in (A, B)]
        frames = np.zeros(2, capi.REPROJ_FRAME_DTYPE)
        kf_at = pt_at = ob_at = 0
        pts_all, obs_all = [], []
        for r, P in enumerate((A, B)):
            q, t = P["T_cur_w"].to_arrays()
            frames[r]["cur_frame_id"], frames[r]["q"], frames[r]["t"] = P["cur_frame_id"], q, t
            frames[r]["cur_exposure_time"], frames[r]["cur_keyframe_id"] = P["cur_exposure"], P["cur_keyframe_id"]
            frames[r]["kf_begin"], frames[r]["kf_count"] = kf_at, len(P["kfs"])
            frames[r]["point_begin"], frames[r]["point_count"] = pt_at, len(P["points"])
            pts = P["points"].copy(); pts["obs_begin"] += ob_at              # obs_begin is absolute, keyframe indices stay relative
            pts_all.append(pts); obs_all.append(P["obs"])
            kf_at += len(P["kfs"]); pt_at += len(pts); ob_at += len(P["obs"])
        proj, match = gpu_ctx.reproject_match_multi(cam, frames, np.concatenate([A["kfs"], B["kfs"]]), np.concatenate(pts_all),
                                                    np.concatenate(obs_all), A["cell_size"], A["grid_n_cols"])
        at = ob = 0
        for (sp, sm), P in zip(solo, (A, B)):
            n = len(P["points"])
            want = sp.copy()
            want["ref_obs"] = np.where(want["ref_obs"] >= 0, want["ref_obs"] + ob, -1)
            assert proj[at:at + n].tobytes() == want.tobytes()
            for i in range(n):
                g, w = match[at + i], sm[i]
                assert bytes(C.string_at(C.addressof(g), C.sizeof(g))) == bytes(C.string_at(C.addressof(w), C.sizeof(w))), i
            at += n; ob += len(P["obs"])
        assert sum(m.success for m in match) > 350
        # a point outside every frame's range, and overlapping ranges, are refused
        bad = frames.copy(); bad[1]["point_count"] -= 1
        with pytest.raises(capi.HsoGpuError, match="no frame"):
            gpu_ctx.reproject_match_multi(cam, bad, np.concatenate([A["kfs"], B["kfs"]]), np.concatenate(pts_all), np.concatenate(obs_all), 23, 28)
        bad = frames.copy(); bad[1]["point_begin"] -= 1
        with pytest.raises(capi.HsoGpuError, match="overlap|no frame"):
            gpu_ctx.reproject_match_multi(cam, bad, np.concatenate([A["kfs"], B["kfs"]]), np.concatenate(pts_all), np.concatenate(obs_all), 23, 28)
    finally:
        for i in up:
            gpu_ctx.frame_release(i)
