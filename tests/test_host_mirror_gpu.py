"""The C++ host mirror (hso_amd/host: reference class names over the C-ABI) driven the way
FrameHandlerMono::processFrame drives the reference, compared with the direct C-ABI call."""
import os
import subprocess

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu
HOST_EXE = os.path.join(os.path.dirname(capi.__file__), "host", "hso_host_test")


class MirrorRun:
    """One run of the C++ mirror's test driver (hso_amd/host/hso_host_test) on a synthetic pair, plus the direct C-ABI calls
    on the same inputs that several of the per-adapter tests below share (each computed once, on first use)."""

    def __init__(self, gpu_ctx, tmp_path, d, cam, inverse):
        self.ctx, self.cam, self.d, self.inverse = gpu_ctx, cam, d, inverse
        n = len(d["feats"])
        self.feats = d["feats"].copy()
        self.idist = 1.0 / self.feats["dist"]
        self.idist[::17] = -1.0                      # features without a point keep their slot
        tab = np.zeros((n, 6))
        tab[:, 0:2], tab[:, 2:5], tab[:, 5] = self.feats["px"], self.feats["f"], self.idist
        case = tmp_path / "case.bin"
        with open(case, "wb") as f:
            f.write(np.array([640, 480, n, inverse], np.int32).tobytes())
            f.write(np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float64).tobytes())
            f.write(d["ref"].tobytes()); f.write(d["cur"].tobytes()); f.write(tab.tobytes())
        out = subprocess.run([HOST_EXE, str(case)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        self.lines = out.stdout.strip().splitlines()
        v = self.lines[0].split()
        self.threw, self.n_tracked = int(v[0]), int(v[1])
        self.q, self.t = np.array(v[2:6], float), np.array(v[6:9], float)
        self.a, self.exposure_time = float(v[9]), float(v[10])
        self.iters = [int(x) for x in v[11:16]]
        self.ii_ref, self.ii_cur = float(v[16]), float(v[17])
        self.has_pt = np.nonzero(self.idist > 0)[0]
        self.T_cw = capi.SE3.from_arrays(self.q, self.t)
        for i in (41, 42):
            try:
                gpu_ctx.frame_release(i)
            except capi.HsoGpuError:
                pass
        self.st_r, self.st_c = gpu_ctx.frame_upload(41, d["ref"]), gpu_ctx.frame_upload(42, d["cur"])
        self.K = int(self.lines[1].split()[0])
        self._got = None

    def matches(self):
        """hso_gpu_align_batch on the jobs the mirror's Matcher::findMatchDirect calls amount to (driver line 2)."""
        if self._got is None:
            cam, feats, idist = self.cam, self.feats, self.idist
            R = synth.quat_to_R(self.q)
            jobs = []
            for k in range(self.K):
                i = self.has_pt[k]
                pos = feats["f"][i] / idist[i]
                pc = R @ pos + self.t
                j = capi.AlignJob()
                j.ref_frame_id = 41; j.ref_level = 0; j.type = capi.FTR_CORNER
                j.px_ref[:] = list(feats["px"][i]); j.f_ref[:] = list(feats["f"][i]); j.grad[:] = [1.0, 0.0]
                j.depth = 1.0 / idist[i]
                j.T_cur_ref = self.T_cw
                j.px_cur[:] = [cam.fx * pc[0] / pc[2] + cam.cx, cam.fy * pc[1] / pc[2] + cam.cy]
                j.exposure_rat = float(np.float32(self.exposure_time / 1.0)); j.kf_gap_lt4 = 1
                jobs.append(j)
            self._got = self.ctx.align_batch(cam, 42, jobs)
        return self._got


@pytest.fixture(scope="module", params=[0, 1], ids=["forward", "inverse_comp"])
def mirror(request, gpu_ctx, tmp_path_factory, pair200, cam):
    return MirrorRun(gpu_ctx, tmp_path_factory.mktemp("mirror"), pair200, cam, request.param)


def test_frame_adapter_rejects_a_wrong_image_size(mirror):
    assert mirror.threw == 1                       # wrong image size -> std::runtime_error (frame.cpp:85-86)
    assert (mirror.st_r.integral_image, mirror.st_c.integral_image) == pytest.approx((mirror.ii_ref, mirror.ii_cur), rel=1e-7)


def test_coarse_tracker_adapter_matches_cabi(mirror):
    """CoarseTracker(...).run(ref, cur) of the mirror == hso_gpu_coarse_track_batch on the same job; dist = |f / idist| as
    makeDepthRef computes it for a point hosted in the reference frame itself (CoarseTracker.cpp:219-235)."""
    m = mirror
    f2 = m.feats.copy()
    p = m.feats["f"] * (1.0 / np.where(m.idist > 0, m.idist, 1.0))[:, None]
    f2["dist"] = np.where(m.idist > 0, np.linalg.norm(p, axis=1), -1.0)
    a0 = float(np.float32(m.st_c.integral_image) / np.float32(m.st_r.integral_image))
    r = m.ctx.coarse_track_batch(m.cam, capi.TrackParams(m.inverse, 4, 1, 50), [m.ctx.make_job(41, 42, f2, capi.SE3.identity(), a0)])[0]
    assert m.iters == list(r.iters) and m.n_tracked == r.n_tracked
    # identity ref pose: cur.T_f_w_ = T_cur_ref * I; dist differs from the adapter's by fp64 rounding only
    assert np.allclose(m.q, r.T_cur_ref.q[:], atol=1e-9) and np.allclose(m.t, r.T_cur_ref.t[:], atol=1e-8)
    assert m.a == pytest.approx(r.exposure_rat, abs=1e-6)
    # write-back rule of CoarseTracker.cpp:200-202 with ref exposure time 1.0
    assert m.exposure_time == (1.0 if 0.99 < m.a < 1.01 else pytest.approx(m.a, rel=1e-6))


def test_matcher_adapter_matches_cabi(mirror):
    """Matcher::findMatchDirect (driver line 2) against hso_gpu_align_batch with the same inputs."""
    mv = mirror.lines[1].split()
    assert mirror.K == 96
    n_ok = 0
    for k, g in enumerate(mirror.matches()):
        ok, px0, px1, sl = int(mv[1 + 4 * k]), float(mv[2 + 4 * k]), float(mv[3 + 4 * k]), int(mv[4 + 4 * k])
        assert (ok, sl) == (g.success, g.search_level)
        assert (px0, px1) == pytest.approx((g.px_cur[0], g.px_cur[1]), abs=1e-6)
        n_ok += ok
    assert n_ok >= 70


def test_depth_filter_adapter_matches_cabi(mirror):
    """DepthFilter::observeDepth (driver line 3) against hso_gpu_seed_observe."""
    import math
    m = mirror
    sv = m.lines[2].split()
    n_seed_ok, n_left = int(sv[0]), int(sv[1])
    seeds = []
    for k in range(m.K, min(len(m.has_pt), 2 * m.K)):
        i = m.has_pt[k]
        sd = capi.Seed()
        sd.ref_frame_id = 41; sd.level = 0; sd.type = capi.FTR_CORNER
        sd.px[:] = list(m.feats["px"][i]); sd.f[:] = list(m.feats["f"][i]); sd.grad[:] = [1.0, 0.0]
        sd.T_ref_w = capi.SE3.identity(); sd.ref_exposure = 1.0
        depth_mean, depth_min = np.float32(1.1 / m.idist[i]), np.float32(0.5 / m.idist[i])
        z_range = np.float32(1.0) / depth_min
        sd.mu = float(np.float32(1.0) / depth_mean); sd.sigma2 = float(z_range * z_range / np.float32(36)); sd.b = 10.0
        seeds.append(sd)
    pea = math.atan(1.0 / (2.0 * abs((m.cam.fx + m.cam.fy) * 0.5))) * 2.0
    so = m.ctx.seed_observe(m.cam, 42, m.T_cw, m.exposure_time, pea, seeds)
    kept = [o for o in so if o.is_valid]
    assert n_left == len(kept) and n_seed_ok == sum(o.result == 1 for o in kept) and n_seed_ok >= 16
    for k, o in enumerate(kept):
        mu, s2, b = (float(x) for x in sv[2 + 3 * k: 5 + 3 * k])
        assert (mu, s2, b) == pytest.approx((o.mu, o.sigma2, o.b), rel=1e-6)


def test_pose_optimizer_adapter_matches_cabi(mirror):
    """pose_optimizer::optimizeLevenbergMarquardt3rd (driver line 4) against hso_gpu_pose_optimize_batch."""
    import math
    m = mirror
    cam = m.cam
    pv = m.lines[3].split()
    n_fts, nobs, culled = int(pv[0]), int(pv[1]), int(pv[2])
    qp, tp = np.array(pv[3:7], float), np.array(pv[7:10], float)
    scale, e0, e1, err_px = float(pv[10]), float(pv[11]), float(pv[12]), float(pv[13])
    got = m.matches()
    matched = [k for k, g in enumerate(got) if g.success]
    assert n_fts == len(matched)
    pf = np.zeros(len(matched), capi.POSE_FEAT_DTYPE)
    for r, k in enumerate(matched):
        i = m.has_pt[k]
        g = got[k]
        x, y = (g.px_cur[0] - cam.cx) / cam.fx, (g.px_cur[1] - cam.cy) / cam.fy
        nrm = math.sqrt(x * x + y * y + 1.0)
        pf[r]["has_point"] = 1; pf[r]["type"] = capi.FTR_CORNER; pf[r]["level"] = g.search_level; pf[r]["host_pose"] = 0
        pf[r]["f"] = [x / nrm, y / nrm, 1.0 / nrm]; pf[r]["grad"] = [1.0, 0.0]
        pf[r]["host_f"] = m.feats["f"][i]; pf[r]["idist"] = m.idist[i]
    T_start = capi.SE3.from_arrays(m.q, m.t + np.array([0.004, -0.003, 0.0]))
    (rg,), (mg,) = m.ctx.pose_optimize_batch(cam, [capi.make_pose_job(pf, [capi.SE3.identity()], T_start)])
    assert (nobs, culled) == (rg.num_obs, int(mg.sum()))
    assert np.allclose(qp, rg.T_f_w.q[:], atol=1e-9) and np.allclose(tp, rg.T_f_w.t[:], atol=1e-8)
    assert (scale, e0, e1) == pytest.approx((rg.estimated_scale, rg.error_init, rg.error_final), rel=1e-6)
    assert err_px == pytest.approx(rg.error_in_px, rel=1e-5)
    # the refinement pulls the perturbed start back to the tracked pose
    assert np.linalg.norm(tp - m.t) < 1.5e-3 and e1 <= e0


def _min_thresh(m):
    kv = m.lines[4].split()
    grad_mean = float(kv[1])
    assert grad_mean == pytest.approx(m.st_c.grad_mean, rel=1e-7)
    return int(grad_mean)


def test_add_keyframe_adapter_matches_cabi(mirror):
    """DepthFilter::addKeyframe -> FeatureExtractor::detect -> seeds (driver line 5) against the C-ABI pieces."""
    m = mirror
    cam = m.cam
    mv = m.lines[1].split()
    kv = m.lines[4].split()
    n_new = int(kv[0])
    min_thresh = _min_thresh(m)
    co, cc, eo, ec = m.ctx.detect_candidates([42], n_levels=3, min_thresh=min_thresh, corner_cap=16384, edgelet_cap=4800)
    occupied = [(float(mv[2 + 4 * k]), float(mv[3 + 4 * k])) for k in range(m.K) if int(mv[1 + 4 * k])]
    keys = np.zeros(len(occupied) + int(cc.sum() + ec.sum()), capi.KEYPOINT_DTYPE)
    keys["x"][:len(occupied)] = [p[0] for p in occupied]
    keys["y"][:len(occupied)] = [p[1] for p in occupied]
    keys["species"][:len(occupied)] = capi.KP_OCCUR
    at = len(occupied)
    for L in range(3):
        c, e = co[0, L, :cc[0, L]], eo[0, L, :ec[0, L]]
        k = keys[at:at + len(c)]
        k["x"], k["y"], k["response"], k["level"], k["species"] = c["x"].astype(np.int32) << L, c["y"].astype(np.int32) << L, c["response"], L, capi.KP_CORNER_HIGH
        at += len(c)
        k = keys[at:at + len(e)]
        k["x"], k["y"], k["response"], k["level"], k["species"] = e["x"].astype(np.int32) << L, e["y"].astype(np.int32) << L, e["grad"], L, capi.KP_EDGELET
        k["gx"], k["gy"] = e["gx"], e["gy"]
        at += len(e)
    sel = capi.select_octree(keys, 640, 480, 300)
    assert n_new == len(sel) and 150 <= n_new <= 303     # ~300 nodes minus those an existing feature occupies
    rec = np.array(kv[2:], float).reshape(n_new, 11)
    n_edgelets = 0
    for r, s in zip(rec, sel):
        is_corner = s["species"] == capi.KP_CORNER_HIGH
        assert (int(r[0]), int(r[1])) == (capi.FTR_CORNER if is_corner else capi.FTR_EDGELET, s["level"])
        assert (r[2], r[3]) == (s["x"], s["y"])
        if is_corner:
            assert (r[4], r[5]) == (1.0, 0.0)
        else:
            g = np.array([s["gx"], s["gy"]], float)
            assert np.allclose(r[4:6], g / np.linalg.norm(g), atol=1e-15)
            n_edgelets += 1
        f = np.array([(s["x"] - cam.cx) / cam.fx, (s["y"] - cam.cy) / cam.fy, 1.0])
        assert np.allclose(r[6:9], f / np.linalg.norm(f), atol=1e-12)
        # Seed(ftr, depth_mean 2.0, depth_min 0.5): mu = 1/2, sigma2 = (1/0.5)^2 / 36
        assert r[9] == 0.5 and r[10] == pytest.approx(4.0 / 36.0, rel=1e-6)
    # (on this densely textured frame every node holds a corner, and corners outrank edgelets:
    # n_edgelets is normally 0 here; tests/test_octree.py covers the edgelet-winning nodes)
    assert 0 <= n_edgelets <= n_new


def test_init_detect_adapter_matches_cabi(mirror):
    """FeatureExtractor(isInit=true).detect (driver line 6): fastDetectMT + fillingHole + oct-tree with 2000 features."""
    assert _init_branch_check(mirror.ctx, mirror.lines[5], 42, _min_thresh(mirror)) >= 0


@pytest.mark.parametrize("line,budget", [(6, 200), (7, 40)], ids=["reprojectCellAll", "three_cell_passes"])
def test_reprojector_adapter_matches_cabi(mirror, line, budget):
    """Reprojector::reprojectMap (driver lines 7-8): few candidates -> reprojectCellAll; small budget -> the three cell passes."""
    m = mirror
    n_sel, n_cand = _reprojector_check(m.ctx, m.cam, m.lines[line], budget, 41, 42, m.T_cw, m.exposure_time, m.feats, m.idist, m.has_pt)
    if budget == 200:
        assert n_cand > 150 and n_sel > 120
    else:
        assert n_sel == 40


def _init_branch_check(gpu_ctx, line, frame_id, min_thresh):
    """Line 6 of the driver: FeatureExtractor(isInit=true).detect against the C-ABI pieces."""
    v = line.split()
    n = int(v[0])
    rec = np.array(v[1:], float).reshape(n, 4)
    co, cc, fo, fc = gpu_ctx.detect_candidates_init([frame_id], n_levels=3, min_thresh=min_thresh, corner_cap=16384, fill_cap=4800)
    parts = []
    for L in range(3):
        c = co[0, L, :cc[0, L]]
        k = np.zeros(len(c), capi.KEYPOINT_DTYPE)
        k["x"], k["y"], k["response"], k["level"], k["species"] = c["x"].astype(np.int32) << L, c["y"].astype(np.int32) << L, c["response"], L, capi.KP_CORNER_HIGH
        parts.append(k)
        if L == 0:
            f = fo[0, :fc[0]]
            k = np.zeros(len(f), capi.KEYPOINT_DTYPE)
            k["x"], k["y"], k["response"], k["level"], k["species"] = f["x"], f["y"], f["response"], 0, capi.KP_GRAD
            parts.append(k)
    sel = capi.select_octree(np.concatenate(parts), 640, 480, 2000)
    assert n == len(sel) and n > 1500
    want_type = np.where(sel["species"] == capi.KP_CORNER_HIGH, 0, 2)       # Feature::CORNER / GRADIENT
    assert (rec[:, 0] == want_type).all() and (rec[:, 1] == sel["level"]).all()
    assert (rec[:, 2] == sel["x"]).all() and (rec[:, 3] == sel["y"]).all()
    return int((want_type == 2).sum())


def _reprojector_check(gpu_ctx, cam, line, budget, kf_id, cur_id, T_cw, exposure_time, feats, idist, has_pt):
    """Lines 7-8 of the driver: hso::Reprojector::reprojectMap against hso_gpu_reproject_match + the
    cell passes of src/reprojector.cpp:261-306 / :352-429 / :556-612 restated here."""
    v = line.split()
    n_matches, n_trials, n_feat, n_overlap, cell_size, gcols = (int(x) for x in v[:6])
    rec = np.array(v[6:], float).reshape(-1, 4)
    assert cell_size == int(np.floor(np.float32(np.sqrt(np.float32(640 * 480) / budget)) * 0.6))
    assert gcols == int(np.ceil(640 / cell_size))
    n_cells = gcols * int(np.ceil(480 / cell_size))
    kfs = np.zeros(1, capi.KF_DTYPE)
    kfs[0]["frame_id"], kfs[0]["q"], kfs[0]["exposure_time"], kfs[0]["keyframe_id"] = kf_id, [0, 0, 0, 1], 1.0, 0
    pts = np.zeros(len(has_pt), capi.MAP_POINT_DTYPE)
    obs = np.zeros(len(has_pt), capi.OBS_DTYPE)
    for r, i in enumerate(has_pt):
        pts[r]["pos"] = feats["f"][i] * (1.0 / idist[i])
        pts[r]["idist"], pts[r]["host_f"], pts[r]["host_kf"] = idist[i], feats["f"][i], 0
        pts[r]["obs_begin"], pts[r]["obs_count"] = r, 1
        obs[r]["kf"], obs[r]["level"], obs[r]["type"] = 0, 0, capi.FTR_CORNER
        obs[r]["px"], obs[r]["f"], obs[r]["grad"] = feats["px"][i], feats["f"][i], [1.0, 0.0]
    proj, match = gpu_ctx.reproject_match(cam, cur_id, T_cw, exposure_time, 0, kfs, pts, obs, cell_size, gcols)
    ok = lambda i: proj["ref_obs"][i] >= 0 and match[i].success == 1
    cand = [i for i in range(len(pts)) if proj["projected"][i]]
    assert n_feat == n_overlap == len(cand)
    sel, trials = [], 0
    if len(cand) < budget + 50:
        for i in cand:
            trials += 1
            if ok(i):
                sel.append(i)
                if len(sel) >= budget:
                    break
    else:
        cells = [[] for _ in range(n_cells)]
        for i in cand:
            cells[proj["cell"][i]].append(i)
        for c in range(n_cells):                                   # 1st pass: one match per cell
            while cells[c]:
                trials += 1
                i = cells[c].pop(0)
                if ok(i):
                    sel.append(i)
                    break
            if len(sel) >= budget:
                break
        if len(sel) < budget:
            for c in range(n_cells - 1, 0, -1):                    # 2nd pass, backwards, cell 0 skipped
                while cells[c]:
                    trials += 1
                    i = cells[c].pop(0)
                    if ok(i):
                        sel.append(i)
                        break
                if len(sel) >= budget:
                    break
        if len(sel) < budget:
            for c in range(n_cells):                               # 3rd pass: up to three more per cell
                got = 0
                while cells[c]:
                    trials += 1
                    i = cells[c].pop(0)
                    if ok(i):
                        sel.append(i); got += 1
                        if got >= 3 or len(sel) >= budget:
                            break
                if len(sel) >= budget:
                    break
    assert (n_matches, n_trials) == (len(sel), trials), (budget, n_matches, len(sel), n_trials, trials)
    assert [int(x) for x in rec[:, 0]] == sel
    # the same policy evaluated on the device (hso_gpu_reproject_select): the mirror visits the cells in index order here
    flags = np.array([1 if ok(i) else 0 for i in cand], np.uint8)
    ex, counts = gpu_ctx.reproject_select([0, len(cand)], proj["cell"][cand], np.full(len(cand), (3 << 4) | 0, np.uint8), flags,
                                          np.arange(n_cells, dtype=np.int32), budget)
    assert (int(counts[0, 1]), int(counts[0, 0])) == (n_matches, n_trials)
    assert [cand[j] for j, taken in ex[0] if taken] == sel
    for r, i in zip(rec, sel):
        assert int(r[1]) == match[i].search_level
        assert (r[2], r[3]) == pytest.approx((match[i].px_cur[0], match[i].px_cur[1]), abs=1e-9)
    return len(sel), len(cand)
