"""Replays the engine's device-call trace (hso_amd/host/hso_engine.h: Trace) through the CPU restatement: one checker per recorded
call, each with exactly the inputs the product saw (per-call replay is the unit of parity).  Used by
tests/test_chain_gpu.py (the engine on the GPU) and tests/test_engine_cpu.py (the engine over the restatement itself, where
every comparison must hold exactly: that run checks the trace and this file, not the kernels)."""
import ctypes as C

import numpy as np
import pytest

from hso_amd import capi, vo


def _arr(ctype, raw):
    n = len(raw) // C.sizeof(ctype)
    return (ctype * max(n, 1)).from_buffer_copy(raw if n else bytes(C.sizeof(ctype))), n


def _rot_err(qa, qb):
    return 2 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


class Replayer:
    """Feeds the recorded calls to the oracle; counts compared items and near-tie escapes per stage."""

    def __init__(self, orc):
        self.orc = orc
        self.frames = {}
        self.stat = {}
        lib = orc.load()
        lib.hso_or_compute_tau.argtypes = [C.POINTER(capi.SE3), C.c_void_p, C.c_double, C.c_double]
        lib.hso_or_compute_tau.restype = C.c_double
        self.tau = lib.hso_or_compute_tau
        lib.hso_or_update_seed.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.hso_or_update_seed.restype = None
        self.upd = lib.hso_or_update_seed

    def bump(self, stage, key, n=1):
        self.stat.setdefault(stage, {}).setdefault(key, 0)
        self.stat[stage][key] += n

    def frame(self, fid):
        f = self.frames[int(fid)]
        if "pyr" not in f:
            f["pyr"] = self.orc.create_pyramid(f["img"])
            f["sobel"] = [self.orc.sobel5(np.ascontiguousarray(f["pyr"][L])) for L in range(3)]
        return f

    # ---- one checker per recorded call
    def frame_upload(self, r):
        w, h = int(vo.scalar(r, "width")), int(vo.scalar(r, "height"))
        img = np.frombuffer(r["img"], np.uint8).reshape(h, w).copy()
        self.frames[int(vo.scalar(r, "frame_id"))] = {"img": img}
        st = capi.FrameStats.from_buffer_copy(r["stats"])
        f = self.frame(vo.scalar(r, "frame_id"))
        so = self.orc.frame_stats(f["pyr"][0], *f["sobel"][0])
        # the reference sums ~3.4e5 pixels serially in fp32 (src/frame.cpp:223-238): its own rounding walk is ~3e-5 relative;
        # the device sums exactly.  On a textureless image (tests/test_relocalise.py) the serial sum's rounding is systematic instead
        # of a random walk — above 2^24 every addend of 117 rounds the same way — and the reference's own mean is off by up to 1 %
        rel = 2e-4 if float(img.std()) > 1.0 else 1e-2
        assert st.integral_image == pytest.approx(so.integral_image, rel=rel) and st.grad_mean == pytest.approx(so.grad_mean, rel=rel)
        self.bump("frame", "n")

    def klt_track(self, r):
        """initialization::trackKlt's device call (tests/test_klt.py states the bar: status equal, position within 2e-3 px unless the
        restatement's own decision margin was below 1e-3; the patch check restated on the device's position)."""
        kp = capi.KltParams.from_buffer_copy(r["params"])
        prev, cur = self.frames[int(vo.scalar(r, "prev_frame_id"))]["img"], self.frames[int(vo.scalar(r, "cur_frame_id"))]["img"]
        a = np.frombuffer(r["px_prev"], np.float32).reshape(-1, 2); b = np.frombuffer(r["px_init"], np.float32).reshape(-1, 2)
        g = np.frombuffer(r["result"], capi.KLT_RESULT_DTYPE)
        o_px, o_st, o_mg = self.orc.klt_track(prev, cur, a, b, kp.win_size, kp.max_level, kp.max_iter, kp.epsilon, bool(kp.use_initial_flow))
        excused = o_mg < 1e-3
        tracked = (g["status"] & capi.KLT_TRACKED) != 0
        assert not ((tracked != (o_st > 0)) & ~excused).any()
        both = tracked & (o_st > 0)
        assert not (both & ~excused & (np.abs(g["px"] - o_px).max(axis=1) > 2e-3)).any()
        for i in np.flatnonzero(both):
            ok, ncc = self.orc.patch_check(prev, cur, a[i], g["px"][i])
            assert abs(ncc - g["ncc"][i]) <= 1e-5
            assert ok == bool(g["status"][i] & capi.KLT_PATCH_OK) or abs(ncc - 0.8) < 1e-4
        self.bump("klt", "n"); self.bump("klt", "points", len(a)); self.bump("klt", "excused", int(excused.sum()))

    def coarse_track(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"]); p = capi.TrackParams.from_buffer_copy(r["params"])
        feats = np.frombuffer(r["feats"], capi.REF_FEAT_DTYPE)
        ref, cur = self.frame(vo.scalar(r, "ref_frame_id")), self.frame(vo.scalar(r, "cur_frame_id"))
        g = capi.TrackResult.from_buffer_copy(r["result"])
        T0 = capi.SE3.from_buffer_copy(r["T_cur_ref"])
        a0 = float(np.float32(vo.scalar(r, "exposure_rat")))
        o = self.orc.Tracker(cam, p, ref["pyr"], cur["pyr"], feats).run(T0, a0)
        qg, tg = g.T_cur_ref.to_arrays(); qo, to = o.T_cur_ref.to_arrays()
        self.bump("track", "n")
        if p.min_level == 0 and p.n_iter == 15:
            self.bump("track", "reloc")     # relocalizeFrame's tracker (src/frame_handler_mono.cpp:366: levels 4..0, 15 iterations)
        seq = lambda x: (list(x.iters), list(x.accept_mask))
        if seq(g) == seq(o):
            assert _rot_err(qg, qo) <= 2e-6 and np.linalg.norm(tg - to) <= 8e-6
            assert g.n_tracked == pytest.approx(o.n_tracked, abs=2)
            self.track_dev = max(getattr(self, "track_dev", 0.0), _rot_err(qg, qo), float(np.linalg.norm(tg - to)))
            return
        # A different LM accept sequence is excused by margin, not by count (tests/test_parity_gpu.py::
        # test_accept_decisions_over_many_scenes): the device sums the bit-identical fp32 energy terms in a tree, the reference
        # serially in fp32, so only an accept decision inside the serial sum's own rounding noise may differ.  Evidence required
        # per call: the device equals the restatement that decides on the fp64 sum of the same terms (sequence and pose), and the
        # serial-sum restatement itself differs from that form.
        self.bump("track", "iter_mismatch")
        t64 = self.orc.Tracker(cam, p, ref["pyr"], cur["pyr"], feats); t64.decide_on_f64_sum(True)
        self.orc.margins_reset()
        r64 = t64.run(T0, a0)
        m64 = self.orc.margins()
        q6, t6 = r64.T_cur_ref.to_arrays()
        if seq(g) == seq(r64):
            assert seq(o) != seq(r64), (seq(g), seq(o), seq(r64))
            assert _rot_err(qg, q6) <= 2e-7 and np.linalg.norm(tg - t6) <= 8e-7
            assert g.n_tracked == pytest.approx(r64.n_tracked, abs=2)
        else:
            # differs from the exact-sum form as well: only where that form itself met an accept test (energy_new < energy_old on
            # float quotients, CoarseTracker.cpp:143) whose two energies agree to within 10x the rounding of the device's fp32 per-feature
            # partial sums (3e-7): a decision no arithmetic pins
            self.bump("track", "accept_tie")
            assert m64.track_accept < 3e-6, (seq(g), seq(o), seq(r64), m64.track_accept)
        # both converge to the same minimum — where there is one: against a textureless image (tests/test_relocalise.py: the frames
        # that make the handler lose track) the photometric energy is flat and two runs that part at a tie go where rounding takes them
        if float(cur["img"].std()) > 1.0 and float(ref["img"].std()) > 1.0:
            assert _rot_err(qg, qo) <= 1e-4 and np.linalg.norm(tg - to) <= 1e-4

    def reproject_match(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"])
        kfs = np.frombuffer(r["kfs"], capi.KF_DTYPE); pts = np.frombuffer(r["points"], capi.MAP_POINT_DTYPE)
        obs = np.frombuffer(r["obs"], capi.OBS_DTYPE)
        proj = np.frombuffer(r["proj"], capi.REPROJ_POINT_DTYPE); match, _ = _arr(capi.AlignOut, r["match"])
        self.last_proj, self.last_match = proj, match
        cur = self.frame(vo.scalar(r, "cur_frame_id"))
        T = capi.SE3.from_buffer_copy(r["T_cur_w"])
        kf_pyrs = [self.frame(k["frame_id"])["pyr"] for k in kfs]
        margins = []
        wproj, wmatch = self.orc.reproject_match(cam, T, vo.scalar(r, "cur_exposure"), int(vo.scalar(r, "cur_keyframe_id")), kfs, pts, obs,
                                                 int(vo.scalar(r, "cell_size")), int(vo.scalar(r, "grid_n_cols")), kf_pyrs, cur["pyr"], cur["sobel"],
                                                 margins_out=margins)
        radtan = cam.model == capi.CAM_PINHOLE and cam.distortion
        # radtan: cam2world runs OpenCV's five fp32 undistortion iterations (src/camera.cpp:171-194); host and device round them
        # differently at the 1e-7 level of the bearing, so A_cur_ref — and with it the warped patch — carries a 2e-5 tolerance
        # instead of 1e-8; the decision margins scale with it
        k = 5.0 if radtan else 1.0
        for i in range(len(pts)):
            g, w = proj[i], wproj[i]
            self.bump("reproject", "points")
            if g["projected"] != w["projected"] or (g["projected"] and g["cell"] != w["cell"]):
                px = w["px"] if w["projected"] else g["px"]
                assert min(abs(px[0] - round(px[0])), abs(px[1] - round(px[1]))) < 1e-6, i   # only on a pixel / cell border
                self.bump("reproject", "tie")
                continue
            if not g["projected"]:
                continue
            assert np.allclose(g["px"], w["px"], atol=1e-8, rtol=0) and g["ref_obs"] == w["ref_obs"]
            if g["ref_obs"] < 0:
                continue
            m, o, mg = match[i], wmatch[i], margins[i]
            self.bump("reproject", "matched_calls")
            assert m.search_level == o.search_level
            assert np.allclose(m.A_cur_ref[:], o.A_cur_ref[:], atol=2e-5 if radtan else 1e-8)
            if o.stage == 1:
                assert m.stage == 1 and not m.success
                continue
            # the margin rule of tests/test_align.py: a differing decision only where the restatement's own comparison was within
            # 10x the tolerance of the compared quantity
            if m.iters != o.iters:
                assert mg.lk_update < 1e-2 * k, (i, m.iters, o.iters, mg.lk_update)
                self.bump("reproject", "tie")
                continue
            if (m.success, m.stage) != (o.success, o.stage):
                assert min(mg.ncc / 1e-3, mg.normal / 1e-3, mg.lk_chi2 / 1e-2, mg.lk_update / 1e-2, mg.jump / 1e-2) < k, \
                    (i, m.stage, o.stage, [getattr(mg, f) for f in self.orc.MARGIN_FIELDS])
                self.bump("reproject", "tie")
                continue
            if o.success:
                assert np.allclose(m.px_cur[:], o.px_cur[:], atol=2e-3 * (1 << m.search_level))
                self.bump("reproject", "success")

    def reproject_select(self, r):
        """The grid selection of the chained call against the sequential walk (tests/test_select.py: select_reference) over the
        projection / match results of the reproject_match record just before it: the examined candidates, their order, which
        became features, the counters; then the feature table the device built from them (next record) is checked in pose_optimize."""
        from test_select import select_reference
        proj, match = self.last_proj, self.last_match
        quality = np.frombuffer(r["quality"], np.uint8)
        order = np.frombuffer(r["cell_order"], np.int32)
        ex = np.frombuffer(r["examined"], capi.MATCH_BRIEF_DTYPE)
        counts = np.frombuffer(r["counts"], np.int32)
        projected = np.frombuffer(r["projected"], np.uint8)
        assert np.array_equal(projected != 0, proj["projected"] != 0)
        idx = np.flatnonzero(proj["projected"])
        cell = proj["cell"][idx].astype(np.int32)
        flags = np.array([1 if (proj["ref_obs"][i] >= 0 and match[i].success) else 0 for i in idx], np.uint8) | ((quality[idx] >> 4) == 0).astype(np.uint8) << 1
        want, n_match, passes = select_reference(cell, quality[idx], flags, order, len(order), int(vo.scalar(r, "max_fts")))
        assert len(ex) == len(want) == counts[0] and n_match == counts[1] and passes == counts[2]
        for e, (k, taken) in zip(ex, want):
            assert e["pad_"] == idx[k] and bool(e["success"]) == taken
        self.bump("select", "n"); self.bump("select", "examined", len(ex)); self.bump("select", "taken", int(n_match))
        self.last_examined = ex


    def pose_optimize(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"])
        feats = np.frombuffer(r["feats"], capi.POSE_FEAT_DTYPE).copy()
        poses, n = _arr(capi.SE3, r["poses"])
        job = capi.make_pose_job(feats, list(poses[:n]), capi.SE3.from_buffer_copy(r["T_f_w"]), vo.scalar(r, "reproj_thresh"), int(vo.scalar(r, "n_iter")))
        g = capi.PoseResult.from_buffer_copy(r["result"]); gmask = np.frombuffer(r["mask"], np.uint8)
        self.orc.margins_reset()
        o, omask = self.orc.pose_optimize(cam, job)
        mg = self.orc.margins()
        qg, tg = g.T_f_w.to_arrays(); qo, to = o.T_f_w.to_arrays()
        self.bump("pose", "n")
        same = (g.iters, g.n_trials_total) == (o.iters, o.n_trials_total)
        if not same:
            # tests/test_pose.py: once converged rho = chi2 - new_chi2 is rounding noise; only then may the serial and the tree sums
            # accept / reject a last no-op step differently — the restatement itself must have seen |rho| / chi2 < 1e-12
            assert mg.pose_rho < 1e-12 and abs(g.iters - o.iters) <= 2 and abs(g.n_trials_total - o.n_trials_total) <= 6, mg.pose_rho
            self.bump("pose", "iter_mismatch")
        assert _rot_err(qg, qo) <= 1e-7 and np.linalg.norm(tg - to) <= 1e-7
        assert g.status == o.status and g.num_obs == o.num_obs
        assert np.array_equal(gmask, omask) and g.estimated_scale == pytest.approx(o.estimated_scale, rel=1e-6)

    def seed_observe(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"])
        seeds, n = _arr(capi.Seed, r["seeds"]); got, _ = _arr(capi.SeedOut, r["out"])
        cur = self.frame(vo.scalar(r, "cur_frame_id")); T = capi.SE3.from_buffer_copy(r["T_f_w"])
        # radtan: cam2world of the matched pixel runs OpenCV's five fp32 undistortion iterations (src/camera.cpp:78-85); host and
        # device round them differently at the 1e-7 level of the bearing, which the triangulation amplifies by depth / baseline
        kz = 30.0 if (cam.model == capi.CAM_PINHOLE and cam.distortion) else 1.0
        kz_dec = 5.0 if kz > 1 else 1.0
        for i in range(n):
            s, g = seeds[i], got[i]
            self.orc.margins_reset()
            o = self.orc.seed_observe(cam, s, T, vo.scalar(r, "exposure"), vo.scalar(r, "px_error_angle"), self.frame(s.ref_frame_id)["pyr"],
                                      cur["pyr"], cur["sobel"])
            mg = self.orc.margins()
            self.bump("seed", "n")
            assert g.is_update == o.is_update and g.is_valid == o.is_valid
            if o.result == 0:
                assert g.result == 0 and g.mu == o.mu and g.sigma2 == o.sigma2
                continue
            assert g.search_level == o.search_level
            if o.result == -1 or g.result == -1:
                assert g.result == o.result
                continue
            if g.n_steps != o.n_steps:
                # the epipolar march (src/matcher.cpp:893-1000) walks unit steps from px_far - inc to px_close + inc and stops when the
                # position passes px_close; a segment shorter than 2 px is padded to exactly 4 units, so in exact arithmetic the last step
                # lands ON the end point and the reference's own `>` there is decided by rounding.  Excused only when the restatement
                # saw that: a tested position within 1e-9 px of the end point; then one sample more or less, same best score.
                assert abs(g.n_steps - o.n_steps) == 1 and mg.march_end < 1e-9, (g.n_steps, o.n_steps, mg.march_end)
                self.bump("seed", "march_end_tie")
                if g.zmncc_best != pytest.approx(o.zmncc_best, abs=1e-4):
                    continue      # the sample only one side visited was the best one: everything downstream follows from that tie
            elif o.n_steps > 0 and o.zmncc_best > 0.1:
                assert g.zmncc_best == pytest.approx(o.zmncc_best, abs=1e-4)
            if g.result != o.result:
                # tests/test_seed.py's rule: the gate that separates the two codes had its operands within 10x their tolerance
                codes = {g.result, o.result}
                near = False
                if codes == {-3, -4} or codes == {1, -4}:
                    near = min(mg.zmncc_best, mg.zmncc_ambig, mg.zmncc_order) < 1e-3
                if codes == {1, -3}:
                    near = min(mg.klt_energy / 1e-2, mg.klt_accept / 1e-2, mg.klt_step / 1e-1, mg.ncc / 1e-3, mg.normal / 1e-3) < kz_dec
                assert near, (i, g.result, o.result, [getattr(mg, f) for f in self.orc.MARGIN_FIELDS])
                self.bump("seed", "tie")
                continue
            if o.result == 1:
                # the matched pixel against the restatement's (the LK tolerance of tests/test_align.py); the depth against the
                # restatement's triangulation (src/matcher.cpp:242-255) of the device's own pixel — with a baseline of one frame,
                # d ln z / d px is ~1e-1 per pixel, so comparing z across the two pixels would only re-measure the LK tolerance
                assert np.allclose(list(g.px_cur), list(o.px_cur), atol=2e-3 * (1 << g.search_level), rtol=0)   # 2e-3 px on the search level
                T_cur_ref = self.orc.se3_mul(T, self.orc.se3_inverse(s.T_ref_w))
                fc = self.orc.cam2world(cam, g.px_cur[0], g.px_cur[1])
                a0 = self.orc.so3_matrix(np.array(T_cur_ref.q[:])) @ np.array(s.f[:]); a1 = fc
                m00, m01, m11 = a0 @ a0, a0 @ a1, a1 @ a1
                inv = 1.0 / (m00 * m11 - m01 * m01)
                z_at_g = abs(((-m11 * inv) * a0 + (m01 * inv) * a1) @ np.array(T_cur_ref.t[:]))
                assert g.z == pytest.approx(z_at_g, rel=1e-5 * kz)
                # computeTau (src/depth_filter.cpp:539-555) is z_plus - z with z_plus = |t| sin(beta+) / sin(pi - alpha - beta+): one
                # frame after a keyframe the parallax is about the pixel angle, the denominator passes through zero and d ln(tau^2) /
                # d ln(z) reaches several hundred.  So the Gaussian update is checked for what it is — the restatement's computeTau +
                # updateSeed evaluated at the device's own triangulated z must give the device's mu and sigma2 — and z itself against
                # the restatement's z above.
                T_ref_cur = self.orc.se3_mul(s.T_ref_w, self.orc.se3_inverse(T))
                f3 = np.array(s.f[:], float)
                tau = self.tau(C.byref(T_ref_cur), f3.ctypes.data, g.z, vo.scalar(r, "px_error_angle"))
                tau_inverse = 0.5 * (1.0 / max(0.0000001, g.z - tau) - 1.0 / (g.z + tau))
                mu, sigma2 = C.c_float(s.mu), C.c_float(s.sigma2)
                self.upd(1. / g.z, tau_inverse * tau_inverse, C.byref(mu), C.byref(sigma2))
                assert g.mu == pytest.approx(mu.value, rel=1e-5) and g.sigma2 == pytest.approx(sigma2.value, rel=1e-4)
                self.bump("seed", "updated")
            else:
                assert g.mu == o.mu and g.sigma2 == o.sigma2 and g.b == o.b

    def seed_observe_previous(self, r):
        """observeDepthWithPreviousFrameOnce: the seeds of one keyframe against the earlier frame they observed (tests/test_seed.py's rules:
        the march is bit-equal, codes differ only at the refinement's gates, excused by margin; b is never touched)"""
        cam = capi.Camera.from_buffer_copy(r["cam"])
        seeds, n = _arr(capi.Seed, r["seeds"]); got, _ = _arr(capi.SeedOut, r["out"])
        pre = self.frame(vo.scalar(r, "pre_frame_id")); T = capi.SE3.from_buffer_copy(r["T_f_w"])
        kz = 30.0 if (cam.model == capi.CAM_PINHOLE and cam.distortion) else 1.0
        kz_dec = 5.0 if kz > 1 else 1.0
        for i in range(n):
            s, g = seeds[i], got[i]
            self.orc.margins_reset()
            o = self.orc.seed_observe_previous(cam, s, T, vo.scalar(r, "exposure"), vo.scalar(r, "px_error_angle"), self.frame(s.ref_frame_id)["pyr"],
                                               pre["pyr"], pre["sobel"])
            mg = self.orc.margins()
            self.bump("seed_previous", "n")
            assert g.is_update == o.is_update and g.b == s.b
            if o.is_update == 0:
                assert g.result == 0 and g.mu == s.mu and g.sigma2 == s.sigma2
                continue
            assert g.search_level == o.search_level and g.n_steps == o.n_steps
            if o.result == -1 or g.result == -1:
                assert g.result == o.result
                continue
            if kz == 1.0:
                assert g.zmncc_best == o.zmncc_best and g.zmncc_second == o.zmncc_second     # the same host patch: bit-equal march
            elif o.n_steps > 0 and o.zmncc_best > 0.1:
                # radtan: the warp matrix goes through cam2world's fp32 undistortion iterations, which host and device round differently
                # (see seed_observe above): the host patch differs at the 1e-6 level and the scores with it
                assert g.zmncc_best == pytest.approx(o.zmncc_best, abs=1e-4)
            if g.result != o.result:
                codes = {g.result, o.result}
                near = False
                if kz > 1.0 and (codes == {-3, -4} or codes == {1, -4}):
                    near = min(mg.zmncc_best, mg.zmncc_ambig, mg.zmncc_order) < 1e-3
                if codes == {1, -3}:
                    near = min(mg.klt_energy / 1e-2, mg.klt_accept / 1e-2, mg.klt_step / 1e-1, mg.ncc / 1e-3, mg.normal / 1e-3) < kz_dec
                assert near, (i, g.result, o.result, [getattr(mg, f) for f in self.orc.MARGIN_FIELDS])
                self.bump("seed_previous", "tie")
                continue
            if o.result == 1:
                assert np.allclose(list(g.px_cur), list(o.px_cur), atol=2e-3 * (1 << g.search_level), rtol=0)
                T_cur_ref = self.orc.se3_mul(T, self.orc.se3_inverse(s.T_ref_w))
                fc = self.orc.cam2world(cam, g.px_cur[0], g.px_cur[1])
                a0 = self.orc.so3_matrix(np.array(T_cur_ref.q[:])) @ np.array(s.f[:]); a1 = fc
                m00, m01, m11 = a0 @ a0, a0 @ a1, a1 @ a1
                inv = 1.0 / (m00 * m11 - m01 * m01)
                z_at_g = abs(((-m11 * inv) * a0 + (m01 * inv) * a1) @ np.array(T_cur_ref.t[:]))
                assert g.z == pytest.approx(z_at_g, rel=1e-5 * kz)
                T_ref_cur = self.orc.se3_mul(s.T_ref_w, self.orc.se3_inverse(T))
                f3 = np.array(s.f[:], float)
                tau = self.tau(C.byref(T_ref_cur), f3.ctypes.data, g.z, vo.scalar(r, "px_error_angle"))
                tau_inverse = 0.5 * (1.0 / max(0.0000001, g.z - tau) - 1.0 / (g.z + tau))
                mu, sigma2 = C.c_float(s.mu), C.c_float(s.sigma2)
                self.upd(1. / g.z, tau_inverse * tau_inverse, C.byref(mu), C.byref(sigma2))
                assert g.mu == pytest.approx(mu.value, rel=1e-5) and g.sigma2 == pytest.approx(sigma2.value, rel=1e-4)
                self.bump("seed_previous", "updated")
            else:
                assert g.mu == s.mu and g.sigma2 == s.sigma2

    def seed_activate(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"])
        seeds, n = _arr(capi.Seed, r["seeds"]); got, _ = _arr(capi.ActivateOut, r["out"])
        begin = np.frombuffer(r["target_begin"], np.int32); tg, _ = _arr(capi.ActivateTarget, r["targets"])
        n_mean = int(vo.scalar(r, "n_mean_converge_frame"))
        kd = 5.0 if (cam.model == capi.CAM_PINHOLE and cam.distortion) else 1.0
        for i in range(n):
            tl = [tg[k] for k in range(begin[i], begin[i + 1])]
            fr = [self.frame(t.frame_id) for t in tl]
            self.orc.margins_reset()
            o, _ = self.orc.seed_activate(cam, seeds[i], tl, self.frame(seeds[i].ref_frame_id)["pyr"], [f["pyr"] for f in fr],
                                          [f["sobel"] for f in fr], n_mean)
            mg = self.orc.margins()
            g = got[i]
            self.bump("activate", "n")
            assert g.n_targets == o.n_targets
            if g.n_matched != o.n_matched:
                # one of the seed's findMatchSeed calls decided differently: only with a gate of that matcher inside its tolerance
                # (the margins of the restatement's run over all targets of this seed)
                assert min(mg.ncc / 1e-3, mg.normal / 1e-3, mg.lk_chi2 / 1e-2, mg.lk_update / 1e-2, mg.jump / 1e-2) < kd, \
                    (i, g.n_matched, o.n_matched, [getattr(mg, f) for f in self.orc.MARGIN_FIELDS])
                self.bump("activate", "tie")
                continue
            if o.n_matched >= 1:
                assert g.dist_mean == pytest.approx(o.dist_mean, abs=2e-3 * kd)
            if min(abs(o.dist_mean - t) for t in (2.0, 2.5, 3.2)) < 1e-2 * kd:     # the drift gates (depth_filter.cpp:846-870), 10x the tolerance above
                self.bump("activate", "tie")
                continue
            assert g.is_valid == o.is_valid and g.activated == o.activated
            if o.activated:
                assert g.opt_id == pytest.approx(o.opt_id, rel=1e-3)
                self.bump("activate", "activated")

    def seed_reproject_match(self, r):
        cam = capi.Camera.from_buffer_copy(r["cam"])
        seeds, n = _arr(capi.Seed, r["seeds"]); match, _ = _arr(capi.AlignOut, r["match"])
        proj = np.frombuffer(r["proj"], capi.REPROJ_POINT_DTYPE)
        cur = self.frame(vo.scalar(r, "cur_frame_id"))
        t = capi.ActivateTarget()
        t.frame_id, t.T_f_w, t.exposure = int(vo.scalar(r, "cur_frame_id")), capi.SE3.from_buffer_copy(r["T_f_w"]), vo.scalar(r, "exposure")
        kd = 5.0 if (cam.model == capi.CAM_PINHOLE and cam.distortion) else 1.0
        for i in range(n):
            # findMatchSeed of one (seed, frame) pair = the oracle's activation matcher with a single target
            self.orc.margins_reset()
            o, mo = self.orc.seed_activate(cam, seeds[i], [t], self.frame(seeds[i].ref_frame_id)["pyr"], [cur["pyr"]], [cur["sobel"]], 6)
            mg = self.orc.margins()
            self.bump("seed_reproject", "n")
            assert int(proj[i]["projected"]) == o.n_targets
            if not o.n_targets:
                continue
            if float(cur["img"].std()) <= 1.0:
                # a textureless current frame (tests/test_relocalise.py): its pose is wherever the tracker drifted on a flat
                # energy, the warp of a seed seen from there is degenerate (NaN / huge determinants), and nothing can match
                assert not match[i].success
                continue
            assert match[i].search_level == mo[0].search_level
            if match[i].success != mo[0].success:
                assert min(mg.ncc / 1e-3, mg.normal / 1e-3, mg.lk_chi2 / 1e-2, mg.lk_update / 1e-2, mg.jump / 1e-2) < kd, \
                    (i, [getattr(mg, f) for f in self.orc.MARGIN_FIELDS])
                self.bump("seed_reproject", "tie")
                continue
            if mo[0].success:
                assert np.allclose(match[i].px_cur[:], mo[0].px_cur[:], atol=2e-3)

    def ba_huber_deltas(self, r):
        poses, n = _arr(capi.SE3, r["poses"])
        hc, he = self.orc.ba_huber_deltas(list(poses[:n]), np.frombuffer(r["idist"], np.float64), np.frombuffer(r["edges"], capi.BA_EDGE_DTYPE),
                                          np.frombuffer(r["obs_uv"], np.float64), vo.scalar(r, "error_multiplier2"))
        assert np.float32(vo.scalar(r, "huber_corner")) == np.float32(hc) and np.float32(vo.scalar(r, "huber_edge")) == np.float32(he)
        self.bump("ba", "deltas")

    def ba_optimize(self, r):
        poses, n = _arr(capi.SE3, r["poses_in"]); pg, _ = _arr(capi.SE3, r["poses_out"])
        fixed = np.frombuffer(r["fixed"], np.uint8); edges = np.frombuffer(r["edges"], capi.BA_EDGE_DTYPE)
        po, io, co, ro = self.orc.ba_optimize(list(poses[:n]), fixed, np.frombuffer(r["idist_in"], np.float64), edges, vo.scalar(r, "huber_corner"),
                                              vo.scalar(r, "huber_edge"), int(vo.scalar(r, "n_iter")))
        rg = capi.BaResult.from_buffer_copy(r["result"])
        self.bump("ba", "n"); self.bump("ba", "edges", len(edges)); self.bump("ba", "unknowns", len(io) + 6 * int((fixed == 0).sum()))
        assert (rg.iterations, rg.n_solves, rg.n_accepted, rg.stop) == (ro.iterations, ro.n_solves, ro.n_accepted, ro.stop)
        assert np.abs(np.frombuffer(r["idist_out"], np.float64) - io).max() <= 1e-9
        for a, b in zip(pg[:n], po):
            assert np.abs(np.array(a.q[:]) - np.array(b.q[:])).max() <= 1e-9 and np.abs(np.array(a.t[:]) - np.array(b.t[:])).max() <= 1e-9
        assert np.allclose(np.frombuffer(r["edge_chi2"], np.float64), co, rtol=1e-7, atol=1e-16)
        assert rg.final_chi2 == pytest.approx(ro.final_chi2, rel=1e-8)

    def detect_candidates(self, r):
        # FAST / Canny / arg-max are bit-exact: the recorded lists must equal the oracle's
        f = self.frame(vo.scalar(r, "frame_id"))
        if vo.scalar(r, "init"):
            return
        W, H = f["img"].shape[1], f["img"].shape[0]
        for L in range(int(vo.scalar(r, "n_levels"))):
            co, ed, _ = self.orc.detect_candidates_level(np.ascontiguousarray(f["pyr"][L]), *f["sobel"][L], L, W, H, int(vo.scalar(r, "min_thresh")))
            gc = np.frombuffer(r["corners%d" % L], capi.CORNER_DTYPE); ge = np.frombuffer(r["edgelets%d" % L], capi.EDGELET_DTYPE)
            assert len(gc) == len(co) and np.array_equal(gc["x"], co["x"]) and np.array_equal(gc["y"], co["y"]) and np.array_equal(gc["score"], co["score"])
            assert len(ge) == len(ed) and np.array_equal(ge["x"], ed["x"]) and np.array_equal(ge["y"], ed["y"])
            self.bump("detect", "corners", len(gc)); self.bump("detect", "edgelets", len(ge))

    def select_octree(self, r):
        # FeatureExtractor::computeKeyPointsOctTree: the product's index-range implementation against the list-of-lists
        # restatement (oracle/octree_py.py) on the keys the driver passed — bit for bit, order included
        from oracle.octree_py import octree_py
        keys = np.frombuffer(r["keys"], capi.KEYPOINT_DTYPE)
        want = octree_py(list(keys), int(vo.scalar(r, "width")), int(vo.scalar(r, "height")), int(vo.scalar(r, "n_features")))
        assert r["out"] == np.array(want, capi.KEYPOINT_DTYPE).tobytes()
        self.bump("detect", "octree"); self.bump("detect", "octree_selected", len(want))
