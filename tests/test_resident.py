"""Resident tables behind handles (seed tables, maps): same kernels as the value-passing calls, state kept in HBM between
calls — results must equal the value-passing entry points bit for bit, and the state must evolve like the host-side copy."""
import ctypes as C

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu
PX_ERROR_ANGLE = 2 * np.arctan(1.0 / (2.0 * 480.6))


def test_seed_table_matches_value_passing_calls(gpu_ctx, cam, pair2000):
    d = pair2000
    gpu_ctx.frame_upload(9301, d["ref"]); gpu_ctx.frame_upload(9302, d["cur"]); gpu_ctx.frame_upload(9303, d["cur"])
    try:
        seeds, T_cur, _ = synth.seeds_for_pair(d, 700, 9301, seed=5)
        t = gpu_ctx.seed_table_create()
        first = gpu_ctx.seed_table_append(t, seeds[:400])
        assert first == 0 and gpu_ctx.seed_table_append(t, seeds[400:]) == 400 and gpu_ctx.seed_table_size(t) == (700, 700)
        host = [capi.Seed.from_buffer_copy(bytes(s)) for s in seeds]          # the value-passing side keeps its own state
        T2 = capi.SE3.from_arrays(T_cur.q[:], np.array(T_cur.t[:]) * 1.3)
        alive = np.ones(700, bool)
        for rnd, (fid, T, expo) in enumerate([(9302, T_cur, 1.05), (9303, T2, 1.02), (9302, T_cur, 1.05)]):
            brief, full = gpu_ctx.seed_table_observe(cam, t, [(fid, T, expo)], PX_ERROR_ANGLE, want_full=(rnd == 1))
            ref = gpu_ctx.seed_observe(cam, fid, T, expo, PX_ERROR_ANGLE, host)
            for i, (s, o) in enumerate(zip(host, ref)):
                if not alive[i]:
                    assert brief[i]["result"] == 0 and brief[i]["mu"] == 0
                    continue
                b = brief[i]
                assert (b["mu"], b["sigma2"], b["b"], b["result"], b["is_update"], b["is_valid"], b["search_level"]) == \
                       (o.mu, o.sigma2, o.b, o.result, o.is_update, o.is_valid, o.search_level), (rnd, i)
                if full is not None:
                    assert bytes(full[i]) == bytes(o)
                s.mu, s.sigma2, s.b = o.mu, o.sigma2, o.b                      # what DepthFilter::updateSeed leaves in the seed
            if rnd == 0:                                                       # erase every 7th seed: slots keep their index
                gone = np.arange(0, 700, 7)
                gpu_ctx.seed_table_erase(t, gone); alive[gone] = False
                assert gpu_ctx.seed_table_size(t) == (700, 600)
        # the table caches the host frame's base pointer per live seed: that frame cannot be released under it
        with pytest.raises(capi.HsoGpuError, match="live seeds"):
            gpu_ctx.frame_release(9301)
        back = gpu_ctx.seed_table_read(t, 0, 700)
        for i in np.where(alive)[0][:50]:
            assert (back[i].mu, back[i].sigma2, back[i].b) == (host[i].mu, host[i].sigma2, host[i].b)
        # compaction: the 600 live records move to slots 0..599 in order, the table shrinks, and the next observation equals the
        # value-passing call over the live seeds alone
        n_new, remap = gpu_ctx.seed_table_compact(t)
        live = np.where(alive)[0]
        assert n_new == 600 and gpu_ctx.seed_table_size(t) == (600, 600)
        assert np.array_equal(remap[live], np.arange(600)) and (remap[~alive] == -1).all()
        moved = gpu_ctx.seed_table_read(t, 0, 600)
        assert all(bytes(moved[k]) == bytes(back[i]) for k, i in enumerate(live))
        brief, _ = gpu_ctx.seed_table_observe(cam, t, [(9303, T2, 1.02)], PX_ERROR_ANGLE)
        ref = gpu_ctx.seed_observe(cam, 9303, T2, 1.02, PX_ERROR_ANGLE, [host[i] for i in live])
        assert len(brief) == 600
        assert all((b["mu"], b["sigma2"], b["b"], b["result"]) == (o.mu, o.sigma2, o.b, o.result) for b, o in zip(brief, ref))
        assert gpu_ctx.seed_table_compact(t)[0] == 600                         # nothing erased: a no-op
        assert gpu_ctx.seed_table_append(t, seeds[:5]) == 600                  # appends continue behind the compacted records
        gpu_ctx.seed_table_destroy(t)
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.seed_table_size(t)
    finally:
        for i in (9301, 9302, 9303):
            gpu_ctx.frame_release(i)


def test_seed_table_groups_serve_many_sequences(gpu_ctx, cam, pair2000):
    d = pair2000
    gpu_ctx.frame_upload(9311, d["ref"]); gpu_ctx.frame_upload(9312, d["cur"])
    try:
        seeds, T_cur, _ = synth.seeds_for_pair(d, 120, 9311, seed=9)
        t = gpu_ctx.seed_table_create()
        gpu_ctx.seed_table_append(t, seeds * 3, group=np.repeat(np.arange(3), 120))
        T_b = capi.SE3.from_arrays(T_cur.q[:], np.array(T_cur.t[:]) * 0.5)
        brief, _ = gpu_ctx.seed_table_observe(cam, t, [(9312, T_cur, 1.05), (9312, T_b, 1.05), (9312, T_cur, 1.05)], PX_ERROR_ANGLE)
        assert np.array_equal(brief[:120], brief[240:]) and not np.array_equal(brief[:120], brief[120:240])
        solo = gpu_ctx.seed_observe(cam, 9312, T_b, 1.05, PX_ERROR_ANGLE, seeds)
        assert [o.result for o in solo] == list(brief[120:240]["result"])
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.seed_table_observe(cam, t, [(9312, T_cur, 1.05)], PX_ERROR_ANGLE)       # groups 1, 2 have no frame
        gpu_ctx.seed_table_destroy(t)
    finally:
        gpu_ctx.frame_release(9311); gpu_ctx.frame_release(9312)


def test_resident_maps_match_value_passing_call(gpu_ctx):
    spec = synth.ICL_NUIM
    cam = synth.camera(spec)
    P = synth.map_problem(n_points=700, spec=spec, first_frame_id=9400)
    Q = synth.map_problem(n_points=500, spec=spec, first_frame_id=9400, seed=72)        # a second map over the same frames
    ids = [int(k["frame_id"]) for k in P["kfs"]]
    for i, f in zip(ids, P["frames"]):
        gpu_ctx.frame_upload(i, f)
    gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"])
    try:
        gpu_ctx.map_reserve(3, 16, 800, 4000)
        gpu_ctx.map_store(0, P["kfs"], P["points"], P["obs"])
        gpu_ctx.map_store(2, Q["kfs"], Q["points"], Q["obs"])
        calls = np.zeros(3, capi.MAP_CALL_DTYPE)
        q, t = P["T_cur_w"].to_arrays()
        for c, m in enumerate((0, 2, 0)):
            calls[c]["map"], calls[c]["cur_keyframe_id"], calls[c]["cur_frame_id"] = m, P["cur_keyframe_id"], P["cur_frame_id"]
            calls[c]["q"], calls[c]["t"], calls[c]["cur_exposure_time"] = q, t, P["cur_exposure"]
        calls[2]["t"] = t * 1.2
        out = gpu_ctx.reproject_match_maps(cam, calls, P["cell_size"], P["grid_n_cols"], 2000)
        assert len(out) == 700 + 500 + 700
        T3 = capi.SE3.from_arrays(q, t * 1.2)
        for lo, M, T in ((0, P, P["T_cur_w"]), (700, Q, P["T_cur_w"]), (1200, P, T3)):
            proj, match = gpu_ctx.reproject_match(cam, P["cur_frame_id"], T, P["cur_exposure"], P["cur_keyframe_id"], M["kfs"], M["points"],
                                                  M["obs"], P["cell_size"], P["grid_n_cols"])
            b = out[lo:lo + len(proj)]
            assert np.array_equal(b["cell"], np.where(proj["projected"] == 1, proj["cell"], -1)) and np.array_equal(b["ref_obs"], proj["ref_obs"])
            assert np.array_equal(b["px"], proj["px"])
            assert [m.success for m in match[:len(proj)]] == list(b["success"]) and [m.search_level for m in match[:len(proj)]] == list(b["search_level"])
            ok = b["success"] == 1
            assert np.array_equal(b["px_cur"][ok], np.array([m.px_cur[:] for m in match[:len(proj)]])[ok]) and ok.sum() > 100
        # a map larger than its region, or a call naming a map that does not exist
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.map_store(1, P["kfs"], np.concatenate([P["points"]] * 2), P["obs"])
        calls[0]["map"] = 7
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.reproject_match_maps(cam, calls, P["cell_size"], P["grid_n_cols"], 2000)
    finally:
        for i in ids + [P["cur_frame_id"]]:
            gpu_ctx.frame_release(i)


def test_pose_chained_behind_the_selection_equals_the_value_passing_optimiser(gpu_ctx, orc):
    """hso_gpu_reproject_select_pose_maps: the selection's result feeds optimizeLevenbergMarquardt3rd without leaving the device.
    Check: (1) the selection part returns exactly what hso_gpu_reproject_select_maps returns; (2) the pose results equal
    hso_gpu_pose_optimize_batch fed with the feature tables a host adapter builds from those records the documented way
    (f = cam2world(px_cur), level / type / grad of the record, the point's host bearing, inverse depth, host keyframe, temporary
    flag) — same kernel; the bearings come from the host's cam2world here and from the device's there (last-bit differences),
    and the optimiser stops on a 1e-10 step, so the poses agree to 1e-9 with identical counts and masks; (3) the CPU restatement of the optimiser on the same tables agrees (pose
    1e-9, outlier mask, iteration counts by the margin rule of tests/test_pose.py)."""
    for spec in (synth.ICL_NUIM, synth.EUROC):
        cam = synth.camera(spec)
        P = synth.map_problem(n_points=900, spec=spec, first_frame_id=9600)
        rng = np.random.default_rng(5)
        t = rng.integers(1, 5, size=len(P["points"]))
        P["points"]["pad_"] = (t << 4) | rng.integers(0, 3, size=len(t))          # incl. TYPE_TEMPORARY (1) points
        ids = [int(k["frame_id"]) for k in P["kfs"]]
        for i, f in zip(ids, P["frames"]):
            gpu_ctx.frame_upload(i, f)
        gpu_ctx.frame_upload(P["cur_frame_id"], P["cur"])
        try:
            gpu_ctx.map_reserve(2, 16, 1000, 5000)
            gpu_ctx.map_store(0, P["kfs"], P["points"], P["obs"]); gpu_ctx.map_store(1, P["kfs"], P["points"], P["obs"])
            calls = np.zeros(2, capi.MAP_CALL_DTYPE)
            q, tt = P["T_cur_w"].to_arrays()
            for c in range(2):
                calls[c]["map"], calls[c]["cur_keyframe_id"], calls[c]["cur_frame_id"] = c, P["cur_keyframe_id"], P["cur_frame_id"]
                calls[c]["q"], calls[c]["t"], calls[c]["cur_exposure_time"] = q, tt, P["cur_exposure"]
            calls[1]["t"] = tt + np.array([0.004, -0.003, 0.002])              # a perturbed start for the second sequence
            n_cells = P["grid_n_cols"] * int(np.ceil(spec["height"] / P["cell_size"]))
            order = np.random.default_rng(3).permutation(n_cells).astype(np.int32)
            budget = 300
            ref_out, ref_begin, ref_counts = gpu_ctx.reproject_select_maps(cam, calls, P["cell_size"], P["grid_n_cols"], order, budget, 2000)
            out, begin, counts, res, nf, mask = gpu_ctx.reproject_select_pose_maps(cam, calls, P["cell_size"], P["grid_n_cols"], order, budget, 2000)
            assert out.tobytes() == ref_out.tobytes() and np.array_equal(begin, ref_begin) and np.array_equal(counts, ref_counts)
            poses = [capi.SE3.from_arrays(k["q"], k["t"]) for k in P["kfs"]]
            for c in range(2):
                rec = out[begin[c]:begin[c + 1]]
                taken = rec[rec["success"] == 1]
                assert nf[c] == len(taken) == counts[c, 1] and 100 < nf[c] <= budget
                pf = np.zeros(len(taken), capi.POSE_FEAT_DTYPE)
                for k, r in enumerate(taken):
                    p = P["points"][r["pad_"]]
                    pf[k]["has_point"], pf[k]["type"], pf[k]["level"] = 1, r["ref_type"], r["search_level"]
                    pf[k]["temporary"] = 1 if (p["pad_"] >> 4) == 1 else 0
                    pf[k]["host_pose"] = p["host_kf"]
                    pf[k]["f"] = orc.cam2world(cam, r["px_cur"][0], r["px_cur"][1])
                    pf[k]["grad"] = [float(r["grad"][0]), float(r["grad"][1])]
                    pf[k]["host_f"], pf[k]["idist"] = p["host_f"], p["idist"]
                T0 = capi.SE3.from_arrays(calls[c]["q"], calls[c]["t"])
                job = capi.make_pose_job(pf, poses, T0)
                (rv,), (mv,) = gpu_ctx.pose_optimize_batch(cam, [job])
                g = res[c]
                assert (g.status, g.iters, g.n_trials_total, g.num_obs, g.n_deleted) == (rv.status, rv.iters, rv.n_trials_total, rv.num_obs, rv.n_deleted)
                # radtan: cam2world runs OpenCV's five fp32 undistortion iterations (src/camera.cpp:78-85), which host and device round
                # differently at the 1e-7 level of the bearing (tests/test_reproject.py): the host-built table differs from the
                # device-built one by that much, and the poses by ~1e-7 (rotation) / (translation, scene depth 2..6 m)
                tol = 1e-9 if spec is synth.ICL_NUIM else 5e-7
                assert np.allclose(g.T_f_w.q[:], rv.T_f_w.q[:], atol=tol, rtol=0) and np.allclose(g.T_f_w.t[:], rv.T_f_w.t[:], atol=tol, rtol=0)
                assert np.array_equal(mask[c, :nf[c]], mv) and g.estimated_scale == pytest.approx(rv.estimated_scale, rel=100 * tol)
                # the restatement on the same table
                orc.margins_reset()
                ro, mo = orc.pose_optimize(cam, job)
                mg = orc.margins()
                if (rv.iters, rv.n_trials_total) != (ro.iters, ro.n_trials_total):
                    assert mg.pose_rho < 1e-12
                assert np.allclose(rv.T_f_w.q[:], ro.T_f_w.q[:], atol=1e-9, rtol=0) and np.allclose(rv.T_f_w.t[:], ro.T_f_w.t[:], atol=1e-9, rtol=0)
                assert np.array_equal(mv, mo)
            assert int((P["points"]["pad_"][out[out["success"] == 1]["pad_"]] >> 4 == 1).sum()) > 5      # temporary points took part
        finally:
            for i in ids + [P["cur_frame_id"]]:
                gpu_ctx.frame_release(i)
