"""hso_gpu_seq_local_ba — ba::LocalBundleAdjustment on a sequence map (src/bundle_adjustment.cpp:577-892) — against the CPU restatement
(tests/fakegpu: the graph built by the reference's sequential walk, oracle/'s deltas and Levenberg loop), call for call.

A product run on the device records the map right before a keyframe's local BA (hso_vo_trace_state: "seq_ba_state") and what the
call returned ("seq_ba_result").  The state is rebuilt in a fresh context of libhso_gpu.so and in the restatement through the public
hso_gpu_seqmap_* calls, the same call is made in both, and

  * the WINDOW the device assembled — the points (ascending rows), the vertex of every keyframe (the core first, then the other
    keyframes in the order the reference's walk meets them), every edge record (indices, bearings, measurements, normals, levels),
    project2d of every observation, the observation row behind every edge — equals the restatement's BYTE FOR BYTE;
  * the same window through the value-passing call (hso_gpu_ba_local_multi, whose tables the host builds) gives the device call's
    poses, inverse depths and chi2 values BIT FOR BIT — the resident assembly, its pose-pair lists included, changes nothing;
  * deltas, poses, inverse depths, positions and the culling list agree with the restatement within the optimiser's tolerance
    (tests/test_ba.py), and the map the call leaves (point rows: idist_, pos_) is what it returned.
"""
import ctypes as C
import os

import numpy as np
import pytest

from hso_amd import capi, synth, vo

import chain_state as cs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_lib(orc):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fakegpu", "libhso_host_fake.so")
    if not os.path.exists(path):
        pytest.skip("tests/fakegpu/libhso_host_fake.so not built")
    return C.CDLL(path)


def _capture(spec, max_fts, n_frames, first, tmp_path):
    """the product engine on the device; every local BA from frame `first` on is recorded with the map it ran on"""
    S = synth.sequence(n_frames, spec=spec, workers=4)
    odo = vo.VisualOdometry(synth.camera(spec), max_fts)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    recs = []
    for k in range(1, n_frames):
        path = str(tmp_path / ("ba%d.bin" % k))
        if k >= first:
            odo.trace(path, state=True)
        st = odo.add_image(S["images"][k], float(k))
        assert st.stage == 3 and st.result != 2, k
        if k >= first:
            odo.trace(None)
            r = dict(vo.read_trace(path))
            if "seq_ba_state" in r:
                recs.append((k, cs.ba_state_from_record(r["seq_ba_state"]), cs.ba_result_from_record(r["seq_ba_result"]), r["ba_optimize"]))
            os.remove(path)
    odo.close()
    return recs


def _run_both(st, fake_lib, **kw):
    out = []
    for lib in (cs.ChainLib(capi.load()), cs.ChainLib(fake_lib)):
        ls = cs.LoadedState(lib, st, None)
        out.append(ls.run_ba(st["core"], st["fixed"], st["n_iter"], st["error_multiplier2"], st["chi2_corner"], st["chi2_edgelet"], **kw))
        ls.close()
    return out


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _assert_window_equal(dev, cpu, what):
    a, b = dev["result"], cpu["result"]
    assert (a["status"], a["n_poses"], a["n_points"], a["n_edges"]) == (b["status"], b["n_poses"], b["n_points"], b["n_edges"]), (what, a, b)
    assert np.array_equal(dev["point_ids"], cpu["point_ids"]), what
    assert np.all(np.diff(dev["point_ids"]) > 0), what
    Wa, Wb = dev["window"], cpu["window"]
    assert np.array_equal(Wa["vertex_rows"], Wb["vertex_rows"]) and np.array_equal(Wa["fixed"], Wb["fixed"]), (what, Wa["vertex_rows"], Wb["vertex_rows"])
    assert Wa["poses_in"].tobytes() == Wb["poses_in"].tobytes(), what
    if a["status"] != 0:
        return
    for key in ("edges", "obs_uv", "edge_obs", "idist_in"):
        if Wa[key].tobytes() != Wb[key].tobytes():
            if key == "edges":
                bad = [n for n in Wa[key].dtype.names if n != "_pad" and not np.array_equal(Wa[key][n], Wb[key][n])]
                assert not bad, (what, key, bad)
            else:
                assert False, (what, key, int((Wa[key] != Wb[key]).sum()))


def _assert_results_close(dev, cpu, st, what):
    a, b = dev["result"], cpu["result"]
    assert a["status"] == b["status"] == 0, what
    assert abs(float(a["huber_corner"]) - float(b["huber_corner"])) <= 1e-6 * max(1e-12, abs(float(b["huber_corner"]))), (what, a["huber_corner"], b["huber_corner"])
    assert abs(float(a["huber_edge"]) - float(b["huber_edge"])) <= 1e-6 * max(1e-12, abs(float(b["huber_edge"]))), (what, a["huber_edge"], b["huber_edge"])
    assert abs(a["lm"]["init_chi2"] - b["lm"]["init_chi2"]) <= 1e-9 * abs(b["lm"]["init_chi2"]), what
    # the optimiser's tolerance (tests/test_ba.py): same accepted steps, the state after them to ~1e-7 relative
    assert (a["lm"]["iterations"], a["lm"]["n_solves"], a["lm"]["n_accepted"], a["lm"]["stop"]) == (b["lm"]["iterations"], b["lm"]["n_solves"], b["lm"]["n_accepted"], b["lm"]["stop"]), (what, a["lm"], b["lm"])
    assert abs(a["lm"]["final_chi2"] - b["lm"]["final_chi2"]) <= 1e-6 * abs(b["lm"]["final_chi2"]), (what, a["lm"]["final_chi2"], b["lm"]["final_chi2"])
    n_core = len(st["core"])
    for c in range(n_core):
        pa, pb = a["core_pose"][c], b["core_pose"][c]
        assert np.abs(pa["q"] - pb["q"]).max() < 1e-7 and np.abs(pa["t"] - pb["t"]).max() < 1e-6, (what, c, pa, pb)
        if st["fixed"][c]:
            assert pa.tobytes() == st["kfs"][st["core"][c]]["T_f_w"].tobytes(), (what, c)
    sa, sb = dev["point_state"], cpu["point_state"]
    rel = np.abs(sa[:, 0] - sb[:, 0]) / np.maximum(np.abs(sb[:, 0]), 1e-9)
    assert rel.max() < 1e-5, (what, float(rel.max()))
    assert np.abs(sa[:, 1:] - sb[:, 1:]).max() < 1e-4 * max(1.0, np.abs(sb[:, 1:]).max()), what
    # the culling list: the same observations, but an edge within rounding of the threshold may fall either way
    ca, cb = set(dev["culled"].tolist()), set(cpu["culled"].tolist())
    assert len(ca ^ cb) <= max(2, 0.02 * max(len(ca), len(cb))), (what, len(ca), len(cb), len(ca ^ cb))
    if ca == cb:
        assert np.array_equal(dev["culled"], cpu["culled"]) and np.array_equal(a["n_culled"], b["n_culled"]), what


def _assert_map_written(out, st, what):
    """the rows the call left = what it returned; everything else untouched"""
    r, after = out["result"], out["after"]
    ids = out["point_ids"]
    if r["status"] == 0:
        assert np.array_equal(after["points"]["idist"][ids], out["point_state"][:, 0]) and np.array_equal(after["points"]["pos"][ids], out["point_state"][:, 1:]), what
        # pos_ = T_host^-1 (host_f / idist) with the pose the call returned for the host keyframe
        poses = {int(row): out["window"]["poses_out"][v] for v, row in enumerate(out["window"]["vertex_rows"])}
        for i in range(0, len(ids), max(1, len(ids) // 200)):
            P = after["points"][ids[i]]
            T = poses[int(P["host_kf"])]
            R = _rot(T["q"])
            want = R.T @ (P["host_f"] / P["idist"] - T["t"])
            assert np.abs(want - P["pos"]).max() < 1e-9 * max(1.0, np.abs(want).max()), (what, i)
        for c, row in enumerate(st["core"]):
            assert after["kfs"][row]["T_f_w"].tobytes() == r["core_pose"][c].tobytes(), (what, c)
    others = np.ones(len(st["points"]), bool); others[ids] = False
    for key in ("idist", "pos", "host_f", "host_kf", "obs_begin", "obs_count", "pad_"):
        keep = others if key in ("idist", "pos") and r["status"] == 0 else np.ones(len(st["points"]), bool)
        assert np.array_equal(after["points"][key][keep], st["points"][key][keep]), (what, key)
    assert after["obs"].tobytes() == st["obs"].tobytes() and np.array_equal(after["obs_point"], st["obs_point"]), what


def _replay_value_passing(dev, st, what):
    """the device-built window through hso_gpu_ba_local_multi (host-built adjacency tables): bit for bit the resident call's state"""
    W = dev["window"]
    ctx = capi.Context()
    poses = [capi.SE3.from_buffer_copy(W["poses_in"][v].tobytes()) for v in range(len(W["poses_in"]))]
    (res,), hub = ctx.ba_local_multi([(poses, W["fixed"], W["idist_in"], W["edges"], W["obs_uv"].reshape(-1, 2), st["n_iter"])], st["error_multiplier2"])
    ctx.close()
    poses_out, idist_out, chi2, R = res
    r = dev["result"]
    assert (float(hub[0, 0]), float(hub[0, 1])) == (float(r["huber_corner"]), float(r["huber_edge"])), what
    assert np.array_equal(np.asarray(idist_out), dev["point_state"][:, 0]), (what, "idist")
    assert np.array_equal(np.asarray(chi2), W["edge_chi2"]), (what, "chi2")
    for v in range(len(poses_out)):
        assert bytes(poses_out[v]) == W["poses_out"][v].tobytes(), (what, "pose", v)
    for key in ("init_chi2", "final_chi2", "robust_chi2", "iterations", "n_solves", "n_accepted", "stop"):
        assert getattr(R, key) == r["lm"][key], (what, key)
    assert R.lambda_ == r["lm"]["lambda_"], what


BA_CASES = [("euroc_200", 200, 70, 20), ("euroc_2000", 2000, 52, 8)]


@pytest.mark.parametrize("name,max_fts,n_frames,first", BA_CASES, ids=[c[0] for c in BA_CASES])
def test_local_ba_on_recorded_map_states_device_vs_restatement(orc, fake_lib, tmp_path, name, max_fts, n_frames, first):
    recs = _capture(synth.EUROC, max_fts, n_frames, first, tmp_path)
    assert len(recs) >= 2, len(recs)
    n_free_seen = 0
    for k, st, want, trace_rec in recs:
        what = "%s frame %d" % (name, k)
        dev, cpu = _run_both(st, fake_lib)
        assert dev["result"]["status"] == 0 and dev["result"]["n_edges"] >= 50 and dev["result"]["n_poses"] >= 2, (what, dev["result"])
        n_free_seen = max(n_free_seen, int((np.asarray(st["fixed"]) == 0).sum()))
        _assert_window_equal(dev, cpu, what)
        _assert_results_close(dev, cpu, st, what)
        _assert_map_written(dev, st, what)
        _assert_map_written(cpu, st, what + " (restatement)")
        _replay_value_passing(dev, st, what)
        # the transplanted call repeats the engine's own: same window, same state afterwards, bit for bit
        assert np.array_equal(dev["point_ids"], want["point_ids"]) and dev["point_state"].tobytes() == want["point_state"].tobytes(), what
        assert np.array_equal(dev["culled"], want["culled"]) and dev["result"]["lm"].tobytes() == want["result"]["lm"].tobytes(), what
        assert dev["window"]["edges"].tobytes() == trace_rec["edges"], what
    assert n_free_seen >= 2, n_free_seen


def test_local_ba_windows_that_have_nothing_to_optimise_and_bad_jobs(orc, fake_lib, tmp_path):
    recs = _capture(synth.EUROC, 200, 40, 15, tmp_path)
    k, st, want, _ = recs[0]
    # a window whose only keyframe has no features: no points, no edges -> status 1, nothing written
    newest = len(st["kfs"]) - 1
    lone = dict(st); lone["core"] = np.array([newest], np.int32); lone["fixed"] = np.array([0], np.uint8)
    lone["kf_fts"] = list(st["kf_fts"]); lone["kf_fts"][newest] = np.zeros(0, np.int32)
    dev, cpu = _run_both(lone, fake_lib)
    assert dev["result"]["status"] == 1 and dev["result"]["n_points"] == 0
    _assert_window_equal(dev, cpu, "lone core")
    for out in (dev, cpu):
        _assert_map_written(out, lone, "lone core")
    # every core keyframe fixed: the loop has no free unknown but the points
    allfixed = dict(st); allfixed["fixed"] = np.ones(len(st["core"]), np.uint8)
    dev, cpu = _run_both(allfixed, fake_lib)
    _assert_window_equal(dev, cpu, "all fixed")
    _assert_results_close(dev, cpu, allfixed, "all fixed")
    _replay_value_passing(dev, allfixed, "all fixed")
    # a culling list longer than what rides in the final read-back: thresholds of zero cull every edge with a positive chi2
    everything = dict(st); everything["chi2_corner"] = 0.0; everything["chi2_edgelet"] = 0.0
    dev, cpu = _run_both(everything, fake_lib)
    n = int(dev["result"]["n_culled"].sum())
    assert n > 2046 or dev["result"]["n_edges"] <= 2046, (n, dev["result"]["n_edges"])
    W = dev["window"]
    is_edgelet = W["edges"]["type"] == capi.FTR_EDGELET
    expect = np.concatenate([W["edge_obs"][(~is_edgelet) & (W["edge_chi2"] > 0)], W["edge_obs"][is_edgelet & (W["edge_chi2"] > 0)]])
    assert np.array_equal(dev["culled"], expect), (len(dev["culled"]), len(expect))
    # refusals
    for lib in (cs.ChainLib(capi.load()), cs.ChainLib(fake_lib)):
        ls = cs.LoadedState(lib, st, None)
        with pytest.raises(capi.HsoGpuError):
            ls.run_ba([len(st["kfs"])], [0], 5, 1.0, 1.0, 1.0, point_cap=100)                                # no such keyframe row
        with pytest.raises(capi.HsoGpuError):
            ls.run_ba([0, 0], [0, 0], 5, 1.0, 1.0, 1.0)                                         # a core keyframe twice
        with pytest.raises(capi.HsoGpuError):
            ls.run_ba(st["core"], st["fixed"], st["n_iter"], 1.0, 1.0, 1.0, point_cap=3)        # point_cap below the window
        ls.close()
