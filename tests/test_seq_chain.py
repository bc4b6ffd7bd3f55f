"""The resident per-frame chain (hso_gpu_seq_chain: tracker table, visiting order, point list, projection + matching, grid selection,
frame features, pose optimiser, the candidates' failure / success bookkeeping with its kind changes, needNewKf's flow sums, the
covisibility ranking, the scene depth) on the device against its sequential restatement (tests/fakegpu: the same entry point
implemented on oracle/, the reference's statements in the reference's order).

Both run the SAME host engine (hso_amd/host) on the same images; what differs is every device function.  The per-call numerics are
the business of the replay tests (tests/test_chain_gpu.py: every recorded tracker / matcher / pose / seed / BA call against the
restatement on identical inputs); here the two evolving states are compared frame by frame — which needs the chain's bookkeeping
(which points a frame lists and in which order, which candidates it examines, whose counters cross which threshold, when a
keyframe is due, which keyframes are connected) to agree step after step, since any deviation there changes the next frames'
lists, trial counts and keyframe timing."""
import os
import subprocess

import numpy as np
import pytest

from hso_amd import capi, synth, vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = dict(synth.EUROC, width=384, height=256, fx=240.0, fy=240.0, cx=191.5, cy=127.5)
MOTION = dict(step=(0.05, 0.015, 0.02), rot_deg_per_frame=(0.1, -0.3, 0.08))


def _rot_err(qa, qb):
    return 2 * np.arccos(min(1.0, abs(float(np.dot(qa, qb)))))


def _run(lib, S, n, max_fts):
    odo = vo.VisualOdometry(synth.camera(S["spec"]), max_fts, lib=lib)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    out = [odo.add_image(S["images"][k], float(k)) for k in range(1, n)]
    kfs = odo.keyframes()
    odo.close()
    return out, kfs


@pytest.mark.parametrize("max_fts", [120, 400])
def test_device_chain_follows_the_restatement_frame_by_frame(orc, max_fts):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fakegpu")])
    fake = vo.load_from(os.path.join(HERE, "fakegpu", "libhso_host_fake.so"))
    n = 40
    S = synth.sequence(n, spec=SMALL, workers=4, **MOTION)
    dev, kf_dev = _run(None, S, n, max_fts)
    cpu, kf_cpu = _run(fake, S, n, max_fts)
    # keyframes are taken at the same frames, and every frame reports the same stage / result
    assert [st.is_keyframe for st in dev] == [st.is_keyframe for st in cpu]
    assert [(st.stage, st.result) for st in dev] == [(st.stage, st.result) for st in cpu]
    assert len(kf_dev) == len(kf_cpu) >= 4
    worst = dict(rot=0.0, trans=0.0, matches=0, trials=0, cands=0, seeds=0)
    for k, (a, b) in enumerate(zip(dev, cpu)):
        qa, ta = a.T_f_w.to_arrays(); qb, tb = b.T_f_w.to_arrays()
        worst["rot"] = max(worst["rot"], _rot_err(qa, qb)); worst["trans"] = max(worst["trans"], float(np.linalg.norm(ta - tb)))
        worst["matches"] = max(worst["matches"], abs(a.n_matches - b.n_matches)); worst["trials"] = max(worst["trials"], abs(a.n_trials - b.n_trials))
        worst["cands"] = max(worst["cands"], abs(a.n_candidates - b.n_candidates)); worst["seeds"] = max(worst["seeds"], abs(a.n_seeds - b.n_seeds))
        assert a.used_inverse == b.used_inverse and a.n_features == b.n_features or abs(a.n_features - b.n_features) <= 3, k
    print("device engine vs restatement engine over %d frames at %d features:" % (n - 1, max_fts), worst)
    # the two states stay together: poses within the tracker's / optimiser's tolerances accumulated over the run, the list-driven
    # counters within a handful (a matcher decision inside its stated margin moves one candidate, never a keyframe or a whole list)
    assert worst["rot"] < 2e-4 and worst["trans"] < 5e-4, worst
    assert worst["matches"] <= 4 and worst["trials"] <= max(12, max_fts // 20) and worst["cands"] <= 12 and worst["seeds"] <= 12, worst


# ------------------------------------------------------------------------------------------------------------------------------
# Per-call parity of the chain's index work: ONE sequence-map state handed to hso_gpu_seq_chain in libhso_gpu.so and to its
# sequential restatement (tests/fakegpu), compared bit for bit where the quantity is an index, a count, a flag or an order
# statistic.  What each stage restates: Reprojector::reprojectMap's keyframe walk and point list (src/reprojector.cpp:98-202),
# CoarseTracker::makeDepthRef (src/CoarseTracker.cpp:210-240), reprojectCell's bookkeeping (src/reprojector.cpp:352-429), needNewKf
# (src/frame_handler_mono.cpp:428-507), createCovisibilityGraph (:559-647), getSceneDepth (src/frame.cpp:323-366).
import chain_state as cs


def _capture(spec, max_fts, n_frames, at, tmp_path, **motion):
    """run the product engine on the device, record the chain call of the frames in `at` together with the sequence map it ran on"""
    S = synth.sequence(n_frames, spec=spec, workers=4, **motion)
    odo = vo.VisualOdometry(synth.camera(spec), max_fts)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    recs = {}
    for k in range(1, n_frames):
        path = str(tmp_path / ("state%d.bin" % k))
        if k in at:
            odo.trace(path, state=True)
        st = odo.add_image(S["images"][k], float(k))
        assert st.stage == 3 and st.result != 2, k
        if k in at:
            odo.trace(None)
            r = dict(vo.read_trace(path))
            recs[k] = (cs.state_from_record(r["seq_chain_state"]), cs.result_from_record(r["seq_chain_result"]))
    n_kf = len(odo.keyframes())
    odo.close()
    return S, recs, n_kf


def _both(st, images, fake_lib, flags=None):
    out = []
    for lib in (cs.ChainLib(capi.load()), cs.ChainLib(fake_lib)):
        ls = cs.LoadedState(lib, st, images)
        out.append(ls.run(flags=flags))
        ls.close()
    return out


def _assert_index_work_equal(dev, cpu, st, what, pose_dependent=True):
    """dev / cpu: what LoadedState.run returned for the device library and for the restatement"""
    a, b = dev["result"], cpu["result"]
    # ---- makeDepthRef: the tracker's reference table (ids = positions in the reference frame's feature list)
    assert len(dev["ref_table"]) == len(cpu["ref_table"]), what
    if len(dev["ref_table"]):
        assert np.array_equal(dev["ref_table"]["px"], cpu["ref_table"]["px"]) and np.array_equal(dev["ref_table"]["f"], cpu["ref_table"]["f"]), what
        da, db = dev["ref_table"]["dist"], cpu["ref_table"]["dist"]
        assert np.array_equal(da < 0, db < 0), what                                  # which features have a usable point
        assert np.all(np.abs(da - db) <= 1e-14 * np.maximum(1.0, np.abs(db))), (what, float(np.abs(da - db).max()))
    # ---- the keyframe walk and the point list
    assert a["n_visit"] == b["n_visit"] and np.array_equal(a["visit"], b["visit"]), (what, a["visit"], b["visit"])
    assert (a["n_listed"], a["n_kf_points"], a["n_candidates"]) == (b["n_listed"], b["n_kf_points"], b["n_candidates"]), what
    assert np.array_equal(dev["list_ids"], cpu["list_ids"]) and np.array_equal(dev["list_quality"], cpu["list_quality"]), what
    if not pose_dependent:
        return
    # ---- the selection's counts, the frame's features (which candidates became features, in fts_ order)
    assert np.array_equal(a["counts"], b["counts"]), (what, a["counts"], b["counts"])
    fa, fb = dev["features"], cpu["features"]
    assert a["n_feats"] == b["n_feats"] == len(fa) == len(fb), what
    for key in ("point", "level", "type"):
        assert np.array_equal(fa[key], fb[key]), (what, key)
    # the matcher's tolerance (tests/test_align.py: 1e-3 px; an LK run that stops one iteration apart in the two lands up to a few 1e-3 away)
    dpx = np.abs(fa["px"] - fb["px"]).max(axis=1) if len(fa) else np.zeros(1)
    assert (dpx >= 1e-3).sum() <= max(2, 0.005 * len(dpx)) and dpx.max() < 5e-3 and np.abs(fa["f"] - fb["f"]).max() < 2e-5 and np.abs(fa["grad"] - fb["grad"]).max() < 1e-3, (what, float(dpx.max()))
    # ---- reprojectCell's bookkeeping: every kind change in the reference's order, every counter of every point row afterwards
    assert a["n_events"] == b["n_events"] and np.array_equal(dev["events"], cpu["events"]), (what, a["n_events"], b["n_events"])
    assert np.array_equal(a["events"], b["events"]), what
    wa, wb = dev["after"]["points"]["pad_"], cpu["after"]["points"]["pad_"]
    assert np.array_equal(wa, wb), (what, int((wa != wb).sum()))
    # ---- the pose optimiser's culling as the frame's feature table carries it, createCovisibilityGraph's votes and ranking
    assert (a["pose"]["num_obs"], a["pose"]["n_deleted"], a["pose"]["status"]) == (b["pose"]["num_obs"], b["pose"]["n_deleted"], b["pose"]["status"]), what
    for key in ("n_with_point", "n_covis", "covis_best", "flow_count", "make_kf", "seeds_observed"):
        assert a[key] == b[key], (what, key, a[key], b[key])
    assert np.array_equal(a["covis"], b["covis"]) and np.array_equal(a["covis_votes"], b["covis_votes"]), what
    # ---- needNewKf's sums: serial fp32 in list order on both sides (hso_select.hip: k_chain_finish adds (float)((double)s + term) on
    # one lane like the restatement).  Their input is the OPTIMISED pose, which inherits the matcher's 1e-3 px tolerance through
    # the features (~1e-6 m at 2000 features): the sums agree as far as that pose does, the decision they feed (make_kf) exactly
    for key in ("flow_full", "flow_shift"):
        assert abs(float(a[key]) - float(b[key])) <= 2e-4 * max(1.0, abs(float(b[key]))), (what, key, a[key], b[key])
    # ---- getSceneDepth / getSceneDistance: order statistics (the element at n / 2, the minimum) of the depths of the same points under
    # the two optimised poses: the same element, its value moved by what the poses differ (device LM vs the restatement's: 1e-8)
    dq, dt = _rot_err(a["pose"]["T_f_w"]["q"], b["pose"]["T_f_w"]["q"]), float(np.linalg.norm(a["pose"]["T_f_w"]["t"] - b["pose"]["T_f_w"]["t"]))
    assert dq < 1e-5 and dt < 1e-5, (what, dq, dt)
    if a["depth_min"] >= 0:
        for key in ("depth_median", "dist_median", "depth_min"):
            assert abs(float(a[key]) - float(b[key])) <= 2 * (dt + dq * abs(float(b[key]))) + 1e-12, (what, key, a[key], b[key], dq, dt)
    # the map's new frame table after the call: the same one in both
    assert dev["after"]["ff_frame"][dev["after"]["ff_newest"]] == cpu["after"]["ff_frame"][cpu["after"]["ff_newest"]] == int(st["job"]["cur_frame_id"][0])


@pytest.fixture(scope="module")
def fake_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fakegpu")])
    return vo.load_from(os.path.join(HERE, "fakegpu", "libhso_host_fake.so"))


CHAIN_CASES = [("euroc_200", 200, 60, (40, 59)), ("euroc_2000", 2000, 40, (24, 39))]


@pytest.mark.parametrize("name,max_fts,n_frames,at", CHAIN_CASES, ids=[c[0] for c in CHAIN_CASES])
def test_chain_call_on_one_map_state_device_vs_restatement(orc, fake_lib, tmp_path, name, max_fts, n_frames, at):
    S, recs, n_kf = _capture(synth.EUROC, max_fts, n_frames, at, tmp_path)
    assert n_kf >= 3, n_kf
    images = {k: S["images"][k] for k in range(n_frames)}          # one sequence: frame id = image index (Seq::new_frame)
    assert len(recs[at[-1]][0]["kfs"]) >= 3                          # the later state: a snapshot after at least three keyframes
    for k in at:
        st, want = recs[k]
        assert len(st["kfs"]) >= 2 and len(st["cands"]) > 0 and int(st["job"]["n_ref_feats"][0]) >= min(max_fts, 150)
        # (0) the record is complete: a fresh device context rebuilt from it repeats the engine's own call bit for bit
        lib = cs.ChainLib(capi.load())
        ls = cs.LoadedState(lib, st, images)
        again = ls.run()
        ls.close()
        assert not cs.fields_differ(again["result"], want["result"], skip=("coop_workgroups", "coop_same_xcd")), (k, cs.fields_differ(again["result"], want["result"]))
        assert np.array_equal(again["events"], want["events"]) and again["features"].tobytes() == want["features"].tobytes()
        # (1) the call as the engine made it (tracker included): the reference table exactly; the walk and the list are functions of the
        # tracked pose only through visibility tests and distances, which a 1e-7 pose difference does not move on these scenes
        dev, cpu = _both(st, images, fake_lib)
        _assert_index_work_equal(dev, cpu, st, "%s frame %d, tracker on" % (name, k), pose_dependent=False)
        qa, ta = dev["result"]["T_tracked"]["q"], dev["result"]["T_tracked"]["t"]; qb, tb = cpu["result"]["T_tracked"]["q"], cpu["result"]["T_tracked"]["t"]
        assert _rot_err(qa, qb) < 5e-5 and np.linalg.norm(ta - tb) < 2e-4                      # the tracker's tolerance
        # (2) the same state with the tracker switched off and the engine's tracked pose as the prior: both sides project, match,
        # select, optimise and book-keep from bit-identical inputs — every index, count, flag and order statistic must agree
        st2 = dict(st); st2["job"] = st["job"].copy()
        st2["job"]["T_cur_w"] = want["result"]["T_tracked"]
        for flags in (cs.SEQ_NO_TRACK | (int(st["job"]["flags"][0]) & cs.SEQ_SEED_BRANCH), cs.SEQ_NO_TRACK | cs.SEQ_DEPTH_STATS):
            dev, cpu = _both(st2, images, fake_lib, flags=flags)
            _assert_index_work_equal(dev, cpu, st2, "%s frame %d, flags %d" % (name, k, flags))
            assert dev["result"]["n_listed"] > 3 * min(max_fts, 150) and dev["result"]["counts"][1] >= min(max_fts, 100)   # (without the tracker the frame has no exposure estimate: fewer matches than the engine's own call)
            if flags & cs.SEQ_DEPTH_STATS:
                assert dev["result"]["depth_min"] > 0 and dev["result"]["make_kf"] == 1
        print("%s frame %d: %d keyframes in the map, %d listed (%d keyframe points, %d candidates), %d trials, %d features, %d events" %
              (name, k, len(st["kfs"]), dev["result"]["n_listed"], dev["result"]["n_kf_points"], dev["result"]["n_candidates"], dev["result"]["counts"][0],
               dev["result"]["n_feats"], dev["result"]["n_events"]))


def _hand_made(st, rng):
    """a recorded state pushed into the corners the recorded runs do not reach: more visitable keyframes than HSO_SEQ_MAX_VISIT,
    deleted points still linked from feature lists, temporary points, counters one step from every threshold of
    src/reprojector.cpp:366-425 (more than HSO_SEQ_EVENTS kind changes in one frame)"""
    S = dict(st)
    for key in ("job", "cfg", "kfs", "points", "key_points", "kf_nfts", "cands"):
        S[key] = st[key].copy()
    S["kf_fts"] = [l.copy() for l in st["kf_fts"]]
    nk0 = len(S["kfs"])
    # 30 more keyframe rows: copies of the real ones a little to the side (so the distance order is a strict one), each seeing the
    # frame through its source's key points and listing every seventh feature of its source (points listed once: the stamps)
    extra, keys, lists = [], [], []
    for i in range(30):
        src = i % nk0
        row = S["kfs"][src].copy()
        row["T_f_w"]["t"] += rng.normal(0, 0.02, 3)
        row["keyframe_id"] = int(S["kfs"]["keyframe_id"].max()) + 1 + i
        extra.append(row); keys.append(S["key_points"][5 * src:5 * src + 5]); lists.append(S["kf_fts"][src][(i % 7)::7].copy())
    S["kfs"] = np.concatenate([S["kfs"], np.array(extra, cs.KF)])
    S["key_points"] = np.concatenate([S["key_points"]] + keys).astype(np.int32)
    S["kf_fts"] += lists
    S["kf_nfts"] = np.array([len(l) for l in S["kf_fts"]], np.int32)
    S["cfg"]["max_kfs"] = 40
    # state words: keep every key, set the counters to the brink
    w = S["points"]["pad_"].copy()
    key = cs.pt_key(w).astype(np.int64)
    kind = key >> 4
    r = rng.random(len(w))
    for i in range(len(w)):
        k = int(kind[i])
        if k == 3:      # TYPE_UNKNOWN: one failure from deletion / one success from TYPE_GOOD
            w[i] = cs.pt_word(int(key[i]), n_fail=15 if r[i] < 0.5 else 3, n_ok=10 if r[i] >= 0.5 else 2)
        elif k == 2:    # a candidate: one failure (or one missed projection: +3) from deletion
            w[i] = cs.pt_word(int(key[i]), n_fail=30 if r[i] < 0.5 else 28)
        elif k == 4 and r[i] < 0.05:
            w[i] = cs.pt_word(int(key[i]) & 0x0f)           # TYPE_DELETED, still linked from its keyframes' feature lists
    # temporary points: forty candidates change kind and move from the candidate list to the job's list
    cand = S["cands"]
    take = cand[:: max(1, len(cand) // 40)][:40]
    for p in take:
        w[p] = cs.pt_word((1 << 4) | (int(key[p]) & 0x0f), n_fail=30 if r[p] < 0.5 else 0)
    S["cands"] = np.array([p for p in cand if p not in set(take.tolist())], np.int32)
    S["temps"] = take.astype(np.int32)
    S["job"]["n_temps"] = len(take); S["job"]["temps_begin"] = 0
    S["points"]["pad_"] = w
    S["job"]["covis"] = [nk0 + 3, 0, nk0 + 11, -1, -1]
    return S


def test_chain_call_on_a_hand_made_state(orc, fake_lib, tmp_path):
    S, recs, n_kf = _capture(synth.EUROC, 2000, 34, (33,), tmp_path)
    st, want = recs[33]
    images = {k: S["images"][k] for k in range(34)}
    hm = _hand_made(st, np.random.default_rng(5))
    hm["job"]["T_cur_w"] = want["result"]["T_tracked"]
    dev, cpu = _both(hm, images, fake_lib, flags=cs.SEQ_NO_TRACK | cs.SEQ_DEPTH_STATS)
    _assert_index_work_equal(dev, cpu, hm, "hand-made state")
    r = dev["result"]
    codes = (dev["events"].astype(np.int64) & 0xffffffff) >> 28
    print("hand-made state: %d keyframe rows, %d visited, %d listed, %d events" % (len(hm["kfs"]), r["n_visit"], r["n_listed"], r["n_events"]),
          {c: int((codes == c).sum()) for c in (1, 2, 3, 4)})
    assert r["n_visit"] == cs.MAX_VISIT and r["n_events"] > cs.N_EVENTS
    assert all((codes == c).any() for c in (cs.EV_ERASE_POINT, cs.EV_ERASE_CANDIDATE, cs.EV_GOOD))
    # a tracked call on the same state (the reference table over a map with deleted points)
    dev, cpu = _both(hm, images, fake_lib)
    _assert_index_work_equal(dev, cpu, hm, "hand-made state, tracker on", pose_dependent=False)
