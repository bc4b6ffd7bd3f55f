"""The sequence engine's host logic (hso_amd/host/hso_engine*.cpp) without a GPU: the engine is built against tests/fakegpu —
the C-ABI entry points it calls, implemented on the CPU restatement — and driven through include/hso_vo.h like the product.
What this covers: the tables and their bookkeeping (keyframes, observation lists, candidates, seeds, the local BA window), the
device mirror (rows patched = rows the kernels would read: the trace reads them back), the trace format and tests/replay.py
(device == restatement here, so every replayed comparison must hold exactly), several sequences in one bank = the same sequences
alone.  The kernels themselves are the `-m gpu` tests' business."""
import os
import subprocess

import numpy as np
import pytest

from hso_amd import synth, vo

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = dict(synth.EUROC, width=384, height=256, fx=240.0, fy=240.0, cx=191.5, cy=127.5)
MOTION = dict(step=(0.05, 0.015, 0.02), rot_deg_per_frame=(0.1, -0.3, 0.08))


@pytest.fixture(scope="session")
def fake(orc):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fakegpu")])
    return vo.load_from(os.path.join(HERE, "fakegpu", "libhso_host_fake.so"))


@pytest.fixture(scope="session")
def small_seq():
    return synth.sequence(34, spec=SMALL, workers=4, **MOTION)


def _run(lib, S, n, max_fts, trace=None):
    odo = vo.VisualOdometry(synth.camera(S["spec"]), max_fts, lib=lib)
    if trace:
        odo.trace(trace)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    out = []
    for k in range(1, n):
        st = odo.add_image(S["images"][k], float(k))
        out.append(bytes(st))
    kfs = odo.keyframes()
    odo.close()
    return out, kfs


def test_sequence_and_replay(fake, small_seq, orc, tmp_path):
    from replay import Replayer
    S = small_seq
    trace = str(tmp_path / "trace.bin")
    sts, kfs = _run(fake, S, 34, 120, trace)
    last = vo.VoStatus.from_buffer_copy(sts[-1])
    q, t = last.T_f_w.to_arrays()
    assert np.linalg.norm(t - S["T_f_w"][33][1]) < 0.03 and last.stage == 3
    assert len(kfs) >= 4 and max(vo.VoStatus.from_buffer_copy(b).n_candidates for b in sts) > 50
    rp = Replayer(orc)
    for call, r in vo.read_trace(trace):
        getattr(rp, call)(r)
    s = rp.stat
    print(s)
    assert s["track"]["n"] == 33 and s["pose"]["n"] == 33 and s["select"]["n"] == 33 and s["ba"]["n"] == len(kfs) - 1
    # the restatement replays itself: no decision may differ
    assert "iter_mismatch" not in s["track"] and "tie" not in s["reproject"] and "tie" not in s.get("seed", {}) and "tie" not in s.get("activate", {})
    assert s["seed"]["updated"] > 0.3 * s["seed"]["n"] and s["activate"]["n"] > 20 and s["detect"]["octree"] == len(kfs) + 1
    # the idle-time pass (observeDepthWithPreviousFrameOnce) ran: one sweep per frame while a keyframe's list of earlier frames lasts
    assert s["seed_previous"]["n"] > 500 and s["seed_previous"]["updated"] > 0.3 * s["seed_previous"]["n"] and "tie" not in s["seed_previous"]


def test_bank_of_sequences_equals_solo_runs(fake, small_seq):
    """three sequences of different lengths in one bank (one sits steps out) = each alone, status record by status record"""
    S = small_seq
    other = synth.sequence(22, spec=SMALL, seed=2031, workers=4, step=(0.04, 0.02, 0.015), rot_deg_per_frame=(0.08, -0.2, 0.1))
    runs = [(S, 15), (other, 13), (S, 8)]
    solo = [_run(fake, s, n, 80) for s, n in runs]
    bank = vo.MultiVisualOdometry(synth.camera(SMALL), 3, 80, lib=fake)
    bank.set_first_frames([s["images"][0] for s, _ in runs], [s["depth0"] for s, _ in runs])
    got = [[] for _ in runs]
    for k in range(1, 15):
        imgs = [s["images"][k] if k < n else None for s, n in runs]
        bank.add_images(imgs, [float(k)] * 3)
        for i, (s, n) in enumerate(runs):
            if k < n:
                got[i].append(bytes(bank.status(i)))
    for i in range(3):
        assert got[i] == solo[i][0], "sequence %d differs from its solo run" % i
        assert [(a, bytes(b), c) for a, b, c in bank.keyframes(i)] == [(a, bytes(b), c) for a, b, c in solo[i][1]]
    bank.close()


def test_worker_pool_hands_out_every_item_once(tmp_path):
    """The engine's parallel-for under stress (tests/pool_stress.cpp): two pools, 60 000 phases each of 2..121 near-empty items,
    with and without more workers than cores; every item exactly once, no hang (the timeout is the hang check)."""
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pool_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(here, "pool_stress.cpp"), "-o", exe])
    for threads in (3, 12):
        out = subprocess.run([exe, str(threads), "60000"], capture_output=True, text=True, timeout=240)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-400:]


def test_chain_state_record_reproduces_the_call(fake, small_seq, tmp_path):
    """hso_vo_trace_state (include/hso_vo.h): the recorded sequence map + job, rebuilt in a fresh context through the public
    hso_gpu_seqmap_* calls, gives the recorded result of hso_gpu_seq_chain bit for bit — i.e. the record holds everything the call
    reads.  Here on the restatement (the `-m gpu` half, tests/test_seq_chain.py, hands the same kind of record to the device)."""
    import chain_state as cs
    S = small_seq
    odo = vo.VisualOdometry(synth.camera(S["spec"]), 120, lib=fake)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    recs = []
    for k in range(1, 30):
        path = str(tmp_path / ("t%d.bin" % k))
        if k in (12, 29):
            odo.trace(path, state=True)
        odo.add_image(S["images"][k], float(k))
        if k in (12, 29):
            odo.trace(None)
            recs.append(vo.read_trace(path))
    n_kf = len(odo.keyframes())
    odo.close()
    assert n_kf >= 3
    lib = cs.ChainLib(fake)
    images = {k: S["images"][k] for k in range(30)}                 # one sequence: frame id = image index (hso_engine_impl.h: Seq::new_frame)
    for rec in recs:
        names = [n for n, _ in rec]
        assert names.count("seq_chain_state") == 1 and names.count("seq_chain_result") == 1
        st = cs.state_from_record(dict(rec)["seq_chain_state"])
        want = cs.result_from_record(dict(rec)["seq_chain_result"])
        assert len(st["kfs"]) >= 2 and len(st["points"]) > 100
        ls = cs.LoadedState(lib, st, images)
        back = ls.dump()                                            # what went in comes out: every table of the rebuilt map
        for key in ("kfs", "points", "obs", "obs_point", "key_points", "kf_nfts", "cands"):
            assert back[key].tobytes() == st[key].tobytes(), key
        assert all(a.tobytes() == b.tobytes() for a, b in zip(back["kf_fts"], st["kf_fts"]))
        got = ls.run()
        assert not cs.fields_differ(got["result"], want["result"])
        assert np.array_equal(got["events"], want["events"]) and got["features"].tobytes() == want["features"].tobytes()
        ls.close()
    assert len(st["kfs"]) >= 3 and len(st["cands"]) > 0 and want["result"]["n_events"] >= 0


def test_local_ba_state_record_reproduces_the_call(fake, small_seq, tmp_path):
    """the same for hso_gpu_seq_local_ba: the map recorded before a keyframe's local BA, rebuilt in a fresh context of the
    restatement, gives the recorded window, state and culling list bit for bit; and the restatement's window is the graph of
    src/bundle_adjustment.cpp:592-812 (checked here against an independent statement in numpy)."""
    import chain_state as cs
    S = small_seq
    odo = vo.VisualOdometry(synth.camera(S["spec"]), 120, lib=fake)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    recs = []
    for k in range(1, 30):
        path = str(tmp_path / ("b%d.bin" % k))
        odo.trace(path, state=True)
        odo.add_image(S["images"][k], float(k))
        odo.trace(None)
        r = dict(vo.read_trace(path))
        if "seq_ba_state" in r:
            recs.append(r)
    odo.close()
    assert len(recs) >= 2
    lib = cs.ChainLib(fake)
    for r in recs:
        st = cs.ba_state_from_record(r["seq_ba_state"])
        want = cs.ba_result_from_record(r["seq_ba_result"])
        ls = cs.LoadedState(lib, st, None)
        got = ls.run_ba(st["core"], st["fixed"], st["n_iter"], st["error_multiplier2"], st["chi2_corner"], st["chi2_edgelet"])
        ls.close()
        assert got["result"].tobytes() == want["result"].tobytes()
        assert np.array_equal(got["point_ids"], want["point_ids"]) and got["point_state"].tobytes() == want["point_state"].tobytes() and np.array_equal(got["culled"], want["culled"])
        W = got["window"]
        assert W["edges"].tobytes() == r["ba_optimize"]["edges"] and W["edge_obs"].tobytes() == r["ba_optimize"]["edge_obs"]
        # ---- the graph, restated: points of the core keyframes' features (ascending), vertices in order of first appearance
        core = [int(c) for c in st["core"]]
        pts = sorted({int(st["obs_point"][f]) for row in core for f in st["kf_fts"][row] if st["obs_point"][f] >= 0})
        assert pts == got["point_ids"].tolist()
        vertex = {row: v for v, row in enumerate(core)}
        edges = []
        for i, p in enumerate(pts):
            P = st["points"][p]
            vertex.setdefault(int(P["host_kf"]), len(vertex))
            row = int(P["obs_begin"])
            for _ in range(int(P["obs_count"])):
                ob = st["obs"][row]
                if ob["kf"] != P["host_kf"]:
                    vertex.setdefault(int(ob["kf"]), len(vertex))
                    edges.append((i, vertex[int(P["host_kf"])], vertex[int(ob["kf"])], row, 1 if ob["type"] == 1 else 0, int(ob["level"])))
                row = int(ob["pad_"])
        assert [row for row, v in sorted(vertex.items(), key=lambda kv: kv[1])] == W["vertex_rows"].tolist()
        E = W["edges"]
        assert [(int(e["point"]), int(e["host"]), int(e["target"])) for e in E] == [(a, b, c) for a, b, c, _, _, _ in edges]
        assert W["edge_obs"].tolist() == [e[3] for e in edges] and E["type"].tolist() == [e[4] for e in edges] and E["level"].tolist() == [e[5] for e in edges]
        ob = st["obs"][W["edge_obs"]]
        uv = ob["f"][:, :2] / ob["f"][:, 2:3]
        assert np.array_equal(W["obs_uv"].reshape(-1, 2), uv)
        corner = E["type"] == 0
        assert np.array_equal(E["meas"][corner], uv[corner]) and np.array_equal(E["meas"][~corner, 0], ob["grad"][~corner, 0] * uv[~corner, 0] + ob["grad"][~corner, 1] * uv[~corner, 1])
        assert np.array_equal(E["fH"], st["points"]["host_f"][got["point_ids"]][E["point"]])
        assert W["fixed"].tolist() == [int(x) for x in st["fixed"]] + [1] * (len(W["fixed"]) - len(core))
