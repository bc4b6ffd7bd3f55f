"""hso_vo_multi_* (the sequence engine, hso_amd/host/hso_engine*.cpp): N sequences over one device context in lockstep, every numeric
stage of a step running as one batched C-ABI call for all of them.  The claim to check: a sequence run inside the multi-sequence driver equals
the same sequence run alone through hso_vo_* bit for bit — every per-frame status record (pose, exposure, match / seed / keyframe
counters, pose-optimisation and BA errors) and the keyframe trajectory — although its tracker jobs, reprojections, pose
optimisations, seed updates, activations and BA windows travelled in batches with the other sequences' (BASELINE configs[4]:
8 sequences, here 4 with different scenes, lengths and keyframe timing)."""
import ctypes as C

import numpy as np
import pytest

from hso_amd import synth, vo

pytestmark = pytest.mark.gpu


def _status_bytes(st):
    return bytes(st)


def test_four_sequences_in_lockstep_equal_four_single_runs():
    spec = synth.EUROC
    cam = synth.camera(spec)
    lengths = [34, 40, 28, 40]
    seqs = [synth.sequence(n, spec=spec, seed=3100 + 17 * k, step=(0.016 + 0.002 * k, 0.005, 0.007 - 0.001 * k)) for k, n in enumerate(lengths)]
    # ---- reference: each sequence alone
    solo_status, solo_kfs = [], []
    for S in seqs:
        odo = vo.VisualOdometry(cam, 200)
        odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
        sts = [_status_bytes(odo.status())]
        for k in range(1, len(S["images"])):
            sts.append(_status_bytes(odo.add_image(S["images"][k], float(k))))
        solo_status.append(sts)
        solo_kfs.append([(ts, bytes(T), fid) for ts, T, fid in odo.keyframes()])
        odo.close()
    # ---- the four in lockstep over one context
    multi = vo.MultiVisualOdometry(cam, len(seqs), 200)
    multi.set_first_frames([S["images"][0] for S in seqs], [S["depth0"] for S in seqs])
    got = [[_status_bytes(multi.status(q))] for q in range(len(seqs))]
    for k in range(1, max(lengths)):
        imgs = [S["images"][k] if k < len(S["images"]) else None for S in seqs]       # shorter sequences sit the later steps out
        multi.add_images(imgs, [float(k)] * len(seqs))
        for q, im in enumerate(imgs):
            if im is not None:
                got[q].append(_status_bytes(multi.status(q)))
    counts = multi.call_counts()
    kfs = [[(ts, bytes(T), fid) for ts, T, fid in multi.keyframes(q)] for q in range(len(seqs))]
    multi.close()
    n_kf = 0
    for q in range(len(seqs)):
        assert len(got[q]) == len(solo_status[q]) == lengths[q]
        for k, (a, b) in enumerate(zip(got[q], solo_status[q])):
            assert a == b, "sequence %d frame %d differs from its solo run" % (q, k)
        assert kfs[q] == solo_kfs[q] and len(kfs[q]) >= 2
        n_kf += len(kfs[q])
    # the calls really travelled together: far fewer tracker / pose / seed launches than frames, several requests per launch
    n_frames = sum(lengths) - len(seqs)
    for kind in ("track", "reproject_select_pose", "frame_upload"):
        calls, items = counts[kind]
        assert items >= n_frames and calls <= max(lengths) + 8 and items / calls > 2.5, (kind, counts)
    calls, items = counts["seed_observe"]          # only frames that have seeds to update (not the keyframes themselves)
    assert items > n_frames // 2 and items / calls > 2.0, counts
    assert counts["ba"][1] >= n_kf - len(seqs) - 4 and counts["seed_activate"][1] > 0 and counts["other"][0] > 0
    print("multi-sequence driver: batched calls / requests per kind", counts)


def test_two_sequences_started_from_images_equal_their_single_runs():
    """hso_vo_multi_start: both sequences run the two-view initialisation (at different frames: different speeds) inside the
    lockstep driver — its KLT calls one per sequence — and every status record equals the solo run's."""
    spec = dict(synth.EUROC, texture_om=((0.004, 0.05), (0.05, 0.6)))            # see tests/test_init.py: init_seq
    cam = synth.camera(spec)
    steps = [(0.05, 0.015, 0.01), (0.07, 0.01, 0.015)]
    seqs = [synth.sequence(22, spec=spec, seed=4100 + 13 * k, step=steps[k], rot_deg_per_frame=(0.05, -0.1, 0.03)) for k in range(2)]
    solo = []
    for S in seqs:
        odo = vo.VisualOdometry(cam, 200)
        odo.start()
        solo.append([_status_bytes(odo.add_image(im, float(k))) for k, im in enumerate(S["images"])])
        odo.close()
    multi = vo.MultiVisualOdometry(cam, 2, 200)
    multi.start()
    got = [[], []]
    for k in range(22):
        multi.add_images([S["images"][k] for S in seqs], [float(k)] * 2)
        for q in range(2):
            got[q].append(_status_bytes(multi.status(q)))
    stages = [[multi.status(q).stage for q in range(2)]]
    multi.close()
    for q in range(2):
        for k, (a, b) in enumerate(zip(got[q], solo[q])):
            assert a == b, "sequence %d frame %d differs from its solo run" % (q, k)
    init_at = [next(k for k, b in enumerate(got[q]) if vo.VoStatus.from_buffer_copy(b).stage == 3) for q in range(2)]
    assert init_at[0] != init_at[1] and stages[0] == [3, 3], (init_at, stages)


def test_depth_filter_stream_gives_the_results_of_the_synchronous_pass(monkeypatch):
    """The idle-time pass of the depth filter (observeDepthWithPreviousFrameOnce) runs on its own stream beside the next frame's
    tracking and is collected before the first thing that reads seeds: every status record and keyframe equals the run in which
    the pass executes inside the step (hso_vo_options.sync_previous), bit for bit."""
    spec = synth.EUROC
    cam = synth.camera(spec)
    seqs = [synth.sequence(30, spec=spec, seed=3300 + 11 * k, step=(0.018 + 0.002 * k, 0.005, 0.006)) for k in range(3)]

    def run(sync_previous=False):
        multi = vo.MultiVisualOdometry(cam, len(seqs), 300)
        multi.set_options(sync_previous=sync_previous)
        multi.set_first_frames([S["images"][0] for S in seqs], [S["depth0"] for S in seqs])
        got = []
        for k in range(1, 30):
            multi.add_images([S["images"][k] for S in seqs], [float(k)] * len(seqs))
            got.append([_status_bytes(multi.status(q)) for q in range(len(seqs))])
        kfs = [[(ts, bytes(T), fid) for ts, T, fid in multi.keyframes(q)] for q in range(len(seqs))]
        counts = multi.call_counts()
        multi.close()
        return got, kfs, counts

    overlapped, kfs_o, counts = run()
    inside, kfs_i, _ = run(sync_previous=True)
    assert overlapped == inside and kfs_o == kfs_i
    assert all(len(k) >= 2 for k in kfs_o) and counts["other"][0] > 10        # the pass ran (it is counted with the other calls)


def test_bank_of_96_at_2000_features_equals_solo_runs(monkeypatch):
    """The end-to-end figure's shape — a bank of 96 sequences at 2000 features — against the same sequences alone, status record by
    status record.  Both runs keep the tracker on its one-workgroup-per-job shape (what a bank that shares the device runs,
    hso_gpu_set_shared_device; a lone sequence would otherwise split its job over workgroups, whose partial sums add in another
    order: equal within the tracker's tolerance only)."""
    spec = synth.EUROC
    cam = synth.camera(spec)
    n_frames = 18
    seqs = [synth.sequence(n_frames, spec=spec, seed=5200 + 19 * k, step=(0.016 + 0.003 * k, 0.005, 0.007)) for k in range(4)]
    solo = []
    for S in seqs:
        odo = vo.VisualOdometry(cam, 2000)
        odo.set_options(track_no_coop=True)
        odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
        solo.append([_status_bytes(odo.add_image(S["images"][k], float(k))) for k in range(1, n_frames)])
        odo.close()
    bank = vo.MultiVisualOdometry(cam, 96, 2000)
    bank.set_options(track_no_coop=True)
    pick = [seqs[q % 4] for q in range(96)]
    bank.set_first_frames([S["images"][0] for S in pick], [S["depth0"] for S in pick])
    for k in range(1, n_frames):
        bank.add_images([S["images"][k] for S in pick], [float(k)] * 96)
        for q in (0, 1, 2, 3, 49, 94, 95):
            assert _status_bytes(bank.status(q)) == solo[q % 4][k - 1], "sequence %d (scene %d) frame %d differs from its solo run" % (q, q % 4, k)
    assert all(bank.status(q).n_matches >= 1500 for q in range(96))
    bank.close()


def test_traced_sequence_inside_three_concurrent_banks_replays(orc, tmp_path):
    """What `sequences_frames_per_s` times: three engines of 96 sequences at 2000 features side by side on one device, each on its own
    thread / context / stream.  One sequence of the middle bank records its device calls; the trace replays through the restatement
    under the margin rules of tests/replay.py — the concurrent banks' kernels and copies interleave with it on the device."""
    import threading
    from replay import Replayer
    spec = synth.EUROC
    cam = synth.camera(spec)
    n_frames, n_seq = 14, 96
    seqs = [synth.sequence(n_frames, spec=spec, seed=5300 + 23 * k, step=(0.017 + 0.002 * k, 0.005, 0.007)) for k in range(4)]
    lib = vo.load()
    lib.hso_vo_host_share(3)
    trace = str(tmp_path / "trace.bin")
    errors, last = [], [None] * 3
    gate = threading.Barrier(3)

    def work(b):
        try:
            m = vo.MultiVisualOdometry(cam, n_seq, 2000)
            pick = [seqs[(q + b) % 4] for q in range(n_seq)]
            if b == 1:
                m.trace(37, trace)
            m.set_first_frames([S["images"][0] for S in pick], [S["depth0"] for S in pick])
            gate.wait()
            for k in range(1, n_frames):
                m.add_images([S["images"][k] for S in pick], [float(k)] * n_seq)
            last[b] = [m.status(q) for q in range(n_seq)]
            m.close()
        except Exception as e:   # noqa: BLE001
            errors.append((b, repr(e)))
            try:
                gate.abort()
            except Exception:   # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(b,)) for b in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    lib.hso_vo_host_share(1)
    assert not errors, errors
    for b in range(3):
        assert all(st.stage == 3 and st.result != 2 and st.n_matches >= 1500 for st in last[b])
    rp = Replayer(orc)
    for call, r in vo.read_trace(trace):
        getattr(rp, call)(r)
    s = rp.stat
    print("traced sequence inside 3 x 96 x 2000:", s)
    assert s["track"]["n"] == n_frames - 1 and s["pose"]["n"] == n_frames - 1 and s["select"]["n"] == n_frames - 1
    assert s["reproject"]["success"] >= 0.8 * s["reproject"]["matched_calls"]


def test_engine_threads_stay_on_the_devices_numa_node():
    """hso_vo_options.no_numa_pin: by default the thread that drives a handle runs, from its first step on, on the CPUs of the NUMA
    node the device is attached to (hso_gpu_device_cpulist ∩ the affinity it had); with the option set nothing changes.  Results do not
    depend on it (same status records)."""
    import os
    import threading
    from hso_amd import capi
    ctx = capi.Context()
    node = ctx.device_cpulist()
    ctx.close()
    before = os.sched_getaffinity(0)
    expect = node & before if len(node & before) >= 2 and (node & before) != before else before
    small = dict(synth.EUROC, width=384, height=256, fx=240.0, fy=240.0, cx=191.5, cy=127.5)
    S = synth.sequence(6, spec=small, workers=4)
    cam = synth.camera(small)
    seen, status = {}, {}

    def run(tag, no_pin):
        odo = vo.VisualOdometry(cam, 120)
        if no_pin:
            odo.set_options(no_numa_pin=True)
        odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
        status[tag] = [bytes(odo.add_image(S["images"][k], float(k))) for k in range(1, 6)]
        seen[tag] = os.sched_getaffinity(0)                         # of this thread
        odo.close()
    for tag, no_pin in (("left alone", True), ("pinned", False)):   # a thread each: the pinned one keeps its affinity
        t = threading.Thread(target=run, args=(tag, no_pin))
        t.start(); t.join()
    assert seen["left alone"] == before
    assert seen["pinned"] == expect, (sorted(seen["pinned"])[:8], sorted(expect)[:8], len(node))
    assert status["pinned"] == status["left alone"]
    assert os.sched_getaffinity(0) == before                        # the test's own thread was never touched
