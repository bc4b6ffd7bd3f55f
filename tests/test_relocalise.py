"""Losing track and finding it again: FrameHandlerMono::relocalizeFrame (reference src/frame_handler_mono.cpp:357-407) and the
RESULT_FAILURE -> STAGE_RELOCALIZING transition of FrameHandlerBase::finishFrameProcessingCommon (src/frame_handler_base.cpp:140-147).

A rendered sequence is interrupted by textureless frames (nothing to match: fewer than Config::qualityMinFts() reprojected points ->
RESULT_FAILURE -> the handler relocalises); when the scene comes back, the LAST frame is aligned against the closest keyframe with
the relocalisation tracker (inverse compositional, levels 4..0, 15 iterations), and with more than 30 tracked features the new frame
goes through the normal path with that keyframe as its reference (:366-386) until tracking is good again.

CPU: the engine over the restatement (tests/fakegpu), every recorded device call replayed exactly.  GPU (-m gpu): the same run on the
device, replayed through the restatement under the margin rules of tests/replay.py."""
import os
import subprocess

import numpy as np
import pytest

from hso_amd import synth, vo

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = dict(synth.EUROC, width=384, height=256, fx=240.0, fy=240.0, cx=191.5, cy=127.5)
MOTION = dict(step=(0.02, 0.006, 0.008), rot_deg_per_frame=(0.04, -0.12, 0.03))
STAGE_RUNNING, STAGE_RELOC = 3, 4
RESULT_FAILURE = 2


def _lose_and_find(lib, S, max_fts, trace, n_good, n_blank, n_after):
    """-> list of (frame index or None for a blank one, VoStatus)"""
    odo = vo.VisualOdometry(synth.camera(S["spec"]), max_fts, lib=lib)
    if trace:
        odo.trace(trace)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    h, w = S["images"][0].shape
    blank = np.full((h, w), 117, np.uint8)
    out = []
    k = 1
    for _ in range(n_good):
        out.append((k, odo.add_image(S["images"][k], float(k)))); k += 1
    for b in range(n_blank):
        out.append((None, odo.add_image(blank, float(k) + 0.1 * b)))
    for _ in range(n_after):
        out.append((k, odo.add_image(S["images"][k], float(k)))); k += 1
    kfs = odo.keyframes()
    odo.close()
    return out, kfs


def _check_story(out, S, n_good, n_blank):
    good, blank, after = out[:n_good], out[n_good:n_good + n_blank], out[n_good + n_blank:]
    assert all(st.stage == STAGE_RUNNING and st.result != RESULT_FAILURE for _, st in good)
    # the first textureless frame fails (no matches), and from then on the handler relocalises
    assert blank[0][1].result == RESULT_FAILURE and blank[0][1].stage == STAGE_RELOC, (blank[0][1].result, blank[0][1].stage)
    assert all(st.stage == STAGE_RELOC and st.result == RESULT_FAILURE for _, st in blank)
    # "reset to avoid crazy pose jumps" (:228-229): the frame that lost track keeps the pose of the frame before it.  (The frames
    # after it go through relocalizeFrame, which first re-aligns the LAST frame — a textureless one here — against the keyframe
    # and resets to that pose, :371-384: wherever the tracker leaves a frame without texture; only finiteness is asserted.)
    q_last, t_last = good[-1][1].T_f_w.to_arrays()
    q, t = blank[0][1].T_f_w.to_arrays()
    assert np.allclose(t, t_last, atol=1e-12) and np.allclose(q, q_last, atol=1e-12)
    for _, st in blank:
        q, t = st.T_f_w.to_arrays()
        assert np.isfinite(t).all() and np.isfinite(q).all()
    # the scene is back: within a few frames the handler tracks again and stays on the trajectory
    stages = [st.stage for _, st in after]
    assert STAGE_RUNNING in stages, stages
    first_ok = stages.index(STAGE_RUNNING)
    assert first_ok <= 2 and all(s == STAGE_RUNNING for s in stages[first_ok:]), stages
    for k, st in after[first_ok:]:
        q, t = st.T_f_w.to_arrays()
        assert np.linalg.norm(t - S["T_f_w"][k][1]) < 0.02, (k, np.linalg.norm(t - S["T_f_w"][k][1]))
        assert st.n_matches >= 60
    return first_ok


def _replay(orc, trace):
    from replay import Replayer
    rp = Replayer(orc)
    names = []
    for call, r in vo.read_trace(trace):
        names.append(call)
        getattr(rp, call)(r)
    return rp, names


def test_lose_track_and_relocalise_cpu(orc, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fakegpu")])
    fake = vo.load_from(os.path.join(HERE, "fakegpu", "libhso_host_fake.so"))
    S = synth.sequence(30, spec=SMALL, workers=4, **MOTION)
    trace = str(tmp_path / "trace.bin")
    out, kfs = _lose_and_find(fake, S, 120, trace, n_good=12, n_blank=3, n_after=10)
    first_ok = _check_story(out, S, 12, 3)
    rp, names = _replay(orc, trace)
    s = rp.stat
    # the relocalisation tracker ran (levels 4..0, 15 iterations: one extra coarse_track record per relocalising step) and the
    # restatement replays itself: no decision may differ
    assert names.count("coarse_track") > 12 + 10 and "iter_mismatch" not in s["track"] and "tie" not in s["reproject"]
    assert s["track"].get("reloc", 0) >= 1 + first_ok, s["track"]


@pytest.mark.gpu
def test_lose_track_and_relocalise_gpu(orc, tmp_path):
    S = synth.sequence(34, spec=synth.EUROC, workers=4, **MOTION)
    trace = str(tmp_path / "trace.bin")
    out, kfs = _lose_and_find(None, S, 200, trace, n_good=14, n_blank=3, n_after=12)
    first_ok = _check_story(out, S, 14, 3)
    rp, names = _replay(orc, trace)
    s = rp.stat
    assert s["track"].get("reloc", 0) >= 1 + first_ok and s["pose"]["n"] >= 14 + 12 - 1, s
