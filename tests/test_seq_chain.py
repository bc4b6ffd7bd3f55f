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

from hso_amd import synth, vo

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
