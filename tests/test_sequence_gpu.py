"""A short synthetic sequence through the tracker the way FrameHandlerMono::processFrame chains
frames (src/frame_handler_mono.cpp:173-209): every new frame is tracked against the keyframe
with the previous frame's pose as the motion-model start.  Reports the metrics of SURVEY.md
section 8(d): per-frame SE(3) difference between the HIP path and the CPU restatement, and the
trajectory error against the scene's ground truth (ATE, RMSE of camera positions)."""
import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_sequence_tracking_matches_oracle_and_ground_truth(gpu_ctx, orc, cam):
    sc = synth.Scene(synth.ICL_NUIM, 4321)
    n_frames = 6
    qi, ti = np.array([0, 0, 0, 1.0]), np.zeros(3)
    key = sc.render(qi, ti, 1.0, 1.0, 100)
    feats = sc.features(qi, ti, 800, seed=101)
    # smooth camera motion: constant twist per frame (~1.5 cm, 0.25 deg)
    rng = np.random.default_rng(7)
    tdir = rng.normal(size=3); tdir /= np.linalg.norm(tdir)
    rdir = rng.normal(size=3); rdir /= np.linalg.norm(rdir)
    truth, frames = [], []
    for k in range(1, n_frames + 1):
        q = synth.rotvec_to_quat(rdir * np.deg2rad(0.25 * k))
        t = tdir * 0.015 * k
        truth.append((q, t))
        frames.append(sc.render(q, t, 1.0 + 0.01 * k, 1.0, 200 + k))
    gpu_ctx.frame_upload(9400, key)
    kp = orc.create_pyramid(key)
    p = capi.TrackParams(0, 4, 1, 50)
    T_prev_g, T_prev_o = capi.SE3.identity(), capi.SE3.identity()
    a_g = a_o = 1.0
    pos_err, rot_diff, tra_diff = [], [], []
    try:
        for k, img in enumerate(frames):
            fid = 9401 + k
            gpu_ctx.frame_upload(fid, img)
            rg = gpu_ctx.coarse_track_batch(cam, p, [gpu_ctx.make_job(9400, fid, feats, T_prev_g, a_g)])[0]
            ro = orc.Tracker(cam, p, kp, orc.create_pyramid(img), feats).run(T_prev_o, a_o)
            gpu_ctx.frame_release(fid)
            assert list(rg.iters) == list(ro.iters), "frame %d" % k
            qg, tg = rg.T_cur_ref.to_arrays()
            qo, to = ro.T_cur_ref.to_arrays()
            if qg @ qo < 0:
                qg = -qg
            rot_diff.append(2 * np.linalg.norm(qg - qo)); tra_diff.append(np.linalg.norm(tg - to))
            # camera position in the keyframe's frame: -R^T t
            Rg = synth.quat_to_R(qg)
            Rt = synth.quat_to_R(truth[k][0])
            pos_err.append(np.linalg.norm(-Rg.T @ tg - (-Rt.T @ truth[k][1])))
            T_prev_g, T_prev_o, a_g, a_o = rg.T_cur_ref, ro.T_cur_ref, rg.exposure_rat, ro.exposure_rat
    finally:
        gpu_ctx.frame_release(9400)
    # HIP path vs CPU restatement, frame by frame (errors do not accumulate: each frame starts
    # from its own previous estimate and the two stay within rounding of each other)
    assert max(rot_diff) <= 2e-6 and max(tra_diff) <= 8e-6, (rot_diff, tra_diff)
    # trajectory against ground truth: ATE (RMSE of positions) well below a millimetre-scale bound
    ate = float(np.sqrt(np.mean(np.square(pos_err))))
    assert ate < 2e-3, pos_err


@pytest.mark.gpu
def test_two_handles_with_different_max_fts_do_not_interfere():
    """Config is per handle (the reference's is a process singleton): a 2000-feature handle created next to a 200-feature one
    leaves the first one's results exactly as they are alone."""
    from hso_amd import vo
    spec = synth.EUROC
    cam = synth.camera(spec)
    S = synth.sequence(14, spec=spec)

    def run(odo):
        odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
        return [bytes(odo.add_image(S["images"][k], float(k))) for k in range(1, 14)]
    a = vo.VisualOdometry(cam, 200)
    alone = run(a)
    a.close()
    a = vo.VisualOdometry(cam, 200)
    b = vo.VisualOdometry(cam, 2000)                     # created after a: with a process-wide Config this changed a's max_fts
    a.set_first_frame(S["images"][0], S["depth0"], 0.0)
    b.set_first_frame(S["images"][0], S["depth0"], 0.0)
    mixed, nb = [], []
    for k in range(1, 14):                                # interleaved calls
        mixed.append(bytes(a.add_image(S["images"][k], float(k))))
        nb.append(b.add_image(S["images"][k], float(k)).n_matches)
    a.close(); b.close()
    # frame / keyframe / point counters are per THREAD (the reference's are process globals): ids differ, everything else must not
    def strip(rec):
        st = vo.VoStatus.from_buffer_copy(rec)
        return (bytes(st.T_f_w), st.stage, st.result, st.n_features, st.n_inliers, st.n_tracked, st.n_matches, st.n_trials, st.n_seeds,
                st.n_candidates, st.is_keyframe, st.pose_error_final)
    assert [strip(r) for r in mixed] == [strip(r) for r in alone]
    assert max(nb) > 1000                                 # b really ran with its own budget
