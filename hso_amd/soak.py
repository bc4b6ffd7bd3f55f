"""`python -m hso_amd.soak [n_frames]`: a longer rendered sequence through the C++ driver, started from images (hso_vo_start), at 200 and
2000 features: initialisation frame, tracking failures, keyframes, ATE against the renderer's ground truth, per-frame wall time; and the
call time of the initialisation's KLT entry point.  Measurement helper, not part of the product path."""
import numpy as np, time, sys
from hso_amd import synth, vo, formats, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
spec = dict(synth.EUROC, texture_om=((0.004, 0.05), (0.05, 0.6)))
S = synth.sequence(n_frames=n, spec=spec, step=(0.03, 0.008, 0.006), rot_deg_per_frame=(0.04, -0.08, 0.02), workers=32)
cam = synth.camera(spec)
# KLT timing
ctx = capi.Context(0)
ctx.frame_upload(1, S["images"][0]); ctx.frame_upload(2, S["images"][3])
rng = np.random.default_rng(0)
px = np.stack([rng.uniform(20, 732, 2000), rng.uniform(20, 460, 2000)], 1).astype(np.float32)
for _ in range(3): ctx.klt_track(1, 2, px, px)
t0 = time.perf_counter()
for _ in range(20): r = ctx.klt_track(1, 2, px, px)
print("klt_track 2000 points: %.3f ms per call, tracked %d" % ((time.perf_counter() - t0) / 20 * 1e3, int(((r["status"] & 3) == 3).sum())))
ctx.close()
for max_fts in (200, 2000):
    odo = vo.VisualOdometry(cam, max_fts)
    odo.start()
    est, stages, res, tms = [], [], [], []
    for k, img in enumerate(S["images"]):
        t0 = time.perf_counter()
        st = odo.add_image(img, float(k))
        tms.append(time.perf_counter() - t0)
        if st.stage == 0: odo.start()
        est.append((np.array(st.T_f_w.q), np.array(st.T_f_w.t))); stages.append(st.stage); res.append(st.result)
    k_init = stages.index(3) if 3 in stages else -1
    gt = np.array([-(synth.quat_to_R(q).T @ t) for q, t in S["T_f_w"]]); ex = np.array([-(synth.quat_to_R(q).T @ t) for q, t in est])
    idx = np.arange(k_init, n)
    rmse, scale, _, _ = formats.ate_rmse(gt[idx], ex[idx])
    print("max_fts", max_fts, "init at", k_init, "failures", sum(1 for s, r in zip(stages[k_init:], res[k_init:]) if s != 3 or r == 2), "keyframes", len(odo.keyframes()),
          "ATE %.4f m over %.2f m (scale %.3f)" % (rmse, np.linalg.norm(gt[-1] - gt[k_init]), scale),
          "ms/frame median %.2f p95 %.2f max %.2f" % (np.median(tms) * 1e3, np.percentile(tms, 95) * 1e3, max(tms) * 1e3), "seeds", st.n_seeds, "cand", st.n_candidates)
    odo.close()
