"""The C++ host mirror (hso_amd/host: reference class names over the C-ABI) driven the way
FrameHandlerMono::processFrame drives the reference, compared with the direct C-ABI call."""
import os
import subprocess

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu
HOST_EXE = os.path.join(os.path.dirname(capi.__file__), "host", "hso_host_test")


@pytest.mark.parametrize("inverse", [0, 1])
def test_coarse_tracker_adapter_matches_cabi(gpu_ctx, tmp_path, pair200, cam, inverse):
    d = pair200
    n = len(d["feats"])
    feats = d["feats"].copy()
    idist = 1.0 / feats["dist"]
    idist[::17] = -1.0                      # features without a point keep their slot
    tab = np.zeros((n, 6))
    tab[:, 0:2], tab[:, 2:5], tab[:, 5] = feats["px"], feats["f"], idist
    case = tmp_path / "case.bin"
    with open(case, "wb") as f:
        f.write(np.array([640, 480, n, inverse], np.int32).tobytes())
        f.write(np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float64).tobytes())
        f.write(d["ref"].tobytes()); f.write(d["cur"].tobytes()); f.write(tab.tobytes())
    out = subprocess.run([HOST_EXE, str(case)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    v = out.stdout.split()
    threw, n_tracked = int(v[0]), int(v[1])
    q, t = np.array(v[2:6], float), np.array(v[6:9], float)
    a, exposure_time = float(v[9]), float(v[10])
    iters = [int(x) for x in v[11:16]]
    ii_ref, ii_cur = float(v[16]), float(v[17])
    assert threw == 1                       # wrong image size -> std::runtime_error (frame.cpp:85-86)

    # the same job through the C-ABI directly; dist = |f / idist| as makeDepthRef computes it
    # for a point hosted in the reference frame itself (CoarseTracker.cpp:219-235)
    for i in (41, 42):
        try:
            gpu_ctx.frame_release(i)
        except capi.HsoGpuError:
            pass
    st_r, st_c = gpu_ctx.frame_upload(41, d["ref"]), gpu_ctx.frame_upload(42, d["cur"])
    assert (st_r.integral_image, st_c.integral_image) == pytest.approx((ii_ref, ii_cur), rel=1e-7)
    f2 = feats.copy()
    p = feats["f"] * (1.0 / np.where(idist > 0, idist, 1.0))[:, None]
    f2["dist"] = np.where(idist > 0, np.linalg.norm(p, axis=1), -1.0)
    a0 = float(np.float32(st_c.integral_image) / np.float32(st_r.integral_image))
    r = gpu_ctx.coarse_track_batch(cam, capi.TrackParams(inverse, 4, 1, 50),
                                   [gpu_ctx.make_job(41, 42, f2, capi.SE3.identity(), a0)])[0]
    assert iters == list(r.iters) and n_tracked == r.n_tracked
    # identity ref pose: cur.T_f_w_ = T_cur_ref * I; dist differs from the adapter's by fp64 rounding only
    assert np.allclose(q, r.T_cur_ref.q[:], atol=1e-9) and np.allclose(t, r.T_cur_ref.t[:], atol=1e-8)
    assert a == pytest.approx(r.exposure_rat, abs=1e-6)
    # write-back rule of CoarseTracker.cpp:200-202 with ref exposure time 1.0
    assert exposure_time == (1.0 if 0.99 < a < 1.01 else pytest.approx(a, rel=1e-6))
