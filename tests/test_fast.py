"""FAST-9 corner candidates (FeatureExtractor::fastDetect: fast_corner_detect_9_sse2 +
fast_corner_score_9 + fast_nonmax_3x3 + shiTomasiScore).

This row is PINNED by the reference's own code: tests/golden/fast9.json was produced by the
compiled thirdparty/fast library (tests/golden/make_fast_golden.py, oracle/_ref/libfast_ref.so).
CPU: the C restatement reproduces every golden case bit for bit (and the live library where it is
present).  GPU: the HIP path equals the restatement bit for bit on the golden images, on full
pyramids (VGA and EuRoC sizes) and on adversarial images (ties, flat, saturated)."""
import base64
import json
import os

import numpy as np
import pytest

from hso_amd import capi, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fast9.json")


@pytest.fixture(scope="module")
def gold():
    g = json.load(open(GOLD))
    imgs = {k: np.frombuffer(base64.b64decode(v["data"]), np.uint8).reshape(v["h"], v["w"]).copy()
            for k, v in g["images"].items()}
    return imgs, g["cases"]


def test_oracle_matches_reference_library_golden(orc, gold):
    imgs, cases = gold
    n_corners = 0
    for c in cases:
        img = imgs[c["image"]]
        xy, sc = orc.fast9_detect(img, c["threshold"])
        gxy = np.array(c["xy"], np.int16).reshape(-1, 2)
        assert xy.shape == gxy.shape and (xy == gxy).all(), (c["image"], c["threshold"])
        assert (sc == np.array(c["scores"], np.int32)).all()
        # non-max suppression + border (border 0 keeps everything fast_nonmax_3x3 keeps)
        out, n = orc.fast_detect_level(img, c["threshold"], border=0)
        keep = c["nonmax"]
        assert n == len(keep)
        assert [(int(o["x"]), int(o["y"]), int(o["score"])) for o in out] == \
               [(int(gxy[k, 0]), int(gxy[k, 1]), int(c["scores"][k])) for k in keep]
        n_corners += len(gxy)
    assert n_corners > 20000 and len(cases) == 21


def test_oracle_matches_live_reference_when_present(orc):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (75, 130), dtype=np.uint8)
    r = orc.ref_fast(img, 15)
    if r is None:
        pytest.skip("oracle/_ref/libfast_ref.so absent (reference not on this machine)")
    rxy, rsc, keep = r
    xy, sc = orc.fast9_detect(img, 15)
    assert (xy == rxy).all() and (sc == rsc).all()
    out, n = orc.fast_detect_level(img, 15, border=0)
    assert [(o["x"], o["y"]) for o in out] == [(rxy[k, 0], rxy[k, 1]) for k in keep]


def test_oracle_shi_tomasi_and_border(orc):
    img = np.zeros((40, 40), np.uint8)
    img[:, 20:] = 200                                   # vertical step edge: one zero eigenvalue
    assert orc.shi_tomasi(img, 20, 20) == 0.0
    img[20:, :] = np.where(img[20:, :] > 0, 0, 200)     # checkerboard corner: both eigenvalues large
    assert orc.shi_tomasi(img, 20, 20) > 1000
    assert orc.shi_tomasi(img, 4, 20) == 0.0            # box touches the image border -> 0 (vision.cpp:126)
    # corners in the border band are dropped after non-max suppression (feature_detection.cpp:573)
    rng = np.random.default_rng(8)
    noise = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    all_, n_all = orc.fast_detect_level(noise, 20, border=0)
    in_, n_in = orc.fast_detect_level(noise, 20, border=8)
    sel = [(o["x"], o["y"]) for o in all_ if not (o["x"] < 8 or o["x"] > 56 or o["y"] < 8 or o["y"] > 56)]
    assert n_in == len(sel) < n_all and [(o["x"], o["y"]) for o in in_] == sel


def _check_level(got, want):
    assert len(got) == len(want)
    for name in ("x", "y", "score"):
        assert (got[name] == want[name]).all(), name
    assert (got["response"].view(np.uint32) == want["response"].view(np.uint32)).all(), "Shi-Tomasi bits"


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [synth.ICL_NUIM, synth.EUROC], ids=["640x480", "752x480"])
def test_fast_detect_full_pyramid_bit_exact(gpu_ctx, orc, spec):
    d = synth.config2_pair(10, spec=spec, seed=31)
    fid = 9300
    gpu_ctx.frame_upload(fid, d["ref"])
    try:
        pyr = orc.create_pyramid(d["ref"])
        for thr in (10, 20):
            levels, counts = gpu_ctx.fast_detect(fid, n_levels=3, threshold=thr, border=8, cap=60000)
            for L in range(3):
                want, n = orc.fast_detect_level(pyr[L], thr, border=8)
                assert counts[L] == n and n > 50
                _check_level(levels[L], want)
        # cap smaller than the count: the first `cap` corners in raster order, the full count reported
        levels, counts = gpu_ctx.fast_detect(fid, n_levels=1, threshold=10, border=8, cap=100)
        want, n = orc.fast_detect_level(pyr[0], 10, border=8)
        assert counts[0] == n > 100 and len(levels[0]) == 100
        _check_level(levels[0], want[:100])
    finally:
        gpu_ctx.frame_release(fid)


@pytest.mark.gpu
def test_fast_detect_batch_equals_single(gpu_ctx):
    """Frames of independent sequences in one call: every frame equals its solo result."""
    frames = [synth.config2_pair(10, seed=40 + k)["ref"] for k in range(3)]
    ids = [9330, 9331, 9332, 9330]                       # a frame may appear twice
    for i, f in zip(ids[:3], frames):
        gpu_ctx.frame_upload(i, f)
    try:
        out, counts = gpu_ctx.fast_detect_batch(ids, n_levels=3, threshold=20, border=8, cap=8192)
        _, counts_only = gpu_ctx.fast_detect_batch(ids, n_levels=3, threshold=20, border=8, cap=0)
        assert (counts == counts_only).all() and counts.min() > 20
        for k, i in enumerate(ids):
            solo, c = gpu_ctx.fast_detect(i, n_levels=3, threshold=20, border=8, cap=8192)
            assert list(counts[k]) == c
            for L in range(3):
                assert out[k, L, :c[L]].tobytes() == solo[L].tobytes()
    finally:
        for i in ids[:3]:
            gpu_ctx.frame_release(i)


@pytest.mark.gpu
def test_fast_detect_adversarial_images(gpu_ctx, orc):
    """Ties in the non-max suppression, flat and saturated images, dense noise."""
    rng = np.random.default_rng(12)
    w, h = 640, 480
    imgs = {
        "four_levels": (rng.integers(0, 4, (h, w)) * 80).astype(np.uint8),
        "flat": np.full((h, w), 128, np.uint8),
        "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
        "saturated_blocks": (np.kron(rng.integers(0, 2, (h // 4, w // 4)), np.ones((4, 4))) * 255).astype(np.uint8),
    }
    fid = 9310
    for name, img in imgs.items():
        gpu_ctx.frame_upload(fid, img)
        try:
            levels, counts = gpu_ctx.fast_detect(fid, n_levels=1, threshold=20, border=8, cap=200000)
        finally:
            gpu_ctx.frame_release(fid)
        want, n = orc.fast_detect_level(img, 20, border=8)
        assert counts[0] == n, name
        _check_level(levels[0], want)
    assert True


@pytest.mark.gpu
def test_fast_detect_golden_and_errors(gpu_ctx, gold):
    """Directly against the reference library's outputs (no oracle in between): embed a golden image
    in a frame-sized canvas whose surroundings cannot create or suppress corners inside it."""
    imgs, cases = gold
    img = imgs["scene_crop_160x120"]
    case = next(c for c in cases if c["image"] == "scene_crop_160x120" and c["threshold"] == 20)
    canvas = np.zeros((128, 160), np.uint8)            # 160x128: a legal frame size (multiples of 16)
    canvas[:120, :] = img
    gxy = np.array(case["xy"], np.int16).reshape(-1, 2)
    sc = np.array(case["scores"])
    keep = [k for k in case["nonmax"] if gxy[k, 1] < 120 - 8]   # rows near the pasted edge see the zero padding
    gpu_ctx.frame_upload(9320, canvas)
    try:
        levels, counts = gpu_ctx.fast_detect(9320, n_levels=1, threshold=20, border=0, cap=20000)
        got = [(int(o["x"]), int(o["y"]), int(o["score"])) for o in levels[0] if o["y"] < 120 - 8]
        assert got == [(int(gxy[k, 0]), int(gxy[k, 1]), int(sc[k])) for k in keep] and len(got) > 100
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.fast_detect(424242)                                  # frame not resident
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.fast_detect(9320, n_levels=6)
    finally:
        gpu_ctx.frame_release(9320)


# ---- FAST-12: FeatureExtractor::fillingHole (src/feature_detection.cpp:1125-1154), the initialisation
# branch of detect.  PINNED like FAST-9: tests/golden/fast12.json holds the outputs of the compiled
# reference library (fast_12_detect.cpp, fast_12_score.cpp, nonmax_3x3.cpp; tests/golden/make_fast12_golden.py).
def test_oracle_fast12_matches_reference_library_golden(orc, gold):
    imgs, _ = gold
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fast12.json")))["cases"]
    n_corners = 0
    for c in cases:
        img = imgs[c["image"]]
        xy, sc = orc.fast_detect_arc(img, c["threshold"], 12)
        gxy = np.array(c["xy"], np.int16).reshape(-1, 2)
        assert xy.shape == gxy.shape and (xy == gxy).all(), (c["image"], c["threshold"])
        assert (sc == np.array(c["scores"], np.int32)).all()
        n_corners += len(gxy)
    assert n_corners > 10000 and len(cases) == 21


def test_oracle_fast12_matches_live_reference_when_present(orc):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (61, 97), dtype=np.uint8)
    r = orc.ref_fast12(img, 8)
    if r is None:
        pytest.skip("oracle/_ref/libfast_ref.so absent (reference not on this machine)")
    rxy, rsc, keep = r
    xy, sc = orc.fast_detect_arc(img, 8, 12)
    assert (xy == rxy).all() and (sc == rsc).all() and len(keep) > 20


def test_oracle_filling_hole_respects_occupancy(orc):
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    have = np.zeros(4800, np.uint8)
    free = orc.filling_hole_level(img, 0, 640, 480, 20, have)
    assert len(free) > 500 and have.sum() == len(free)        # one feature per grid index, each index now occupied
    idx = [orc.cell_index(c["x"], c["y"], 8, 80, 60) for c in free]
    assert len(set(idx)) == len(idx)
    # every index already taken stays silent; the others keep their first survivor
    have2 = np.zeros(4800, np.uint8)
    have2[idx[::2]] = 1
    part = orc.filling_hole_level(img, 0, 640, 480, 20, have2)
    assert part.tobytes() == free[1::2].tobytes()
    # barrier: max(0.6 * minThresh, 6) truncated to short -> 6 for minThresh <= 10, 12 for 20
    a = orc.filling_hole_level(img, 0, 640, 480, 7, np.zeros(4800, np.uint8))
    b = orc.filling_hole_level(img, 0, 640, 480, 10, np.zeros(4800, np.uint8))
    assert a.tobytes() == b.tobytes() and len(a) >= len(free)


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [synth.ICL_NUIM, synth.EUROC], ids=["640x480", "752x480"])
def test_detect_candidates_init_bit_exact(gpu_ctx, orc, spec):
    base = synth.config2_pair(10, spec=spec, seed=90)["ref"].astype(np.float32)
    # lower contrast leaves grid indices without a FAST-9 corner for fillingHole's FAST-12 pass to fill
    frames = [(base * s + 128 * (1 - s)).astype(np.uint8) for s in (1.0, 0.5, 0.35)]
    rng = np.random.default_rng(3)
    frames.append((np.kron(rng.integers(0, 2, (spec["height"] // 8, spec["width"] // 8)), np.ones((8, 8))) * 60 + 50).astype(np.uint8))
    ids = [9350 + k for k in range(len(frames))]
    for i, f in zip(ids, frames):
        gpu_ctx.frame_upload(i, f)
    try:
        h, w = frames[0].shape
        for thr in (7, 20):
            co, cc, fo, fc = gpu_ctx.detect_candidates_init(ids, n_levels=3, min_thresh=thr, corner_cap=30000, fill_cap=6000)
            for k, img in enumerate(frames):
                pyr = orc.create_pyramid(img)
                have0 = None
                for L in range(3):
                    want, n = orc.fast_detect_level(np.ascontiguousarray(pyr[L]), thr, border=8)
                    assert cc[k, L] == n
                    _check_level(co[k, L, :n], want)
                    if L == 0:
                        g, gc, gr = orc.detect_grid(w, h, 0)
                        have0 = np.zeros(gc * gr, np.uint8)
                        for c in want:
                            have0[orc.cell_index(c["x"], c["y"], g, gc, gr)] = 1
                fill = orc.filling_hole_level(np.ascontiguousarray(pyr[0]), 0, w, h, thr, have0)
                assert fc[k] == len(fill), (thr, k)
                _check_level(fo[k, :fc[k]], fill)
            assert fc.max() > (50 if thr == 20 else 0)
        # caps smaller than the counts
        co, cc, fo, fc2 = gpu_ctx.detect_candidates_init(ids[:1], n_levels=1, min_thresh=20, corner_cap=10, fill_cap=10)
        assert fc2[0] == fc[0] and fo[0, :min(10, fc2[0])].tobytes() == orc.filling_hole_level(
            np.ascontiguousarray(orc.create_pyramid(frames[0])[0]), 0, w, h, 20,
            _have_after_fast(orc, frames[0], 20, w, h))[:10].tobytes()
        co, cc, fo, fc2 = gpu_ctx.detect_candidates_init(ids[1:2], n_levels=1, min_thresh=20, corner_cap=10, fill_cap=10)
        assert fc2[0] == fc[1] > 10 and fo[0].tobytes() == orc.filling_hole_level(
            np.ascontiguousarray(orc.create_pyramid(frames[1])[0]), 0, w, h, 20, _have_after_fast(orc, frames[1], 20, w, h))[:10].tobytes()
    finally:
        for i in ids:
            gpu_ctx.frame_release(i)


def _have_after_fast(orc, img, thr, w, h):
    want, _ = orc.fast_detect_level(np.ascontiguousarray(orc.create_pyramid(img)[0]), thr, border=8)
    g, gc, gr = orc.detect_grid(w, h, 0)
    have = np.zeros(gc * gr, np.uint8)
    for c in want:
        have[orc.cell_index(c["x"], c["y"], g, gc, gr)] = 1
    return have
