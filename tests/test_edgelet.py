"""Corner + edgelet candidates of a keyframe (FeatureExtractor::detect, non-init branch up to the
oct-tree: fastDetectMT + edgeLetDetectMT, src/feature_detection.cpp:408-447, 518-545, 749-830).

The FAST half is pinned by the reference's own library (test_fast.py).  The edgelet half calls
cv::Canny, an absent dependency: PARITY UNPINNED — the C restatement is checked here against a
second, independently written numpy restatement of the published algorithm (vectorised
non-maximum suppression + connected-component hysteresis) and against hand-built cases; the HIP
path equals the C restatement bit for bit."""
import numpy as np
import pytest
from scipy import ndimage

from hso_amd import capi, synth


def _canny_numpy(gx, gy, low_t, high_t):
    """cv::Canny(dx, dy, low, high, L2gradient=True), written differently from the C oracle."""
    lo, hi = min(low_t, high_t), max(low_t, high_t)
    lo, hi = min(32767.0, lo), min(32767.0, hi)
    low, high = int(np.floor(lo * lo)), int(np.floor(hi * hi))
    gx, gy = gx.astype(np.int64), gy.astype(np.int64)
    m = gx * gx + gy * gy
    p = np.pad(m, 1)
    h, w = m.shape

    def nb(dx, dy):
        return p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ax, ay = np.abs(gx), np.abs(gy) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    s_neg = (gx ^ gy) < 0                                  # opposite signs: the anti-diagonal pair
    keep_h = (m > nb(-1, 0)) & (m >= nb(1, 0))
    keep_v = (m > nb(0, -1)) & (m >= nb(0, 1))
    keep_d = np.where(s_neg, (m > nb(1, -1)) & (m > nb(-1, 1)), (m > nb(-1, -1)) & (m > nb(1, 1)))
    surv = (m > low) & ((horiz & keep_h) | (vert & keep_v) | (diag & keep_d))
    strong = surv & (m > high)
    lab, n = ndimage.label(surv, structure=np.ones((3, 3), int))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    return (good[lab] * 255).astype(np.uint8)


def _frame_images(orc, img):
    pyr = orc.create_pyramid(img)
    sob = [orc.sobel5(np.ascontiguousarray(pyr[L])) for L in range(3)]
    return pyr, sob


def test_oracle_canny_matches_independent_restatement(orc):
    rng = np.random.default_rng(5)
    d = synth.config2_pair(10, seed=77)
    _, sob = _frame_images(orc, d["ref"])
    n_edge = 0
    for L in range(3):
        for thr in (5, 20, 60):
            got = orc.canny_l2(sob[L][0], sob[L][1], 31 * thr, 70 * thr)
            want = _canny_numpy(sob[L][0], sob[L][1], 31 * thr, 70 * thr)
            assert (got == want).all(), (L, thr)
            n_edge += int((got > 0).sum())
    assert n_edge > 5000
    # arbitrary int16 gradients (ties, all sectors, both diagonal signs), swapped and clamped thresholds
    gx = rng.integers(-3000, 3001, (90, 140)).astype(np.int16)
    gy = rng.integers(-3000, 3001, (90, 140)).astype(np.int16)
    gx[::7], gy[:, ::5] = 0, 0
    gx[10:20, 10:60] = 1200; gy[10:20, 10:60] = 1200      # plateaus: strict / non-strict comparisons decide
    gx[30:40, 10:60] = -900; gy[30:40, 10:60] = 900
    for lo, hi in ((800, 2500), (2500, 800), (1.5, 40000.0), (0, 0)):
        assert (orc.canny_l2(gx, gy, lo, hi) == _canny_numpy(gx, gy, lo, hi)).all(), (lo, hi)


def test_oracle_canny_hand_cases(orc):
    # a vertical step edge: gx large on two columns, the larger one survives; weak tail joins a strong head
    h, w = 20, 30
    gx = np.zeros((h, w), np.int16); gy = np.zeros((h, w), np.int16)
    gx[:, 10] = 500; gx[:, 11] = 900; gx[:, 12] = 400
    gx[12:, 11] = 300                                       # lower part only weak
    gx[12:, 10] = 100; gx[12:, 12] = 100
    e = orc.canny_l2(gx, gy, 200, 800)
    assert (e[:, 11] == 255).all() and e.sum() == 255 * h   # the weak tail hangs on the strong head
    e = orc.canny_l2(gx[12:], gy[12:], 200, 800)
    assert e.sum() == 0                                     # the tail alone has no strong seed
    # horizontal sector tie rule: m > left && m >= right keeps the left pixel of an equal pair
    gx[:] = 0
    gx[:, 10] = 900; gx[:, 11] = 900
    e = orc.canny_l2(gx, gy, 200, 800)
    assert (e[:, 10] == 255).all() and (e[:, 11] == 0).all()


def test_oracle_grid_and_cell_quirks(orc):
    # FeatureExtractor ctor :393-400 on 640x480, 3 levels: 80 x 60 cells of 8, 4, 2 pixels
    assert [orc.detect_grid(640, 480, L) for L in range(3)] == [(8, 80, 60), (4, 80, 60), (2, 80, 60)]
    assert orc.detect_grid(752, 480, 1) == (4, 94, 60)
    # getCellIndex divides x by the number of grid ROWS (feature_detection.h:298)
    assert orc.cell_index(639, 479, 8, 80, 60) == 59 * 80 + 10
    assert orc.cell_index(100, 17, 8, 80, 60) == 2 * 80 + 1
    # the cell origin uses index / gridRows for y (:770): index 61 -> x = 61 % 80 * 8, y = 61 / 60 * 8
    h, w = 480, 640
    gx = np.zeros((h, w), np.int16); gy = np.zeros((h, w), np.int16)
    gx[10, 490] = 2000                                      # one isolated strong edge pixel
    have = np.zeros(4800, np.uint8)
    out = orc.edgelet_level(gx, gy, 0, w, h, 20, have)
    assert len(out) == 1 and (out[0]["x"], out[0]["y"], out[0]["gx"], out[0]["grad"]) == (490, 10, 2000, 2000.0)
    assert have[61] == 1 and have.sum() == 1
    # an occupied index is skipped
    have[:] = 0; have[61] = 1
    assert len(orc.edgelet_level(gx, gy, 0, w, h, 20, have)) == 0
    # the strongest edge pixel of a window wins, the first one on ties (strict >)
    gx[:] = 0
    gx[9, 489] = 2500; gx[12, 493] = 2600; gx[14, 495] = 2600
    have[:] = 0
    out = orc.edgelet_level(gx, gy, 0, w, h, 20, have)
    assert [(o["x"], o["y"]) for o in out] == [(493, 12)]


def _check(got, want, kind):
    assert len(got) == len(want), kind
    assert got.tobytes() == want.tobytes(), kind


@pytest.mark.gpu
@pytest.mark.parametrize("spec", [synth.ICL_NUIM, synth.EUROC], ids=["640x480", "752x480"])
def test_detect_candidates_bit_exact(gpu_ctx, orc, spec):
    d = synth.config2_pair(10, spec=spec, seed=52)
    fid = 9700
    gpu_ctx.frame_upload(fid, d["ref"])
    try:
        pyr, sob = _frame_images(orc, d["ref"])
        h, w = d["ref"].shape
        for thr in (8, 20):
            co, cc, eo, ec = gpu_ctx.detect_candidates([fid], n_levels=3, min_thresh=thr, corner_cap=30000, edgelet_cap=8000)
            n_e = 0
            for L in range(3):
                corners, edgelets, _ = orc.detect_candidates_level(np.ascontiguousarray(pyr[L]), sob[L][0], sob[L][1], L, w, h, thr)
                assert cc[0, L] == len(corners) and ec[0, L] == len(edgelets)
                _check(co[0, L, :cc[0, L]], corners, "corners")
                _check(eo[0, L, :ec[0, L]], edgelets, "edgelets")
                n_e += len(edgelets)
            assert n_e > 300
        # caps smaller than the counts: the first entries, full counts
        co, cc, eo, ec = gpu_ctx.detect_candidates([fid], n_levels=1, min_thresh=8, corner_cap=50, edgelet_cap=40)
        corners, edgelets, _ = orc.detect_candidates_level(np.ascontiguousarray(pyr[0]), sob[0][0], sob[0][1], 0, w, h, 8)
        assert cc[0, 0] == len(corners) > 50 and ec[0, 0] == len(edgelets) > 40
        _check(co[0, 0], corners[:50], "corners cap"); _check(eo[0, 0], edgelets[:40], "edgelets cap")
    finally:
        gpu_ctx.frame_release(fid)


@pytest.mark.gpu
def test_detect_candidates_long_weak_chains_and_batch(gpu_ctx, orc):
    """Edge closure across many tiles (at min_thresh 3 the outer ring is a 1444-pixel weak chain
    whose only strong seed is a 20-pixel stretch: the closure needs more than the first four
    passes), flat images, and a batch whose frames equal their solo results."""
    h, w = 480, 640
    yy, xx = np.mgrid[0:h, 0:w]
    spiral = np.full((h, w), 100.0)
    # concentric rectangular steps of 3 grey levels: Sobel-5 magnitude 144, weak for 93 < m < 210
    for k, r in enumerate(range(30, 230, 25)):
        ring = (np.maximum(np.abs(xx - 320), np.abs(yy - 240) * 1.3) < r)
        spiral += np.where(ring, 3.0, 0.0)
    spiral[230:250, 116:140] += 2             # a step of 5 on 20 rows of the outer ring's left side: strong
    rng = np.random.default_rng(2)
    imgs = [np.clip(spiral, 0, 255).astype(np.uint8),
            np.full((h, w), 77, np.uint8),
            synth.config2_pair(10, seed=61)["ref"],
            (np.kron(rng.integers(0, 2, (h // 8, w // 8)), np.ones((8, 8))) * 40 + 60).astype(np.uint8)]
    ids = [9710 + k for k in range(len(imgs))]
    for i, im in zip(ids, imgs):
        gpu_ctx.frame_upload(i, im)
    try:
        for thr in (1, 3):
            co, cc, eo, ec = gpu_ctx.detect_candidates(ids, n_levels=3, min_thresh=thr, corner_cap=40000, edgelet_cap=4800)
            for k, im in enumerate(imgs):
                pyr, sob = _frame_images(orc, im)
                for L in range(3):
                    corners, edgelets, _ = orc.detect_candidates_level(np.ascontiguousarray(pyr[L]), sob[L][0], sob[L][1], L, w, h, thr)
                    assert (cc[k, L], ec[k, L]) == (len(corners), len(edgelets)), (thr, k, L)
                    _check(co[k, L, :cc[k, L]], corners, "corners")
                    _check(eo[k, L, :ec[k, L]], edgelets, "edgelets")
            assert ec[1].sum() == 0 and ec[0].sum() > 300
            solo = gpu_ctx.detect_candidates([ids[2]], n_levels=3, min_thresh=thr, corner_cap=40000, edgelet_cap=4800)
            assert (solo[1][0] == cc[2]).all() and (solo[3][0] == ec[2]).all()
            assert solo[2][0].tobytes() == eo[2].tobytes()
    finally:
        for i in ids:
            gpu_ctx.frame_release(i)


@pytest.mark.gpu
def test_detect_candidates_multi_one_barrier_per_frame(gpu_ctx, orc):
    """hso_gpu_detect_candidates_multi: the keyframes of different sequences carry different minThresh_ (Frame::gradMean_,
    src/feature_detection.cpp:408-445).  Four frames at barriers 12, 5, 27, 5 in one call: every frame's corner and edgelet lists
    equal the oracle's at its own barrier and the single-barrier call's bytes; a barrier outside [0, 255] is refused."""
    h, w = 480, 640
    rng = np.random.default_rng(5)
    imgs = [synth.config2_pair(10, seed=71)["ref"], synth.config2_pair(10, seed=72)["ref"],
            (np.kron(rng.integers(0, 2, (h // 8, w // 8)), np.ones((8, 8))) * 40 + 60).astype(np.uint8), synth.config2_pair(10, seed=73)["ref"]]
    ths = [12, 5, 27, 5]
    ids = [9760 + k for k in range(len(imgs))]
    for i, im in zip(ids, imgs):
        gpu_ctx.frame_upload(i, im)
    try:
        co, cc, eo, ec = gpu_ctx.detect_candidates_multi(ids, ths, n_levels=3, corner_cap=40000, edgelet_cap=4800)
        for k, (im, thr) in enumerate(zip(imgs, ths)):
            pyr, sob = _frame_images(orc, im)
            for L in range(3):
                corners, edgelets, _ = orc.detect_candidates_level(np.ascontiguousarray(pyr[L]), sob[L][0], sob[L][1], L, w, h, thr)
                assert (cc[k, L], ec[k, L]) == (len(corners), len(edgelets)), (thr, k, L)
                _check(co[k, L, :cc[k, L]], corners, "corners")
                _check(eo[k, L, :ec[k, L]], edgelets, "edgelets")
            solo = gpu_ctx.detect_candidates([ids[k]], n_levels=3, min_thresh=thr, corner_cap=40000, edgelet_cap=4800)
            assert (solo[1][0] == cc[k]).all() and (solo[3][0] == ec[k]).all()
            assert solo[0][0].tobytes() == co[k].tobytes() and solo[2][0].tobytes() == eo[k].tobytes()
        assert cc[1].sum() != cc[3].sum() or ec[1].sum() != ec[3].sum()          # different images at the same barrier
        with pytest.raises(Exception):
            gpu_ctx.detect_candidates_multi(ids, [12, 5, 256, 5])
    finally:
        for i in ids:
            gpu_ctx.frame_release(i)


@pytest.mark.gpu
def test_detect_candidates_serpentine_chain_needs_many_closure_passes(gpu_ctx, orc):
    """One weak contour that winds through far more than 64 tile borders (a ribbon of +3 grey levels snaking over the
    whole frame) with a single strong stretch: cv::Canny's flood fill follows it to the end, so must the tiled closure,
    however many cross-tile passes that takes (there is no iteration limit)."""
    h, w = 480, 640
    img = np.full((h, w), 100.0)
    runs = list(range(24, h - 40, 36))
    for k, y0 in enumerate(runs):
        img[y0:y0 + 12, 24:w - 24] += 3.0                                   # horizontal run of the ribbon
        if k + 1 < len(runs):                                               # connector on alternating sides
            x0 = w - 36 if k % 2 == 0 else 24
            img[y0:runs[k + 1] + 12, x0:x0 + 12] = 103.0
    img[runs[0]:runs[0] + 12, 24:44] += 2.0                                 # the only strong stretch, at one end of the chain
    img = np.clip(img, 0, 255).astype(np.uint8)
    gpu_ctx.frame_upload(9740, img)
    try:
        co, cc, eo, ec = gpu_ctx.detect_candidates([9740], n_levels=3, min_thresh=3, corner_cap=40000, edgelet_cap=4800)
        pyr, sob = _frame_images(orc, img)
        for L in range(3):
            corners, edgelets, _ = orc.detect_candidates_level(np.ascontiguousarray(pyr[L]), sob[L][0], sob[L][1], L, w, h, 3)
            assert (cc[0, L], ec[0, L]) == (len(corners), len(edgelets)), L
            _check(eo[0, L, :ec[0, L]], edgelets, "edgelets")
        # the far end of the ribbon (last run) carries edgelets only if the closure walked the whole chain
        far = eo[0, 0, :ec[0, 0]]
        assert (far["y"] > runs[-1] - 4).sum() > 10
    finally:
        gpu_ctx.frame_release(9740)
