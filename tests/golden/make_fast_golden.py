"""Generate tests/golden/fast9.json from the COMPILED REFERENCE FAST library
(oracle/_ref/libfast_ref.so = /root/reference/thirdparty/fast/src/{fast_9, fast_9_score, nonmax_3x3,
faster_corner_9_sse}.cpp built by oracle/Makefile).  Run in the authoring container only (the
reference does not travel): `python tests/golden/make_fast_golden.py`.  The JSON holds input images
(base64) and the library's outputs — corner list, scores, indices kept by fast_nonmax_3x3 — for
several barriers; no reference source is stored.
"""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402
from hso_amd import synth  # noqa: E402


def main():
    oracle_py.build()
    rng = np.random.default_rng(20260928)
    scene = synth.config2_pair(10)["ref"]
    images = {
        "scene_crop_160x120": scene[100:220, 200:360].copy(),
        "scene_half_crop_96x80": scene[::2, ::2][60:140, 100:196].copy(),
        "noise_97x60": rng.integers(0, 256, (60, 97), dtype=np.uint8),
        "four_levels_64x48": (rng.integers(0, 4, (48, 64)) * 80).astype(np.uint8),      # many equal scores: non-max ties
        "narrow_20x30": rng.integers(0, 256, (30, 20), dtype=np.uint8),                 # width < 22: the plain detector
        "short_22x7": rng.integers(0, 256, (7, 22), dtype=np.uint8),
        "flat_21x20": np.full((20, 21), 77, np.uint8),
    }
    cases = []
    for name, img in images.items():
        for thr in (5, 20, 40):
            r = oracle_py.ref_fast(img, thr)
            if r is None:
                raise SystemExit("oracle/_ref/libfast_ref.so not built (reference absent)")
            xy, sc, keep = r
            cases.append({"image": name, "threshold": thr, "xy": xy.reshape(-1).tolist(), "scores": sc.tolist(),
                          "nonmax": keep.tolist()})
    out = {"images": {k: {"w": int(v.shape[1]), "h": int(v.shape[0]), "data": base64.b64encode(v.tobytes()).decode()}
                      for k, v in images.items()},
           "cases": cases}
    with open(os.path.join(os.path.dirname(__file__), "fast9.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote fast9.json:", len(cases), "cases,", sum(len(c["scores"]) for c in cases), "corners")


if __name__ == "__main__":
    main()
