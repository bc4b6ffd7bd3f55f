"""Generate tests/golden/fast12.json from the COMPILED REFERENCE FAST library
(oracle/_ref/libfast_ref.so: fast_12_detect.cpp, fast_12_score.cpp, nonmax_3x3.cpp of
/root/reference/thirdparty/fast/src, built by oracle/Makefile) — the calls
FeatureExtractor::fillingHole makes (src/feature_detection.cpp:1125-1154).  Run in the authoring
container only: `python tests/golden/make_fast12_golden.py`.  The images are those of fast9.json
(referenced by name, not repeated); the JSON holds the library's outputs only.
"""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402


def main():
    oracle_py.build()
    here = os.path.dirname(os.path.abspath(__file__))
    g9 = json.load(open(os.path.join(here, "fast9.json")))
    cases = []
    for name, v in g9["images"].items():
        img = np.frombuffer(base64.b64decode(v["data"]), np.uint8).reshape(v["h"], v["w"]).copy()
        for thr in (6, 12, 30):                 # fillingHole's barrier is max(0.6 * minThresh, 6): 6..12 for minThresh 7..20
            r = oracle_py.ref_fast12(img, thr)
            if r is None:
                raise SystemExit("oracle/_ref/libfast_ref.so not built (reference absent)")
            xy, sc, keep = r
            cases.append({"image": name, "threshold": thr, "xy": xy.reshape(-1).tolist(), "scores": sc.tolist(),
                          "nonmax": keep.tolist()})
    with open(os.path.join(here, "fast12.json"), "w") as f:
        json.dump({"images_from": "fast9.json", "cases": cases}, f, separators=(",", ":"))
    print("wrote fast12.json:", len(cases), "cases,", sum(len(c["scores"]) for c in cases), "corners")


if __name__ == "__main__":
    main()
