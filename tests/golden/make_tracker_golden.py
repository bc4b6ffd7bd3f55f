"""Generate tests/golden/tracker_small.json: a small seeded CoarseTracker case and
the CPU restatement's outputs on it.

The reference holds no golden vectors for this path and cannot be built here
(SURVEY.md §8c), so this fixture is a SELF-golden: it pins the oracle against
accidental change and lets `-m gpu` tests check the HIP path against a committed
vector.  It does NOT pin the oracle against the reference ("parity unpinned").
Inputs are regenerated from seeds by hso_amd/synth.py, so only outputs are stored.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hso_amd import synth, capi  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402

SPEC = dict(model=capi.CAM_PINHOLE, width=320, height=240, fx=240.6, fy=240.0, cx=159.5, cy=119.5)


def case():
    return synth.config2_pair(150, spec=SPEC, seed=4242, exposure=0.97, noise=0.5, rot_deg=0.3)


def main():
    d = case()
    cam = synth.camera(SPEC)
    rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
    st_r = orc.frame_stats(rp[0], *orc.sobel5(rp[0]))
    st_c = orc.frame_stats(cp[0], *orc.sobel5(cp[0]))
    out = {"image_sha": [int(d["ref"].astype(np.uint64).sum()), int(d["cur"].astype(np.uint64).sum())],
           "stats": [st_r.integral_image, st_r.grad_mean, st_c.integral_image, st_c.grad_mean], "runs": {}}
    a0 = np.float32(st_c.integral_image / st_r.integral_image)
    for inv in (0, 1):
        p = capi.TrackParams(inv, 4, 1, 50)
        tr = orc.Tracker(cam, p, rp, cp, d["feats"])
        r = tr.run(capi.SE3.identity(), float(a0))
        out["runs"][str(inv)] = {
            "q": list(r.T_cur_ref.q), "t": list(r.T_cur_ref.t), "a": r.exposure_rat, "n_tracked": r.n_tracked,
            "iters": list(r.iters), "accept": [int(x) for x in r.accept_mask],
            "huber": [float(x) for x in r.huber], "outlier": [float(x) for x in r.outlier],
            "n_select": list(r.n_select), "energy": list(r.energy)}
    with open(os.path.join(os.path.dirname(__file__), "tracker_small.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tracker_small.json")


if __name__ == "__main__":
    main()
