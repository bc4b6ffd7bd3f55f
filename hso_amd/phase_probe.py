"""Developer probe: per-phase shader cycles of the tracker kernel (library must be built
with -DHSO_PHASE_TIMERS: `HSO_EXTRA_FLAGS=-DHSO_PHASE_TIMERS python -m hso_amd.build --force`)."""
import sys
import numpy as np
from hso_amd import capi, synth


def main(n_feats=2000, batch=1, inverse=0):
    d = synth.config2_pair(n_feats)
    cam = synth.camera()
    ctx = capi.Context(0)
    ids = []
    for i in range(batch):
        ctx.frame_upload(2 * i, d["ref"]); ctx.frame_upload(2 * i + 1, d["cur"]); ids.append((2 * i, 2 * i + 1))
    p = capi.TrackParams(inverse, 4, 1, 50)
    jobs = [ctx.make_job(a, b, d["feats"], capi.SE3.identity(), 1.05) for a, b in ids]
    for rep in range(3):
        res = ctx.coarse_track_batch(cam, p, jobs)
    r = res[0]
    ph = np.array(r.phase_cycles[:], float)
    n_eval = sum(r.n_eval[:])
    names = ["stage+precompute", "thresholds", "evaluations", "lm_solve", "job total"]
    for k in range(5):
        print("%-18s %10.0f cycles  %5.1f%%" % (names[k], ph[k], 100 * ph[k] / max(ph[4], 1)))
    import os
    labels = ["  project", "  pixel loop", "  expansion", "  wave exchange", "  wg combine"]
    if os.environ.get("HSO_PROBE_BASE3"):   # library built with -DHSO_PHASE_TIMERS_BASE=3
        labels = ["  wave exchange", "  wg combine (all)", "   coop: sync+store", "   coop: poll", "   coop: sum"]
    for k, nm in enumerate(labels):
        print("%-18s %10.0f cycles  %5.1f%%" % (nm, ph[5 + k], 100 * ph[5 + k] / max(ph[4], 1)))
    print("evals", n_eval, "cycles/eval", ph[2] / max(n_eval, 1), "iters", list(r.iters), "coop K", r.coop_workgroups, "same xcd", r.coop_same_xcd)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
