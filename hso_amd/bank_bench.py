"""`python -m hso_amd.bank_bench <sequences> <frames> <max_fts> [distinct]`: the sequence engine end to end (hso_vo_multi_*: frame
construction, tracker, reprojection + matching + selection + pose, local BA, depth filter, all on one evolving state per sequence)
for many sequences in lockstep on one GPU.  `distinct` rendered sequences are replicated to `sequences` (same images = same work).
Prints one JSON line; HSO_ENGINE_TIMING=1 adds the engine's phase split on stderr."""
import json
import sys
import time

import numpy as np


def run(n_seq, frames, max_fts, distinct=8, spec=None, device=0, seqs=None):
    from hso_amd import synth, vo
    spec = spec or synth.EUROC
    cam = synth.camera(spec)
    if seqs is None:
        seqs = synth.sequences(min(distinct, n_seq), frames, spec=spec, seed0=777)
    pick = [seqs[q % len(seqs)] for q in range(n_seq)]
    m = vo.MultiVisualOdometry(cam, n_seq, max_fts, device=device)
    m.set_first_frames([q["images"][0] for q in pick], [q["depth0"] for q in pick])
    step_ms, fails = [], 0
    for k in range(1, frames):
        imgs = [q["images"][k] for q in pick]
        t0 = time.perf_counter()
        m.add_images(imgs, [float(k)] * n_seq)
        step_ms.append(1e3 * (time.perf_counter() - t0))
    kfs = [len(m.keyframes(q)) for q in range(n_seq)]
    sts = [m.status(q) for q in range(n_seq)]
    fails = sum(int(st.stage != 3 or st.result == 2) for st in sts)
    err = [float(np.linalg.norm(np.array(sts[q].T_f_w.t[:]) - pick[q]["T_f_w"][frames - 1][1])) for q in range(n_seq)]
    counts = m.call_counts()
    m.close()
    warm = step_ms[2:] if len(step_ms) > 4 else step_ms
    return dict(sequences=n_seq, distinct=len(seqs), frames=frames - 1, max_fts=max_fts, frames_per_s=1e3 * n_seq / float(np.mean(warm)),
                ms_per_step_mean=float(np.mean(warm)), ms_per_step_median=float(np.median(warm)), ms_per_step_max=float(np.max(warm)),
                ms_first_steps=[round(x, 2) for x in step_ms[:3]], keyframes_per_sequence=float(np.mean(kfs)) - 1, failures=fails,
                trans_err_max=max(err), n_matches_last=int(np.mean([st.n_matches for st in sts])), n_seeds_last=int(np.mean([st.n_seeds for st in sts])),
                calls=counts)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(a[0], a[1], a[2], a[3] if len(a) > 3 else 8)))
