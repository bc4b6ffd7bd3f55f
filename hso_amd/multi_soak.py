"""`python -m hso_amd.multi_soak <sequences> <frames>`: a long lockstep run of the multi-sequence driver (failures, keyframes, drift, resident set size)."""
import sys, time, numpy as np, resource
from hso_amd import synth, vo
S, frames = int(sys.argv[1]), int(sys.argv[2])
spec = synth.EUROC; cam = synth.camera(spec)
seqs = synth.sequences(S, frames, spec=spec, seed0=777)
m = vo.MultiVisualOdometry(cam, S, 200)
m.set_first_frames([q["images"][0] for q in seqs], [q["depth0"] for q in seqs])
t0 = time.perf_counter(); fails = 0; rss0 = None
for k in range(1, frames):
    m.add_images([q["images"][k] for q in seqs], [float(k)] * S)
    for q in range(S):
        st = m.status(q)
        fails += int(st.stage != 3 or st.result == 2)
    if k == 20: rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
dt = time.perf_counter() - t0
err = [float(np.linalg.norm(np.array(m.status(q).T_f_w.t) - seqs[q]["T_f_w"][frames - 1][1])) for q in range(S)]
print("multi soak: %d sequences x %d frames: %.1f frames/s, failures %d, keyframes %s, max |t - t_gt| %.4f m, maxrss %d -> %d MB" % (
    S, frames - 1, S * (frames - 1) / dt, fails, [len(m.keyframes(q)) for q in range(S)][:6], max(err), rss0 // 1024, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024))
m.close()
