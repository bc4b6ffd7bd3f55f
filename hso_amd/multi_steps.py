"""`python -m hso_amd.multi_steps <n_sequences>`: per-step wall time and batched-call counts of the multi-sequence driver (hso_vo_multi_*);
`HSO_MULTI_TIMING=1` adds the wall time of every batched call.  Measurement helper (profiles/r3_host_memory_eviction.txt)."""
import sys, time, numpy as np
from hso_amd import synth, vo
S = int(sys.argv[1]); frames = 24
spec = synth.EUROC; cam = synth.camera(spec)
seqs = synth.sequences(S, frames, spec=spec, seed0=2024)
m = vo.MultiVisualOdometry(cam, S, 200)
t0 = time.perf_counter(); m.set_first_frames([q["images"][0] for q in seqs], [q["depth0"] for q in seqs]); print("first frames %.1f ms" % ((time.perf_counter() - t0) * 1e3))
def cg():
    d = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat'))
    return int(d['usage_usec']), int(d['nr_throttled']), int(d['throttled_usec'])
c0 = cg(); tw0 = time.perf_counter()
prev = m.call_counts()
for k in range(1, frames):
    t0 = time.perf_counter()
    m.add_images([q["images"][k] for q in seqs], [float(k)] * S)
    dt = (time.perf_counter() - t0) * 1e3
    c = m.call_counts()
    d = {kk: (c[kk][0] - prev[kk][0], c[kk][1] - prev[kk][1]) for kk in c if c[kk] != prev[kk]}
    prev = c
    nkf = sum(m.status(q).is_keyframe for q in range(S))
    print("step %2d %7.2f ms kf %2d %s" % (k, dt, nkf, d))
c1 = cg(); print('stepping phase: wall %.3f s, cpu %.3f s, throttled periods %d, throttled %.3f s' % (time.perf_counter() - tw0, (c1[0] - c0[0]) / 1e6, c1[1] - c0[1], (c1[2] - c0[2]) / 1e6))
m.close()
