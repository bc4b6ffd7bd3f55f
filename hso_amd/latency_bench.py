"""Single-sequence latency at the metric's shape (EuRoC 752x480, radtan): BASELINE configs[2]/[3] are ONE sequence on ONE
MI355X, so what matters there is the time of one call, not the throughput of a batch.

    python -m hso_amd.latency_bench [--reps 30] [--frames 24] [--feats 200,2000]

Prints one JSON object:
  track      per feature count: wall time of hso_gpu_coarse_track_batch with ONE job (levels 4..1, forward mode; includes the
             feature-table upload, the launch, the result read-back and the stream synchronise) and the time of the tracker
             launch alone between two HIP events on the launch stream (prepare / launch / collect form);
  sequence   per max_fts: ms per frame of one synthetic sequence through libhso_host.so (FrameHandlerMono::addImage:
             frame build, tracker, reprojection + matching + selection, pose optimisation, seed updates; keyframes with
             detection, activation and local BA), non-keyframes and keyframes apart.
Development / measurement tool; bench.py embeds the same measurement in its JSON line.  Uses no CPU reference.
"""
import argparse
import json
import time

import numpy as np

from hso_amd import capi, synth


def track_latency(ctx, stream, cam, spec, n_feats, reps, seed=1234):
    import torch
    d = synth.config2_pair(n_feats, spec=spec, seed=seed)
    ctx.frame_upload(900001, d["ref"]); ctx.frame_upload(900002, d["cur"])
    p = capi.TrackParams(0, 4, 1, 50)
    job = [ctx.make_job(900001, 900002, d["feats"], capi.SE3.identity(), 1.05)]
    for _ in range(3):
        res = ctx.coarse_track_batch(cam, p, job)
    t0 = time.perf_counter()
    for _ in range(reps):
        res = ctx.coarse_track_batch(cam, p, job)
    wall = (time.perf_counter() - t0) / reps
    # the launch alone, on the context's stream
    ctx.coarse_track_prepare(cam, p, job)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(stream)
        ctx.coarse_track_launch()
        b.record(stream)
        ctx.coarse_track_collect()
    kern = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    r = res[0]
    ctx.frame_release(900001); ctx.frame_release(900002)
    return dict(feats=n_feats, call_ms=wall * 1e3, launch_ms=kern, evaluations=int(sum(r.n_eval[:])), iters=list(r.iters)[:5],
                workgroups_per_job=int(r.coop_workgroups) or 1, same_xcd=int(r.coop_same_xcd),
                trans_err=float(np.linalg.norm(r.T_cur_ref.to_arrays()[1] - d["t_true"])))


def sequence_latency(cam, S, max_fts):
    from hso_amd import vo
    odo = vo.VisualOdometry(cam, max_fts)
    odo.set_first_frame(S["images"][0], S["depth0"], 0.0)
    t_kf, t_nkf = [], []
    for k in range(1, len(S["images"])):
        t0 = time.perf_counter()
        st = odo.add_image(S["images"][k], float(k))
        dt = time.perf_counter() - t0
        (t_kf if st.is_keyframe else t_nkf).append(dt)
    st = odo.status()
    q, t = st.T_f_w.to_arrays()
    err = float(np.linalg.norm(t - S["T_f_w"][len(S["images"]) - 1][1]))
    odo.close()
    allf = t_kf + t_nkf
    return dict(max_fts=max_fts, frames=len(allf), keyframes=len(t_kf), ms_per_frame=1e3 * float(np.mean(allf)),
                ms_per_non_keyframe=1e3 * float(np.mean(t_nkf)) if t_nkf else None,
                ms_per_keyframe=1e3 * float(np.mean(t_kf)) if t_kf else None, frames_per_s=len(allf) / sum(allf),
                n_matches_last=int(st.n_matches), trans_err_last=err)


def multi_sequence_run(cam, seqs, max_fts, device=0):
    """S sequences in lockstep through hso_vo_multi_* (one context; the device calls of all sequences batched per kind).
    -> (dict for the bench line, per-sequence trajectories [(timestamp, (q, t))])."""
    from hso_amd import vo
    S = len(seqs)
    n_frames = min(len(q["images"]) for q in seqs)
    m = vo.MultiVisualOdometry(cam, S, max_fts, device)
    m.set_first_frames([q["images"][0] for q in seqs], [q["depth0"] for q in seqs])
    traj = [[(0.0, m.status(q).T_f_w.to_arrays())] for q in range(S)]
    t_steps = []
    for k in range(1, n_frames):
        t0 = time.perf_counter()
        m.add_images([q["images"][k] for q in seqs], [float(k)] * S)
        t_steps.append(time.perf_counter() - t0)
        for q in range(S):
            traj[q].append((float(k), m.status(q).T_f_w.to_arrays()))
    err = [float(np.linalg.norm(traj[q][-1][1][1] - seqs[q]["T_f_w"][n_frames - 1][1])) for q in range(S)]
    n_kf = [len(m.keyframes(q)) for q in range(S)]
    counts = m.call_counts()
    m.close()
    total = sum(t_steps)
    return dict(sequences=S, frames_per_sequence=n_frames - 1, max_fts=max_fts, frames_per_s=S * (n_frames - 1) / total,
                ms_per_step=1e3 * total / (n_frames - 1), keyframes=n_kf, trans_err_last_max=max(err),
                batched_calls={k: v[0] for k, v in counts.items()}, requests={k: v[1] for k, v in counts.items()}), traj


def measure(reps=30, frames=24, feats=(200, 2000), spec=None, device=0):
    import torch
    spec = spec or synth.EUROC
    cam = synth.camera(spec)
    stream = torch.cuda.Stream(device)
    ctx = capi.Context(device, stream.cuda_stream)
    out = {"shape": "%dx%d" % (spec["width"], spec["height"]), "track": [], "sequence": []}
    for n in feats:
        out["track"].append(track_latency(ctx, stream, cam, spec, n, reps))
    ctx.close()
    if frames > 1:
        S = synth.sequence(frames, spec=spec)
        for n in feats:
            out["sequence"].append(sequence_latency(cam, S, n))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--feats", default="200,2000")
    args = ap.parse_args()
    print(json.dumps(measure(args.reps, args.frames, tuple(int(x) for x in args.feats.split(",")))))


if __name__ == "__main__":
    main()
