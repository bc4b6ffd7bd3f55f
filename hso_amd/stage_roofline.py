"""One later-pipeline stage at its multi-sequence size, with the algorithmic bytes SURVEY.md section 8(d) assigns to it, for
the per-kernel roofline table (profiles/collect_stages.sh runs every stage under `rocprofv3 --kernel-trace --stats`;
profiles/summarize_stages.py divides the bytes printed here by the kernel durations of the same run).

    python -m hso_amd.stage_roofline --stage align|seed|pose|frame [--reps 5]

Algorithmic bytes per unit ("Other stages' units"):
  align  100 taps x 4 B for the warped reference patch + iterations x 64 x 4 B of current-image samples, per candidate that
         reaches the LK stage (iterations as reported by the kernel);
  seed   epipolar steps x 64 x 4 B per seed (steps as reported by the kernel);
  pose   (iterations + 1) x n_features x sizeof(hso_pose_feat) per frame (every LM iteration re-reads the feature table);
  frame  1.33 WH read + 0.33 WH written (pyramid) + 1.31 WH read + 5.25 WH written (Sobel of levels 0-2) per frame.
Development / measurement tool; uses no CPU reference.
"""
import argparse
import ctypes as C
import json
import math
import time

import numpy as np

from hso_amd import capi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", required=True, choices=["align", "seed", "pose", "frame"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nseq", type=int, default=128)
    args = ap.parse_args()
    ctx = capi.Context(0)
    cam = synth.camera()
    pair = synth.config2_pair(2000, trans_frac=0.03)
    nseq = args.nseq
    rec = dict(stage=args.stage, sequences=nseq, reps=args.reps)
    if args.stage in ("align", "seed"):
        for k in range(2 * nseq):
            ctx.frame_upload(1000 + k, pair["ref"] if k % 2 == 0 else pair["cur"])
        gy0, gx0 = np.gradient(pair["ref"].astype(np.float64))
    if args.stage == "align":
        jobs = synth.align_jobs(pair, 2000, 1000, gx=gx0, gy=gy0)
        big = (capi.AlignJob * (nseq * len(jobs)))()
        cur_ids = []
        for q in range(nseq):
            for i, j in enumerate(jobs):
                C.memmove(C.byref(big[q * len(jobs) + i]), C.byref(j), C.sizeof(capi.AlignJob))
                big[q * len(jobs) + i].ref_frame_id = 1000 + 2 * q
            cur_ids += [1001 + 2 * q] * len(jobs)
        out = ctx.align_multi(cam, cur_ids, big, as_list=False)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = ctx.align_multi(cam, cur_ids, big, as_list=False)
        dt = (time.perf_counter() - t0) / args.reps
        iters = np.array([o.iters for o in out]); stage = np.array([o.stage for o in out])
        warped = int((iters > 0).sum())
        rec.update(kernel="k_align", units="candidates", n=len(cur_ids), ms_per_call=dt * 1e3,
                   algorithmic_bytes_per_call=float(warped * 400 + int(iters.sum()) * 256), mean_iters=float(iters.mean()),
                   reached_lk=warped)
    elif args.stage == "seed":
        seeds, T_cur, _ = synth.seeds_for_pair(pair, 900, 1000, gx=gx0, gy=gy0)
        pea = math.atan(1.0 / (2.0 * 480.6)) * 2.0
        big_s = (capi.Seed * (nseq * len(seeds)))()
        s_frame = np.repeat(np.arange(nseq, dtype=np.int32), len(seeds))
        for q in range(nseq):
            for i, sd in enumerate(seeds):
                C.memmove(C.byref(big_s[q * len(seeds) + i]), C.byref(sd), C.sizeof(capi.Seed))
                big_s[q * len(seeds) + i].ref_frame_id = 1000 + 2 * q
        act = [(1001 + 2 * q, T_cur, 1.05) for q in range(nseq)]
        out = ctx.seed_observe_multi(cam, act, s_frame, pea, big_s, as_list=False)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = ctx.seed_observe_multi(cam, act, s_frame, pea, big_s, as_list=False)
        dt = (time.perf_counter() - t0) / args.reps
        steps = np.array([max(o.n_steps, 0) for o in out])
        rec.update(kernel="k_seed_observe", units="seeds", n=len(s_frame), ms_per_call=dt * 1e3,
                   algorithmic_bytes_per_call=float(int(steps.sum()) * 256), mean_steps=float(steps.mean()))
    elif args.stage == "pose":
        n_frames = 2 * nseq
        feats_p, poses, T0, _ = synth.pose_problem(300, seed=5)
        pj = [capi.make_pose_job(feats_p, poses, T0) for _ in range(n_frames)]
        arr = (capi.PoseJob * n_frames)(*pj); res = (capi.PoseResult * n_frames)()
        masks = [np.zeros(max(j.n_feats, 1), np.uint8) for j in pj]
        mptr = (C.c_void_p * n_frames)(*[m.ctypes.data for m in masks])
        call = lambda: ctx._check(ctx.lib.hso_gpu_pose_optimize_batch(ctx.h, C.byref(cam), arr, n_frames, res, mptr), "pose")
        call()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            call()
        dt = (time.perf_counter() - t0) / args.reps
        its = np.array([r.iters for r in res]); trials = np.array([r.n_trials_total for r in res])
        rec.update(kernel="k_pose", units="frames", n=n_frames, ms_per_call=dt * 1e3,
                   algorithmic_bytes_per_call=float(int((trials + its + 1).sum()) * len(feats_p) * C.sizeof(capi.PoseFeat)),
                   mean_iters=float(its.mean()), mean_evaluations=float((trials + its + 1).mean()), feats_per_frame=len(feats_p))
    else:
        import torch
        n_frames = 8 * nseq
        img = torch.from_numpy(pair["cur"].copy()).cuda()
        ptrs = np.array([img.data_ptr()] * n_frames, np.uint64)
        ids = list(range(5000, 5000 + n_frames))
        H, W = pair["cur"].shape
        call = lambda: ctx.frame_upload_batch(ids, device_ptrs=ptrs, width=W, height=H, want_stats=False)
        call(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            call()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        rec.update(kernel="k_pyramid+k_sobel", units="frames", n=n_frames, ms_per_call=dt * 1e3,
                   algorithmic_bytes_per_call=float(n_frames * W * H * (1.33 + 0.33 + 1.31 + 5.25)), width=W, height=H)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
