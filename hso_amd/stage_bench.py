"""Timing of the later pipeline stages at keyframe-sized batches:
reprojection matching, pose optimisation, seed observation, seed activation and BA linearisation.

Prints one JSON object per stage: wall time of the C-ABI call (includes the host<->device copies
of the job / result records and the stream synchronise the ABI performs) and the work done per
call.  Run it under `rocprofv3 --kernel-trace --stats` (profiles/collect_r1.sh does) to get the
kernel-only durations that go with these numbers.  Development / measurement tool: it is not part
of the product path and uses no CPU reference.

    python -m hso_amd.stage_bench [--reps 20]
"""
import argparse
import json
import math
import time

import numpy as np

from hso_amd import capi, synth


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--candidates", type=int, default=2000)
    ap.add_argument("--seeds", type=int, default=900)
    ap.add_argument("--pose-frames", type=int, default=256)
    args = ap.parse_args()
    ctx = capi.Context(0)
    cam = synth.camera()
    out = []

    # a11-a14: one frame's reprojection candidates in one call
    pair = synth.config2_pair(2000, trans_frac=0.03)
    ctx.frame_upload(1, pair["ref"]); ctx.frame_upload(2, pair["cur"])
    gy0, gx0 = np.gradient(pair["ref"].astype(np.float64))
    jobs = synth.align_jobs(pair, args.candidates, 1, gx=gx0, gy=gy0)
    jobs_arr = (capi.AlignJob * len(jobs))(*jobs)            # marshalled once: the timed call is the C-ABI call
    dt = timed(lambda: ctx.align_batch(cam, 2, jobs_arr, as_list=False), args.reps)
    ok = sum(o.success for o in ctx.align_batch(cam, 2, jobs))
    out.append(dict(stage="align_batch (findMatchDirect)", units="candidates", n=len(jobs), ms_per_call=dt * 1e3,
                    units_per_s=len(jobs) / dt, matched=ok))

    # a16-a17: one active frame against all seeds
    seeds, T_cur, feats = synth.seeds_for_pair(pair, args.seeds, 1, gx=gx0, gy=gy0)
    pea = math.atan(1.0 / (2.0 * 480.6)) * 2.0
    seeds_arr = (capi.Seed * len(seeds))(*seeds)
    dt = timed(lambda: ctx.seed_observe(cam, 2, T_cur, 1.05, pea, seeds_arr, as_list=False), args.reps)
    ok = sum(o.result == 1 for o in ctx.seed_observe(cam, 2, T_cur, 1.05, pea, seeds))
    out.append(dict(stage="seed_observe (observeDepthRow/doLineStereo)", units="seeds", n=len(seeds), ms_per_call=dt * 1e3,
                    units_per_s=len(seeds) / dt, matched=ok))
    # section 8f rank 1, first stage: FAST-9 corner candidates of pyramid levels 0..2 of one frame
    dt = timed(lambda: ctx.fast_detect(1, 3, 20, 8, 20000), args.reps)
    _, counts = ctx.fast_detect(1, 3, 20, 8, 20000)
    npx = sum((640 >> l) * (480 >> l) for l in range(3))
    out.append(dict(stage="fast_detect (FAST-9 + score + nonmax + Shi-Tomasi, levels 0-2)", units="pixels", n=npx,
                    ms_per_call=dt * 1e3, units_per_s=npx / dt, corners=sum(counts)))
    # the same for the new keyframes of 256 independent sequences in one call (counts only: no list copy)
    nb = 256
    for k in range(nb):
        ctx.frame_upload(1000 + k, pair["ref"] if k % 2 == 0 else pair["cur"])
    bid = list(range(1000, 1000 + nb))
    # reprojection candidates of 128 sequences (frame pairs 1000+2k -> 1001+2k) in one launch
    import ctypes as C
    nseq = nb // 2
    big = (capi.AlignJob * (nseq * len(jobs)))()
    cur_ids = []
    for q in range(nseq):
        for i, j in enumerate(jobs):
            C.memmove(C.byref(big[q * len(jobs) + i]), C.byref(j), C.sizeof(capi.AlignJob))
            big[q * len(jobs) + i].ref_frame_id = 1000 + 2 * q
        cur_ids += [1001 + 2 * q] * len(jobs)
    dt = timed(lambda: ctx.align_multi(cam, cur_ids, big, as_list=False), max(args.reps // 4, 2))
    out.append(dict(stage="align_multi x128 frames", units="candidates", n=len(cur_ids), ms_per_call=dt * 1e3,
                    units_per_s=len(cur_ids) / dt))
    # the seeds of 128 sequences observed in their own active frames in one launch
    big_s = (capi.Seed * (nseq * len(seeds)))()
    s_frame = np.repeat(np.arange(nseq, dtype=np.int32), len(seeds))
    for q in range(nseq):
        for i, sd in enumerate(seeds):
            C.memmove(C.byref(big_s[q * len(seeds) + i]), C.byref(sd), C.sizeof(capi.Seed))
            big_s[q * len(seeds) + i].ref_frame_id = 1000 + 2 * q
    act_frames = [(1001 + 2 * q, T_cur, 1.05) for q in range(nseq)]
    dt = timed(lambda: ctx.seed_observe_multi(cam, act_frames, s_frame, pea, big_s, as_list=False), max(args.reps // 4, 2))
    out.append(dict(stage="seed_observe_multi x128 frames", units="seeds", n=len(s_frame), ms_per_call=dt * 1e3,
                    units_per_s=len(s_frame) / dt))
    dt = timed(lambda: ctx.fast_detect_batch(bid, 3, 20, 8, 0), max(args.reps // 2, 2))
    out.append(dict(stage="fast_detect_batch x256 frames (levels 0-2, counts only)", units="pixels", n=npx * nb,
                    ms_per_call=dt * 1e3, units_per_s=npx * nb / dt))
    # section 8f rank 1, second stage: corners + edgelets (FAST, cell occupancy, Canny on the Sobel images, per-cell arg-max)
    dt = timed(lambda: ctx.detect_candidates([1], 3, 20, 8192, 4800), args.reps)
    _, cc, _, ec = ctx.detect_candidates([1], 3, 20, 8192, 4800)
    out.append(dict(stage="detect_candidates (FAST + Canny + edgelets, levels 0-2)", units="pixels", n=npx,
                    ms_per_call=dt * 1e3, units_per_s=npx / dt, corners=int(cc.sum()), edgelets=int(ec.sum())))
    dt = timed(lambda: ctx.detect_candidates(bid, 3, 20, 0, 0), max(args.reps // 2, 2))
    out.append(dict(stage="detect_candidates x256 frames (counts only)", units="pixels", n=npx * nb,
                    ms_per_call=dt * 1e3, units_per_s=npx * nb / dt))
    for k in bid:
        ctx.frame_release(k)
    ctx.frame_release(1); ctx.frame_release(2)

    # a18: activation of converged seeds against 8 observing frames
    P = synth.activation_problem(n_seeds=300, n_targets=8)
    ctx.frame_upload(P["host_frame_id"], P["host"])
    for t, f in zip(P["targets"], P["frames"]):
        ctx.frame_upload(t.frame_id, f)
    dt = timed(lambda: ctx.seed_activate(cam, P["seeds"], P["per_seed"], 6), args.reps)
    act = sum(o.activated for o in ctx.seed_activate(cam, P["seeds"], P["per_seed"], 6))
    n_pairs = sum(len(t) for t in P["per_seed"])
    out.append(dict(stage="seed_activate (activatePoint/seedOptimizer)", units="seed-target pairs", n=n_pairs,
                    ms_per_call=dt * 1e3, units_per_s=n_pairs / dt, activated=act))

    # a15: a batch of frames, 300 features each
    feats_p, poses, T0, _ = synth.pose_problem(300, seed=5)
    pj = [capi.make_pose_job(feats_p, poses, T0) for _ in range(args.pose_frames)]
    dt = timed(lambda: ctx.pose_optimize_batch(cam, pj), max(args.reps // 4, 2))
    out.append(dict(stage="pose_optimize_batch (LM3rd)", units="frames", n=len(pj), ms_per_call=dt * 1e3,
                    units_per_s=len(pj) / dt))

    # a19: one local BA window
    bposes, fixed, idist, edges = synth.ba_problem(9, 3000, 5, seed=9)
    dt = timed(lambda: ctx.ba_linearize(bposes, fixed, idist, edges, 1.0, 0.7), args.reps)
    out.append(dict(stage="ba_linearize", units="edges", n=len(edges), ms_per_call=dt * 1e3, units_per_s=len(edges) / dt))

    # a19 complete: the Levenberg optimisation of one window, and of the windows of 32 sequences in one lockstep call
    prob = (bposes, fixed, idist, edges, 1.0, 0.7, 10)
    dt1 = timed(lambda: ctx.ba_optimize(*prob), max(args.reps // 4, 2))
    r1 = ctx.ba_optimize(*prob)[3]
    out.append(dict(stage="ba_optimize (LM; linearisation, Schur complement, dense LDL^T, updates and errors on the device), one window", units="edges", n=len(edges),
                    ms_per_call=dt1 * 1e3, units_per_s=len(edges) / dt1, iterations=r1.iterations, solves=r1.n_solves))
    probs = [prob] * 32
    dtm = timed(lambda: ctx.ba_optimize_multi(probs), 2)
    out.append(dict(stage="ba_optimize_multi x32 windows (lockstep)", units="edges", n=32 * len(edges), ms_per_call=dtm * 1e3,
                    units_per_s=32 * len(edges) / dtm, ms_per_window=dtm * 1e3 / 32))

    # section 8f rank 2: project the map points of 6 keyframes, choose reference observations, match — one call
    M = synth.map_problem(n_points=3000, first_frame_id=7000)
    for k, f in zip(M["kfs"], M["frames"]):
        ctx.frame_upload(int(k["frame_id"]), f)
    ctx.frame_upload(M["cur_frame_id"], M["cur"])
    rp_args = (cam, M["cur_frame_id"], M["T_cur_w"], M["cur_exposure"], M["cur_keyframe_id"], M["kfs"], M["points"], M["obs"],
               M["cell_size"], M["grid_n_cols"])
    dt = timed(lambda: ctx.reproject_match(*rp_args), args.reps)
    proj, match = ctx.reproject_match(*rp_args)
    out.append(dict(stage="reproject_match (reprojectPoint + getCloseViewObs + findMatchDirect)", units="map points", n=len(proj),
                    ms_per_call=dt * 1e3, units_per_s=len(proj) / dt, projected=int(proj["projected"].sum()),
                    matched=sum(m.success for m in match)))

    # the same for 64 sequences in one launch (the tables replicated; the resident frames are shared)
    nrep = 64
    q_cur, t_cur = M["T_cur_w"].to_arrays()
    frames = np.zeros(nrep, capi.REPROJ_FRAME_DTYPE)
    pts_all, obs_all = [], []
    for r in range(nrep):
        frames[r]["cur_frame_id"], frames[r]["q"], frames[r]["t"] = M["cur_frame_id"], q_cur, t_cur
        frames[r]["cur_exposure_time"], frames[r]["cur_keyframe_id"] = M["cur_exposure"], M["cur_keyframe_id"]
        frames[r]["kf_begin"], frames[r]["kf_count"] = r * len(M["kfs"]), len(M["kfs"])
        frames[r]["point_begin"], frames[r]["point_count"] = r * len(M["points"]), len(M["points"])
        pts = M["points"].copy(); pts["obs_begin"] += r * len(M["obs"])
        pts_all.append(pts); obs_all.append(M["obs"])
    kfs_all, pts_all, obs_all = np.concatenate([M["kfs"]] * nrep), np.concatenate(pts_all), np.concatenate(obs_all)
    dt = timed(lambda: ctx.reproject_match_multi(cam, frames, kfs_all, pts_all, obs_all, M["cell_size"], M["grid_n_cols"]),
               max(args.reps // 4, 2))
    out.append(dict(stage="reproject_match_multi x64 frames", units="map points", n=len(pts_all), ms_per_call=dt * 1e3,
                    units_per_s=len(pts_all) / dt))

    # ---- the per-frame chain for 256 sequences, every stage in its multi-sequence form (inputs are
    # independent synthetic problems of realistic size sharing resident frames, not one chained
    # state: a sum of stage times, not an end-to-end VO run)
    nseq = 256
    pairs = [synth.config2_pair(2000, seed=1234 + 7 * k) for k in range(4)]
    ref_ids, cur_ids2 = list(range(20000, 20000 + nseq)), list(range(21000, 21000 + nseq))
    st_r = ctx.frame_upload_batch(ref_ids, imgs=[pairs[i % 4]["ref"] for i in range(nseq)])
    st_c = ctx.frame_upload_batch(cur_ids2, imgs=[pairs[i % 4]["cur"] for i in range(nseq)])
    tjobs = [ctx.make_job(ref_ids[i], cur_ids2[i], pairs[i % 4]["feats"], capi.SE3.identity(),
                          float(np.float32(st_c[i].integral_image / st_r[i].integral_image))) for i in range(nseq)]
    ctx.coarse_track_prepare(cam, capi.TrackParams(0, 4, 1, 50), tjobs)

    def track():
        ctx.frame_upload_batch(cur_ids2, imgs=[pairs[i % 4]["cur"] for i in range(nseq)], want_stats=False)
        ctx.coarse_track_launch()
        return ctx.coarse_track_collect()
    t_track = timed(track, 3)
    M1 = synth.map_problem(n_points=1000, first_frame_id=7000)      # frames 7000.. are still resident
    fr = np.zeros(nseq, capi.REPROJ_FRAME_DTYPE)
    pa, oa = [], []
    for r in range(nseq):
        fr[r]["cur_frame_id"], fr[r]["q"], fr[r]["t"] = M1["cur_frame_id"], q_cur, t_cur
        fr[r]["cur_exposure_time"], fr[r]["cur_keyframe_id"] = M1["cur_exposure"], M1["cur_keyframe_id"]
        fr[r]["kf_begin"], fr[r]["kf_count"] = r * len(M1["kfs"]), len(M1["kfs"])
        fr[r]["point_begin"], fr[r]["point_count"] = r * len(M1["points"]), len(M1["points"])
        pts = M1["points"].copy(); pts["obs_begin"] += r * len(M1["obs"])
        pa.append(pts); oa.append(M1["obs"])
    ka, pa, oa = np.concatenate([M1["kfs"]] * nseq), np.concatenate(pa), np.concatenate(oa)
    t_reproj = timed(lambda: ctx.reproject_match_multi(cam, fr, ka, pa, oa, M1["cell_size"], M1["grid_n_cols"]), 3)
    # marshalled once: the timed call is the C-ABI call (as for the alignment and seed stages above)
    pj_arr = (capi.PoseJob * nseq)(*pj[:nseq]); pj_res = (capi.PoseResult * nseq)()
    pj_masks = [np.zeros(max(j.n_feats, 1), np.uint8) for j in pj[:nseq]]
    pj_mptr = (C.c_void_p * nseq)(*[m.ctypes.data for m in pj_masks])
    t_pose = timed(lambda: ctx._check(ctx.lib.hso_gpu_pose_optimize_batch(ctx.h, C.byref(cam), pj_arr, nseq, pj_res, pj_mptr), "pose"), 5)
    ctx.frame_upload(1, pair["ref"]); ctx.frame_upload(2, pair["cur"])
    big2 = (capi.Seed * (nseq * len(seeds)))()
    for q in range(nseq):
        for i, sd in enumerate(seeds):
            C.memmove(C.byref(big2[q * len(seeds) + i]), C.byref(sd), C.sizeof(capi.Seed))
    sf2 = np.repeat(np.arange(nseq, dtype=np.int32), len(seeds))
    t_seed = timed(lambda: ctx.seed_observe_multi(cam, [(2, T_cur, 1.05)] * nseq, sf2, pea, big2, as_list=False), 3)
    total = t_track + t_reproj + t_pose + t_seed
    out.append(dict(stage="per-frame chain x256 sequences, value-passing calls (frame build + track 2000 pts, 1000 map points, pose 300 fts, 900 seeds)",
                    units="frames", n=nseq, ms_per_call=total * 1e3, units_per_s=nseq / total,
                    ms_track=t_track * 1e3, ms_reproject=t_reproj * 1e3, ms_pose=t_pose * 1e3, ms_seeds=t_seed * 1e3))

    # ---- the same chain with the state resident behind handles: raw images already in HBM (as in bench.py), the maps of the
    # 256 sequences stored once (hso_gpu_map_store: they change at keyframe rate), the seeds in one resident table with one group
    # per sequence; per frame only poses go in and compact records come out
    import torch
    cur_dev = [torch.from_numpy(pairs[i % 4]["cur"].copy()).cuda() for i in range(nseq)]
    cur_ptrs = np.array([t_.data_ptr() for t_ in cur_dev], np.uint64)

    def track_res():
        ctx.frame_upload_batch(cur_ids2, device_ptrs=cur_ptrs, width=640, height=480, want_stats=False)
        ctx.coarse_track_launch()
        return ctx.coarse_track_collect()
    t_track_r = timed(track_res, 5)
    ctx.map_reserve(nseq, 16, len(M1["points"]), len(M1["obs"]))
    M1["points"]["pad_"] = (4 << 4) | 0        # quality keys of the on-device selection: every point TYPE_GOOD, corner
    for r in range(nseq):
        ctx.map_store(r, M1["kfs"], M1["points"], M1["obs"])
    calls = np.zeros(nseq, capi.MAP_CALL_DTYPE)
    calls["map"] = np.arange(nseq); calls["cur_keyframe_id"] = M1["cur_keyframe_id"]; calls["cur_frame_id"] = M1["cur_frame_id"]
    calls["q"], calls["t"], calls["cur_exposure_time"] = q_cur, t_cur, M1["cur_exposure"]
    cap = nseq * len(M1["points"])
    t_reproj_r = timed(lambda: ctx.reproject_match_maps(cam, calls, M1["cell_size"], M1["grid_n_cols"], cap), 5)
    # ... and with the grid selection chained on the device: only the examined candidates come back (max_fts = 200)
    n_cells_sel = M1["grid_n_cols"] * int(math.ceil(480 / M1["cell_size"]))
    order_sel = np.random.default_rng(3).permutation(n_cells_sel).astype(np.int32)
    t_reproj_sel = timed(lambda: ctx.reproject_select_maps(cam, calls, M1["cell_size"], M1["grid_n_cols"], order_sel, 200, cap), 5)
    sel_out, _, sel_counts = ctx.reproject_select_maps(cam, calls, M1["cell_size"], M1["grid_n_cols"], order_sel, 200, cap)
    tab = ctx.seed_table_create()
    ctx.seed_table_append(tab, big2, group=sf2)
    frs = [(2, T_cur, 1.05)] * nseq
    t_seed_r = timed(lambda: ctx.seed_table_observe(cam, tab, frs, pea), 5)
    total_r = t_track_r + t_reproj_r + t_pose + t_seed_r
    out.append(dict(stage="per-frame chain x256 sequences, resident tables (images, maps and seeds in HBM; poses in, compact records out)",
                    units="frames", n=nseq, ms_per_call=total_r * 1e3, units_per_s=nseq / total_r,
                    ms_track=t_track_r * 1e3, ms_reproject=t_reproj_r * 1e3, ms_pose=t_pose * 1e3, ms_seeds=t_seed_r * 1e3,
                    ms_reproject_with_device_selection=t_reproj_sel * 1e3, examined_per_frame=float(sel_counts[:, 0].mean()),
                    matches_per_frame=float(sel_counts[:, 1].mean())))

    total_s = t_track_r + t_reproj_sel + t_pose + t_seed_r
    out.append(dict(stage="per-frame chain x256 sequences, resident tables + the grid selection on the device (only examined candidates return)",
                    units="frames", n=nseq, ms_per_call=total_s * 1e3, units_per_s=nseq / total_s,
                    ms_track=t_track_r * 1e3, ms_reproject_select=t_reproj_sel * 1e3, ms_pose=t_pose * 1e3, ms_seeds=t_seed_r * 1e3))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
