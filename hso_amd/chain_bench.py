"""The whole per-frame chain for many independent sequences at the metric's shape, through the resident-table entry points:

    frame build + CoarseTracker::run          hso_gpu_frame_upload_batch (device images) + hso_gpu_coarse_track_launch / collect
    reprojectMap (project, choose the view,   hso_gpu_reproject_select_maps
      findMatchDirect, grid selection)
    optimizeLevenbergMarquardt3rd             hso_gpu_pose_optimize_batch
    DepthFilter::updateSeeds                  hso_gpu_seed_table_observe

(reference src/frame_handler_mono.cpp:173-355 in processFrame order; keyframe-rate work — detection, activation, local BA — is
not part of the per-frame chain).  Every sequence has its own resident frames, map, seed group and feature table; the inputs
are `n_distinct` synthetic scenes replicated over the sequences (own copies in HBM), each a consistent little world: six
keyframes + the current frame rendered from one analytic scene, map points observed in the keyframes, seeds hosted in
keyframe 0, reference features of keyframe 0 for the tracker.  The stages are timed one after the other on the same state
(the chain's data dependencies between stages — the tracker's pose feeding the reprojection, the matches feeding the pose
optimiser — are replaced by the scene's true pose: a sum of stage times over consistent inputs, not an end-to-end VO run; the
C++ driver `hso_vo` is the end-to-end form, see hso_amd/latency_bench.py).

Algorithmic bytes per stage follow SURVEY.md section 8(d) ("Other stages' units"): tracker B_frame (bench.algorithmic_bytes) +
the frame build; per candidate that reaches the LK stage 400 B of warped reference taps + iterations x 256 B; per pose
evaluation n_features x sizeof(hso_pose_feat); per seed epipolar steps x 256 B.  Iteration / step counts come from one
value-passing call per distinct scene (the resident forms return compact records without them).

Development / measurement code: bench.py embeds `measure()` in its JSON line; uses no CPU reference."""
import ctypes as C
import math
import time

import numpy as np

from hso_amd import capi, synth

KF_PER_SEQ = 8          # frame ids reserved per sequence: six keyframes + the far one + the current frame


def build_scene(job):
    """One distinct scene (numpy only; runs in a worker process before any GPU runtime exists)."""
    spec, feats, n_points, n_seeds, seed = job
    M = synth.map_problem(n_points=n_points, spec=spec, seed=seed, first_frame_id=0, max_fts=feats)
    sc = M["scene"]
    q_cur, t_cur = M["T_cur_w"].to_arrays()
    tf = sc.features(np.array([0, 0, 0, 1.0]), np.zeros(3), feats, seed=seed + 40)
    gy0, gx0 = np.gradient(M["frames"][0].astype(np.float64))
    pair = dict(scene=sc, q_true=q_cur, t_true=t_cur)
    seeds, _, _ = synth.seeds_for_pair(pair, n_seeds, 0, seed=seed + 50, gx=gx0, gy=gy0)
    seeds_bytes = b"".join(bytes(s) for s in seeds)
    M = {k: v for k, v in M.items() if k != "scene"}
    M["T_cur_w"] = (q_cur, t_cur)
    return dict(M=M, track_feats=tf, seeds_bytes=seeds_bytes, n_seeds=n_seeds)


def scene_jobs(spec, feats, n_points, n_seeds, n_distinct, seed0=7100):
    return [(dict(spec), feats, n_points, n_seeds, seed0 + 13 * k) for k in range(n_distinct)]


class Chain:
    def __init__(self, ctx, stream, spec, scenes, nseq, feats, first_frame_id=400000):
        import torch
        self.ctx, self.stream, self.spec, self.nseq, self.feats = ctx, stream, spec, nseq, feats
        self.cam = synth.camera(spec)
        self.W, self.H = spec["width"], spec["height"]
        nd = len(scenes)
        self.scenes = scenes
        fid = lambda q, k: first_frame_id + q * KF_PER_SEQ + k
        self.fid = fid
        # ---- resident frames: every sequence its own copies
        ids, imgs = [], []
        for q in range(nseq):
            M = scenes[q % nd]["M"]
            for k, f in enumerate(M["frames"]):
                ids.append(fid(q, k)); imgs.append(f)
        self.st_kf = ctx.frame_upload_batch(ids, imgs=imgs)
        self.cur_ids = [fid(q, KF_PER_SEQ - 1) for q in range(nseq)]
        self.cur_dev = [torch.from_numpy(scenes[q % nd]["M"]["cur"].copy()).cuda() for q in range(nseq)]
        self.cur_ptrs = np.array([t.data_ptr() for t in self.cur_dev], np.uint64)
        st_cur = ctx.frame_upload_batch(self.cur_ids, device_ptrs=self.cur_ptrs, width=self.W, height=self.H)
        n_fr = len(scenes[0]["M"]["frames"])
        # ---- tracker jobs: keyframe 0 -> current frame, motion-model-like start (0.8 of the true motion)
        self.params = capi.TrackParams(0, 4, 1, 50)
        jobs = []
        for q in range(nseq):
            S = scenes[q % nd]
            qc, tc = S["M"]["T_cur_w"]
            rv = 2 * np.arctan2(np.linalg.norm(qc[:3]), qc[3]) * qc[:3] / max(np.linalg.norm(qc[:3]), 1e-12)
            T0 = capi.SE3.from_arrays(synth.rotvec_to_quat(0.8 * rv), 0.8 * tc)
            a0 = float(np.float32(st_cur[q].integral_image / self.st_kf[q * n_fr].integral_image))
            jobs.append(ctx.make_job(fid(q, 0), self.cur_ids[q], S["track_feats"], T0, a0))
        self.jobs = jobs
        ctx.coarse_track_prepare(self.cam, self.params, jobs)
        # ---- resident maps
        M0 = scenes[0]["M"]
        ctx.map_reserve(nseq, 16, len(M0["points"]), max(len(S["M"]["obs"]) for S in scenes))
        for q in range(nseq):
            M = scenes[q % nd]["M"]
            kfs = M["kfs"].copy(); kfs["frame_id"] = [fid(q, k) for k in range(len(kfs))]
            pts = M["points"].copy(); pts["pad_"] = (4 << 4) | 0        # quality keys of the device selection: TYPE_GOOD corners
            ctx.map_store(q, kfs, pts, M["obs"])
        calls = np.zeros(nseq, capi.MAP_CALL_DTYPE)
        calls["map"] = np.arange(nseq); calls["cur_keyframe_id"] = M0["cur_keyframe_id"]; calls["cur_frame_id"] = self.cur_ids
        for q in range(nseq):
            calls[q]["q"], calls[q]["t"] = scenes[q % nd]["M"]["T_cur_w"]
        calls["cur_exposure_time"] = M0["cur_exposure"]
        self.calls = calls
        self.map_ids = np.arange(nseq, dtype=np.int32)
        self.quality = np.full(nseq * len(M0["points"]), (4 << 4) | 0, np.uint8)
        self.cell_size, self.grid_n_cols = M0["cell_size"], M0["grid_n_cols"]
        n_cells = self.grid_n_cols * int(math.ceil(self.H / self.cell_size))
        self.cell_order = np.random.default_rng(3).permutation(n_cells).astype(np.int32)
        self.sel_cap = nseq * len(M0["points"])
        # ---- pose jobs (a synthetic problem of `feats` features per frame; the optimiser sees no images)
        pf, poses, T0, _ = synth.pose_problem(feats, seed=5)
        self.pose_feats = pf
        pj = [capi.make_pose_job(pf, poses, T0) for _ in range(nseq)]
        self._pj = pj
        self.pj_arr = (capi.PoseJob * nseq)(*pj); self.pj_res = (capi.PoseResult * nseq)()
        self._masks = [np.zeros(max(j.n_feats, 1), np.uint8) for j in pj]
        self.pj_mptr = (C.c_void_p * nseq)(*[m.ctypes.data for m in self._masks])
        # ---- resident seed table: one group per sequence, hosted in that sequence's keyframe 0
        n_seeds = scenes[0]["n_seeds"]
        raw = bytearray(b"".join(scenes[q % nd]["seeds_bytes"] for q in range(nseq)))
        arr = (capi.Seed * (nseq * n_seeds)).from_buffer(raw)
        ref_ids = np.frombuffer(raw, np.int64)[:: C.sizeof(capi.Seed) // 8]
        assert capi.Seed.ref_frame_id.offset == 0
        ref_ids[:] = np.repeat([fid(q, 0) for q in range(nseq)], n_seeds)
        self.tab = ctx.seed_table_create()
        ctx.seed_table_append(self.tab, arr, group=np.repeat(np.arange(nseq, dtype=np.int32), n_seeds))
        self._seed_raw = raw
        self.seed_frames = [(self.cur_ids[q], capi.SE3.from_arrays(*scenes[q % nd]["M"]["T_cur_w"]), M0["cur_exposure"]) for q in range(nseq)]
        self.pea = math.atan(1.0 / (2.0 * spec["fx"])) * 2.0
        self.n_seeds, self.n_points = n_seeds, len(M0["points"])
        # result tables in page-locked memory (hso_gpu_host_alloc), allocated once like a caller's per-sequence buffers would be:
        # the match / seed records of a step (tens of MB at 256 sequences) are then DMA'd straight into them
        self.sel_out = ctx.host_array(self.sel_cap, capi.MATCH_BRIEF_DTYPE)
        self.seed_out = ctx.host_array(nseq * n_seeds, capi.SEED_BRIEF_DTYPE)

    # ---- the four stages
    def track(self):
        self.ctx.frame_upload_batch(self.cur_ids, device_ptrs=self.cur_ptrs, width=self.W, height=self.H, want_stats=False)
        self.ctx.coarse_track_launch()
        return self.ctx.coarse_track_collect(as_list=False)

    def reproject(self):
        return self.ctx.reproject_select_maps(self.cam, self.calls, self.cell_size, self.grid_n_cols, self.cell_order, self.feats, self.sel_cap)

    def reproject_pose(self):
        """reprojectMap + optimizeLevenbergMarquardt3rd chained on the device (the selected matches ARE the frame's features).
        The maps' per-frame part goes in first: the quality keys of every point (Point::type_ promotions and deletions happen
        between keyframes, src/reprojector.cpp:376-423) — one byte per stored point and frame."""
        self.ctx.map_update_quality(self.map_ids, self.quality)
        return self.ctx.reproject_select_pose_maps(self.cam, self.calls, self.cell_size, self.grid_n_cols, self.cell_order, self.feats, self.sel_cap,
                                                   want_mask=True, out=self.sel_out)

    def pose(self):
        self.ctx._check(self.ctx.lib.hso_gpu_pose_optimize_batch(self.ctx.h, C.byref(self.cam), self.pj_arr, self.nseq, self.pj_res, self.pj_mptr), "pose")
        return self.pj_res

    def seeds(self):
        return self.ctx.seed_table_observe(self.cam, self.tab, self.seed_frames, self.pea, brief_out=self.seed_out)

    # ---- work counts for the algorithmic bytes (one value-passing call per distinct scene)
    def work_counts(self):
        out = []
        for d, S in enumerate(self.scenes):
            M = S["M"]
            q = d                                                 # sequence d uses scene d
            kfs = M["kfs"].copy(); kfs["frame_id"] = [self.fid(q, k) for k in range(len(kfs))]
            T = capi.SE3.from_arrays(*M["T_cur_w"])
            proj, match = self.ctx.reproject_match(self.cam, self.cur_ids[q], T, M["cur_exposure"], M["cur_keyframe_id"], kfs, M["points"],
                                                   M["obs"], self.cell_size, self.grid_n_cols)
            iters = np.array([m.iters for m in match]); reached = int((iters > 0).sum())
            n = S["n_seeds"]
            seeds = (capi.Seed * n).from_buffer_copy(S["seeds_bytes"])
            for s in seeds:
                s.ref_frame_id = self.fid(q, 0)
            so = self.ctx.seed_observe(self.cam, self.cur_ids[q], T, M["cur_exposure"], self.pea, seeds, as_list=False)
            steps = np.array([max(o.n_steps, 0) for o in so])
            out.append(dict(reached_lk=reached, lk_iters=int(iters.sum()), projected=int(proj["projected"].sum()),
                            obs=int(M["points"]["obs_count"].sum()), steps=int(steps.sum()),
                            seed_updates=int(sum(o.result == 1 for o in so)), matched=int(sum(m.success for m in match))))
        return out


def timed(fn, reps, stream=None):
    import torch
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)] if stream is not None else None
    t0 = time.perf_counter()
    for k in range(reps):
        if ev:
            ev[k][0].record(stream)
        fn()
        if ev:
            ev[k][1].record(stream)
    wall = (time.perf_counter() - t0) / reps
    if ev:
        stream.synchronize()        # the closing events are recorded behind the calls' own synchronisation points
    dev = float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3 if ev else None
    return wall, dev


def measure(ctx, stream, spec, scenes, nseq=256, feats=2000, reps=5, algorithmic_bytes=None):
    """-> dict for the bench line: per-stage ms (wall time of the C-ABI call incl. its copies and synchronise; `device_ms` =
    between two HIP events on the launch stream), frames/s of the chain, algorithmic-byte fraction per stage."""
    ch = Chain(ctx, stream, spec, scenes, nseq, feats)
    res = ch.track()
    counts = ch.work_counts()
    nd = len(scenes)
    rep = lambda key: sum(counts[q % nd][key] for q in range(nseq))
    t = {}
    for name, fn in (("track", ch.track), ("reproject_select", ch.reproject), ("pose", ch.pose), ("seeds", ch.seeds),
                     ("reproject_select_pose", ch.reproject_pose)):
        t[name] = timed(fn, reps, stream)
    res = ch.track()
    pres = ch.pose()
    sel_out, _, sel_counts = ch.reproject()
    brief, _ = ch.seeds()
    W, H = ch.W, ch.H
    n_valid = [int((scenes[q % nd]["track_feats"]["dist"] >= 0).sum()) for q in range(nseq)]
    b_track = (algorithmic_bytes(res, n_valid, False, (4, 3, 2, 1)) if algorithmic_bytes else 0) + nseq * W * H * (1.33 + 0.33 + 1.31 + 5.25)
    b_repr = nseq * ch.n_points * capi.MAP_POINT_DTYPE.itemsize + rep("obs") * capi.OBS_DTYPE.itemsize + rep("reached_lk") * 400 + rep("lk_iters") * 256
    evals = sum(int(r.n_trials_total) + int(r.iters) + 1 for r in pres)
    b_pose = evals * len(ch.pose_feats) * C.sizeof(capi.PoseFeat)
    b_seed = rep("steps") * 256 + nseq * ch.n_seeds * C.sizeof(capi.Seed)
    _, _, _, cres, cnf, _ = ch.reproject_pose()
    c_evals = sum(int(r.n_trials_total) + int(r.iters) + 1 for r in cres[:nseq])
    b_cpose = c_evals * float(np.mean(cnf)) * C.sizeof(capi.PoseFeat)
    # the chain as it runs: the pose optimisation consumes the selection's result on the device
    total = t["track"][0] + t["reproject_select_pose"][0] + t["seeds"][0]
    total_value_passing = t["track"][0] + t["reproject_select"][0] + t["pose"][0] + t["seeds"][0]
    stage = lambda k, b, **kw: dict(ms=t[k][0] * 1e3, device_ms=t[k][1] * 1e3, algorithmic_bytes=float(b),
                                    frac_of_hbm_peak=float(b) / t[k][0] / 8e12, **kw)
    out = {
        "sequences": nseq, "shape": "%dx%d" % (W, H), "features": feats, "map_points_per_sequence": ch.n_points,
        "seeds_per_sequence": ch.n_seeds, "distinct_scenes": nd,
        "frames_per_s": nseq / total, "ms_per_step": total * 1e3,
        "frames_per_s_pose_tables_through_pcie": nseq / total_value_passing,
        "stages": {
            "track": stage("track", b_track, mean_evaluations=float(np.mean([sum(r.n_eval[L] for L in (4, 3, 2, 1)) for r in res]))),
            "reproject_select": stage("reproject_select", b_repr, candidates_matched=rep("reached_lk"), lk_iterations=rep("lk_iters"),
                                      examined_per_frame=float(sel_counts[:, 0].mean()), matches_per_frame=float(sel_counts[:, 1].mean())),
            "reproject_select_pose": stage("reproject_select_pose", b_repr + b_cpose, pose_evaluations=c_evals, features_per_frame=float(np.mean(cnf)),
                                           note="the chain's form: selection and pose optimisation in one call, feature tables built on the device"),
            "pose": stage("pose", b_pose, evaluations=evals, features_per_frame=len(ch.pose_feats),
                          note="value-passing form on a synthetic table of the same size (96-byte feature records through PCIe): not in frames_per_s"),
            "seeds": stage("seeds", b_seed, epipolar_steps=rep("steps"), seeds_updated=int((brief["result"] == 1).sum())),
        },
        "what": "sum of the per-frame stage calls track + reproject_select_pose + seeds (resident tables; images, maps, seeds in HBM; poses "
                "in, compact records out) over %d independent sequences; stage inputs are consistent synthetic worlds, not one chained VO "
                "state; reproject_select and pose alone are listed for the split" % nseq,
    }
    # sanity: the stages did real work
    qc, tc = scenes[0]["M"]["T_cur_w"]
    terr = float(np.linalg.norm(np.array(res[0].T_cur_ref.t[:]) - tc))
    out["track_trans_err_vs_truth"] = terr
    assert terr < 2e-2, terr
    assert out["stages"]["seeds"]["seeds_updated"] > 0.2 * nseq * ch.n_seeds and sel_counts[:, 1].mean() > 0.3 * min(feats, ch.n_points)
    ctx.seed_table_destroy(ch.tab)
    return out, ch


def main():
    import argparse
    import json
    import multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", type=int, default=256)
    ap.add_argument("--feats", type=int, default=2000)
    ap.add_argument("--points", type=int, default=4000)
    ap.add_argument("--seeds", type=int, default=6000)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--shape", default="euroc")
    args = ap.parse_args()
    spec = synth.EUROC if args.shape == "euroc" else synth.ICL_NUIM
    jobs = scene_jobs(spec, args.feats, args.points, args.seeds, args.distinct)
    with mp.get_context("fork").Pool(min(len(jobs), 8)) as pool:
        scenes = pool.map(build_scene, jobs, chunksize=1)
    import torch
    import bench
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = capi.Context(0, stream.cuda_stream)
        out, _ = measure(ctx, stream, spec, scenes, args.nseq, args.feats, algorithmic_bytes=bench.algorithmic_bytes)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
