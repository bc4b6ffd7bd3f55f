"""TEST INFRASTRUCTURE for tests/test_seq_chain.py: one sequence-map state handed to hso_gpu_seq_chain in libhso_gpu.so and to its
sequential restatement (tests/fakegpu, built on oracle/) through the same entry points.

A state is what the engine records with hso_vo_trace_state (include/hso_vo.h): the job, the call's configuration and every table
of the sequence map as the device held it right before the call (hso_gpu_seqmap_debug_dump, include/hso_gpu_debug.h).  `load()`
rebuilds it in a fresh context of either library with the public calls (frame upload, hso_gpu_seqmap_*), `run()` makes the chain
call and reads back everything the call produced."""
import ctypes as C

import numpy as np

from hso_amd import capi, vo

SE3 = np.dtype([("q", "<f8", 4), ("t", "<f8", 3)])
KF = np.dtype([("frame_id", "<i8"), ("T_f_w", SE3), ("exposure_time", "<f8"), ("keyframe_id", "<i4"), ("pad_", "<i4")], align=True)
OBS = np.dtype([("kf", "<i4"), ("level", "<i4"), ("type", "<i4"), ("pad_", "<i4"), ("px", "<f8", 2), ("f", "<f8", 3), ("grad", "<f8", 2)], align=True)
MAP_POINT = np.dtype([("pos", "<f8", 3), ("idist", "<f8"), ("host_f", "<f8", 3), ("host_kf", "<i4"), ("obs_begin", "<i4"), ("obs_count", "<i4"),
                      ("pad_", "<i4")], align=True)
SEQ_FEATURE = np.dtype([("px", "<f8", 2), ("f", "<f8", 3), ("grad", "<f4", 2), ("point", "<i4"), ("level", "i1"), ("type", "i1"), ("pad_", "<i2")], align=True)
REF_FEAT = capi.REF_FEAT_DTYPE
TRACK_PARAMS = np.dtype([("inverse_composition", "<i4"), ("max_level", "<i4"), ("min_level", "<i4"), ("n_iter", "<i4")], align=True)
TRACK_RESULT = np.dtype([("T_cur_ref", SE3), ("exposure_rat", "<f4"), ("n_tracked", "<i4"), ("n_terms_last", "<i4"), ("n_saturated_last", "<i4"),
                         ("iters", "<i4", 5), ("n_eval", "<i4", 5), ("accept_mask", "<u8", 5), ("huber", "<f4", 5), ("outlier", "<f4", 5),
                         ("n_select", "<i4", 5), ("energy", "<f8", 5), ("phase_cycles", "<u8", 10), ("status", "<i4"), ("coop_workgroups", "<i2"),
                         ("coop_same_xcd", "<i2")], align=True)
POSE_RESULT = np.dtype([("T_f_w", SE3), ("cov", "<f8", 36), ("estimated_scale", "<f8"), ("error_init", "<f8"), ("error_final", "<f8"),
                        ("error_in_px", "<f4"), ("num_obs", "<i4"), ("n_deleted", "<i4"), ("iters", "<i4"), ("n_trials_total", "<i4"), ("status", "<i4")], align=True)
SEQ_JOB = np.dtype([("map", "<i4"), ("flags", "<i4"), ("ref_frame_id", "<i8"), ("cur_frame_id", "<i8"), ("T_ref_w", SE3), ("T_cur_w", SE3),
                    ("ref_exposure", "<f8"), ("ref_kf_row", "<i4"), ("n_ref_feats", "<i4"), ("cur_keyframe_id", "<i4"), ("last_kf_row", "<i4"),
                    ("covis", "<i4", 5), ("temps_begin", "<i4"), ("n_temps", "<i4"), ("exposure_rat", "<f4"), ("seed_group", "<i4"), ("pad_", "<i4")], align=True)
SEQ_CFG = np.dtype([("track", TRACK_PARAMS), ("cell_size", "<i4"), ("grid_n_cols", "<i4"), ("n_cells", "<i4"), ("max_fts", "<i4"), ("cell_order", "<u8"),
                    ("max_kfs", "<i4"), ("pose_n_iter", "<i4"), ("pose_reproj_thresh", "<f8"), ("quality_min_fts", "<i4"), ("want_debug", "<i4"),
                    ("seed_table", "<i4"), ("n_seed_groups", "<i4"), ("px_error_angle", "<f8"), ("seed_brief_out", "<u8"), ("seed_brief_cap", "<i4"),
                    ("pad_", "<i4")], align=True)
MAX_VISIT, MAX_COVIS, N_EVENTS = 24, 8, 120
SEQ_RESULT = np.dtype([("track", TRACK_RESULT), ("pose", POSE_RESULT), ("T_tracked", SE3), ("exposure", "<f8"), ("counts", "<i4", 4), ("n_feats", "<i4"),
                       ("n_listed", "<i4"), ("n_kf_points", "<i4"), ("n_candidates", "<i4"), ("n_visit", "<i4"), ("visit", "<i4", MAX_VISIT),
                       ("flow_full", "<f4"), ("flow_shift", "<f4"), ("flow_count", "<i4"), ("n_with_point", "<i4"), ("n_covis", "<i4"),
                       ("covis", "<i4", MAX_COVIS), ("covis_votes", "<i4", MAX_COVIS), ("covis_best", "<i4"), ("make_kf", "<i4"), ("seeds_observed", "<i4"),
                       ("depth_median", "<f8"), ("dist_median", "<f8"), ("depth_min", "<f8"), ("n_events", "<i4"), ("events", "<i4", N_EVENTS)], align=True)
LIST_PATCH = np.dtype([("map", "<i4"), ("list", "<i4"), ("first", "<i4"), ("n", "<i4"), ("ids", "<u8")], align=True)
SEQ_NO_TRACK, SEQ_SEED_BRANCH, SEQ_DEPTH_STATS = 1, 2, 4
EV_ERASE_POINT, EV_ERASE_CANDIDATE, EV_TEMP_BAD, EV_GOOD = 1, 2, 3, 4
DUMP = dict(sizes=0, kfs=1, points=2, obs=3, obs_point=4, key_points=5, kf_nfts=6, kf_fts=7, cands=8, frame_feats0=9, frame_feats1=10)


SEQ_BA_JOB = np.dtype([("map", "<i4"), ("n_core", "<i4"), ("core", "<i4", 16), ("fixed", "u1", 16), ("n_iter", "<i4"), ("point_cap", "<i4"), ("cull_cap", "<i4"),
                       ("pad_", "<i4"), ("point_ids", "<u8"), ("point_state", "<u8"), ("culled", "<u8")])
BA_RESULT = np.dtype([("init_chi2", "<f8"), ("final_chi2", "<f8"), ("robust_chi2", "<f8"), ("lambda_", "<f8"), ("iterations", "<i4"), ("n_solves", "<i4"),
                      ("n_accepted", "<i4"), ("stop", "<i4")])
SEQ_BA_RESULT = np.dtype([("lm", BA_RESULT), ("huber_corner", "<f4"), ("huber_edge", "<f4"), ("status", "<i4"), ("n_poses", "<i4"), ("n_points", "<i4"),
                          ("n_edges", "<i4"), ("n_culled", "<i4", 2), ("core_pose", SE3, 16)])
assert SEQ_BA_JOB.itemsize == 128 and SEQ_BA_RESULT.itemsize == 80 + 16 * 56
BAW = dict(sizes=0, vertex_rows=1, fixed=2, edges=3, obs_uv=4, edge_obs=5, edge_chi2=6, poses_out=7, poses_in=8, idist_in=9)


def pt_word(key, n_fail=0, bad=False, n_ok=0):
    """HSO_PT_WORD (include/hso_gpu.h): the state word of a sequence map's point row"""
    w = (key & 0xff) | ((n_fail & 0x3ff) << 8) | ((1 << 18) if bad else 0) | ((n_ok & 0x7ff) << 20)
    return np.int32(np.uint32(w).view(np.int32))


def pt_key(w):
    return np.asarray(w).view(np.uint32) & 0xff if isinstance(w, np.ndarray) else (int(np.uint32(w)) & 0xff)


def pt_nfail(w):
    return (np.asarray(w).astype(np.int64) & 0xffffffff) >> 8 & 0x3ff


def pt_nok(w):
    return (np.asarray(w).astype(np.int64) & 0xffffffff) >> 20 & 0x7ff


def fields_differ(a, b, skip=(), prefix=""):
    """names of the fields (recursively) in which two records of one structured dtype differ — padding bytes are not compared"""
    bad = []
    for name in a.dtype.names:
        if name in skip or name.startswith("pad_") or name == "phase_cycles":
            continue
        x, y = a[name], b[name]
        if x.dtype.names:
            bad += fields_differ(x, y, skip, prefix + name + ".")
        elif not np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True):
            bad.append(prefix + name)
    return bad


def state_from_record(rec):
    """a "seq_chain_state" trace record (hso_amd/host/hso_engine_init.cpp: trace_chain_state) -> dict of arrays"""
    sz = np.frombuffer(rec["sizes"], "<i8")
    assert (sz[10], sz[11], sz[12], sz[13], sz[14], sz[15]) == (KF.itemsize, MAP_POINT.itemsize, OBS.itemsize, SEQ_FEATURE.itemsize, SEQ_JOB.itemsize,
                                                               SEQ_RESULT.itemsize), ("struct layouts of this file differ from the library's", sz[10:16])
    S = dict(sizes=sz.copy(), cam=capi.Camera.from_buffer_copy(rec["cam"]), job=np.frombuffer(rec["job"], SEQ_JOB).copy(), cfg=np.frombuffer(rec["cfg"], SEQ_CFG).copy(),
             cell_order=np.frombuffer(rec["cell_order"], "<i4").copy(), temps=np.frombuffer(rec["temps"], "<i4").copy(), kfs=np.frombuffer(rec["kfs"], KF).copy(),
             points=np.frombuffer(rec["points"], MAP_POINT).copy(), obs=np.frombuffer(rec["obs"], OBS).copy(), obs_point=np.frombuffer(rec["obs_point"], "<i4").copy(),
             key_points=np.frombuffer(rec["key_points"], "<i4").copy(), kf_nfts=np.frombuffer(rec["kf_nfts"], "<i4").copy(), cands=np.frombuffer(rec["cands"], "<i4").copy(),
             ff=[np.frombuffer(rec["frame_feats0"], SEQ_FEATURE).copy(), np.frombuffer(rec["frame_feats1"], SEQ_FEATURE).copy()],
             ff_frame=[int(sz[7]), int(sz[8])], ff_newest=int(sz[9]), fts_cap=int(sz[3]))
    nk = len(S["kfs"])
    lists = np.frombuffer(rec["kf_fts"], "<i4").reshape(nk, S["fts_cap"]) if nk else np.zeros((0, 0), "<i4")
    S["kf_fts"] = [lists[r, :S["kf_nfts"][r]].copy() for r in range(nk)]
    S["job"]["temps_begin"] = 0
    return S


def _scalar(rec, key):
    return float(np.frombuffer(rec[key], "<f8")[0])


def ba_state_from_record(rec):
    """a "seq_ba_state" trace record (trace_ba_state) -> the map's tables and what the call was asked"""
    sz = np.frombuffer(rec["sizes"], "<i8")
    assert (sz[10], sz[11], sz[12], sz[13]) == (KF.itemsize, MAP_POINT.itemsize, OBS.itemsize, SEQ_FEATURE.itemsize), ("struct layouts of this file differ from the library's", sz[10:14])
    S = dict(sizes=sz.copy(), job=None, kfs=np.frombuffer(rec["kfs"], KF).copy(), points=np.frombuffer(rec["points"], MAP_POINT).copy(), obs=np.frombuffer(rec["obs"], OBS).copy(),
             obs_point=np.frombuffer(rec["obs_point"], "<i4").copy(), key_points=np.frombuffer(rec["key_points"], "<i4").copy(), kf_nfts=np.frombuffer(rec["kf_nfts"], "<i4").copy(),
             cands=np.frombuffer(rec["cands"], "<i4").copy(), ff=[np.frombuffer(rec["frame_feats0"], SEQ_FEATURE).copy(), np.frombuffer(rec["frame_feats1"], SEQ_FEATURE).copy()],
             ff_frame=[int(sz[7]), int(sz[8])], ff_newest=int(sz[9]), fts_cap=int(sz[3]),
             core=np.frombuffer(rec["core"], "<i4").copy(), fixed=np.frombuffer(rec["fixed"], "u1").copy(), n_iter=int(_scalar(rec, "n_iter")),
             error_multiplier2=_scalar(rec, "error_multiplier2"), chi2_corner=_scalar(rec, "chi2_corner"), chi2_edgelet=_scalar(rec, "chi2_edgelet"))
    nk = len(S["kfs"])
    lists = np.frombuffer(rec["kf_fts"], "<i4").reshape(nk, S["fts_cap"]) if nk else np.zeros((0, 0), "<i4")
    S["kf_fts"] = [lists[r, :S["kf_nfts"][r]].copy() for r in range(nk)]
    return S


def ba_result_from_record(rec):
    return dict(result=np.frombuffer(rec["result"], SEQ_BA_RESULT).copy()[0], point_ids=np.frombuffer(rec["point_ids"], "<i4").copy(),
                point_state=np.frombuffer(rec["point_state"], "<f8").reshape(-1, 4).copy(), culled=np.frombuffer(rec["culled"], "<i4").copy())


def result_from_record(rec):
    return dict(result=np.frombuffer(rec["result"], SEQ_RESULT).copy()[0], events=np.frombuffer(rec["events"], "<i4").copy(),
                features=np.frombuffer(rec["features"], SEQ_FEATURE).copy())


class ChainLib:
    """The entry points the test needs, declared on a CDLL: libhso_gpu.so (capi.load()) or tests/fakegpu's library"""

    def __init__(self, cdll):
        L = self.L = cdll
        vp, i32, i64, P = C.c_void_p, C.c_int, C.c_int64, C.POINTER
        L.hso_gpu_create.argtypes = [P(vp), i32, vp]
        L.hso_gpu_destroy.argtypes = [vp]; L.hso_gpu_destroy.restype = None
        L.hso_gpu_last_error.argtypes = [vp]; L.hso_gpu_last_error.restype = C.c_char_p
        L.hso_gpu_frame_upload_batch.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        L.hso_gpu_seqmap_create.argtypes = [vp, P(i32)]
        L.hso_gpu_seqmap_configure.argtypes = [vp, i32, i32]
        L.hso_gpu_seqmap_set_keyframes.argtypes = [vp, i32, vp, i32]
        L.hso_gpu_seqmap_set_key_points.argtypes = [vp, i32, vp, i32]
        L.hso_gpu_seqmap_patch.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32]
        L.hso_gpu_seqmap_patch_links.argtypes = [vp, i32, vp, vp, i32]
        L.hso_gpu_seqmap_patch_lists.argtypes = [vp, vp, i32]
        L.hso_gpu_seq_set_frame_features.argtypes = [vp, i32, i64, vp, i32]
        L.hso_gpu_seq_frame_features.argtypes = [vp, vp, vp, i32, vp, i32, vp]
        L.hso_gpu_seq_chain.argtypes = [vp, P(capi.Camera), vp, vp, i32, vp, i32, vp]
        L.hso_gpu_seq_events.argtypes = [vp, i32, vp, i32]
        L.hso_gpu_seq_debug_list.argtypes = [vp, i32, vp, vp, i32]
        L.hso_gpu_seq_debug_ref_table.argtypes = [vp, i32, vp, i32]
        L.hso_gpu_seqmap_debug_dump.argtypes = [vp, i32, i32, vp, C.c_size_t]
        L.hso_gpu_seq_local_ba.argtypes = [vp, vp, i32, C.c_double, C.c_double, C.c_double, vp]
        L.hso_gpu_seq_ba_debug_window.argtypes = [vp, i32, i32, vp, C.c_size_t]

    def check(self, ctx, rc, what):
        if rc < 0:
            raise capi.HsoGpuError("%s failed (%d): %s" % (what, rc, (self.L.hso_gpu_last_error(ctx) or b"?").decode()))
        return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LoadedState:
    """One state in one library's fresh context"""

    def __init__(self, lib, S, images):
        """images: frame_id -> level-0 image (every keyframe of the map, the job's reference and current frame)"""
        self.lib, self.S = lib, S
        L = lib.L
        self.ctx = C.c_void_p()
        lib.check(None, L.hso_gpu_create(C.byref(self.ctx), 0, None), "create")
        ctx = self.ctx
        need = set(int(k) for k in S["kfs"]["frame_id"])
        if S.get("job") is not None:
            need |= {int(S["job"]["ref_frame_id"][0]), int(S["job"]["cur_frame_id"][0])}
        need = sorted(need)
        if images is None:                                           # a call that never looks at pixels (local BA): any resident frame will do
            images = {i: np.zeros((64, 64), np.uint8) for i in need}
        h, w = next(iter(images.values())).shape
        imgs = [np.ascontiguousarray(images[i], np.uint8) for i in need]
        ids = np.array(need, np.int64)
        ptrs = (C.c_void_p * len(imgs))(*[im.ctypes.data for im in imgs])
        stats = np.zeros(len(imgs) * 4, np.float32)
        lib.check(ctx, L.hso_gpu_frame_upload_batch(ctx, _p(ids), ptrs, len(imgs), w, h, 0, _p(stats)), "frame_upload_batch")
        m = C.c_int(-1)
        lib.check(ctx, L.hso_gpu_seqmap_create(ctx, C.byref(m)), "seqmap_create")
        self.map = m.value
        lib.check(ctx, L.hso_gpu_seqmap_configure(ctx, self.map, S["fts_cap"]), "seqmap_configure")
        kfs = np.ascontiguousarray(S["kfs"])
        lib.check(ctx, L.hso_gpu_seqmap_set_keyframes(ctx, self.map, _p(kfs), len(kfs)), "seqmap_set_keyframes")
        pts, obs = np.ascontiguousarray(S["points"]), np.ascontiguousarray(S["obs"])
        pid, oid = np.arange(len(pts), dtype=np.int32), np.arange(len(obs), dtype=np.int32)
        # observation rows first (a point row's obs_begin must exist), in two calls: the library checks a patch's rows against the
        # tables as they stand
        lib.check(ctx, L.hso_gpu_seqmap_patch(ctx, self.map, None, None, 0, _p(oid), _p(obs), len(obs)), "seqmap_patch (observations)")
        lib.check(ctx, L.hso_gpu_seqmap_patch(ctx, self.map, _p(pid), _p(pts), len(pts), None, None, 0), "seqmap_patch (points)")
        link = np.ascontiguousarray(S["obs_point"], np.int32)
        lib.check(ctx, L.hso_gpu_seqmap_patch_links(ctx, self.map, _p(oid), _p(link), len(oid)), "seqmap_patch_links")
        keep = []
        patches = np.zeros(len(kfs) + 1, LIST_PATCH)
        for r in range(len(kfs)):
            ids_r = np.ascontiguousarray(S["kf_fts"][r], np.int32); keep.append(ids_r)
            patches[r] = (self.map, r, 0, len(ids_r), ids_r.ctypes.data if len(ids_r) else 0)
        cands = np.ascontiguousarray(S["cands"], np.int32); keep.append(cands)
        patches[len(kfs)] = (self.map, -1, 0, len(cands), cands.ctypes.data if len(cands) else 0)
        lib.check(ctx, L.hso_gpu_seqmap_patch_lists(ctx, _p(patches), len(patches)), "seqmap_patch_lists")
        keys = np.ascontiguousarray(S["key_points"], np.int32)
        lib.check(ctx, L.hso_gpu_seqmap_set_key_points(ctx, self.map, _p(keys), len(kfs)), "seqmap_set_key_points")
        # the two frame feature tables, the older one first so that the map's "newest" ends up where it was
        order = [1 - S["ff_newest"], S["ff_newest"]]
        for b in order:
            if S["ff_frame"][b] < 0:
                continue
            ff = np.ascontiguousarray(S["ff"][b])
            lib.check(ctx, L.hso_gpu_seq_set_frame_features(ctx, self.map, int(S["ff_frame"][b]), _p(ff) if len(ff) else None, len(ff)), "seq_set_frame_features")

    def dump(self):
        """the map as this library holds it now, in the layout of state_from_record (tables only)"""
        L, ctx = self.lib.L, self.ctx
        sz = np.zeros(16, np.int64)
        self.lib.check(ctx, L.hso_gpu_seqmap_debug_dump(ctx, self.map, 0, _p(sz), sz.nbytes), "dump sizes")
        nk, npnt, nobs, cap, nc = (int(x) for x in sz[:5])

        def get(what, dtype, n):
            a = np.zeros(n, dtype)
            if a.nbytes:
                self.lib.check(ctx, L.hso_gpu_seqmap_debug_dump(ctx, self.map, DUMP[what], _p(a), a.nbytes), "dump " + what)
            return a
        D = dict(sizes=sz, kfs=get("kfs", KF, nk), points=get("points", MAP_POINT, npnt), obs=get("obs", OBS, nobs), obs_point=get("obs_point", "<i4", nobs),
                 key_points=get("key_points", "<i4", 5 * nk), kf_nfts=get("kf_nfts", "<i4", nk), cands=get("cands", "<i4", nc),
                 ff=[get("frame_feats0", SEQ_FEATURE, int(sz[5])), get("frame_feats1", SEQ_FEATURE, int(sz[6]))], ff_frame=[int(sz[7]), int(sz[8])], ff_newest=int(sz[9]))
        lists = get("kf_fts", "<i4", nk * cap).reshape(nk, cap) if nk else np.zeros((0, 0), "<i4")
        D["kf_fts"] = [lists[r, :D["kf_nfts"][r]].copy() for r in range(nk)]
        return D

    def run(self, flags=None, want_debug=1):
        """one hso_gpu_seq_chain call of the state's job -> everything the call produced"""
        L, ctx, S = self.lib.L, self.ctx, self.S
        job = S["job"].copy()
        job["map"] = self.map
        job["seed_group"] = -1
        if flags is not None:
            job["flags"] = flags
        cfg = S["cfg"].copy()
        order = np.ascontiguousarray(S["cell_order"], np.int32)
        cfg["cell_order"] = order.ctypes.data
        cfg["seed_table"] = -1; cfg["seed_brief_out"] = 0; cfg["seed_brief_cap"] = 0; cfg["want_debug"] = want_debug
        temps = np.ascontiguousarray(S["temps"], np.int32)
        res = np.zeros(1, SEQ_RESULT)
        self.lib.check(ctx, L.hso_gpu_seq_chain(ctx, C.byref(S["cam"]), _p(cfg), _p(job), 1, _p(temps) if len(temps) else None, len(temps), _p(res)), "seq_chain")
        r = res[0]
        out = dict(result=r)
        ev = np.zeros(max(int(r["n_events"]), 1), np.int32)
        n = self.lib.check(ctx, L.hso_gpu_seq_events(ctx, 0, _p(ev), len(ev)), "seq_events") if r["n_events"] > 0 else 0
        out["events"] = ev[:n].copy()
        ids = np.zeros(max(int(r["n_listed"]), 1), np.int32); q = np.zeros(max(int(r["n_listed"]), 1), np.uint8)
        n = self.lib.check(ctx, L.hso_gpu_seq_debug_list(ctx, 0, _p(ids), _p(q), len(ids)), "seq_debug_list") if r["n_listed"] > 0 else 0
        out["list_ids"], out["list_quality"] = ids[:n].copy(), q[:n].copy()
        n_ref = 0 if (int(job["flags"][0]) & SEQ_NO_TRACK) else int(job["n_ref_feats"][0])
        tab = np.zeros(max(n_ref, 1), REF_FEAT)
        n = self.lib.check(ctx, L.hso_gpu_seq_debug_ref_table(ctx, 0, _p(tab), len(tab)), "seq_debug_ref_table") if n_ref > 0 else 0
        out["ref_table"] = tab[:n].copy()
        cap = max(int(cfg["max_fts"][0]), 1)
        ff = np.zeros(cap, SEQ_FEATURE); n_out = np.zeros(1, np.int32)
        maps = np.array([self.map], np.int32); fid = np.array([int(job["cur_frame_id"][0])], np.int64)
        self.lib.check(ctx, L.hso_gpu_seq_frame_features(ctx, _p(maps), _p(fid), 1, _p(ff), cap, _p(n_out)), "seq_frame_features")
        out["features"] = ff[:int(n_out[0])].copy()
        out["after"] = self.dump()
        return out

    def run_ba(self, core, fixed, n_iter, error_multiplier2, chi2_corner, chi2_edgelet, cull_cap=None, point_cap=None):
        """one hso_gpu_seq_local_ba call on this state -> result record, the window's points / state / culled observations, the window
        as the library assembled it (hso_gpu_seq_ba_debug_window) and the map afterwards"""
        L, ctx, S = self.lib.L, self.ctx, self.S
        core = np.asarray(core, np.int32)
        job = np.zeros(1, SEQ_BA_JOB)
        job["map"] = self.map; job["n_core"] = len(core); job["core"][0, :len(core)] = core; job["fixed"][0, :len(core)] = np.asarray(fixed, np.uint8)
        job["n_iter"] = n_iter
        pcap = int(point_cap if point_cap is not None else min(len(S["points"]), sum(len(S["kf_fts"][r]) for r in core)))
        ccap = int(cull_cap if cull_cap is not None else len(S["obs"]))
        ids = np.full(max(pcap, 1), -1, np.int32); state = np.zeros(4 * max(pcap, 1)); culled = np.full(max(ccap, 1), -1, np.int32)
        job["point_cap"] = pcap; job["cull_cap"] = ccap
        job["point_ids"] = ids.ctypes.data; job["point_state"] = state.ctypes.data; job["culled"] = culled.ctypes.data
        res = np.zeros(1, SEQ_BA_RESULT)
        self.lib.check(ctx, L.hso_gpu_seq_local_ba(ctx, _p(job), 1, float(error_multiplier2), float(chi2_corner), float(chi2_edgelet), _p(res)), "seq_local_ba")
        r = res[0]
        n_pts, n_cull = int(r["n_points"]), int(r["n_culled"].sum())
        out = dict(result=r, point_ids=ids[:n_pts].copy(), point_state=state[:4 * n_pts].reshape(-1, 4).copy(), culled=culled[:min(n_cull, ccap)].copy())
        sz = np.zeros(4, np.int32)
        self.lib.check(ctx, L.hso_gpu_seq_ba_debug_window(ctx, 0, BAW["sizes"], _p(sz), sz.nbytes), "ba window sizes")
        assert (int(sz[0]), int(sz[1]), int(sz[2]), int(sz[3])) == (int(r["n_poses"]), n_pts, int(r["n_edges"]), int(r["status"]))
        npo, ne = int(sz[0]), int(sz[2]) if int(sz[3]) == 0 else 0

        def get(what, dtype, n):
            a = np.zeros(n, dtype)
            self.lib.check(ctx, L.hso_gpu_seq_ba_debug_window(ctx, 0, BAW[what], _p(a) if a.nbytes else _p(np.zeros(1)), a.nbytes), "ba window " + what)
            return a
        W = dict(vertex_rows=get("vertex_rows", "<i4", npo), fixed=get("fixed", "u1", npo), poses_in=get("poses_in", SE3, npo))
        if int(sz[3]) == 0:
            W.update(edges=get("edges", capi.BA_EDGE_DTYPE, ne), obs_uv=get("obs_uv", "<f8", 2 * ne), edge_obs=get("edge_obs", "<i4", ne), edge_chi2=get("edge_chi2", "<f8", ne),
                     poses_out=get("poses_out", SE3, npo), idist_in=get("idist_in", "<f8", n_pts))
        out["window"] = W
        out["after"] = self.dump()
        return out

    def close(self):
        if self.ctx:
            self.lib.L.hso_gpu_destroy(self.ctx)
            self.ctx = C.c_void_p()
