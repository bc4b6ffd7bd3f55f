#!/usr/bin/env python
"""bench.py — frames/sec of the MI355X-native HSO per-frame hot path.

One "step" = one pass of the hot path over one batch of synthetic input, per GPU:
B independent (reference, current) frame pairs — by default EuRoC-shaped 752x480 8-bit
images with the radtan camera and 2000 sparse points each, the shape BASELINE.json's
`metric` is quoted on (`--shape vga` = BASELINE configs[1], 640x480 pinhole) — go through
  Frame construction of the current frame  (5-level pyramid + Sobel-5 + frame statistics)
  CoarseTracker::run                        (levels 4..1, device-resident LM loop)
  result read-back                          (per-frame pose / exposure / iteration records, D2H)
with the raw level-0 images, the reference frames and the feature tables already resident
in HBM when the timed region starts.  value = frames / second over all GPUs.

Input variety: `--scenes` distinct synthetic scenes per rank (own texture, depth surface,
motion, feature set), each job starting from a motion-model-like prediction of its motion
(not the identity), replicated to B resident pairs; two sets of current images (the second
with an independent +-1 grey-level perturbation) alternate between steps, so no step re-reads
the images of the step before.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), B pairs per rank (weak
scaling: independent sequences shard with no data-path collective); the only exchange is one
all_gather of the per-frame result records at the end of the timed region, as BASELINE.json's
north_star prescribes.  `python bench.py --gpus N` without a launcher starts the N ranks itself.

The JSON line also carries
  roofline     — dominant kernel (k_track): algorithmic bytes per launch (SURVEY.md §8(d)
                 B_frame summed over the batch with the evaluation counts the kernel reports)
                 / mean launch duration from HIP events on the launch stream.  `bound` names
                 the roofline the fraction is priced against (HBM);
                 `limiter` says what actually limits the kernel;
                 `traffic` is the PMC-measured HBM bytes per launch for exactly this
                 shape/batch if a matching record exists under profiles/, else null;
  cpu_baseline — the CPU restatement (oracle/, single thread, rebuilt on this host with
                 -O3 -march=native for the timing) on the same frames, bounded sample;
  roofline_valu — the same kernel priced against what actually limits it, VALU issue: wave-level VALU instructions per launch
                 (SQ_INSTS_VALU of a PMC pass on exactly this workload, profiles/) / the live launch duration, against
                 1024 SIMDs x clock / 4 (one wave64 VALU instruction per four clocks and SIMD), plus the PMC pass's own
                 VALU-busy fraction;
  sequences_frames_per_s — the WHOLE per-frame chain end to end on evolving state (FrameHandlerMono::addImage: frame build, tracker,
                 reprojection + matching + grid selection + pose optimisation, local BA, depth filter) through the sequence
                 engine (libhso_host.so, hso_vo_multi_*): `--banks` engines x `--sequences` sequences of `--seq-feats` features
                 per GPU, images resident in HBM; sequences_cpu_frames_per_s = the same engine over the CPU restatement
                 (one sequence, one thread; tests/fakegpu);
  single_sequence_ms_200 / _2000 — BASELINE configs[2]/[3] are ONE sequence: ms per addImage of one sequence through the engine;
                 single_track_call_ms_2000: one tracker call (the cooperative shape splits a job across the CUs of an XCD);
  (everything else — per-bank step times, call counts, the single-sequence tables, the gathered trajectory shape — goes to
   bench_detail.json beside this file, or the path in HSO_BENCH_DETAIL: the printed line stays small)
  se3_vs_cpu   — per-frame SE(3) deviation GPU vs the strict CPU restatement on the distinct
                 scenes (rotation angle, translation, iteration-count agreement) — the second
                 half of BASELINE.json's metric; `vs_f64_energy_sum` repeats it against the restatement
                 deciding on the fp64 sum of the same fp32 energy terms (a diagnostic form, not the
                 reference's: it shows which differences come from the reference's serial fp32 sum).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PA = {4: 9, 3: 13, 2: 13, 1: 21, 0: 25}      # include/hso/CoarseTracker.h:100-109 via :80
PAD = {4: 1, 3: 2, 2: 2, 1: 3, 0: 2}         # CoarseTracker.h:111-120
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_CLOCK_GHZ = 2.4                         # MI355X_MICROARCH.md: max clock


def algorithmic_bytes(results, n_valid, inverse, levels):
    """B_frame = sum_L [ n_eval(L) * B_alg(N,L) + B_pre(N,L) + B_sel(N,L) ]."""
    total = 0
    for r, nv in zip(results, n_valid):
        for L in levels:
            pa, pad = PA[L], PAD[L]
            u_fwd, u_ic = (2 * pad + 4) ** 2, (2 * pad + 2) ** 2
            if inverse:
                b_alg = nv * (32 + 28 * pa + u_ic)
                b_pre = nv * (16 + u_fwd + 4 * pa + 24 * pa)
            else:
                b_alg = nv * (32 + 4 * pa + u_fwd)
                b_pre = nv * (16 + u_ic + 4 * pa)
            b_sel = nv * (32 + 4 * pa + u_ic)
            total += r.n_eval[L] * b_alg + b_pre + b_sel
    return total


def host_cpu_share():
    """CPUs this rank may keep busy while it prepares its input: the cgroup's quota (cpu.max) or the hardware threads, divided among
    the ranks of the node (LOCAL_WORLD_SIZE of the launcher) — 8 ranks each forking 64 renderers on a 16-CPU quota would spend the
    set-up in the scheduler's throttle."""
    n = os.cpu_count() or 2
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(float(q) / float(per)))))
    except (OSError, ValueError):
        pass
    return max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))


def _render_scene(job):
    """One distinct scene (worker process; numpy only): reference + current image, features,
    true motion and a motion-model-like initial guess."""
    from hso_amd import synth
    shape, feats, seed = job
    spec = synth.EUROC if shape == "euroc" else synth.ICL_NUIM
    rng = np.random.default_rng(seed + 77)
    d = synth.config2_pair(feats, spec=spec, seed=seed, exposure=float(rng.uniform(0.92, 1.08)),
                           trans_frac=float(rng.uniform(0.012, 0.028)), rot_deg=float(rng.uniform(0.3, 0.7)))
    # constant-velocity prediction (frame_handler_mono.cpp:176: T_f_w = motionModel * last T): the last
    # inter-frame motion, i.e. the true one up to an acceleration term
    rv = 2 * np.arctan2(np.linalg.norm(d["q_true"][:3]), d["q_true"][3]) * d["q_true"][:3] / max(np.linalg.norm(d["q_true"][:3]), 1e-12)
    rv0 = rv * rng.uniform(0.6, 1.1) + rng.normal(0, np.deg2rad(0.04), 3)
    t0 = d["t_true"] * rng.uniform(0.6, 1.1) + rng.normal(0, 0.1 * np.linalg.norm(d["t_true"]), 3)
    cur_b = np.clip(d["cur"].astype(np.int16) + rng.integers(-1, 2, d["cur"].shape), 0, 255).astype(np.uint8)
    return dict(ref=d["ref"], cur=d["cur"], cur_b=cur_b, feats=d["feats"], q_true=d["q_true"], t_true=d["t_true"],
                q_init=synth.rotvec_to_quat(rv0), t_init=t0)


def render_scenes(shape, feats, seeds):
    """Distinct scenes render in worker processes (forked before any GPU runtime exists in this process) and are cached under the
    temporary directory, keyed by their parameters: the 1 / 2 / 4 / 8-GPU runs of one node, and repeated runs on one box, render once."""
    import multiprocessing as mp
    import pickle
    import tempfile
    cache = os.path.join(tempfile.gettempdir(), "hso_bench_scenes_v4")
    os.makedirs(cache, exist_ok=True)
    out, todo = {}, []
    for s in seeds:
        fn = os.path.join(cache, "%s_%d_%d.pkl" % (shape, feats, int(s)))
        try:
            with open(fn, "rb") as f:
                out[int(s)] = pickle.load(f)
        except (OSError, EOFError, pickle.UnpicklingError):
            todo.append((shape, feats, int(s)))
    if todo:
        nproc = max(1, min(len(todo), host_cpu_share(), 64))
        if nproc == 1:
            res = [_render_scene(j) for j in todo]
        else:
            with mp.get_context("fork").Pool(nproc) as pool:
                res = pool.map(_render_scene, todo, chunksize=1)
        for j, r in zip(todo, res):
            out[j[2]] = r
            try:
                tmp = os.path.join(cache, "%s_%d_%d.pkl.%d" % (j[0], j[1], j[2], os.getpid()))
                with open(tmp, "wb") as f:
                    pickle.dump(r, f, protocol=4)
                os.replace(tmp, tmp[:tmp.rindex(".")])
            except OSError:
                pass
    return [out[int(s)] for s in seeds]


def render_sequences(n_seq, n_frames, spec, seed0):
    """The rendered sequences of the end-to-end run, cached like the scenes (the 1 / 2 / 4 / 8-GPU runs of a node render once)."""
    import pickle
    import tempfile
    from hso_amd import synth
    cache = os.path.join(tempfile.gettempdir(), "hso_bench_sequences_v1")
    os.makedirs(cache, exist_ok=True)
    fn = os.path.join(cache, "%dx%d_%d_%d_%d.pkl" % (spec["width"], spec["height"], n_seq, n_frames, seed0))
    try:
        with open(fn, "rb") as f:
            return pickle.load(f)
    except (OSError, EOFError, pickle.UnpicklingError):
        pass
    seqs = synth.sequences(n_seq, n_frames, spec=spec, seed0=seed0, workers=host_cpu_share())
    try:
        tmp = fn + ".%d" % os.getpid()
        with open(tmp, "wb") as f:
            pickle.dump(seqs, f, protocol=4)
        os.replace(tmp, fn)
    except OSError:
        pass
    return seqs


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one per GPU, RCCL) ourselves."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def bind_to_device_node(ctx):
    """numactl --cpunodebind for this process, from the inside: all its threads (the ones that exist; later ones inherit) onto the CPUs
    of the NUMA node the context's device hangs off, intersected with what the process may use.  -> description for the JSON line"""
    try:
        node = ctx.device_cpulist()
        allowed = os.sched_getaffinity(0)
        cpus = node & allowed
        if len(cpus) < 2 or cpus == allowed:
            return "unchanged (%d CPUs; the device's node has %d)" % (len(allowed), len(node))
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        return "%d CPUs of the device's NUMA node (of %d allowed)" % (len(cpus), len(allowed))
    except Exception as e:                                            # a stand-in context without the call (tests/test_bench_cpu.py)
        return "unchanged (%s)" % type(e).__name__


class GpuSide:
    """Everything bench.py needs of the device side in one place: HIP stream / events through torch, the C-ABI context, device
    copies of images, the collective backend.  tests/test_bench_cpu.py swaps it (HSO_BENCH_SIDE="module:Class") for a stand-in over
    the CPU restatement and gloo, so that the multi-rank control flow of this file — self-launch, sharding, both gathers, the JSON
    line — runs without a GPU."""
    backend = "nccl"
    engine_lib = None            # None = hso_amd/host/libhso_host.so

    def __init__(self, local_rank):
        import torch
        self.torch, self.rank = torch, local_rank
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)

    def init_group(self, rank, world):
        import torch.distributed as dist
        dist.init_process_group(self.backend, rank=rank, world_size=world, device_id=self.dev)

    def new_stream(self):
        return self.torch.cuda.Stream()

    def on(self, stream):
        return self.torch.cuda.stream(stream)

    def context(self, stream):
        from hso_amd import capi
        return capi.Context(self.rank, stream.cuda_stream)

    def to_device(self, img):
        return self.torch.from_numpy(img).cuda()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def pinned_batch(self, imgs):
        """the images back to back in ONE page-locked host buffer (what a camera driver's ring buffer is): [n, h, w] uint8"""
        t = self.torch.empty((len(imgs),) + tuple(imgs[0].shape), dtype=self.torch.uint8, pin_memory=True)
        for i, im in enumerate(imgs):
            t[i].copy_(self.torch.from_numpy(im))
        return t

    def device_batch(self, n, h, w):
        return self.torch.empty((n, h, w), dtype=self.torch.uint8, device=self.dev)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def release_cached(self):
        self.torch.cuda.empty_cache()


def make_side(local_rank):
    spec = os.environ.get("HSO_BENCH_SIDE")
    if not spec:
        return GpuSide(local_rank)
    import importlib
    mod, cls = spec.split(":")
    return getattr(importlib.import_module(mod), cls)(local_rank)


def start_pose(sc, variant):
    """The motion-model prediction a job starts from.  Variant 0 is the scene's own; 1..3 are other plausible predictions of the
    same motion (scaled / perturbed deterministically), so that replicas of a scene walk different Levenberg-Marquardt paths: the
    per-frame SE(3) comparison with the CPU restatement then covers 4 x scenes distinct (scene, start) frames."""
    if variant == 0:
        return sc["q_init"], sc["t_init"]
    rng = np.random.default_rng(9000 + 31 * variant + int(1e6 * abs(float(sc["t_true"][0]))) % 100000)
    from hso_amd import synth
    q = np.asarray(sc["q_true"], float)
    rv = 2 * np.arctan2(np.linalg.norm(q[:3]), q[3]) * q[:3] / max(np.linalg.norm(q[:3]), 1e-12)
    rv0 = rv * rng.uniform(0.5, 1.2) + rng.normal(0, np.deg2rad(0.05), 3)
    t0 = np.asarray(sc["t_true"]) * rng.uniform(0.5, 1.2) + rng.normal(0, 0.12 * np.linalg.norm(sc["t_true"]), 3)
    return synth.rotvec_to_quat(rv0), t0


def deal_features(feats, q, t, spec, level):
    """Experiment (profiles/r5_headline_levers.md): the same features in another table order — dealt round-robin over
    the 32 LDS banks of the dword their pattern's row 0 starts at in the CURRENT frame's level-`level` image (positions predicted with
    the job's start pose), so that the 32 lanes an LDS access serves mostly hit distinct banks.  The tracker sums over the features, so
    only the order of the fp32 sums changes."""
    from hso_amd import synth
    X = feats["f"] * feats["dist"][:, None]
    X = X @ synth.quat_to_R(np.asarray(q, float)).T + np.asarray(t, float)
    z = np.where(np.abs(X[:, 2]) > 1e-9, X[:, 2], 1.0)
    x, y = X[:, 0] / z, X[:, 1] / z
    d = spec.get("d")
    if d is not None:
        r2 = x * x + y * y
        rad = 1 + d[0] * r2 + d[1] * r2 * r2 + d[4] * r2 ** 3
        x, y = x * rad + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x), y * rad + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
    u = np.floor((spec["fx"] * x + spec["cx"]) / (1 << level)).astype(np.int64)
    v = np.floor((spec["fy"] * y + spec["cy"]) / (1 << level)).astype(np.int64)
    stride = spec["width"] >> level
    bank = (((v * stride + u - 3) >> 2) % 32 + 32) % 32
    buckets = [list(np.nonzero(bank == b)[0][::-1]) for b in range(32)]
    order = []
    while any(buckets):
        for b in range(32):
            if buckets[b]:
                order.append(buckets[b].pop())
    return np.ascontiguousarray(feats[np.asarray(order)])


def rot_angle(qa, qb):
    """Angle of qa * qb^-1 for unit quaternions (x, y, z, w)."""
    d = abs(float(np.dot(qa, qb)))
    return 2 * np.arccos(min(1.0, d))


def single_sequence(args, ctx, stream, spec, seq_S, cam):
    """BASELINE configs[2]/[3] are ONE sequence: latency of one tracker call and ms per addImage of one sequence through the engine."""
    from hso_amd import latency_bench
    single = {"shape": "%dx%d" % (spec["width"], spec["height"]), "track": [], "sequence": []}
    for n in (200, args.feats):
        single["track"].append(latency_bench.track_latency(ctx, stream, cam, spec, n, 20))
    if seq_S is not None:
        for n in (200, args.feats):
            single["sequence"].append(latency_bench.sequence_latency(cam, seq_S, n))
    single["what"] = ("track: ONE job per call (hso_gpu_coarse_track_batch wall time incl. table upload, launch, read-back; launch_ms "
                      "between HIP events); sequence: ms per addImage of one synthetic %d-frame sequence through libhso_host.so" % args.seq_frames)
    return single


def cpu_sequence_baseline(cam, seq_S, max_fts, n_frames):
    """The CPU path of a WHOLE evolving sequence, timed beside the GPU's: the same engine (hso_amd/host/hso_engine*.cpp) linked
    against the CPU restatement instead of libhso_gpu.so (tests/fakegpu: every numeric stage = the oracle's function, one thread) —
    frame construction, tracker, reprojection + matching + selection, pose optimisation, local BA, depth filter."""
    from hso_amd import vo
    path = os.path.join(ROOT, "tests", "fakegpu", "libhso_host_fake.so")
    if not os.path.exists(path):
        return None
    lib = vo.load_from(path)
    odo = vo.VisualOdometry(cam, max_fts, lib=lib)
    odo.set_first_frame(seq_S["images"][0], seq_S["depth0"], 0.0)
    t0 = time.perf_counter()
    n_kf = 0
    for k in range(1, n_frames):
        st = odo.add_image(seq_S["images"][k], float(k))
        n_kf += st.is_keyframe
    dt = time.perf_counter() - t0
    q, t = odo.status().T_f_w.to_arrays()
    err = float(np.linalg.norm(t - seq_S["T_f_w"][n_frames - 1][1]))
    odo.close()
    return {"value": (n_frames - 1) / dt, "unit": "frames/s", "cores": 1, "kind": "port", "ms_per_frame": 1e3 * dt / (n_frames - 1),
            "sample": "%d frames (%d keyframes) of one %d-feature sequence through the engine over oracle/ (strict build), %.1f s"
                      % (n_frames - 1, n_kf, max_fts, dt), "trans_err_last": err}


def main():
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="frame pairs per GPU per step (19 GB of resident frames at 4096 EuRoC pairs)")
    ap.add_argument("--feats", type=int, default=2000)
    ap.add_argument("--scenes", type=int, default=64, help="distinct synthetic scenes rendered per rank")
    ap.add_argument("--inverse", type=int, default=0)
    ap.add_argument("--min-level", type=int, default=1, help="developer knob: stop the tracker above level 1 (the judged line uses 1)")
    ap.add_argument("--cpu-frames", type=int, default=400, help="frames in the cpu_baseline sample (about 7 s on one host core)")
    ap.add_argument("--seq-frames", type=int, default=121, help="frames per sequence of the end-to-end runs through libhso_host.so (0 = skip): 120 steps = 8 "
                    "keyframes per sequence, i.e. full local-BA windows (7 core keyframes) and three live seed batches in the second half — the steady state")
    ap.add_argument("--seq-warmup-frames", type=int, default=121, help="frames of the untimed warm-up pass of the end-to-end engines (0 = none).  A full-length "
                    "pass: on a box whose device memory has not been touched since it came up, the FIRST run that grows to the full footprint (768 sequences: "
                    "~30 GB of frames and maps) runs at 60-70 %% of every later one, whatever ran before it at a smaller footprint — a 30-step pass did not "
                    "always cover that (profiles/r6_engine_host.md section 8)")
    ap.add_argument("--sequences", type=int, default=128, help="sequences per engine (bank) of the end-to-end run (hso_vo_multi_*; 0 = skip)")
    ap.add_argument("--banks", type=int, default=6, help="engines per GPU, each on its own host thread and stream")
    ap.add_argument("--seq-feats", type=int, default=2000, help="Config::maxFts() of the end-to-end run")
    ap.add_argument("--seq-distinct", type=int, default=8, help="distinct rendered sequences per rank (replicated to --sequences x --banks)")
    ap.add_argument("--single", type=int, default=1, help="0: skip the single-sequence latency section (N = 1 only)")
    ap.add_argument("--se3-frames", type=int, default=256, help="frames of the per-frame SE(3) comparison with the CPU restatement")
    ap.add_argument("--native-gather", type=int, default=-1,
                    help="1: repeat the trajectory gather through libhso_gather.so (ncclAllGather from C, include/hso_vo.h) and "
                         "require it to equal the torch.distributed one; reported in bench_detail.json.  -1 (default): on when N > 1")
    ap.add_argument("--deal-features", type=int, default=0, help="experiment: > 0 = reorder every job's features by LDS bank at this pyramid level (deal_features)")
    ap.add_argument("--overlap-readback", type=int, default=1, help="1: step k's result read-back is waited for after step k + 1 is enqueued (collect_begin / _end); 0: the synchronous collect")
    ap.add_argument("--ic-steps", type=int, default=3, help="tracker launches in inverse-compositional mode on the same pairs, reported as tracker_inverse_mode_launch_ms (0 = skip)")
    ap.add_argument("--numa-bind", type=int, default=1, help="1: every thread of the rank's process is confined to the CPUs of the NUMA node its GPU is attached to "
                    "(what `numactl --cpunodebind` does for a deployment; the engines do it for their own threads anyway: hso_vo_options.no_numa_pin). "
                    "Reported as host_affinity.  The timed headline region has no host work in it; the end-to-end figures do")
    ap.add_argument("--h2d", type=int, default=1, help="1: also time the headline steps and the end-to-end run with the image upload (page-locked host memory -> HBM) inside the timed region")
    ap.add_argument("--shape", choices=["euroc", "vga"], default="euroc",
                    help="euroc (default, the judged line): EuRoC-shaped 752x480 frames, radtan camera — the shape "
                         "BASELINE.json's metric is quoted on; vga: BASELINE configs[1], 640x480 pinhole")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    from hso_amd import dist as hdist
    # ---- synthetic input (before the GPU runtime is touched: the renderers fork)
    n_sc = max(1, min(args.scenes, args.batch))
    seq_ids = hdist.shard_sequences(world * n_sc, rank, world)      # distinct sequences per rank
    t_r0 = time.perf_counter()
    from hso_amd import synth as _synth
    spec0 = _synth.EUROC if args.shape == "euroc" else _synth.ICL_NUIM
    extras = rank == 0 and world == 1
    scenes = render_scenes(args.shape, args.feats, [1234 + 7 * s for s in seq_ids])
    # sequences for the end-to-end engine: `--seq-distinct` rendered per rank (replicated to banks x sequences) at every N — their
    # trajectories are what the ranks gather; the first one also serves the single-sequence latency and the CPU sequence baseline
    seq_list = (render_sequences(max(1, min(args.seq_distinct, args.sequences)), args.seq_frames, spec0, 2024 + 1000 * rank)
                if args.seq_frames > 1 and args.sequences > 0 else [])
    seq_S = seq_list[0] if extras and seq_list else None
    t_render = time.perf_counter() - t_r0

    import torch
    import torch.distributed as dist
    from hso_amd import capi, synth

    side = make_side(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        side.init_group(rank, world)
    dev = side.dev

    stream = side.new_stream()
    spec = synth.EUROC if args.shape == "euroc" else synth.ICL_NUIM
    B, W, H = args.batch, spec["width"], spec["height"]
    cam = synth.camera(spec)
    params = capi.TrackParams(args.inverse, 4, args.min_level, 50)   # frame_handler_mono.cpp:190,203
    levels = tuple(range(4, args.min_level - 1, -1))

    with side.on(stream):
        ctx = side.context(stream)
        numa = bind_to_device_node(ctx) if args.numa_bind else None
        ref_ids = list(range(0, B))
        cur_ids = list(range(B, 2 * B))
        st_ref = ctx.frame_upload_batch(ref_ids, imgs=[scenes[i % n_sc]["ref"] for i in range(B)])
        # raw level-0 images of the two alternating current-frame sets, resident in HBM
        cur_dev = [[side.to_device(scenes[i % n_sc][k]) for i in range(B)] for k in ("cur", "cur_b")]
        cur_ptrs = [np.array([t.data_ptr() for t in cd], np.uint64) for cd in cur_dev]
        st_cur = ctx.frame_upload_batch(cur_ids, device_ptrs=cur_ptrs[0], width=W, height=H)
        jobs, a0s, starts = [], [], []
        for i in range(B):
            sc = scenes[i % n_sc]
            a0 = float(np.float32(st_cur[i].integral_image / st_ref[i].integral_image))  # CoarseTracker.cpp:60
            a0s.append(a0)
            starts.append(start_pose(sc, (i // n_sc) % 4))
            fts = deal_features(sc["feats"], starts[i][0], starts[i][1], spec, args.deal_features) if args.deal_features > 0 else sc["feats"]
            jobs.append(ctx.make_job(ref_ids[i], cur_ids[i], fts, capi.SE3.from_arrays(*starts[i]), a0))
        ctx.coarse_track_prepare(cam, params, jobs)

        overlap = bool(args.overlap_readback) and hasattr(ctx, "coarse_track_collect_begin")

        def enqueue(k, ev=None):
            ctx.frame_upload_batch(cur_ids, device_ptrs=cur_ptrs[k & 1], width=W, height=H, want_stats=False)
            if ev is not None:
                ev[0].record(stream)
            ctx.coarse_track_launch()
            if ev is not None:
                ev[1].record(stream)

        def step(k, ev=None):
            enqueue(k, ev)
            return ctx.coarse_track_collect(as_list=False)   # synchronises the stream, D2H of the result records

        for k in range(args.warmup):
            step(k)
        stream.synchronize()
        if world > 1:
            dist.barrier()
        side.synchronize()
        events = [(side.event(), side.event())
                  for _ in range(args.steps)]
        res_set = [None, None]
        # the interpreter's cyclic collector would otherwise walk the rendered scenes (millions of objects) somewhere inside
        # the timed loop: a 40 ms stall that has nothing to do with the path being measured
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        if overlap:
            # the read-back of step k is waited for after step k + 1 has been enqueued (hso_gpu_coarse_track_collect_begin / _end): the
            # stream never runs dry between steps; every step's records still reach the host inside the timed region
            for k in range(args.steps):
                enqueue(k, events[k])
                ctx.coarse_track_collect_begin()
                if k > 0:
                    res_set[(k - 1) & 1] = ctx.coarse_track_collect_end(as_list=False)
            res_set[(args.steps - 1) & 1] = ctx.coarse_track_collect_end(as_list=False)
            stream.synchronize()
        else:
            for k in range(args.steps):
                res_set[k & 1] = step(k, events[k])
        results = res_set[(args.steps - 1) & 1]
        rec = hdist.pack_records(results)
        # the path's only exchange: gather every rank's per-frame records (RCCL all_gather)
        tg0 = time.perf_counter()
        allrec = hdist.gather_records(rec, device=dev)
        assert allrec.shape == (world, B, 8)
        side.synchronize()
        t_gather = time.perf_counter() - tg0
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()

        # ---- the same steps with the image upload inside the timed region (FrameHandlerMono::addImage takes a HOST image,
        # src/frame_handler_mono.cpp:80-123, src/frame.cpp:82-96): the level-0 images of a step lie in page-locked host memory and
        # cross PCIe on a copy stream, double-buffered against the kernels — copy k + 1 runs beside the frame build + tracker of step k
        h2d = None
        if args.h2d and hasattr(side, "pinned_batch"):
            host_sets = [side.pinned_batch([scenes[i % n_sc][k] for i in range(B)]) for k in ("cur", "cur_b")]
            dev_in = [side.device_batch(B, H, W) for _ in range(2)]
            in_ptrs = [np.array([t.data_ptr() + i * W * H for i in range(B)], np.uint64) for t in dev_in]
            copy_stream = side.new_stream()
            ev_copied = [side.event(), side.event()]
            ev_built = [side.event(), side.event()]

            def copy_in(k):
                with side.on(copy_stream):
                    if k >= 2:
                        copy_stream.wait_event(ev_built[k & 1])      # the frame build of step k - 2 has read this buffer
                    dev_in[k & 1].copy_(host_sets[k & 1], non_blocking=True)
                    ev_copied[k & 1].record(copy_stream)

            def enqueue_h2d(k):
                stream.wait_event(ev_copied[k & 1])
                ctx.frame_upload_batch(cur_ids, device_ptrs=in_ptrs[k & 1], width=W, height=H, want_stats=False)
                ev_built[k & 1].record(stream)
                ctx.coarse_track_launch()

            n_h = args.steps
            copy_in(0)
            for k in range(min(2, args.warmup)):                      # warm: both buffers touched, both events recorded once
                copy_in(k + 1) if k + 1 < 2 else None
                enqueue_h2d(k)
                ctx.coarse_track_collect(as_list=False)
            side.synchronize()
            if world > 1:
                dist.barrier()
            th0 = time.perf_counter()
            copy_in(0)
            for k in range(n_h):
                if k + 1 < n_h:
                    copy_in(k + 1)
                enqueue_h2d(k)
                if overlap:
                    ctx.coarse_track_collect_begin()
                    if k > 0:
                        ctx.coarse_track_collect_end(as_list=False)
                else:
                    ctx.coarse_track_collect(as_list=False)
            if overlap:
                ctx.coarse_track_collect_end(as_list=False)
            stream.synchronize()
            side.synchronize()
            if world > 1:
                dist.barrier()
            th1 = time.perf_counter()
            el_h = hdist.max_over_ranks(th1 - th0, device=dev)
            h2d = {"value_with_h2d": B * world * n_h / el_h, "ms_per_step": 1e3 * el_h / n_h, "pcie_gb_per_s_per_gpu": B * W * H * n_h / (th1 - th0) / 1e9,
                   "bytes_per_frame": W * H, "what": "the headline's steps with every step's %d level-0 images copied from page-locked host memory on a copy stream "
                                                     "inside the timed region, double-buffered against frame build + tracker" % B}
            del host_sets, dev_in

        # ---- the tracker's other mode on the same resident pairs (untimed for `value`): the reference runs CoarseTracker in
        # inverse-compositional mode whenever the new frame's gradient mean does not exceed the last frame's by 0.5
        # (src/frame_handler_mono.cpp:184) — the mode the end-to-end engines spend their tracker time in
        ic_ms = None
        if not args.inverse and args.ic_steps > 0:
            ctx.coarse_track_prepare(cam, capi.TrackParams(1, 4, args.min_level, 50), jobs)
            ev_ic = [(side.event(), side.event()) for _ in range(args.ic_steps + 1)]
            for k in range(args.ic_steps + 1):
                ev_ic[k][0].record(stream)
                ctx.coarse_track_launch()
                ev_ic[k][1].record(stream)
                ctx.coarse_track_collect(as_list=False)
            ic_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_ic[1:]]))

    elapsed = hdist.max_over_ranks(t1 - t0, device=dev)
    local_fps = B * args.steps / (t1 - t0)
    per_gpu = [local_fps]
    if world > 1:
        tl = torch.tensor([local_fps], dtype=torch.float64, device=dev)
        out_l = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(out_l, tl)
        per_gpu = [float(x.item()) for x in out_l]

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    n_valid = [int((scenes[i % n_sc]["feats"]["dist"] >= 0).sum()) for i in range(B)]
    # both image sets contribute launches: average the algorithmic bytes of the sets that ran
    sets_run = [r for r in res_set if r is not None]
    bytes_launch = float(np.mean([algorithmic_bytes(r, n_valid, bool(args.inverse), levels) for r in sets_run]))
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
    evals = float(np.mean([sum(r.n_eval[L] for L in levels) for r in results]))

    # HBM-side bytes of the same kernel: only a PMC record collected for exactly this workload counts
    traffic, traffic_src, valu = None, None, None
    for fn in ("r6_pmc_k_track.json", "r5_pmc_k_track.json", "r4_pmc_k_track.json", "r3_pmc_k_track.json", "r2_pmc_k_track.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
            for e in pmc["records"]:
                if (e["shape"], e["batch"], e["feats"], e["inverse"], e["scenes"], e.get("min_level", 1)) == \
                        (args.shape, B, args.feats, args.inverse, n_sc, args.min_level):
                    if traffic is None and e.get("hbm_bytes_per_launch"):
                        traffic, traffic_src = e["hbm_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc, separate passes)" % fn
                    if valu is None and e.get("valu_insts_per_launch"):
                        valu = dict(e, source="profiles/%s" % fn)
        except (OSError, KeyError, ValueError):
            pass

    # sanity: every frame converged to its scene's motion (guards against timing a broken run)
    terr = [float(np.linalg.norm(rec[i, 4:7] - scenes[i % n_sc]["t_true"])) for i in range(min(B, 4 * n_sc))]
    assert max(terr) < 1e-2, "tracking diverged in the benchmark (%.3g)" % max(terr)

    shape_txt = ("EuRoC-shaped synthetic 752x480 (radtan camera, test/cameras/euroc.txt) 5-level pyramid" if args.shape == "euroc"
                 else "BASELINE configs[1]: synthetic 640x480 5-level pyramid")
    out = {
        "metric": "frames/sec on %s %dx%d, %d pts; per-frame SE(3) vs CPU ref" % ("EuRoC-shaped" if args.shape == "euroc" else "synthetic", W, H, args.feats),
        "value": B * world * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 residuals / f64 geometry",
        "data": "synthetic (%d distinct scenes per rank, motion-model initial poses, replicated to %d resident pairs; "
                "two alternating current-image sets); `value` and `sequences_frames_per_s` start from images ALREADY IN HBM (device-resident kernel "
                "rates, not FrameHandlerMono::addImage rates); `value_with_h2d` and `sequences_with_h2d_frames_per_s` include the copy of every "
                "frame's level-0 image from page-locked host memory over PCIe" % (n_sc, B),
        "config": {"workload": shape_txt + ", %d points, frame build (pyramid/Sobel/stats) + CoarseTracker levels 4..1 + result read-back" % args.feats,
                   "shape": args.shape, "min_level": args.min_level, "frames_per_gpu_per_step": B, "mode": "inverse_compositional" if args.inverse else "forward",
                   "parallelism": "independent sequences, %d per GPU x %d GPU(s)" % (B, world),
                   "mean_evaluations_per_frame": evals, "distinct_scenes_per_rank": n_sc},
        "per_gpu_frames_per_s": per_gpu, "gather_ms": 1e3 * t_gather, "setup_render_s": t_render,
        "roofline": {"bound": "hbm", "kernel": "k_track", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "limiter": "VALU issue + serial per-job phases (thresholds, LM solve); taps are LDS-served, HBM traffic is below "
                                "the algorithmic bytes (DESIGN.md section 3.2)",
                     "launch_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch},
    }
    if ic_ms is not None:
        out["tracker_inverse_mode_launch_ms"] = ic_ms     # beside roofline.launch_ms (forward mode): same pairs, same poses
    if h2d:
        out["value_with_h2d"] = h2d["value_with_h2d"]
        out["h2d_pcie_gb_per_s_per_gpu"] = h2d["pcie_gb_per_s_per_gpu"]
        out["h2d_ms_per_step"] = h2d["ms_per_step"]
    # the limiter itself: VALU issue.  Instructions from a PMC pass on exactly this workload (SQ_INSTS_VALU counts wave-level
    # instructions, summed over the tracker's launches of one batch); time = the live launch duration above.
    if valu is not None:
        peak_ginst = 1024 * VALU_CLOCK_GHZ / 4.0          # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 clocks each
        ach = valu["valu_insts_per_launch"] / (kern_ms * 1e-3) / 1e9
        out["roofline_valu"] = {"bound": "valu_issue", "kernel": "k_track", "achieved": ach, "peak": peak_ginst, "unit": "G wave-instructions/s",
                                "frac": ach / peak_ginst, "valu_insts_per_launch": valu["valu_insts_per_launch"],
                                "valu_busy_frac_pmc": valu.get("valu_busy_frac"), "clock_ghz": VALU_CLOCK_GHZ, "source": valu["source"],
                                "note": "peak = 1024 SIMDs x 2.4 GHz / 4 clocks per wave64 instruction (MI355X_MICROARCH.md max clock; the "
                                        "sustained clock under this kernel is lower, so frac is a lower bound); valu_busy_frac_pmc = "
                                        "4 x SQ_ACTIVE_INST_VALU / (SIMDs x busy cycles) of the PMC pass, time-weighted over the two launches"}

    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import oracle_py as orc   # checker / baseline only, never the product path
        # (1) per-frame SE(3) GPU vs the strict restatement (the parity build) on the distinct scenes
        orc.load()
        res_a = res_set[0] if res_set[0] is not None else results
        rot, tr, it_eq, acc_eq = [], [], 0, 0
        rot64, tr64, acc_eq64, self_eq64 = [], [], 0, 0
        pyr = {}
        n_cmp = min(4 * n_sc, B, max(1, args.se3_frames))
        for i in range(n_cmp):
            d = scenes[i % n_sc]
            if i % n_sc not in pyr:
                pyr[i % n_sc] = (orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"]))
            P = pyr[i % n_sc]
            T0 = capi.SE3.from_arrays(*starts[i])
            ro = orc.Tracker(cam, params, P[0], P[1], d["feats"]).run(T0, a0s[i])
            qg, tg = res_a[i].T_cur_ref.to_arrays(); qo, to = ro.T_cur_ref.to_arrays()
            rot.append(rot_angle(qg, qo)); tr.append(float(np.linalg.norm(tg - to)))
            it_eq += int(list(res_a[i].iters) == list(ro.iters))
            acc_eq += int(list(res_a[i].accept_mask) == list(ro.accept_mask))
            # diagnostics: the same restatement deciding on the fp64 sum of the same fp32 energy terms (not the reference's
            # behaviour): separates "the device decides differently" from "the reference's serial fp32 sum decided by its rounding"
            t64 = orc.Tracker(cam, params, P[0], P[1], d["feats"]); t64.decide_on_f64_sum(True)
            r64 = t64.run(T0, a0s[i])
            q6, t6 = r64.T_cur_ref.to_arrays()
            rot64.append(rot_angle(qg, q6)); tr64.append(float(np.linalg.norm(tg - t6)))
            acc_eq64 += int(list(res_a[i].accept_mask) == list(r64.accept_mask) and list(res_a[i].iters) == list(r64.iters))
            self_eq64 += int(list(ro.accept_mask) == list(r64.accept_mask) and list(ro.iters) == list(r64.iters))
        out["se3_vs_cpu"] = {"frames": n_cmp, "rot_rad_max": max(rot), "trans_max": max(tr),
                             "rot_rad_mean": float(np.mean(rot)), "trans_mean": float(np.mean(tr)),
                             "iters_equal_frac": it_eq / n_cmp, "accept_sequence_equal_frac": acc_eq / n_cmp,
                             "vs_f64_energy_sum": {"accept_sequence_equal_frac": acc_eq64 / n_cmp, "rot_rad_max": max(rot64),
                                                   "trans_max": max(tr64), "cpu_serial_vs_f64_equal_frac": self_eq64 / n_cmp,
                                                   "note": "diagnostics: CPU restatement deciding on the fp64 sum of the same fp32 "
                                                           "energy terms instead of the reference's serial fp32 sum"},
                             "cpu": "oracle/ strict build (-O3 -ffp-contract=off), scene depth 2..6 m",
                             "trans_err_vs_truth_max": max(terr)}
        # (2) timing: the same restatement rebuilt for this host (-O3 -march=native, contraction on)
        flags = orc.use_native_build()
        n_cpu = args.cpu_frames
        tc0 = time.perf_counter()
        for i in range(n_cpu):
            k = i % min(n_cmp, n_sc)
            d = scenes[k]
            cp = orc.create_pyramid(d["cur"])
            for l in range(3):
                g = orc.sobel5(cp[l])
                if l == 0:
                    stc = orc.frame_stats(cp[0], *g)
            tr_ = orc.Tracker(cam, params, pyr[k][0], cp, d["feats"])
            tr_.run(capi.SE3.from_arrays(d["q_init"], d["t_init"]),
                    float(np.float32(stc.integral_image / st_ref[k].integral_image)))
        tc1 = time.perf_counter()
        out["cpu_baseline"] = {"value": n_cpu / (tc1 - tc0), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames of the same workload (pyramid + Sobel + stats + CoarseTracker from the same "
                                         "initial poses), oracle/ C restatement, 1 thread, %.1f s" % (n_cpu, tc1 - tc0),
                               "build": flags, "host_cpus": os.cpu_count()}
    # ---- the end-to-end engine at every N: `--banks` engines x `--sequences` sequences per rank (hso_vo_multi_*: every numeric stage
    # of a step one batched C-ABI call for a bank's sequences; the banks on their own threads and streams), images resident in
    # HBM; the ranks' per-frame trajectories are gathered over RCCL — the path's only exchange (BASELINE north_star / configs[4])
    detail_extra = {}
    if seq_list:
        from hso_amd import bank_bench
        ctx.close()                                  # the headline's 4096 resident pairs leave HBM before the engines start
        del cur_dev
        side.release_cached()
        if world > 1:
            dist.barrier()
        # an untimed pass of the same engines over the first frames (like --warmup of the headline): on a fresh box the first engines of
        # a process page the libraries' code in and load every kernel of the keyframe path for the first time — the first of four
        # identical runs in one process measured 18.5 k frames/s steady, the next three 22-25 k (tools/r6_repeat_banks.py, profiles/r6_*)
        t_w0 = time.perf_counter()
        if args.seq_warmup_frames > 1:
            bank_bench.run_banks(args.banks, args.sequences, min(args.seq_warmup_frames, args.seq_frames), args.seq_feats, spec=spec, device=local_rank,
                                 seqs=seq_list, lib_path=side.engine_lib, to_device=side.to_device)
        t_seq_warm = time.perf_counter() - t_w0
        if world > 1:
            dist.barrier()
        mres, traj = bank_bench.run_banks(args.banks, args.sequences, args.seq_frames, args.seq_feats, spec=spec, device=local_rank, seqs=seq_list,
                                          want_traj=True, lib_path=side.engine_lib, to_device=side.to_device)
        mres_h = None
        if args.h2d and torch.cuda.is_available():
            # the same run with every frame's image handed over in page-locked host memory (the copy is part of the step)
            mres_h = bank_bench.run_banks(args.banks, args.sequences, args.seq_frames, args.seq_feats, spec=spec, device=local_rank, seqs=seq_list,
                                          lib_path=side.engine_lib, host_images=True)
        n_seq_rank = args.banks * args.sequences
        tr_rec = np.zeros((n_seq_rank, args.seq_frames, 8))
        for q, T in enumerate(traj):
            tr_rec[q, :len(T), :7] = T[:args.seq_frames]
            tr_rec[q, :len(T), 7] = 1.0
        all_tr = hdist.gather_records(tr_rec.reshape(n_seq_rank * args.seq_frames, 8), device=dev)
        assert all_tr.shape[0] == world and (world == 1 or dist.get_world_size() == world)
        native = None
        if (args.native_gather == 1 or (args.native_gather < 0 and world > 1)) and torch.cuda.is_available():
            # the same exchange through the C interface (hso_gather_*): rank 0's communicator id travels by a broadcast
            uid = [hdist.NativeGather.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            ng = hdist.NativeGather(uid[0], rank, world, local_rank)
            t0 = time.perf_counter()
            nat = ng.gather(tr_rec.reshape(n_seq_rank * args.seq_frames, 8))
            native = {"ms": 1e3 * (time.perf_counter() - t0), "equal_to_torch_gather": bool(np.array_equal(nat, all_tr)), "shape": list(nat.shape)}
            ng.close()
            assert native["equal_to_torch_gather"]
        tl = torch.tensor([mres["frames_per_s"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tl)
        detail_extra["sequences"] = dict(mres, ranks=world, sequences_total=world * n_seq_rank, frames_per_s_all_ranks=float(tl.item()),
                                   gathered_trajectory_shape=list(all_tr.shape), native_gather=native,
                                   what="end-to-end FrameHandlerMono::addImage on evolving state: frame build, tracker, reprojection + matching + grid "
                                        "selection + pose optimisation, local BA, depth filter (seed observation, activation, new seeds); %d engines x %d "
                                        "sequences per GPU, %d features, %d distinct rendered sequences replicated; images resident in HBM; "
                                        "trajectories of all ranks gathered (torch.distributed %s, world %d)"
                                        % (args.banks, args.sequences, args.seq_feats, len(seq_list), "nccl = RCCL" if world > 1 else "not initialised", world))
        # the end-to-end figure = the STEADY STATE (second half of the run: full BA windows, three seed batches alive), all ranks;
        # the whole run incl. the warm-up beside it
        ts = torch.tensor([mres.get("steady_frames_per_s") or mres["frames_per_s"]], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(ts)
        out["sequences_frames_per_s"] = float(ts.item())
        out["sequences_whole_run_frames_per_s"] = float(tl.item())
        out["sequences_warmup_frames_per_s"] = mres.get("warmup_frames_per_s")
        out["sequences_config"] = ("%d engines x %d sequences x %d features per GPU, %d frames (%.1f keyframes per sequence), end to end on evolving state; "
                                   "steady state = steps %d..%d" % (args.banks, args.sequences, args.seq_feats, args.seq_frames - 1, mres["keyframes_per_sequence"],
                                                                   mres.get("steady_from_step", 0), args.seq_frames - 1))
        if mres_h:
            th = torch.tensor([mres_h.get("steady_frames_per_s") or mres_h["frames_per_s"], mres_h["frames_per_s"]], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(th)
            out["sequences_with_h2d_frames_per_s"] = float(th[0].item())
            out["sequences_with_h2d_whole_run_frames_per_s"] = float(th[1].item())
            out["sequences_h2d_pcie_gb_per_s_per_gpu"] = float(th[0].item()) / world * spec["width"] * spec["height"] / 1e9
            detail_extra["sequences_with_h2d"] = mres_h
        out["sequences_warmup_pass_s"] = t_seq_warm
        out["sequences_failures"] = mres["failures"]
        out["sequences_roofline_frac"] = mres.get("roofline_frac_hbm")      # sum of SURVEY 8(d) algorithmic bytes of the chain's kernels / wall / 8 TB/s
        out["sequences_gpu_busy_frac"] = mres.get("steady_gpu_busy_frac")   # amdgpu gpu_busy_percent sampled every 20 ms over the steady window
        # what that figure is worth: it counts queues that are not empty; a kernel timeline of the same run has a kernel in flight for
        # about half of the time (profiles/r5_banks_overlap.json, profiles/r5_engine_host.md section 7)
        out["sequences_gpu_busy_note"] = "sysfs gpu_busy_percent (non-empty queues); kernels in flight ~0.5-0.6 of the run by rocprofv3 timeline"
        out["host_cpu_quota"] = mres.get("host_cpu_quota"); out["threads_per_bank"] = mres.get("threads_per_bank")
        out["host_affinity"] = numa
        out["host_cpus_used"] = mres.get("host_cpus_used")
    if extras and seq_S is not None and args.single:
        ctx2 = side.context(stream)
        with side.on(stream):
            single = single_sequence(args, ctx2, stream, spec, seq_S, cam)
        ctx2.close()
        detail_extra["single_sequence"] = single
        out["single_sequence_ms_200"] = single["sequence"][0]["ms_per_frame"]
        out["single_sequence_ms_2000"] = single["sequence"][1]["ms_per_frame"]
        out["single_track_call_ms_2000"] = single["track"][1]["call_ms"]
        if args.cpu_frames > 0:
            cb = cpu_sequence_baseline(cam, seq_S, args.seq_feats, min(args.seq_frames, 41))   # bounded sample: 40 frames, ~10 s on one core
            if cb:
                detail_extra["sequences_cpu_baseline"] = cb
                out["sequences_cpu_frames_per_s"] = cb["value"]
    if rank == 0:
        # the judged line stays small (the driver keeps what fits its parser); everything else goes to a side file
        try:
            detail = dict(out)
            detail.update(detail_extra)
            with open(os.environ.get("HSO_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")), "w") as f:
                json.dump(detail, f, indent=1)
        except OSError:
            pass
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
