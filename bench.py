#!/usr/bin/env python
"""bench.py — frames/sec of the MI355X-native HSO per-frame hot path.

One "step" = one pass of the hot path over one batch of synthetic input, per GPU:
B independent (reference, current) frame pairs — 640x480 8-bit images, 2000 sparse
points each (BASELINE.json configs[1]) — go through
  Frame construction of the current frame  (5-level pyramid + Sobel-5 + frame statistics)
  CoarseTracker::run                        (levels 4..1, device-resident LM loop)
with the level-0 images, the reference frames and the feature tables already resident
in HBM when the timed region starts.  value = frames / second over all GPUs.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), B pairs per rank
(weak scaling: independent sequences shard with no data-path collective); the only
exchange is one all_gather of the per-frame result records (pose + timing) at the end of
the timed region, as BASELINE.json's north_star prescribes.

The JSON line also carries
  roofline     — dominant kernel (k_track): algorithmic bytes per launch (SURVEY.md §8(d)
                 B_frame summed over the batch, with the evaluation counts the kernel
                 reports) / mean launch duration from HIP events on the launch stream;
  cpu_baseline — the CPU restatement (oracle/, single thread) timed on the same frames on
                 this box's host cores, bounded sample (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PA = {4: 9, 3: 13, 2: 13, 1: 21, 0: 25}      # include/hso/CoarseTracker.h:100-109 via :80
PAD = {4: 1, 3: 2, 2: 2, 1: 3, 0: 2}         # CoarseTracker.h:111-120
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(results, n_valid, inverse, levels):
    """B_frame = sum_L [ n_eval(L) * B_alg(N,L) + B_pre(N,L) + B_sel(N,L) ]."""
    total = 0
    for r in results:
        for L in levels:
            pa, pad = PA[L], PAD[L]
            u_fwd, u_ic = (2 * pad + 4) ** 2, (2 * pad + 2) ** 2
            if inverse:
                b_alg = n_valid * (32 + 28 * pa + u_ic)
                b_pre = n_valid * (16 + u_fwd + 4 * pa + 24 * pa)
            else:
                b_alg = n_valid * (32 + 4 * pa + u_fwd)
                b_pre = n_valid * (16 + u_ic + 4 * pa)
            b_sel = n_valid * (32 + 4 * pa + u_ic)
            total += r.n_eval[L] * b_alg + b_pre + b_sel
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="frame pairs per GPU per step (16 GB of resident frames at 4096)")
    ap.add_argument("--feats", type=int, default=2000)
    ap.add_argument("--pairs", type=int, default=8, help="distinct synthetic pairs rendered per rank")
    ap.add_argument("--inverse", type=int, default=0)
    ap.add_argument("--cpu-frames", type=int, default=600, help="frames in the cpu_baseline sample (about 10 s on one host core)")
    ap.add_argument("--shape", choices=["vga", "euroc"], default="vga",
                    help="vga: BASELINE configs[1] (640x480 pinhole, the default and the judged line); "
                         "euroc: the same workload on EuRoC-shaped 752x480 frames with the radtan camera")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hso_amd import capi, synth
    from hso_amd import dist as hdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    stream = torch.cuda.Stream()
    spec = synth.EUROC if args.shape == "euroc" else synth.ICL_NUIM
    B, W, H = args.batch, spec["width"], spec["height"]
    cam = synth.camera(spec)
    params = capi.TrackParams(args.inverse, 4, 1, 50)   # frame_handler_mono.cpp:190,203
    levels = (4, 3, 2, 1)

    # ---- synthetic input: `pairs` distinct scenes per rank, replicated into B distinct resident frames
    pairs = [synth.config2_pair(args.feats, spec=spec, seed=1234 + 100 * rank + 7 * k) for k in range(args.pairs)]
    with torch.cuda.stream(stream):
        ctx = capi.Context(local_rank, stream.cuda_stream)
        ref_ids = list(range(0, B))
        cur_ids = list(range(B, 2 * B))
        st_ref = ctx.frame_upload_batch(ref_ids, imgs=[pairs[i % len(pairs)]["ref"] for i in range(B)])
        cur_dev = [torch.from_numpy(pairs[i % len(pairs)]["cur"].copy()).cuda() for i in range(B)]
        cur_ptrs = np.array([t.data_ptr() for t in cur_dev], np.uint64)
        st_cur = ctx.frame_upload_batch(cur_ids, device_ptrs=cur_ptrs, width=W, height=H)
        jobs = []
        for i in range(B):
            a0 = float(np.float32(st_cur[i].integral_image / st_ref[i].integral_image))  # CoarseTracker.cpp:60
            jobs.append(ctx.make_job(ref_ids[i], cur_ids[i], pairs[i % len(pairs)]["feats"], capi.SE3.identity(), a0))
        ctx.coarse_track_prepare(cam, params, jobs)

        def step(ev=None):
            ctx.frame_upload_batch(cur_ids, device_ptrs=cur_ptrs, width=W, height=H, want_stats=False)
            if ev is not None:
                ev[0].record(stream)
            ctx.coarse_track_launch()
            if ev is not None:
                ev[1].record(stream)

        for _ in range(args.warmup):
            step()
        stream.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(events[k])
        results = ctx.coarse_track_collect()       # synchronises the stream
        rec = hdist.pack_records(results)
        # the path's only exchange: gather every rank's per-frame records (RCCL all_gather)
        allrec = hdist.gather_records(rec, device=torch.device("cuda", local_rank))
        assert allrec.shape == (world, B, 8)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()

    elapsed = hdist.max_over_ranks(t1 - t0, device=torch.device("cuda", local_rank))

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    n_valid = int((pairs[0]["feats"]["dist"] >= 0).sum())
    bytes_launch = algorithmic_bytes(results, n_valid, bool(args.inverse), levels)
    achieved = bytes_launch / (kern_ms * 1e-3) / 1e9
    evals = float(np.mean([sum(r.n_eval[L] for L in levels) for r in results]))
    # HBM-side bytes of the same kernel from the PMC passes kept under profiles/ (collected by
    # profiles/collect_r1.sh with rocprofv3 --pmc, separate runs); valid for the profiled shape only
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_k_track.json")))
        if pmc["batch"] == B and pmc["feats"] == args.feats and not args.inverse:
            traffic = (pmc["fetch_size_kb"] + pmc["write_size_kb"]) * 1024.0
    except (OSError, KeyError, ValueError):
        pass

    # sanity: every frame converged to its scene's motion (guards against timing a broken run)
    for i in (0, B // 2, B - 1):
        t_true = pairs[i % len(pairs)]["t_true"]
        assert np.linalg.norm(rec[i, 4:7] - t_true) < 5e-3, "tracking diverged in the benchmark"

    out = {
        "metric": "frames/sec on synthetic %dx%d 5-level pyramids, 2000 pts (CoarseTracker + frame build)" % (W, H),
        "value": B * world * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 residuals / f64 geometry", "data": "synthetic (%d distinct scenes per rank replicated to %d resident pairs)" % (len(pairs), B),
        "config": {"workload": ("BASELINE configs[1]: synthetic 640x480 5-level pyramid" if args.shape == "vga" else "EuRoC-shaped synthetic 752x480 (radtan camera) 5-level pyramid")
                   + ", %d points, CoarseTracker levels 4..1 (+ pyramid/Sobel/stats of the current frame)" % args.feats,
                   "frames_per_gpu_per_step": B, "mode": "inverse_compositional" if args.inverse else "forward",
                   "parallelism": "independent sequences, %d per GPU x %d GPU(s)" % (B, world),
                   "mean_evaluations_per_frame": evals},
        "roofline": {"bound": "hbm", "kernel": "k_track", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "launch_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_launch},
    }

    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle import oracle_py as orc   # checker / baseline only, never the product path
        orc.load()
        n_cpu = args.cpu_frames
        tc0 = time.perf_counter()
        for i in range(n_cpu):
            d = pairs[i % len(pairs)]
            cp = orc.create_pyramid(d["cur"])
            for l in range(3):
                g = orc.sobel5(cp[l])
                if l == 0:
                    stc = orc.frame_stats(cp[0], *g)
            if i < len(pairs):
                d["_rp"] = orc.create_pyramid(d["ref"])
            tr = orc.Tracker(cam, params, d["_rp"], cp, d["feats"])
            tr.run(capi.SE3.identity(), float(np.float32(stc.integral_image / st_ref[i % len(pairs)].integral_image)))
        tc1 = time.perf_counter()
        out["cpu_baseline"] = {"value": n_cpu / (tc1 - tc0), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "%d frames of the same workload (pyramid + Sobel + stats + CoarseTracker), "
                                         "oracle/ C restatement, 1 thread, %.1f s" % (n_cpu, tc1 - tc0),
                               "host_cpus": os.cpu_count()}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
