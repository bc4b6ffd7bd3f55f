"""`python -m hso_amd.bank_bench <sequences> <frames> <max_fts> [distinct]`: the sequence engine end to end (hso_vo_multi_*: frame
construction, tracker, reprojection + matching + selection + pose, local BA, depth filter, all on one evolving state per sequence)
for many sequences in lockstep on one GPU.  `distinct` rendered sequences are replicated to `sequences` (same images = same work).
Prints one JSON line; HSO_ENGINE_TIMING=1 adds the engine's phase split on stderr."""
import os
import json
import sys
import time

import numpy as np


def run(n_seq, frames, max_fts, distinct=8, spec=None, device=0, seqs=None, on_device=True):
    """on_device: the images are copied to HBM once, before the timed steps, and handed over as device pointers (throughput with
    inputs resident in HBM, as bench.py's headline); False: host images, 361 KB of PCIe per frame inside every step."""
    from hso_amd import synth, vo
    spec = spec or synth.EUROC
    cam = synth.camera(spec)
    if seqs is None:
        seqs = synth.sequences(min(distinct, n_seq), frames, spec=spec, seed0=777)
    pick = [seqs[q % len(seqs)] for q in range(n_seq)]
    m = vo.MultiVisualOdometry(cam, n_seq, max_fts, device=device)
    m.set_first_frames([q["images"][0] for q in pick], [q["depth0"] for q in pick])
    step_ms, fails = [], 0
    dev = None
    if on_device:
        import torch
        dev = [[torch.from_numpy(np.ascontiguousarray(im)).cuda(device) for im in q["images"]] for q in seqs]
        torch.cuda.synchronize(device)
    h, w = pick[0]["images"][0].shape
    for k in range(1, frames):
        if on_device:
            ptrs = [dev[q % len(seqs)][k].data_ptr() for q in range(n_seq)]
            t0 = time.perf_counter()
            m.add_images_device(ptrs, w, h, [float(k)] * n_seq)
        else:
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
    return dict(sequences=n_seq, distinct=len(seqs), frames=frames - 1, max_fts=max_fts, images="device" if on_device else "host", frames_per_s=1e3 * n_seq / float(np.mean(warm)),
                ms_per_step_mean=float(np.mean(warm)), ms_per_step_median=float(np.median(warm)), ms_per_step_max=float(np.max(warm)),
                ms_first_steps=[round(x, 2) for x in step_ms[:3]], keyframes_per_sequence=float(np.mean(kfs)) - 1, failures=fails,
                trans_err_max=max(err), n_matches_last=int(np.mean([st.n_matches for st in sts])), n_seeds_last=int(np.mean([st.n_seeds for st in sts])),
                calls=counts)


def _cgroup_cpu():
    """(usage_usec, nr_throttled, nr_periods) of the process's cgroup (v2), or None"""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d["usage_usec"]), int(d.get("nr_throttled", 0)), int(d.get("nr_periods", 0))
    except (OSError, ValueError, KeyError):
        return None


def _gpu_busy_reader(device):
    """-> a callable returning the device's busy percentage now (amdgpu sysfs), or None where the file is not there"""
    import glob
    cards = sorted(glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"))
    if not cards:
        return None
    path = None
    try:
        # the card whose PCI address is the HIP device's (sysfs numbers every card of the host; the process may see one of them)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
            bus = buf.value.decode().lower()
            for c in cards:
                if os.path.realpath(os.path.dirname(c)).lower().endswith(bus):
                    path = c
    except (OSError, AttributeError):
        pass
    if path is None:
        if len(cards) > 1:
            return None                      # which of the host's cards is ours is not known: no figure rather than another GPU's
        path = cards[0]

    def read():
        try:
            return float(open(path).read().strip())
        except (OSError, ValueError):
            return None
    return read if read() is not None else None


def run_banks(n_banks, n_seq, frames, max_fts, distinct=8, spec=None, device=0, seqs=None, want_traj=False, lib_path=None, to_device=None, steady_from=None,
              host_images=False, no_numa_pin=False):
    """n_banks engines of n_seq sequences each, every one on its own host thread with its own device context / stream: the
    device work of one bank overlaps the bookkeeping and the PCIe traffic of the others (independent sequences shard freely, also
    within one GPU).  Whole-run throughput: all frames / wall time from the first step to the last bank's last step.
    host_images: every step's images are handed over as page-locked HOST buffers (FrameHandlerMono::addImage takes a host image): 361 KB
    of PCIe per frame inside the step, one bank's copies beside the other banks' kernels."""
    import threading
    from hso_amd import synth
    spec = spec or synth.EUROC
    if seqs is None:
        seqs = synth.sequences(min(distinct, n_seq), frames, spec=spec, seed0=777)
    res = [None] * n_banks
    traj = [None] * n_banks
    threads_per_bank = [0] * n_banks
    gate = threading.Barrier(n_banks + 1)
    # every bank sizes its worker pool to its share of the host's CPU budget (include/hso_vo.h: hso_vo_host_share)
    from hso_amd import vo as _vo
    _lib = _vo.load_from(lib_path) if lib_path else _vo.load()
    _lib.hso_vo_host_share(n_banks)
    host_quota = int(_lib.hso_vo_host_cpu_quota())
    t_end = [0.0] * n_banks
    step_end = [None] * n_banks          # per bank: perf_counter at the end of every step
    alg = [None] * n_banks
    # steady state: from the step at which every sequence carries a full local-BA window and three live seed batches
    # (Config::coreNKfs() = 7 keyframes, src/config.cpp:34; DepthFilter::Options::max_n_kfs = 3): by default the second half of the run
    if steady_from is None:
        steady_from = (frames - 1) // 2

    def work(b):
        from hso_amd import vo
        cam = synth.camera(spec)
        pick = [seqs[q % len(seqs)] for q in range(n_seq)]
        m = vo.MultiVisualOdometry(cam, n_seq, max_fts, device=device, lib=vo.load_from(lib_path) if lib_path else None)
        if no_numa_pin:
            m.set_options(no_numa_pin=True)
        m.set_first_frames([q["images"][0] for q in pick], [q["depth0"] for q in pick])
        threads_per_bank[b] = int(m.lib.hso_vo_multi_threads(m.h))
        if host_images:
            import torch
            dev = [[torch.from_numpy(np.ascontiguousarray(im)).pin_memory() for im in q["images"]] for q in seqs]
        elif to_device is None:
            import torch
            dev = [[torch.from_numpy(np.ascontiguousarray(im)).cuda(device) for im in q["images"]] for q in seqs]
            torch.cuda.synchronize(device)
        else:
            dev = [[to_device(np.ascontiguousarray(im)) for im in q["images"]] for q in seqs]
        h, w = pick[0]["images"][0].shape
        gate.wait()
        ms, ends = [], []
        for k in range(1, frames):
            ptrs = [dev[q % len(seqs)][k].data_ptr() for q in range(n_seq)]
            t0 = time.perf_counter()
            (m.add_images_host_ptrs if host_images else m.add_images_device)(ptrs, w, h, [float(k)] * n_seq)
            t1 = time.perf_counter()
            ms.append(1e3 * (t1 - t0))
            ends.append(t1)
        t_end[b] = time.perf_counter()
        step_end[b] = ends
        alg[b] = m.alg_bytes()
        sts = [m.status(q) for q in range(n_seq)]
        if want_traj:
            traj[b] = [m.trajectory(q)[1] for q in range(n_seq)]
        kfs = [len(m.keyframes(q)) - 1 for q in range(n_seq)]
        res[b] = dict(ms_per_step_mean=float(np.mean(ms[2:] or ms)), ms_per_step_median=float(np.median(ms[2:] or ms)), keyframes=float(np.mean(kfs)), failures=sum(int(st.stage != 3 or st.result == 2) for st in sts),
                      trans_err_max=max(float(np.linalg.norm(np.array(sts[q].T_f_w.t[:]) - pick[q]["T_f_w"][frames - 1][1])) for q in range(n_seq)))
        m.close()

    th = [threading.Thread(target=work, args=(b,)) for b in range(n_banks)]
    for t in th:
        t.start()
    gate.wait()
    c0 = _cgroup_cpu()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    busy_read = _gpu_busy_reader(device)
    busy = []
    if busy_read:
        stop = threading.Event()

        def sample():
            while not stop.wait(0.02):
                v = busy_read()
                if v is not None:
                    busy.append((time.perf_counter(), v))
        sampler = threading.Thread(target=sample)
        sampler.start()
    for t in th:
        t.join()
    if busy_read:
        stop.set(); sampler.join()
    wall = max(t_end) - t0
    c1 = _cgroup_cpu()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    out = dict(banks=n_banks, sequences_per_bank=n_seq, sequences=n_banks * n_seq, frames=frames - 1, max_fts=max_fts, images="page-locked host" if host_images else "device",
               distinct=len(seqs), frames_per_s=n_banks * n_seq * (frames - 1) / wall, wall_s=wall,
               ms_per_step_mean_per_bank=[r["ms_per_step_mean"] for r in res], ms_per_step_median_per_bank=[r["ms_per_step_median"] for r in res],
               keyframes_per_sequence=float(np.mean([r["keyframes"] for r in res])),
               failures=sum(r["failures"] for r in res), trans_err_max=max(r["trans_err_max"] for r in res))
    # what the host side asked of the kernel over the timed steps: page faults (fresh allocations), context switches (waits, pool hand-offs)
    out["host_rusage"] = dict(minor_faults=ru1.ru_minflt - ru0.ru_minflt, major_faults=ru1.ru_majflt - ru0.ru_majflt,
                              voluntary_switches=ru1.ru_nvcsw - ru0.ru_nvcsw, involuntary_switches=ru1.ru_nivcsw - ru0.ru_nivcsw,
                              user_s=ru1.ru_utime - ru0.ru_utime, system_s=ru1.ru_stime - ru0.ru_stime)
    out["host_cpu_quota"] = host_quota
    out["threads_per_bank"] = threads_per_bank[0]
    # steady state: the window in which every bank is past step `steady_from` and none has finished
    if all(e is not None and len(e) > steady_from + 1 for e in step_end):
        wa, wb = max(e[steady_from] for e in step_end), min(e[-1] for e in step_end)
        if wb > wa:
            steps_in = sum(sum(1 for t in e if wa < t <= wb) for e in step_end)
            out["steady_from_step"] = steady_from
            out["steady_window_s"] = wb - wa
            out["steady_frames_per_s"] = steps_in * n_seq / (wb - wa)
            out["warmup_frames_per_s"] = n_banks * n_seq * steady_from / (max(e[steady_from - 1] for e in step_end) - t0) if steady_from > 0 else None
            in_win = [v for t, v in busy if wa <= t <= wb]
            out["steady_gpu_busy_frac"] = float(np.mean(in_win)) / 100.0 if in_win else None
            # the algorithmic bytes of the chain's kernels over the whole run / its wall time / the HBM peak
            tot = {k: sum(a[k] for a in alg) for k in alg[0]}
            out["alg_bytes_per_frame"] = {k: v / (n_banks * n_seq * (frames - 1)) for k, v in tot.items()}
            out["roofline_frac_hbm"] = sum(tot.values()) / wall / 8.0e12
    out["gpu_busy_frac"] = float(np.mean([v for _, v in busy])) / 100.0 if busy else None
    if c0 and c1:   # host CPUs the process kept busy over the timed steps (incl. the banks' teardown) and CFS periods it was throttled in
        out["host_cpus_used"] = (c1[0] - c0[0]) / 1e6 / max(time.perf_counter() - t0, 1e-9)
        out["host_throttled_periods"] = [c1[1] - c0[1], c1[2] - c0[2]]
    if want_traj:
        return out, [t for b in traj for t in b]
    return out


if __name__ == "__main__":
    if sys.argv[1] == "banks":
        a = [int(x) for x in sys.argv[2:]]
        print(json.dumps(run_banks(a[0], a[1], a[2], a[3], a[4] if len(a) > 4 else 8, no_numa_pin=(len(a) > 5 and a[5] != 0))))   # sixth number != 0: hso_vo_options.no_numa_pin
    else:
        a = [int(x) for x in sys.argv[1:]]
        print(json.dumps(run(a[0], a[1], a[2], a[3] if len(a) > 3 else 8, on_device=(a[4] != 0 if len(a) > 4 else True))))
