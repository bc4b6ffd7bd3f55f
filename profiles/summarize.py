"""Condense the rocprofv3 output of profiles/collect_rN.sh (gpurun_out/prof_<tag>/) into the small
tracked summaries under profiles/:
  <tag>_kernel_stats.csv      (copied from the --kernel-trace --stats pass)
  <tag>_pmc_summary.csv       (mean counter value per kernel and dispatch, every PMC pass)
  <tag>_sq_counters.csv       (SQ passes of k_track with the derived busy / wait fractions)
  <round>_pmc_k_track.json    (HBM-side bytes and VALU instructions per k_track launch, keyed by the workload they
                               were measured on — what bench.py reports as roofline.traffic / roofline_valu)
usage: python profiles/summarize.py [round-prefix=r2] [tag=<round-prefix>]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r2"
TAG = sys.argv[2] if len(sys.argv) > 2 else RND
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
OUT = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def bench_line(log):
    try:
        return json.loads([l for l in open(os.path.join(SRC, log)) if l.startswith("{")][-1])
    except (OSError, IndexError, ValueError):
        return None


def main():
    for f in glob.glob(os.path.join(SRC, "*_kernel_stats.csv")) + glob.glob(os.path.join(SRC, "*_domain_stats.csv")):
        shutil.copy(f, os.path.join(OUT, os.path.basename(f)))
    rows = []
    per = {}
    for path in sorted(glob.glob(os.path.join(SRC, "*_counter_collection.csv"))):
        acc = defaultdict(list)
        with open(path) as fh:
            for r in csv.DictReader(fh):
                acc[(r["Counter_Name"], short(r["Kernel_Name"]))].append(float(r["Counter_Value"]))
        for (cname, kname), vals in acc.items():
            rows.append((cname, kname, len(vals), sum(vals) / len(vals)))
            per[(cname, kname)] = sum(vals) / len(vals)
    with open(os.path.join(OUT, TAG + "_pmc_summary.csv"), "w") as fh:
        fh.write("counter,kernel,dispatches,mean_value_per_dispatch\n")
        for cname, kname, n, mean in sorted(rows):
            fh.write("%s,%s,%d,%.1f\n" % (cname, kname, n, mean))

    # the tracker is two kernels per batch since the two-launch split (trk2:: coarse levels, trk1:: the finest level)
    kt = sorted({k for (c, k) in per if "k_track" in k})
    # SQ counters of the tracker with the derived fractions (MI355X_MICROARCH.md "rocprofv3 PMC slots":
    # WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES; SQ_* cycle counters tick once per 4 clocks;
    # SQ_BUSY_CYCLES sums the 32 shader engines)
    if kt and ("SQ_WAVE_CYCLES", kt[0]) in per:
        with open(os.path.join(OUT, TAG + "_sq_counters.csv"), "w") as fh:
            fh.write("# rocprofv3 --pmc SQ passes of `python bench.py --cpu-frames 0 --steps 2 --warmup 1` (profiles/collect_%s.sh); mean per dispatch, per tracker kernel\n" % RND)
            fh.write("kernel,counter,value\n")
            for k in kt:
                g = lambda c: per.get((c, k), float("nan"))
                for (c, kk), v in sorted(per.items()):
                    if kk == k and c.startswith("SQ_"):
                        fh.write("%s,%s,%.0f\n" % (k, c, v))
                wc = g("SQ_WAVE_CYCLES")
                fh.write("%s,valu_busy_frac,%.3f\n" % (k, 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * g("SQ_BUSY_CYCLES") / 32)))
                fh.write("%s,wave_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES),%.3f\n" % (k, g("SQ_WAIT_ANY") / wc))
                fh.write("%s,issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (k, g("SQ_WAIT_INST_ANY") / wc))
                fh.write("%s,lds_issue_stall_frac (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES),%.3f\n" % (k, g("SQ_WAIT_INST_LDS") / wc))
                fh.write("%s,active_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (k, g("SQ_ACTIVE_INST_ANY") / wc))
                if ("SQ_INSTS_LDS", k) in per:
                    fh.write("%s,lds_bank_conflict_cycles_per_lds_inst,%.3f\n" % (k, g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))

    # the bench line's roofline fraction, reproduced from the kernel-stats pass: algorithmic bytes per batch (the line's own
    # figure) / summed average duration of the tracker kernels of one batch / 8 TB/s
    line = bench_line("bench_trace.log")
    stats = [os.path.join(SRC, TAG + "_kernel_stats.csv")] if os.path.exists(os.path.join(SRC, TAG + "_kernel_stats.csv")) else glob.glob(os.path.join(SRC, "*_kernel_stats.csv"))   # the pass of THIS tag (round 6 adds a <tag>_ic pass)
    if line and stats:
        dur = {}
        with open(stats[0]) as fh:
            for r in csv.DictReader(fh):
                if "k_track" in r["Name"]:
                    dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
        tot_ms = sum(v[1] for v in dur.values()) * 1e-6
        rf = line["roofline"]
        with open(os.path.join(OUT, TAG + "_roofline_check.txt"), "w") as fh:
            fh.write("bench line under rocprofv3 --kernel-trace --stats (bench_trace.log): launch_ms %.3f, frac %.4f, algorithmic bytes / batch %.0f\n"
                     % (rf["launch_ms"], rf["frac"], rf["algorithmic_bytes_per_launch"]))
            for k, (calls, avg) in sorted(dur.items()):
                fh.write("kernel stats: %s  calls %d  average %.3f ms\n" % (k, calls, avg * 1e-6))
            fh.write("sum of the tracker kernels of one batch: %.3f ms -> %.1f GB/s -> frac %.4f of 8000 GB/s\n"
                     % (tot_ms, rf["algorithmic_bytes_per_launch"] / (tot_ms * 1e-3) / 1e9, rf["algorithmic_bytes_per_launch"] / (tot_ms * 1e-3) / 8e12))

    if kt and ("FETCH_SIZE", kt[0]) in per and ("WRITE_SIZE", kt[0]) in per:
        k = " + ".join(kt)
        line = bench_line("bench_fetch.log")
        cfg = line["config"] if line else {}
        # one batch = one dispatch of every tracker kernel: the per-dispatch means add up
        fetch_kb = sum(per[("FETCH_SIZE", x)] for x in kt)
        write_kb = sum(per[("WRITE_SIZE", x)] for x in kt)
        rec = {"round": RND, "tag": TAG, "kernel": k,
               "shape": cfg.get("shape"), "batch": cfg.get("frames_per_gpu_per_step"), "feats": 2000,
               "inverse": 1 if cfg.get("mode") == "inverse_compositional" else 0,
               "scenes": cfg.get("distinct_scenes_per_rank"), "min_level": cfg.get("min_level", 1),
               "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
               # the guide's gfx950 correction: FETCH_SIZE counts 128-B read requests at 64 B -> x2 on the read
               # side; WRITE_SIZE is taken as reported
               "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, mean over the k_track dispatches of "
                       "`python bench.py --cpu-frames 0 --steps 3 --warmup 1`; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                       "(MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports half the bytes of wide reads; this kernel's "
                       "global reads are partly 4-8 B/lane, for which the x2 is an upper bound)."}
        if all(("SQ_INSTS_VALU", x) in per and ("SQ_ACTIVE_INST_VALU", x) in per and ("SQ_BUSY_CYCLES", x) in per for x in kt):
            # what bench.py's roofline_valu needs: wave-level VALU instructions of one batch (both launches) and the VALU-busy
            # fraction over the two launches together (busy cycles add up: the launches run one after the other)
            rec["valu_insts_per_launch"] = sum(per[("SQ_INSTS_VALU", x)] for x in kt)
            rec["valu_busy_frac"] = 4 * sum(per[("SQ_ACTIVE_INST_VALU", x)] for x in kt) / (1024 * sum(per[("SQ_BUSY_CYCLES", x)] for x in kt) / 32)
        path = os.path.join(OUT, RND + "_pmc_k_track.json")
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            doc = {"records": []}
        doc["records"] = [e for e in doc["records"] if (e["shape"], e["batch"], e["inverse"], e["scenes"], e.get("min_level", 1)) !=
                          (rec["shape"], rec["batch"], rec["inverse"], rec["scenes"], rec["min_level"])] + [rec]
        json.dump(doc, open(path, "w"), indent=1)
    print("wrote summaries from", SRC)


if __name__ == "__main__":
    main()
