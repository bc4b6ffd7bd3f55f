"""Condense the rocprofv3 output of profiles/collect_rN.sh (gpurun_out/prof_<tag>/) into the small
tracked summaries under profiles/:
  <tag>_kernel_stats.csv      (copied from the --kernel-trace --stats pass)
  <tag>_pmc_summary.csv       (mean counter value per kernel and dispatch, every PMC pass)
  <tag>_sq_counters.csv       (SQ passes of k_track with the derived busy / wait fractions)
  r2_pmc_k_track.json         (HBM-side bytes per k_track launch, keyed by the workload it was
                               measured on — what bench.py reports as roofline.traffic)
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

    kt = sorted({k for (c, k) in per if k.startswith("k_track")})
    # SQ counters of the tracker with the derived fractions (MI355X_MICROARCH.md "rocprofv3 PMC slots":
    # WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES; SQ_* cycle counters tick once per 4 clocks;
    # SQ_BUSY_CYCLES sums the 32 shader engines)
    if kt and ("SQ_WAVE_CYCLES", kt[0]) in per:
        k = kt[0]
        g = lambda c: per.get((c, k), float("nan"))
        with open(os.path.join(OUT, TAG + "_sq_counters.csv"), "w") as fh:
            fh.write("# rocprofv3 --pmc SQ passes of `python bench.py --cpu-frames 0 --steps 2 --warmup 1` (profiles/collect_%s.sh); mean per k_track dispatch\n" % RND)
            fh.write("counter,value\n")
            for (c, kk), v in sorted(per.items()):
                if kk == k and c.startswith("SQ_"):
                    fh.write("%s,%.0f\n" % (c, v))
            wc = g("SQ_WAVE_CYCLES")
            fh.write("valu_busy_frac,%.3f\n" % (4 * g("SQ_ACTIVE_INST_VALU") / (1024 * g("SQ_BUSY_CYCLES") / 32)))
            fh.write("wave_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES),%.3f\n" % (g("SQ_WAIT_ANY") / wc))
            fh.write("issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (g("SQ_WAIT_INST_ANY") / wc))
            fh.write("lds_issue_stall_frac (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES),%.3f\n" % (g("SQ_WAIT_INST_LDS") / wc))
            fh.write("active_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (g("SQ_ACTIVE_INST_ANY") / wc))
            if ("SQ_INSTS_LDS", k) in per:
                fh.write("lds_bank_conflict_cycles_per_lds_inst,%.3f\n" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS")))

    if kt and ("FETCH_SIZE", kt[0]) in per and ("WRITE_SIZE", kt[0]) in per:
        k = kt[0]
        line = bench_line("bench_fetch.log")
        cfg = line["config"] if line else {}
        fetch_kb, write_kb = per[("FETCH_SIZE", k)], per[("WRITE_SIZE", k)]
        rec = {"round": RND, "tag": TAG, "kernel": k,
               "shape": cfg.get("shape"), "batch": cfg.get("frames_per_gpu_per_step"), "feats": 2000,
               "inverse": 1 if cfg.get("mode") == "inverse_compositional" else 0,
               "scenes": cfg.get("distinct_scenes_per_rank"),
               "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
               # the guide's gfx950 correction: FETCH_SIZE counts 128-B read requests at 64 B -> x2 on the read
               # side; WRITE_SIZE is taken as reported
               "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, mean over the k_track dispatches of "
                       "`python bench.py --cpu-frames 0 --steps 3 --warmup 1`; hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE "
                       "(MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports half the bytes of wide reads; this kernel's "
                       "global reads are partly 4-8 B/lane, for which the x2 is an upper bound)."}
        path = os.path.join(OUT, "r2_pmc_k_track.json")
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            doc = {"records": []}
        doc["records"] = [e for e in doc["records"] if (e["shape"], e["batch"], e["inverse"], e["scenes"]) !=
                          (rec["shape"], rec["batch"], rec["inverse"], rec["scenes"])] + [rec]
        json.dump(doc, open(path, "w"), indent=1)
    print("wrote summaries from", SRC)


if __name__ == "__main__":
    main()
