"""Condense the rocprofv3 output of profiles/collect_r1.sh (gpurun_out/prof_r1/) into the small
tracked summaries under profiles/: r1_kernel_stats.csv (copied), r1_pmc_summary.csv (mean counter
value per kernel) and pmc_k_track.json (what bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r1")
OUT = os.path.join(ROOT, "profiles")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def main():
    for f in ("r1_kernel_stats.csv", "r1_domain_stats.csv", "r1_stages_kernel_stats.csv", "stage_bench.log"):
        p = os.path.join(SRC, f)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(OUT, f if f.startswith("r1_") else "r1_" + f))
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
    with open(os.path.join(OUT, "r1_pmc_summary.csv"), "w") as fh:
        fh.write("counter,kernel,dispatches,mean_value_KB_per_dispatch\n")
        for cname, kname, n, mean in sorted(rows):
            fh.write("%s,%s,%d,%.1f\n" % (cname, kname, n, mean))
    kt = [k for (c, k) in per if k.startswith("k_track")]
    if kt:
        k = kt[0]
        note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, mean over the dispatches of "
                "`python bench.py --steps 3 --warmup 1 --cpu-frames 0` (profiles/collect_r1.sh). Raw counter values: the "
                "gfx950 x2 read-side correction of MI355X_MICROARCH.md is calibrated for 16 B/lane streams only; this "
                "kernel's global reads are 4-8 B/lane, so the raw value is a lower bound and 2x the read part an upper bound.")
        batch = 512
        try:   # the batch the PMC runs used: the bench line in their log
            line = [l for l in open(os.path.join(SRC, "bench_fetch.log")) if l.startswith("{")][-1]
            batch = json.loads(line)["config"]["frames_per_gpu_per_step"]
        except (OSError, IndexError, KeyError, ValueError):
            pass
        json.dump({"round": 1, "kernel": k, "batch": batch, "feats": 2000,
                   "fetch_size_kb": per.get(("FETCH_SIZE", k)), "write_size_kb": per.get(("WRITE_SIZE", k)), "note": note},
                  open(os.path.join(OUT, "pmc_k_track.json"), "w"), indent=1)
    print("wrote summaries from", SRC)


if __name__ == "__main__":
    main()
