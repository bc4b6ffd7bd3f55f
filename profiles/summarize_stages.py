"""profiles/collect_stages.sh output -> profiles/<tag>_stage_roofline.csv: per stage, the kernels that ran (calls, average
duration) and, for the stage's dominant kernel, algorithmic bytes per call (hso_amd/stage_roofline.py, SURVEY.md section 8(d)
units) / average kernel duration = achieved GB/s and its fraction of the 8 TB/s HBM peak.
usage: python profiles/summarize_stages.py [tag=r2]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
SRC = os.path.join(ROOT, "gpurun_out", "stages_" + TAG)
KERNELS = {"align": ["k_align"], "seed": ["k_seed_observe"], "pose": ["k_pose"], "frame": ["k_pyramid", "k_sobel"]}


def main():
    rows = []
    for stage, kerns in KERNELS.items():
        try:
            rec = json.loads([l for l in open(os.path.join(SRC, stage + ".log")) if l.startswith("{")][-1])
        except (OSError, IndexError, ValueError):
            continue
        ks = []
        with open(os.path.join(SRC, stage + "_kernel_stats.csv")) as fh:
            for r in csv.DictReader(fh):
                name = r["Name"].replace("void ", "").split("(")[0]
                if any(name.startswith(k) or ("::" + k) in name for k in kerns):
                    ks.append((name, int(r["Calls"]), float(r["AverageNs"])))
        if not ks:
            continue
        ms = sum(a for _, _, a in ks) * 1e-6          # one call of the stage = one dispatch of each of its kernels
        gbs = rec["algorithmic_bytes_per_call"] / (ms * 1e-3) / 1e9
        rows.append((stage, " + ".join(n for n, _, _ in ks), ks[0][1], ms, rec["n"], rec["units"], rec["algorithmic_bytes_per_call"],
                     gbs, gbs / 8000.0, rec["ms_per_call"], json.dumps({k: v for k, v in rec.items() if k.startswith("mean_") or k in ("reached_lk", "feats_per_frame")})))
    path = os.path.join(ROOT, "profiles", TAG + "_stage_roofline.csv")
    with open(path, "w") as fh:
        fh.write("# python -m hso_amd.stage_roofline --stage S under rocprofv3 --kernel-trace --stats (profiles/collect_stages.sh); VGA frames, 128 sequences\n")
        fh.write("stage,kernels,dispatches,kernel_ms_per_call,units_per_call,unit,algorithmic_bytes_per_call,achieved_GBps,frac_of_8TBps,abi_call_ms,detail\n")
        for r in rows:
            fh.write("%s,%s,%d,%.4f,%d,%s,%.0f,%.1f,%.4f,%.3f,\"%s\"\n" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10].replace('"', "'")))
    print(open(path).read())


if __name__ == "__main__":
    main()
