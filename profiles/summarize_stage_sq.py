"""profiles/collect_r3_stage_sq.sh output (gpurun_out/stage_sq_<tag>/) -> profiles/<tag>_stage_sq_<stage>.csv: the SQ counters of the
later-stage kernels (k_align_t, k_seed_observe, k_pose) with the same derived fractions profiles/summarize.py reports for
k_track (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_* cycle counters tick once per 4 clocks; SQ_BUSY_CYCLES sums the 32
shader engines; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).
usage: python profiles/summarize_stage_sq.py [tag=r3]"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r3"
SRC = os.path.join(ROOT, "gpurun_out", "stage_sq_" + TAG)
KERNELS = {"align": ["k_align_t"], "seed": ["k_seed_observe", "k_seed_pre", "k_seed_image", "k_seed_post"], "pose": ["k_pose"]}


def main():
    for stage, kern in [(st, k) for st, ks in KERNELS.items() for k in ks]:
        per = defaultdict(list)
        geom = {}
        for path in sorted(glob.glob(os.path.join(SRC, stage + "_sq*_counter_collection.csv"))):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    if kern in r["Kernel_Name"]:
                        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                        geom = dict(grid=r["Grid_Size"], wg=r["Workgroup_Size"], lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"],
                                    sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"])
        if not per:
            continue
        m = {k: sum(v) / len(v) for k, v in per.items()}
        g = lambda c: m.get(c, float("nan"))
        wc = g("SQ_WAVE_CYCLES")
        out = os.path.join(ROOT, "profiles", "%s_stage_sq_%s%s.csv" % (TAG, stage, "" if len(KERNELS[stage]) == 1 or kern == "k_seed_observe" else "_" + kern))
        with open(out, "w") as fh:
            fh.write("# rocprofv3 --pmc SQ passes of `python -m hso_amd.stage_roofline --stage %s --reps 2` (profiles/collect_r3_stage_sq.sh); mean per dispatch of %s\n" % (stage, kern))
            fh.write("# launch: grid %(grid)s threads, workgroup %(wg)s, LDS %(lds)s B, VGPRs %(vgpr)s, SGPRs %(sgpr)s, scratch %(scratch)s B\n" % geom)
            fh.write("kernel,counter,value\n")
            for c in sorted(m):
                fh.write("%s,%s,%.0f\n" % (kern, c, m[c]))
            fh.write("%s,valu_busy_frac (4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * SQ_BUSY_CYCLES / 32)),%.3f\n" % (kern, 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * g("SQ_BUSY_CYCLES") / 32)))
            fh.write("%s,wave_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES),%.3f\n" % (kern, g("SQ_WAIT_ANY") / wc))
            fh.write("%s,issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (kern, g("SQ_WAIT_INST_ANY") / wc))
            fh.write("%s,lds_issue_stall_frac (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES),%.3f\n" % (kern, g("SQ_WAIT_INST_LDS") / wc))
            fh.write("%s,active_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES),%.3f\n" % (kern, g("SQ_ACTIVE_INST_ANY") / wc))
            fh.write("%s,waves_resident_per_simd (SQ_WAVE_CYCLES / (1024 * SQ_BUSY_CYCLES / 32)),%.2f\n" % (kern, wc / (1024 * g("SQ_BUSY_CYCLES") / 32)))
            if "SQ_INSTS_LDS" in m:
                fh.write("%s,lds_bank_conflict_cycles_per_lds_inst,%.3f\n" % (kern, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_INSTS_LDS"), 1)))
                fh.write("%s,valu_insts_per_vmem_rd,%.1f\n" % (kern, g("SQ_INSTS_VALU") / max(g("SQ_INSTS_VMEM_RD"), 1)))
        print(open(out).read())


if __name__ == "__main__":
    main()
