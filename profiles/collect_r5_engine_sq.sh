#!/bin/bash
# SQ counters of the engine's OWN launches: one engine x 128 sequences x 2000 features, 40 steps, the kernels as
# the resident chain / BA / seed calls launch them.  Counter passes only (no trace domains alongside --pmc).  Run on the GPU box:
#   bash profiles/collect_r5_engine_sq.sh  &&  python profiles/summarize_r5_engine.py      (the second step also works off-box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_engine_sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT -o sq1 -- python -m hso_amd.bank_bench 128 41 2000 8 > $OUT/sq1.log 2>&1 || echo "sq1 failed" >> $OUT/errors.txt
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES \
  --output-format csv -d $OUT -o sq2 -- python -m hso_amd.bank_bench 128 41 2000 8 > $OUT/sq2.log 2>&1 || echo "sq2 failed" >> $OUT/errors.txt
# condensed on the box (the per-dispatch files are large)
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        d = (f, k, r["Dispatch_Id"])
        if (d, r["Counter_Name"]) not in seen: n[(k, r["Counter_Name"])] += 1; seen.add((d, r["Counter_Name"]))
with open("$OUT/engine_sq_per_kernel.csv", "w") as o:
    o.write("kernel,counter,mean_per_dispatch,dispatches\n")
    for key in sorted(acc): o.write("%s,%s,%.6g,%d\n" % (key[0], key[1], acc[key] / max(n[key], 1), n[key]))
PY
rm -f $OUT/*/*counter_collection.csv $OUT/*counter_collection.csv
ls -la $OUT
