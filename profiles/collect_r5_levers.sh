#!/bin/bash
# Round-5 headline levers, run on the GPU box from the repo root:  bash profiles/collect_r5_levers.sh
#  (b) collect_begin / _end: --overlap-readback 0 / 1 alternating;  (c) features dealt to waves by LDS bank: --deal-features 0 / 1 / 2
#  alternating, then the LDS counters of the tracker with and without the dealing (separate PMC passes).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_levers
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
B="python $ROOT/bench.py --cpu-frames 0 --seq-frames 0 --sequences 0 --single 0 --se3-frames 0"
line() { python -c "
import sys, json
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][-1]
print(sys.argv[2], round(d['value']), round(d['ms_per_step'], 3), round(d['roofline']['launch_ms'], 3))" $1 "$2" >> $OUT/levers.txt; }
for rep in 1 2; do
  for ov in 0 1; do $B --steps 20 --warmup 4 --overlap-readback $ov > $OUT/ov${ov}_$rep.log 2>&1; line $OUT/ov${ov}_$rep.log "overlap-readback $ov"; done
done
for rep in 1 2 3; do
  for dl in 0 1 2; do $B --steps 20 --warmup 4 --deal-features $dl > $OUT/deal${dl}_$rep.log 2>&1; line $OUT/deal${dl}_$rep.log "deal-features $dl"; done
done
for dl in 0 1; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
    --output-format csv -d $OUT -o deal${dl}_lds -- $B --steps 2 --warmup 1 --deal-features $dl > $OUT/deal${dl}_lds.log 2>&1 || echo "lds pass $dl failed" >> $OUT/errors.txt
done
python - <<PY
import csv, glob, collections
for dl in (0, 1):
    fs = glob.glob("$OUT/**/deal%d_lds_counter_collection.csv" % dl, recursive=True)
    if not fs: print("no counter file for", dl); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "k_track" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:24], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:24], r["Counter_Name"])] += 1
    with open("$OUT/levers.txt", "a") as f:
        for k in sorted(acc): f.write("deal-features %d  %-26s %-22s %.4g per launch (%d launches)\n" % (dl, k[0], k[1], acc[k] / n[k], n[k]))
PY
cat $OUT/levers.txt
