#!/bin/bash
# Kernel timeline of several engines on one GPU (run on the GPU box from the repo root): bash tools/collect_overlap.sh [banks] [sequences] [frames] [tag]
B=${1:-6}; N=${2:-128}; F=${3:-61}; TAG=${4:-overlap}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o banks -- python -m hso_amd.bank_bench banks $B $N $F 2000 8 > $OUT/banks.log 2>&1 || echo failed >> $OUT/errors.txt
T=$(ls $OUT/*kernel_trace.csv | head -1)
python $ROOT/tools/kernel_overlap.py $T $OUT/overlap.json > /dev/null
# the second half of the run only (steady state): rows whose start lies past the median start
python - $T $OUT <<'PY'
import csv, sys, subprocess
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = rows[len(rows) // 2:]
with open(sys.argv[2] + "/second_half.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(half)
PY
python $ROOT/tools/kernel_overlap.py $OUT/second_half.csv $OUT/overlap_second_half.json > /dev/null
head -2 $T > $OUT/trace_head.csv
gzip -9 $OUT/second_half.csv
rm -f $T
ls -la $OUT
