#!/bin/bash
# Kernel split and GPU-busy fraction of the end-to-end run in bench.py's shape (run on the GPU box from the repo root):
#   bash profiles/collect_r5_banks.sh [banks] [sequences] [frames] [max_fts] [tag]
# One rocprofv3 --kernel-trace --stats pass over hso_amd.bank_bench banks ...; sum of kernel + copy durations / wall = busy fraction.
set -e
B=${1:-6}; N=${2:-128}; F=${3:-121}; M=${4:-2000}; TAG=${5:-r5_banks}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o banks -- python -m hso_amd.bank_bench banks $B $N $F $M 8 > $OUT/banks.log 2>&1 || echo "failed" >> $OUT/errors.txt
rm -f $OUT/*kernel_trace.csv
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/banks_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
line = [l for l in open("$OUT/banks.log") if l.startswith("{")][-1]
d = json.loads(line)
out = dict(banks=$B, sequences_per_bank=$N, frames=$F - 1, max_fts=$M, kernel_and_copy_s=tot * 1e-9, wall_s=d["wall_s"], gpu_busy_frac=tot * 1e-9 / d["wall_s"],
           frames_per_s_under_rocprof=d["frames_per_s"], note="sum of kernel + copy durations (rocprofv3 --kernel-trace --stats) / wall time of the timed steps; concurrent kernels of different banks overlap, so this can exceed the wall fraction actually occupied")
json.dump(out, open("$OUT/gpu_busy.json", "w"), indent=1)
print(out)
PY
ls -la $OUT
