#!/bin/bash
# Kernel split of the sequence engine end to end (run on the GPU box from the repo root):
#   bash profiles/collect_r4_bank.sh [sequences] [frames] [max_fts] [tag]
# One rocprofv3 --kernel-trace --stats pass over hso_amd.bank_bench; the engine's own phase split (host wall time) goes to the log.
set -e
N=${1:-64}; F=${2:-30}; M=${3:-2000}; TAG=${4:-r4_bank}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT HSO_ENGINE_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bank -- python -m hso_amd.bank_bench $N $F $M 8 > $OUT/bank.log 2>&1 || echo "failed" >> $OUT/errors.txt
rm -f $OUT/*kernel_trace.csv
ls -la $OUT
