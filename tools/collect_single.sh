#!/bin/bash
# One sequence through the engine: per-step phases and the kernel split (run on the GPU box from the repo root): bash tools/collect_single.sh [tag] [feats]
TAG=${1:-single}; N=${2:-2000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
python -m hso_amd.latency_bench --reps 5 --frames 121 --feats $N > $OUT/plain.json 2> $OUT/plain.err
HSO_ENGINE_TIMING=2 python -m hso_amd.latency_bench --reps 5 --frames 121 --feats $N > $OUT/timing.json 2> $OUT/timing.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o single -- python -m hso_amd.latency_bench --reps 5 --frames 121 --feats $N > $OUT/prof.json 2> $OUT/prof.err
T=$(ls $OUT/*kernel_trace.csv | head -1)
python $ROOT/tools/kernel_overlap.py $T $OUT/overlap.json > /dev/null
rm -f $T
tail -1 $OUT/plain.json | cut -c1-1200
