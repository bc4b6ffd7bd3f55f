#!/bin/bash
# Kernel-level split of the whole per-frame chain (hso_amd/chain_bench.py: 256 sequences, EuRoC shape, 2000 features, 4000 map
# points, 6000 seeds per sequence): run on the GPU box from the repo root:
#   bash profiles/collect_r3_chain.sh [tag]
# -> gpurun_out/chain_<tag>/chain_kernel_stats.csv (copy to profiles/<tag>_chain_kernel_stats.csv) + the bench line of the same run
set -e
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/chain_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o chain -- python -m hso_amd.chain_bench --nseq 256 --distinct 2 > $OUT/chain.log 2>&1
rm -f $OUT/*kernel_trace.csv
ls -la $OUT
