#!/bin/bash
# Per-kernel roofline of the later stages (run on the GPU box from the repo root):
#   bash profiles/collect_stages.sh [tag]  &&  python profiles/summarize_stages.py [tag]
# One rocprofv3 --kernel-trace --stats pass per stage, each over hso_amd.stage_roofline at its multi-sequence size.
set -e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stages_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for S in align seed pose frame; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $S -- python -m hso_amd.stage_roofline --stage $S > $OUT/$S.log 2>&1 || echo "$S failed" >> $OUT/errors.txt
done
rm -f $OUT/*kernel_trace.csv
ls -la $OUT
