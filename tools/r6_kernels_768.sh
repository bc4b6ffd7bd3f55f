#!/bin/bash
# kernel totals of ONE engine at two bank sizes (how the chain's kernels scale with the number of jobs per launch)
#   bash tools/r6_kernels_768.sh   (GPU box, repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for N in 128 768; do
  OUT=$ROOT/gpurun_out/r6_kern_$N
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bank -- python -m hso_amd.bank_bench $N 121 2000 8 > $OUT/bank.log 2>&1 || echo failed >> $OUT/errors.txt
  rm -f $OUT/*kernel_trace.csv
  tail -1 $OUT/bank.log | cut -c1-300
done
