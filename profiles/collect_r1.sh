#!/bin/bash
# Round-1 profile recipe (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect_r1.sh && python profiles/summarize.py   (the second step also works here)
# 1) kernel trace + stats of the default bench command, 2)+3) PMC passes (separate runs:
# FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# 4) kernel trace + stats of the later pipeline stages (hso_amd/stage_bench.py).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r1
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r1 -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-frames 0 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o r1_fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-frames 0 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o r1_write -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-frames 0 > $OUT/bench_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r1_stages -- python -m hso_amd.stage_bench --reps 10 > $OUT/stage_bench.log 2>&1
rm -f $OUT/*kernel_trace.csv   # per-dispatch traces are large; the stats files carry what is summarised
ls -la $OUT
