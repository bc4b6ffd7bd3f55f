#!/bin/bash
# Round-6 profile recipe (collect_r4.sh with the image-upload leg off and an inverse-compositional kernel-trace pass) (run on the GPU box through gpurun from the repo root):
#   bash profiles/collect_r6.sh [tag]  &&  python profiles/summarize.py r4 [tag]   (the second step also works off-box)
# All passes profile the DEFAULT bench command (EuRoC-shaped 752x480, 2000 points, 4096 pairs, 64 scenes):
#  1) kernel trace + stats, 2)+3) HBM-side PMC passes (separate runs: FETCH_SIZE and WRITE_SIZE do not
#  fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), 4)+5) SQ counter passes for k_track
#  (issue / wait split; LDS instructions and bank conflicts; VMEM instructions incl. scratch).
set -e
TAG=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
B="python $ROOT/bench.py --cpu-frames 0 --seq-frames 0 --sequences 0 --single 0 --se3-frames 0 --h2d 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG} -- $B --steps 10 --warmup 2 > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o ${TAG}_fetch -- $B --steps 3 --warmup 1 > $OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o ${TAG}_write -- $B --steps 3 --warmup 1 > $OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS \
  --output-format csv -d $OUT -o ${TAG}_sq1 -- $B --steps 2 --warmup 1 > $OUT/bench_sq1.log 2>&1 || echo "sq1 pass failed" >> $OUT/errors.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT \
  --output-format csv -d $OUT -o ${TAG}_sq2 -- $B --steps 2 --warmup 1 > $OUT/bench_sq2.log 2>&1 || echo "sq2 pass failed" >> $OUT/errors.txt
rm -f $OUT/*kernel_trace.csv   # per-dispatch traces are large; the stats files carry what is summarised
ls -la $OUT
# the inverse-compositional mode (the engine's mode) under the same kernel trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_ic -- $B --inverse 1 --steps 10 --warmup 2 > $OUT/bench_trace_ic.log 2>&1 || echo "ic pass failed" >> $OUT/errors.txt
rm -f $OUT/*kernel_trace.csv
