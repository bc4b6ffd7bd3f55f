#!/bin/bash
# Hardware-queue count and per-phase wall time of six engines on one GPU (run on the GPU box from the repo root)
OUT=${1:-gpurun_out/qsweep}; mkdir -p $OUT
V="python tools/seq_shape_variants.py"
$V $OUT/q4_a.json 121 plain > $OUT/q4_a.log 2>&1
for q in 8 16 4 8 24 4 8; do
  n=$(ls $OUT | grep -c "^q${q}_.*json$")
  GPU_MAX_HW_QUEUES=$q $V $OUT/q${q}_r$n.json 121 plain > $OUT/q${q}_r$n.log 2>&1
done
GPU_MAX_HW_QUEUES=8 $V $OUT/q8_8banks.json 121 plain8 > $OUT/q8_8banks.log 2>&1
GPU_MAX_HW_QUEUES=16 $V $OUT/q16_12x64.json 121 plain12x64 > $OUT/q16_12x64.log 2>&1
HSO_ENGINE_TIMING=1 $V $OUT/timing6.json 121 plain > $OUT/timing6.log 2>&1
HSO_ENGINE_TIMING=1 python -m hso_amd.bank_bench 128 121 2000 8 > $OUT/timing1.log 2>&1
grep -h '^{"variant' $OUT/q*.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['variant'], r['banks'], 'steady %.0f whole %.0f warm %.0f busy %s cpus %.1f' % (r['steady'], r['whole'], r['warmup'], r['gpu_busy'], r['cpus']))
" > $OUT/summary.txt
ls $OUT/q*.json | paste - $OUT/summary.txt
