#!/bin/bash
# Does the allocator's page-fault / trim churn cost the six engines anything?  (run on the GPU box from the repo root)
OUT=${1:-gpurun_out/malloc}; mkdir -p $OUT
V="python tools/seq_shape_variants.py"
$V $OUT/plain_a.json 121 plain > $OUT/plain_a.log 2>&1
MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 $V $OUT/keep_a.json 121 plain > $OUT/keep_a.log 2>&1
$V $OUT/plain_b.json 121 plain > $OUT/plain_b.log 2>&1
MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 $V $OUT/keep_b.json 121 plain > $OUT/keep_b.log 2>&1
MALLOC_ARENA_MAX=64 MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967295 MALLOC_TOP_PAD_=268435456 $V $OUT/keep_arena.json 121 plain > $OUT/keep_arena.log 2>&1
for f in plain_a keep_a plain_b keep_b keep_arena; do grep -h '^{"variant' $OUT/$f.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$f', 'steady %.0f whole %.0f warm %.0f cpus %.1f' % (r['steady'], r['whole'], r['warmup'], r['cpus']), r['rusage'])
"; done | tee $OUT/summary.txt
