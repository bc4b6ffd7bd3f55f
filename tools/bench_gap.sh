#!/bin/bash
# bench.py's end-to-end section against the standalone run_banks on the SAME rendered sequences, same box (run on the GPU box)
OUT=${1:-gpurun_out/gap}; mkdir -p $OUT
V="python tools/seq_shape_variants.py"
B="python bench.py --cpu-frames 0 --se3-frames 0 --single 0 --steps 3 --warmup 1"
$V $OUT/alone_2024_a.json 121 plain 2024 > $OUT/alone_2024_a.log 2>&1
$B > $OUT/bench_a.log 2>&1
$V $OUT/alone_777.json 121 plain 777 > $OUT/alone_777.log 2>&1
$V $OUT/alone_2024_b.json 121 plain 2024 > $OUT/alone_2024_b.log 2>&1
$B > $OUT/bench_b.log 2>&1
for f in alone_2024_a alone_777 alone_2024_b; do grep -h '^{"variant' $OUT/$f.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$f', 'steady %.0f whole %.0f warm %.0f cpus %.1f kf %.1f' % (r['steady'], r['whole'], r['warmup'], r['cpus'], r['keyframes']))
"; done
for f in bench_a bench_b; do python - $OUT/$f.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print(sys.argv[1], "steady %.0f whole %.0f warmup %.0f cpus %s" % (d["sequences_frames_per_s"], d["sequences_whole_run_frames_per_s"], d["sequences_warmup_frames_per_s"], d.get("host_cpus_used")))
PY
done
