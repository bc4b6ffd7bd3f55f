#!/bin/bash
# the device library's host loops on the engine's pool against the calling thread, same box (run on the GPU box from the repo root)
OUT=${1:-gpurun_out/pool_ab}; mkdir -p $OUT
for r in 1 2; do
HSO_ENGINE_TIMING=1 python -m hso_amd.bank_bench 128 121 2000 8 > $OUT/solo_pool_$r.json 2> $OUT/solo_pool_$r.err
HSO_ENGINE_NO_LIB_POOL=1 HSO_ENGINE_TIMING=1 python -m hso_amd.bank_bench 128 121 2000 8 > $OUT/solo_nopool_$r.json 2> $OUT/solo_nopool_$r.err
done
for f in solo_pool_1 solo_nopool_1 solo_pool_2 solo_nopool_2; do echo "== $f"; grep "hso engine\] [0-9]* steps" $OUT/$f.err | cut -c60-400; grep -o "ba: [a-z ]*call [0-9.]*\|flush: device calls [0-9.]*" $OUT/$f.err | tr '\n' ' '; grep -o '"frames_per_s": [0-9.]*' $OUT/$f.json; done
V="python tools/seq_shape_variants.py"
for r in 1 2; do
$V $OUT/six_pool_$r.json 121 plain > $OUT/six_pool_$r.log 2>&1
HSO_ENGINE_NO_LIB_POOL=1 $V $OUT/six_nopool_$r.json 121 plain > $OUT/six_nopool_$r.log 2>&1
done
for f in six_pool_1 six_nopool_1 six_pool_2 six_nopool_2; do grep -h '^{"variant' $OUT/$f.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$f', 'steady %.0f whole %.0f warm %.0f cpus %.1f' % (r['steady'], r['whole'], r['warmup'], r['cpus']))
"; done
