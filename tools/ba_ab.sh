#!/bin/bash
# local BA in one call against the two-call form, same box (run on the GPU box from the repo root)
OUT=${1:-gpurun_out/ba_ab}; mkdir -p $OUT
for r in 1 2; do
HSO_ENGINE_TIMING=1 python -m hso_amd.bank_bench 128 121 2000 8 > $OUT/one_$r.json 2> $OUT/one_$r.err
HSO_BA_TWO_CALLS=1 HSO_ENGINE_TIMING=1 python -m hso_amd.bank_bench 128 121 2000 8 > $OUT/two_$r.json 2> $OUT/two_$r.err
done
for f in one_1 two_1 one_2 two_2; do echo "== $f"; grep "hso engine\] [0-9]* steps" $OUT/$f.err | cut -c1-400; grep "  local BA" $OUT/$f.err | cut -c1-250; grep -o "ba: [a-z ]*call [0-9.]*" $OUT/$f.err; grep -o '"frames_per_s": [0-9.]*' $OUT/$f.json; done
