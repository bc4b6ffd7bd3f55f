#!/bin/bash
# Development aid (GPU box): the engine's end-to-end throughput for several bank shapes, each repeated, one JSON line per run.
#   bash tools/bank_sweep.sh <outfile> <repeats> "<banks> <sequences>" ...
OUT=$1; REP=$2; shift 2
: > $OUT
for cfg in "$@"; do
  for r in $(seq $REP); do
    python -m hso_amd.bank_bench banks $cfg 24 2000 8 2>/dev/null | tail -1 >> $OUT
  done
done
python - "$OUT" <<'P'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    print(d["banks"], "x", d["sequences_per_bank"], "frames/s %.0f" % d["frames_per_s"], "median ms/step", [round(x, 1) for x in d["ms_per_step_median_per_bank"]])
P
