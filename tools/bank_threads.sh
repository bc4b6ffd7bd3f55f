#!/bin/bash
# Development aid (GPU box): engine throughput against the worker threads per bank.  bash tools/bank_threads.sh "<banks> <seqs>" t1 t2 ...
CFG=$1; shift
for t in "$@"; do
  for r in 1 2; do
    echo -n "cfg $CFG threads $t: "
    HSO_ENGINE_THREADS=$t timeout 120 python -m hso_amd.bank_bench banks $CFG 24 2000 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']))"
  done
done
