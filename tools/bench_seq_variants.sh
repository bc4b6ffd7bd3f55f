#!/bin/bash
# bench.py's end-to-end section under a few host-side knobs (run on the GPU box from the repo root): bash tools/bench_seq_variants.sh [out dir]
OUT=${1:-gpurun_out/r5_v}
mkdir -p $OUT
B="python bench.py --cpu-frames 0 --se3-frames 0 --single 0 --steps 4 --warmup 1"
$B > $OUT/default.log 2>&1
HSO_HOST_SERIAL=1 $B > $OUT/serial.log 2>&1
HSO_POLLING_SYNC=1 $B > $OUT/poll.log 2>&1
HSO_HOST_SERIAL=1 HSO_POLLING_SYNC=1 $B > $OUT/serial_poll.log 2>&1
for f in default serial poll serial_poll; do
python - $OUT/$f.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print(sys.argv[1], "steady %.0f whole %.0f warmup %.0f cpus %s busy %s" % (d["sequences_frames_per_s"], d["sequences_whole_run_frames_per_s"], d["sequences_warmup_frames_per_s"], d.get("host_cpus_used"), d.get("sequences_gpu_busy_frac")))
PY
done
