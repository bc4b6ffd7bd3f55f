#!/bin/bash
# The engine's threads on the device's NUMA node (hso_vo_options.no_numa_pin = 0, the default) against affinities left alone; six
# engines x 128 x 2000, alternating on one box after an untimed full run.   bash tools/r6_numa_ab.sh   (GPU box, repo root)
cd ${GRAFT_REPO_ROOT:-.}
python -c "
from hso_amd import capi
c = capi.Context(); s = sorted(c.device_cpulist()); print('device node CPUs:', len(s), s[:2], '...', s[-2:]); c.close()"
python -m hso_amd.bank_bench banks 6 128 121 2000 8 > /dev/null 2>&1
for rep in 1 2 3; do
  for v in 0 1; do
    python -m hso_amd.bank_bench banks 6 128 121 2000 8 $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('no_numa_pin=$v', 'steady %.0f whole %.0f warm-up %.0f cpus %.1f' % (d['steady_frames_per_s'], d['frames_per_s'], d['warmup_frames_per_s'], d['host_cpus_used']))"
  done
done
