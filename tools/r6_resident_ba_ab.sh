#!/bin/bash
# Same-box A/B of the keyframe path: build/exp/old = the tree of a build of an earlier commit (575adf1: local BA windows assembled on the host, value-passing
# call, patch of the window's points afterwards; 53d7eb2: the Levenberg loop decided on the host; `git worktree add build/exp/old <commit> && (cd build/exp/old && python -m hso_amd.build)`)
# against the working tree (hso_gpu_seq_local_ba + this round's later kernel work); six engines x 128 x 2000, alternating.
#   bash tools/r6_resident_ba_ab.sh   (GPU box, repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT && python -m hso_amd.bank_bench banks 6 128 31 2000 8 > /dev/null 2>&1
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cd $ROOT/build/exp/old; else cd $ROOT; fi
    python -m hso_amd.bank_bench banks 6 128 121 2000 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', 'steady %.0f whole %.0f warm-up %.0f cpus %.1f failures %d' % (d['steady_frames_per_s'], d['frames_per_s'], d['warmup_frames_per_s'], d['host_cpus_used'], d['failures']))"
  done
done
