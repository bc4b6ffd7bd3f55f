#!/bin/bash
# Alternating A/B of the end-to-end run: the built library (inverse-compositional row windows) against build/exp/libhso_gpu_noicrows.so,
# one process per run, A B A B A B on one box (the box-to-box and first-run effects of profiles/r6_engine_host.md section 2 taken out).
cd ${GRAFT_REPO_ROOT:-.}
cp hso_amd/csrc/libhso_gpu.so /tmp/libhso_gpu_main.so
python tools/r6_repeat_banks.py 1 6 128 2>&1 | grep "^{" | sed 's/^/warm-up of the box /'
for i in 1 2 3 4; do
  cp /tmp/libhso_gpu_main.so hso_amd/csrc/libhso_gpu.so
  python tools/r6_repeat_banks.py 1 6 128 2>&1 | grep "^{" | sed 's/^/rows    /'
  cp build/exp/libhso_gpu_noicrows.so hso_amd/csrc/libhso_gpu.so
  python tools/r6_repeat_banks.py 1 6 128 2>&1 | grep "^{" | sed 's/^/generic /'
done
cp /tmp/libhso_gpu_main.so hso_amd/csrc/libhso_gpu.so
