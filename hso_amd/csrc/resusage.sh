#!/bin/bash
# developer helper: VGPR / spill / scratch summary of one .hip translation unit
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I/root/repo/include $HSO_EXTRA_FLAGS \
  -c "$1" -o /tmp/resusage.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - -
