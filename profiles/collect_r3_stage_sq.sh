#!/bin/bash
# SQ counter passes for the later-stage kernels (k_align_t, k_seed_observe, k_pose): the evidence behind "issue-bound, not
# bandwidth-bound" (VERDICT r2 item 4).  Run on the GPU box from the repo root:
#   bash profiles/collect_r3_stage_sq.sh [tag]  &&  python profiles/summarize_stage_sq.py [tag]
# Same two SQ passes profiles/collect_r2.sh runs for k_track, over hso_amd.stage_roofline at its multi-sequence size
# (counter passes only: no trace domains alongside --pmc).
set -e
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stage_sq_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for S in ${STAGES:-align seed pose}; do
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS \
    --output-format csv -d $OUT -o ${S}_sq1 -- python -m hso_amd.stage_roofline --stage $S --reps 2 > $OUT/${S}_sq1.log 2>&1 || echo "$S sq1 failed" >> $OUT/errors.txt
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT \
    --output-format csv -d $OUT -o ${S}_sq2 -- python -m hso_amd.stage_roofline --stage $S --reps 2 > $OUT/${S}_sq2.log 2>&1 || echo "$S sq2 failed" >> $OUT/errors.txt
done
ls -la $OUT
