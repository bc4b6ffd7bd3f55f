# kernel times of the seed stage (rocprofv3 kernel trace over hso_amd.stage_roofline --stage seed)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/seed_stage
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o seed -- python -m hso_amd.stage_roofline --stage seed > $OUT/seed.log 2>&1
grep -h "k_seed" $OUT/seed_kernel_stats.csv | head -5
rm -f $OUT/*kernel_trace.csv
