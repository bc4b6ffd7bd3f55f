set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
nproc; cat /sys/fs/cgroup/cpu.max
for n in 128 384 768; do
  HSO_ENGINE_TIMING=1 timeout 600 python -m hso_amd.bank_bench $n 121 2000 8 > gpurun_out/r6a/one_$n.json 2> gpurun_out/r6a/one_$n.err
  tail -3 gpurun_out/r6a/one_$n.err
done
timeout 600 python -m hso_amd.bank_bench banks 6 128 121 2000 8 > gpurun_out/r6a/banks6.json 2> gpurun_out/r6a/banks6.err
cat gpurun_out/r6a/banks6.json
