#!/bin/bash
# frame build in cache-sized chunks (HSO_FRAME_CHUNK_MB; hso_frame.hip): the headline step with 0 (one launch pair for the batch) / 48 / 96 / 192 MB
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --seq-frames 0 --single 0 --cpu-frames 0 --h2d 0 --steps 20"
for rep in 1 2; do
for v in 0 96 48 192; do
  if [ $v = 96 ]; then L=$PWD/hso_amd/csrc/libhso_gpu.so; else L=$PWD/build/exp/libhso_gpu_chunk$v.so; fi
  HSO_GPU_LIB=$L $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
print('chunk %3s MB: value %.0f  ms_per_step %.3f  tracker launch_ms %.3f' % ('$v', d['value'], d['ms_per_step'], d['roofline']['launch_ms']))"
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 96; do
  if [ $v = 96 ]; then L=$GRAFT_REPO_ROOT/hso_amd/csrc/libhso_gpu.so; else L=$GRAFT_REPO_ROOT/build/exp/libhso_gpu_chunk$v.so; fi
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r6_chunk$v; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r6_chunk$v
  HSO_GPU_LIB=$L PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6_chunk$v -o b -- python $GRAFT_REPO_ROOT/bench.py --seq-frames 0 --single 0 --cpu-frames 0 --h2d 0 --steps 10 > /dev/null 2>&1
  rm -f $GRAFT_REPO_ROOT/gpurun_out/r6_chunk$v/*kernel_trace.csv
  python - <<PY
import csv
rows=list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r6_chunk$v/b_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("k_sobel","k_pyramid","k_frame_stats","k_track")):
        print("chunk $v:", n.replace("void ","").split("(")[0][:40], "calls", r["Calls"], "total ms", float(r["TotalDurationNs"])/1e6, "per step of 4096 (13 steps + setup):", float(r["TotalDurationNs"])/1e6/14)
PY
done
