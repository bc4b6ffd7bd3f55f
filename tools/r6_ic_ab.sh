#!/bin/bash
# A/B of the inverse-compositional pixel loop (TRK_IC_ROWS): the tracker alone in bench.py's batch shape, then the engine end to end.
#   build/exp/libhso_gpu_noicrows.so = the same library with hso_tracker*.hip compiled -DTRK_IC_ROWS=0 (see profiles/r6_ic_rows.md)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/r6_ic; mkdir -p $OUT
B="python bench.py --inverse 1 --seq-frames 0 --single 0 --cpu-frames 0 --h2d 0 --steps 10"
$B > $OUT/ic_rows.json 2> $OUT/ic_rows.err
HSO_GPU_LIB=$PWD/build/exp/libhso_gpu_noicrows.so $B > $OUT/ic_generic.json 2> $OUT/ic_generic.err
$B > $OUT/ic_rows_2.json 2>> $OUT/ic_rows.err
for f in ic_rows ic_generic ic_rows_2; do python - <<PY
import json
d=json.loads([l for l in open("$OUT/$f.json") if l.startswith("{")][-1])
print("$f", "value %.0f  launch_ms %.3f  frac %.3f" % (d["value"], d["roofline"]["launch_ms"], d["roofline"]["frac"]))
PY
done
python tools/r6_repeat_banks.py 3 6 128 2>&1 | grep "^{" | sed 's/^/rows /'
cp hso_amd/csrc/libhso_gpu.so $OUT/libhso_gpu_main.so; cp build/exp/libhso_gpu_noicrows.so hso_amd/csrc/libhso_gpu.so
python tools/r6_repeat_banks.py 3 6 128 2>&1 | grep "^{" | sed 's/^/generic /'
cp $OUT/libhso_gpu_main.so hso_amd/csrc/libhso_gpu.so; rm $OUT/libhso_gpu_main.so
