#!/bin/bash
# ThreadSanitizer on the worker pool and on a bank of sequences, AddressSanitizer + UBSan on the same bank:
# the engine's host code (hso_amd/host/hso_engine*.cpp) over tests/fakegpu, no GPU needed.  Logs -> build/sanitize/*.log
#   bash tools/run_sanitizers.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/sanitize
mkdir -p $OUT
cd $ROOT
make -s -C oracle
python - <<PY
import numpy as np, struct, sys
sys.path.insert(0, "$ROOT")
from hso_amd import synth, capi
SMALL = dict(synth.EUROC, width=384, height=256, fx=240.0, fy=240.0, cx=191.5, cy=127.5)
S = synth.sequence(30, spec=SMALL, workers=4, step=(0.05, 0.015, 0.02), rot_deg_per_frame=(0.1, -0.3, 0.08))
cam = synth.camera(SMALL)
with open("$OUT/seq.bin", "wb") as f:
    f.write(struct.pack("<3i", SMALL["width"], SMALL["height"], len(S["images"])))
    f.write(bytes(cam))
    for im in S["images"]:
        f.write(np.ascontiguousarray(im, np.uint8).tobytes())
    f.write(np.ascontiguousarray(S["depth0"], np.float32).tobytes())
print("sequence written")
PY
HOST=hso_amd/host
ENGINE="$HOST/hso_math.cpp $HOST/hso_init.cpp $HOST/hso_engine.cpp $HOST/hso_engine_step.cpp $HOST/hso_engine_kf.cpp $HOST/hso_engine_init.cpp $HOST/hso_engine_c.cpp"
SRC="$ENGINE tests/fakegpu/hso_fake_gpu.cpp hso_amd/csrc/hso_octree.cpp tests/engine_sanitize.cpp"
LINK="-Loracle -lhso_oracle -Wl,-rpath,$ROOT/oracle -pthread"
set -x
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread tests/pool_stress.cpp -o $OUT/pool_stress_tsan || exit 1
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer $SRC $LINK -o $OUT/engine_tsan || exit 1
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer $SRC $LINK -o $OUT/engine_asan || exit 1
set +x
( time $OUT/pool_stress_tsan 6 20000 ) > $OUT/pool_stress_tsan.log 2>&1; echo "pool_stress_tsan rc=$?" | tee -a $OUT/pool_stress_tsan.log
( time HSO_ENGINE_THREADS=4 $OUT/engine_tsan $OUT/seq.bin 4 120 3 ) > $OUT/engine_tsan.log 2>&1; echo "engine_tsan rc=$?" | tee -a $OUT/engine_tsan.log
( time HSO_ENGINE_THREADS=4 ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/engine_asan $OUT/seq.bin 4 120 3 ) > $OUT/engine_asan.log 2>&1; echo "engine_asan rc=$?" | tee -a $OUT/engine_asan.log
tail -n 3 $OUT/pool_stress_tsan.log $OUT/engine_tsan.log $OUT/engine_asan.log
