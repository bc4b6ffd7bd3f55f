import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from hso_amd import bank_bench, synth
seqs = synth.sequences(8, 61, spec=synth.EUROC, seed0=777)
for r in range(3):
    sys.stderr.write("=== run %d\n" % r); sys.stderr.flush()
    m = bank_bench.run(128, 61, 2000, seqs=seqs)
    print(json.dumps(dict(run=r, fps=m["frames_per_s"], ms_mean=m["ms_per_step_mean"], ms_median=m["ms_per_step_median"], ms_max=m["ms_per_step_max"])), flush=True)
