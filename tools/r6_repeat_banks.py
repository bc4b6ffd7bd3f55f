"""python tools/r6_repeat_banks.py [runs] [banks] [sequences]: the end-to-end run (bank_bench.run_banks) several times in ONE process on the same rendered
sequences: is the first set of engines of a process slower than the following ones (bench.py's end-to-end figure is the first)?"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from hso_amd import bank_bench, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
runs = int(args[0]) if len(args) > 0 else 3
banks = int(args[1]) if len(args) > 1 else 6
nseq = int(args[2]) if len(args) > 2 else 128
seqs = synth.sequences(8, 121, spec=synth.EUROC, seed0=777)
empty = "--empty-cache" in sys.argv
for r in range(runs):
    if empty:
        import gc
        gc.collect(); torch.cuda.empty_cache()      # is the slow first half of a process's later runs tied to torch's cached blocks?
    t0 = time.time()
    m = bank_bench.run_banks(banks, nseq, 121, 2000, seqs=seqs)
    print(json.dumps(dict(run=r, banks=banks, sequences=nseq, steady=m.get("steady_frames_per_s"), whole=m["frames_per_s"], warmup=m.get("warmup_frames_per_s"),
                          cpus=m.get("host_cpus_used"), throttled=m.get("host_throttled_periods"), set_up_s=time.time() - t0 - m["wall_s"])), flush=True)
