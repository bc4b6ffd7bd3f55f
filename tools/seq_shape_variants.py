"""End-to-end sequence throughput (bench.py's `sequences` shape) under the tracker-shape knobs of a shared device, one process,
one rendering: `python tools/seq_shape_variants.py [out.json] [frames]` on the GPU box from the repo root.
Every variant is a fresh set of engines (hso_amd.bank_bench.run_banks); the knobs are read by the library per call."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = [
    ("plain", 6, 128, {}),
    ("plain8", 8, 128, {}),
    ("plain7", 7, 128, {}),
    ("plain5", 5, 128, {}),
    ("plain6x160", 6, 160, {}),
    ("plain_again", 6, 128, {}),
    ("plain12x64", 12, 64, {}),
    ("default", 6, 128, {}),
    ("split", 6, 128, {"HSO_TRACK_SPLIT_MIN_JOBS": "1"}),
    ("split_all_trk2", 6, 128, {"HSO_TRACK_SPLIT_MIN_JOBS": "1", "HSO_TRACK_ALL_TRK2": "1"}),
    ("split_8x128", 8, 128, {"HSO_TRACK_SPLIT_MIN_JOBS": "1"}),
    ("shared_coop", 6, 128, {"HSO_TRACK_SHARED_COOP": "1"}),
    ("default_again", 6, 128, {}),
    ("split_again", 6, 128, {"HSO_TRACK_SPLIT_MIN_JOBS": "1"}),
]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/seq_shape_variants.json"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 121
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 777          # bench.py renders its sequences from 2024 + 1000 * rank
    from hso_amd import synth, bank_bench
    import pickle
    cache = "/tmp/hso_seqs_%d_%d.pkl" % (frames, seed0)          # one rendering for several processes of a box
    if os.path.exists(cache):
        seqs = pickle.load(open(cache, "rb"))
    else:
        seqs = [dict(images=q["images"], depth0=q["depth0"], T_f_w=q["T_f_w"]) for q in synth.sequences(8, frames, spec=synth.EUROC, seed0=seed0)]
        pickle.dump(seqs, open(cache, "wb"))
    rows = []
    for name, banks, n, env in VARIANTS:
        if only and name not in only:
            continue
        for k, v in env.items():
            os.environ[k] = v
        try:
            r = bank_bench.run_banks(banks, n, frames, 2000, seqs=seqs)
        finally:
            for k in env:
                del os.environ[k]
        row = dict(variant=name, banks=banks, sequences_per_bank=n, env=env, steady=r.get("steady_frames_per_s"), whole=r["frames_per_s"],
                   warmup=r.get("warmup_frames_per_s"), failures=r["failures"], trans_err_max=r["trans_err_max"],
                   keyframes=r["keyframes_per_sequence"], gpu_busy=r.get("steady_gpu_busy_frac"), cpus=r.get("host_cpus_used"), rusage=r.get("host_rusage"), wall_s=r["wall_s"])
        rows.append(row)
        print(json.dumps(row), flush=True)
        json.dump(rows, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
