"""bench.py's N > 1 path without hardware: `python bench.py --gpus 2` launches its two ranks itself (torch.distributed.run), each
rank shards the distinct scenes, times its steps between barriers, gathers the per-frame records and — through the sequence
engine — the trajectories of its sequences, and rank 0 prints the one JSON line.  The device side is swapped for the CPU restatement
and gloo (tests/bench_cpu_side.py), so only the plumbing is under test: the first real 8-GPU run must not fail on it."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_self_launch_and_gathers(gpus, orc, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "fakegpu")])
    env = dict(os.environ, HSO_BENCH_SIDE="bench_cpu_side:CpuSide", PYTHONPATH=HERE + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               MASTER_ADDR="127.0.0.1", HSO_BENCH_DETAIL=str(tmp_path / "bench_detail.json"))   # not the tracked file
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--batch", "4", "--scenes", "2",
           "--feats", "150", "--shape", "vga", "--cpu-frames", "0", "--sequences", "2", "--banks", "2", "--seq-feats", "60", "--seq-frames", "4", "--seq-distinct", "2", "--single", "0", "--seq-warmup-frames", "0"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 4096
    out = json.loads(line)
    assert out["n_gpus"] == gpus and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert len(out["per_gpu_frames_per_s"]) == gpus and out["value"] > 0 and out["unit"] == "frames/s"
    assert out["config"]["frames_per_gpu_per_step"] == 4 and out["roofline"]["frac"] > 0
    assert out["sequences_frames_per_s"] > 0 and out["sequences_failures"] == 0
    # the host budget: every bank's pool is its share of the CPU quota (ranks of the node x banks of the process), never the whole
    assert out["host_cpu_quota"] >= 1 and 1 <= out["threads_per_bank"] <= max(1, (7 * out["host_cpu_quota"]) // (2 * gpus * 2))
    detail = json.load(open(tmp_path / "bench_detail.json"))
    # both gathers: [world, records per rank, 8]
    assert detail["sequences"]["gathered_trajectory_shape"] == [gpus, 2 * 2 * 4, 8] and detail["sequences"]["sequences_total"] == gpus * 4
