"""The measurement helpers under tools/ that the profile records rest on (no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_overlap_counts_time_weighted_concurrency(tmp_path):
    """tools/kernel_overlap.py on a hand-made rocprofv3 kernel trace: A runs 0..100, B 50..150 on another queue, then nothing
    until C 300..400 — no kernel in flight for 150 of 400 ns, one for 200, two for 50; A and B are overlapped for half of
    their durations, C never."""
    trace = tmp_path / "trace.csv"
    head = '"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y","Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n'
    row = '"KERNEL_DISPATCH","Agent 2",%d,1,1,%d,1,"%s",1,%d,%d,512,0,8,0,16,256,1,1,1024,1,1\n'
    trace.write_text(head + row % (1, 1, "void k_a<true>(int)", 1000, 1100) + row % (2, 2, "k_b(float*)", 1050, 1150) + row % (1, 3, "k_c()", 1300, 1400))
    out = tmp_path / "out.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_overlap.py"), str(trace), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    d = json.loads(out.read_text())
    assert d["kernels"] == 3 and abs(d["span_ms"] - 400e-6) < 1e-12
    share = d["concurrency_share"]
    assert abs(share["0"] - 150 / 400) < 1e-9 and abs(share["1"] - 200 / 400) < 1e-9 and abs(share["2"] - 50 / 400) < 1e-9
    assert abs(d["mean_in_flight"] - 300 / 400) < 1e-9
    by = {k["name"]: k for k in d["kernels_by_time"]}
    assert set(by) == {"k_a<true>", "k_b", "k_c"}
    assert abs(by["k_a<true>"]["overlapped_frac"] - 0.5) < 1e-9 and abs(by["k_b"]["overlapped_frac"] - 0.5) < 1e-9 and by["k_c"]["overlapped_frac"] == 0
    assert d["queues"]["1"]["kernels"] == 2 and d["queues"]["2"]["kernels"] == 1
    assert by["k_a<true>"]["lds"] == 512 and by["k_a<true>"]["wg"] == 256 and by["k_a<true>"]["grid"] == 1024
