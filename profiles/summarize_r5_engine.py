"""Round 5: the engine's own launches — per-stage roofline and SQ counters.
  profiles/r5_stage_roofline.csv   per stage of the resident chain: kernel time per step of 128 sequences (rocprofv3 --kernel-trace --stats of
                                   one engine, profiles/r5_bank_solo_kernel_stats.csv), algorithmic bytes per step (SURVEY.md section 8(d),
                                   accumulated by the engine itself: hso_vo_multi_alg_bytes), achieved GB/s and the fraction of 8 TB/s
  profiles/r5_engine_sq_counters.csv  per kernel: VALU-busy, wait and LDS fractions from the SQ counters of the same launches
usage: python profiles/summarize_r5_engine.py"""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
G = os.path.join(ROOT, "gpurun_out")

STAGES = {   # stage -> (alg-bytes key of hso_vo_multi_alg_bytes, kernel-name patterns)
    "frame construction": ("frame", [r"^k_pyramid", r"^k_sobel", r"^k_frame_stats"]),
    "tracker": ("track", [r"k_track"]),
    "reprojection + matching": ("match", [r"^k_chain_list", r"^k_chain_reproject", r"^k_chain_visit"]),   # + the chain's share of k_align_t (below)
    "pose optimisation": ("pose", [r"^k_pose"]),
    "seed observation": ("seed", [r"^k_seed_"]),
}


def main():
    rows = list(csv.DictReader(open(os.path.join(P, "r5_bank_solo_kernel_stats.csv"))))
    steps = 120
    t = {r["Name"].replace("void ", "").split("(")[0]: (float(r["TotalDurationNs"]) / steps * 1e-6, int(r["Calls"])) for r in rows}
    line = [l for l in open(os.path.join(G, "r5_bank_solo", "banks.log")) if l.startswith("{")][-1]
    alg = json.loads(line)["alg_bytes_per_frame"]
    n_seq = 128
    out = []
    for stage, (key, pats) in STAGES.items():
        ms = sum(v[0] for k, v in t.items() if any(re.search(p, k) for p in pats))
        if stage == "reprojection + matching":
            a = [v for k, v in t.items() if k.startswith("k_align_t")]
            # k_align_t runs for the chain (every step) and for the activation (the steps with converged seeds): equal launches, the chain's half
            ms += sum(v[0] for v in a) * 0.5
        b = alg[key] * n_seq
        out.append((stage, ms, b, b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0))
    with open(os.path.join(P, "r5_stage_roofline.csv"), "w") as f:
        f.write("stage,kernel_ms_per_step_of_128_sequences,algorithmic_bytes_per_step,achieved_GB_per_s,frac_of_8_TB_per_s\n")
        for s, ms, b, g in out:
            f.write("%s,%.4f,%.0f,%.1f,%.4f\n" % (s, ms, b, g, g / 8000.0))
        tot_ms = sum(v[0] for v in t.values()); tot_b = sum(alg.values()) * n_seq
        f.write("all kernels + copies of a step,%.4f,%.0f,%.1f,%.4f\n" % (tot_ms, tot_b, tot_b / (tot_ms * 1e-3) / 1e9, tot_b / (tot_ms * 1e-3) / 1e9 / 8000.0))
    print(open(os.path.join(P, "r5_stage_roofline.csv")).read())
    src = os.path.join(G, "r5_engine_sq", "engine_sq_per_kernel.csv")
    if not os.path.exists(src):
        print("no SQ counters collected (gpurun_out/r5_engine_sq)")
        return
    c = {}
    for r in csv.DictReader(open(src)):
        c.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))
    with open(os.path.join(P, "r5_engine_sq_counters.csv"), "w") as f:
        # valu_busy_frac as profiles/summarize.py forms it for k_track: 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32) — of the
        # WHOLE chip's SIMDs: a launch of 128 workgroups cannot exceed what its share of the CUs allows
        f.write("kernel,dispatches,valu_busy_frac,wait_any_frac_of_wave_cycles,wait_lds_frac_of_wave_cycles,valu_insts_per_dispatch,lds_insts_per_dispatch,lds_bank_conflict_cycles_per_active_lds_cycle,vmem_rd_per_dispatch,waves_per_dispatch\n")
        for k in sorted(c, key=lambda k: -c[k].get("SQ_BUSY_CYCLES", (0, 0))[0] * c[k].get("SQ_BUSY_CYCLES", (0, 0))[1]):
            g = lambda n: c[k].get(n, (0.0, 0))[0]
            busy, wave = g("SQ_BUSY_CYCLES"), g("SQ_WAVE_CYCLES")
            if busy <= 0:
                continue
            f.write("%s,%d,%.3f,%.3f,%.4f,%.4g,%.4g,%.3f,%.4g,%.4g\n" % (
                k, c[k]["SQ_BUSY_CYCLES"][1], 4 * g("SQ_ACTIVE_INST_VALU") / (1024 * busy / 32) if busy else 0, g("SQ_WAIT_ANY") / wave if wave else 0,
                g("SQ_WAIT_INST_LDS") / wave if wave else 0, g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"),
                g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS") if g("SQ_ACTIVE_INST_LDS") else 0, g("SQ_INSTS_VMEM_RD"), g("SQ_WAVES")))
    print(open(os.path.join(P, "r5_engine_sq_counters.csv")).read()[:3000])


if __name__ == "__main__":
    main()
