"""Scan the scenes of tests/test_parity_gpu.py::test_accept_decisions_over_many_scenes for frames whose device accept sequence
differs from the fp64-sum restatement's, and print both sides' energies (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from hso_amd import capi, synth
from oracle import oracle_py


def main(shape="euroc", n_scenes=64):
    oracle_py.build(); oracle_py.load(); orc = oracle_py
    ctx = capi.Context(0)
    spec = synth.EUROC if shape == "euroc" else synth.ICL_NUIM
    camS = synth.camera(spec)
    p = capi.TrackParams(0, 4, 1, 50)
    rng = np.random.default_rng(5)
    for k in range(n_scenes):
        d = synth.config2_pair(600, spec=spec, seed=4000 + 13 * k, exposure=float(rng.uniform(0.92, 1.08)),
                               trans_frac=float(rng.uniform(0.012, 0.028)), rot_deg=float(rng.uniform(0.3, 0.7)))
        for i in (3, 4):
            try:
                ctx.frame_release(i)
            except capi.HsoGpuError:
                pass
        ctx.frame_upload(3, d["ref"]); ctx.frame_upload(4, d["cur"])
        rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
        starts = [(capi.SE3.from_arrays(synth.rotvec_to_quat(rng.normal(0, np.deg2rad(0.05), 3)), rng.uniform(0.5, 1.2) * np.array(d["t_true"])),
                   float(np.float32(rng.uniform(0.95, 1.05)))) for _ in range(4)]
        jobs = [ctx.make_job(3, 4, d["feats"], T0, a0) for T0, a0 in starts]
        got = ctx.coarse_track_batch(camS, p, jobs)
        solo = [ctx.coarse_track_batch(camS, p, [j])[0] for j in jobs]
        for q, ((T0, a0), rg) in enumerate(zip(starts, got)):
            t64 = orc.Tracker(camS, p, rp, cp, d["feats"]); t64.decide_on_f64_sum(True)
            r64 = t64.run(T0, a0)
            same = list(rg.iters) == list(r64.iters) and list(rg.accept_mask) == list(r64.accept_mask)
            same_solo = list(solo[q].iters) == list(rg.iters) and list(solo[q].accept_mask) == list(rg.accept_mask)
            if not same or not same_solo:
                print("scene", k, "start", q, "iters", list(rg.iters), list(r64.iters), "mask", list(rg.accept_mask), list(r64.accept_mask),
                      "solo mask", list(solo[q].accept_mask))
                print("  energy gpu ", [repr(float(e)) for e in rg.energy])
                print("  energy solo", [repr(float(e)) for e in solo[q].energy])
                print("  energy r64 ", [repr(float(e)) for e in r64.energy])
                print("  n_eval", list(rg.n_eval), list(r64.n_eval), "coop", rg.coop_workgroups)
                for lvl in (4, 3, 2, 1):
                    ge, _, _, _ = ctx.tracker_eval(camS, p, jobs[q], lvl, r64.T_cur_ref, r64.exposure_rat, huber=float(r64.huber[lvl]), outlier=float(r64.outlier[lvl]))
                    t64.set_level(lvl); t64.set_thresholds(float(r64.huber[lvl]), float(r64.outlier[lvl]))
                    eo = t64.eval(r64.T_cur_ref, r64.exposure_rat)
                    print("  level", lvl, "eval at r64 pose: gpu sum %r n %d | oracle f64 %r n %d" % (float(ge.energy_sum), ge.n_terms, t64.energy_f64(), eo.n_terms))
    print("scan done")


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["euroc"]), *(int(a) for a in sys.argv[2:3]))
