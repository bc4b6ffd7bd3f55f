"""Previous-frame seed pass on the EuRoC (radtan) camera: device vs restatement, print what differs (GPU box only)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from hso_amd import capi, synth
from oracle import oracle_py as orc

orc.build(); orc.load()
spec = synth.EUROC
cam = synth.camera(spec)
d = synth.config2_pair(300, spec=spec, trans_frac=0.05)
rp, cp = orc.create_pyramid(d["ref"]), orc.create_pyramid(d["cur"])
sob = [orc.sobel5(cp[l]) for l in range(3)]
gx0, gy0 = orc.sobel5(rp[0])
seeds, T_cur, feats = synth.seeds_for_pair(d, 300, 9101, gx=gx0, gy=gy0)
pea = math.atan(1.0 / (2.0 * 458.0)) * 2.0
ctx = capi.Context(0)
ctx.frame_upload(9101, d["ref"]); ctx.frame_upload(9102, d["cur"])
tab = ctx.seed_table_create()
ctx.seed_table_append(tab, seeds)
brief, full = ctx.seed_table_observe_previous(cam, tab, [(9101, (9102, T_cur, 1.05))], pea, want_full=True)
nd = 0
for k, s in enumerate(seeds):
    o = orc.seed_observe_previous(cam, s, T_cur, 1.05, pea, rp, cp, sob)
    g = full[k]
    if (g.is_update, g.result, g.n_steps, g.zmncc_best, g.zmncc_second) != (o.is_update, o.result, o.n_steps, o.zmncc_best, o.zmncc_second):
        nd += 1
        if nd <= 12:
            print(k, "gpu", g.is_update, g.result, g.n_steps, g.search_level, repr(g.zmncc_best), repr(g.zmncc_second), "| cpu", o.is_update, o.result, o.n_steps, o.search_level, repr(o.zmncc_best), repr(o.zmncc_second))
print("differing", nd, "of", len(seeds))
