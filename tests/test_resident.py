"""Resident seed tables behind handles: same kernels as the value-passing calls, state kept in HBM between calls — results must
equal the value-passing entry points bit for bit, and the state must evolve like the host-side copy.  (The sequence maps and the
resident per-frame chain have their own tests: tests/test_seq_chain.py.)"""
import ctypes as C

import numpy as np
import pytest

from hso_amd import capi, synth

pytestmark = pytest.mark.gpu
PX_ERROR_ANGLE = 2 * np.arctan(1.0 / (2.0 * 480.6))


def test_seed_table_matches_value_passing_calls(gpu_ctx, cam, pair2000):
    d = pair2000
    gpu_ctx.frame_upload(9301, d["ref"]); gpu_ctx.frame_upload(9302, d["cur"]); gpu_ctx.frame_upload(9303, d["cur"])
    try:
        seeds, T_cur, _ = synth.seeds_for_pair(d, 700, 9301, seed=5)
        t = gpu_ctx.seed_table_create()
        first = gpu_ctx.seed_table_append(t, seeds[:400])
        assert first == 0 and gpu_ctx.seed_table_append(t, seeds[400:]) == 400 and gpu_ctx.seed_table_size(t) == (700, 700)
        host = [capi.Seed.from_buffer_copy(bytes(s)) for s in seeds]          # the value-passing side keeps its own state
        T2 = capi.SE3.from_arrays(T_cur.q[:], np.array(T_cur.t[:]) * 1.3)
        alive = np.ones(700, bool)
        for rnd, (fid, T, expo) in enumerate([(9302, T_cur, 1.05), (9303, T2, 1.02), (9302, T_cur, 1.05)]):
            brief, full = gpu_ctx.seed_table_observe(cam, t, [(fid, T, expo)], PX_ERROR_ANGLE, want_full=(rnd == 1))
            ref = gpu_ctx.seed_observe(cam, fid, T, expo, PX_ERROR_ANGLE, host)
            for i, (s, o) in enumerate(zip(host, ref)):
                if not alive[i]:
                    assert brief[i]["result"] == 0 and brief[i]["mu"] == 0
                    continue
                b = brief[i]
                assert (b["mu"], b["sigma2"], b["b"], b["result"], b["is_update"], b["is_valid"], b["search_level"]) == \
                       (o.mu, o.sigma2, o.b, o.result, o.is_update, o.is_valid, o.search_level), (rnd, i)
                if full is not None:
                    assert bytes(full[i]) == bytes(o)
                s.mu, s.sigma2, s.b = o.mu, o.sigma2, o.b                      # what DepthFilter::updateSeed leaves in the seed
            if rnd == 0:                                                       # erase every 7th seed: slots keep their index
                gone = np.arange(0, 700, 7)
                gpu_ctx.seed_table_erase(t, gone); alive[gone] = False
                assert gpu_ctx.seed_table_size(t) == (700, 600)
        # the table caches the host frame's base pointer per live seed: that frame cannot be released under it
        with pytest.raises(capi.HsoGpuError, match="live seeds"):
            gpu_ctx.frame_release(9301)
        back = gpu_ctx.seed_table_read(t, 0, 700)
        for i in np.where(alive)[0][:50]:
            assert (back[i].mu, back[i].sigma2, back[i].b) == (host[i].mu, host[i].sigma2, host[i].b)
        # compaction: the 600 live records move to slots 0..599 in order, the table shrinks, and the next observation equals the
        # value-passing call over the live seeds alone
        n_new, remap = gpu_ctx.seed_table_compact(t)
        live = np.where(alive)[0]
        assert n_new == 600 and gpu_ctx.seed_table_size(t) == (600, 600)
        assert np.array_equal(remap[live], np.arange(600)) and (remap[~alive] == -1).all()
        moved = gpu_ctx.seed_table_read(t, 0, 600)
        assert all(bytes(moved[k]) == bytes(back[i]) for k, i in enumerate(live))
        brief, _ = gpu_ctx.seed_table_observe(cam, t, [(9303, T2, 1.02)], PX_ERROR_ANGLE)
        ref = gpu_ctx.seed_observe(cam, 9303, T2, 1.02, PX_ERROR_ANGLE, [host[i] for i in live])
        assert len(brief) == 600
        assert all((b["mu"], b["sigma2"], b["b"], b["result"]) == (o.mu, o.sigma2, o.b, o.result) for b, o in zip(brief, ref))
        assert gpu_ctx.seed_table_compact(t)[0] == 600                         # nothing erased: a no-op
        assert gpu_ctx.seed_table_append(t, seeds[:5]) == 600                  # appends continue behind the compacted records
        gpu_ctx.seed_table_destroy(t)
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.seed_table_size(t)
    finally:
        for i in (9301, 9302, 9303):
            gpu_ctx.frame_release(i)


def test_seed_table_groups_serve_many_sequences(gpu_ctx, cam, pair2000):
    d = pair2000
    gpu_ctx.frame_upload(9311, d["ref"]); gpu_ctx.frame_upload(9312, d["cur"])
    try:
        seeds, T_cur, _ = synth.seeds_for_pair(d, 120, 9311, seed=9)
        t = gpu_ctx.seed_table_create()
        gpu_ctx.seed_table_append(t, seeds * 3, group=np.repeat(np.arange(3), 120))
        T_b = capi.SE3.from_arrays(T_cur.q[:], np.array(T_cur.t[:]) * 0.5)
        brief, _ = gpu_ctx.seed_table_observe(cam, t, [(9312, T_cur, 1.05), (9312, T_b, 1.05), (9312, T_cur, 1.05)], PX_ERROR_ANGLE)
        assert np.array_equal(brief[:120], brief[240:]) and not np.array_equal(brief[:120], brief[120:240])
        solo = gpu_ctx.seed_observe(cam, 9312, T_b, 1.05, PX_ERROR_ANGLE, seeds)
        assert [o.result for o in solo] == list(brief[120:240]["result"])
        with pytest.raises(capi.HsoGpuError):
            gpu_ctx.seed_table_observe(cam, t, [(9312, T_cur, 1.05)], PX_ERROR_ANGLE)       # groups 1, 2 have no frame
        gpu_ctx.seed_table_destroy(t)
    finally:
        gpu_ctx.frame_release(9311); gpu_ctx.frame_release(9312)
