"""ZMNCC_F<4> (include/hso/vikit/patch_score.h:268-305), the score Matcher::doLineStereo ranks its
epipolar candidates with (src/matcher.cpp:918-960).  PINNED by the reference's own code: the header
compiles standalone (oracle/_ref/libpatch_score_ref.so) and tests/golden/zmncc.json holds its
outputs (tests/golden/make_zmncc_golden.py); the restatement in oracle/hso_oracle_seed.c — the one
the seed-observation parity tests check the HIP path against — reproduces every case bit for bit."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "zmncc.json")


def _cases():
    for c in json.load(open(GOLD))["cases"]:
        yield (c["name"], np.frombuffer(bytes.fromhex(c["host"]), np.float32).copy(),
               np.frombuffer(bytes.fromhex(c["target"]), np.float32).copy(), c["score_bits"])


def test_oracle_zmncc_matches_reference_golden(orc):
    n = 0
    scores = []
    for name, host, target, bits in _cases():
        got = np.float32(orc.zmncc_f8(host, target))
        assert int(got.view(np.uint32)) == bits, name
        scores.append(float(got))
        n += 1
    assert n == 64
    by_name = {c[0]: s for c, s in zip(_cases(), scores)}
    assert by_name["identical"] > 0.999999 and by_name["negated"] < -0.999999
    assert by_name["flat_host"] == 0.0 and by_name["flat_both"] == 0.0       # zero variance: 0 / (0 + 1e-12)
    assert sum(s > 0.8 for s in scores[:40]) >= 5 and min(scores[:40]) < 0.5   # shifted scene patches: matches and mismatches


def test_oracle_zmncc_matches_live_reference_when_present(orc):
    import pytest
    rng = np.random.default_rng(4)
    if orc.ref_zmncc_f8(np.zeros(64, np.float32), np.zeros(64, np.float32)) is None:
        pytest.skip("oracle/_ref/libpatch_score_ref.so absent (reference not on this machine)")
    for _ in range(500):
        h = rng.uniform(0, 255, 64).astype(np.float32)
        t = (h * np.float32(rng.uniform(0.5, 1.5)) + rng.normal(0, rng.uniform(0, 60), 64)).astype(np.float32)
        a, b = np.float32(orc.zmncc_f8(h, t)), np.float32(orc.ref_zmncc_f8(h, t))
        assert a.view(np.uint32) == b.view(np.uint32)
