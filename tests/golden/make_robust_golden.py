"""Generate tests/golden/robust_cost.json from the COMPILED REFERENCE
(oracle/_ref/librobust_cost_ref.so = /root/reference/src/vikit/robust_cost.cpp built
by oracle/Makefile).  Run in the authoring container only (the reference does not
travel): `python tests/golden/make_robust_golden.py`.  The JSON holds inputs and the
reference's outputs as exact float32 bit patterns; no reference source is stored.
"""
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402


def bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def main():
    oracle_py.build()
    ref = oracle_py.load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(20260928)
    xs = np.concatenate([rng.normal(0, 3, 200), rng.uniform(-20, 20, 56),
                         [0.0, 1.345, -1.345, 1.3449999, 4.6851, -4.6851, 1e-8, 1e8]]).astype(np.float32)
    out = {"huber_k": bits(ref.ref_huber_default_k()), "tukey_b": bits(ref.ref_tukey_default_b()),
           "x": [bits(x) for x in xs], "huber": [], "tukey": [], "tdist": [], "mad": [], "tdist_scale": []}
    k, b = ref.ref_huber_default_k(), ref.ref_tukey_default_b()
    for x in xs:
        out["huber"].append(bits(ref.ref_huber_weight(k, float(x))))
        out["tukey"].append(bits(ref.ref_tukey_weight(b, float(x))))
        out["tdist"].append(bits(ref.ref_tdist_weight(5.0, float(x))))
    for n in (1, 2, 3, 10, 11, 100, 1001):
        e = np.abs(rng.normal(0, 2, n)).astype(np.float32)
        out["mad"].append({"errors": [bits(v) for v in e], "scale": bits(ref.ref_mad_scale(e.ctypes.data, n))})
        if n >= 10:
            out["tdist_scale"].append({"errors": [bits(v) for v in e],
                                       "scale": bits(ref.ref_tdist_scale(5.0, e.ctypes.data, n))})
    with open(os.path.join(os.path.dirname(__file__), "robust_cost.json"), "w") as f:
        json.dump(out, f)
    print("wrote robust_cost.json:", len(xs), "weights,", len(out["mad"]), "MAD cases")


if __name__ == "__main__":
    main()
