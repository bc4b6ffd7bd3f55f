"""Bank load of the tracker's LDS row reads versus the row stride of the staged level image (CPU simulation).
A wave instruction of the pixel loop reads, in every lane, two dwords (ds_read2_b32) of one row of the lane's feature window; lanes
0-31 and 32-63 are served separately, 32 banks of 4 B; extra cycles = (largest number of distinct dwords on one bank) - 1."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from hso_amd import synth

def conflicts(addr_dw):
    extra = 0
    for half in (addr_dw[:32], addr_dw[32:]):
        for k in range(2):                      # the two dwords of a ds_read2_b32 are separate accesses
            a = np.unique(half + k)
            extra += np.bincount(a % 32, minlength=32).max() - 1
    return extra

d = synth.config2_pair(2000, spec=synth.EUROC)
px = np.array([[f.px[0], f.px[1]] for f in d["feats"]]) if hasattr(d["feats"][0], "px") else None
if px is None:
    rng = np.random.default_rng(3); px = np.column_stack([rng.uniform(20, 730, 2000), rng.uniform(20, 460, 2000)])
out = []
for level in (1, 2, 3):
    cols = [752, 376, 188, 94, 47][level]
    u = np.floor(px[:, 0] / (1 << level)).astype(int); v = np.floor(px[:, 1] / (1 << level)).astype(int)
    n = len(u)
    for name, order in (("table order", np.arange(n)), ("features dealt round-robin over the 32 banks of their row-0 dword", None)):
        for stride in (cols, (cols + 3) & ~3, ((cols + 3) & ~3) + 4, ((cols + 3) & ~3) + 12, ((cols + 63) & ~63) + 4):
            base = v * stride + u - 3
            if order is None:
                bank = (base >> 2) % 32
                buckets = [list(np.nonzero(bank == b)[0]) for b in range(32)]
                o = []
                while any(buckets):
                    for b in range(32):
                        if buckets[b]: o.append(buckets[b].pop())
                idx = np.array(o)
            else:
                idx = order
            tot = cnt = 0
            for w0 in range(0, n - 63, 64):
                lanes = idx[w0:w0 + 64]
                for R in range(-4, 6):          # the rows of the 21-pixel pattern
                    tot += conflicts((base[lanes] + R * stride) >> 2); cnt += 1
            out.append("level %d  %-70s stride %4d: %.2f extra LDS cycles per ds_read2_b32" % (level, name, stride, tot / cnt))
print("\n".join(out))
