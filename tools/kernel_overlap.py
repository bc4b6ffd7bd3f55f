"""What runs beside what: time-weighted concurrency of a rocprofv3 --kernel-trace CSV (several engines on one GPU).
`python tools/kernel_overlap.py <kernel_trace.csv[.gz]> [out.json]` — prints the share of the traced span spent with k kernels in
flight, the busy time per hardware queue, and for the heaviest kernels how much of their own duration another kernel overlapped."""
import csv
import gzip
import json
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").strip()


def main():
    path = sys.argv[1]
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"),
                         int(r.get("LDS_Block_Size", 0) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # drop the set-up: start at the first tracker launch of the run's second half
    ev = []
    for i, r in enumerate(rows):
        ev.append((r[0], 1, i))
        ev.append((r[1], -1, i))
    ev.sort()
    active = set()
    last = ev[0][0]
    hist = defaultdict(int)
    overlapped = defaultdict(int)      # per kernel name: ns of its duration with >= 1 other kernel in flight
    weighted_peers = defaultdict(float)  # per kernel name: integral of (number of other kernels in flight)
    dur = defaultdict(int)
    calls = defaultdict(int)
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            k = len(active)
            hist[k] += dt
            for j in active:
                nm = rows[j][2]
                dur[nm] += dt
                if k > 1:
                    overlapped[nm] += dt
                    weighted_peers[nm] += dt * (k - 1)
        last = t
        if d > 0:
            active.add(i)
            calls[rows[i][2]] += 1
        else:
            active.discard(i)
    span = t1 - t0
    queues = defaultdict(lambda: [0, 0])
    for r in rows:
        queues[r[3]][0] += 1
        queues[r[3]][1] += r[1] - r[0]
    out = dict(span_ms=span * 1e-6, kernels=len(rows),
               concurrency_share={str(k): v / span for k, v in sorted(hist.items())},
               mean_in_flight=sum(k * v for k, v in hist.items()) / span,
               queues={q: dict(kernels=c, busy_ms=b * 1e-6) for q, (c, b) in sorted(queues.items())},
               kernels_by_time=[dict(name=n, calls=calls[n], total_ms=dur[n] * 1e-6, mean_us=dur[n] * 1e-3 / max(calls[n], 1),
                                     overlapped_frac=overlapped[n] / max(dur[n], 1), mean_peers=weighted_peers[n] / max(dur[n], 1))
                                for n in sorted(dur, key=lambda n: -dur[n])[:30]])
    shapes = {}
    for r in rows:
        shapes.setdefault(r[2], dict(lds=r[4], wg=r[5], grid=r[6]))
    for k in out["kernels_by_time"]:
        k.update(shapes.get(k["name"], {}))
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
