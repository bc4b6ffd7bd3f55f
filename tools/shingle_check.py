"""Development aid: 6-token shingle overlap of every source file of this repo with the reference tree (comments and
whitespace stripped), the measure the review applies to find renamed / reflowed copies.  Needs /root/reference (this
container only).  Usage: python tools/shingle_check.py [path ...]"""
import os
import re
import sys

REF = "/root/reference"
TOK = re.compile(r"[A-Za-z_][A-Za-z_0-9]*|\d+\.?\d*(?:[eE][-+]?\d+)?[fFuUlL]*|->|::|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|[-+*/%=<>!&|^~?:;,.(){}\[\]#]")


def strip(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return src


def shingles(path, k=6):
    try:
        t = TOK.findall(strip(open(path, errors="ignore").read()))
    except OSError:
        return set()
    return {tuple(t[i:i + k]) for i in range(len(t) - k + 1)}


def ref_files():
    for d, _, fs in os.walk(REF):
        if "/.git" in d:
            continue
        for f in fs:
            if f.endswith((".cpp", ".h", ".hpp", ".c", ".cc")):
                yield os.path.join(d, f)


def main(argv):
    paths = argv or []
    if not paths:
        for d, _, fs in os.walk("."):
            if any(s in d for s in ("/.git", "/build", "/gpurun_out", "/__pycache__")):
                continue
            for f in fs:
                if f.endswith((".cpp", ".h", ".hip", ".c", ".py")):
                    paths.append(os.path.join(d, f))
    per_ref = {p: shingles(p) for p in ref_files()}
    allref = set().union(*per_ref.values()) if per_ref else set()
    rows = []
    for p in sorted(paths):
        s = shingles(p)
        if len(s) < 50:
            continue
        hit = s & allref
        best = max(per_ref.items(), key=lambda kv: len(s & kv[1]))
        rows.append((len(hit) / len(s), p, len(s), os.path.relpath(best[0], REF), len(s & best[1]) / len(s)))
    for frac, p, n, bf, bfrac in sorted(rows, reverse=True):
        print("%5.1f%%  %-44s %6d shingles; closest %s (%.1f%%)" % (100 * frac, p, n, bf, 100 * bfrac))


if __name__ == "__main__":
    main(sys.argv[1:])
