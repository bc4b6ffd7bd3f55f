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


def hot_lines(path, k=6):
    """Print the lines of `path` whose tokens take part in shingles that also occur in the reference (where to look first)."""
    per_ref = [shingles(p) for p in ref_files()]
    allref = set().union(*per_ref)
    src = open(path, errors="ignore").read()
    clean = strip(src)
    # token -> line number (approximate: searched sequentially in the comment-stripped text, which keeps line breaks of code lines)
    toks, lines, pos = [], [], 0
    for m in TOK.finditer(clean):
        toks.append(m.group(0)); lines.append(clean.count("\n", 0, m.start()) + 1)
    hits = {}
    for i in range(len(toks) - k + 1):
        if tuple(toks[i:i + k]) in allref:
            for j in range(i, i + k):
                hits[lines[j]] = hits.get(lines[j], 0) + 1
    text = clean.split("\n")
    for ln in sorted(hits):
        if hits[ln] >= 8:
            print("%5d %3d  %s" % (ln, hits[ln], text[ln - 1].strip()[:150]))
