"""side-by-side per-layer conv times of tools/prof_convs.py outputs: python tools/cmp_convs.py a.txt b.txt ..."""
import sys, re
def rows(f):
    out = []
    for l in open(f):
        m = re.search(r"([\d.]+) us +([\d.]+) TF", l)
        if not m or not (l.startswith("in ") or l.startswith("convT")): continue
        key = re.sub(r" +", " ", l[:l.index(m.group(0))]).strip()
        out.append((key, float(m.group(1)), float(m.group(2))))
    return out
tabs = [rows(f) for f in sys.argv[1:]]
keys = []
for t in tabs:
    for k, _, _ in t:
        if k not in keys: keys.append(k)
for k in keys:
    if not re.search(r"cout +(128|256|512) |^convT", k): continue
    cells = []
    for t in tabs:
        v = [r for r in t if r[0] == k]
        cells.append("%7.1f(%5.1f)" % (sum(r[1] for r in v), v[0][2]) if v else "      -       ")
    print("%-62s %s" % (k[:62], " ".join(cells)))
