"""Reduce the two SQ counter passes of tools/pmc_sq.sh: per kernel (short name) means per dispatch and the derived shares."""
import csv
import glob
import re
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]


for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
rows = []
for k in acc:
    m = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    d = dur.get(k, [0.0])
    rows.append((sum(d), k, len(d), sum(d) / max(len(d), 1), m))
rows.sort(reverse=True)
print("%-70s %5s %8s | %6s %6s %6s(%5s) %6s | %6s | %9s %9s %9s | %8s" % ("kernel", "n", "us", "park%", "stall%", "issue%", "ldsS%", "mfma%", "valuA%", "VALU/wave", "LDS/wave", "MFMA/wave", "ldsConf%"))
for tot, k, n, us, m in rows[:40]:
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    busy = m.get("SQ_BUSY_CYCLES", 0.0)
    waves = 1.0
    mf_cycles = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0) or 1.0
    print("%-70s %5d %8.1f | %6.1f %6.1f %6.1f(%5.1f) %6.1f | %6.1f | %9.0f %9.0f %9.0f | %8.1f" % (
        k, n, us, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * m.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * mf_cycles / (gui * 1024.0 / 8.0 * 8.0) if gui > 1 else 0.0,
        100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_MFMA", 0),
        100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)))
