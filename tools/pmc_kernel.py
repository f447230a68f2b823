#!/usr/bin/env python3
"""Per-kernel means of the counters of one rocprofv3 --pmc pass:  python tools/pmc_kernel.py OUT_DIR [kernel substring]"""
import collections, csv, glob, os, re, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if pat not in n:
        continue
    per[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in disp[n]:
        disp[n].add(r["Dispatch_Id"])
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, c in per.items():
    k = len(disp[n])
    print("%s: %d dispatches, %.1f us" % (n, k, dur[n] / k))
    for name, v in sorted(c.items()):
        print("    %-34s %.4g per dispatch" % (name, v / k))
