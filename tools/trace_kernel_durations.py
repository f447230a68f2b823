"""Durations of every launch of kernels whose name contains PATTERN, in launch order, from a rocprofv3 kernel trace directory.
usage: python tools/trace_kernel_durations.py DIR PATTERN [PATTERN ...]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for pat in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    print(pat, len(d), "launches; us:", " ".join("%.0f" % x for x in d))
