"""Per-kernel MFMA utilisation of one PGRM forward from a rocprofv3 PMC pass:
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES \
            --output-format csv -d OUT -- python tools/prof_pgrm.py 48
  python tools/pmc_pgrm_util.py OUT profiles/<tag>_pmc_pgrm_forward_mfma_util.csv
util = MFMA-busy cycles / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 * 1024 SIMDs); TFLOP/s = MOPS * 512 / time."""
import collections
import csv
import glob
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    per[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"],)
    if key not in seen:
        seen.add(key)
        cnt[n] += 1
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
nfwd = sum(v for k_, v in cnt.items() if k_.startswith("k_dwconv_gelu")) / 2.0 or 1.0          # two Mlp blocks per PGRM forward
rows, tot_busy, tot_avail, tot_ms = [], 0.0, 0.0, 0.0
for n, c in per.items():
    if not n.startswith("k_"):
        continue
    busy, act, mops = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
    avail = act / 8.0 * 1024.0
    us = dur[n] / cnt[n]
    rows.append((dur[n] / nfwd / 1e3, n, cnt[n], us, busy / cnt[n], act / cnt[n], mops / cnt[n], 100.0 * busy / avail if avail else 0.0,
                 mops / cnt[n] * 512 / us / 1e6))
    tot_busy += busy; tot_avail += avail; tot_ms += dur[n] / nfwd / 1e3
rows.sort(reverse=True)
with open(dst, "w") as o:
    o.write("k,launches,avg_us,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE_sum8xcd,SQ_INSTS_VALU_MFMA_MOPS_F32,mfma_util_pct,mfma_tflops,ms_per_pgrm_forward\n")
    for ms, n, c, us, busy, act, mops, util, tf in rows:
        o.write('"%s",%d,%.3f,%.1f,%.3f,%.1f,%.3f,%.3f,%.3f\n' % (n, c, us, busy, act, mops, util, tf, ms))
    o.write('"TOTAL (%d forwards)",,,,,,%.3f,,%.3f\n' % (nfwd, 100.0 * tot_busy / tot_avail, tot_ms))
print(open(dst).read())
