"""Copy the summaries of tools/final_profiles.sh (gpurun_out/final/) into profiles/ under a round tag and derive the
roofline-kernel HBM traffic JSON that bench.py reads:  python tools/final_profiles_collect.py r01z"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01z"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "final"), os.path.join(root, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    assert hits, pattern
    return hits[-1]          # gpurun merges into gpurun_out/: an earlier run's files may still be there


shutil.copy(one("fwd/**/*kernel_stats.csv"), os.path.join(dst, tag + "_fwd_cfg1_kernel_stats.csv"))
shutil.copy(one("train/**/*kernel_stats.csv"), os.path.join(dst, tag + "_train_cfg1_kernel_stats.csv"))
for name in ("bench_default", "bench_train", "bench_train_drop", "bench_cfg3"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p) and open(p).read().strip().startswith("{"):
        shutil.copy(p, os.path.join(dst, "%s_%s.json" % (tag, name)))
rec = {"kernel": "k_gemm_pw<192>", "workload": "B=48: z[b] = Wp(384x384) . g[b](384x1024)",
       "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md, HBM) -> hbm_bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE) * 1024"}
for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = one(sub + "/**/*counter_collection.csv")
    shutil.copy(f, os.path.join(dst, "%s_pmc_%s_roofline_kernel.csv" % (tag, "fetch" if ctr == "FETCH_SIZE" else "write")))
    vals, durs = [], []
    for r in csv.DictReader(open(f)):
        if "k_gemm_pw" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            vals.append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rec[ctr + "_KB_mean"] = sum(vals) / len(vals)
    rec[ctr + "_launches"] = len(vals)
    if durs:
        rec[ctr + "_pass_avg_us"] = sum(durs) / len(durs)
rec["hbm_bytes_per_launch"] = (2 * rec["FETCH_SIZE_KB_mean"] + rec["WRITE_SIZE_KB_mean"]) * 1024
rec["algorithmic_bytes_per_launch"] = 4.0 * (2 * 48 * 384 * 1024 + 384 * 384 + 384)
json.dump(rec, open(os.path.join(dst, tag + "_pmc_k_gemm_pw.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
for r in csv.DictReader(open(os.path.join(dst, tag + "_fwd_cfg1_kernel_stats.csv"))):
    if "k_gemm_pw" in r["Name"]:
        print("fwd profile k_gemm_pw:", r["Calls"], "calls avg", float(r["AverageNs"]) / 1e3, "us")
