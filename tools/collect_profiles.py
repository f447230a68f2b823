#!/usr/bin/env python3
"""Reduce / collect the evidence of tools/collect_profiles.sh.
  on the GPU box:   python tools/collect_profiles.py --reduce gpurun_out/prof     (per-dispatch PMC csv -> per-kernel json)
  in the repo:      python tools/collect_profiles.py r02a                          (gpurun_out/prof -> profiles/r02a_*)
HBM traffic per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: on gfx950 rocprofv3's FETCH_SIZE counts wide coalesced
reads at one half (MI355X_MICROARCH.md, HBM section); both counters are in KB and come from SEPARATE passes.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 * 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# rocprof kernel name -> the family names bench.py's `kernels` array uses (dpmn_profile tags)
FAMILIES = [("k_conv_igemm<128, 128", "k_conv_igemm<128,128>"), ("k_conv_igemm<64, 64", "k_conv_igemm<64,64>"),
            ("k_conv_igemm<128, 16", "k_conv_igemm<128,16|32>"), ("k_conv_igemm<128, 32", "k_conv_igemm<128,16|32>"),
            ("k_conv_igemm_x3<128, 128", "k_conv_igemm<128,128>"), ("k_conv_igemm_x3<64, 64", "k_conv_igemm<64,64>"), ("k_conv_igemm_x3<128, 64", "k_conv_igemm<128,64>"),
            ("k_conv_splitk_reduce", "k_conv_splitk_reduce"), ("k_conv_halo_c4", "k_conv_halo_c4"), ("k_conv_halo<", "k_conv_halo"), ("k_conv_halo_x3", "k_conv_halo"),
            ("k_conv_wgrad", "k_conv_wgrad"),
            ("k_gemm_pw", "k_gemm_pw"), ("k_gemm_wstat<96, 1", "k_gemm_wstat|rowreg<LN prologue>"), ("k_gemm_wstat<192, 1", "k_gemm_wstat|rowreg<LN prologue>"),
            ("k_gemm_rowreg<96, 1", "k_gemm_wstat|rowreg<LN prologue>"), ("k_gemm_rowreg<192, 1", "k_gemm_wstat|rowreg<LN prologue>"),
            ("k_gemm_wstat", "k_gemm_wstat|rowreg"), ("k_gemm_rowreg", "k_gemm_wstat|rowreg"), ("k_gemm_kloop", "k_gemm_kloop"),
            ("k_gemm_tn", "k_gemm_tn_reg"), ("k_affine_act_bwd", "k_affine_act_bwd"), ("k_ln_bwd", "k_ln_bwd"),
            ("k_window_attn_mfma<", "k_window_attn_mfma<4|8|16,32>"), ("k_conv_pack", "k_conv_pack_multi"), ("k_wgrad_unpack", "k_wgrad_unpack_multi"), ("k_dwconv_gelu", "k_dwconv_gelu"),
            ("k_window_attn8_mfma", "k_window_attn8_mfma"), ("k_window_attn<", "k_window_attn<2|4|16>"), ("k_ln_qkv_window_attn_bwd", "k_ln_qkv_window_attn_bwd"), ("k_ln_qkv_window_attn", "k_ln_qkv_window_attn"),
            ("k_bigru", "k_bigru"), ("k_mha32", "k_mha32")]


def family(name):
    name = name.replace("(anonymous namespace)::", "")
    for pat, fam in FAMILIES:
        if pat in name:
            return fam
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else None


def reduce_pmc(path):
    """{family: {counter: mean per dispatch, 'launches': n, 'avg_us': t}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp, dur = collections.defaultdict(set), collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[fam]:
            disp[fam].add(r["Dispatch_Id"])
            dur[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = {}
    for fam, c in per.items():
        n = len(disp[fam])
        out[fam] = {k: v / n for k, v in c.items()}
        out[fam]["launches"] = n
        out[fam]["avg_us"] = dur[fam] / n
    return out


def latest(base, pattern):
    hits = sorted(glob.glob(os.path.join(base, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def do_reduce(out):
    for sub in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_fetch_train", "pmc_write_train", "pmc_mfma_train", "pmc_fetch_x3", "pmc_write_x3", "pmc_mfma_x3",
                "pmc_fetch_train_x3", "pmc_write_train_x3", "pmc_mfma_train_x3"):
        f = latest(out, sub + "/**/*counter_collection.csv")
        if f:
            json.dump(reduce_pmc(f), open(os.path.join(out, sub + ".json"), "w"), indent=1)
            print("reduced", sub, "from", os.path.basename(f))


def do_collect(tag):
    src, dst = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
    for sub, name in (("fwd", "fwd_cfg1_kernel_stats.csv"), ("train", "train_cfg1_kernel_stats.csv"),
                      ("fwd_x3", "x3_fwd_cfg1_kernel_stats.csv"), ("train_x3", "x3_train_cfg1_kernel_stats.csv")):
        f = latest(src, sub + "/**/*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, name)))
    B = 48
    f = os.path.join(src, "pmc_pgrm_mfma_util.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_pmc_pgrm_mfma_util.csv" % tag))
    for name in ("bench_default", "bench_train", "bench_x3", "bench_train_x3", "bench_cfg4_x3", "bench_cfg4_train_x3", "bench_cfg3_x3", "bench_train_drop", "bench_train_nodrop", "bench_cfg3", "bench_cfg4", "bench_cfg4_train"):
        p = os.path.join(src, name + ".json")
        if os.path.exists(p) and open(p).read().strip().startswith("{"):
            shutil.copy(p, os.path.join(dst, "%s_%s.json" % (tag, name)))
            if name == "bench_default":
                B = json.load(open(p))["config"]["per_gpu_batch"]
    for suffix, mode, text in (("", "fwd", "cfg1 forward, bench.py --steps 3 --warmup 2 under rocprofv3 --pmc (one counter per pass)"),
                               ("_train", "train", "cfg1 TRAINING step, bench.py --mode train --steps 3 --warmup 2 under rocprofv3 --pmc (one counter group per "
                                "pass; branch and weight-gradient streams active, so kernels overlap: the per-launch durations are inflated, the byte counts are not)"),
                               ("_x3", "fwd_x3", "cfg1 forward in mode 2 (f32 via bf16x3), bench.py --dtype x3 --steps 3 --warmup 2 under rocprofv3 --pmc"),
                               ("_train_x3", "train_x3", "cfg1 TRAINING step in mode 2 (f32 via bf16x3), bench.py --dtype x3 --mode train --steps 3 --warmup 2 under "
                                "rocprofv3 --pmc (streams overlap: per-launch durations inflated, byte counts not)")):
        fe, wr, mf = (json.load(open(os.path.join(src, n + suffix + ".json"))) if os.path.exists(os.path.join(src, n + suffix + ".json")) else {}
                      for n in ("pmc_fetch", "pmc_write", "pmc_mfma"))
        pre = tag + {"fwd": "", "train": "_train", "fwd_x3": "_x3", "train_x3": "_x3_train"}[mode]
        if fe and wr:
            rec = {"per_gpu_batch": B, "mode": mode, "workload": text,
                   "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
                   "kernels": {}}
            for fam in sorted(set(fe) & set(wr)):
                rec["kernels"][fam] = {"FETCH_SIZE_KB": fe[fam].get("FETCH_SIZE"), "WRITE_SIZE_KB": wr[fam].get("WRITE_SIZE"),
                                       "launches": fe[fam]["launches"], "avg_us_in_pmc_pass": fe[fam]["avg_us"],
                                       "hbm_bytes_per_launch": (2 * fe[fam].get("FETCH_SIZE", 0.0) + wr[fam].get("WRITE_SIZE", 0.0)) * 1024}
            json.dump(rec, open(os.path.join(dst, pre + "_pmc_traffic.json"), "w"), indent=1)
        if mf:
            rows, tb, ta = [], 0.0, 0.0
            for fam, c in mf.items():
                busy, act, mops = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
                avail = act / 8.0 * 1024.0
                mops16 = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)  # same 512-flop unit (checked against the six-product count of k_conv_igemm_x3)
                rows.append((c["avg_us"] * c["launches"], fam, c["launches"], c["avg_us"], 100.0 * busy / avail if avail else 0.0,
                             mops * 512 / c["avg_us"] / 1e6 if c["avg_us"] else 0.0, mops16 * 512 / c["avg_us"] / 1e6 if c["avg_us"] else 0.0))
                tb += busy * c["launches"]; ta += avail * c["launches"]
            rows.sort(reverse=True)
            with open(os.path.join(dst, pre + "_pmc_mfma_util.csv"), "w") as o:
                o.write("kernel_family,launches,avg_us,mfma_busy_pct,mfma_f32_tflops,mfma_bf16_tflops\n")
                for _, fam, n, us, util, tf, tf16 in rows:
                    o.write('"%s",%d,%.2f,%.2f,%.2f,%.2f\n' % (fam, n, us, util, tf, tf16))
                o.write('"ALL dpmn kernels (time-weighted)",,,%.2f,,\n' % (100.0 * tb / ta if ta else 0.0))
            print(open(os.path.join(dst, pre + "_pmc_mfma_util.csv")).read())
    for name in ("pmc_sq_f32.txt", "pmc_sq_x3.txt", "torch_ops_per_step.txt", "gpu_suite.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
    rl = os.path.join(src, "rccl_world1.log")
    if os.path.exists(rl):
        shutil.copy(rl, os.path.join(dst, tag + "_rccl_world1.log"))
    pe = os.path.join(ROOT, "gpurun_out", "parity_errors.json")
    if os.path.exists(pe):
        shutil.copy(pe, os.path.join(dst, tag + "_parity_errors.json"))
    print("collected into profiles/%s_*" % tag)


if __name__ == "__main__":
    if sys.argv[1] == "--reduce":
        do_reduce(sys.argv[2])
    else:
        do_collect(sys.argv[1])
