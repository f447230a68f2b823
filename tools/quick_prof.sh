#!/bin/bash
# kernel-trace summary of the training step and the forward (top families by time), plus the bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- $B --mode train --steps 5 --warmup 3 > $O/train.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd -- $B --steps 10 --warmup 3 > $O/fwd.log 2>&1
cd $R
python - <<'P'
import glob, csv, re, collections
for leg, steps in (("train", 8), ("fwd", 13)):
    f = glob.glob("gpurun_out/qprof/%s/**/*kernel_stats.csv" % leg, recursive=True)
    if not f: print(leg, "no stats"); continue
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("== %s: %.2f ms of kernels per step (%d launches per step)" % (leg, tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
    fam = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r["Name"]; m = re.search(r"(k_\w+)", n); key = m.group(1) if m else n[:50]
        if "igemm" in n or "wgrad<" in n or "k_gemm_rowreg" in n or "k_gemm_wstat" in n:
            t = re.search(r"(k_\w+<[^>(]{0,24})", n); key = t.group(1) if t else key
        fam[key][0] += int(r["Calls"]); fam[key][1] += float(r["TotalDurationNs"])
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:32]:
        print("  %-44s %6.1f calls/step %8.1f us/call %7.3f ms/step" % (k, c / steps, t / c / 1e3, t / steps / 1e6))
P
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python bench.py --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd', d['ms_per_step'], 'train', d['train']['ms_per_step'], 'drop', d['train']['with_dropout_0.1']['ms_per_step'])"
