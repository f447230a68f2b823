#!/bin/bash
# single-stream kernel-trace summary of the training step (kernels do not overlap: their durations are their own)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qprof1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DPMN_TRAIN_BRANCH_STREAMS=0 DPMN_WGRAD_STREAM=0 DPMN_BRANCH_STREAMS=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train --mode train --steps 5 --warmup 3 > $O/train.log 2>&1
cd $R
python - <<'P'
import glob, csv, re, collections
f = glob.glob("gpurun_out/qprof1/train/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
steps = 8
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== train single stream: %.2f ms of kernels per step (%d launches per step)" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
fam = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Name"]; m = re.search(r"(k_\w+)", n); key = m.group(1) if m else n[:40]
    if "igemm" in n or "wgrad<" in n or "k_gemm_rowreg" in n or "k_gemm_wstat" in n or "halo" in n:
        t = re.search(r"(k_\w+<[^>(]{0,26})", n); key = t.group(1) if t else key
    fam[key][0] += int(r["Calls"]); fam[key][1] += float(r["TotalDurationNs"])
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %-46s %6.1f calls/step %8.1f us/call %7.3f ms/step" % (k, c / steps, t / c / 1e3, t / steps / 1e6))
P
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
tail -1 $O/train.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('single-stream train step under the tracer', d['ms_per_step'])"
