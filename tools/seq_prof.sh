#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/seq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fwd -- python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train --steps 2 --warmup 2 > $O/fwd.log 2>&1
cd $R
python - <<'P'
import glob, csv, re
f = glob.glob("gpurun_out/seq/fwd/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = []
for r in rows:
    m = re.search(r"(k_\w+|__amd_rocclr_\w+|elementwise\w*|at::native::\w+)", r["Kernel_Name"])
    names.append((m.group(1) if m else r["Kernel_Name"][:40], r.get("Stream_Id", r.get("Queue_Id", "?"))))
n = len(names) // 4
last = names[-n:]
print("launches in the last step:", n)
out, prev, cnt = [], None, 0
for nm in last:
    if nm == prev: cnt += 1
    else:
        if prev: out.append("%s[q%s]x%d" % (prev[0], prev[1], cnt))
        prev, cnt = nm, 1
out.append("%s[q%s]x%d" % (prev[0], prev[1], cnt))
print(" ".join(out))
P
rm -rf $O/fwd
