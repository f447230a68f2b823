#!/bin/bash
# how busy is the GPU inside one step?  kernel-trace of the forward (one batch at a time) / the training step, then per step:
# wall time, time with >= 1 kernel running, idle time, and the time-weighted mean number of kernels in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ovl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-profile --no-train"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/fwd -- $B --pipeline 1 --steps 6 --warmup 3 > $O/fwd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/train -- $B --mode train --steps 5 --warmup 3 > $O/train.log 2>&1
cd $R
python - <<'P'
import glob, csv, collections
for leg in ("fwd", "train"):
    f = glob.glob("gpurun_out/ovl/%s/**/*kernel_trace.csv" % leg, recursive=True)
    if not f: print(leg, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows)
    # inside the timed region: the last ~4 steps' worth of kernels (setup + warm-up come before)
    ev = ev[-(2000 if leg == "fwd" else 5600):]
    T0, T1 = ev[0][0], max(e[1] for e in ev)
    pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
    busy = idle = 0; depth = 0; last = T0; wsum = 0; hist = collections.Counter()
    for t, d in pts:
        dt = t - last
        if depth > 0: busy += dt; wsum += dt * depth
        else: idle += dt
        hist[min(depth, 6)] += dt
        depth += d; last = t
    tot = T1 - T0
    print("== %s: window %.1f ms, %d kernels: busy %.1f %%, idle %.1f %%, mean kernels in flight while busy %.2f" % (leg, tot / 1e6, len(ev), 100 * busy / tot, 100 * idle / tot, wsum / max(busy, 1)))
    print("   time share by kernels in flight: " + "  ".join("%d%s: %.1f %%" % (k, "+" if k == 6 else "", 100 * v / tot) for k, v in sorted(hist.items())))
    q = collections.Counter()
    for e in ev: q[e[3]] += e[1] - e[0]
    print("   kernel time per queue (ms): " + "  ".join("%s: %.1f" % (k, v / 1e6) for k, v in q.most_common(8)))
    # idle gaps: how many and how long
    gaps = []; depth = 0; last = None
    for t, d in pts:
        if depth == 0 and last is not None and t > last: gaps.append(t - last)
        depth += d
        if depth == 0: last = t
    gaps.sort(reverse=True)
    print("   idle gaps: %d, total %.2f ms, the 10 longest (us): %s" % (len(gaps), sum(gaps) / 1e6, [round(g / 1e3, 1) for g in gaps[:10]]))
P
rm -rf $O/fwd $O/train
