"""Concurrency of the training step's kernels from a rocprofv3 kernel trace: eager (branch / weight-gradient streams) against the
hipGraph replay of the same step.
  rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --mode train [--graph] --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile
  python tools/graph_timeline.py OUT [n_last_kernels]
Prints, over the LAST n kernels of the trace (default: the last 40 %): hardware queues used, wall span, sum of kernel durations, and
how much of the span had 0 / 1 / 2 / 3+ kernels resident."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else int(len(rows) * 0.4)
rows = rows[-n:]
ev = []
queues = {}
tot = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    tot += e - s
    q = r.get("Queue_Id", "?")
    queues[q] = queues.get(q, 0) + 1
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
level, last, hist = 0, t0, {}
for t, d in ev:
    hist[min(level, 3)] = hist.get(min(level, 3), 0) + (t - last)
    level += d
    last = t
span = t1 - t0
print("kernels %d, hardware queues %s" % (len(rows), dict(sorted(queues.items(), key=lambda kv: -kv[1]))))
print("wall span %.2f ms, sum of kernel durations %.2f ms (average concurrency %.2f)" % (span / 1e6, tot / 1e6, tot / span))
for k in sorted(hist):
    print("  %s kernels resident: %5.1f %% of the span" % ("3+" if k == 3 else str(k), 100.0 * hist[k] / span))
