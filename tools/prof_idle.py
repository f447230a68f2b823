"""GPU idle time inside the timed steps of a rocprofv3 kernel trace: union of the kernel intervals (all streams) over the last
`steps` steps vs their span, and the largest gaps with the kernels on both sides.
usage: python tools/prof_idle.py <trace dir> <steps in the trace> [steps to analyse = 3]"""
import glob, re, sys
import pandas as pd

d = pd.read_csv(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]).sort_values('Start_Timestamp').reset_index(drop=True)
steps = int(sys.argv[2])
take = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = len(d) // steps * take if steps else len(d)      # the steps are equally long in launches once warmed up
x = d.iloc[-n:]
span = (x.End_Timestamp.max() - x.Start_Timestamp.min()) / 1e6
busy, cur_s, cur_e, gaps, prev = 0.0, None, None, [], None


def nm(s):
    m = re.search(r'(k_\w+)', s)
    return m.group(0) if m else s[:40]


for s, e, k in zip(x.Start_Timestamp, x.End_Timestamp, x.Kernel_Name):
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, prev, nm(k)))
        cur_s, cur_e = s, e
    prev = nm(k)
busy += cur_e - cur_s
print("span %.3f ms over %d launches (~%d steps): busy %.3f ms, idle %.3f ms = %.1f %%" % (span, n, take, busy / 1e6, span - busy / 1e6,
                                                                                      100 * (1 - busy / 1e6 / span)))
print("gaps > 5 us: %d, total %.3f ms" % (sum(g[0] > 5 for g in gaps), sum(g[0] for g in gaps if g[0] > 5) / 1e3))
for g in sorted(gaps, reverse=True)[:15]:
    print("  %8.1f us  after %-28s before %s" % g)
