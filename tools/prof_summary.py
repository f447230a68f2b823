"""Summarise a rocprofv3 kernel trace: per-kernel ms/step and (optionally) per-grid breakdown.
usage: python tools/prof_summary.py <dir> <steps> [kernel-substring ...]"""
import sys, glob, re
import pandas as pd

d = pd.read_csv(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])
steps = float(sys.argv[2])
d['dur'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3


def nm(s):
    m = re.search(r'(k_\w+)(<[^>]*>)?', s)
    if m:
        return m.group(0)
    m = re.search(r'(FillFunctor|CUDAFunctor_add|direct_copy_kernel_cuda|rocclr\w+|CatArray\w+|index\w+|reduce_kernel|\w+Functor\w*)', s)
    return 'torch:' + m.group(0) if m else s[:50]


d['name'] = d.Kernel_Name.map(nm)
g = d.groupby('name').dur.agg(['count', 'sum', 'mean']).sort_values('sum', ascending=False)
g['count'] /= steps
g['ms_step'] = g['sum'] / steps / 1e3
print(g.drop(columns='sum').head(int(30)).to_string())
print('total ms/step', d.dur.sum() / steps / 1e3)
for k in sys.argv[3:]:
    x = d[d.Kernel_Name.str.contains(k, regex=False)].copy()
    x['grid'] = x.Grid_Size_X.astype(str) + ',' + x.Grid_Size_Y.astype(str) + ',' + x.Grid_Size_Z.astype(str)
    x['key'] = x.name + ' ' + x.grid
    gg = x.groupby('key').dur.agg(['count', 'mean', 'sum']).sort_values('sum', ascending=False)
    gg['count'] /= steps
    gg['ms_step'] = gg['sum'] / steps / 1e3
    print(k)
    print(gg.drop(columns='sum').head(14).to_string())
