#!/usr/bin/env python3
"""Classified instruction stream of a gfx950 .s file: M mfma, v valu, s salu, L lds, G global/buffer, X scratch, W waitcnt,
B barrier, J branch; labels between bars.  python tools/isa_stream.py file.s [first_label] [n_chars]
(how the scheduling experiments of DESIGN.md were read: is the vector work inside the MFMA shadows or in front of them?)"""
import sys
lines = open(sys.argv[1]).read().split('\n')
first = sys.argv[2] if len(sys.argv) > 2 else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
on = first is None
seq = []
for l in lines:
    t = l.strip()
    if not on:
        if t.startswith(first + ':'):
            on = True
        else:
            continue
    if not t or t.startswith(';'):
        continue
    if t.startswith('.LBB'):
        seq.append('|' + t.split(':')[0][1:] + '|')
        continue
    if t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    c = ('M' if op.startswith('v_mfma') else 'v' if op.startswith('v_') else 'W' if op.startswith('s_waitcnt') else
         'B' if op.startswith('s_barrier') else 'J' if op.startswith(('s_cbranch', 's_branch')) else 's' if op.startswith('s_') else
         'L' if op.startswith('ds_') else 'G' if op.startswith(('global_', 'buffer_')) else 'X' if op.startswith('scratch_') else '?')
    seq.append(c)
    if op == 's_endpgm':
        break
print(''.join(seq)[:n])
