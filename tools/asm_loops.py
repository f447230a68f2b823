#!/usr/bin/env python3
"""Loops of one kernel in /tmp/<stem>.s (written by tools/isa_stats.py): MFMA / scratch / barrier / waitcnt counts per loop body.
usage: python tools/asm_loops.py attn_fused_bwd <mangled-name-substring>"""
import re, sys
L = open("/tmp/%s.s" % sys.argv[1]).read().split("\n")
st = [i for i, l in enumerate(L) if sys.argv[2] in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l)][0]
en = [i for i, l in enumerate(L) if i > st and l.startswith(".Lfunc_end")][0]
body = L[st:en]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > 150:
        a = labels[m.group(1)]
        ch = body[a:i]
        c = lambda k: sum(1 for x in ch if k in x)
        print("lines %5d-%5d mfma %4d scratch_st %3d scratch_ld %3d barrier %2d vmcnt-waits %3d global_ld %3d global_st %3d" % (
            a, i, c("v_mfma"), c("scratch_store"), c("scratch_load"), c("s_barrier"), sum(1 for x in ch if "s_waitcnt" in x and "vmcnt" in x),
            c("global_load") + c("buffer_load"), c("global_store")))
