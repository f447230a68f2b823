"""Run-length trace of a kernel's instruction classes (MFMA / global load / LDS / waits / barriers / branches) from hipcc -S
output: where do the loads and their s_waitcnt sit relative to the MFMA block?   usage: python tools/isa_shape.py file.s kernel_substring"""
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
i = s.index("\n", s.index(sys.argv[2] + "", i))
body = s[i:s.index("s_endpgm", i)]
out = []
for l in body.split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    m = l.split()[0]
    if m.endswith(":"):
        out.append(l.split(":")[0] + ":")
        continue
    if m.startswith("v_mfma"): c = "MFMA"
    elif m.startswith(("global_load", "buffer_load")): c = "GLOAD"
    elif m.startswith(("global_store", "buffer_store")): c = "GSTORE"
    elif m.startswith(("ds_read", "ds_load")): c = "DSR"
    elif m.startswith(("ds_write", "ds_store")): c = "DSW"
    elif m.startswith("s_waitcnt"): c = "WAIT " + l.split(None, 1)[1]
    elif m.startswith("s_barrier"): c = "BARRIER"
    elif m.startswith(("s_cbranch", "s_branch")): c = "BR " + l.split()[-1]
    elif m.startswith("v_"): c = "V"
    else: c = "S"
    out.append(c)
res, prev, n = [], None, 0
for c in out + [None]:
    if c == prev:
        n += 1
    else:
        if prev is not None:
            res.append("%s x%d" % (prev, n) if n > 1 else prev)
        prev, n = c, 1
print(" | ".join(res))
