#!/usr/bin/env python3
"""Phase timeline of k_ln_qkv_window_attn_bwd from a FAB_TIMING build (tools/build_variants.sh attn_fused_bwd "-DFAB_TIMING=1" fabtm):
DPMN_HIP_LIB=tools/variants/libdpmn_fabtm.so python tools/fab_timeline.py  -- s_memtime ticks (100 MHz constant clock -> 10 ns)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dpmn_amd import ops, _abi
from dpmn_amd.utils import synth
B, H, W, C = 48, 16, 64, 96
dev = torch.device("cuda:0")
u = lambda n, s, lo=-1, hi=1: synth.uniform(n, s, lo, hi, 3).to(dev)
tq, tkv, dout = u("tq", (B, H * W, C)), u("tkv", (B, H * W, C)), u("do", (B, H * W, C))
ln = [u("a", (C,), .5, 1.5), u("b", (C,)), u("c", (C,), .5, 1.5), u("d", (C,))]
wq, bq, wkv, bkv = u("wq", (C, C), -.1, .1), u("bq", (C,)), u("wkv", (2 * C, C), -.1, .1), u("bkv", (2 * C,))
tables = [u("t%d" % i, ((2 * w - 1) ** 2, 2)) for i, w in enumerate((2, 4, 8))]
fold = []
ops.ln_qkv_window_attn_train(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], [1, 2, 4], 2, H, W, save_qkv=False, fold_out=fold)
run = lambda: ops.ln_qkv_window_attn_bwd(tq, tkv, *ln, wq, bq, wkv, bkv, tables, [2, 4, 8], [1, 2, 4], 2, H, W, dout, fold=fold[0])
lib = ctypes.CDLL(_abi.LIB_PATH)
for _ in range(10):
    run()
torch.cuda.synchronize()
lib.dpmn_fab_timing_clear()
run()
torch.cuda.synchronize()
buf = np.zeros((512, 9, 8), dtype=np.uint64)
lib.dpmn_fab_timing_dump(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.astype(np.int64)
names = ["stats+proj", "sync(prev B)", "tiles->LDS+sync", "pass A", "sync", "pass B"]
T0 = min(int(x) for x in t.reshape(-1) if x > 0)
# per class of block (by unit count): median cycles of each phase, of a unit, of the staging, the loop and the flush; end times
import collections
cls = collections.defaultdict(list)
for blk in range(512):
    tb = t[blk]
    nu = int((tb[:8, 0] > 0).sum())
    if nu == 0 or tb[8][0] == 0:
        continue
    ph = [[int(tb[u_][k + 1]) - int(tb[u_][k]) for k in range(6)] for u_ in range(nu)]
    cls[nu].append(dict(ph=np.median(np.array(ph), axis=0), unit=np.median([int(tb[u_][6]) - int(tb[u_][0]) for u_ in range(nu)]),
                        stage=int(tb[8][1]) - int(tb[8][0]), loop=int(tb[8][2]) - int(tb[8][1]), flush=int(tb[8][3]) - int(tb[8][2]),
                        start=int(tb[8][0]) - T0, end=int(tb[8][3]) - T0))
for nu, rows in sorted(cls.items()):
    med = lambda k: float(np.median([r[k] for r in rows]))
    ph = np.median(np.array([r["ph"] for r in rows]), axis=0)
    print("blocks with %d units (%d blocks): unit %.0f cycles [%s]; staging %.0f, loop %.0f, flush %.0f; start %.0f .. end median %.0f max %.0f" % (
        nu, len(rows), med("unit"), "  ".join("%s %.0f" % (n, x) for n, x in zip(names, ph)), med("stage"), med("loop"), med("flush"), med("start"),
        med("end"), max(r["end"] for r in rows)))
